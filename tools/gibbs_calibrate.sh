#!/bin/bash
# How much do two REFERENCE Gibbs runs (different seeds) differ, and where does the drop-in's parallel sampler fall?
# usage: tools/gibbs_calibrate.sh <n_reads> <M> <threads>
N=${1:-1000000}; M=${2:-20000}; P=${3:-64}; D=/tmp/e2egc_$N
rm -rf $D; tools/bin/gen_temp $D $N $M 1 | tail -1
rsem_amd/bin/rsem-run-em $D/ref 1 $D/s $D/temp/s $D/stat/s --gibbs-out -q > /dev/null
cp $D/temp/s.iso_res $D/iso_res.pre; cp $D/temp/s.gene_res $D/gene_res.pre
run() { # name program seed extra
  cp $D/iso_res.pre $D/temp/s.iso_res; cp $D/gene_res.pre $D/temp/s.gene_res
  ( time $2 $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P --seed $3 -q $4 $5 ) 2>&1 | grep real | sed "s/^/$1 /"
  cp $D/temp/s.iso_res $D/iso_res.$1
}
run refA oracle/_ref/rsem-run-gibbs 5
run refB oracle/_ref/rsem-run-gibbs 77
run newP rsem_amd/bin/rsem-run-gibbs 5 --gibbs-mode parallel
run newP2 rsem_amd/bin/rsem-run-gibbs 99 --gibbs-mode parallel
python - <<PY
import numpy as np
def pm(name):
    r = [l.split("\t") for l in open("$D/iso_res.%s" % name).read().strip().split("\n")]
    return np.array(r[-5], float), np.array(r[-4], float)
A, sA = pm("refA"); B, sB = pm("refB"); P, sP = pm("newP"); P2, sP2 = pm("newP2")
def cmp(x, y, s, what):
    d = np.abs(x - y)
    scale = s + 0.5
    q = d / scale
    print("%-22s |diff|/(posterior sd + 0.5): median %.4f  99%% %.4f  max %.4f ; corr %.8f" % (what, np.median(q), np.percentile(q, 99), q.max(), np.corrcoef(x, y)[0, 1]))
cmp(A, B, sA, "ref(seed 5) vs ref(77)")
cmp(P, A, sA, "drop-in vs ref(5)")
cmp(P, B, sA, "drop-in vs ref(77)")
cmp(P, P2, sA, "drop-in(5) vs drop-in(99)")
print("posterior sd ratio drop-in/ref: median %.3f" % np.median(sP[sA > 1] / sA[sA > 1]))
PY
rm -rf $D
