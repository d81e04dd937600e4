#!/bin/bash
# Which sampler is closer to the posterior?  (NOT yet run to completion: the first attempt also ran the drop-in's exact
# mode with 64 chains -- one wave per chain, sequential on one GPU -- and spent the round's GPU budget; that leg is gone.)
#  "Truth" = one long reference chain (-p 1, BURNIN 2000, 4000 samples);
# then the reference and the drop-in with the pipeline's settings (200 / 1000 / 1, -p P).
N=${1:-200000}; M=${2:-4000}; P=${3:-64}; D=/tmp/e2egt_$N
rm -rf $D; tools/bin/gen_temp $D $N $M 1 | tail -1
rsem_amd/bin/rsem-run-em $D/ref 1 $D/s $D/temp/s $D/stat/s --gibbs-out -q > /dev/null
cp $D/temp/s.iso_res $D/iso_res.pre; cp $D/temp/s.gene_res $D/gene_res.pre
run() { # name program burnin nsamples p seed extra...
  cp $D/iso_res.pre $D/temp/s.iso_res; cp $D/gene_res.pre $D/temp/s.gene_res
  local name=$1 prog=$2 b=$3 n=$4 p=$5 seed=$6; shift 6
  ( time timeout 300 $prog $D/ref $D/temp/s $D/stat/s $b $n 1 -p $p --seed $seed -q "$@" ) 2>&1 | grep real | sed "s/^/$name /"
  cp $D/temp/s.iso_res $D/iso_res.$name
}
run truth oracle/_ref/rsem-run-gibbs 2000 4000 1 11
run truth2 oracle/_ref/rsem-run-gibbs 2000 4000 1 12
run ref oracle/_ref/rsem-run-gibbs 200 1000 $P 5
run new rsem_amd/bin/rsem-run-gibbs 200 1000 $P 5 --gibbs-mode parallel
run new1 rsem_amd/bin/rsem-run-gibbs 200 1000 $P 5 --gibbs-mode parallel --gibbs-thin 1
python - <<PY
import numpy as np
def pm(name):
    r = [l.split("\t") for l in open("$D/iso_res.%s" % name).read().strip().split("\n")]
    return np.array(r[-5], float), np.array(r[-4], float)
T, sT = pm("truth"); T2, _ = pm("truth2")
def cmp(name, what):
    x, _ = pm(name)
    q = np.abs(x - T) / (sT + 0.5)
    print("%-34s vs long chain: |diff|/(sd+0.5) median %.4f 99%% %.4f max %.4f; rms %.4f" % (what, np.median(q), np.percentile(q, 99), q.max(), np.sqrt((q**2).mean())))
cmp("truth2", "second long chain")
cmp("ref", "reference 200/1000 -p $P")
cmp("new", "drop-in parallel thin 8")
cmp("new1", "drop-in parallel thin 1")
PY
rm -rf $D
