#!/bin/bash
# rsem-run-gibbs end to end, drop-in (parallel sampler) vs reference binary, on a generated data set
# usage: tools/e2e_gibbs.sh <n_reads> <M> <threads> [ref]
N=${1:-10000000}; M=${2:-50000}; P=${3:-64}; D=/tmp/e2eg_$N
rm -rf $D; tools/bin/gen_temp $D $N $M 1 | tail -1
rsem_amd/bin/rsem-run-em $D/ref 1 $D/s $D/temp/s $D/stat/s --gibbs-out -q > /dev/null
ls -la $D/temp/s.ofg | awk '{print ".ofg bytes", $5}'
cp $D/temp/s.iso_res $D/iso_res.pre; cp $D/temp/s.gene_res $D/gene_res.pre
echo "== drop-in Gibbs (-p $P)"; ( time rsem_amd/bin/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P --seed 5 ) 2>&1 | grep -E "sampler|real|sweeps"
cp $D/temp/s.iso_res $D/iso_res.new
if [ "$4" == "ref" ]; then
  cp $D/iso_res.pre $D/temp/s.iso_res; cp $D/gene_res.pre $D/temp/s.gene_res
  echo "== reference Gibbs (-p $P)"; ( time oracle/_ref/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P --seed 5 -q ) 2>&1 | grep real
  python - <<PY
import numpy as np
def rows(p): return [l.split("\t") for l in open(p).read().strip().split("\n")]
a, b = rows("$D/iso_res.new"), rows("$D/temp/s.iso_res")
# appended rows: pme_c, sd, pme_TPM, pme_FPKM, IsoPct_pme (WriteResults.h:407-476)
pa, sa = np.array(a[-5], float), np.array(a[-4], float)
pb, sb = np.array(b[-5], float), np.array(b[-4], float)
em = np.array(b[4], float)  # expected_count row of the EM result
se = np.sqrt(sa**2 + sb**2) / np.sqrt(1000.0) * 3 + 0.02 * np.maximum(pb, 1.0) + 0.5
z = np.abs(pa - pb) / se
print("posterior mean counts: %d transcripts, max |diff|/tol %.2f, fraction within tol %.4f, corr %.8f; sum %.1f vs %.1f" % (
    len(pa), z.max(), (z <= 1).mean(), np.corrcoef(pa, pb)[0, 1], pa.sum(), pb.sum()))
print("posterior sd: median ratio drop-in/reference %.3f" % np.median((sa[sb > 1] / sb[sb > 1])))
PY
fi
rm -rf $D
