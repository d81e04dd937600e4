#!/bin/bash
# rsem-calculate-credibility-intervals end to end, drop-in vs reference binary, on a generated data set:
#   gen_temp -> rsem-run-em --gibbs-out -> rsem-run-gibbs (count vectors) -> CI (both programs on the same vectors)
# usage: tools/e2e_ci.sh <n_reads> <M> <threads> [ref]
N=${1:-1000000}; M=${2:-20000}; P=${3:-64}; D=/tmp/e2eci_$N
rm -rf $D; tools/bin/gen_temp $D $N $M 1 | tail -1
rsem_amd/bin/rsem-run-em $D/ref 1 $D/s $D/temp/s $D/stat/s --gibbs-out -q > /dev/null
( time rsem_amd/bin/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P --seed 5 -q ) 2>&1 | grep real | sed 's/^/gibbs (drop-in, parallel mode) /'
cp $D/temp/s.iso_res $D/iso_res.pre; cp $D/temp/s.gene_res $D/gene_res.pre
echo "== drop-in CI"; ( time rsem_amd/bin/rsem-calculate-credibility-intervals $D/ref $D/temp/s $D/stat/s 0.95 1000 50 1024 -p $P --seed 7 ) 2>&1 | grep -E "device|real"
cp $D/temp/s.iso_res $D/iso_res.new; cp $D/temp/s.gene_res $D/gene_res.new
if [ "$4" == "ref" ]; then
  cp $D/iso_res.pre $D/temp/s.iso_res; cp $D/gene_res.pre $D/temp/s.gene_res
  echo "== reference CI (-p $P)"; ( time oracle/_ref/rsem-calculate-credibility-intervals $D/ref $D/temp/s $D/stat/s 0.95 1000 50 1024 -p $P --seed 7 -q ) 2>&1 | grep real
  python - <<PY
import numpy as np
def rows(p): return [np.array(l.split("\t"), float) for l in open(p).read().strip().split("\n")[-6:]]
for f in ("iso_res", "gene_res"):
    a, b = rows("$D/%s.new" % f), rows("$D/temp/s.%s" % f)
    for k in (0, 3):
        w = b[k + 1] - b[k]
        tol = 0.1 * w + 1e-3 * np.abs(b[k + 1]) + 1e-6
        print(f, "rows", k, k + 1, "max |diff|/tol: lb %.3f ub %.3f; cqv max abs diff %.4f; n=%d" % (
            np.max(np.abs(a[k] - b[k]) / tol), np.max(np.abs(a[k + 1] - b[k + 1]) / tol), np.max(np.abs(a[k + 2] - b[k + 2])), len(w)))
PY
fi
rm -rf $D
