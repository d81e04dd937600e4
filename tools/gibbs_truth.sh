#!/bin/bash
# Which sampler is closer to the posterior, and is the exact mode the reference's chain at scale?   (GPU box, repo root)
#   tools/gibbs_truth.sh [n_reads] [M] [P]
# "Truth" = long REFERENCE collapsed chains (rsem-run-gibbs -p 16, BURNIN 2000, 4000 samples; two seeds), run in the
# background on the host cores together with the reference at the pipeline's settings (200 / 1000 / 1, -p P), while the
# GPU runs the drop-in: exact mode (count vectors must be byte-equal to the reference's with the same seed), exact mode
# with ONE chain of the same length (the concurrency check: P chains should cost about what one does), and the
# data-augmentation sampler with 8 and 1 sweeps per round.
N=${1:-1000000}; M=${2:-20000}; P=${3:-64}; D=/tmp/e2egt_$N
rm -rf $D $D.*; tools/bin/gen_temp $D $N $M 1 | tail -1
rsem_amd/bin/rsem-run-em $D/ref 1 $D/s $D/temp/s $D/stat/s --gibbs-out -q > /dev/null
rm -f $D/temp/*.fq $D/temp/s.dat   # only .ofg / .model / results are needed from here on
run() { # name program burnin nsamples p seed extra...   (each run in its own copy: the programs append to iso_res in place)
  local name=$1 prog=$2 b=$3 n=$4 p=$5 seed=$6; shift 6
  rm -rf $D.$name; cp -r $D $D.$name
  local t0=$(date +%s.%N)
  timeout 600 $prog $D.$name/ref $D.$name/temp/s $D.$name/stat/s $b $n 1 -p $p --seed $seed -q "$@" 2> $D.$name.err
  local rc=$?
  echo "$name rc=$rc wall=$(awk -v a=$t0 -v b=$(date +%s.%N) 'BEGIN{printf "%.2f", b-a}') s  ($(basename $prog) $b $n 1 -p $p --seed $seed $*)"
}
run truth oracle/_ref/rsem-run-gibbs 2000 4000 16 11 &
run truth2 oracle/_ref/rsem-run-gibbs 2000 4000 16 12 &
run ref oracle/_ref/rsem-run-gibbs 200 1000 $P 5 &
run exact rsem_amd/bin/rsem-run-gibbs 200 1000 $P 5 --gibbs-mode exact
run exact1 rsem_amd/bin/rsem-run-gibbs 200 $(( (1000 + P - 1) / P )) 1 5 --gibbs-mode exact
run new rsem_amd/bin/rsem-run-gibbs 200 1000 $P 5 --gibbs-mode parallel
run new1 rsem_amd/bin/rsem-run-gibbs 200 1000 $P 5 --gibbs-mode parallel --gibbs-thin 1
wait
same=0; diff=0
for k in $(seq 0 $((P - 1))); do
  if cmp -s $D.ref/temp/s.countvectors$k $D.exact/temp/s.countvectors$k; then same=$((same + 1)); else diff=$((diff + 1)); fi
done
echo "exact mode vs reference (same seed, -p $P): $same count-vector files byte-equal, $diff differ"
cmp -s $D.ref/temp/s.iso_res $D.exact/temp/s.iso_res && echo "exact mode vs reference: iso_res byte-equal" || echo "exact mode vs reference: iso_res differ (the pme_TPM/FPKM sums are floating point: compared below)"
python - <<PY
import numpy as np
def pm(name):
    r = [l.split("\t") for l in open("$D.%s/temp/s.iso_res" % name).read().strip().split("\n")]
    return np.array(r[-5], float), np.array(r[-4], float)
T, sT = pm("truth"); T2, _ = pm("truth2")
def cmp(name, what):
    x, _ = pm(name)
    q = np.abs(x - T) / (sT + 0.5)
    q2 = np.abs(x - T2) / (sT + 0.5)
    print("%-40s vs long chains: |diff|/(sd+0.5) median %.4f 99%% %.4f max %.4f rms %.4f corr %.6f | vs 2nd long run: max %.4f rms %.4f" % (
        what, np.median(q), np.percentile(q, 99), q.max(), np.sqrt((q**2).mean()), np.corrcoef(x, T)[0, 1], q2.max(), np.sqrt((q2**2).mean())))
cmp("truth2", "second set of long chains")
cmp("ref", "reference 200/1000 -p $P")
cmp("exact", "drop-in exact 200/1000 -p $P")
cmp("new", "drop-in parallel thin 8")
cmp("new1", "drop-in parallel thin 1")
PY
rm -rf $D $D.*
