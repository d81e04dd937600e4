#!/bin/bash
# BASELINE configs[3] on ONE GPU through the drop-in programs (GIBBS_MODE=parallel unless set: `auto` takes the reference's chains here, ~24 min): Gibbs posterior (rsem-run-gibbs) on the 50 M-pair /
# 200 k-transcript input, 8 chains (the config's "-p 8"; on an 8-GPU node they are dealt one per GPU and meet in one
# RCCL reduce), pipeline settings BURNIN 200, NSAMPLES 1000, GAP 1.  Also times rsem-run-em on the binary hand-off
# (imdName.rsb/, what rsem-parse-alignments --binary writes) for the end-to-end table.
#   GPU box, repo root:   tools/e2e_c4.sh [n_pairs_alignable=50000000] [M=200000] [chains=8] > gpurun_out/e2e_c4.log
N1=${1:-50000000}; M=${2:-200000}; P=${3:-8}; ISO=${ISO:-5-16}
NF=$(( N1 * 20 / 19 ))
export RSEM_HIP_TIMING=1
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
D=/tmp/e2e_c4
rm -rf $D
t=$(now); tools/bin/gen_temp $D $NF $M 3 20250925 100 nosam $ISO | tail -1; echo "gen_s $(el $t)"
t=$(now); tools/bin/temp_to_rsb $D/temp/s $D/stat/s 3; echo "to_rsb_s $(el $t) (not part of any timing below: the parser writes this directly)"
rm -f $D/temp/s.dat $D/temp/*.fq; du -sh $D/temp/s.rsb | cut -f1
echo "== rsem-run-em on the binary hand-off, --gibbs-out as arrays too (RSEM_HIP_BINARY=1: imdName.ofb/, no .ofg text)"; t=$(now)
RSEM_HIP_BINARY=1 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 --gibbs-out > $D/em.log 2>&1; echo "em_rsb_rc $? em_rsb_s $(el $t)"
grep -E "^\[timing\]" $D/em.log; grep ROUND $D/em.log | tail -1; du -sh $D/temp/s.ofb | cut -f1
echo "== rsem-run-gibbs 200 1000 1 -p $P, --gibbs-mode ${GIBBS_MODE:-parallel} (auto: the reference's chains unless they would take > 30 min)"; t=$(now)
rsem_amd/bin/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P --seed 1 --gibbs-mode ${GIBBS_MODE:-parallel} > $D/gibbs.log 2>&1; echo "gibbs_rc $? gibbs_s $(el $t)"
tail -4 $D/gibbs.log; cat $D/stat/s.gibbs_sampler
python - <<PY
import numpy as np
rows = [l.split("\t") for l in open("$D/temp/s.iso_res").read().strip().split("\n")]
print("iso_res rows", len(rows))
em = np.array(rows[4], float); pme = np.array(rows[-5], float); sd = np.sqrt(np.array(rows[-4], float))
big = em > 50
print("posterior mean counts vs EM expected counts on %d transcripts with > 50 reads: median |diff|/em %.4f, corr %.6f; sum pme %.1f" % (
    big.sum(), np.median(np.abs(pme[big] - em[big]) / em[big]), np.corrcoef(pme, em)[0, 1], pme.sum()))
import os
print("countvector files:", sorted(f for f in os.listdir("$D/temp") if "countvectors" in f))
PY
rm -rf $D
