#!/bin/bash
# The UNMODIFIED reference rsem-run-em on BASELINE configs[2] AS NAMED (50 M read pairs, 560 M alignments, 200 k transcripts, the
# files bench.py generates for its full-size run), ALONE on the host and pinned the way bench.py pins it (-p 64 on the 64 physical
# cores of one socket: its fastest setting, profiles/r04p_ref_threads_probe.json), then the drop-in on the same files.  Round 4
# measured the reference once beside this repo's GPU jobs (1 676.5 s; 1 551 s estimated undisturbed): this is the undisturbed run.
#   GPU box, repo root:   tools/ref_full_size_alone.sh  > gpurun_out/r05u_call.log      (about 27 minutes, nearly all of it the reference)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-r05u}; mkdir -p $O
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
N1=${N1:-50000000}; M=200000; NF=$(( N1 * 20 / 19 )); P=64; D=/tmp/c3full
CORES=$(python - <<PY
import sys
sys.path.insert(0, ".")
import bench
c = bench.one_socket_cores(64)
print(",".join(map(str, c)) if c else "")
PY
)
rm -rf $D
t=$(now); tools/bin/gen_temp $D $NF $M 3 20250925 100 nosam 5-16 | tail -1; echo "gen_s $(el $t)"
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
echo "== reference -p $P on cpus ${CORES:-unpinned}, alone"
t=$(now); T0=$(date +%s.%N)
( if [ -n "$CORES" ]; then taskset -c $CORES oracle/_ref/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p $P; else oracle/_ref/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p $P; fi ) 2>&1 |
  while IFS= read -r line; do printf "%s %s\n" "$(awk -v a=$T0 -v b=$(date +%s.%N) 'BEGIN{printf "%.2f", b-a}')" "$line"; done > /tmp/ref_full_stamped.log
echo "reference_s $(el $t)"
grep -c " ROUND = " /tmp/ref_full_stamped.log | sed 's/^/reference_rounds /'
grep " ROUND = " /tmp/ref_full_stamped.log | awk 'NR==1{print "first ROUND line at", $1} NR==12{print "round 12 at", $1} {last=$1; n=NR} END{print "last ROUND line at", last, "of", n}'
grep " ROUND = " /tmp/ref_full_stamped.log | awk '{print $1, $4}' | sed 's/,//' | awk 'NR%50==1' > $O/reference_full_size_alone_round_arrivals.txt
grep -v " ROUND = " /tmp/ref_full_stamped.log | tail -5
cp $D/stat/s.theta /tmp/ref_alone.theta
export RSEM_HIP_TIMING=1
echo "== drop-in on the same files"
t=$(now); rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p $P > $O/dropin_full.log 2>&1; echo "dropin_rc $? dropin_s $(el $t)"
grep -E "^\[timing\]" $O/dropin_full.log | tr '\n' ';'; echo; grep "^ROUND" $O/dropin_full.log | tail -1
grep -v "^ROUND" $O/dropin_full.log > $O/dropin_full.tmp; mv $O/dropin_full.tmp $O/dropin_full.log
python - <<PY
import numpy as np, gzip
def theta(p):
    l = open(p).read().split("\n")
    return np.array(l[1].split(), float)
a, b = theta("/tmp/ref_alone.theta"), theta("$D/stat/s.theta")
big = a >= 1e-7
print("theta: drop-in vs this reference run, max rel diff over %d entries >= 1e-7: %.3g" % (big.sum(), np.max(np.abs(a - b)[big] / a[big])))
try:
    old = np.array(gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1].split(), float)
    print("theta: this reference run vs round 4's reference run: max abs diff %.3g" % np.abs(a - old).max())
except Exception as e:
    print("(no round-4 theta to compare with: %s)" % e)
PY
rm -rf $D
