#!/bin/bash
# End-to-end wall clock on BASELINE configs[2] scale (PairedEndQModel, 50 M alignable read pairs, 200 k transcripts):
# the drop-in rsem-run-em at FULL size, and the reference rsem-run-em (oracle/_ref, -p REF_P) on a 1/SUB-size input from
# the same generator (same transcriptome size and gene structure), run to convergence, alone on the host.  The
# reference's full-size time is then extrapolated: everything it does is linear in the number of reads / alignments
# (parsing, rounds 1-11, every later round; EM.cpp:97-174,199-236); the number of rounds is taken from the drop-in's
# full-size run (the two programs stop at the same ROUND on the same input -- checked here at the 1/SUB size).
#   GPU box, repo root:   tools/e2e_c3.sh [n_pairs_alignable=50000000] [M=200000] [SUB=10] > gpurun_out/e2e_c3.log
N1=${1:-50000000}; M=${2:-200000}; SUB=${3:-10}; P=${REF_P:-64}; ISO=${ISO:-5-16}
NF=$(( N1 * 20 / 19 )); NS=$(( NF / SUB ))
export RSEM_HIP_TIMING=1
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
DS=/tmp/e2e_c3_sub; DF=/tmp/e2e_c3_full
rm -rf $DS $DF
df -h /tmp | tail -1
echo "== generate 1/$SUB size"; t=$(now); tools/bin/gen_temp $DS $NS $M 3 20250925 100 nosam $ISO | tail -1; echo "gen_sub_s $(el $t)"
oracle/_ref/rsem-build-read-index 32 1 1 $DS/temp/s_alignable_1.fq $DS/temp/s_alignable_2.fq > /dev/null
echo "== drop-in, 1/$SUB size"; t=$(now)
rsem_amd/bin/rsem-run-em $DS/ref 3 $DS/s $DS/temp/s $DS/stat/s -p $P > $DS/new.log 2>&1; echo "new_sub_rc $? new_sub_s $(el $t)"
grep -E "^\[timing\]" $DS/new.log; grep ROUND $DS/new.log | tail -1
cp $DS/stat/s.theta $DS/new.theta
echo "== generate full size"; t=$(now); tools/bin/gen_temp $DF $NF $M 3 20250925 100 nosam $ISO | tail -1; echo "gen_full_s $(el $t)"; du -sh $DF | cut -f1
echo "== drop-in, full size"; t=$(now)
rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p $P > $DF/new.log 2>&1; echo "new_full_rc $? new_full_s $(el $t)"
grep -E "^\[timing\]" $DF/new.log; grep ROUND $DF/new.log | tail -1
python - <<PY
print("theta_sum_full %.12f" % sum(float(x) for x in open("$DF/stat/s.theta").read().split("\n")[1].split()))
PY
cp $DF/stat/s.theta $DS/full_text.theta
echo "== drop-in, full size, binary hand-off (imdName.rsb/ = what rsem-parse-alignments --binary writes; conversion not timed)"
t=$(now); tools/bin/temp_to_rsb $DF/temp/s $DF/stat/s 3; echo "to_rsb_s $(el $t)"
rm -f $DF/temp/s.dat $DF/temp/*.fq
t=$(now)
rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p $P > $DF/new_rsb.log 2>&1; echo "new_full_rsb_rc $? new_full_rsb_s $(el $t)"
grep -E "^\[timing\]" $DF/new_rsb.log; grep ROUND $DF/new_rsb.log | tail -1
python - <<PY
import numpy as np
a=np.array(open("$DF/stat/s.theta").read().split("\n")[1].split(),float); b=np.array(open("$DS/full_text.theta").read().split("\n")[1].split(),float)
m=b>=1e-7
print("theta_max_rel_diff_rsb_vs_text_full %.3g" % np.max(np.abs(a[m]-b[m])/b[m]))
PY
rm -rf $DF
echo "== reference -p $P, 1/$SUB size (alone on the host)"
( t=$(now); oracle/_ref/rsem-run-em $DS/ref 3 $DS/s $DS/temp/s $DS/stat/s -p $P > $DS/ref.log 2>&1; echo "ref_sub_rc $? ref_sub_s $(el $t)" > $DS/ref.time ) &
REFPID=$!
# arrival times of the reference's ROUND lines: startup, rounds 1-11, later rounds
( t0=$(now); while kill -0 $REFPID 2>/dev/null; do r=$(grep -c "^ROUND" $DS/ref.log 2>/dev/null); echo "$(el $t0) $r"; sleep 0.5; done ) > $DS/ref.progress
wait $REFPID; cat $DS/ref.time; grep ROUND $DS/ref.log | tail -1; grep "Time Used" $DS/ref.log
python - <<PY
import numpy as np
a=[np.array(l.split(),float) for l in open("$DS/new.theta").read().split("\n")[1:3]]
b=[np.array(l.split(),float) for l in open("$DS/stat/s.theta").read().split("\n")[1:3]]
m=b[0]>=1e-7
print("theta_max_rel_diff_sub %.3g" % np.max(np.abs(a[0][m]-b[0][m])/b[0][m]))
pr=[l.split() for l in open("$DS/ref.progress") if len(l.split()) == 2]
t1=next((float(t) for t,r in pr if int(r)>=1), None); t11=next((float(t) for t,r in pr if int(r)>=11), None)
tl=float(pr[-1][0]); rl=int(pr[-1][1])
print("ref_sub_startup_s %.1f ref_sub_rounds1_11_s %.1f ref_sub_late_s %.1f ref_sub_late_rounds %d" % (t1, t11-t1, tl-t11, rl-11))
PY
rm -rf $DS $DF
