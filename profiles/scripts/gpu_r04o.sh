#!/bin/bash
# Round 4, call O: configs[2] through the final program, twice (text inputs), theta against the reference's own; then the binary hand-off.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04o; mkdir -p $out
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
export RSEM_HIP_TIMING=1
DF=/tmp/c3full; rm -rf $DF
tools/bin/gen_temp $DF 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in 1 2; do
  t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/text_$v.log 2>&1; echo "text run $v rc $? wall $(el $t) s"
  grep -E "^\[timing\]" $out/text_$v.log | grep -v "model round " | tr '\n' ';'; echo
  grep -E "model round " $out/text_$v.log | awk '{printf "%s ", $(NF-1)}'; echo
  grep ROUND $out/text_$v.log | tail -1
  grep -v "^ROUND" $out/text_$v.log > $out/tmp; mv $out/tmp $out/text_$v.log
done
python - $DF/stat/s.theta <<'PY'
import gzip, sys, numpy as np
a = [np.array(l.split(), float) for l in open(sys.argv[1]).read().split("\n")[1:3]]
b = [np.array(l.split(), float) for l in gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1:3]]
m = b[0] >= 1e-7
print("full size: theta vs the REFERENCE's own (round 4 call A): max rel diff %.3g (polished %.3g)" % (np.max(np.abs(a[0][m] - b[0][m]) / b[0][m]), np.max(np.abs(a[1][b[1] >= 1e-7] - b[1][b[1] >= 1e-7]) / b[1][b[1] >= 1e-7])))
PY
tools/bin/temp_to_rsb $DF/temp/s $DF/stat/s 3 > /dev/null
rm -f $DF/temp/s.dat $DF/temp/*.fq
t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/rsb.log 2>&1; echo "binary hand-off rc $? wall $(el $t) s"
grep -E "^\[timing\]" $out/rsb.log | grep -v "model round " | tr '\n' ';'; echo
grep -v "^ROUND" $out/rsb.log > $out/tmp; mv $out/tmp $out/rsb.log
rm -rf $DF
