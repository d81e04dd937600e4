#!/bin/bash
# Round 4, call I: configs[2] through the program with per-round times of the model rounds, with and without the helper thread
# that gives the parsed inputs back during the device loop.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04i; mkdir -p $out
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
export RSEM_HIP_TIMING=1
DF=/tmp/c3full; rm -rf $DF
tools/bin/gen_temp $DF 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in release norelease release2 norelease2; do
  unset RSEM_HIP_NO_RELEASE; case $v in norelease*) export RSEM_HIP_NO_RELEASE=1;; esac
  t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/$v.log 2>&1; echo "$v rc $? wall $(el $t) s"
  grep -E "^\[timing\]" $out/$v.log | grep -v "model round" | tr '\n' ';'; echo
  grep -E "model round" $out/$v.log | awk '{printf "%s ", $(NF-1)}'; echo
  grep -v "^ROUND" $out/$v.log > $out/tmp; mv $out/tmp $out/$v.log
done
rm -rf $DF
