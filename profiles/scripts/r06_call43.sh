# Round 6, call 43: the host stages' thread counts following the cgroup's cores (16 on the boxes) instead of the hardware threads: configs[2] at full size
# and the 5 % input, with and without (RSEM_HIP_IGNORE_CGROUP).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06aq; mkdir -p $out
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in cgroup hw cgroup hw; do
  if [ $v = hw ]; then export RSEM_HIP_IGNORE_CGROUP=1; else unset RSEM_HIP_IGNORE_CGROUP; fi
  ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/full_$v.log 2>&1
  echo "$v: $(grep -E 'refs \+|parse read|estimateFrom|contexts \+|device loop|main\(\) total' $out/full_$v.log | sed 's/\[timing\] //' | tr -s ' ' | tr '\n' ';') $(grep real $out/full_$v.log)"
done
rm -rf $D
D=/tmp/c3_5pct; rm -rf $D
tools/bin/gen_temp $D 2631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in cgroup hw cgroup hw; do
  if [ $v = hw ]; then export RSEM_HIP_IGNORE_CGROUP=1; else unset RSEM_HIP_IGNORE_CGROUP; fi
  ( time rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/small_$v.log 2>&1; echo "5 % $v: $(grep real $out/small_$v.log)"
done
rm -rf $D
