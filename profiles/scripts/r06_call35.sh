# Round 6, call 35: the device loop of the program is 5.8-5.9 s where its kernels take 5.1 s: with and without the helper thread that gives the
# parsed inputs back during the loop (RSEM_HIP_NO_RELEASE), configs[2] at full size.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ai; mkdir -p $out
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in release norelease release norelease; do
  if [ $v = norelease ]; then export RSEM_HIP_NO_RELEASE=1; else unset RSEM_HIP_NO_RELEASE; fi
  ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_$v.log 2>&1
  echo "$v: $(grep -E 'device loop|main\(\) total' $out/dropin_$v.log | tr '\n' ' ') $(grep real $out/dropin_$v.log)"
done
rm -rf $D
