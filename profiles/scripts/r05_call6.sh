cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05t; mkdir -p $O; D=/tmp/e2e5
rm -rf $D; tools/bin/gen_temp $D 2631578 200000 3 20250925 100 nosam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null 2>&1
export RSEM_HIP_TIMING=1
for i in 1 2; do ( time -p rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 ) > $O/em_$i.log 2>&1; grep -E "^\[timing\]|^real" $O/em_$i.log | tr '\n' ';'; echo; done
export RSEM_HIP_NORMAL_EXIT=1
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
rm -rf $D
