#!/bin/bash
# Round 4, call T: the model rounds' kernel without loads under conditions against the kernel before (variants/model_before = the
# library of commit 577e7ad), a fifth of configs[2] through the program; then the CLI tests (every model type against the reference).
# (variants/model_before/ was a copy of rsem_amd/librsem_hip.so and rsem_amd/bin/rsem-run-em built at commit 577e7ad; it is not kept.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_HIP_TIMING=1
start=$(date +%s)
D5=/tmp/c3fifth; rm -rf $D5
tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D5/temp/s_alignable_1.fq $D5/temp/s_alignable_2.fq > /dev/null 2>&1
for v in before after before after; do
  exe=rsem_amd/bin/rsem-run-em; [ $v = before ] && exe=variants/model_before/bin/rsem-run-em
  $exe $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s -q > /tmp/run_$v.out 2>&1; echo "$v rc=$? $(grep -E 'model round|rounds 1-11|rounds >= 12' /tmp/run_$v.out | tr -s ' ' | tr '\n' ';' | cut -c1-700)"
  cp $D5/stat/s.theta /tmp/theta_$v; cp $D5/stat/s.model /tmp/model_$v
done
cmp /tmp/theta_before /tmp/theta_after && echo "theta identical"; cmp /tmp/model_before /tmp/model_after && echo "model identical"
echo "== runs $(( $(date +%s) - start )) s"
rm -rf $D5
timeout 100 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== total $(( $(date +%s) - start )) s"
