# Round 6, call 40: the -b pass at 10 % of configs[2], BAM input, on the final tree against the reference -p 64 in the same call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06an; mkdir -p $out
echo "nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( TAG=r06an timeout 1500 tools/e2e_bam.sh > $out/e2e_bam.log 2>&1; echo "e2e_bam rc=$?" ); cat $out/e2e_bam.log
