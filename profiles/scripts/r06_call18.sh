# Round 6, call 18: the -b pass with the folded CRC-32: 10 % of configs[2], BAM input, drop-in against the reference -p 64 in the same call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06r; mkdir -p $out
echo "nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -x -k "bam" > $out/cli_bam_tests.log 2>&1; echo "cli bam tests rc=$?" ); tail -3 $out/cli_bam_tests.log
( TAG=r06r timeout 1500 tools/e2e_bam.sh > $out/e2e_bam.log 2>&1; echo "e2e_bam rc=$?" ); cat $out/e2e_bam.log
