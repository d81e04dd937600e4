# Round 6, call 20: the whole GPU suite on the tree (after the fix of test_dist_gpu), then the -b pass at 10 % of configs[2] with BAM input
# (own inflate + own deflate + folded CRC) against the reference -p 64.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06t; mkdir -p $out
( timeout 2400 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?" ); tail -6 $out/gpu_tests.log
( TAG=r06t timeout 1500 tools/e2e_bam.sh > $out/e2e_bam.log 2>&1; echo "e2e_bam rc=$?" ); cat $out/e2e_bam.log
