# Round 6, call 7: the far queue (units with ids outside their window: their partial counts for such ids wait in LDS, the loop keeps
# its prefetch) -- tests, then the probe with and without it; the -b pass with BAM input once more (framer thread).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06g; mkdir -p $out
( timeout 900 python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py -m gpu -q -x > $out/em_tests.log 2>&1; echo "em tests rc=$?" ); tail -3 $out/em_tests.log
( timeout 700 python tools/xrows_probe.py C3X,C3X30,C2R,C3 most,most_noq > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
( timeout 600 python -m pytest tests/test_cli_gpu.py tests/test_dist_gpu.py -m gpu -q -x -k "em or EM or shard or matches_reference" > $out/cli_tests.log 2>&1; echo "cli tests rc=$?" ); tail -3 $out/cli_tests.log
( TAG=r06g timeout 1500 tools/e2e_bam.sh > $out/e2e_bam.log 2>&1; echo "e2e_bam rc=$?" ); cat $out/e2e_bam.log
