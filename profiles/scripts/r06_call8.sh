# Round 6, call 8: the far-queue loop with theta of the far ids requested a slice ahead (ring of 3 register sets; a build with 4), the
# rule for which units take it; tests on it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06h; mkdir -p $out
( timeout 900 python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py -m gpu -q -x > $out/em_tests.log 2>&1; echo "em tests rc=$?" ); tail -3 $out/em_tests.log
( timeout 700 python tools/xrows_probe.py C3X,C3X30,C2R,C3 most,most_noq > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
( RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_fq4.so timeout 500 python tools/xrows_probe.py C3X,C3X30 most > $out/xrows_probe_fq4.log 2>&1; echo "xrows fq4 rc=$?" ); cat $out/xrows_probe_fq4.log
