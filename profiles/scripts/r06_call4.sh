# Round 6, call 4: units cut to fit one LDS window (sell_build_units) -- the layout policies for reads that reach beyond their gene
# once more; the layout-dependent GPU tests on the new units.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06d; mkdir -p $out
( timeout 700 python tools/xrows_probe.py C3X,C3X30,C2R,C3 most,all_1s,all > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
( timeout 900 python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_gibbs_gpu.py -m gpu -q -x > $out/layout_tests.log 2>&1; echo "layout tests rc=$?" ); tail -3 $out/layout_tests.log
( RSEM_HIP_SPLIT_POLICY=all timeout 400 python -m pytest tests/test_em_gpu.py -m gpu -q -x > $out/em_tests_all.log 2>&1; echo "em tests (policy all) rc=$?" ); tail -3 $out/em_tests_all.log
