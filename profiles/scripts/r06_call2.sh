# Round 6, call 2: the layout policies for reads that reach beyond their gene (tools/xrows_probe.py), the counters of k_model_group
# once more (call 1 lost them to a pass that hung), where the drop-in's 2 seconds go at 5 % of configs[2].
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06b; mkdir -p $out
( timeout 900 python tools/xrows_probe.py C3X,C3X30,C2R most,all_1s,all,all_noids > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
( timeout 240 python -m pytest tests/test_em_gpu.py -m gpu -q -x -k "another_gene or split or unstructured" > $out/em_tests_default.log 2>&1; echo "em tests (default policy) rc=$?" ); tail -3 $out/em_tests_default.log
( RSEM_HIP_SPLIT_POLICY=all timeout 400 python -m pytest tests/test_em_gpu.py -m gpu -q -x > $out/em_tests_all.log 2>&1; echo "em tests (policy all) rc=$?" ); tail -3 $out/em_tests_all.log
( timeout 700 tools/model_group_pmc.sh $out/model_pmc > $out/model_pmc.log 2>&1; echo "pmc rc=$?" ); tail -100 $out/model_pmc.log
D=/tmp/c3_5pct; rm -rf $D
tools/bin/gen_temp $D 2631578 200000 3 20250925 100 nosam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
for i in 1 2; do ( time RSEM_HIP_TIMING=1 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_5pct_$i.log 2>&1; done
grep -E "timing|real" $out/dropin_5pct_2.log
ls -la rsem_amd/librsem_hip.so
rm -rf $D
