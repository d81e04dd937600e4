# Round 6, call 22 (k_code_reads as one coalesced pass): the model rounds' kernel with a read position as ONE 16-bit code (quality x base; table index = one multiply-add of the
# reference base): parity (CLI tests against the reference's goldens and binary), then per-kernel times at a fifth of configs[2] against HEAD's library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06v; mkdir -p $out
( timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x > $out/cli_tests.log 2>&1; echo "cli tests rc=$?" ); tail -3 $out/cli_tests.log
( MODES="default lib:mbase" timeout 900 tools/profile_model_rounds.sh 10526315 200000 > $out/model_rounds.log 2>&1; echo "model rounds rc=$?" ); cat $out/model_rounds.log
