# Round 6, call 44: the final tree: smoke, the whole GPU suite, the default bench line (its -b leg with BAM input now).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ar; mkdir -p $out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" ); tail -2 $out/smoke.log
( timeout 2400 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?" ); tail -4 $out/gpu_tests.log
( timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" ); wc -c $out/bench_line.json; cat $out/bench_line.json; tail -3 $out/bench.err
cp profiles/bench_detail_latest.json $out/bench_detail.json 2>/dev/null
