# Round 6, call 19: the tree as it stands: smoke, the whole GPU suite, rocprofv3 kernel stats + PMC traffic of the headline command, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06s; mkdir -p $out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" ); tail -2 $out/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?" ); tail -4 $out/gpu_tests.log
( timeout 900 tools/profile_round.sh r06s "C3" > $out/profile_round.log 2>&1; echo "profile rc=$?" ); tail -25 $out/profile_round.log
( timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" ); wc -c $out/bench_line.json; cat $out/bench_line.json; tail -3 $out/bench.err
cp profiles/bench_detail_latest.json $out/bench_detail.json 2>/dev/null
