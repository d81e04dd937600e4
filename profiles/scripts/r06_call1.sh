# Round 6, call 1: the team barrier's way out (new test), counters of k_model_group per wait reason, the compact bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06a; mkdir -p $out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" ); tail -2 $out/smoke.log
( timeout 900 python -m pytest tests/test_gibbs_gpu.py -m gpu -q -x -k "team or resident" -s > $out/gibbs_team_tests.log 2>&1; echo "team tests rc=$?" ); tail -5 $out/gibbs_team_tests.log
( timeout 900 tools/model_group_pmc.sh $out/model_pmc > $out/model_pmc.log 2>&1; echo "pmc rc=$?" ); tail -90 $out/model_pmc.log
( timeout 600 python bench.py --legs C3X,C2R --no-cpu-baseline --no-gibbs --no-ci --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" ); wc -c $out/bench_line.json; cat $out/bench_line.json
