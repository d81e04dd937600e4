# Round 6, call 5: the units with ids outside their window dealt evenly over the launch order (their gathers and atomics beside the
# compact units' streaming); then the default bench line on this tree (new Gibbs legs, the reference's rsem-run-gibbs and calcCI in the
# same run), and the GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06e; mkdir -p $out
( timeout 600 python tools/xrows_probe.py C3X,C3X30,C2R,C3 most,most_spread,all_spread > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
( timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" ); wc -c $out/bench_line.json; cat $out/bench_line.json; tail -3 $out/bench.err
cp profiles/bench_detail_latest.json $out/bench_detail.json 2>/dev/null
