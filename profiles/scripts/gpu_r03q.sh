#!/bin/bash
# Round 3: the second layout pass (stray reads sorted apart) -- tests, far-unit counts, E step F64 / Q32 and Gibbs sweep on C3 / C3X.
budget=${1:-400}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03q; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests 300 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_gibbs_gpu.py tests/test_dist_gpu.py -q -m gpu -s > $out/tests.log 2>&1; grep -E 'passed|failed|rror|units with ids' $out/tests.log | tail -8"
B="python bench.py --config C3 --legs C2,C2R,C3X --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
step bench 120 bash -c "$B > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.err"
step bench_c3x_q32 120 bash -c "python bench.py --config C3X --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream > $out/bench_c3x.json 2> $out/bench_c3x.err; tail -1 $out/bench_c3x.err"
step gibbs_sweep 90 bash -c "for c in C3 C2 C3X; do python tools/gibbs_profile.py 1.0 40 \$c 2>&1 | tail -1; done"
python - $out <<'PY'
import json, sys
for n in ("bench", "bench_c3x"):
    try:
        d = json.loads(open("%s/%s.json" % (sys.argv[1], n)).read().strip().split("\n")[-1])
        print("%-9s %s far units %s; launch ms %.4f q32 %.4f" % (n, d["config"]["synthetic_config"], d["config"].get("units_with_ids_outside_their_window"), d["roofline"]["avg_launch_ms"], d["q32_value_planes"]["estep_avg_launch_ms"]),
              " ".join("%s %.4f [far %s] (parity %.1e)" % (k, v["estep_avg_launch_ms"], v.get("units_with_ids_outside_their_window"), v["parity_one_step"]["max_rel_diff_counts_vs_oracle"]) for k, v in d.get("other_configs", {}).items()))
    except Exception as e:
        print(n, "unreadable:", e)
PY
echo "== total $(( $(date +%s) - start )) s"
