#!/bin/bash
# Round 3: the software pipeline the compiler's waits allow (no flat loads, global atomics waited for in their own block,
# unconditional issues, peeled tails) against HEAD's library, and ring depths on top of it.
budget=${1:-500}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03p; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests 300 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_gibbs_gpu.py tests/test_dist_gpu.py -q -m gpu > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -8"
B="python bench.py --config C3 --legs C2,C2R,C3X --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
for v in "" head q2222 f3322; do
  n=${v:-product}
  step bench_$n 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so $B > $out/bench_$n.json 2> $out/bench_$n.err; tail -1 $out/bench_$n.err"
done
for v in "" head; do
  step gibbs_sweep_${v:-product} 90 bash -c "for c in C3 C2 C3X; do RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so python tools/gibbs_profile.py 1.0 40 \$c 2>&1 | tail -1; done"
done
python - $out <<'PY'
import json, sys
for n in ("product", "head", "q2222", "f3322"):
    try:
        d = json.loads(open("%s/bench_%s.json" % (sys.argv[1], n)).read().strip().split("\n")[-1])
        print("%-8s far units %s; C3 launch ms %.4f" % (n, " | ".join([str(d["config"].get("units_with_ids_outside_their_window"))] + [str(v.get("units_with_ids_outside_their_window")) for v in d["other_configs"].values()]), d["roofline"]["avg_launch_ms"]), "q32 %.4f" % d["q32_value_planes"]["estep_avg_launch_ms"],
              " ".join("%s %.4f (parity %.1e)" % (k, v["estep_avg_launch_ms"], v["parity_one_step"]["max_rel_diff_counts_vs_oracle"]) for k, v in d["other_configs"].items()))
    except Exception as e:
        print(n, "unreadable:", e)
PY
echo "== total $(( $(date +%s) - start )) s"
