#!/bin/bash
# Round 3: guest ids (ids outside a unit's LDS window in slots behind it) against the same library with guest_ids=0:
# the EM tests, then bench legs C3 (headline), C2, C2R, C3X with both settings.
budget=${1:-420}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
tag=r03m
out=gpurun_out/$tag; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_em 300 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_dist_gpu.py tests/test_gibbs_gpu.py -q -m gpu > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -8"
B="python bench.py --config C3 --legs C2,C2R,C3X --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
for v in apart_guests:1:1 apart:1:0 guests:0:1 plain:0:0; do
  IFS=: read n a g <<< "$v"
  step bench_$n 200 bash -c "RSEM_HIP_APART=$a RSEM_GUEST_IDS=$g $B > $out/bench_$n.json 2> $out/bench_$n.err; tail -1 $out/bench_$n.err"
done
for a in 1 0; do
  step gibbs_c3x_apart$a 120 bash -c "RSEM_HIP_APART=$a python tools/gibbs_profile.py 1.0 40 C3X 2>&1 | tail -2"
done
python - $out <<'PY'
import json, sys
for n in ("apart_guests", "apart", "guests", "plain"):
    try:
        d = json.loads(open("%s/bench_%s.json" % (sys.argv[1], n)).read().strip().split("\n")[-1])
        print(n, "C3 launch ms", d["roofline"].get("avg_launch_ms"), "frac", d["roofline"]["frac"], "ms_per_step", d["ms_per_step"], "q32", d.get("q32_value_planes", {}).get("estep_avg_launch_ms"))
        for k, v in d["other_configs"].items():
            print("   ", k, {kk: v.get(kk) for kk in ("estep_avg_launch_ms", "frac", "ms_per_step")}, v.get("parity_one_step"))
    except Exception as e:
        print(n, "unreadable:", e)
PY
echo "== total $(( $(date +%s) - start )) s"
