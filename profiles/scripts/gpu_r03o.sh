#!/bin/bash
# Round 3: (1) exact Gibbs with item-major resolve rounds against the read-major version (xbm) and under the phase profile;
# (2) E step: the new tuples' theta gather before the next slice's loads are issued, against after (takelast).
budget=${1:-400}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03o; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests 240 bash -c "python -m pytest tests/test_gibbs_gpu.py tests/test_em_gpu.py -q -m gpu > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -8"
for v in "" xbm xprof; do
  step "exact_C3x0.2_${v:-product}" 100 env RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so python tools/gibbs_exact_profile.py 0.2 8 6 C3 wg
done
step "exact_C2_product" 100 python tools/gibbs_exact_profile.py 1.0 8 6 C2 wg
step "exact_C5_product" 100 python tools/gibbs_exact_profile.py 0.02 8 6 C5 wg
B="python bench.py --config C3 --legs C2,C2R,C3X --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
for v in "" takelast; do
  n=${v:-product}
  step bench_$n 200 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so $B > $out/bench_$n.json 2> $out/bench_$n.err; tail -1 $out/bench_$n.err"
done
python - $out <<'PY'
import json, sys
for n in ("product", "takelast"):
    try:
        d = json.loads(open("%s/bench_%s.json" % (sys.argv[1], n)).read().strip().split("\n")[-1])
        print(n, "C3 launch ms %.4f" % d["roofline"]["avg_launch_ms"], "q32 %.4f" % d["q32_value_planes"]["estep_avg_launch_ms"],
              " ".join("%s %.4f (parity %.1e)" % (k, v["estep_avg_launch_ms"], v["parity_one_step"]["max_rel_diff_counts_vs_oracle"]) for k, v in d["other_configs"].items()))
    except Exception as e:
        print(n, "unreadable:", e)
PY
echo "== total $(( $(date +%s) - start )) s"
