#!/bin/bash
# Round 4, call F: the state at the end of the round -- smoke(), the whole -m gpu suite, the bench line with the driver's arguments,
# per-kernel times of the E step with split rows (C2R).
budget=${1:-1000}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04f; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step smoke 120 python -c "import __graft_entry__ as g; g.smoke()"
rm -rf /tmp/prof_c2r
step stats_C2R 150 env RSEM_WL_CACHE=/dev/shm/rsem_wl rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2r -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/C2R.json 2> $out/C2R.err
python - /tmp/prof_c2r $out/C2R_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:6]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-46s calls %6s avg %10.1f us total %9.1f ms" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf /tmp/prof_c2r
step tests_all 700 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -12"
step bench_default 600 bash -c "python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err; python -c \"
import json; d=json.load(open('$out/bench_default.json'))
print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','checks')}, indent=None)[:800])
r=d['roofline']; print({k: r.get(k) for k in ('achieved','frac','frac_physical','traffic','frac_of_traffic','avg_launch_ms')}); print(r.get('physical'))
e=d.get('e2e_wall_clock',{}); print(json.dumps(e.get('measured'))[:900]); f=e.get('full_size',{}); print({k: f.get(k) for k in ('dropin_s','dropin_rounds','reference_s','speedup','speedup_against_undisturbed_reference','same_round_count_as_the_reference')})
print(json.dumps(d.get('cpu_baseline'))[:700])
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','split_rows','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v.get('physical', {}).get('physical_over_pmc') for k, v in d.get('other_configs', {}).items()})
g=d.get('gibbs',{}); print({k: g.get(k) if not isinstance(g.get(k), dict) else {kk: g[k].get(kk) for kk in ('ms_per_sweep','ms_per_round','frac_of_hbm_peak_per_gpu')} for k in ('parallel','exact','error')})\""
echo "== total $(( $(date +%s) - start )) s"
