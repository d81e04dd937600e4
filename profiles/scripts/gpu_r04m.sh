#!/bin/bash
# Round 4, call M: the final tree -- smoke(), the whole -m gpu suite, headline + legs (no reference runs), C2R per-kernel times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04m; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
start=$(date +%s)
timeout 120 python -c "import __graft_entry__ as g; g.smoke()"; echo "== smoke rc=$?"
rm -rf /tmp/prof_m
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/C2R_prof.json 2> $out/C2R_prof.err
python - /tmp/prof_m $out/C2R_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:4]:
    print("   %-46s calls %6s avg %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/prof_m
timeout 700 python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -8
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ci > $out/bench_nocpu.json 2> $out/bench_nocpu.err; python -c "
import json; d=json.load(open('$out/bench_nocpu.json'))
r=d['roofline']; print({k: r.get(k) for k in ('frac','frac_physical','avg_launch_ms')}, d['ms_per_step'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle'))
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v.get('parity_one_step', {}).get('max_rel_diff_counts_vs_oracle') for k, v in d.get('other_configs', {}).items()})"
echo "== total $(( $(date +%s) - start )) s"
