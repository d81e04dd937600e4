#!/bin/bash
# Round 4, call K: the far passes reworked (row pass: coalesced loads staged in LDS; column pass: id-range windows in LDS) at C2R;
# parity tests of the split rows.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04k; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
rm -rf /tmp/prof_k
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/C2R_prof.json 2> $out/C2R_prof.err
python - /tmp/prof_k $out/C2R_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:5]:
    print("   %-46s calls %6s avg %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/prof_k
python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('C2R (no profiler): launch %.4f ms frac %.4f frac_physical %.4f parity %s' % (r['avg_launch_ms'], r['frac'], r['frac_physical'], d['checks']['parity_one_step']))"
timeout 200 python -m pytest tests/test_em_gpu.py -q -m gpu -k "unstructured or another_gene" 2>&1 | tail -3
