#!/bin/bash
# Round 4, call L: the column pass (restored: runs of (block of 2^16 row slots, id), one atomic per run) against the number of its workgroups.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04l; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
for k in 8 16 32 1000; do
  rm -rf /tmp/prof_l
  RSEM_HIP_COLSUM_WG_PER_CU=$k timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/k$k.json 2> $out/k$k.err
  python - /tmp/prof_l $k <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("workgroups per CU %s:" % sys.argv[2], "  ".join("%s %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:14], float(r["AverageNs"]) / 1e3) for r in rows[:3]))
PY
done
rm -rf /tmp/prof_l
python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('C2R (no profiler): launch %.4f ms frac %.4f frac_physical %.4f parity %s' % (r['avg_launch_ms'], r['frac'], r['frac_physical'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle')))"
timeout 200 python -m pytest tests/test_em_gpu.py -q -m gpu -k "unstructured or another_gene" 2>&1 | tail -2
