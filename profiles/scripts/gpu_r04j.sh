#!/bin/bash
# Round 4, call J: the column pass of the split rows against the size of its row-slot blocks (gathers in L2 vs atomics per block and id).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04j; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
for lg in ${LGS:-17 19 20 21 22 31}; do
  rm -rf /tmp/prof_j
  RSEM_HIP_CSC_BLOCK_LG=$lg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/lg$lg.json 2> $out/lg$lg.err
  python - /tmp/prof_j $lg <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("block 2^%s slots:" % sys.argv[2], "  ".join("%s %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:14], float(r["AverageNs"]) / 1e3) for r in rows[:3]))
PY
done
rm -rf /tmp/prof_j
