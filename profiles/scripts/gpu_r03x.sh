#!/bin/bash
# Round 3: units whose reads leave their window take the new tuples (spill, theta gather) BEFORE the next slice's loads are issued
# (product) against after (head = the commit before).
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03x; mkdir -p $out
timeout 100 python -m pytest tests/test_em_gpu.py -q -m gpu -x -k "another_gene or unstructured or step_matches or edge or long_rows" > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -3
B="python bench.py --config C3 --legs C2,C2R,C3X --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
for v in "" head ""; do
  n=${v:-product}
  RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so timeout 100 $B > $out/bench_$n.json 2> $out/bench_$n.err
  python - $out/bench_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("%-9s C3 launch ms %.4f q32 %.4f" % (sys.argv[2], d["roofline"]["avg_launch_ms"], d["q32_value_planes"]["estep_avg_launch_ms"]), " ".join("%s %.4f (%.1e)" % (k, v["estep_avg_launch_ms"], v["parity_one_step"]["max_rel_diff_counts_vs_oracle"]) for k, v in d["other_configs"].items()))
PY
done
echo "== total $(( $(date +%s) - start )) s"
