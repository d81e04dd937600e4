#!/bin/bash
# Round 3: the sid planes of slices without a new tuple: dummy loads (product) or none (F64 only: idsq; F64 and Q32: idsnever).
budget=${1:-200}
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03t; mkdir -p $out
B="python bench.py --config C3 --legs C2 --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-stream"
for rep in 1 2; do for v in "" idsq idsnever; do
  n=${v:-product}
  RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so timeout 100 $B > $out/bench_${n}_$rep.json 2> $out/bench_${n}_$rep.err
  python - $out/bench_${n}_$rep.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("%-9s C3 launch ms %.4f q32 %.4f" % (sys.argv[2], d["roofline"]["avg_launch_ms"], d["q32_value_planes"]["estep_avg_launch_ms"]), " ".join("%s %.4f" % (k, v["estep_avg_launch_ms"]) for k, v in d["other_configs"].items()))
PY
done; done
echo "== total $(( $(date +%s) - start )) s"
