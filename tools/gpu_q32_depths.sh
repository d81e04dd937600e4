#!/bin/bash
# One gpurun call: the Q32 parity tests and the bench line (F64 headline + Q32 leg, C3 and C2) for every library variant
# given (tools/build_variants.sh; "default" = the product's library), then the whole -m gpu suite on the default library.
#   tools/gpu_q32_depths.sh <budget seconds> default q4433 ...
budget=${1:-600}; shift
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/q32_depths; mkdir -p $out
step() { name=$1; lim=$2; shift 2; l=$(left); [ $l -lt 20 ] && { echo "== $name: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  t0=$(date +%s); timeout $lim "$@"; echo "== $name: rc=$? $(( $(date +%s) - t0 )) s"; }
for v in "$@"; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  extra=""; [[ "$v" == g1* ]] && extra="--lane-policy 1"
  step bench_$v 120 bash -c "RSEM_HIP_LIB=$lib python bench.py $extra --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err; python - $out/bench_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q, c2 = d['q32_value_planes'], d['other_configs']['C2']
print('C3 f64 estep %.4f ms step %.4f | q32 estep %.4f ms step %.4f dtheta %.2e || C2 f64 estep %.4f step %.4f | q32 estep %.4f step %.4f' % (
    d['roofline']['avg_launch_ms'], d['ms_per_step'], q['estep_avg_launch_ms'], q['ms_per_step'], q['theta_max_rel_diff_vs_f64_after_20_rounds'],
    c2['estep_avg_launch_ms'], c2['ms_per_step'], c2['q32_value_planes']['estep_avg_launch_ms'], c2['q32_value_planes']['ms_per_step']))
PY"
  step tests_$v 120 bash -c "RSEM_HIP_LIB=$lib python -m pytest tests/test_em_q32_gpu.py tests/test_em_gpu.py tests/test_em_lane_policy_gpu.py -x -q -k 'not full_size_c3 and not rsem_run_em' > $out/tests_$v.log 2>&1; tail -2 $out/tests_$v.log"
done
step tests_all 300 bash -c "python -m pytest tests -x -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|error' $out/tests_all.log | tail -3"
echo "== total $(( $(date +%s) - start )) s"
