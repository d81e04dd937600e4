#!/bin/bash
# Round 3, second GPU call: the workgroup-per-chain exact Gibbs sampler (tests + per-round time at C2 / C3 against the
# one-wave kernel, variants: 4 / 6 waves, workgroup-scope counts), the candidates NT level 2 / Gibbs NT / Gibbs RNG spread,
# the adopted set (NT + DPP + DS) as the default library.
#   tools/build_prepared_variants.sh && gpurun --timeout 900 -- 'tools/gpu_r03b.sh 840'
budget=${1:-840}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03b; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 300 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg,coop 2>&1 | tee $out/exact_c2.log"
step exact_c3 200 bash -c "python tools/gibbs_exact_profile.py 1.0 8 2 C3 wg 2>&1 | tee $out/exact_c3.log"
step exact_c3_64 120 bash -c "python tools/gibbs_exact_profile.py 0.2 64 2 C3 wg 2>&1 | tee $out/exact_c3_64chains.log"
for v in xw4 xw6 xwg; do
  lib=$PWD/rsem_amd/librsem_hip_$v.so; [ -f $lib ] || continue
  step exact_$v 120 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_$v.log"
done
step exact_default_c3x0.2 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg,coop 2>&1 | tee $out/exact_default_c3x0.2.log"
BV="python bench.py --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline --no-stream"
for v in default nt2 default nt2; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  step bench_$v 90 bash -c "RSEM_HIP_LIB=$lib $BV > $out/bench_$v.json 2> $out/bench_$v.err; python - $out/bench_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q, c2 = d['q32_value_planes'], d['other_configs']['C2']
print('%-7s C3 f64 estep %.4f ms step %.4f | q32 estep %.4f ms step %.4f dtheta %.2e || C2 f64 estep %.4f step %.4f | q32 estep %.4f step %.4f' % (sys.argv[2],
    d['roofline']['avg_launch_ms'], d['ms_per_step'], q['estep_avg_launch_ms'], q['ms_per_step'], q['theta_max_rel_diff_vs_f64_after_20_rounds'],
    c2['estep_avg_launch_ms'], c2['ms_per_step'], c2['q32_value_planes']['estep_avg_launch_ms'], c2['q32_value_planes']['ms_per_step']))
PY"
done
for v in default gnt grs gboth default gboth; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  step gibbs_$v 90 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 40 C3 2>&1 | tail -1 | tee $out/gibbs_$v.log; RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 60 C2 2>&1 | tail -1 | tee -a $out/gibbs_$v.log"
done
step tests_em 200 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py -x -q -k 'not full_size' > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -3"
echo "== total $(( $(date +%s) - start )) s"
