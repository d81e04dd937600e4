#!/bin/bash
# Round 3, fourth GPU call: the exact sampler with the transposed tile layout (tests, per-round time, phase profile, 2 / 4
# waves), the sweep kernel at 5 waves per SIMD, bench.py's new flow on a small config, then the default bench line.
budget=${1:-900}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03d; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 200 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg,coop 2>&1 | tee $out/exact_c2.log"
step exact_c3x02 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_c3x0.2.log"
step exact_prof_c2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_prof_c2.log"
step exact_prof_c3 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2.log"
step exact_xw2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xw2.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_xw2.log"
step exact_c5 120 bash -c "python tools/gibbs_exact_profile.py 0.02 8 2 C5 wg,coop 2>&1 | tee $out/exact_c5x0.02.log"
for v in default g5w default g5w; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  step gibbs_$v 90 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 40 C3 2>&1 | tail -1 | tee $out/gibbs_$v.log; RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 60 C2 2>&1 | tail -1 | tee -a $out/gibbs_$v.log"
done
step bench_small 300 bash -c "python bench.py --config C2 --legs C3X@0.05,C5@0.01 --steps 20 --warmup 5 > $out/bench_small.json 2> $out/bench_small.err; tail -3 $out/bench_small.err; python -c \"
import json; d=json.load(open('$out/bench_small.json'))
print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','roofline','checks','gibbs','e2e_wall_clock','cpu_baseline')}, indent=None)[:3500])
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','parity_one_step','error','generate_s')} for k, v in d.get('other_configs', {}).items()})\""
step bench_default 600 bash -c "python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err; python -c \"
import json; d=json.load(open('$out/bench_default.json'))
print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','roofline','checks','gibbs','e2e_wall_clock','cpu_baseline')}, indent=None)[:4000])
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','parity_one_step','error','generate_s')} for k, v in d.get('other_configs', {}).items()})\""
echo "== total $(( $(date +%s) - start )) s"
