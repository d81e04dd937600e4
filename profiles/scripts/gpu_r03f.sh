#!/bin/bash
# Round 3, sixth GPU call: exact sampler with the shared-items resolve (tests, time, profile), the anchor-keyed layout
# (EM tests, C3X / C2R / C2 / C3 E-step times).
budget=${1:-600}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03f; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 200 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_prof_c2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_prof_c2.log"
step exact_prof_c3 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2.log"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_c2.log"
step exact_c3x02 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_c3x0.2.log"
step exact_c5 120 bash -c "python tools/gibbs_exact_profile.py 0.02 8 2 C5 wg,coop 2>&1 | tee $out/exact_c5x0.02.log"
step tests_em 200 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_dist_gpu.py -x -q -k 'not full_size' > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -3"
step bench 200 bash -c "python bench.py --steps 20 --warmup 5 --legs C2,C2R,C3X --no-gibbs --no-ci --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python -c \"
import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print('C3 estep %.4f step %.4f step/launch %.4f frac %.3f q32 %.4f' % (r['avg_launch_ms'], d['ms_per_step'], r['step_over_launch'], r['frac'], d['q32_value_planes']['estep_avg_launch_ms']))
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v['parity_one_step'].get('ok') for k, v in d.get('other_configs', {}).items()}, d['checks']['parity_one_step'].get('ok'))\""
echo "== total $(( $(date +%s) - start )) s"
