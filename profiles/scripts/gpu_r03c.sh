#!/bin/bash
# Round 3, third GPU call: the restructured workgroup-per-chain exact sampler (tests, per-round time, phase profile), the
# Gibbs sweep candidates (NT, DPP scan, RNG spread) with checksums.
budget=${1:-600}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03c; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 200 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg,coop 2>&1 | tee $out/exact_c2.log"
step exact_c3x02 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg,coop 2>&1 | tee $out/exact_c3x0.2.log"
step exact_prof_c2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_prof_c2.log"
step exact_prof_c3 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2.log"
step exact_prof_c3_1chain 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 1 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2_1chain.log"
step exact_c3 150 bash -c "python tools/gibbs_exact_profile.py 1.0 8 2 C3 wg 2>&1 | tee $out/exact_c3.log"
for v in default gnt gdpp gdppb default gnt gdpp; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  step gibbs_$v 90 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 40 C3 2>&1 | tail -1 | tee $out/gibbs_$v.log; RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 60 C2 2>&1 | tail -1 | tee -a $out/gibbs_$v.log"
done
echo "== total $(( $(date +%s) - start )) s"
