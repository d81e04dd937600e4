#!/bin/bash
# Round 3, seventh + eighth GPU calls: exact sampler with the move-endpoint hash table (tests, time, profile), atomics microbenchmark.
budget=${1:-420}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03h; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 200 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_prof_c2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_prof_c2.log"
step exact_prof_c3 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2.log"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_c2.log"
step exact_c3x02 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_c3x0.2.log"
step exact_c3x02_64 120 bash -c "python tools/gibbs_exact_profile.py 0.2 64 2 C3 wg 2>&1 | tee $out/exact_c3x0.2_64chains.log"
step exact_c5 120 bash -c "python tools/gibbs_exact_profile.py 0.02 8 2 C5 wg,coop 2>&1 | tee $out/exact_c5x0.02.log"
step tests_cli_gibbs 200 bash -c "python -m pytest tests/test_cli_gpu.py -x -q -k gibbs > $out/tests_cli.log 2>&1; grep -E 'passed|failed|rror' $out/tests_cli.log | tail -3"
step atomics 60 bash -c "tools/microbench/atomics 2>&1 | tee $out/atomics.log"
echo "== total $(( $(date +%s) - start )) s"
