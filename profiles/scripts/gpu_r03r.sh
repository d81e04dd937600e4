#!/bin/bash
# Round 3: exact Gibbs with the next tile's staging data loaded ahead into registers.
budget=${1:-300}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03r; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests 240 bash -c "python -m pytest tests/test_gibbs_gpu.py tests/test_cli_gpu.py -q -m gpu -k 'gibbs or Gibbs' > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -8"
for v in "" xprof; do
  step "exact_C3x0.2_${v:-product}" 100 env RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so python tools/gibbs_exact_profile.py 0.2 8 6 C3 wg
done
step "exact_C2_product" 100 python tools/gibbs_exact_profile.py 1.0 8 6 C2 wg
step "exact_C5_product" 100 python tools/gibbs_exact_profile.py 0.02 8 6 C5 wg
echo "== total $(( $(date +%s) - start )) s"
