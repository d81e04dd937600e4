#!/bin/bash
# Round 3: exact Gibbs, the resolve rounds under a finer phase profile, with and without a bitmap filter for the scan.
budget=${1:-300}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
for v in "" $VARIANTS; do
  lib=rsem_amd/librsem_hip${v:+_$v}.so
  step "C3x0.2_${v:-product}" 100 env RSEM_HIP_LIB=$PWD/$lib python tools/gibbs_exact_profile.py 0.2 8 6 C3 wg
done
for v in "" $VARIANTS2; do
  lib=rsem_amd/librsem_hip${v:+_$v}.so
  step "C2_${v:-product}" 100 env RSEM_HIP_LIB=$PWD/$lib python tools/gibbs_exact_profile.py 1.0 8 6 C2 wg
done
echo "== total $(( $(date +%s) - start )) s"
