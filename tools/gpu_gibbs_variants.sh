#!/bin/bash
# One gpurun call: the Gibbs GPU tests and the sweep time (tools/gibbs_profile.py, C2-shaped matrix) per library variant.
#   tools/gpu_gibbs_variants.sh <budget seconds> default gsa ...
budget=${1:-400}; shift
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gibbs_variants; mkdir -p $out
step() { name=$1; lim=$2; shift 2; l=$(left); [ $l -lt 20 ] && { echo "== $name: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  t0=$(date +%s); timeout $lim "$@"; echo "== $name: rc=$? $(( $(date +%s) - t0 )) s"; }
for v in "$@"; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  step sweep_$v 150 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 60 > $out/sweep_$v.log 2>&1; tail -3 $out/sweep_$v.log"
  [[ "$v" == gd* ]] && continue   # elimination builds: times only, their results are meaningless
  step tests_$v 200 bash -c "RSEM_HIP_LIB=$lib python -m pytest tests/test_gibbs_gpu.py -x -q > $out/tests_$v.log 2>&1; tail -2 $out/tests_$v.log"
done
echo "== total $(( $(date +%s) - start )) s"
