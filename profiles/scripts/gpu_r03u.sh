#!/bin/bash
# Round 3: Gibbs sweep with the sid planes loaded only where a tuple starts (product) or always (gidsalways); EM / Gibbs tests on the product.
budget=${1:-260}
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03u; mkdir -p $out
for rep in 1 2; do for v in "" gidsalways; do
  for c in C3 C2; do echo -n "${v:-product} $c: "; RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so timeout 60 python tools/gibbs_profile.py 1.0 40 $c 2>&1 | tail -1; done
done; done
timeout 150 python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_gibbs_gpu.py -q -m gpu -x > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -3
echo "== total $(( $(date +%s) - start )) s"
