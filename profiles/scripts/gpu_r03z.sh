#!/bin/bash
# Round 3: Gibbs sweep, the G lanes of a read taking turns at drawing its uniforms (product) against lane 0 drawing every slice (rng0).
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03z; mkdir -p $out
timeout 100 python -m pytest tests/test_gibbs_gpu.py -q -m gpu -x > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -3
for v in "" rng0 ""; do for c in C3 C2; do echo -n "${v:-product} $c: "; RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so timeout 60 python tools/gibbs_profile.py 1.0 40 $c 2>&1 | tail -1; done; done
echo "== total $(( $(date +%s) - start )) s"
