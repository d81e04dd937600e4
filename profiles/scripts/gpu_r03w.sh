#!/bin/bash
# Round 3: exact Gibbs with the look-ups of a round on one work list (balanced over the threads); smoke().
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03w; mkdir -p $out
timeout 120 python -m pytest tests/test_gibbs_gpu.py -q -m gpu -x > $out/tests.log 2>&1; grep -E 'passed|failed|rror' $out/tests.log | tail -3
for v in "" xprof; do
  echo -n "${v:-product} "; RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip${v:+_$v}.so timeout 100 python tools/gibbs_exact_profile.py 0.2 8 6 C3 wg
done
timeout 60 python tools/gibbs_exact_profile.py 1.0 8 6 C2 wg
timeout 60 python tools/gibbs_exact_profile.py 0.02 8 6 C5 wg
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== total $(( $(date +%s) - start )) s"
