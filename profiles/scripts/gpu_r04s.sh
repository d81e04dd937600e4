#!/bin/bash
# Round 4, call S: the exact-Gibbs kernel after "no load under a condition" + narrow tail steps + the count-update wait moved behind the
# next tile's staging: its phase profile, then every GPU test of the exact chain (library and programs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
start=$(date +%s)
RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gxprof.so timeout 60 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | tr '|' '\n' | cut -c1-300
echo "== profile $(( $(date +%s) - start )) s"
timeout 150 python -m pytest tests/test_gibbs_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "exact or one_launch or chain_groups or gibbs_binary_handoff" 2>&1 | tail -4
echo "== tests $(( $(date +%s) - start )) s"
