#!/bin/bash
# Round 4, call R: where a tile's cycles go in the final exact-Gibbs kernel (RSEM_GX_PROFILE build), C3 shape at 5 % of its reads, 8 chains.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gxprof.so timeout 150 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | tr '|' '\n' | cut -c1-400
