#!/bin/bash
# Round 4, call U (the round's last 30 GPU seconds): the exact Gibbs chain with 256 threads per chain (the product) and with 512
# (-DRSEM_GX_THREADS=512: two waves per SIMD, the extra waves share the item-major phases), C3 shape at 5 % of its reads, 8 chains.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 20 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | cut -c1-200
RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gx512.so timeout 20 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | cut -c1-200
