#!/bin/bash
# Round 4, call V: is the product build of the exact chain the same speed before and after the thread-count parameter?  (gxprev = gibbs.hip of commit 6baed31)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gxprev.so timeout 12 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | cut -c1-200
timeout 12 python tools/gibbs_exact_profile.py 0.05 8 3 C3 wg 2>&1 | cut -c1-200
