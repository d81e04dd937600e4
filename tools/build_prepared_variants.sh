#!/bin/bash
# The library variants prepared at the end of round 2 ("Prepared, not measured"), for one GPU call:
#   tools/build_prepared_variants.sh && gpurun --timeout 900 -- 'tools/gpu_q32_depths.sh 860 default rcp dpp ds clamp fma nt magic neff all allm g1' ; gpurun -- 'tools/gpu_gibbs_variants.sh 400 default gsa gsap gd1 gd2 gd4 gd7'
# (tools/gpu_q32_depths.sh: per variant the EM / Q32 GPU tests and the bench line with the Q32 leg on C3 and C2; the
#  lane-policy tests and a bench with lane_policy need the g1 library: RSEM_HIP_LIB=rsem_amd/librsem_hip_g1.so.)
cd "$(dirname "$0")/.."
exec tools/build_variants.sh \
  rcp "-DRSEM_FAST_RCP=1" dpp "-DRSEM_DPP_REDUCE=1" ds "-DRSEM_SPILL_DS=1" clamp "-DRSEM_CLAMP_FAST=1" fma "-DRSEM_FMA_ACC=1" nt "-DRSEM_NT_LOADS=1" magic "-DRSEM_Q32_MAGIC=1 -DRSEM_Q32_DEPTHS=8,6,4,2" neff "-DRSEM_NEFF_BALLOT=1" \
  all "-DRSEM_FAST_RCP=1 -DRSEM_DPP_REDUCE=1 -DRSEM_SPILL_DS=1 -DRSEM_CLAMP_FAST=1 -DRSEM_FMA_ACC=1 -DRSEM_NEFF_BALLOT=1" allm "-DRSEM_FAST_RCP=1 -DRSEM_DPP_REDUCE=1 -DRSEM_CLAMP_FAST=1 -DRSEM_FMA_ACC=1 -DRSEM_Q32_MAGIC=1 -DRSEM_NEFF_BALLOT=1" \
  g1 "-DRSEM_GENERAL_G=1" gd1 "-DRSEM_GDIAG=1" gd2 "-DRSEM_GDIAG=2" gd4 "-DRSEM_GDIAG=4" gd7 "-DRSEM_GDIAG=7" gsa "-DRSEM_GIBBS_SCALAR_ADDR=1" gsap "-DRSEM_GIBBS_SCALAR_ADDR=1 -DRSEM_GIBBS_PHILOX2=1"
