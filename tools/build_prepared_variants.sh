#!/bin/bash
# The library variants waiting for a measurement, next to the product's library: rsem_amd/librsem_hip_<tag>.so, selected with
# RSEM_HIP_LIB.  Earlier sets: profiles/r03a_variants_and_steps.log, r03b_variants.log, r03c_exact_and_sweep.log,
# r03d_exact_sweep_bench.log -- winners adopted, the rest deleted.
cd "$(dirname "$0")/.."
exec tools/build_variants.sh xprof "-DRSEM_GX_PROFILE=1"
