#!/bin/bash
# The library variants waiting for a measurement (gibbs_block.hpp / gibbs_exact_wg.hpp say what each one is), next to the
# product's library: rsem_amd/librsem_hip_<tag>.so, selected with RSEM_HIP_LIB.  Earlier sets: profiles/r03a_variants_and_steps.log,
# profiles/r03b_variants.log -- winners adopted, the rest deleted.
cd "$(dirname "$0")/.."
exec tools/build_variants.sh xprof "-DRSEM_GX_PROFILE=1" gnt "-DRSEM_GIBBS_NT=1" gdpp "-DRSEM_GIBBS_NT=1 -DRSEM_GIBBS_DPP=1" gdppb "-DRSEM_GIBBS_NT=1 -DRSEM_GIBBS_DPP=1 -DRSEM_GIBBS_RNG_SPREAD=1"
