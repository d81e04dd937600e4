#!/bin/bash
# The library variants waiting for a measurement (estep_block.hpp / gibbs_block.hpp say what each one is), next to the
# product's library: rsem_amd/librsem_hip_<tag>.so, selected with RSEM_HIP_LIB.  Round 3's first set (reciprocal by Newton, DPP
# reduction, clamp fast path, ...) was measured in profiles/r03a_variants_and_steps.log; winners adopted, the rest deleted.
cd "$(dirname "$0")/.."
exec tools/build_variants.sh nt2 "-DRSEM_NT_LEVEL=2" gnt "-DRSEM_GIBBS_NT=1" grs "-DRSEM_GIBBS_RNG_SPREAD=1" gboth "-DRSEM_GIBBS_NT=1 -DRSEM_GIBBS_RNG_SPREAD=1" xw4 "-DRSEM_GX_W=4" xw6 "-DRSEM_GX_W=6" xwg "-DRSEM_GX_SCOPE=__HIP_MEMORY_SCOPE_WORKGROUP"
