#!/bin/bash
# Tuning experiments: builds of librsem_hip.so that differ in compile-time constants of em.hip / gibbs.hip, next to the product's
# library (rsem_amd/librsem_hip_<tag>.so, selected with RSEM_HIP_LIB; git-ignored, shipped to the GPU box).
#   tools/build_variants.sh tag1 "-DRSEM_Q32_DEPTHS=4,4,3,3" tag2 "-D... -D..." ...
set -e
cd "$(dirname "$0")/.."
python -m rsem_amd.build > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=off"
B=rsem_amd/build
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  ( hipcc $FLAGS $defs -c rsem_amd/csrc/em.hip -o $B/em_$tag.o && hipcc $FLAGS $defs -c rsem_amd/csrc/gibbs.hip -o $B/gibbs_$tag.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o rsem_amd/librsem_hip_$tag.so $B/status.hip.o $B/comm.hip.o $B/em_$tag.o $B/gibbs_$tag.o $B/model.hip.o $B/ci.hip.o -ldl &&
    echo "built rsem_amd/librsem_hip_$tag.so ($defs)" ) &
done
wait
