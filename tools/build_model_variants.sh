#!/bin/bash
# Builds of librsem_hip.so that differ in compile-time constants of model.hip (rsem_amd/librsem_hip_<tag>.so; tools/profile_model_rounds.sh
# MODES="default lib:<tag>" runs them).   tools/build_model_variants.sh tag1 "-DRSEM_GROUP_CHUNK=0" tag2 "..." ...
set -e
cd "$(dirname "$0")/.."
python -m rsem_amd.build > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=off"
B=rsem_amd/build
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  ( hipcc $FLAGS $defs -c rsem_amd/csrc/model.hip -o $B/model_$tag.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o rsem_amd/librsem_hip_$tag.so $B/status.hip.o $B/comm.hip.o $B/em.hip.o $B/gibbs.hip.o $B/model_$tag.o $B/ci.hip.o -ldl &&
    echo "built rsem_amd/librsem_hip_$tag.so ($defs)" ) &
done
wait
