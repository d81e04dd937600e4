# Round 6, call 10: exact chain, same box: HEAD's library (base), random numbers made ahead (on a second stream / on the teams' stream),
# take-backs behind / in front of the commit barrier; phases of each.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06j; mkdir -p $out
export RSEM_GX_VERBOSE=1
L=$PWD/rsem_amd
run() { tag=$1; shift; ( "$@" timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $out/$tag.log 2>&1 ); echo "$tag: $(grep 'ms/round' $out/$tag.log | sed 's/.*team=32: //;s/, 0.0.*//') $(grep -o 'stage.*phases [0-9.]*' $out/$tag.log)"; }
run base env RSEM_HIP_LIB=$L/librsem_hip_base.so
run lazy env
run lazy_serial env RSEM_GX_RND_SERIAL=1
run eager env RSEM_HIP_LIB=$L/librsem_hip_eager.so
run eager_serial env RSEM_HIP_LIB=$L/librsem_hip_eager.so RSEM_GX_RND_SERIAL=1
run base2 env RSEM_HIP_LIB=$L/librsem_hip_base.so
run baseprof env RSEM_HIP_LIB=$L/librsem_hip_baseprof.so
run lazyprof env RSEM_HIP_LIB=$L/librsem_hip_gxprof.so
run lazyprof_serial env RSEM_HIP_LIB=$L/librsem_hip_gxprof.so RSEM_GX_RND_SERIAL=1
run eagerprof env RSEM_HIP_LIB=$L/librsem_hip_eagerprof.so
