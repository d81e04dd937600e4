# Round 6, call 27: exact chain: the confirming phases' look-up by ring (ids whose cells changed in the phase before) against HEAD's library, same box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06aa; mkdir -p $out
export RSEM_GX_VERBOSE=1
L=$PWD/rsem_amd
run() { tag=$1; shift; ( "$@" timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $out/$tag.log 2>&1 ); echo "$tag: $(grep 'ms/round' $out/$tag.log | sed 's/.*team=32: //;s/, 0.0.*//') $(grep -o 'stage.*phases [0-9.]*' $out/$tag.log)"; }
run base env RSEM_HIP_LIB=$L/librsem_hip_base.so
run new env
run base2 env RSEM_HIP_LIB=$L/librsem_hip_base.so
run new2 env
run baseprof env RSEM_HIP_LIB=$L/librsem_hip_baseprof.so
run newprof env RSEM_HIP_LIB=$L/librsem_hip_gxprof.so
( timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 64 > $out/one_chain.log 2>&1 ); grep "ms/round" $out/one_chain.log
( RSEM_HIP_LIB=$L/librsem_hip_base.so timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 64 > $out/one_chain_base.log 2>&1 ); grep "ms/round" $out/one_chain_base.log
( timeout 600 python -m pytest tests/test_gibbs_gpu.py -m gpu -q -x > $out/gibbs_tests.log 2>&1; echo "gibbs tests rc=$?" ); tail -3 $out/gibbs_tests.log
