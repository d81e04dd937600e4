cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b
export RSEM_GX_VERBOSE=1
L=$PWD/rsem_amd
( timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 1,16,32 > gpurun_out/r05b/product.log 2>&1; echo "product rc=$?" ); grep "ms/round" gpurun_out/r05b/product.log
for v in gxfence gx256; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > gpurun_out/r05b/$v.log 2>&1; echo "$v rc=$?" ); grep "ms/round" gpurun_out/r05b/$v.log; done
for v in gxprof gx256prof; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > gpurun_out/r05b/$v.log 2>&1; echo "$v rc=$?" ); grep "ms/round\|cycles per tile" gpurun_out/r05b/$v.log; done
( timeout 300 python tools/gibbs_team_profile.py 0.5 8 4 C2 1,32 > gpurun_out/r05b/c2.log 2>&1 ); grep "ms/round" gpurun_out/r05b/c2.log
( timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 1,64 > gpurun_out/r05b/c3_1chain.log 2>&1 ); grep "ms/round" gpurun_out/r05b/c3_1chain.log
( timeout 300 python tools/gibbs_team_profile.py 0.2 64 4 C3 1,4 > gpurun_out/r05b/c3_64chains.log 2>&1 ); grep "ms/round" gpurun_out/r05b/c3_64chains.log
( timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05b/gpu_tests.log 2>&1; echo "suite rc=$?" )
tail -5 gpurun_out/r05b/gpu_tests.log
