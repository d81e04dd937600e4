cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05d
export RSEM_GX_VERBOSE=1
L=$PWD/rsem_amd
( timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 1,8,16,32 > gpurun_out/r05d/product.log 2>&1; echo "product rc=$?" ); grep "ms/round\|barriers per" gpurun_out/r05d/product.log
for v in gx512; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > gpurun_out/r05d/$v.log 2>&1; echo "$v rc=$?" ); grep "ms/round" gpurun_out/r05d/$v.log; done
for v in gxprof; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > gpurun_out/r05d/$v.log 2>&1; echo "$v rc=$?" ); grep "ms/round\|cycles per tile" gpurun_out/r05d/$v.log; done
( timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 64 > gpurun_out/r05d/c3_1chain.log 2>&1 ); grep "ms/round" gpurun_out/r05d/c3_1chain.log
( timeout 300 python tools/gibbs_team_profile.py 0.05 8 4 C5 1,32 > gpurun_out/r05d/c5.log 2>&1 ); grep "ms/round" gpurun_out/r05d/c5.log
( timeout 600 python -m pytest tests/test_gibbs_gpu.py tests/test_cli_gpu.py -m gpu -q -x -k "exact or gibbs" > gpurun_out/r05d/gibbs_tests.log 2>&1; echo "gibbs tests rc=$?" )
tail -3 gpurun_out/r05d/gibbs_tests.log
