cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$1
export RSEM_GX_VERBOSE=1
L=$PWD/rsem_amd
( timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 1,32 > gpurun_out/$1/product.log 2>&1; echo "product rc=$?" ); grep "ms/round\|barriers per" gpurun_out/$1/product.log
for v in gxprof; do ( RSEM_HIP_LIB=$L/librsem_hip_$v.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > gpurun_out/$1/$v.log 2>&1; echo "$v rc=$?" ); grep "ms/round\|cycles per tile" gpurun_out/$1/$v.log; done
( timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 64 > gpurun_out/$1/c3_1chain.log 2>&1 ); grep "ms/round" gpurun_out/$1/c3_1chain.log
