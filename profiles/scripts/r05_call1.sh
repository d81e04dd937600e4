cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
export RSEM_GX_VERBOSE=1
( timeout 600 python -m pytest tests/test_gibbs_gpu.py -m gpu -q -x -s -k "team_size or at_scale" > gpurun_out/r05a/team_tests.log 2>&1; echo "team tests rc=$?" ) 
tail -15 gpurun_out/r05a/team_tests.log
( timeout 500 python tools/gibbs_team_profile.py 0.2 8 5 C3 1,4,8,16,32,0 > gpurun_out/r05a/team_profile.log 2>&1; echo "profile rc=$?" )
cat gpurun_out/r05a/team_profile.log | grep -v "^$" | tail -20
( RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gxprof.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 4 C3 1,32 > gpurun_out/r05a/team_profile_phases.log 2>&1; echo "phases rc=$?" )
grep -v "^$" gpurun_out/r05a/team_profile_phases.log | tail -12
( RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gx256.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 4 C3 1,32 > gpurun_out/r05a/team_profile_256.log 2>&1; echo "256 rc=$?" )
grep -v "^$" gpurun_out/r05a/team_profile_256.log | tail -6
( timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05a/gpu_tests.log 2>&1; echo "suite rc=$?" )
tail -8 gpurun_out/r05a/gpu_tests.log
