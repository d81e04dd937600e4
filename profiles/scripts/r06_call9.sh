# Round 6, call 9: the exact chain with its random numbers made a sweep ahead by a workgroup of their own (k_gibbs_mt_stream on a
# second stream) and the published moves taken back behind the commit barrier (two copies of the team's tables): GPU tests, time per
# round by team size, where a tile's cycles go now, tiles closed at fewer items; configs[2] at full size alone on the host, finer marks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06i; mkdir -p $out
export RSEM_GX_VERBOSE=1
( timeout 900 python -m pytest tests/test_gibbs_gpu.py -m gpu -q -x > $out/gibbs_tests.log 2>&1; echo "gibbs tests rc=$?" ); tail -3 $out/gibbs_tests.log
( timeout 400 python tools/gibbs_team_profile.py 0.2 8 5 C3 1,16,32 > $out/team_profile.log 2>&1; echo "profile rc=$?" ); grep "ms/round\|barriers per" $out/team_profile.log
( RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_gxprof.so timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $out/team_phases.log 2>&1; echo "phases rc=$?" ); grep "ms/round\|cycles per tile" $out/team_phases.log
( timeout 300 python tools/gibbs_team_profile.py 0.2 1 4 C3 64 > $out/one_chain.log 2>&1 ); grep "ms/round" $out/one_chain.log
for cap in 3300 3500 3700; do ( RSEM_GX_TILE_ITEMS=$cap timeout 300 python tools/gibbs_team_profile.py 0.2 8 5 C3 32 > $out/team_cap_$cap.log 2>&1 ); echo "cap $cap: $(grep 'ms/round' $out/team_cap_$cap.log)"; done
# configs[2] at full size through the program, alone on the host
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for i in 1 2; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_full_$i.log 2>&1; done
grep -E "timing|real" $out/dropin_full_2.log
rm -rf $D
