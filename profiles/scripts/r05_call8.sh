cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05z; mkdir -p $O
export RSEM_GX_VERBOSE=1
( timeout 120 python tools/gibbs_team_profile.py 0.05 8 4 C3 0 > $O/two_a.log 2>&1 ) &
( timeout 120 python tools/gibbs_team_profile.py 0.05 8 4 C3 0 > $O/two_b.log 2>&1 ) &
wait
for f in two_a two_b; do echo "== $f: $(grep -c 'teams of' $O/$f.log) team run(s); $(grep 'ms/round' $O/$f.log | cut -c1-140)"; done
ls /tmp/rsem_hip_team_*.lock 2>/dev/null
timeout 120 python -m pytest tests/test_gibbs_gpu.py -m gpu -q -x -k "team_size or bit_identical" > $O/gibbs_tests.log 2>&1; grep -E "passed|failed" $O/gibbs_tests.log | tail -1
