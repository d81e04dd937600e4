# Round 6, call 37: + mapped text files dropped under the shared lock before they are unmapped: CLI + pipeline tests, the 5 % input (whole program,
# three runs) and configs[2] at full size, finer marks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ak; mkdir -p $out
( timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x > $out/cli_tests.log 2>&1; echo "cli tests rc=$?" ); tail -3 $out/cli_tests.log
D=/tmp/c3_5pct; rm -rf $D
tools/bin/gen_temp $D 2631578 200000 3 20250925 100 nosam 5-16 | tail -1
for i in 1 2 3; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_5pct_$i.log 2>&1; grep real $out/dropin_5pct_$i.log; done
grep -E "timing" $out/dropin_5pct_3.log | grep -v "model round"
rm -rf $D
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for i in 1 2; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_full_$i.log 2>&1; done
grep -E "timing|real" $out/dropin_full_2.log | grep -v "model round"
python - $D <<'PY'
import gzip, sys, numpy as np
a = np.array(open(sys.argv[1] + "/stat/s.theta").read().split("\n")[1].split(), float)
b = np.array(gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1].split(), float)
m = b >= 1e-7
print("full size: theta max rel diff vs the reference's own theta of round 4: %.3g over %d transcripts" % (np.max(np.abs(a[m] - b[m]) / b[m]), m.sum()))
PY
rm -rf $D
