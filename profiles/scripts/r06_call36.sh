# Round 6, call 36: the parsed inputs' pages given back on 8 threads by MADV_DONTNEED slices (no exclusive lock) before they are freed: the device
# loop of configs[2] at full size, against one thread (RSEM_HIP_RELEASE_THREADS=1) and no release at all; CLI tests first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06aj; mkdir -p $out
( timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x > $out/cli_tests.log 2>&1; echo "cli tests rc=$?" ); tail -2 $out/cli_tests.log
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in t8 t1 t8 t16 norelease t8; do
  unset RSEM_HIP_NO_RELEASE RSEM_HIP_RELEASE_THREADS
  case $v in norelease) export RSEM_HIP_NO_RELEASE=1;; t1) export RSEM_HIP_RELEASE_THREADS=1;; t16) export RSEM_HIP_RELEASE_THREADS=16;; esac
  ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_$v.log 2>&1
  echo "$v: $(grep -E 'device loop|main\(\) total' $out/dropin_$v.log | tr '\n' ' ') $(grep real $out/dropin_$v.log)"
done
python - $D <<'PY'
import gzip, sys, numpy as np
a = np.array(open(sys.argv[1] + "/stat/s.theta").read().split("\n")[1].split(), float)
b = np.array(gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1].split(), float)
m = b >= 1e-7
print("full size: theta max rel diff vs the reference's own theta of round 4: %.3g over %d transcripts" % (np.max(np.abs(a[m] - b[m]) / b[m]), m.sum()))
PY
rm -rf $D
