# Round 6, call 34: rocprofv3 --kernel-trace --stats of the PROGRAM (rsem-run-em on configs[2] as named, full size): which kernels the 10 seconds are made of.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ah; mkdir -p $out
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
export RSEM_HIP_NORMAL_EXIT=1 RSEM_HIP_TIMING=2
( time rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prog_stats -o p -- rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_full_under_rocprofv3.log 2>&1
cp $(find /tmp/prog_stats -name "*kernel_stats.csv" | head -1) $out/dropin_full_kernel_stats.csv
grep -E "timing|real" $out/dropin_full_under_rocprofv3.log | grep -v "model round"
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r06ah/dropin_full_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("device time in kernels: %.3f s over %d kernels" % (tot / 1e9, len(rows)))
for r in rows[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-46s calls %6s avg %10.1f us total %8.1f ms %5.1f %%" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
rm -rf $D /tmp/prog_stats
