#!/bin/bash
# One gpurun call for the Q32 value planes: parity tests, the bench line with the Q32 leg, the whole -m gpu suite, then
# rocprofv3 kernel stats and PMC traffic of a run whose headline context streams Q32 planes.  Every step is bounded by
# what is left of the budget given as $1 (seconds); logs under gpurun_out/q32/.
budget=${1:-720}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/q32; mkdir -p $out
step() {  # step <name> <max seconds> <cmd...>
  name=$1; lim=$2; shift 2
  l=$(left); [ $l -lt 20 ] && { echo "== $name: skipped, $l s left"; return; }
  [ $lim -gt $l ] && lim=$l
  t0=$(date +%s)
  timeout $lim "$@"
  echo "== $name: rc=$? $(( $(date +%s) - t0 )) s"
}
step tests_q32 240 bash -c "python -m pytest tests/test_em_q32_gpu.py -x -q > $out/tests_q32.log 2>&1; tail -15 $out/tests_q32.log"
step bench 300 bash -c "python bench.py --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-6000 $out/bench.json"
step tests_all 420 bash -c "python -m pytest tests -x -q -m gpu --deselect tests/test_em_q32_gpu.py > $out/tests_all.log 2>&1; tail -8 $out/tests_all.log"
B="python bench.py --value-bits 32 --no-q32 --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci"
step prof_stats 150 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- $B > $out/bench_q32_under_rocprof.json 2> $out/prof_stats.err; cut -c1-1500 $out/bench_q32_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc_$c 150 bash -c "rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- $B > /dev/null 2> $out/pmc_$c.err"
done
python - <<'PY'
import csv, glob, json
rows = []
f = glob.glob("gpurun_out/q32/stats/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/q32/q32_bench_C3_kernel_stats.csv", "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows[:5]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/q32/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f: continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c and "k_estep_lane" in r["Kernel_Name"]]
    if vals:
        res[c] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals)}
        with open("gpurun_out/q32/q32_pmc_C3_%s.csv" % c, "w") as fo:
            fo.write("counter,launches,mean_KB\n%s,%d,%.3f\n" % (c, len(vals), res[c]["mean_KB"]))
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["traffic_bytes_per_launch"] = int(1024 * (2.0 * res["FETCH_SIZE"]["mean_KB"] + res["WRITE_SIZE"]["mean_KB"]))
print(json.dumps(res))
json.dump(res, open("gpurun_out/q32/q32_pmc_traffic.json", "w"), indent=1)
PY
echo "== total $(( $(date +%s) - start )) s"
