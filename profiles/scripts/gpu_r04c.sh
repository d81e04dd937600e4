#!/bin/bash
# Round 4, call C: split rows (the transposed pass for alignments outside a read's window) on the GPU for the first time, the round
# kernel writing the value planes in place, the text parsers without intermediate vectors, evidence for profiles/.
#   1. tests: EM / Q32 / sharded / CLI (parity of everything that changed);
#   2. E-step legs C2R, C3X, C3X30 with and without split rows (RSEM_HIP_SPLIT=0), C3 for reference;
#   3. the model rounds at a fifth of configs[2]: kernel stats + SQ counters of k_model_group;
#   4. configs[2] at full size through the program, theta against the reference's own;
#   5. rocprofv3 --kernel-trace --stats and the two PMC passes of the bench command at C3.
budget=${1:-900}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04c; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
export RSEM_HIP_TIMING=1 RSEM_WL_CACHE=/dev/shm/rsem_wl
step tests_em 400 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_dist_gpu.py -q -m gpu > $out/tests_em.log 2>&1; tail -12 $out/tests_em.log"
step tests_cli 300 bash -c "python -m pytest tests/test_cli_gpu.py -q -m gpu -k 'matches_reference or binary' > $out/tests_cli.log 2>&1; tail -6 $out/tests_cli.log"
legs() {  # tag, env assignment
  step legs_$1 400 bash -c "$2 python bench.py --config C2R --legs C3X,C3X30 --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/legs_$1.json 2> $out/legs_$1.err; tail -2 $out/legs_$1.err; python -c \"
import json; d=json.load(open('$out/legs_$1.json'))
r=d['roofline']; print('C2R', {k: r.get(k) for k in ('avg_launch_ms','frac','frac_physical')}, d['checks']['parity_one_step'], d['config'].get('units_with_ids_outside_their_window'))
for k, v in d.get('other_configs', {}).items(): print(k, {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','split_rows','units_with_ids_outside_their_window','error')}, v.get('parity_one_step'))\""
}
legs split ""
legs whole "RSEM_HIP_SPLIT=0"
# ---- model rounds at a fifth of configs[2] ---------------------------------------------------------------------------
D5=/tmp/c3fifth; rm -rf $D5
step gen_fifth 120 bash -c "tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1"
step model_stats 200 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s > $out/model.out 2>&1"
grep -E "^\[timing\] rounds" $out/model.out
python - /tmp/prof_m $out/model_fifth_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:8]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-50s calls %6s avg %11.1f us total %9.1f ms" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep -v "^ROUND" $out/model.out > $out/model.tmp; mv $out/model.tmp $out/model.out
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  step pmc_model_$tagp 200 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_pmc_$tagp -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s -q > /dev/null 2> $out/pmc_model_$tagp.err"
done
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for tagp in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
    f = glob.glob("/tmp/prof_pmc_%s/**/*counter_collection.csv" % tagp, recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f[0])):
        k = x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        if "k_model_group" not in k: continue
        acc[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + "/model_group_pmc_fifth_size.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v["mean"], 1) for c, v in d.items()})
PY
rm -rf $D5 /tmp/prof_m /tmp/prof_pmc_*
# ---- configs[2] at full size ----------------------------------------------------------------------------------------
DF=/tmp/c3full; rm -rf $DF
step gen_full 200 bash -c "tools/bin/gen_temp $DF 52631578 200000 3 20250925 100 nosam 5-16 | tail -1"
t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/dropin_full.log 2>&1; echo "new_full_rc $? new_full_s $(el $t)"
grep -E "^\[timing\]" $out/dropin_full.log; grep ROUND $out/dropin_full.log | sed -n '11,12p;$p'
python - $DF/stat/s.theta <<'PY'
import gzip, sys, numpy as np
a = [np.array(l.split(), float) for l in open(sys.argv[1]).read().split("\n")[1:3]]
b = [np.array(l.split(), float) for l in gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1:3]]
m = b[0] >= 1e-7
print("full size: theta vs the REFERENCE's own (round 4 call A): max rel diff %.3g (polished %.3g)" % (np.max(np.abs(a[0][m] - b[0][m]) / b[0][m]), np.max(np.abs(a[1][b[1] >= 1e-7] - b[1][b[1] >= 1e-7]) / b[1][b[1] >= 1e-7])))
PY
grep -v "^ROUND" $out/dropin_full.log > $out/dropin_full.tmp; mv $out/dropin_full.tmp $out/dropin_full.log
rm -rf $DF
# ---- the bench command at C3 under rocprofv3: kernel stats, then the two PMC passes ---------------------------------------
B="python bench.py --config C3 --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream"
step stats_C3 150 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3_stats -o s -- $B > $out/C3_stats.out 2> $out/C3_stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc_C3_$c 150 bash -c "rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/c3_pmc_$c -o p -- $B > /dev/null 2> $out/C3_pmc_$c.err"
done
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
kern = "k_estep_lane<true, true>"
r = {}
f = glob.glob("/tmp/c3_stats/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("%s/C3_kernel_stats.csv" % out, "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for row in rows:
        if kern in row["Name"]:
            r["kernel"] = {"name": row["Name"][:100], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/c3_pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f: continue
    rows = [x for x in csv.DictReader(open(f[0])) if x["Counter_Name"] == c and kern in x["Kernel_Name"]]
    vals = [float(x["Counter_Value"]) for x in rows]
    if vals:
        r[c] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals)}
        keep = rows[::max(1, len(rows) // 100)]
        with open("%s/pmc_C3_%s.csv" % (out, c), "w") as fo:
            w = csv.DictWriter(fo, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], extrasaction="ignore"); w.writeheader()
            for x in keep:
                x = dict(x); x["Kernel_Name"] = x["Kernel_Name"][:60]; w.writerow(x)
if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
    r["traffic_bytes_per_launch"] = int(1024 * (2.0 * r["FETCH_SIZE"]["mean_KB"] + r["WRITE_SIZE"]["mean_KB"]))
try:
    d = json.loads(open("%s/C3_stats.out" % out).read().strip().split("\n")[-1])
    r["bench_under_rocprof"] = {"avg_launch_ms": d["roofline"]["avg_launch_ms"], "ms_per_step": d["ms_per_step"], "frac": d["roofline"]["frac"],
                                "frac_physical": d["roofline"]["frac_physical"], "physical_bytes": d["roofline"]["physical"]["physical_bytes_per_launch"]}
except Exception as e:
    r["bench_error"] = str(e)
json.dump(r, open("%s/C3_summary.json" % out, "w"), indent=1)
print(json.dumps(r, indent=1))
PY
rm -rf /tmp/c3_stats /tmp/c3_pmc_*
echo "== total $(( $(date +%s) - start )) s"
