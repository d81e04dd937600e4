#!/bin/bash
# Round 4, call A: BASELINE configs[2] AS NAMED against the reference itself (VERDICT r03 "next" 1a).
#   * generate the full-size input (50 M alignable read pairs, 200 k transcripts, 30 GB of text);
#   * the drop-in rsem-run-em on it (wall clock, ROUND lines, theta);
#   * the UNMODIFIED reference rsem-run-em -p 64 on the same files, to convergence (about half an hour), with the arrival
#     times of its ROUND lines;
#   * while the reference runs: side jobs on the GPU, pinned to 32 hardware threads the reference is kept off
#     (taskset on both sides), so that the half hour of GPU time is not wasted: profiles of the model rounds at full size,
#     PMC passes, exact Gibbs against the reference binary at 5 M pairs, the new configs[0] CLI test.  The ROUND arrival
#     log shows whether the reference's rate differs while side jobs run and after they have ended.
budget=${1:-2700}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04a; mkdir -p $out
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
{ nproc; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|MHz" ; free -g | head -2; df -h /tmp | tail -1; } > $out/host.txt 2>&1
NCPU=$(nproc)
# side jobs: the last 16 cores of the second socket and their SMT siblings (standard Linux numbering: n/2.. are the siblings)
if [ $NCPU -ge 256 ]; then SIDE="112-127,240-255"; REFSET="0-111,128-239"; else SIDE="$((NCPU*3/4))-$((NCPU-1))"; REFSET="0-$((NCPU*3/4-1))"; fi
echo "side jobs on CPUs $SIDE, reference on $REFSET" >> $out/host.txt
P=${REF_P:-64}; N1=${N1:-50000000}; M=200000
NF=$(( N1 * 20 / 19 ))
DF=/tmp/c3full
rm -rf $DF
echo "== generate full size"; t=$(now); tools/bin/gen_temp $DF $NF $M 3 20250925 100 nosam 5-16 | tail -1; echo "gen_full_s $(el $t)"; du -sh $DF | cut -f1
t=$(now); oracle/_ref/rsem-build-read-index 32 1 1 $DF/temp/s_alignable_1.fq $DF/temp/s_alignable_2.fq > /dev/null; echo "build_read_index_s $(el $t)"
# a second directory with the same inputs (symlinks) for every program that runs beside the reference: outputs stay apart
mkview() { local V=$1; rm -rf $V; mkdir -p $V/temp $V/stat; for f in $DF/ref.*; do ln -s $f $V/; done
  for f in $DF/temp/*; do ln -s $f $V/temp/; done; for f in $DF/stat/*; do ln -s $f $V/stat/; done; }
export RSEM_HIP_TIMING=1
echo "== drop-in, full size, text inputs (alone on the host)"
mkview /tmp/c3new; t=$(now)
rsem_amd/bin/rsem-run-em /tmp/c3new/ref 3 /tmp/c3new/s /tmp/c3new/temp/s /tmp/c3new/stat/s -p $P > $out/dropin_full.log 2>&1; echo "new_full_rc $? new_full_s $(el $t)"
grep -E "^\[timing\]" $out/dropin_full.log; grep ROUND $out/dropin_full.log | tail -1
grep ROUND $out/dropin_full.log | gzip > $out/dropin_full_rounds.txt.gz
cp /tmp/c3new/stat/s.theta /tmp/dropin_full.theta; gzip -c /tmp/dropin_full.theta > $out/dropin_full.theta.gz
grep -v -E "^ROUND" $out/dropin_full.log > $out/dropin_full.log.tmp; mv $out/dropin_full.log.tmp $out/dropin_full.log
echo "== reference -p $P, full size (CPUs $REFSET)"
( t=$(now); taskset -c $REFSET oracle/_ref/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p $P > /tmp/ref_full.log 2>&1; echo "ref_full_rc $? ref_full_s $(el $t)" > /tmp/ref_full.time ) &
REFPID=$!
( t0=$(now); while kill -0 $REFPID 2>/dev/null; do r=$(grep -c "^ROUND" /tmp/ref_full.log 2>/dev/null); echo "$(el $t0) $r $(cat /tmp/side_state 2>/dev/null)"; sleep 2; done ) > $out/ref_full_progress.txt &
MONPID=$!
# ---- side jobs (GPU), pinned ----------------------------------------------------------------------------------
side() { local name=$1 lim=$2; shift 2; echo "$name" > /tmp/side_state; local t0=$(date +%s)
  taskset -c $SIDE timeout $lim "$@"; echo "== side $name: rc=$? $(( $(date +%s) - t0 )) s"; echo idle > /tmp/side_state; }
stats_top() { python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
with open(sys.argv[2], "w") as fo:
    if rows:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-50s calls %6s avg %11.1f us total %9.1f ms" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}
echo side_start > /tmp/side_state
# (1) model rounds at FULL size under rocprofv3 --kernel-trace --stats
mkview /tmp/c3prof
side model_stats_full 400 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_model_full -o p -- rsem_amd/bin/rsem-run-em /tmp/c3prof/ref 3 /tmp/c3prof/s /tmp/c3prof/temp/s /tmp/c3prof/stat/s -q > $out/model_stats_full.out 2>&1"
stats_top /tmp/prof_model_full $out/model_full_kernel_stats.csv
# (2) PMC passes of the model kernels on a fifth of the reads (bytes scale with the reads; 5 767 serialised dispatches of the
#     device loop at full size would take the side jobs' time)
D5=/tmp/c3fifth; rm -rf $D5
side gen_fifth 200 tools/bin/gen_temp $D5 $(( NF / 5 )) $M 3 20250925 100 nosam 5-16
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  side pmc_model_$tagp 300 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_model_pmc_$tagp -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s -q > /dev/null 2> $out/pmc_model_$tagp.err"
done
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for tagp in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
    f = glob.glob("/tmp/prof_model_pmc_%s/**/*counter_collection.csv" % tagp, recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f[0])):
        k = x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        if "estep_lane" in k or "solo_close" in k: continue
        acc[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + "/model_pmc_fifth_size.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v["mean"], 1) for c, v in d.items()}, "launches", next(iter(d.values()))["launches"])
PY
rm -rf /tmp/prof_model_pmc_* /tmp/prof_model_full
# (3) exact Gibbs against the REFERENCE binary at 5.26 M pairs, 8 chains, the pipeline's 200 / 1000 / 1
G=/tmp/g5m
side gibbs_em 300 bash -c "rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s --gibbs-out -q > /dev/null 2>&1"
rm -f $D5/temp/*.fq $D5/temp/s.dat
gr() { local name=$1 prog=$2; shift 2; rm -rf $G.$name; mkdir -p $G.$name; cp -r $D5/ref.* $D5/temp $D5/stat $G.$name/ 2>/dev/null
  local t0=$(now); "$@" $prog $G.$name/ref $G.$name/temp/s $G.$name/stat/s 200 1000 1 -p 8 --seed 5 -q ${GX:-} > $G.$name.log 2>&1; echo "gibbs_$name rc=$? wall=$(el $t0) s"; }
echo gibbs_ref_and_exact > /tmp/side_state
( gr ref oracle/_ref/rsem-run-gibbs taskset -c $SIDE timeout 1500 ) &
GREF=$!
GX="--gibbs-mode exact" gr exact rsem_amd/bin/rsem-run-gibbs taskset -c $SIDE timeout 1500
wait $GREF
same=0; diff=0
for k in 0 1 2 3 4 5 6 7; do if cmp -s $G.ref/temp/s.countvectors$k $G.exact/temp/s.countvectors$k; then same=$((same+1)); else diff=$((diff+1)); fi; done
echo "exact Gibbs vs reference binary, $(( NF / 5 * 19 / 20 )) pairs, -p 8, 200/1000/1, seed 5: $same count-vector files byte-equal, $diff differ ($(stat -c %s $G.ref/temp/s.countvectors0) bytes each)"
tail -3 $G.exact.log
rm -rf $G.* $D5
echo idle > /tmp/side_state
# (4) the configs[0] / type-2 CLI tests added this round, and the rest of that test
side cli_generated 900 bash -c "python -m pytest tests/test_cli_gpu.py -q -m gpu -k generated_dataset > $out/pytest_generated.log 2>&1; tail -3 $out/pytest_generated.log"
# (5) kernel stats + PMC traffic of the E step at C2 and C5 on HEAD (VERDICT weak 3): separate passes, --no-q32 so that
#     one variant of k_estep_lane is what the counters see
for cfg in C2 C5; do
  B="python bench.py --config $cfg --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream"
  side stats_$cfg 400 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/es_${cfg}_stats -o s -- $B > $out/${cfg}_stats.out 2> $out/${cfg}_stats.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    side pmc_${cfg}_$c 400 bash -c "rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/es_${cfg}_pmc_$c -o p -- $B > /dev/null 2> $out/${cfg}_pmc_$c.err"
  done
done
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for name in ("C2", "C5"):
    kern = "k_estep_lane<true, true>"
    r = {}
    f = glob.glob("/tmp/es_%s_stats/**/*kernel_stats.csv" % name, recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        with open("%s/%s_kernel_stats.csv" % (out, name), "w") as fo:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        for row in rows:
            if kern in row["Name"]:
                r["kernel"] = {"name": row["Name"][:100], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("/tmp/es_%s_pmc_%s/**/*counter_collection.csv" % (name, c), recursive=True)
        if not f: continue
        rows = [x for x in csv.DictReader(open(f[0])) if x["Counter_Name"] == c and kern in x["Kernel_Name"]]
        vals = [float(x["Counter_Value"]) for x in rows]
        if vals:
            r[c] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals)}
            keep = rows[::max(1, len(rows) // 100)]
            with open("%s/pmc_%s_%s.csv" % (out, name, c), "w") as fo:
                w = csv.DictWriter(fo, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], extrasaction="ignore"); w.writeheader()
                for x in keep:
                    x = dict(x); x["Kernel_Name"] = x["Kernel_Name"][:60]; w.writerow(x)
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        r["traffic_bytes_per_launch"] = int(1024 * (2.0 * r["FETCH_SIZE"]["mean_KB"] + r["WRITE_SIZE"]["mean_KB"]))
    try:
        d = json.loads(open("%s/%s_stats.out" % (out, name)).read().strip().split("\n")[-1])
        r["bench_under_rocprof"] = {"avg_launch_ms": d["roofline"]["avg_launch_ms"], "ms_per_step": d["ms_per_step"], "layout": d.get("layout")}
    except Exception as e:
        r["bench_error"] = str(e)
    res[name] = r
json.dump(res, open("%s/estep_pmc_c2_c5.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf /tmp/es_*
echo side_done > /tmp/side_state
echo "== side jobs done after $(( $(date +%s) - start )) s"
# ---- wait for the reference --------------------------------------------------------------------------------------
while kill -0 $REFPID 2>/dev/null; do
  if [ $(left) -lt 60 ]; then echo "budget: stopping the reference"; pkill -P $REFPID; kill $REFPID; break; fi
  sleep 5
done
wait $REFPID 2>/dev/null; kill $MONPID 2>/dev/null
cat /tmp/ref_full.time; grep ROUND /tmp/ref_full.log | tail -1; grep "Time Used" /tmp/ref_full.log
grep ROUND /tmp/ref_full.log | gzip > $out/ref_full_rounds.txt.gz
grep -v -E "^ROUND" /tmp/ref_full.log | tail -20 > $out/ref_full.log
[ -f $DF/stat/s.theta ] && gzip -c $DF/stat/s.theta > $out/ref_full.theta.gz
python - $out <<'PY'
import gzip, json, sys, numpy as np
out = sys.argv[1]
res = {}
try:
    a = [np.array(l.split(), float) for l in gzip.open(out + "/dropin_full.theta.gz", "rt").read().split("\n")[1:3]]
    b = [np.array(l.split(), float) for l in gzip.open(out + "/ref_full.theta.gz", "rt").read().split("\n")[1:3]]
    m = b[0] >= 1e-7
    res["theta_max_rel_diff_full"] = float(np.max(np.abs(a[0][m] - b[0][m]) / b[0][m]))
    m = b[1] >= 1e-7
    res["theta_polished_max_rel_diff_full"] = float(np.max(np.abs(a[1][m] - b[1][m]) / b[1][m]))
except Exception as e:
    res["theta_error"] = str(e)
pr = [l.split() for l in open(out + "/ref_full_progress.txt") if len(l.split()) >= 2]
t1 = next((float(x[0]) for x in pr if int(x[1]) >= 1), None); t11 = next((float(x[0]) for x in pr if int(x[1]) >= 11), None)
if t1 and t11:
    res.update({"ref_startup_s": t1, "ref_rounds1_11_s": t11 - t1, "ref_late_s": float(pr[-1][0]) - t11, "ref_late_rounds": int(pr[-1][1]) - 11})
    # the reference's late-round rate while side jobs ran vs after they had ended
    busy = [(float(x[0]), int(x[1])) for x in pr if int(x[1]) > 11 and len(x) > 2 and x[2] not in ("idle", "side_done")]
    free = [(float(x[0]), int(x[1])) for x in pr if int(x[1]) > 11 and len(x) > 2 and x[2] == "side_done"]
    for name, seg in (("with_side_jobs", busy), ("alone", free)):
        if len(seg) > 5 and seg[-1][1] > seg[0][1]:
            res["ref_ms_per_late_round_" + name] = 1e3 * (seg[-1][0] - seg[0][0]) / (seg[-1][1] - seg[0][1])
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
rm -rf $DF /tmp/c3new /tmp/c3prof
echo "== total $(( $(date +%s) - start )) s"
