#!/bin/bash
# Round 4, call E: split rows only where most of a read is far + side passes with their loads issued together; the round kernel
# without scratch (kernel stats + PMC); configs[2] at full size; configs[0] as named.
budget=${1:-600}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04e; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
export RSEM_HIP_TIMING=1 RSEM_WL_CACHE=/dev/shm/rsem_wl
step tests_em 300 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py -q -m gpu > $out/tests_em.log 2>&1; tail -4 $out/tests_em.log"
top() { python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:5]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-46s calls %6s avg %10.1f us total %9.1f ms" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}
for cfg in C2R C3X; do
  rm -rf /tmp/prof_$cfg
  step stats_$cfg 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python bench.py --config $cfg --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/$cfg.json 2> $out/$cfg.err
  top /tmp/prof_$cfg $out/${cfg}_kernel_stats.csv
  rm -rf /tmp/prof_$cfg
done
step legs 300 bash -c "python bench.py --config C2R --legs C3X,C3X30 --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/legs.json 2> $out/legs.err; python -c \"
import json; d=json.load(open('$out/legs.json'))
r=d['roofline']; print('C2R', {k: r.get(k) for k in ('avg_launch_ms','frac','frac_physical')}, d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle'), d['config'].get('units_with_ids_outside_their_window'))
for k, v in d.get('other_configs', {}).items(): print(k, {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','split_rows','units_with_ids_outside_their_window','error')}, v.get('parity_one_step', {}).get('max_rel_diff_counts_vs_oracle'))\""
# ---- the round kernel at a fifth of configs[2]: kernel stats, PMC ---------------------------------------------------------
D5=/tmp/c3fifth; rm -rf $D5
step gen_fifth 120 bash -c "tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1"
step model_stats 100 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s > $out/model.out 2>&1"
grep -E "^\[timing\] rounds" $out/model.out
top /tmp/prof_m $out/model_fifth_kernel_stats.csv
grep -v "^ROUND" $out/model.out > $out/model.tmp; mv $out/model.tmp $out/model.out
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  step pmc_model_$tagp 100 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_pmc_$tagp -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s -q > /dev/null 2> $out/pmc_model_$tagp.err"
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
for rep in 1 2; do
t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/dropin_full_$rep.log 2>&1; echo "new_full_rc $? new_full_s $(el $t)"
grep -E "^\[timing\]" $out/dropin_full_$rep.log | tr '\n' ';'; echo; grep ROUND $out/dropin_full_$rep.log | tail -1
grep -v "^ROUND" $out/dropin_full_$rep.log > $out/dropin_full.tmp; mv $out/dropin_full.tmp $out/dropin_full_$rep.log
done
python - $DF/stat/s.theta <<'PY'
import gzip, sys, numpy as np
a = [np.array(l.split(), float) for l in open(sys.argv[1]).read().split("\n")[1:3]]
b = [np.array(l.split(), float) for l in gzip.open("profiles/r04a_reference_full_size.theta.gz", "rt").read().split("\n")[1:3]]
m = b[0] >= 1e-7
print("full size: theta vs the REFERENCE's own (round 4 call A): max rel diff %.3g (polished %.3g)" % (np.max(np.abs(a[0][m] - b[0][m]) / b[0][m]), np.max(np.abs(a[1][b[1] >= 1e-7] - b[1][b[1] >= 1e-7]) / b[1][b[1] >= 1e-7])))
PY
echo "== binary hand-off (imdName.rsb/)"
t=$(now); tools/bin/temp_to_rsb $DF/temp/s $DF/stat/s 3 > /dev/null; echo "to_rsb_s $(el $t)"
rm -f $DF/temp/s.dat $DF/temp/*.fq
t=$(now); rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/dropin_full_rsb.log 2>&1; echo "new_full_rsb_rc $? new_full_rsb_s $(el $t)"
grep -E "^\[timing\]" $out/dropin_full_rsb.log | tr '\n' ';'; echo
grep -v "^ROUND" $out/dropin_full_rsb.log > $out/dropin_full.tmp; mv $out/dropin_full.tmp $out/dropin_full_rsb.log
rm -rf $DF
echo "== total $(( $(date +%s) - start )) s"
