#!/bin/bash
# Round 4, call D: where does a round with split rows spend its time?  Per-kernel times (rocprofv3 --kernel-trace --stats) of the
# E-step at C3X and C2R with split rows, for three builds of the lane kernel (the reciprocal stored by every lane of a read /
# by its first lane only / not at all), and with whole rows; the round kernel after its register diet at a fifth of configs[2].
budget=${1:-500}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04d; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
export RSEM_WL_CACHE=/dev/shm/rsem_wl
top() { python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:7]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-46s calls %6s avg %10.1f us total %9.1f ms" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}
run() {  # tag config env...
  local tag=$1 cfg=$2; shift 2
  rm -rf /tmp/prof_$tag
  step $tag 150 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --config $cfg --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/$tag.json 2> $out/$tag.err
  python -c "
import json; d=json.loads(open('$out/$tag.json').read().strip().split('\n')[-1]); r=d['roofline']
print('   $tag: launch %.4f ms, step %.4f ms, parity %s' % (r['avg_launch_ms'], d['ms_per_step'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle')))"
  top /tmp/prof_$tag $out/${tag}_kernel_stats.csv
  rm -rf /tmp/prof_$tag
}
run C3X_split C3X A=1
run C3X_split_g0 C3X RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xg0.so
run C3X_split_nostore C3X RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xnostore.so
run C3X_whole C3X RSEM_HIP_SPLIT=0
run C2R_split C2R A=1
run C2R_whole C2R RSEM_HIP_SPLIT=0
run C3_plain C3 A=1
# ---- the round kernel after the register diet ---------------------------------------------------------------------------
D5=/tmp/c3fifth; rm -rf $D5
step gen_fifth 120 bash -c "tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1"
for v in inplace scatter; do
  [ $v = scatter ] && export RSEM_MODEL_PLANES=0
  step model_$v 100 bash -c "RSEM_HIP_TIMING=1 RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m_$v -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s > $out/model_$v.out 2>&1"
  grep -E "^\[timing\] rounds" $out/model_$v.out
  top /tmp/prof_m_$v $out/model_${v}_kernel_stats.csv
  grep -v "^ROUND" $out/model_$v.out > $out/model.tmp; mv $out/model.tmp $out/model_$v.out
done
unset RSEM_MODEL_PLANES
rm -rf $D5 /tmp/prof_m_*
echo "== total $(( $(date +%s) - start )) s"
