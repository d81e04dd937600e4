#!/bin/bash
# Round 4, call B: the group-per-read model round kernel (k_model_group) on the GPU for the first time.
#   1. the CLI tests against the reference's goldens / binary (parity of the new kernel, all four model types), then the whole suite;
#   2. the model rounds at a fifth of configs[2]: kernel stats of the default build, of the 4-waves-per-SIMD build
#      (rsem_amd/variants/occ4) and of the thread-per-read kernels it replaces (RSEM_MODEL_KERNELS=alignment), theta of all three compared;
#   3. configs[2] at full size through the program (text inputs): phase times, theta against the reference's own (round 4 call A);
#   4. the bench line with the driver's arguments.
budget=${1:-1500}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04b; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
export RSEM_HIP_TIMING=1
step tests_cli 420 bash -c "python -m pytest tests/test_cli_gpu.py -q -m gpu -k 'matches_reference or binary or (generated_dataset and 50000)' > $out/tests_cli.log 2>&1; tail -15 $out/tests_cli.log"
# ---- model rounds at a fifth of configs[2] ---------------------------------------------------------------------------
D5=/tmp/c3fifth; rm -rf $D5
step gen_fifth 120 bash -c "tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1"
top() { python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows[:9]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-50s calls %6s avg %11.1f us total %9.1f ms" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}
for v in group occ4 alignment; do
  unset RSEM_MODEL_KERNELS LD_LIBRARY_PATH
  [ $v = occ4 ] && export LD_LIBRARY_PATH=$PWD/rsem_amd/variants/occ4
  [ $v = alignment ] && export RSEM_MODEL_KERNELS=alignment
  rm -rf /tmp/prof_$v
  step model_$v 200 bash -c "RSEM_HIP_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s > $out/model_$v.out 2>&1"
  grep -E "^\[timing\] rounds|^ROUND" $out/model_$v.out | sed -n '1,2p;11,13p;$p'
  top /tmp/prof_$v
  python - /tmp/prof_$v $out/model_${v}_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
  cp $D5/stat/s.theta /tmp/theta_$v; cp $D5/stat/s.model /tmp/model_$v
  grep -v "^ROUND" $out/model_$v.out > $out/model_$v.tmp; mv $out/model_$v.tmp $out/model_$v.out
done
unset RSEM_MODEL_KERNELS LD_LIBRARY_PATH
python - <<'PY'
import numpy as np
def th(v): return np.array(open("/tmp/theta_%s" % v).read().split("\n")[1].split(), float)
a = th("alignment")
for v in ("group", "occ4"):
    b = th(v); m = a >= 1e-7
    print("theta %s vs thread-per-read kernels (a fifth of configs[2], to convergence): max rel diff %.3g" % (v, np.max(np.abs(a[m] - b[m]) / a[m])))
def nums(v): return np.array([float(x) for x in open("/tmp/model_%s" % v).read().split()])
a = nums("alignment")
for v in ("group", "occ4"):
    b = nums(v)
    print(".model %s vs thread-per-read kernels: %d numbers, max |diff| / max(|x|, 1e-9) %.3g" % (v, len(a), np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-9)) if len(a) == len(b) else -1))
PY
rm -rf $D5 /tmp/prof_*
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
echo "== the same with the 4-waves-per-SIMD build"
t=$(now); LD_LIBRARY_PATH=$PWD/rsem_amd/variants/occ4 rsem_amd/bin/rsem-run-em $DF/ref 3 $DF/s $DF/temp/s $DF/stat/s -p 64 > $out/dropin_full_occ4.log 2>&1; echo "new_full_occ4_rc $? new_full_occ4_s $(el $t)"
grep -E "^\[timing\] (rounds|main)" $out/dropin_full_occ4.log
grep -v "^ROUND" $out/dropin_full_occ4.log > $out/dropin_full.tmp; mv $out/dropin_full.tmp $out/dropin_full_occ4.log
rm -rf $DF
# ---- the rest of the suite, the bench line -----------------------------------------------------------------------------
step tests_all 600 bash -c "python -m pytest tests -q -m gpu --deselect tests/test_cli_gpu.py::test_generated_dataset_vs_reference_binary > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -12"
step bench_default 700 bash -c "python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err; python -c \"
import json; d=json.load(open('$out/bench_default.json'))
print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','checks')}, indent=None)[:800])
r=d['roofline']; print({k: r.get(k) for k in ('achieved','frac','frac_physical','traffic','frac_of_traffic','avg_launch_ms')}); print(r.get('physical'))
print(json.dumps(d.get('e2e_wall_clock'))[:2500])
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v.get('physical', {}).get('physical_over_pmc') for k, v in d.get('other_configs', {}).items()})\""
echo "== total $(( $(date +%s) - start )) s"
