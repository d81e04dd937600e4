#!/bin/bash
# Round 3, first GPU call: (1) the whole -m gpu suite incl. the Perl-pipeline test, (2) rocprofv3 kernel stats + the two PMC
# passes on HEAD's kernels at C3 (F64 E step, Q32 E step, PARALLEL Gibbs sweep), (3) the device STREAM probe, (4) every
# prepared library variant on the same cached workload (tools/build_prepared_variants.sh), (5) the > 2^32 alignments test.
#   tools/build_prepared_variants.sh && gpurun --timeout 1500 -- 'tools/gpu_r03a.sh 1440'
budget=${1:-1400}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
tag=r03a
out=gpurun_out/$tag; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }

step tests_all 420 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -8"

B="python bench.py --config C3 --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream"
step stream 40 bash -c "python -c \"
from rsem_amd import capi
print('stream probe read/copy GB/s: %.0f %.0f' % capi.stream_probe(0, 8 << 30, 5))\" | tee $out/stream.log"
prof() {  # name, command
  local name=$1; shift
  step stats_$name 120 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -o s -- $* > $out/${name}_stats.out 2> $out/${name}_stats.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_${name}_$c 120 bash -c "rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${name}_pmc_$c -o p -- $* > /dev/null 2> $out/${name}_pmc_$c.err"
  done
}
prof C3_f64 $B
prof C3_q32 $B --value-bits 32
prof C3_gibbs python tools/gibbs_profile.py 1.0 40 C3
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for name, kern in (("C3_f64", "k_estep_lane"), ("C3_q32", "k_estep_lane"), ("C3_gibbs", "k_sample_z_lane")):
    r = {}
    f = glob.glob("%s/%s_stats/**/*kernel_stats.csv" % (out, name), recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        with open("%s/%s_kernel_stats.csv" % (out, name), "w") as fo:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        for row in rows:
            if kern in row["Name"]:
                r.setdefault("kernels", []).append({"name": row["Name"][:120], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/%s_pmc_%s/**/*counter_collection.csv" % (out, name, c), recursive=True)
        if not f: continue
        vals = [float(x["Counter_Value"]) for x in csv.DictReader(open(f[0])) if x["Counter_Name"] == c and kern in x["Kernel_Name"]]
        if vals:
            # the first launches of a process include warm-up / other template instantiations: keep the steady ones (median-ish)
            vals.sort(); mid = vals[len(vals) // 4: len(vals) - len(vals) // 4] or vals
            r[c] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals), "mid_half_mean_KB": sum(mid) / len(mid)}
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        r["traffic_bytes_per_launch"] = int(1024 * (2.0 * r["FETCH_SIZE"]["mean_KB"] + r["WRITE_SIZE"]["mean_KB"]))
    res[name] = r
json.dump(res, open("%s/summary.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
for n in C3_f64 C3_q32; do tail -1 $out/${n}_stats.out | cut -c1-600; done; tail -2 $out/C3_gibbs_stats.out

# library variants: EM (bench: C3 F64 headline + Q32 leg + C2 leg) and Gibbs (sweep time at C3)
BV="python bench.py --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline --no-stream"
for v in default rcp dpp ds clamp fma nt magic neff all allm g1; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  extra=""; [[ "$v" == g1* ]] && extra="--lane-policy 1"
  step bench_$v 90 bash -c "RSEM_HIP_LIB=$lib $BV $extra > $out/bench_$v.json 2> $out/bench_$v.err; python - $out/bench_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q, c2 = d['q32_value_planes'], d['other_configs']['C2']
print('%-7s C3 f64 estep %.4f ms step %.4f | q32 estep %.4f ms step %.4f dtheta %.2e || C2 f64 estep %.4f step %.4f | q32 estep %.4f step %.4f' % (sys.argv[2],
    d['roofline']['avg_launch_ms'], d['ms_per_step'], q['estep_avg_launch_ms'], q['ms_per_step'], q['theta_max_rel_diff_vs_f64_after_20_rounds'],
    c2['estep_avg_launch_ms'], c2['ms_per_step'], c2['q32_value_planes']['estep_avg_launch_ms'], c2['q32_value_planes']['ms_per_step']))
PY"
done
for v in default gsa gsap; do
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  [ -f $lib ] || { echo "== $v: no library"; continue; }
  step gibbs_$v 90 bash -c "RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 40 C3 2>&1 | tail -1 | tee $out/gibbs_$v.log; RSEM_HIP_LIB=$lib python tools/gibbs_profile.py 1.0 60 C2 2>&1 | tail -1 | tee -a $out/gibbs_$v.log"
done
step tests_xl 480 bash -c "RSEM_TEST_XL=1 python -m pytest tests/test_em_gpu.py -q -k more_than_2_to_32 > $out/tests_xl.log 2>&1; tail -3 $out/tests_xl.log"
find $out -name '*counter_collection.csv' -size +8M -delete
echo "== total $(( $(date +%s) - start )) s"
