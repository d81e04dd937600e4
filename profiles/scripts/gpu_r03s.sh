#!/bin/bash
# Round 3, last evidence call (after the pipelined-waits work): the whole -m gpu suite, rocprofv3 kernel stats + PMC passes of the final kernels at C3
# (F64 E step, Q32 E step, Gibbs sweep), configs[3] through the programs with the binary hand-offs, the default bench line.
budget=${1:-1300}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
tag=r03s
out=gpurun_out/$tag; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_all 420 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -8"
B="python bench.py --config C3 --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream"
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
for name, kern in (("C3_f64", "k_estep_lane<true, true>"), ("C3_q32", "k_estep_lane<true, true>"), ("C3_gibbs", "k_sample_z_lane")):
    r = {}
    f = glob.glob("%s/%s_stats/**/*kernel_stats.csv" % (out, name), recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        with open("%s/%s_kernel_stats.csv" % (out, name), "w") as fo:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        for row in rows:
            if kern in row["Name"]:
                r["kernel"] = {"name": row["Name"][:100], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/%s_pmc_%s/**/*counter_collection.csv" % (out, name, c), recursive=True)
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
    res[name] = r
json.dump(res, open("%s/summary.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
PY
for n in C3_f64 C3_q32; do python -c "
import json; d=json.loads(open('$out/${n}_stats.out').read().strip().split('\n')[-1]); r=d['roofline']; print('$n under rocprof: HIP events %.4f ms, step %.4f, frac %.4f' % (r['avg_launch_ms'], d['ms_per_step'], r['frac']))"; done; tail -1 $out/C3_gibbs_stats.out
find $out -name '*_stats' -type d -exec rm -rf {} + 2>/dev/null; find $out -name '*_pmc_*' -type d -exec rm -rf {} + 2>/dev/null
step bench_default 700 bash -c "python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err; python -c \"
import json; d=json.load(open('$out/bench_default.json'))
print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','roofline','checks')}, indent=None)[:2500])
print(json.dumps(d.get('gibbs'))[:1800])
print(json.dumps(d.get('e2e_wall_clock'))[:1800])
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','error','generate_s')} for k, v in d.get('other_configs', {}).items()})\""
step e2e_c4 420 bash -c "tools/e2e_c4.sh > $out/e2e_c4.log 2>&1; cat $out/e2e_c4.log | cut -c1-300"
# the DEFAULT mode (auto -> the reference's chains at this size) on a fifth of configs[3]'s reads: 10 M pairs, 8 chains
step e2e_c4_fifth_default 480 bash -c "GIBBS_MODE=auto tools/e2e_c4.sh 10000000 200000 8 > $out/e2e_c4_fifth_default_mode.log 2>&1; cat $out/e2e_c4_fifth_default_mode.log | cut -c1-300"
echo "== total $(( $(date +%s) - start )) s"
