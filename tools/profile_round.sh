#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [configs="C3 C2"]     -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
# Kernel-trace/stats and each PMC counter are separate runs (MI355X_MICROARCH.md, HBM/rocprofv3 section).
set -u
tag=${1:-rXX}; cfgs=${2:-"C3 C2"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for cfg in $cfgs; do
  B="python bench.py --config $cfg --legs= --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci --no-q32"  # (--no-q32: the Q32 leg launches the same kernel name on other planes)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_bench_$cfg -o b -- $B > $out/${tag}_bench_$cfg.json 2> $out/${tag}_bench_$cfg.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${cfg}_$c -o p -- $B > /dev/null 2> $out/${tag}_pmc_${cfg}_$c.err
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_gibbs -o g -- python tools/gibbs_profile.py 1.0 60 > $out/${tag}_gibbs.log 2>&1
python - "$tag" $cfgs <<'PY'
import csv, glob, json, sys
tag, cfgs = sys.argv[1], sys.argv[2:]
def stats(pat):
    f = glob.glob(pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
for name in ["bench_" + c for c in cfgs] + ["gibbs"]:
    rows = stats("gpurun_out/%s_%s/**/*kernel_stats.csv" % (tag, name))
    with open("gpurun_out/%s_%s_kernel_stats.csv" % (tag, name), "w") as fo:
        if rows:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows[:6]:
        print(name, r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
traffic = {}
for cfg in cfgs:
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/%s_pmc_%s_%s/**/*counter_collection.csv" % (tag, cfg, c), recursive=True)
        if not f: continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c and "k_estep_lane" in r["Kernel_Name"]]
        if vals: res[c] = {"launches": len(vals), "mean_KB": sum(vals) / len(vals)}
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        # FETCH_SIZE counts 64 B per 128 B request on wide streaming reads on gfx950: x2 (the guide; calibrated in round 1 with
        # tools/microbench/stream.hip: 1 GiB read -> 524298.5 KB, 1 GiB written -> 1048576 KB)
        traffic[cfg] = {"kernel": "k_estep_lane", "launches_sampled": res["FETCH_SIZE"]["launches"], "FETCH_SIZE_KB_raw_mean": res["FETCH_SIZE"]["mean_KB"],
                        "FETCH_SIZE_correction": 2.0, "WRITE_SIZE_KB_mean": res["WRITE_SIZE"]["mean_KB"],
                        "traffic_bytes_per_launch": int(1024 * (2.0 * res["FETCH_SIZE"]["mean_KB"] + res["WRITE_SIZE"]["mean_KB"])),
                        "source": "profiles/%s_pmc_%s_*.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; tools/profile_round.sh)" % (tag, cfg)}
json.dump(traffic, open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
for cfg in $cfgs; do tail -1 $out/${tag}_bench_$cfg.json | cut -c1-400; done
tail -3 $out/${tag}_gibbs.log
