#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
# Kernel-trace/stats and each PMC counter are separate runs (MI355X_MICROARCH.md, HBM/rocprofv3 section).
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-gibbs --no-ci"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_bench -o b -- $B > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -o p -- $B > /dev/null 2> $out/${tag}_pmc_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_gibbs -o g -- python tools/gibbs_profile.py 1.0 60 > $out/${tag}_gibbs.log 2>&1
python - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
def stats(pat):
    f = glob.glob(pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
for name in ("bench", "gibbs"):
    rows = stats("gpurun_out/%s_%s/**/*kernel_stats.csv" % (tag, name))
    with open("gpurun_out/%s_%s_kernel_stats.csv" % (tag, name), "w") as fo:
        if rows:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows[:8]:
        print(name, r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (tag, c), recursive=True)
    if not f: continue
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            per.setdefault(next((k for k in ("k_estep_lane", "k_mstep_fused", "k_sample_z_lane") if k in r["Kernel_Name"]), "other"), []).append(float(r["Counter_Value"]))
    res[c] = {k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in per.items()}
json.dump(res, open("gpurun_out/%s_pmc_summary.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
tail -3 $out/${tag}_gibbs.log; cat $out/${tag}_bench.json | tail -1 | cut -c1-600
