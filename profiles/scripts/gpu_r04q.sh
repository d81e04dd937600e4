#!/bin/bash
# Round 4, call Q: counters on the two side passes of the split rows (C2R), then the bench line with the driver's arguments
# on the final tree (the reference pinned to one socket).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04q; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
start=$(date +%s)
B="python bench.py --config C2R --legs= --steps 10 --warmup 2 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_q_$tagp -o p -- $B > /dev/null 2> $out/pmc_$tagp.err; echo "== pmc $tagp rc=$? $(( $(date +%s) - start )) s"
done
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in glob.glob("/tmp/prof_q_*"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f[0])):
        k = x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:24]
        if not ("k_far_" in k or "k_estep_lane" in k): continue
        acc[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, dd in acc.items():
        for c, v in dd.items():
            res[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + "/C2R_side_passes_pmc.json", "w"), indent=1)
for k, dd in res.items():
    print(k, {c: round(v["mean"], 1) for c, v in dd.items()})
PY
rm -rf /tmp/prof_q_*
timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "== bench rc=$? $(( $(date +%s) - start )) s"
python - $out/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "frac_physical", "avg_launch_ms")}, d["value"], d["ms_per_step"])
c = d.get("cpu_baseline") or {}; print({k: c.get(k) for k in ("value", "cores", "kind", "ms_per_round", "pinned_cpus")}); print((c.get("sample") or "")[:300])
e = d.get("e2e_wall_clock") or {}; print(json.dumps(e.get("measured"))[:600]); fs = e.get("full_size") or {}; print({k: fs.get(k) for k in ("dropin_s", "dropin_rounds", "reference_s", "speedup")})
print({k: {kk: v.get(kk) for kk in ("estep_avg_launch_ms", "frac", "frac_physical", "error")} for k, v in d.get("other_configs", {}).items()})
PY
echo "== total $(( $(date +%s) - start )) s"
