cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do
export RSEM_HIP_DBG=$d
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd$d -o b -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-gibbs --no-ci > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pd$d/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_mstep_fast" in r["Name"] or "k_estep_lane" in r["Name"]: print("dbg $d", r["Name"][22:36], r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
