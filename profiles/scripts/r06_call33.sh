# Round 6, call 33: counters of k_model_group with the position codes (a subset of tools/model_group_pmc.sh's groups): did the instructions go where the time did not?
#!/bin/bash
# Counters of k_model_group (rounds 1-11 of rsem-run-em) at a fifth of configs[2], one rocprofv3 --pmc pass per group (the
# guide: counter passes on their own, with --kernel-trace only).  tools/model_group_pmc.sh <out_dir> [n_reads] [M]
out=${1:-gpurun_out/model_pmc}; N=${2:-10526315}; M=${3:-200000}; D=/tmp/mgp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out; rm -rf $D
tools/bin/gen_temp $D $N $M 3 20250925 100 nosam 5-16 | tail -1
tools/bin/temp_to_rsb $D/temp/s $D/stat/s 3 > /dev/null
export RSEM_HIP_NORMAL_EXIT=1 RSEM_HIP_MAX_ROUND=20 RSEM_HIP_TIMING=1
run() { rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -q; }
rm -rf /tmp/mgp_stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mgp_stats -o p -- rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -q > $out/stats_run.log 2>&1
cp $(find /tmp/mgp_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1)); rm -rf /tmp/mgp_pmc_$i
  timeout 90 rocprofv3 --pmc $group --kernel-trace --output-format csv -d /tmp/mgp_pmc_$i -o p -- rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -q > $out/pmc_$i.log 2>&1
  echo "group $i ($group): rc=$?"
  # (the counters of the round kernel leave the box with every pass: a later pass that hangs must not take them along)
  f=$(find /tmp/mgp_pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|k_model_group" "$f" > $out/counters_$i.csv
  rm -rf /tmp/mgp_pmc_$i
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD
GROUPS
python - $out $i <<'PY'
import csv, glob, json, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(dict)
for i in range(1, n + 1):
    for f in glob.glob(out + "/counters_%d.csv" % i):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for x in csv.DictReader(open(f)):
            k = x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
            if "k_model_group" not in k: continue
            acc[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
        for k, d in acc.items():
            for cn, v in d.items():
                res[k][cn] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + "/model_group_pmc.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for cn in sorted(d): print("   %-44s %16.1f  (%d launches)" % (cn, d[cn]["mean"], d[cn]["launches"]))
PY
grep -E "^\[timing\] rounds|k_model_group" $out/stats_run.log $out/kernel_stats.csv | head
rm -rf $D /tmp/mgp_pmc_* /tmp/mgp_stats
