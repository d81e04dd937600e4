#!/bin/bash
# One gpurun call: where does the E step's time go?  (a) bench runs of library variants that leave one component out
# (tools/build_variants.sh -DRSEM_DIAG=bits; results meaningless, times comparable), (b) SQ counters of the real kernel.
#   tools/gpu_diag.sh <budget seconds> <scale> v1 v2 ... -- v5 v6 ...     (PMC passes run at the "--")
budget=${1:-280}; scale=${2:-0.4}; shift 2
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/diag; mkdir -p $out
step() { name=$1; lim=$2; shift 2; l=$(left); [ $l -lt 15 ] && { echo "== $name: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  t0=$(date +%s); timeout $lim "$@"; echo "== $name: rc=$? $(( $(date +%s) - t0 )) s"; }
B="python bench.py --config C3 --scale $scale --legs= --no-gibbs --no-ci --no-cpu-baseline --steps 20 --warmup 5"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE"
for v in "$@"; do
  if [ "$v" == "--" ]; then
    for bits in 64 32; do for p in 1 2; do
      pm=$P1; [ $p == 2 ] && pm=$P2
      step pmc_${bits}_$p 60 bash -c "rocprofv3 --pmc $pm --kernel-trace --output-format csv -d $out/pmc_${bits}_$p -o p -- $B --value-bits $bits --no-q32 > /dev/null 2> $out/pmc_${bits}_$p.err"
    done; done
    continue
  fi
  lib=$PWD/rsem_amd/librsem_hip.so; [ "$v" != default ] && lib=$PWD/rsem_amd/librsem_hip_$v.so
  step bench_$v 60 bash -c "RSEM_HIP_LIB=$lib $B > $out/bench_$v.json 2> $out/bench_$v.err; python - $out/bench_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q = d['q32_value_planes']
print('%-8s f64 estep %.4f ms step %.4f | q32 estep %.4f ms step %.4f' % (sys.argv[2], d['roofline']['avg_launch_ms'], d['ms_per_step'], q.get('estep_avg_launch_ms', -1), q.get('ms_per_step', -1)))
PY"
done
python - <<'PY'
import csv, glob, json
res = {}
for bits in (64, 32):
    for p in (1, 2):
        f = glob.glob("gpurun_out/diag/pmc_%d_%d/**/*counter_collection.csv" % (bits, p), recursive=True)
        if not f: continue
        acc = {}
        for r in csv.DictReader(open(f[0])):
            if "k_estep_lane" not in r["Kernel_Name"]: continue
            a = acc.setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, s) in acc.items():
            res.setdefault(str(bits), {})[k] = {"launches": n, "mean": s / n}
json.dump(res, open("gpurun_out/diag/sq_counters.json", "w"), indent=1)
for bits, d in res.items():
    print(bits, {k: round(v["mean"]) for k, v in d.items()})
PY
echo "== total $(( $(date +%s) - start )) s"
