#!/bin/bash
# Round 4, call P: the two side passes of split rows once more (row pass: loads of 4 x 64 entries issued together, no
# workgroup barrier; column pass: workgroups renumbered so that an XCD walks one contiguous eighth of the entries), against
# the old ones and against the size of the column pass's row-slot blocks; then the tree as it stands (whole -m gpu suite,
# bench without the reference legs); last: the reference's rounds >= 12 at other thread counts / pinned (tools/ref_threads_probe.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04p; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
start=$(date +%s)
leg() {  # row-batched column-xcd block-lg
  RSEM_HIP_ROWSUM_BATCHED=$1 RSEM_HIP_COLSUM_XCD=$2 RSEM_HIP_CSC_BLOCK_LG=$3 timeout 120 python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('C2R row batched $1, column xcd $2, blocks 2^$3: launch %.4f ms frac %.4f frac_physical %.4f parity %s' % (r['avg_launch_ms'], r['frac'], r['frac_physical'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle')))"
}
leg 0 0 16; leg 1 0 16; leg 0 1 16; leg 1 1 16; leg 1 1 17; leg 1 1 18; leg 1 1 20
echo "== variants done $(( $(date +%s) - start )) s"
rm -rf /tmp/prof_p
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/C2R_prof.json 2> $out/C2R_prof.err
python - /tmp/prof_p $out/C2R_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:4]:
    print("   %-46s calls %6s avg %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/prof_p
timeout 400 python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -8
echo "== tests done $(( $(date +%s) - start )) s"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ci > $out/bench_nocpu.json 2> $out/bench_nocpu.err; python -c "
import json; d=json.load(open('$out/bench_nocpu.json'))
r=d['roofline']; print({k: r.get(k) for k in ('frac','frac_physical','avg_launch_ms')}, d['ms_per_step'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle'))
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v.get('parity_one_step', {}).get('max_rel_diff_counts_vs_oracle') for k, v in d.get('other_configs', {}).items()})"
echo "== bench done $(( $(date +%s) - start )) s"
timeout 220 python tools/ref_threads_probe.py 12 > $out/ref_threads_probe.json 2> $out/ref_threads_probe.err; cat $out/ref_threads_probe.err | cut -c1-260
echo "== total $(( $(date +%s) - start )) s"
