#!/bin/bash
# Round 4, call G: the far row pass two ways (thread per row slot / wave per 256 entries) at C2R, E-step tests on the final code.
budget=${1:-400}
start=$(date +%s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04g; mkdir -p $out
export RSEM_WL_CACHE=/dev/shm/rsem_wl
run() { local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 150 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream > $out/$tag.json 2> $out/$tag.err
  echo "== $tag rc=$?"
  python - /tmp/prof_$tag $out/${tag}_kernel_stats.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows:
    with open(sys.argv[2], "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:5]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-46s calls %6s avg %10.1f us total %9.1f ms" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf /tmp/prof_$tag
}
run C2R_thread A=1
run C2R_wave RSEM_HIP_ROWSUM=wave
for v in thread wave; do
  e=""; [ $v = wave ] && e="RSEM_HIP_ROWSUM=wave"
  env $e python bench.py --config C2R --legs= --steps 20 --warmup 3 --no-cpu-baseline --no-gibbs --no-ci --no-q32 --no-stream 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('C2R $v (no profiler): launch %.4f ms frac %.4f frac_physical %.4f parity %s' % (r['avg_launch_ms'], r['frac'], r['frac_physical'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle')))"
done
timeout 200 python -m pytest tests/test_em_gpu.py -q -m gpu -k "unstructured or another_gene or edge or golden" 2>&1 | tail -3
RSEM_HIP_ROWSUM=wave timeout 200 python -m pytest tests/test_em_gpu.py -q -m gpu -k "unstructured" 2>&1 | tail -2
echo "== total $(( $(date +%s) - start )) s"
