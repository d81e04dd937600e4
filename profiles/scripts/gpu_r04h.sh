#!/bin/bash
# Round 4, call H: the final tree -- smoke(), the whole -m gpu suite, the headline and legs once more (no reference runs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04h; mkdir -p $out
start=$(date +%s)
timeout 120 python -c "import __graft_entry__ as g; g.smoke()"; echo "== smoke rc=$?"
timeout 700 python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1; grep -E 'passed|failed|rror' $out/tests_all.log | tail -8
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ci > $out/bench_nocpu.json 2> $out/bench_nocpu.err; python -c "
import json; d=json.load(open('$out/bench_nocpu.json'))
r=d['roofline']; print({k: r.get(k) for k in ('frac','frac_physical','avg_launch_ms')}, d['ms_per_step'], d['checks']['parity_one_step'].get('max_rel_diff_counts_vs_oracle'))
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','frac','frac_physical','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v.get('units_with_ids_outside_their_window') for k, v in d.get('other_configs', {}).items()}, d['config'].get('units_with_ids_outside_their_window'))"
echo "== total $(( $(date +%s) - start )) s"
