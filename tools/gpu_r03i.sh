#!/bin/bash
# Round 3, ninth GPU call: the foreign side path of the E step (tests; C3X / C2R / C2 / C3 times).
budget=${1:-600}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03i; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_foreign 200 bash -c "python -m pytest tests/test_em_gpu.py -x -q -k 'foreign or unstructured' > $out/tests_foreign.log 2>&1; tail -15 $out/tests_foreign.log | cut -c1-300"
step tests_em 300 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_dist_gpu.py -x -q -k 'not full_size' > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -3"
step bench 200 bash -c "python bench.py --steps 20 --warmup 5 --legs C2,C2R,C3X --no-gibbs --no-ci --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err; python -c \"
import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print('C3 estep %.4f step %.4f step/launch %.4f frac %.3f q32 %.4f' % (r['avg_launch_ms'], d['ms_per_step'], r['step_over_launch'], r['frac'], d['q32_value_planes']['estep_avg_launch_ms']))
print({k: {kk: v.get(kk) for kk in ('estep_avg_launch_ms','ms_per_step','frac','error')} for k, v in d.get('other_configs', {}).items()})
print({k: v['parity_one_step'] for k, v in d.get('other_configs', {}).items()}, d['checks']['parity_one_step'].get('ok'))\""
step tests_cli_em 200 bash -c "python -m pytest tests/test_cli_gpu.py -x -q -k 'run_em or generated' > $out/tests_cli.log 2>&1; grep -E 'passed|failed|rror' $out/tests_cli.log | tail -3"
echo "== total $(( $(date +%s) - start )) s"
