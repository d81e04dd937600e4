#!/bin/bash
# Last call of a round: EM-side GPU tests, then the bench line, then whatever else fits.
budget=${1:-120}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; mkdir -p $out
step() { name=$1; lim=$2; shift 2; l=$(left); [ $l -lt 12 ] && { echo "== $name: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  t0=$(date +%s); timeout $lim "$@"; echo "== $name: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_em 70 bash -c "python -m pytest tests/test_em_q32_gpu.py tests/test_em_gpu.py tests/test_dist_gpu.py -x -q -k 'not full_size_c3' > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -3"
step bench 60 bash -c "python bench.py --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q, c2 = d['q32_value_planes'], d['other_configs']['C2']
print('C3 f64 estep %.4f ms step %.4f frac %.3f | q32 estep %.4f ms step %.4f dtheta %.2e || C2 f64 estep %.4f step %.4f | q32 estep %.4f step %.4f' % (
    d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline']['frac'], q['estep_avg_launch_ms'], q['ms_per_step'], q['theta_max_rel_diff_vs_f64_after_20_rounds'],
    c2['estep_avg_launch_ms'], c2['ms_per_step'], c2['q32_value_planes']['estep_avg_launch_ms'], c2['q32_value_planes']['ms_per_step']))
PY"
step tests_cli 60 bash -c "python -m pytest tests/test_cli_gpu.py -x -q -k 'rsem_run_em_matches_reference' > $out/tests_cli.log 2>&1; grep -E 'passed|failed|rror' $out/tests_cli.log | tail -3"
step tests_rest 200 bash -c "python -m pytest tests -x -q -m gpu --deselect tests/test_em_q32_gpu.py --deselect tests/test_em_gpu.py --deselect tests/test_dist_gpu.py -k 'not rsem_run_em_matches_reference' > $out/tests_rest.log 2>&1; grep -E 'passed|failed|rror' $out/tests_rest.log | tail -3"
echo "== total $(( $(date +%s) - start )) s"
