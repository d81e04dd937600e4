#!/bin/bash
# Round 3, fifth GPU call: the block-synchronous exact sampler (tests, per-round time, phase profile), the new CLI / pipeline
# tests (binary hand-offs), the EM tests after the close-only last launch, the stream probe.
budget=${1:-780}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03e; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step tests_gibbs 200 bash -c "python -m pytest tests/test_gibbs_gpu.py -x -q -s > $out/tests_gibbs.log 2>&1; grep -E 'passed|failed|rror|exact sweeps' $out/tests_gibbs.log | tail -8"
step exact_c2 120 bash -c "python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg,coop 2>&1 | tee $out/exact_c2.log"
step exact_c3x02 120 bash -c "python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_c3x0.2.log"
step exact_prof_c2 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 1.0 8 3 C2 wg 2>&1 | tee $out/exact_prof_c2.log"
step exact_prof_c3 120 bash -c "RSEM_HIP_LIB=$PWD/rsem_amd/librsem_hip_xprof.so python tools/gibbs_exact_profile.py 0.2 8 2 C3 wg 2>&1 | tee $out/exact_prof_c3x0.2.log"
step exact_c5 120 bash -c "python tools/gibbs_exact_profile.py 0.02 8 2 C5 wg,coop 2>&1 | tee $out/exact_c5x0.02.log"
step exact_c3 150 bash -c "python tools/gibbs_exact_profile.py 1.0 8 2 C3 wg 2>&1 | tee $out/exact_c3.log"
step stream 40 bash -c "python -c \"
from rsem_amd import capi
print('stream probe read/copy GB/s: %.0f %.0f' % capi.stream_probe(0, 8 << 30, 5))\" | tee $out/stream.log"
step tests_cli 300 bash -c "python -m pytest tests/test_cli_gpu.py tests/test_pipeline_gpu.py -x -q > $out/tests_cli.log 2>&1; grep -E 'passed|failed|rror' $out/tests_cli.log | tail -5"
step tests_em 200 bash -c "python -m pytest tests/test_em_gpu.py tests/test_em_q32_gpu.py tests/test_dist_gpu.py -x -q -k 'not full_size' > $out/tests_em.log 2>&1; grep -E 'passed|failed|rror' $out/tests_em.log | tail -3"
step bench 120 bash -c "python bench.py --steps 20 --warmup 5 --legs C2 --no-gibbs --no-ci --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python -c \"
import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print('C3 estep %.4f step %.4f step/launch %.4f frac %.3f stream %s' % (r['avg_launch_ms'], d['ms_per_step'], r['step_over_launch'], r['frac'], r['stream']))\""
echo "== total $(( $(date +%s) - start )) s"
