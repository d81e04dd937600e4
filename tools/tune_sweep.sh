#!/bin/bash
# Round-1 experiment: passes of the measured-lifetime reordering of the units (RSEM_HIP_TUNE) vs E-step time.
for t in 0 1 2 3; do
  export RSEM_HIP_TUNE=$t
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gibbs --no-ci 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('tune $t', 'estep_ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'ms/step %.4f' % d['ms_per_step'], 'theta_sum', d['checks']['theta_sum'])"
done
RSEM_HIP_TUNE=1 python tools/trace_estep.py 1.0
