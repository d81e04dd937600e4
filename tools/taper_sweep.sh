#!/bin/bash
# Round-1 experiment: unit tapering (RSEM_HIP_TAPER = fraction of full-size, half-size units) vs E-step time.  Did not pay.
for tp in "1.0,0.0" "0.95,0.05" "0.9,0.1" "0.9,0.05" "0.85,0.1" "0.8,0.2"; do
  export RSEM_HIP_TAPER=$tp
  for i in 1 2; do
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-gibbs --no-ci 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('taper $tp', 'estep_ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'ms/step %.4f' % d['ms_per_step'])"
  done
done
