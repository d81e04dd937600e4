#!/bin/bash
# Three default-workload bench runs in a row (E-step launch time and fraction): run-to-run spread on one box.
for i in 1 2 3; do
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-gibbs --no-ci 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('estep_ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'ms/step %.4f' % d['ms_per_step'])"
done
