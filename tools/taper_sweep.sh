for tp in "1.0,0.0" "0.7,0.2" "0.55,0.25" "0.4,0.3" "0.25,0.35" "0.0,0.5" "0.0,0.0"; do
  export RSEM_HIP_TAPER=$tp
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gibbs --no-ci 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('taper $tp', 'estep_ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'ms/step %.4f' % d['ms_per_step'], 'theta_sum', d['checks']['theta_sum'])"
done
