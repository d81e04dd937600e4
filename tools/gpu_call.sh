#!/bin/bash
# One gpurun call's worth of work, logged under gpurun_out/ (the tail of this script's stdout comes back directly).
#   tools/gpu_call.sh <tag> <step> [<step> ...]     steps: tests[:<pytest -k expr>] truth[:N:M:P] bench[:args] cmd:<shell>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  t0=$(date +%s)
  case $kind in
    tests) if [ -n "$arg" ]; then timeout 1200 python -m pytest tests -m gpu -q -k "$arg" -s > $out/pytest.log 2>&1; else timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; fi
           tail -25 $out/pytest.log ;;
    truth) IFS=: read -r n m p <<< "$arg"; timeout 900 tools/gibbs_truth.sh ${n:-1000000} ${m:-20000} ${p:-64} > $out/gibbs_truth.log 2>&1; cat $out/gibbs_truth.log ;;
    bench) timeout 1500 python bench.py $arg > $out/bench.json 2> $out/bench.err; tail -5 $out/bench.err; cut -c1-3000 $out/bench.json ;;
    cmd)   timeout 1500 bash -c "$arg" > $out/cmd.log 2>&1; tail -40 $out/cmd.log ;;
  esac
  echo "== step $step: $(( $(date +%s) - t0 )) s"
done
