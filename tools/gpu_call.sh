#!/bin/bash
# One gpurun call's worth of work, logged under gpurun_out/ (the tail of this script's stdout comes back directly).
#   tools/gpu_call.sh <tag> <step> [<step> ...]     steps: tests[:<pytest -k expr>] truth[:N:M:P] bench[:args] prof[:tag[:cfgs]] e2e[:N:M:SUB] cmd:<shell>
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
    prof)  IFS=: read -r ptag pcfgs <<< "$arg"; timeout 1500 tools/profile_round.sh ${ptag:-rXX} "${pcfgs:-C3 C2}" > $out/profile.log 2>&1; tail -60 $out/profile.log ;;
    e2e)   IFS=: read -r n m sub <<< "$arg"; timeout 2400 tools/e2e_c3.sh ${n:-50000000} ${m:-200000} ${sub:-10} > $out/e2e_c3.log 2>&1; cat $out/e2e_c3.log ;;
    c4)    IFS=: read -r n m pp <<< "$arg"; timeout 2400 tools/e2e_c4.sh ${n:-50000000} ${m:-200000} ${pp:-8} > $out/e2e_c4.log 2>&1; cat $out/e2e_c4.log ;;
    cmd)   timeout 1500 bash -c "$arg" > $out/cmd.log 2>&1; tail -40 $out/cmd.log ;;
  esac
  echo "== step $step: $(( $(date +%s) - t0 )) s"
done
