#!/bin/bash
# Round 3, tenth GPU call: the foreign side path again (probe with the list sizes, tests).
budget=${1:-420}
start=$(date +%s)
left() { echo $(( budget - ($(date +%s) - start) )); }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RSEM_WL_CACHE=/dev/shm/rsem_wl
out=gpurun_out/r03j; mkdir -p $out
step() { local sname=$1 lim=$2; shift 2; local l=$(left); [ $l -lt 15 ] && { echo "== $sname: skipped, $l s left"; return; }; [ $lim -gt $l ] && lim=$l
  local t0=$(date +%s); timeout $lim "$@"; echo "== $sname: rc=$? $(( $(date +%s) - t0 )) s"; }
step probe_small 60 bash -c "python tools/foreign_probe.py smallX 1.0 2>&1 | tail -4 | tee $out/probe_smallX.log"
step probe_c2r 60 bash -c "python tools/foreign_probe.py C2R 1.0 2>&1 | tail -4 | tee $out/probe_c2r.log"
step probe_c3x 90 bash -c "python tools/foreign_probe.py C3X 1.0 2>&1 | tail -4 | tee $out/probe_c3x.log"
step probe_c3 90 bash -c "python tools/foreign_probe.py C3 1.0 2>&1 | tail -4 | tee $out/probe_c3.log"
step tests_foreign 200 bash -c "python -m pytest tests/test_em_gpu.py -q -k 'foreign or unstructured' > $out/tests_foreign.log 2>&1; tail -12 $out/tests_foreign.log | cut -c1-300"
echo "== total $(( $(date +%s) - start )) s"
