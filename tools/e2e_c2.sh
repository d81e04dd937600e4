#!/bin/bash
# End-to-end timing of the drop-in rsem-run-em (and optionally the reference) on a generated C2-scale input.
# usage: tools/e2e_c2.sh <n_reads> <M> <read_type> [ref]
set -e
N=${1:-10000000}; M=${2:-50000}; RT=${3:-1}; D=/tmp/e2e_$N_$RT
tools/bin/gen_temp $D $N $M $RT | tail -1
export RSEM_HIP_TIMING=1
( time rsem_amd/bin/rsem-run-em $D/ref $RT $D/s $D/temp/s $D/stat/s --gibbs-out ) > gpurun_out/e2e_new_$RT.log 2>&1 || true
grep -E "timing|real" gpurun_out/e2e_new_$RT.log | tail -12
grep ROUND gpurun_out/e2e_new_$RT.log | tail -1
if [ "$4" == "ref" ]; then
  cp $D/stat/s.theta gpurun_out/e2e_new_$RT.theta
  if [ "$RT" == "1" ]; then oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable.fq > /dev/null; else oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null; fi
  ( time timeout 3000 oracle/_ref/rsem-run-em $D/ref $RT $D/s $D/temp/s $D/stat/s -p ${REF_P:-64} --gibbs-out ) > gpurun_out/e2e_ref_$RT.log 2>&1 || true
  grep -E "ROUND|real" gpurun_out/e2e_ref_$RT.log | tail -3
  python - <<PY
import numpy as np
a=[np.array(l.split(),float) for l in open("gpurun_out/e2e_new_$RT.theta").read().split("\n")[1:3]]
b=[np.array(l.split(),float) for l in open("$D/stat/s.theta").read().split("\n")[1:3]]
m=b[0]>=1e-7
print("theta max rel diff", np.max(np.abs(a[0][m]-b[0][m])/b[0][m]))
PY
fi
rm -rf $D
