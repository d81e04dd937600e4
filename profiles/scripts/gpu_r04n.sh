#!/bin/bash
# Round 4, call N: the sharded product path (rsem-run-em --devices 0,0: two contexts, LOCAL communicator) at a fifth of configs[2] with
# this round's model kernel and layouts, against the single-context run; the same on an input without gene structure (split rows).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04n; mkdir -p $out
export RSEM_HIP_TIMING=1
D5=/tmp/c3fifth; rm -rf $D5
tools/bin/gen_temp $D5 10526315 200000 3 20250925 100 nosam 5-16 | tail -1
for v in one two; do
  extra=""; [ $v = two ] && extra="--ngpus 2 --devices 0,0"
  t0=$(date +%s.%N)
  rsem_amd/bin/rsem-run-em $D5/ref 3 $D5/s $D5/temp/s $D5/stat/s $extra > $out/$v.log 2>&1; echo "$v rc=$? wall $(awk -v a=$t0 -v b=$(date +%s.%N) 'BEGIN{printf "%.2f", b-a}') s"
  grep -E "^\[timing\] (rounds|main)|^GPU " $out/$v.log | tr '\n' ';'; echo; grep ROUND $out/$v.log | tail -1
  cp $D5/stat/s.theta /tmp/theta_$v; cp $D5/stat/s.model /tmp/model_$v
  grep -v "^ROUND" $out/$v.log > $out/tmp; mv $out/tmp $out/$v.log
done
python - <<'PY'
import numpy as np
def th(v): return np.array(open("/tmp/theta_%s" % v).read().split("\n")[1].split(), float)
a, b = th("one"), th("two"); m = a >= 1e-7
print("theta two shards vs one context: max rel diff %.3g" % np.max(np.abs(a[m] - b[m]) / a[m]))
x = np.array([float(v) for v in open("/tmp/model_one").read().split()]); y = np.array([float(v) for v in open("/tmp/model_two").read().split()])
print(".model: %d numbers, max |diff| / max(|x|, 1e-9) %.3g" % (len(x), np.max(np.abs(x - y) / np.maximum(np.abs(x), 1e-9))))
PY
rm -rf $D5
