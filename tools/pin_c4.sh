#!/bin/bash
# BASELINE configs[3] AS NAMED (rsem-run-gibbs, 50 M read pairs, 200 k transcripts, 8 chains, the pipeline's 200 / 1000 / 1), pinned
# against the UNMODIFIED reference binary on the same .ofg, in both modes of the drop-in:
#   exact    : the 8 count-vector files must be byte-equal to the reference's (cmp), the appended result rows equal as printed;
#   parallel : posterior mean counts against the reference's in units of (reference posterior sd + 0.5) -- the statistic of
#              tests/test_gibbs_gpu.py's long-chain study -- beside the same distance for a SECOND reference-equivalent run
#              (the exact chains with another seed: what two runs of the reference differ by).  Pass: rms <= 1.25 x, max <= 1.5 x.
# The reference runs in the background on 8 host cores of the second socket (taskset) for as long as it takes (~40 min); what
# is passed as "$@" after the sizes runs on the GPU meanwhile.
#   GPU box, repo root:   TAG=r05p tools/pin_c4.sh [n_pairs=50000000] [M=200000] [chains=8] [-- command to run while waiting]
N1=${1:-50000000}; M=${2:-200000}; P=${3:-8}; shift 3 2>/dev/null
[ "$1" == "--" ] && shift
NF=$(( N1 * 20 / 19 ))
export RSEM_HIP_TIMING=1
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
R=$PWD; D=/tmp/pin_c4; O=$R/gpurun_out/${TAG:-r05p}; mkdir -p $O
rm -rf $D
t=$(now); tools/bin/gen_temp $D $NF $M 3 20250925 100 nosam ${ISO:-5-16} | tail -1; echo "gen_s $(el $t)"
t=$(now); tools/bin/temp_to_rsb $D/temp/s $D/stat/s 3 > /dev/null; echo "to_rsb_s $(el $t)"
rm -f $D/temp/s.dat $D/temp/*.fq
echo "== rsem-run-em (drop-in) on the binary hand-off, --gibbs-out as TEXT (the reference reads imdName.ofg)"; t=$(now)
rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 --gibbs-out > $O/em.log 2>&1; echo "em_rc $? em_s $(el $t)"
grep -E "^\[timing\]" $O/em.log | tr '\n' ';'; echo; grep ROUND $O/em.log | tail -1; ls -la $D/temp/s.ofg | awk '{print ".ofg bytes", $5}'
cp $D/temp/s.iso_res $D/iso_res.em; cp $D/temp/s.gene_res $D/gene_res.em
# ---- the reference, in a directory of its own on the same .ofg --------------------------------------------------------------------
mkdir -p $D/tref; ln -s $D/temp/s.ofg $D/tref/s.ofg
for f in omit iso_res gene_res; do [ -e $D/temp/s.$f ] && cp $D/temp/s.$f $D/tref/s.$f; done
[ -e $D/tref/s.omit ] || : > $D/tref/s.omit
# 8 physical cores of the LAST package (bench.py pins the reference EM of its own baseline leg to the first package's cores)
CORES=$(python - <<PY
import glob, os
by = {}
for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
    cpu = int(os.path.basename(d)[3:])
    try:
        pkg = int(open(d + "/topology/physical_package_id").read())
        first = int(open(d + "/topology/thread_siblings_list").read().strip().replace("-", ",").split(",")[0])
    except OSError:
        continue
    if first == cpu: by.setdefault(pkg, []).append(cpu)
last = sorted(by[max(by)]) if by else list(range(8))
print(",".join(map(str, last[-$P:])))
PY
)
echo "reference on cpus $CORES"
( t=$(now); taskset -c $CORES oracle/_ref/rsem-run-gibbs $D/ref $D/tref/s $D/stat/s 200 1000 1 -p $P --seed 1 -q > $O/reference_gibbs.log 2>&1
  echo "reference_gibbs_rc $? reference_gibbs_s $(el $t) ($P threads on cpus $CORES)" > $D/ref.done ) &
REFJOB=$!; TREF=$(now)
# ---- the drop-in ----------------------------------------------------------------------------------------------------------------------
run_dropin() {  # name, extra arguments
  name=$1; shift
  cp $D/iso_res.em $D/temp/s.iso_res; cp $D/gene_res.em $D/temp/s.gene_res; rm -f $D/temp/s.countvectors*
  t=$(now); rsem_amd/bin/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p $P "$@" > $O/dropin_$name.log 2>&1; echo "dropin_${name}_rc $? dropin_${name}_s $(el $t)"
  grep -E "sampler|sweeps|timing" $O/dropin_$name.log | head -8; cat $D/stat/s.gibbs_sampler | tr '\n' ' '; echo
  mkdir -p $D/$name; mv $D/temp/s.countvectors* $D/$name/ 2>/dev/null; cp $D/temp/s.iso_res $D/$name/iso_res
}
export RSEM_GX_VERBOSE=1
run_dropin auto --seed 1                       # the default: must pick the reference's chains here
run_dropin exact2 --seed 2 --gibbs-mode exact  # a second reference-equivalent run
run_dropin parallel --seed 1 --gibbs-mode parallel
# ---- whatever else was asked for, while the reference is still at it ------------------------------------------------------------------------
if [ $# -gt 0 ]; then echo "== meanwhile: $*"; t=$(now); "$@"; echo "meanwhile_rc $? meanwhile_s $(el $t)"; fi
echo "== waiting for the reference"; t=$(now)
# (REF_LIMIT seconds at most from its start: a reference that is still running then is stopped, and the lines its chains have
# written so far are compared with the same lines of the drop-in's files -- the GPU box is paid by the minute)
while [ ! -e $D/ref.done ]; do
  sleep 5
  if [ $(awk -v a=$TREF -v b=$(now) 'BEGIN{print int(b-a)}') -gt ${REF_LIMIT:-3300} ]; then
    pkill -P $REFJOB 2>/dev/null; kill $REFJOB 2>/dev/null; sleep 2
    echo "reference_gibbs STOPPED after ${REF_LIMIT:-3300} s (not finished); comparing the count vectors written so far" > $D/ref.done; PARTIAL=1
  fi
done; echo "waited_s $(el $t)"; cat $D/ref.done
if [ -n "$PARTIAL" ]; then
  for k in $(seq 0 $(( P - 1 ))); do
    n=$(wc -l < $D/tref/s.countvectors$k); n=$(( n > 0 ? n - 1 : 0 ))   # (the last line may be incomplete)
    if [ $n -gt 0 ] && cmp -s <(head -n $n $D/tref/s.countvectors$k) <(head -n $n $D/auto/s.countvectors$k); then echo "chain $k: first $n count vectors byte-equal"; else echo "chain $k: $n lines, DIFFERENT or none"; fi
  done
  ls -la $O | awk '{print $5, $9}' | tail -12; rm -rf $D; exit 0
fi
cp $D/tref/s.iso_res $D/tref/iso_res
# ---- compare ----------------------------------------------------------------------------------------------------------------------------------
same=0; diff=0
for k in $(seq 0 $(( P - 1 ))); do
  if cmp -s $D/auto/s.countvectors$k $D/tref/s.countvectors$k; then same=$(( same + 1 )); else diff=$(( diff + 1 )); fi
done
echo "exact chains vs the reference binary, $N1 pairs, -p $P, 200/1000/1, seed 1: $same count-vector files byte-equal, $diff differ ($(stat -c %s $D/tref/s.countvectors0) bytes each)"
( cd $D/tref && sha256sum s.countvectors* ) > $O/reference_countvectors.sha256
( cd $D/auto && sha256sum s.countvectors* ) > $O/dropin_countvectors.sha256
python - <<PY
import gzip, numpy as np
def rows(p):
    r = [l.split("\t") for l in open(p).read().rstrip("\n").split("\n")]
    return r
ref, ex, ex2, par = (rows("$D/%s/iso_res" % d) for d in ("tref", "auto", "exact2", "parallel"))
n0 = len(rows("$D/iso_res.em"))
print("iso_res rows before / after Gibbs:", n0, len(ref), len(ex), len(par))
def tail(r): return [np.array(x, float) for x in r[n0:]]
R, E, E2, Pm = tail(ref), tail(ex), tail(ex2), tail(par)
# appended rows (WriteResults.h:407-476): posterior mean count, its standard deviation, pme TPM, pme FPKM, IsoPct from pme TPM
names = ["pme_c", "sd", "pme_TPM", "pme_FPKM", "IsoPct_pme"][:len(R)]
for i, nm in enumerate(names):
    print("exact vs reference, row %-11s: max |diff| %.4g (printed with 2 decimals)" % (nm, np.abs(E[i] - R[i]).max()))
sd = R[1]
def dist(a, b):
    q = np.abs(a - b) / (sd + 0.5)
    return float(np.sqrt((q ** 2).mean())), float(q.max())
r2, m2 = dist(E2[0], R[0]); rp, mp = dist(Pm[0], R[0])
print("posterior mean counts against the reference's, |diff| / (reference sd + 0.5): second reference-equivalent run (exact, seed 2) rms %.4f max %.3f | "
      "data-augmentation sampler rms %.4f max %.3f  => %s (rule: rms <= 1.25 x, max <= 1.5 x)" % (r2, m2, rp, mp, "PASS" if rp <= 1.25 * r2 and mp <= 1.5 * m2 else "FAIL"))
big = R[0] > 50
print("posterior sd, transcripts with > 50 reads (%d): median ratio to the reference's: exact seed 2 %.3f, data-augmentation %.3f" % (
    big.sum(), np.median(E2[1][big] / R[1][big]), np.median(Pm[1][big] / R[1][big])))
t2, tm2 = dist(E2[2] * 0 + E2[2], R[2]) if False else (0, 0)
rel = lambda a, b: float(np.median(np.abs(a[big] - b[big]) / np.maximum(b[big], 1e-9)))
print("pme TPM, median relative difference to the reference's on those transcripts: exact seed 2 %.5f, data-augmentation %.5f" % (rel(E2[2], R[2]), rel(Pm[2], R[2])))
with gzip.open("$O/reference_full_size_gibbs_rows.txt.gz", "wt") as f:
    for r in ref[n0:]: f.write("\t".join(r) + "\n")
with gzip.open("$O/dropin_parallel_gibbs_rows.txt.gz", "wt") as f:
    for r in par[n0:]: f.write("\t".join(r) + "\n")
PY
ls -la $O | awk '{print $5, $9}' | tail -12
rm -rf $D
