#!/bin/bash
# A BASELINE config through the PROGRAMS, pinned against the UNMODIFIED reference binary on the same files:
#   the drop-in rsem-run-em (plus the flags given in DROPIN_FLAGS) and oracle/_ref/rsem-run-em -p 64 on a generated .temp directory;
#   same ROUND count, theta to 1e-6 (observed 1e-12), .iso_res / .gene_res equal as printed.  The reference runs in the background
#   on the 64 physical cores of the LAST package; what follows "--" runs meanwhile (GPU work of the same call).
#   GPU box, repo root:  TAG=r06c tools/pin_config.sh <name> <read_type> <n_reads_total> <M> <iso> [-- command ...]
#   e.g. configs[1]:  tools/pin_config.sh configs1 1 10526315 50000 4-12      configs[4] at a tenth:  ... configs4_tenth 1 10526315 500000 32-64
name=$1; RT=$2; N=$3; M=$4; ISO=$5; shift 5; [ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$PWD; D=/tmp/pin_$name; O=$R/gpurun_out/${TAG:-r06}; mkdir -p $O
export RSEM_HIP_TIMING=1
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
rm -rf $D
t=$(now); tools/bin/gen_temp $D $N $M $RT 20250925 100 nosam $ISO | tail -1; echo "gen_s $(el $t)"; du -sh $D | cut -f1
if [ "$RT" == "1" ] || [ "$RT" == "0" ]; then oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable.f? > /dev/null; else oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.f? $D/temp/s_alignable_2.f? > /dev/null; fi
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
last = sorted(by[max(by)]) if by else list(range(64))
print(",".join(map(str, last[:64])))
PY
)
echo "== drop-in rsem-run-em $DROPIN_FLAGS"; t=$(now)
rsem_amd/bin/rsem-run-em $D/ref $RT $D/s $D/temp/s $D/stat/s -p 64 $DROPIN_FLAGS > $O/${name}_dropin.log 2>&1; echo "dropin_rc $? dropin_s $(el $t)"
grep -E "^\[timing\]" $O/${name}_dropin.log | tr '\n' ';'; echo; grep "^ROUND" $O/${name}_dropin.log | tail -1
grep "^ROUND" $O/${name}_dropin.log | gzip > $O/${name}_dropin_ROUND_lines.txt.gz
grep -v "^ROUND" $O/${name}_dropin.log > $O/${name}_dropin.tmp; mv $O/${name}_dropin.tmp $O/${name}_dropin.log
mkdir -p $D/new; cp $D/stat/s.theta $D/new/theta; cp $D/temp/s.iso_res $D/new/iso_res; cp $D/temp/s.gene_res $D/new/gene_res; cp $D/stat/s.model $D/new/model
echo "== reference rsem-run-em -p 64 on cpus $CORES (background)"
( t=$(now); taskset -c $CORES oracle/_ref/rsem-run-em $D/ref $RT $D/s $D/temp/s $D/stat/s -p 64 > $D/ref.log 2>&1; echo "reference_rc $? reference_s $(el $t)" > $D/ref.done ) &
REFJOB=$!
if [ $# -gt 0 ]; then echo "== meanwhile: $*"; "$@"; fi
wait $REFJOB; cat $D/ref.done; grep "^ROUND" $D/ref.log | tail -1; grep "Time Used" $D/ref.log
grep "^ROUND" $D/ref.log | gzip > $O/${name}_reference_ROUND_lines.txt.gz
python - $D $O $name <<'PY'
import gzip, sys, numpy as np
D, O, name = sys.argv[1:4]
def theta(p): return [np.array(l.split(), float) for l in open(p).read().split("\n")[1:3]]
a, b = theta(D + "/new/theta"), theta(D + "/stat/s.theta")
m = b[0] >= 1e-7
print("theta_max_rel_diff %.3g over %d transcripts with theta >= 1e-7 (of %d)" % (np.max(np.abs(a[0][m] - b[0][m]) / b[0][m]), m.sum(), len(m)))
ra = [l for l in gzip.open("%s/%s_dropin_ROUND_lines.txt.gz" % (O, name), "rt")]
rb = [l for l in gzip.open("%s/%s_reference_ROUND_lines.txt.gz" % (O, name), "rt")]
def parts(l):  # ROUND = r, SUM = s, bChange = b, totNum = t
    f = [x.split("=")[1].strip() for x in l.strip().split(",")]
    return int(f[0]), f[2], int(f[3])
same = len(ra) == len(rb) and all(parts(x)[0] == parts(y)[0] and parts(x)[2] == parts(y)[2] for x, y in zip(ra, rb))
sameb = len(ra) == len(rb) and sum(parts(x)[1] == parts(y)[1] for x, y in zip(ra, rb))
print("ROUND lines: drop-in %d, reference %d; every round's totNum equal: %s; rounds with the same printed bChange: %s" % (len(ra), len(rb), same, sameb))
for f in ("iso_res", "gene_res"):
    x, y = open(D + "/new/" + f).read(), open(D + "/temp/s." + f).read()
    if x == y: print(f, "byte-identical")
    else:
        rx, ry = x.strip().split("\n"), y.strip().split("\n")
        worst = 0.0
        for lx, ly in zip(rx, ry):
            try:
                vx, vy = np.array(lx.split("\t"), float), np.array(ly.split("\t"), float)
                worst = max(worst, float(np.max(np.abs(vx - vy))))
            except ValueError:
                assert lx == ly
        print(f, "rows %d / %d, max |difference| of a printed value %.3g" % (len(rx), len(ry), worst))
with gzip.open("%s/%s_reference.theta.gz" % (O, name), "wt") as f: f.write(open(D + "/stat/s.theta").read())
PY
rm -rf $D
