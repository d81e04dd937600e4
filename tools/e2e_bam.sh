#!/bin/bash
# rsem-run-em WITH the transcript.bam pass (-b: on by default in rsem-calculate-expression, :61,626-632; BamWriter.h:39-146) at a size
# that means something, BAM input (what aligners hand over): the UNMODIFIED reference binary (-p 64, pinned to one socket) and the
# drop-in (-p 64) on the same files, wall clock of the whole programs, theta compared, the drop-in's pass broken down by stage.
#   GPU box, repo root:  TAG=r06d tools/e2e_bam.sh [n_reads_total=5263157 (10 % of configs[2])] [M=200000]
N=${1:-5263157}; M=${2:-200000}; P=${REF_P:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$PWD; D=/tmp/e2e_bam; O=$R/gpurun_out/${TAG:-r06d}; mkdir -p $O
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
rm -rf $D
t=$(now); tools/bin/gen_temp $D $N $M 3 20250925 100 sam 5-16 | tail -1; echo "gen_s $(el $t)"; ls -la $D/aln.sam | awk '{print "aln.sam bytes", $5}'
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
A="$D/ref 3 $D/s $D/temp/s $D/stat/s -p $P"
# BAM input = the records of aln.sam as BAM: the drop-in's own transcript.bam of a first, untimed run (both programs overwrite MAPQ / ZW)
t=$(now); rsem_amd/bin/rsem-run-em $A -b $D/aln.sam 0 -q > $O/convert.log 2>&1; echo "convert_rc $? convert_s $(el $t) (drop-in, SAM input: EM + pass)"
mv $D/s.transcript.bam $D/aln.bam; rm -f $D/aln.sam; ls -la $D/aln.bam | awk '{print "aln.bam bytes", $5}'
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
first = sorted(by[min(by)]) if by else list(range(64))
print(",".join(map(str, first[:64])))
PY
)
echo "== drop-in, BAM input"; for i in 1 2; do t=$(now); RSEM_HIP_TIMING=1 rsem_amd/bin/rsem-run-em $A -b $D/aln.bam 0 > $O/dropin_bam_$i.log 2>&1; echo "dropin_rc $? dropin_s $(el $t)"; done
grep -E "^\[timing\]" $O/dropin_bam_2.log | grep -v "model round"; grep "^ROUND" $O/dropin_bam_2.log | tail -1
grep -v "^ROUND" $O/dropin_bam_2.log > $O/dropin_bam.log; rm -f $O/dropin_bam_1.log $O/dropin_bam_2.log
cp $D/stat/s.theta $D/new.theta; ls -la $D/s.transcript.bam | awk '{print "drop-in transcript.bam bytes", $5}'; mv $D/s.transcript.bam $D/new.transcript.bam
echo "== reference -p $P on cpus $CORES, BAM input"; t=$(now)
taskset -c $CORES oracle/_ref/rsem-run-em $A -b $D/aln.bam 0 > $O/reference_bam.log 2>&1; echo "reference_rc $? reference_s $(el $t)"
grep "^ROUND" $O/reference_bam.log | tail -1; grep "Time Used" $O/reference_bam.log
# when did the reference's EM end and its BAM pass begin: the last ROUND line's arrival is not logged; "Time Used for EM.cpp" covers both.
grep -v "^ROUND" $O/reference_bam.log > $O/reference_bam.tmp; mv $O/reference_bam.tmp $O/reference_bam.log
ls -la $D/s.transcript.bam | awk '{print "reference transcript.bam bytes", $5}'
python - $D <<'PY'
import sys, numpy as np
D = sys.argv[1]
def theta(p): return np.array(open(p).read().split("\n")[1].split(), float)
a, b = theta(D + "/new.theta"), theta(D + "/stat/s.theta")
m = b >= 1e-7
print("theta_max_rel_diff %.3g" % np.max(np.abs(a[m] - b[m]) / b[m]))
PY
# the same records?  tests/test_cli_gpu.py and tests/test_bam_cpu.py compare them one by one on the fixtures (MAPQ / the last bit of a ZW
# float may differ where theta differs in its 13th digit); here: the decompressed streams must be equally long
( gzip -dc $D/new.transcript.bam | wc -c | sed 's/^/dropin decompressed bytes /' ) &
( gzip -dc $D/s.transcript.bam | wc -c | sed 's/^/reference decompressed bytes /' ) &
wait
rm -rf $D
