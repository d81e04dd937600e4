# Round 6, call 41: does the program run faster with as many threads as the box grants cores (16) than with the 64 / 256 it sees?  -b at 10 % by -p,
# and configs[2] at full size under taskset (hardware_concurrency follows the affinity mask).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ao; mkdir -p $out
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 5263157 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
B="$D/ref 3 $D/s $D/temp/s $D/stat/s"
rsem_amd/bin/rsem-run-em $B -p 64 -b $D/aln.sam 0 -q > /dev/null 2>&1; mv $D/s.transcript.bam $D/aln.bam; rm -f $D/aln.sam
for p in 64 16 24 32 64 16; do
  ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $B -p $p -b $D/aln.bam 0 -q ) > $out/bam_p$p.txt 2>&1
  echo "-p $p: $(grep -o 'transcript.bam  *[0-9.]* s' $out/bam_p$p.txt) $(grep real $out/bam_p$p.txt) | $(grep 'transcript.bam pass' $out/bam_p$p.txt | sed 's/.*stages (wall) //' | cut -c1-200)"
done
rm -rf $D
D=/tmp/c3_full; rm -rf $D
tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1
for v in all c16 c32 all c16; do
  case $v in all) T="";; c16) T="taskset -c 0-15";; c32) T="taskset -c 0-31";; esac
  ( time RSEM_HIP_TIMING=2 $T rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/full_$v.log 2>&1
  echo "$v: $(grep -E 'refs \+|parse read|estimateFrom|contexts \+|device loop|main\(\) total' $out/full_$v.log | sed 's/\[timing\] //' | tr -s ' ' | tr '\n' ';') $(grep real $out/full_$v.log)"
done
rm -rf $D
