# Round 6, call 39: the -b pass by super-chunk size (256 MB default; 512 MB, 1 GB, 2 GB): fewer stage barriers under the boxes' CPU quota?  10 % of configs[2], BAM input.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06am; mkdir -p $out
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 5263157 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
A="$D/ref 3 $D/s $D/temp/s $D/stat/s -p 64"
rsem_amd/bin/rsem-run-em $A -b $D/aln.sam 0 -q > /dev/null 2>&1; mv $D/s.transcript.bam $D/aln.bam; rm -f $D/aln.sam
for c in 268435456 536870912 1073741824 2147483648 268435456 1073741824; do
  ( time RSEM_HIP_BAM_CHUNK=$c RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $A -b $D/aln.bam 0 -q ) > $out/log_$c.txt 2>&1
  echo "chunk $c: $(grep -o 'transcript.bam  *[0-9.]* s' $out/log_$c.txt) $(grep real $out/log_$c.txt) | $(grep 'transcript.bam pass' $out/log_$c.txt | sed 's/.*stages (wall) //' | cut -c1-260)"
done
rm -rf $D
