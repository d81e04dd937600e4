# Round 6, call 13: the -b pass's deflate stage by zlib setting (level, memLevel, deflateTune good/lazy/nice/chain): time of the pass and bytes
# of transcript.bam at 2 % of configs[2] (1.05 M reads, SAM input), 64 threads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06m; mkdir -p $out
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 1052631 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
A="$D/ref 3 $D/s $D/temp/s $D/stat/s -p 64"
rsem_amd/bin/rsem-run-em $A -b $D/aln.sam 0 -q > /dev/null 2>&1; mv $D/s.transcript.bam $D/aln.bam
for v in "" "6,9" "5,8" "4,8" "3,8" "1,8" "6,8,8,16,128,32" "6,8,8,16,64,16" "6,9,8,16,64,16" "6,8,8,16,32,8" "6,9,4,8,32,8" "6,9,8,16,258,16"; do
  for i in 1 2; do RSEM_HIP_DEFLATE=$v RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $A -b $D/aln.bam 0 -q > $out/log.txt 2>&1; done
  echo "deflate=[$v] $(grep -o 'transcript.bam  *[0-9.]* s' $out/log.txt) bytes $(stat -c %s $D/s.transcript.bam) $(grep -o 'copy + weigh + deflate [0-9.]*' $out/log.txt) $(grep -o 'deflate [0-9]* MB/s' $out/log.txt)"
done
rm -rf $D
