# Round 6, call 15: the -b pass with this repository's DEFLATE encoder (deflate_fast.hpp) and the parallel framing of BAM input: 2 % of
# configs[2] (own encoder / zlib tuned / zlib's own level 6, pass time and bytes), then 10 % with BAM input against the reference -p 64.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06o; mkdir -p $out
( timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -x -k "bam" > $out/cli_bam_tests.log 2>&1; echo "cli bam tests rc=$?" ); tail -3 $out/cli_bam_tests.log
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 1052631 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
A="$D/ref 3 $D/s $D/temp/s $D/stat/s -p 64"
rsem_amd/bin/rsem-run-em $A -b $D/aln.sam 0 -q > /dev/null 2>&1; mv $D/s.transcript.bam $D/aln.bam
for v in "-" "zlib" "6,8"; do
  for i in 1 2; do if [ "$v" = "-" ]; then RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $A -b $D/aln.bam 0 -q > $out/log_2pct.txt 2>&1; else RSEM_HIP_DEFLATE=$v RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $A -b $D/aln.bam 0 -q > $out/log_2pct.txt 2>&1; fi; done
  echo "deflate=[$v] $(grep -o 'transcript.bam  *[0-9.]* s' $out/log_2pct.txt) bytes $(stat -c %s $D/s.transcript.bam) | $(grep 'transcript.bam pass' $out/log_2pct.txt | sed 's/.*stages (wall) //') | $(grep 'framing:' $out/log_2pct.txt)"
  gzip -dc $D/s.transcript.bam | md5sum
done
rm -rf $D
( TAG=r06o timeout 1500 tools/e2e_bam.sh > $out/e2e_bam.log 2>&1; echo "e2e_bam rc=$?" ); cat $out/e2e_bam.log
