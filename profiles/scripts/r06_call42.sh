# Round 6, call 42: the -b pass with no more threads than the cgroup grants cores (the program is given -p 64 as before): CLI BAM tests, 10 % of configs[2], BAM and SAM input.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ap; mkdir -p $out
( timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -x -k "bam" > $out/cli_bam_tests.log 2>&1; echo "cli bam tests rc=$?" ); tail -2 $out/cli_bam_tests.log
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 5263157 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
B="$D/ref 3 $D/s $D/temp/s $D/stat/s -p 64"
for i in 1 2; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $B -b $D/aln.sam 0 -q ) > $out/sam_$i.txt 2>&1; echo "SAM input: $(grep -o 'transcript.bam  *[0-9.]* s' $out/sam_$i.txt) $(grep real $out/sam_$i.txt)"; done
mv $D/s.transcript.bam $D/aln.bam; rm -f $D/aln.sam
for i in 1 2 3; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $B -b $D/aln.bam 0 -q ) > $out/bam_$i.txt 2>&1; echo "BAM input: $(grep -o 'transcript.bam  *[0-9.]* s' $out/bam_$i.txt) $(grep real $out/bam_$i.txt) | $(grep 'transcript.bam pass' $out/bam_$i.txt | sed 's/.*transcript.bam pass, //' | cut -c1-230)"; done
rm -rf $D
