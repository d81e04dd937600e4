# Round 6, call 29: the -b pass with SAM input at 10 % of configs[2] after encode_sam_line lost its heap traffic (the CLI's BAM tests first).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ac; mkdir -p $out
( timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -x -k "bam" > $out/cli_bam_tests.log 2>&1; echo "cli bam tests rc=$?" ); tail -2 $out/cli_bam_tests.log
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 5263157 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
A="$D/ref 3 $D/s $D/temp/s $D/stat/s -p 64"
for i in 1 2; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $A -b $D/aln.sam 0 -q ) > $out/sam_input_$i.log 2>&1; grep real $out/sam_input_$i.log; done
grep "transcript.bam" $out/sam_input_2.log | cut -c1-700
rm -rf $D
