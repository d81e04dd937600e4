#!/bin/bash
# End-to-end: SAM -> rsem-parse-alignments -> rsem-run-em, drop-in vs reference binaries, on a generated data set.
# usage: tools/e2e_parse.sh <n_reads> <M> <read_type> [ref]
N=${1:-5000000}; M=${2:-50000}; RT=${3:-1}; D=/tmp/e2ep_${N}_$RT
rm -rf $D; tools/bin/gen_temp $D $N $M $RT 7 100 sam | tail -1
ls -la $D/aln.sam | awk '{print "SAM bytes", $5}'
mkdir -p $D/new/temp $D/new/stat $D/ref/temp $D/ref/stat
cp $D/temp/s.mparams $D/new/temp/; cp $D/temp/s.mparams $D/ref/temp/
echo "== drop-in parse"; time rsem_amd/bin/rsem-parse-alignments $D/ref $D/new/temp/s $D/new/stat/s $D/aln.sam $RT -q
export RSEM_HIP_TIMING=1
echo "== drop-in EM"; ( time rsem_amd/bin/rsem-run-em $D/ref $RT $D/new/s $D/new/temp/s $D/new/stat/s -q ) 2>&1 | grep -E "real|ROUND" | tail -2
if [ "$4" == "ref" ]; then
  echo "== reference parse"; time oracle/_ref/rsem-parse-alignments $D/ref $D/ref/temp/s $D/ref/stat/s $D/aln.sam $RT -q
  for f in $(cd $D/ref && find . -type f ! -name s.mparams); do cmp $D/ref/$f $D/new/$f || echo "DIFF $f"; done; echo "compare done"
fi
rm -rf $D
