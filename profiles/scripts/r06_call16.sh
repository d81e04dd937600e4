# Round 6, call 16: the DEFLATE encoder's rate on the GPU box's host: one thread alone, 64 at once (unpinned / pinned to one socket's cores),
# -O2 and -O3, on the record stream of a generated transcript.bam.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06p; mkdir -p $out
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 210526 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -b $D/aln.sam 0 -q > /dev/null 2>&1
gzip -dc $D/s.transcript.bam | head -c 400000000 > /tmp/raw.bin; ls -la /tmp/raw.bin
g++ -O2 -std=c++17 tests/deflate_fast_check.cpp -o /tmp/dfc2 -lz
g++ -O3 -std=c++17 tests/deflate_fast_check.cpp -o /tmp/dfc3 -lz
g++ -O3 -march=native -std=c++17 tests/deflate_fast_check.cpp -o /tmp/dfc3n -lz
lscpu | grep -E "Model name|Thread|Core|Socket|MHz|L2|L3" | head -12
echo "one thread -O2: $(/tmp/dfc2 file /tmp/raw.bin)"
echo "one thread -O3: $(/tmp/dfc3 file /tmp/raw.bin)"
echo "one thread -O3 native: $(/tmp/dfc3n file /tmp/raw.bin)"
echo "64 at once, unpinned:"; for i in $(seq 64); do /tmp/dfc2 file /tmp/raw.bin > $out/par_$i.txt & done; wait; cat $out/par_*.txt | sed 's/.*out (0....), //;s/;.*//' | sort -n | sed -n '1p;32p;64p'
echo "64 at once, one per core of socket 0:"; for i in $(seq 0 63); do taskset -c $i /tmp/dfc2 file /tmp/raw.bin > $out/par_$i.txt & done; wait; cat $out/par_*.txt | sed 's/.*out (0....), //;s/;.*//' | sort -n | sed -n '1p;32p;64p'
echo "128 at once, unpinned:"; for i in $(seq 128); do /tmp/dfc2 file /tmp/raw.bin > $out/par_$i.txt & done; wait; cat $out/par_*.txt | sed 's/.*out (0....), //;s/;.*//' | sort -n | sed -n '1p;64p;128p'
rm -f $out/par_*.txt; rm -rf $D /tmp/raw.bin
