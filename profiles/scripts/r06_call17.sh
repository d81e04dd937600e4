# Round 6, call 17: how many host cores does a call on the GPU box really get?  nproc, the cgroup's quota, and the rate of N copies of a
# compute-only loop (the DEFLATE encoder and zlib on a resident buffer) for N = 1, 8, 16, 32, 64.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06q; mkdir -p $out
echo "nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
grep Cpus_allowed_list /proc/self/status
D=/tmp/e2e_bam; rm -rf $D
tools/bin/gen_temp $D 105263 200000 3 20250925 100 sam 5-16 | tail -1
oracle/_ref/rsem-build-read-index 32 1 1 $D/temp/s_alignable_1.fq $D/temp/s_alignable_2.fq > /dev/null
rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -b $D/aln.sam 0 -q > /dev/null 2>&1
gzip -dc $D/s.transcript.bam 2>/dev/null | head -c 200000000 > /tmp/raw.bin
g++ -O2 -std=c++17 tests/deflate_fast_check.cpp -o /tmp/dfc2 -lz
for n in 1 8 16 32 64; do
  t0=$(date +%s.%N)
  for i in $(seq $n); do /tmp/dfc2 file /tmp/raw.bin > $out/par_$i.txt & done; wait
  t1=$(date +%s.%N)
  echo "N=$n wall $(echo "$t1 - $t0" | bc) s; fast MB/s min/median/max: $(cat $out/par_*.txt | sed 's/.*out (0....), //;s/ MB.*//' | sort -n | sed -n "1p;$(( (n+1)/2 ))p;${n}p" | tr '\n' ' '); zlib MB/s: $(cat $out/par_*.txt | sed 's/.*), //;s/ MB.*//' | sort -n | sed -n "1p;$(( (n+1)/2 ))p;${n}p" | tr '\n' ' ')"
  rm -f $out/par_*.txt
done
rm -rf $D /tmp/raw.bin
