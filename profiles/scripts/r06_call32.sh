# Round 6, call 32: BASELINE configs[3] as named through the programs with this round's exact-chain kernel (take-backs behind the commit
# barrier): the 8 count-vector files' sha256 against the ones committed in round 5 (profiles/r05p_reference_countvectors.sha256: the
# reference binary's own files of that run, byte-equal to the drop-in's then).  No reference run here (it takes 20 minutes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06af; mkdir -p $out
export RSEM_HIP_TIMING=1 RSEM_GX_VERBOSE=1
now() { date +%s.%N; }
el() { awk -v a=$1 -v b=$(now) 'BEGIN{printf "%.2f", b-a}'; }
D=/tmp/pin_c4; rm -rf $D
t=$(now); tools/bin/gen_temp $D 52631578 200000 3 20250925 100 nosam 5-16 | tail -1; echo "gen_s $(el $t)"
t=$(now); tools/bin/temp_to_rsb $D/temp/s $D/stat/s 3 > /dev/null; echo "to_rsb_s $(el $t)"
rm -f $D/temp/s.dat $D/temp/*.fq
t=$(now); rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 --gibbs-out > $out/em.log 2>&1; echo "em_rc $? em_s $(el $t)"
grep ROUND $out/em.log | tail -1; ls -la $D/temp/s.ofg | awk '{print ".ofg bytes", $5}'
[ -e $D/temp/s.omit ] || : > $D/temp/s.omit
t=$(now); rsem_amd/bin/rsem-run-gibbs $D/ref $D/temp/s $D/stat/s 200 1000 1 -p 8 --seed 1 > $out/dropin_gibbs.log 2>&1; echo "dropin_gibbs_rc $? dropin_gibbs_s $(el $t)"
grep -E "sampler|sweeps|timing|barriers" $out/dropin_gibbs.log | head -10; cat $D/stat/s.gibbs_sampler 2>/dev/null | tr '\n' ' '; echo
( cd $D/temp && sha256sum s.countvectors* ) > $out/dropin_countvectors.sha256
if diff -q <(sort $out/dropin_countvectors.sha256) <(sort profiles/r05p_reference_countvectors.sha256) > /dev/null; then echo "count-vector files: 8 of 8 sha256 EQUAL to the reference's files of round 5"; else echo "count-vector files DIFFER from round 5's"; diff <(sort $out/dropin_countvectors.sha256) <(sort profiles/r05p_reference_countvectors.sha256) | head; fi
rm -rf $D
