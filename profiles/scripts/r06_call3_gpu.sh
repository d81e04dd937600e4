# (run by r06_call3.sh while the reference's configs[1] run occupies 64 host cores of the other socket)
out=gpurun_out/r06c; mkdir -p $out
( timeout 700 python tools/xrows_probe.py C3X,C3X30,C2R most,all_1s,all,all_noids > $out/xrows_probe.log 2>&1; echo "xrows rc=$?" ); cat $out/xrows_probe.log
# kernel start / end times of a few rounds with the split rows' chain on its own stream: do the two streams really overlap?
rm -rf /tmp/xtrace; ( timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/xtrace -o t -- python tools/xrows_probe.py C3X all > $out/xtrace_run.log 2>&1; echo "trace rc=$?" )
python - <<'PY' > gpurun_out/r06c/xtrace_overlap.txt 2>&1
import csv, glob
f = glob.glob("/tmp/xtrace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_estep_lane", "k_far_rowsum", "k_far_colsum", "k_mstep"))]
keep.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = keep[len(keep) // 2: len(keep) // 2 + 16]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
    print("%-28s grid %8s  start %9.1f us  end %9.1f us  (%.1f us)  queue %s" % (n, r.get("Grid_Size", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY
cat $out/xtrace_overlap.txt; rm -rf /tmp/xtrace
( RSEM_HIP_SPLIT_POLICY=all timeout 400 python -m pytest tests/test_em_gpu.py -m gpu -q -x > $out/em_tests_all.log 2>&1; echo "em tests (policy all) rc=$?" ); tail -3 $out/em_tests_all.log
# the model rounds' kernel: chunks of rows per workgroup (default 256) against the grid-wide stride of rounds 4-5, and two other chunk sizes
( MODES="default lib:mstride lib:mchunk64 lib:mchunk1k" timeout 600 tools/profile_model_rounds.sh 10526315 200000 > $out/model_rounds_variants.log 2>&1; echo "model variants rc=$?" ); grep -E "^==|k_model_group" $out/model_rounds_variants.log
( timeout 500 tools/model_group_pmc.sh $out/model_pmc > $out/model_pmc.log 2>&1; echo "pmc rc=$?" ); grep -E "FETCH_SIZE|WRITE_SIZE|RDREQ|WAIT_ANY|WAVE_CYCLES|TCC_HIT|TCC_MISS|READ_REQ_sum|k_model_group" $out/model_pmc.log | head -40
( timeout 900 python -m pytest tests/test_cli_gpu.py -m gpu -q -x -k "matches_reference or generated" > $out/cli_tests.log 2>&1; echo "cli tests rc=$?" ); tail -3 $out/cli_tests.log
D=/tmp/c3_5pct; rm -rf $D
tools/bin/gen_temp $D 2631578 200000 3 20250925 100 nosam 5-16 | tail -1
for i in 1 2 3; do ( time RSEM_HIP_TIMING=2 rsem_amd/bin/rsem-run-em $D/ref 3 $D/s $D/temp/s $D/stat/s -p 64 -q ) > $out/dropin_5pct_$i.log 2>&1; done
grep -E "timing|real" $out/dropin_5pct_3.log
rm -rf $D
