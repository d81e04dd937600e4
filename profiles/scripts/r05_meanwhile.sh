#!/bin/bash
# What the GPU does while the reference's rsem-run-gibbs works through BASELINE configs[3] on 8 host cores (tools/pin_c4.sh ... -- this):
# the -m gpu suite, the FETCH_SIZE calibration, the model rounds' kernel in three builds, the bench line, its rocprof / PMC evidence.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-r05q}; mkdir -p $O
now() { date +%s; }
t=$(now); timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$? $(( $(now) - t )) s"; tail -3 $O/gpu_tests.log
t=$(now); rm -rf $O/fetch_calib; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_calib -o f -- tools/microbench/fetch_calib > $O/fetch_calib.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/fetch_calib/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
out = open("$O/fetch_calib_table.txt", "w")
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        line = "%-40s FETCH_SIZE %12.1f KB  (requested 1048576 KB: counter / requested = %.3f)" % (r["Kernel_Name"].split("(")[0][-40:], float(r["Counter_Value"]), float(r["Counter_Value"]) / 1048576.0)
        print(line); out.write(line + "\n")
PY
echo "fetch calibration $(( $(now) - t )) s"
t=$(now); MODES="default lib:mahead lib:mahead3w" timeout 900 tools/profile_model_rounds.sh 10526315 200000 > $O/model_rounds.log 2>&1; echo "model rounds rc=$? $(( $(now) - t )) s"; grep -E "^==|k_model_group" $O/model_rounds.log
t=$(now); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(now) - t )) s"; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
t=$(now); timeout 1200 tools/profile_round.sh ${TAG:-r05q} "C3" > $O/profile.log 2>&1; echo "profile rc=$? $(( $(now) - t )) s"; tail -25 $O/profile.log
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null; find gpurun_out -name "*.db" -size +4M -delete 2>/dev/null; du -sh gpurun_out | cut -f1
