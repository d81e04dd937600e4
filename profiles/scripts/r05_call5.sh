cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05r; mkdir -p $O
t=$(date +%s); timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$? $(( $(date +%s) - t )) s"; tail -4 $O/gpu_tests.log | head -3; grep -E "FAILED|ERROR" $O/gpu_tests.log | head
t=$(date +%s); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t )) s"; tail -2 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
r=d["roofline"]; print("C3 ms/step", d["ms_per_step"], "launch", r["avg_launch_ms"], "frac", r["frac"], "frac_alg", r["frac_algorithmic"])
e=d.get("e2e_wall_clock",{}); print("e2e measured", {k:e.get("measured",{}).get(k) for k in ("reference_s","dropin_s","speedup")}); print("bam_on", e.get("bam_on")); print("full", {k:e.get("full_size",{}).get(k) for k in ("dropin_s","speedup_vs_recorded_reference")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["ms_per_round"]); print("gibbs exact", d["gibbs"]["exact"]["ms_per_round"], d["gibbs"]["exact"]["workgroups_per_chain"], "sweep", d["gibbs"]["parallel"]["ms_per_sweep"])
PY
t=$(date +%s); timeout 900 tools/profile_round.sh r05r "C3" > $O/profile.log 2>&1; echo "profile rc=$? $(( $(date +%s) - t )) s"; tail -12 $O/profile.log | cut -c1-300
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null; find gpurun_out -name "*.db" -size +4M -delete 2>/dev/null
