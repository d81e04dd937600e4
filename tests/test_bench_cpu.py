"""bench.py's control flow without a GPU: the C-ABI wrappers are replaced by stand-ins (in a subprocess), everything else is
the real script -- argument handling, the timed-region loop, the legs, the Q32 leg, the one JSON line on stdout.  Guards the
benchmark contract (keys, one line) against slips that would otherwise only show on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys, os, numpy as np
sys.path.insert(0, %(root)r)
import torch
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda: None
import rsem_amd.build as build
build.build = lambda *a, **k: None
import rsem_amd.capi as capi
class Prof: estep_ms_sum = 2.0; estep_launches = 2
class FakeCtx:
    def __init__(self, M, rp, sid, cp=None, ncp=None, device=0): self.opts = {"value_bits": 64}
    def run(self, th, N0, round0=0, min_round=1, max_round=1, profile=False):
        out = dict(theta=np.array(th, float) / np.sum(th), rounds=max_round)
        if profile: out["profile"] = Prof()
        return out
    def step(self, th, N0):
        n = len(th)
        return np.zeros(n) + 1e300, np.array(th, float), 1.0, 0.0, 0   # (nothing like the oracle's counts: parity must say so)
    def info(self, k): return {"value_plane_bytes": 1000 if self.opts["value_bits"] == 64 else 520, "reads_q32": 9, "value_range_bits": 8, "far_units": 1, "units": 7}[k]
    def set_option(self, k, v): self.opts[k] = v
    def set_comm(self, c): pass
    def close(self): pass
capi.EmContext = FakeCtx
def boom(*a, **k): raise RuntimeError("stand-in: no device")
capi.GibbsContext = boom
capi.ci_calculate = boom
capi.stream_probe = lambda device=0, nbytes=0, reps=0: (6000.0, 5000.0)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(%(root)r, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.MIN_TIMED_S = 0.0
sys.argv = ["bench.py", "--config", "tiny", "--legs", "tinyR,tiny@0.5", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"]
b.main()
'''


def test_bench_main_prints_one_contract_line():
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert d["q32_value_planes"]["value_bits"] == 32 and "tinyR" in d["other_configs"] and "10000 reads" in d["other_configs"]["tiny@0.5"]["workload"]
    assert "error" in d["gibbs"] and "error" in d["credibility_intervals"]  # the EM line survives a failing side leg
    # the parity check beside the measurement really compares with the oracle (the stand-in returns nonsense counts)
    assert d["checks"]["parity_one_step"]["ok"] is False and d["other_configs"]["tinyR"]["parity_one_step"]["ok"] is False
    assert d["roofline"]["stream"]["read_GBps"] == 6000.0 and d["roofline"]["achieved_over_stream_read"] > 0
    assert "frac_of_traffic" in d["roofline"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU,
    127.0.0.1); under a launcher whose world size differs from --gpus it refuses."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"], env=dict(env, BENCH_PRINT_LAUNCH="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    cmd = json.loads(r.stdout.strip().split("\n")[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
