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
    def info(self, k): return {"value_plane_bytes": 1000 if self.opts["value_bits"] == 64 else 520, "reads_q32": 9, "value_range_bits": 8, "far_units": 1, "units": 7,
                               "physical_bytes_per_launch": 1500 if self.opts["value_bits"] == 64 else 1020, "sid_plane_bytes_loaded": 100, "sid_plane_bytes": 500,
                               "slots": 20, "slices": 5, "window_entries": 10}[k]
    def set_option(self, k, v): self.opts[k] = v
    def set_comm(self, c): pass
    def close(self): pass
capi.EmContext = FakeCtx
def boom(*a, **k): raise RuntimeError("stand-in: no device")
capi.GibbsContext = boom
%(extra)s
capi.ci_calculate = boom
capi.stream_probe = lambda device=0, nbytes=0, reps=0: (6000.0, 5000.0)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(%(root)r, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.MIN_TIMED_S = 0.0
sys.argv = ["bench.py", "--config", "tiny", "--legs", "tinyR,tiny@0.5", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + %(argv)r
b.main()
'''

# stand-ins of the N > 1 run: a communicator whose all-reduce does nothing, a Gibbs context that reports rank-dependent times
DIST_EXTRA = r'''
class FakeComm:
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"])
    @staticmethod
    def unique_id(): return b"x" * 128
    @classmethod
    def create(cls, device, rank, world, uid): return cls()
    def allreduce(self, ptr, n, stream=0): pass
    def close(self): pass
capi.Comm = FakeComm
class GP:
    def __init__(self, sweep, red): self.sweep_ms, self.reduce_ms, self.team = sweep, red, 32
class FakeGibbs:
    def __init__(self, *a, **k): pass
    def set_comm(self, c): pass
    def run_chains(self, mode, seeds, *a, **k):
        r = int(os.environ["RANK"])
        return None, [np.ones(3)], None, GP(2.0 + r, 0.25)
    def close(self): pass
capi.GibbsContext = FakeGibbs
capi.gibbs_chain_seeds = lambda seed, n: list(range(n))
'''


def _run_driver(tmp_path, extra="", argv=()):
    detail = os.path.join(str(tmp_path), "detail.json")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "extra": extra, "argv": list(argv)}], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300, env=dict(os.environ, BENCH_DETAIL_PATH=detail))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    with open(detail) as f:
        return lines[0], json.loads(lines[0]), json.load(f)


def test_bench_main_prints_one_contract_line(tmp_path):
    raw, d, det = _run_driver(tmp_path)
    # the driver keeps the last 8 187 bytes of stdout: the line must fit in them whole (round 5's 22.5 KB line was not parsed)
    assert len(raw) < 8000, len(raw)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "detail"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch", "frac_algorithmic"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    # the contract's fraction is the physical one (the layout's own bytes over its own launch time: never above 1 for a kernel that
    # moves what its layout says); the formula's sits beside it
    assert abs(d["roofline"]["frac"] - 1500 / 1e-3 / 1e9 / 8000.0) < 1e-9 and abs(d["roofline"]["achieved"] - d["roofline"]["frac"] * 8000.0) < 1e-3
    assert d["roofline"]["frac_algorithmic"] > 0 and d["roofline"]["physical_bytes_per_launch"] == 1500 and d["roofline"]["stream_read_GBps"] == 6000.0
    # one-number summaries of the legs; a failing side leg does not take the EM line with it
    assert set(d["legs"]) == {"tinyR", "tiny@0.5"} and d["legs"]["tinyR"]["frac"] > 0 and d["legs"]["tinyR"]["ms"] == 1.0
    assert "error" in d["gibbs"] and "error" in d["ci"]
    # the parity check beside the measurement really compares with the oracle (the stand-in returns nonsense counts)
    assert d["checks"]["parity_one_step"]["ok"] is False and d["legs"]["tinyR"]["parity_ok"] is False
    assert d["q32"]["estep_avg_launch_ms"] == 1.0 and abs(d["q32"]["frac_physical"] - 1020 / 1e-3 / 1e9 / 8000.0) < 1e-9
    # everything else is in the side file the line names
    assert det["metric"] == d["metric"] and det["steps"] == 3
    assert det["roofline"]["physical"]["parts"]["sid_planes_loaded"] == 100 and "note" in det["roofline"] and "frac_of_traffic" in det["roofline"]
    assert det["roofline"]["stream"]["read_GBps"] == 6000.0 and det["roofline"]["achieved_over_stream_read"] > 0
    assert det["roofline"]["frac"] == det["roofline"]["frac_physical"] == 1500 / 1e-3 / 1e9 / 8000.0
    assert det["q32_value_planes"]["value_bits"] == 32 and "10000 reads" in det["other_configs"]["tiny@0.5"]["workload"]
    for leg in det["other_configs"].values():
        assert leg["frac_physical"] > 0 and leg["physical"]["physical_bytes_per_launch"] == 1500
    assert det["q32_value_planes"]["frac_physical"] == 1020 / 1e-3 / 1e9 / 8000.0


def test_contract_line_of_a_full_run_fits_the_drivers_tail():
    """The line assembled from a whole default run's record (round 5's 22.5 KB one, committed) stays below 8 000 bytes and keeps
    roofline.frac and cpu_baseline.value; a record ten times as wordy still does."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    with open(os.path.join(ROOT, "profiles", "r05y_bench_default_final.json")) as f:
        det = json.load(f)
    raw = b.contract_line(det, "profiles/bench_detail_latest.json")
    d = json.loads(raw)
    assert len(raw) < 8000 and "\n" not in raw
    assert 0 < d["roofline"]["frac"] <= 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] == 64
    assert d["e2e"]["measured"]["speedup"] > 1 and d["e2e"]["full_size"]["dropin_s"] > 0 and d["checks"]["parity_one_step"]["ok"] is True
    assert set(d["legs"]) == {"C2", "C2R", "C3X", "C3X30", "C5"} and d["gibbs"]["exact"]["ms_per_round"] > 0
    det["other_configs"] = {"leg%d" % i: det["other_configs"]["C2"] for i in range(200)}
    raw = b.contract_line(det, "x.json")
    d = json.loads(raw)
    assert len(raw) < 8000 and d["legs"] == {"see": "detail"} and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_line_carries_per_rank_numbers(tmp_path):
    """The N > 1 control flow of bench.py as two gloo processes on the CPU (BENCH_DIST_BACKEND=gloo; the C-ABI wrappers and
    the communicator are stand-ins): rank 0 prints ONE line, n_gpus = 2, value = the sum over ranks, and the line carries what
    a scaling run is read for -- per-rank E-step times and physical roofline fractions, per-rank sweeps/s of the PARALLEL Gibbs
    split (the split that scales: one chain per GPU) and the cost of its single final reduce."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DIST_BACKEND="gloo", BENCH_DETAIL_PATH=os.path.join(str(tmp_path), "detail%d.json" % rank))
        code = DRIVER % {"root": ROOT, "extra": DIST_EXTRA, "argv": ["--gpus", "2", "--no-ci"]}
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert not outs[1][0].strip()  # only rank 0 prints
    lines = [l for l in outs[0][0].split("\n") if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "other_configs" not in d and "cpu_baseline" not in d
    di = d["distributed"]
    assert di["rccl_ranks"] == 2 and len(di["estep_ms_per_rank"]) == 2 and len(di["frac_physical_per_rank"]) == 2
    assert all(x > 0 for x in di["frac_physical_per_rank"]) and di["allreduce_ms"] >= 0
    g = d["gibbs"]["parallel"]
    assert g["ms_per_sweep_per_rank"] == [2.0, 3.0] and g["ms_per_sweep"] == 3.0  # the slowest GPU sets the job's rate
    assert g["sweeps_per_s_per_rank"] == [500.0, 333.333] and g["sweeps_per_s_all_gpus"] == 666.667  # (six digits in the line)
    assert g["final_reduce_ms"] == 0.25 and len(g["frac_of_hbm_peak_per_rank"]) == 2
    assert d["gibbs"]["exact"]["final_reduce_ms"] == 0.25
    # the split north_star names: 8 chains in all, 4 per GPU here, the slowest GPU sets the rate
    es = d["gibbs"]["exact_strong"]
    assert es["chains_total"] == 8 and es["chains_per_gpu"] == 4 and es["gpus"] == 2 and es["ms_per_round"] == 3.0 and es["workgroups_per_chain"] == 32


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
