"""Q32 value planes: what the rounding does to a whole EM run -- the oracle (CPU) run to convergence on the doubles and on
tools/q32_ref.quantize_q32(values):  python tools/q32_accuracy_study.py C2 1.0   (profiles/r02e_q32_accuracy_c2_full_size.log)"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import pyoracle as orc
from tools.q32_ref import quantize_q32
from tools.synth_data import make_em_workload
cfg, scale = sys.argv[1], float(sys.argv[2])
wl = make_em_workload(cfg, scale=scale)
M = wl["M"]
print("N1", len(wl["row_ptr"]) - 1, "nnz", len(wl["sid"]), flush=True)
t0 = time.time()
th, r, b, t = orc.em_run(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"])
print(f"f64: rounds {r} totNum {t} ({time.time()-t0:.0f} s)", flush=True)
for D in (8,):
    q, ok = quantize_q32(wl["row_ptr"], wl["conprb"], D)
    t0 = time.time()
    th2, r2, _, t2 = orc.em_run(M, wl["row_ptr"], wl["sid"], q, wl["ncp"], wl["N0"], wl["theta0"])
    big = th >= 1e-7
    rel = np.abs(th2 - th)[big] / th[big]
    print(f"q32 D={D}: compressed {ok.mean():.4f} rounds {r2} totNum {t2} max rel dtheta {rel.max():.3e} rms {np.sqrt((rel**2).mean()):.3e} n(theta>=1e-7) {big.sum()} ({time.time()-t0:.0f} s)", flush=True)
