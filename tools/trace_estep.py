#!/usr/bin/env python3
"""Per-workgroup timeline of one E-step launch (rsem_em_debug_trace): python tools/trace_estep.py [scale]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
wl = make_em_workload("C2", scale=scale)
ctx = capi.EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], device=0)
ctx.run(wl["theta0"], wl["N0"], min_round=3, max_round=3)  # first use: the units are re-sorted by measured lifetime
L = capi.lib()
L.rsem_em_debug_trace.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.uint64), C.POINTER(C.c_uint32)]
cap = 1 << 20
buf = np.zeros(2 * cap, np.uint64)
n = C.c_uint32(cap)
rc = L.rsem_em_debug_trace(ctx._h, wl["theta0"], buf, C.byref(n))
assert rc == 0, L.rsem_hip_last_error()
n = n.value
t = buf[: 2 * n].reshape(n, 2).astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
st, en = t[:, 0] - t0, t[:, 1] - t0
total = en.max()
life = en - st
print("units %d, kernel span %.1f us; unit lifetime mean %.1f us (min %.1f, max %.1f)" % (n, total, life.mean(), life.min(), life.max()))
edges = np.linspace(0, total, 21)
print("time window (us)   running WGs (avg)   starts   ends")
for a, b in zip(edges[:-1], edges[1:]):
    running = (np.minimum(en, b) - np.maximum(st, a)).clip(min=0).sum() / (b - a)
    print("%6.1f-%6.1f        %7.1f          %5d   %5d" % (a, b, running, ((st >= a) & (st < b)).sum(), ((en >= a) & (en < b)).sum()))
print("first 1024 starts within %.1f us; last end - 99th pct end = %.1f us; time with < 512 WGs running at the end: %.1f us"
      % (np.sort(st)[min(1023, n - 1)], total - np.percentile(en, 99), total - np.sort(en)[max(0, n - 512)]))
ctx.close()
