#!/usr/bin/env python3
"""The E step with and without the foreign side path on a workload: python tools/foreign_probe.py CONFIG [scale]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "C3X"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
wl = make_em_workload(config, scale=scale)
M, nnz = wl["M"], len(wl["sid"])
ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
ref = None
for mode in (0, 1, -1):
    t0 = time.perf_counter()
    ctx.set_option("foreign_side_path", mode)
    relist_s = time.perf_counter() - t0
    nf = ctx.info("foreign_alignments")
    ctx.run(wl["theta0"], wl["N0"], min_round=5, max_round=5)
    out = ctx.run(wl["theta0"], wl["N0"], min_round=40, max_round=40, profile=True)
    p = out["profile"]
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    if ref is None:
        ref = counts
    err = float(np.max(np.abs(counts - ref) / np.maximum(np.abs(ref), 1e-6)))
    print("%s x%g mode %2d: listed %d of %d (%.2f %%), relist %.2f s, E-step launch %.4f ms, round %.4f ms, step vs mode 0: %.2e" % (
        config, scale, mode, nf, nnz, 100.0 * nf / max(nnz, 1), relist_s, p.estep_ms_sum / max(p.estep_launches, 1), p.total_ms / max(p.rounds, 1), err))
ctx.close()
