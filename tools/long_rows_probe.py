#!/usr/bin/env python3
"""The cliff at 256 alignments per read: reads with more stay in the caller-order CSR and are walked by k_estep_csr (a thread per read)
beside the lane kernel.  BASELINE configs[1]'s matrix (10 M reads) with every k-th read replaced by one of 300 .. 2000 alignments
(a Trinity-shaped tail): E-step time per round against the share of such reads.
    python tools/long_rows_probe.py [scale=1.0] [every=0,100000,10000,1000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
everys = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,100000,10000,1000").split(",")]
wl = make_em_workload("C2", scale=scale)
M, rp, sid, cp, ncp = wl["M"], wl["row_ptr"].astype(np.int64), wl["sid"], wl["conprb"], wl["ncp"]
N1 = len(rp) - 1
for every in everys:
    if every:
        rng = np.random.default_rng(every)
        rows = np.arange(every // 2, N1, every)
        lens = (rp[1:] - rp[:-1]).copy()
        newl = rng.integers(300, 2001, len(rows))
        lens[rows] = newl
        nrp = np.zeros(N1 + 1, np.int64)
        nrp[1:] = np.cumsum(lens)
        nsid = np.empty(int(nrp[-1]), np.int32)
        ncpv = np.empty(int(nrp[-1]), np.float64)
        keep = np.ones(N1, bool)
        keep[rows] = False
        # the unchanged rows: copied run by run (vectorised through repeat)
        src0, dst0, ln = rp[:-1][keep], nrp[:-1][keep], (rp[1:] - rp[:-1])[keep]
        idx = np.repeat(dst0 - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(int(ln.sum()))
        sidx = np.repeat(src0 - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(int(ln.sum()))
        nsid[idx] = sid[sidx]
        ncpv[idx] = cp[sidx]
        for r, l in zip(rows, newl):
            a = int(nrp[r])
            nsid[a:a + l] = np.sort(rng.choice(M, l, replace=False) + 1)
            ncpv[a:a + l] = 10.0 ** rng.uniform(-4, 0, l)
        w = (nrp.astype(np.uint64), nsid, ncpv)
    else:
        w = (rp.astype(np.uint64), sid, cp)
    ctx = capi.EmContext(M, w[0], w[1], w[2], ncp)
    out = ctx.run(wl["theta0"], wl["N0"], max_round=40, min_round=40, profile=True)
    p = out["profile"]
    print("every %7d: %d reads, %d alignments, reads left in the CSR %d (%.4f %%), their alignments %.2f %% of all: %.4f ms per E step" % (
        every, N1, len(w[1]), ctx.info("reads_long"), 100.0 * ctx.info("reads_long") / N1,
        100.0 * (len(w[1]) - len(sid)) / max(len(w[1]), 1) if every else 0.0, p.estep_ms_sum / max(p.estep_launches, 1)), flush=True)
    ctx.close()
