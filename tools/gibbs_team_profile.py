#!/usr/bin/env python3
"""Per-round time of the exact (reference-chain) Gibbs kernel by team size, one context, steady state:
    python tools/gibbs_team_profile.py [scale=0.2] [chains=8] [rounds=6] [config=C3] [teams=1,8,16,32,0]
Synthetic items of the given bench config (scale x its reads); `rounds` burn-in rounds, then 2 kept samples per chain.
Team size 0 = the product's own choice (compute units / chains, at most 64); 1 = one workgroup per chain (rounds 3-4).
Every run must give the same checksum (the count vectors are the same integers whatever the team size)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload, to_gibbs_items  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
config = sys.argv[4] if len(sys.argv) > 4 else "C3"
teams = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "1,8,16,32,0").split(",")]
wl = make_em_workload(config, scale=scale)
M = wl["M"]
irp, isid, icp = to_gibbs_items(wl)
N1 = len(irp) - 1
g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                      np.full(M + 1, 1000.0), np.ones(M + 1), np.array([1, M + 1], np.int32))
seeds = capi.gibbs_chain_seeds(1, chains)
ref = None
for W in teams:
    if W:
        os.environ["RSEM_GX_TEAM"] = str(W)
    else:
        os.environ.pop("RSEM_GX_TEAM", None)
    t0 = time.time()
    cvs, acc, _, p = g.run_chains(capi.GIBBS_EXACT, seeds, rounds, [2] * chains, 1, want_vectors=True)
    chk = float(np.dot(acc[0], np.arange(M + 1) % 97))
    same = True
    if ref is None:
        ref = [c.copy() for c in cvs]
    else:
        same = all(np.array_equal(a, b) for a, b in zip(cvs, ref))
    print("%s N1=%d items=%d chains=%d team=%d: %.2f ms/round, %.4f us per read visit and chain, checksum %.6f, count vectors %s (%.1f s)" % (
        config, N1, len(isid), chains, W, p.sweep_ms, p.sweep_ms * 1e3 / N1, chk, "same" if same else "DIFFERENT", time.time() - t0), flush=True)
g.close()
