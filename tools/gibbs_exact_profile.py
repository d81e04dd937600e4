#!/usr/bin/env python3
"""Where does the exact (reference-chain) Gibbs kernel spend its time?   python tools/gibbs_exact_profile.py [scale] [chains] [rounds]
Synthetic C2-shaped items (scale x 10 M reads); per-round time of k_gibbs_exact_coop in its steady state (after `rounds`
burn-in rounds) with the kernel's debug switches: all on / no commit loop / no draw either; plus the commit loop's counters."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 4 and sys.argv[4] == "child":
    from rsem_amd import capi
    from tools.synth_data import make_em_workload, to_gibbs_items
    scale, chains, rounds = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    wl = make_em_workload("C2", scale=scale)
    M = wl["M"]
    irp, isid, icp = to_gibbs_items(wl)
    N1 = len(irp) - 1
    g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                          np.full(M + 1, 1000.0), np.ones(M + 1), np.array([1, M + 1], np.int32))
    _, _, _, p = g.run_chains(capi.GIBBS_EXACT, capi.gibbs_chain_seeds(1, chains), rounds, [2] * chains, 1, want_vectors=False)
    print("N1=%d items=%d chains=%d: %.2f ms/round, %.4f us per read visit and chain" % (N1, len(isid), chains, p.sweep_ms, p.sweep_ms * 1e3 / N1))
    sys.exit(0)

scale = sys.argv[1] if len(sys.argv) > 1 else "0.1"
chains = sys.argv[2] if len(sys.argv) > 2 else "8"
rounds = sys.argv[3] if len(sys.argv) > 3 else "40"
for dbg, what in ((4, "full kernel (+ counters)"), (0, "full kernel"), (1, "no commit loop"), (3, "no draw, no commit loop")):
    env = dict(os.environ, RSEM_GIBBS_EXACT_DEBUG=str(dbg))
    r = subprocess.run([sys.executable, __file__, scale, chains, rounds, "child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print("%-28s %s %s" % (what, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else "(no output)", " | ".join(l for l in r.stderr.split("\n") if "gibbs exact" in l)))
