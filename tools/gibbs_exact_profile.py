#!/usr/bin/env python3
"""Per-round time of the exact (reference-chain) Gibbs kernels in their steady state:
    python tools/gibbs_exact_profile.py [scale] [chains] [rounds] [config=C2] [impls=wg,coop]
Synthetic items of the given bench config (scale x its reads); `rounds` burn-in rounds, then 2 kept samples.
impl = wg (workgroup per chain, the default), coop (one wave per chain), serial (lane 0 walks)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 6 and sys.argv[6] == "child":
    from rsem_amd import capi
    from tools.synth_data import make_em_workload, to_gibbs_items
    scale, chains, rounds, config = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    wl = make_em_workload(config, scale=scale)
    M = wl["M"]
    irp, isid, icp = to_gibbs_items(wl)
    N1 = len(irp) - 1
    g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                          np.full(M + 1, 1000.0), np.ones(M + 1), np.array([1, M + 1], np.int32))
    _, acc, _, p = g.run_chains(capi.GIBBS_EXACT, capi.gibbs_chain_seeds(1, chains), rounds, [2] * chains, 1, want_vectors=False)
    print("%s N1=%d items=%d chains=%d: %.2f ms/round, %.4f us per read visit and chain, checksum %.6f" % (
        config, N1, len(isid), chains, p.sweep_ms, p.sweep_ms * 1e3 / N1, float(np.dot(acc[0], np.arange(M + 1) % 97))))
    sys.exit(0)

scale = sys.argv[1] if len(sys.argv) > 1 else "0.1"
chains = sys.argv[2] if len(sys.argv) > 2 else "8"
rounds = sys.argv[3] if len(sys.argv) > 3 else "10"
config = sys.argv[4] if len(sys.argv) > 4 else "C2"
impls = (sys.argv[5] if len(sys.argv) > 5 else "wg,coop").split(",")
for impl in impls:
    env = dict(os.environ, RSEM_GIBBS_EXACT_IMPL=impl)
    r = subprocess.run([sys.executable, __file__, scale, chains, rounds, config, impl, "child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print("%-7s %s %s" % (impl, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else "(no output) " + r.stderr[-400:], " | ".join(l for l in r.stderr.split("\n") if "gibbs exact" in l)))
