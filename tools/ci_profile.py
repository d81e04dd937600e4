#!/usr/bin/env python3
"""Time the credibility-interval path at BASELINE scale: python tools/ci_profile.py [M] [nCV] [nSpC]
(defaults: C2's 50 000 transcripts, rsem-calculate-expression's 1000 count vectors x 50 draws)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsem_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
nCV = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nSpC = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(3)
mean = np.exp(rng.normal(3.0, 2.5, M + 1)) * (rng.random(M + 1) > 0.3)
cv = rng.poisson(mean, size=(nCV, M + 1)).astype(np.int32)
eel = np.concatenate([[0.0], rng.uniform(300, 4000, M)])
mw = np.ones(M + 1)
sizes = rng.integers(1, 10, M)
starts = np.concatenate([[1], 1 + np.cumsum(sizes)])
starts = starts[starts < M + 1]
starts = np.concatenate([starts, [M + 1]]).astype(np.int32)
t0 = time.time()
out = capi.ci_calculate(cv, nSpC, eel, mw, starts, 0.95, 1.0, seed=1)
wall = time.time() - t0
p = out["profile"]
nS = nCV * nSpC
print("M=%d genes=%d nSamples=%d: wall %.2f s (device total %.1f ms: draws %.1f ms = %.1f G gamma/s; sort %.1f ms = %.2f G keys/s; "
      "intervals %.1f ms)" % (M, len(starts) - 1, nS, wall, p.total_ms, p.sample_ms, p.n_draws / p.sample_ms / 1e6, p.sort_ms,
                             p.n_keys_sorted / p.sort_ms / 1e6, p.interval_ms))
print("matrix %.2f GB; sample stage writes %.2f GB -> %.2f TB/s" % (M * nS * 4 / 1e9, M * nS * 4 / 1e9, M * nS * 4 / p.sample_ms / 1e9))
