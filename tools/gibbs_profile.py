#!/usr/bin/env python3
"""Time / profile the PARALLEL Gibbs sweep on a synthetic matrix: python tools/gibbs_profile.py [scale] [sweeps] [config=C2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload, to_gibbs_items  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
config = sys.argv[3] if len(sys.argv) > 3 else "C2"
wl = make_em_workload(config, scale=scale)
M = wl["M"]
cache = os.environ.get("RSEM_WL_CACHE")
cdir = os.path.join(cache, "gibbs_%s_%g" % (config, scale)) if cache else None
if cdir and os.path.exists(os.path.join(cdir, "icp.npy")):
    irp, isid, icp = (np.load(os.path.join(cdir, k + ".npy")) for k in ("irp", "isid", "icp"))
else:
    irp, isid, icp = to_gibbs_items(wl)
    if cdir:
        os.makedirs(cdir + ".tmp", exist_ok=True)
        for k, a in (("irp", irp), ("isid", isid), ("icp", icp)):
            np.save(os.path.join(cdir + ".tmp", k + ".npy"), a)
        os.rename(cdir + ".tmp", cdir)
N1 = len(irp) - 1
t0 = time.time()
g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                      np.full(M + 1, 1000.0), np.ones(M + 1), np.arange(1, M + 2, 5, dtype=np.int32) if M % 5 == 0 else np.array([1, M + 1], np.int32))
print("create %.2f s" % (time.time() - t0))
cv, acc, ms = g.run(capi.GIBBS_PARALLEL, 1, sweeps - 2, 2, 1, thin=1, want_vectors=False)
nitems = len(isid)
bytes_sweep = 12 * (nitems - N1) + 16 * N1  # conprb + sid per alignment, noise conprb + (unused) row pointer per read
print("N1=%d items=%d: %.3f ms/sweep, %.1f G items/s, algorithmic %.2f GB/sweep -> %.2f TB/s, checksum %.3f" % (
    N1, nitems, ms, nitems / ms / 1e6, bytes_sweep / 1e9, bytes_sweep / ms / 1e9, float(np.dot(acc[0], np.arange(M + 1) % 97))))
