#!/usr/bin/env python3
"""CPU study (oracle only, no GPU): does the data-augmentation sampler with the pipeline's chain lengths reach the same
posterior means as the reference's collapsed chain?  Truth = long collapsed chains; then 64 short collapsed chains
(the reference with -p 64, BURNIN 200, 1000 samples) and 64 short DA chains at several `thin` values.

  python tools/gibbs_mixing_study.py <dir with ref.* temp/s.ofg>   (e.g. made by gen_temp + rsem-run-em --gibbs-out)
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as orc  # noqa: E402
from tests import rsem_files as rf  # noqa: E402

D = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M, N0, row_ptr, sid, val = rf.read_ofg(os.path.join(D, "temp", "s.ofg"))
N1 = len(row_ptr) - 1
init = np.zeros(M + 1, np.int32)
eel, mw, grp = np.full(M + 1, 1000.0), np.ones(M + 1), np.array([1, M + 1], np.int32)
totc = (M + 1) + N0 + N1


def collapsed(args):
    seed, burnin, ns = args
    cv, acc = orc.gibbs_chain(M, row_ptr, sid, val, init, None, 1.0, totc, N0, eel, mw, grp, seed, burnin, ns, 1)
    return acc[0], acc[1], ns


def da(args):
    seed, burnin, ns, thin = args
    a, b = orc.gibbs_da_chain(M, row_ptr, sid, val, init, 1.0, N0, seed, burnin, ns, 1, thin)
    return a, b, ns


def pooled(fn, jobs):
    with Pool(8) as p:
        res = p.map(fn, jobs)
    n = sum(r[2] for r in res)
    m = sum(r[0] for r in res) / n
    v = sum(r[1] for r in res) / n - m * m
    return m, np.sqrt(np.maximum(v, 0))


if __name__ == "__main__":
    seeds = orc.chain_seeds(5, P)
    q, left = 1000 // P, 1000 % P
    ns = [q + (1 if k < left else 0) for k in range(P)]
    print("M=%d N1=%d items=%d" % (M, N1, len(sid)), flush=True)
    truth, sd = pooled(collapsed, [(101 + k, 2000, 2000) for k in range(8)])
    truth2, _ = pooled(collapsed, [(201 + k, 2000, 2000) for k in range(8)])

    def report(name, m):
        qv = np.abs(m - truth) / (sd + 0.5)
        print("%-40s |diff|/(sd+0.5): median %.4f  99%% %.4f  max %.4f  rms %.4f  corr %.8f" % (
            name, np.median(qv), np.percentile(qv, 99), qv.max(), np.sqrt((qv ** 2).mean()), np.corrcoef(m, truth)[0, 1]), flush=True)
        return qv

    report("second set of long collapsed chains", truth2)
    ref, _ = pooled(collapsed, [(int(seeds[k]), 200, ns[k]) for k in range(P)])
    qref = report("collapsed, %d chains x (200 + %d)" % (P, q), ref)
    for thin in (1, 8, 32):
        m, _ = pooled(da, [(int(seeds[k]), 200, ns[k], thin) for k in range(P)])
        qd = report("DA thin %d, %d chains x (200 + %d)" % (thin, P, q), m)
        worst = np.argsort(-qd)[:5]
        print("   worst transcripts:", [(int(j), round(float(truth[j]), 1), round(float(m[j]), 1), round(float(ref[j]), 1), round(float(sd[j]), 1)) for j in worst], "(sid, truth, DA, collapsed-short, sd)", flush=True)
