#!/usr/bin/env python3
"""Fixture for tests/test_gibbs_gpu.py::test_parallel_sampler_is_no_further_from_long_chains_than_the_reference_setup.

Question: rsem-run-gibbs --gibbs-mode parallel is a different Markov chain than the reference's
collapsed sampler; is it at least as close to the posterior as the reference's OWN configuration (BURNIN 200, 1000
samples over -p 64 chains)?  "The posterior" = long collapsed chains.  All chains here are the oracle's restatement of
Gibbs.cpp (oracle/rsem_oracle.c::orc_gibbs_chain), which reproduces the reference's integer count vectors bit for bit
(tests/test_oracle_golden.py), on a deterministic synthetic input (tools/synth_data.py "small", first 200 k reads, 5 k
transcripts).  Written: tests/golden/gibbs_truth/truth.npz
   long_mean, long_sd : posterior mean / sd of the counts from 8 chains x (2000 burn-in + 500 samples)
   long2_mean         : a second, independent set of such chains (the Monte-Carlo noise floor of the comparison)
   ref_mean           : the reference configuration: 64 chains x (200 burn-in + 15/16 samples), 1000 samples
Run time: a few minutes on 8 cores.    python tests/golden/make_gibbs_truth.py
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N_READS, SEED = 200_000, 5


def items():
    from tools.synth_data import make_em_workload, to_gibbs_items
    wl = make_em_workload("small", seed=SEED)
    rp = wl["row_ptr"][:N_READS + 1]
    nz = int(rp[-1])
    sub = dict(wl, row_ptr=rp, sid=wl["sid"][:nz], conprb=wl["conprb"][:nz], ncp=wl["ncp"][:N_READS])
    irp, isid, icp = to_gibbs_items(sub)
    M = wl["M"]
    N0 = 10_000
    grp = np.arange(1, M + 2, 10, dtype=np.int32)
    if grp[-1] != M + 1:
        grp = np.append(grp, M + 1).astype(np.int32)
    return dict(M=M, irp=irp, isid=isid, icp=icp, N0=N0, init=np.zeros(M + 1, np.int32), pseudoC=1.0,
                totc=(M + 1) * 1.0 + N0 + N_READS, eel=np.full(M + 1, 700.0), mw=np.ones(M + 1), grp=grp)


def chain(args):
    from oracle import pyoracle as orc
    seed, burnin, ns = args
    d = items()
    _, acc = orc.gibbs_chain(d["M"], d["irp"], d["isid"], d["icp"], d["init"], None, d["pseudoC"], d["totc"], d["N0"], d["eel"], d["mw"],
                             d["grp"], seed, burnin, ns, 1)
    return acc[0], acc[1]


def run(seeds, burnin, ns_list):
    with Pool(min(8, os.cpu_count() or 1)) as p:
        res = p.map(chain, [(int(s), burnin, int(n)) for s, n in zip(seeds, ns_list)])
    n = sum(ns_list)
    s1 = sum(r[0] for r in res) / n
    s2 = sum(r[1] for r in res) / n
    return s1, np.sqrt(np.maximum(s2 - s1 * s1, 0.0) * n / (n - 1))


if __name__ == "__main__":
    from oracle import pyoracle as orc
    long_mean, long_sd = run(orc.chain_seeds(11, 8), 2000, [500] * 8)
    long2_mean, _ = run(orc.chain_seeds(12, 8), 2000, [500] * 8)
    ns = [1000 // 64 + (1 if k < 1000 % 64 else 0) for k in range(64)]
    ref_mean, _ = run(orc.chain_seeds(5, 64), 200, ns)
    out = os.path.join(ROOT, "tests", "golden", "gibbs_truth", "truth.npz")
    np.savez_compressed(out, long_mean=long_mean, long_sd=long_sd, long2_mean=long2_mean, ref_mean=ref_mean, n_reads=N_READS, seed=SEED)

    def dist(x):
        q = np.abs(x - long_mean) / (long_sd + 0.5)
        return "rms %.4f max %.4f" % (np.sqrt((q ** 2).mean()), q.max())
    print("second set of long chains:", dist(long2_mean))
    print("reference configuration  :", dist(ref_mean))
