"""Host-side sharding rules for the multi-GPU paths (one process per GPU, torch.distributed).

EM   : reads are independent given theta -> contiguous read ranges balanced by alignment count,
       exactly the reference's thread split (EM.cpp:135-157); one all-reduce of the M+1 counts per
       round (the reference's serial reduction countvs[0] += countvs[i], EM.cpp:385-389).
Gibbs: chains are independent -> rank r runs the chain thread r of the reference would run
       (Gibbs.cpp:211-223: NSAMPLES split, per-chain seed from the seed engine, sampling.h:19-44);
       one reduce of the accumulators at the end (release(), Gibbs.cpp:372-388).
"""
import numpy as np


def shard_rows(row_ptr, world):
    """EM.cpp:135-157: thread i takes reads until it holds >= nHits/T hits (last thread takes the rest);
    every later thread is left at least one read.  Returns world+1 read boundaries."""
    N1 = len(row_ptr) - 1
    nhits = int(row_ptr[-1])
    nhT = nhits // world
    bounds = [0]
    cur = 0
    for i in range(world):
        left_threads = world - i - 1
        if i == world - 1:
            cur = N1
        else:
            # smallest cur' with hits(cur..cur') >= nhT, but leave >= left_threads reads
            target = int(row_ptr[cur]) + nhT
            nxt = int(np.searchsorted(row_ptr, target, side="left"))
            nxt = max(nxt, cur)
            nxt = min(nxt, N1 - left_threads)
            cur = max(cur, nxt)
        bounds.append(cur)
    return bounds


def take_shard(row_ptr, sid, conprb, ncp, lo, hi):
    a, b = int(row_ptr[lo]), int(row_ptr[hi])
    rp = (row_ptr[lo:hi + 1] - row_ptr[lo]).astype(np.uint64)
    return rp, sid[a:b], (None if conprb is None else conprb[a:b]), (None if ncp is None else ncp[lo:hi])


def gibbs_chain_plan(nsamples, world):
    """Gibbs.cpp:215-223: samples per chain."""
    q, left = divmod(nsamples, world)
    return [q + (1 if r < left else 0) for r in range(world)]


def gibbs_rank_chains(nchains, world, rank):
    """Chains are dealt round-robin to the GPUs (rsem-run-gibbs: chain k runs on group k % world, which also writes
    imdName.countvectors<k>, Gibbs.cpp:225-226,257-262)."""
    return list(range(rank, nchains, world))
