"""world_size-2 gloo tests of the multi-GPU host logic (sharding + collectives), CPU only.
The per-shard compute is stood in for by the oracle (tests may use it); the product's GPU kernels
are exercised by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rsem_files as rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _em_worker(rank, world, port, fxname, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle as orc
    from rsem_amd import dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = rf.fixture(fxname)
    M, N0, rpi, sidi, vali = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    rp, sid, cp, ncp = rf.split_noise(rpi, sidi, vali)
    raw, _ = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    b = rd.shard_rows(rp, world)
    srp, ssid, scp, sncp = rd.take_shard(rp, sid, cp, ncp, b[rank], b[rank + 1])
    theta = raw.copy()
    for _ in range(3):
        counts = orc.em_estep(M, srp, np.ascontiguousarray(ssid), np.ascontiguousarray(scp), np.ascontiguousarray(sncp), theta)
        t = torch.from_numpy(counts)
        dist.all_reduce(t)  # EM.cpp:385-389
        _, theta, s, bc, tn = orc.em_mstep(M, N0, t.numpy(), theta)
    if rank == 0:
        q.put((theta, s))
    dist.barrier()
    dist.destroy_process_group()


def test_em_row_sharding_gloo():
    from oracle import pyoracle as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_em_worker, args=(r, world, port, "pe_q", q)) for r in range(world)]
    for p in procs:
        p.start()
    theta_d, s_d = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    fx = rf.fixture("pe_q")
    M, N0, rpi, sidi, vali = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    rp, sid, cp, ncp = rf.split_noise(rpi, sidi, vali)
    theta, _ = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    for _ in range(3):
        counts = orc.em_estep(M, rp, sid, cp, ncp, theta)
        _, theta, s, bc, tn = orc.em_mstep(M, N0, counts, theta)
    assert np.allclose(theta_d, theta, rtol=1e-12, atol=0)
    assert abs(s - s_d) < 1e-9


def test_shard_rows_matches_reference_rule():
    from rsem_amd import dist as rd
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 9, 1000)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for T in (1, 2, 3, 8):
        b = rd.shard_rows(rp, T)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:]))
        # literal restatement of the while loop in EM.cpp:139-153
        nhT, left, cur, exp = int(rp[-1]) // T, 1000, 0, [0]
        for i in range(T):
            ntLeft, hits = T - i - 1, 0
            while left > ntLeft and (i == T - 1 or hits < nhT):
                hits += int(lens[cur]); cur += 1; left -= 1
            exp.append(cur)
        assert b == exp
        from rsem_amd import capi
        assert capi.em_shard_rows(rp, T) == exp  # the C ABI's restatement (what rsem-run-em --ngpus uses)
    # fewer alignments than workers, and fewer reads than workers
    from rsem_amd import capi
    for lens2, T in ((np.array([1, 1, 1]), 8), (np.array([5, 1, 1, 9, 2]), 4), (np.array([3]), 2)):
        rp2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
        n = len(lens2)
        nhT, left, cur, exp = int(rp2[-1]) // T, n, 0, [0]
        for i in range(T):
            ntLeft, hits = T - i - 1, 0
            while left > ntLeft and (i == T - 1 or hits < nhT):
                hits += int(lens2[cur]); cur += 1; left -= 1
            exp.append(cur)
        assert capi.em_shard_rows(rp2, T) == exp, (lens2, T)


def test_gibbs_chain_plan():
    from rsem_amd import dist as rd
    assert rd.gibbs_chain_plan(1000, 8) == [125] * 8
    assert rd.gibbs_chain_plan(10, 4) == [3, 3, 2, 2]
    assert sum(rd.gibbs_chain_plan(1001, 8)) == 1001


def _gibbs_worker(rank, world, port, fxname, nchains, nsamples, tmp, q):
    """Rank r runs the chains k = r, r + world, ... (the product's deal), sums their accumulators in chain order, writes
    their count-vector files; ONE reduce to rank 0 at the end (release(), Gibbs.cpp:372-388)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle as orc
    from rsem_amd import dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = _gibbs_inputs(fxname)
    plan = rd.gibbs_chain_plan(nsamples, nchains)
    seeds = _chain_seeds(orc, 7, nchains)
    acc = None
    for k in rd.gibbs_rank_chains(nchains, world, rank):
        cv, a = orc.gibbs_chain(d["M"], d["rp"], d["sid"], d["val"], d["init"], None, 1.0, d["totc"], d["N0"], d["eel"], d["mw"], d["grp"],
                                seeds[k], 5, plan[k], 2)
        np.savetxt(os.path.join(tmp, "s.countvectors%d" % k), cv, fmt="%d")
        acc = a if acc is None else [x + y for x, y in zip(acc, a)]
    flat = torch.from_numpy(np.concatenate(acc))
    dist.reduce(flat, dst=0)
    if rank == 0:
        q.put(flat.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _gibbs_inputs(fxname):
    fx = rf.fixture(fxname)
    M, N0, rp, sid, val = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    N1 = len(rp) - 1
    return dict(M=M, N0=N0, rp=rp, sid=sid, val=val, init=np.zeros(M + 1, np.int32), totc=float((M + 1) + N0 + N1),
                eel=np.full(M + 1, 300.0), mw=np.ones(M + 1), grp=np.array([1, M + 1], np.int32))


def _chain_seeds(orc, seed, n):
    from rsem_amd import capi
    return capi.gibbs_chain_seeds(seed, n)


def test_gibbs_chain_sharding_gloo(tmp_path):
    """The split north_star names: independent chains per GPU, one reduce at the end.  world_size 2, 5 chains with unequal
    numbers of samples: the reduced accumulators and the count-vector files equal the one-process run."""
    from oracle import pyoracle as orc
    from rsem_amd import dist as rd
    world, nchains, nsamples = 2, 5, 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gibbs_worker, args=(r, world, port, "se_q", nchains, nsamples, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d = _gibbs_inputs("se_q")
    plan = rd.gibbs_chain_plan(nsamples, nchains)
    assert plan == [3, 3, 3, 2, 2]
    seeds = _chain_seeds(orc, 7, nchains)
    assert sorted(k for r in range(world) for k in rd.gibbs_rank_chains(nchains, world, r)) == list(range(nchains))
    acc = None
    for k in range(nchains):
        cv, a = orc.gibbs_chain(d["M"], d["rp"], d["sid"], d["val"], d["init"], None, 1.0, d["totc"], d["N0"], d["eel"], d["mw"], d["grp"],
                                seeds[k], 5, plan[k], 2)
        acc = a if acc is None else [x + y for x, y in zip(acc, a)]
        own = np.loadtxt(os.path.join(str(tmp_path), "s.countvectors%d" % k), dtype=np.int64, ndmin=2)
        assert np.array_equal(own, cv)  # every chain's file exists exactly once, written by the rank that ran it
    want = np.concatenate(acc)
    nM = d["M"] + 1
    assert np.array_equal(got[:2 * nM], want[:2 * nM])          # counts and squared counts: integers in doubles, exact
    assert np.allclose(got[2 * nM:], want[2 * nM:], rtol=1e-12, atol=0)  # tpm / fpkm / group sums: summation order differs
    assert abs(got[:nM].sum() - nsamples * (d["N0"] + len(d["rp"]) - 1)) < 1e-6
