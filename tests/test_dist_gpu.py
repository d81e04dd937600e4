"""Sharded EM through the C ABI: rows split by the reference's thread rule (rsem_em_shard_rows, EM.cpp:135-157), one
rsem_em_ctx per shard, rsem_em_run on every rank with one all-reduce of the counts per round (rsem_em_set_comm).

The GPU box has ONE GPU, and RCCL refuses two ranks on one device, so the sharded path is exercised here with the LOCAL
communicator (ranks = threads sharing GPU 0; same rsem_em_run code, same collective call sites) and the RCCL calls
themselves with a one-rank communicator that is forced to issue its collectives.  The result must equal a
single-context run on the whole matrix: same ROUND count, theta to 1e-9."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shards(wl, world):
    from rsem_amd import capi, dist as rd
    b = capi.em_shard_rows(wl["row_ptr"], world)
    assert b == rd.shard_rows(wl["row_ptr"], world)
    return [rd.take_shard(wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], b[r], b[r + 1]) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_em_run_equals_single_context(world):
    from rsem_amd import capi
    from tools.synth_data import make_em_workload
    wl = make_em_workload("small", seed=9, long_row_every=50000)
    M = wl["M"]
    ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    ref = ctx.run(wl["theta0"], wl["N0"], max_round=400)
    ctx.close()
    comms = capi.Comm.create_local([0] * world)
    out = [None] * world
    lines = []

    def rank(r, shard):
        rp, sid, cp, ncp = shard
        c = capi.EmContext(M, rp, np.ascontiguousarray(sid), np.ascontiguousarray(cp), np.ascontiguousarray(ncp), device=0)
        c.set_comm(comms[r])
        if r == 0:
            c.set_progress(lambda rd_, s, b, t: lines.append((rd_, s, b, t)))
        out[r] = c.run(wl["theta0"], wl["N0"], max_round=400)  # GLOBAL N0 on every rank
        c.close()

    ts = [threading.Thread(target=rank, args=(r, s)) for r, s in enumerate(_shards(wl, world))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in comms:
        c.close()
    assert all(o is not None for o in out)
    for o in out:
        assert o["rounds"] == ref["rounds"]
        assert np.allclose(o["theta"], ref["theta"], rtol=1e-9, atol=1e-18)
        assert np.array_equal(o["theta"], out[0]["theta"])  # every rank computes the same M step from the same sums
    assert np.allclose(out[0]["counts"], ref["counts"], rtol=1e-9, atol=1e-9)
    # one line per round, in order, SUM = N0 + reads with a non-zero normaliser over ALL shards
    assert [l[0] for l in lines] == list(range(1, ref["rounds"] + 1))
    assert abs(lines[-1][1] - (wl["N0"] + len(wl["row_ptr"]) - 1)) < 1e-6
    assert lines[-1][3] == out[0]["totNum"] and lines[-1][2] == out[0]["bChange"]


def test_rccl_calls_with_a_one_rank_communicator(monkeypatch):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce on the ctx stream, forced to run for a single rank."""
    from rsem_amd import capi
    from tools.synth_data import make_em_workload
    monkeypatch.setenv("RSEM_COMM_FORCE", "1")
    wl = make_em_workload("tiny", seed=3)
    uid = capi.Comm.unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    comm = capi.Comm.create(0, 0, 1, uid)
    assert (comm.rank, comm.world) == (0, 1)
    ctx = capi.EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    plain = ctx.run(wl["theta0"], wl["N0"], max_round=60)
    ctx.set_comm(comm)
    viarccl = ctx.run(wl["theta0"], wl["N0"], max_round=60)
    ctx.close()
    comm.close()
    # (floating-point atomics: two runs of the same ctx agree to the last few bits, not bit for bit)
    assert viarccl["rounds"] == plain["rounds"] and np.allclose(viarccl["theta"], plain["theta"], rtol=1e-12, atol=0)


def test_bench_forced_dist_line_on_one_gpu():
    """The N > 1 code path of bench.py, executed on HEAD every round on a one-GPU box (BENCH_FORCE_DIST=1: process group,
    communicator id through torch.distributed, RCCL all-reduce per round from C++, Gibbs chains dealt to ranks, the single
    final reduce) -- the driver's 8-GPU scaling run is the first time this path meets real ranks, so every round must at
    least have executed it and seen the keys a scaling run is read for."""
    import json
    import subprocess
    import sys
    import tempfile
    detail_path = os.path.join(tempfile.mkdtemp(), "detail.json")  # (the printed line is a summary since round 6: the full record goes here)
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_PORT="29631", BENCH_DETAIL_PATH=detail_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--scale", "0.1", "--legs=", "--no-cpu-baseline", "--no-ci",
                        "--steps", "5", "--warmup", "1", "--gibbs-sweeps", "6", "--gibbs-exact-rounds", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout[:1000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["roofline"]["frac"] > 0 and len(lines[0]) < 8000
    di = d["distributed"]
    d = json.load(open(detail_path))
    assert d["distributed"] == di or d["distributed"]["rccl_ranks"] == di["rccl_ranks"]
    assert di["rccl_ranks"] == 1 and len(di["estep_ms_per_rank"]) == 1 and di["frac_physical_per_rank"][0] > 0 and di["allreduce_ms"] > 0
    g = d["gibbs"]
    assert "error" not in g, g
    assert g["parallel"]["sweeps_per_s_per_rank"][0] > 0 and g["parallel"]["final_reduce_ms"] is not None and g["parallel"]["final_reduce_ms"] >= 0
    assert g["exact"]["final_reduce_ms"] is not None


# ---- real ranks: these run where the box has at least two GPUs (the driver's 8-GPU node), RCCL over xGMI --------------------------

def _two_gpus():
    try:
        from rsem_amd import capi
        return capi.device_count() >= 2
    except Exception:
        return False


needs_two_gpus = pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _rccl_ranks(world, body):
    """world threads of this process, rank r on GPU r, one RCCL communicator (ncclCommInitRank from every thread at once)."""
    from rsem_amd import capi
    uid = capi.Comm.unique_id()
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            comm = capi.Comm.create(r, r, world, uid)
            try:
                out[r] = body(r, comm)
            finally:
                comm.close()
        except BaseException as e:  # (a rank that dies must not leave the others waiting without a word)
            err[r] = e

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    assert not any(t.is_alive() for t in ts), "a rank is still waiting in a collective"
    for e in err:
        if e is not None:
            raise e
    return out


@needs_two_gpus
def test_em_on_two_gpus_one_allreduce_per_round_equals_single_context():
    """EM.cpp:385-389 across devices: rows split by the reference's rule, ONE ncclAllReduce of [counts | totals] per round on each
    rank's EM stream (comm.hip), every rank with the same theta and the same ROUND count as one context on the whole matrix."""
    from rsem_amd import capi
    from tools.synth_data import make_em_workload
    world = 2
    wl = make_em_workload("small", seed=9, long_row_every=50000)
    M = wl["M"]
    ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], device=0)
    ref = ctx.run(wl["theta0"], wl["N0"], max_round=400)
    ctx.close()
    shards = _shards(wl, world)

    def body(r, comm):
        assert (comm.rank, comm.world) == (r, world)
        rp, sid, cp, ncp = shards[r]
        c = capi.EmContext(M, rp, np.ascontiguousarray(sid), np.ascontiguousarray(cp), np.ascontiguousarray(ncp), device=r)
        c.set_comm(comm)
        o = c.run(wl["theta0"], wl["N0"], max_round=400)
        c.close()
        return o

    out = _rccl_ranks(world, body)
    for o in out:
        assert o["rounds"] == ref["rounds"]
        assert np.allclose(o["theta"], ref["theta"], rtol=1e-9, atol=1e-18)
        assert np.array_equal(o["theta"], out[0]["theta"])
    assert np.allclose(out[0]["counts"], ref["counts"], rtol=1e-9, atol=1e-9)


@needs_two_gpus
def test_gibbs_chains_dealt_to_two_gpus_one_reduce_equals_single_context():
    """Gibbs.cpp:211-254,372-388 across devices: chain k on rank k % 2 (each GPU advances its chains with teams of workgroups as
    large as ITS compute units allow for ITS number of chains -- the count vectors do not depend on the team), the accumulator
    sums meet in ONE ncclReduce on rank 0: equal to all chains in one context, count vectors chain for chain."""
    from rsem_amd import capi
    from rsem_amd.dist import gibbs_rank_chains
    from tools.synth_data import make_em_workload, to_gibbs_items
    world, nchains = 2, 6
    wl = make_em_workload("small", seed=4)
    M = wl["M"]
    irp, isid, icp = to_gibbs_items(wl)
    N1 = len(irp) - 1
    init = np.zeros(M + 1, np.int32)
    eel, mw, grp = np.full(M + 1, 700.0), np.ones(M + 1), np.array([1, M + 1], np.int32)
    totc = (M + 1) + wl["N0"] + N1
    seeds = capi.gibbs_chain_seeds(11, nchains)
    ns = [3, 3, 2, 2, 2, 2]
    mk = lambda dev: capi.GibbsContext(M, irp, isid, icp, init, None, 1.0, totc, wl["N0"], eel, mw, grp, device=dev)
    g = mk(0)
    cvs1, acc1, _, p1 = g.run_chains(capi.GIBBS_EXACT, seeds, 4, ns, 2)
    g.close()

    def body(r, comm):
        mine = gibbs_rank_chains(nchains, world, r)
        gr = mk(r)
        gr.set_comm(comm)
        cvs, acc, _, prof = gr.run_chains(capi.GIBBS_EXACT, [seeds[k] for k in mine], 4, [ns[k] for k in mine], 2)
        gr.close()
        return mine, cvs, acc, prof

    out = _rccl_ranks(world, body)
    for mine, cvs, _, prof in out:
        assert prof.chains == len(mine) and prof.team >= 1
        for k, cv in zip(mine, cvs):
            assert np.array_equal(cv, cvs1[k])
    assert out[0][3].reduce_ms > 0  # the one collective ran
    for a, b in zip(out[0][2], acc1):  # rank 0 holds the sums over all chains (added in another order: doubles, not bit for bit)
        assert np.allclose(a, b, rtol=1e-12, atol=1e-9)
