"""Parity of the HIP Gibbs samplers (through the C ABI).

EXACT mode must reproduce the reference's integer count vectors bit for bit (same MT19937 stream,
same cumulative sums); PARALLEL mode is a different chain for the same posterior and is held to a
sampling tolerance against a long reference-equivalent (oracle) chain.
"""
import os

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def capi():
    from rsem_amd import capi as c
    return c


def _load(name):
    fx = rf.fixture(name)
    M, N0, rp, sid, val = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    model = rf.read_model(os.path.join(fx, "stat", "s.model"))
    full, tot = rf.read_seq_lens(os.path.join(fx, "ref.seq"))
    grp = rf.read_grp(os.path.join(fx, "ref.grp"))
    eel = orc.calc_eel(M, full, tot, model["gld"])
    N1 = len(rp) - 1
    init, pseudoC, totc = rf.gibbs_setup(fx, M, N0, N1)
    return dict(fx=fx, M=M, N0=N0, N1=N1, rp=rp, sid=sid, val=val, eel=eel, mw=model["mw"], grp=grp,
                meta=rf.read_meta(fx), totc=totc, init=init, pseudoC=pseudoC)


def _ctx(d, alpha=None):
    return capi().GibbsContext(d["M"], d["rp"], d["sid"], d["val"], d["init"], alpha, d["pseudoC"],
                               d["totc"], d["N0"], d["eel"], d["mw"], d["grp"])


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_exact_chain_bit_identical_to_reference(name):
    d = _load(name)
    burnin, nsamples, gap = d["meta"]["gibbs"]
    T = d["meta"]["gibbs_threads"]
    seeds = capi().gibbs_chain_seeds(d["meta"]["gibbs_seed"], T)
    ctx = _ctx(d)
    tot = None
    for k in range(T):
        ns = nsamples // T + (1 if k < nsamples % T else 0)
        cv, acc, _ = ctx.run(capi().GIBBS_EXACT, seeds[k], burnin, ns, gap)
        gold = rf.read_countvectors(os.path.join(d["fx"], "temp", "s.countvectors%d" % k))
        assert np.array_equal(cv, gold)
        ocv, oacc = orc.gibbs_chain(d["M"], d["rp"], d["sid"], d["val"], d["init"], None, d["pseudoC"],
                                    d["totc"], d["N0"], d["eel"], d["mw"], d["grp"], seeds[k], burnin, ns, gap)
        for a, b in zip(acc, oacc):
            assert np.allclose(a, b, rtol=1e-10, atol=1e-9)
        tot = acc if tot is None else [x + y for x, y in zip(tot, acc)]
    pme_c = tot[0] / nsamples
    res = rf.per_target_rows(d["fx"])
    assert np.allclose(pme_c[1:], res["pme_c"], atol=0.00501)
    assert np.allclose((tot[2] / nsamples)[1:], res["pme_tpm"], atol=0.00501)
    ctx.close()


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_all_chains_in_one_launch_match_reference_files(name):
    """rsem_gibbs_run_chains: the reference's -p T chains advance together (one wave each); every chain's count vectors
    equal the reference's imd.countvectors<k> bit for bit and the summed accumulators equal the per-chain runs'."""
    d = _load(name)
    burnin, nsamples, gap = d["meta"]["gibbs"]
    T = d["meta"]["gibbs_threads"]
    seeds = capi().gibbs_chain_seeds(d["meta"]["gibbs_seed"], T)
    ns = [nsamples // T + (1 if k < nsamples % T else 0) for k in range(T)]
    ctx = _ctx(d)
    cvs, acc, _, prof = ctx.run_chains(capi().GIBBS_EXACT, seeds, burnin, ns, gap)
    assert prof.chains == T
    tot = None
    for k in range(T):
        gold = rf.read_countvectors(os.path.join(d["fx"], "temp", "s.countvectors%d" % k))
        assert np.array_equal(cvs[k], gold)
        _, oacc = orc.gibbs_chain(d["M"], d["rp"], d["sid"], d["val"], d["init"], None, d["pseudoC"],
                                  d["totc"], d["N0"], d["eel"], d["mw"], d["grp"], seeds[k], burnin, ns[k], gap)
        tot = oacc if tot is None else [x + y for x, y in zip(tot, oacc)]
    for a, b in zip(acc, tot):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-9)
    ctx.close()


def _synthetic_items(n_reads, config="small", long_row_every=0):
    from tools.synth_data import make_em_workload, to_gibbs_items
    wl = make_em_workload(config, seed=5, long_row_every=long_row_every)
    rp = wl["row_ptr"][:n_reads + 1]
    nz = int(rp[-1])
    sub = dict(wl, row_ptr=rp, sid=wl["sid"][:nz], conprb=wl["conprb"][:nz], ncp=wl["ncp"][:n_reads])
    return wl["M"], to_gibbs_items(sub)


@pytest.mark.parametrize("n_reads,long_every", [(200_000, 0), (30_000, 4000)])
def test_exact_tile_kernel_at_scale_vs_oracle(n_reads, long_every):
    """200 k reads x 5000 transcripts, 5 concurrent chains with unequal lengths: the team kernel's count vectors are the
    oracle's sequential chain's, chain by chain; the second case carries reads with more items than an LDS tile (walked
    alone)."""
    M, (irp, isid, icp) = _synthetic_items(n_reads, long_row_every=long_every)
    init = np.zeros(M + 1, np.int32)
    N0, pseudoC = 12345, 1.0
    eel, mw = np.full(M + 1, 700.0), np.ones(M + 1)
    grp = np.arange(1, M + 2, 10, dtype=np.int32)
    if grp[-1] != M + 1:
        grp = np.append(grp, M + 1).astype(np.int32)
    totc = (M + 1) * pseudoC + N0 + n_reads
    seeds = capi().gibbs_chain_seeds(99, 5)
    ns = [3, 3, 2, 2, 2]
    burnin, gap = 3, 2
    ctx = capi().GibbsContext(M, irp, isid, icp, init, None, pseudoC, totc, N0, eel, mw, grp)
    cvs, acc, _, prof = ctx.run_chains(capi().GIBBS_EXACT, seeds, burnin, ns, gap)
    for k in range(5):
        ocv, _ = orc.gibbs_chain(M, irp, isid, icp, init, None, pseudoC, totc, N0, eel, mw, grp, seeds[k], burnin, ns[k], gap)
        assert np.array_equal(cvs[k], ocv), k
    assert np.all(cvs[0].sum(1) == N0 + n_reads)
    print("exact sweeps: %.3f ms per round (5 chains, %d reads, %d workgroups per chain)" % (prof.sweep_ms, n_reads, prof.team))
    ctx.close()


@pytest.mark.parametrize("n_reads,long_every,chains,prior", [(150_000, 0, 8, False), (30_000, 4000, 3, False), (60_000, 0, 1, False),
                                                             (120_000, 0, 8, True), (30_000, 3500, 2, True)])
def test_exact_chain_is_the_same_for_every_team_size(n_reads, long_every, chains, prior, monkeypatch):
    """k_gibbs_exact_team (gibbs_exact_team.hpp): W workgroups per chain take W tiles of a window at once and settle the moves
    between them through the team's tables.  Whatever W is -- one workgroup per chain (the kernel of rounds 3-4), 2, 7 (no
    divisor of anything), 16 / 17 (the boundary of the group cells), what the device offers for this many chains -- the count
    vectors are the same integers, and they are the oracle's sequential chain's.  prior: per-transcript pseudo counts (--prior,
    Gibbs.cpp:171-194,300-303) -- the second pass of the kernel's headers (tiles of 3072 items)."""
    M, (irp, isid, icp) = _synthetic_items(n_reads, long_row_every=long_every)
    init = np.zeros(M + 1, np.int32)
    N0, pseudoC = 777, 1.0
    alpha = np.concatenate([[1.0], np.random.default_rng(8).uniform(0.05, 3.0, M)]) if prior else None
    eel, mw = np.full(M + 1, 700.0), np.ones(M + 1)
    grp = np.array([1, M + 1], np.int32)
    totc = (M + 1) * pseudoC + N0 + n_reads
    seeds = capi().gibbs_chain_seeds(4242, chains)
    ns = [2 + (k % 2) for k in range(chains)]
    burnin, gap = 2, 2
    ctx = capi().GibbsContext(M, irp, isid, icp, init, alpha, pseudoC, totc, N0, eel, mw, grp)
    monkeypatch.setenv("RSEM_GX_TEAM", "1")
    base, acc1, _, p1 = ctx.run_chains(capi().GIBBS_EXACT, seeds, burnin, ns, gap)
    ocv, _ = orc.gibbs_chain(M, irp, isid, icp, init, alpha, pseudoC, totc, N0, eel, mw, grp, seeds[0], burnin, ns[0], gap)
    assert np.array_equal(base[0], ocv)
    if prior and chains > 1:  # (--prior: a second chain against the oracle as well)
        ocv, _ = orc.gibbs_chain(M, irp, isid, icp, init, alpha, pseudoC, totc, N0, eel, mw, grp, seeds[chains - 1], burnin, ns[chains - 1], gap)
        assert np.array_equal(base[chains - 1], ocv)
    times = {1: p1.sweep_ms}
    for W in (2, 7, 16, 17, 0):
        if W:
            monkeypatch.setenv("RSEM_GX_TEAM", str(W))
        else:
            monkeypatch.delenv("RSEM_GX_TEAM")  # the product's choice: compute units / chains, at most 64
        cvs, acc, _, p = ctx.run_chains(capi().GIBBS_EXACT, seeds, burnin, ns, gap)
        for k in range(chains):
            assert np.array_equal(cvs[k], base[k]), (W, k)
        for a, b in zip(acc, acc1):
            assert np.array_equal(a, b)
        times[W] = p.sweep_ms
    print("exact sweeps by team size (0 = the product's choice), ms per round, %d chains x %d reads: %s" % (chains, n_reads, times))
    ctx.close()


def test_a_team_that_cannot_be_resident_gives_up_and_the_run_starts_over(monkeypatch, capfd):
    """The team barrier's way out (gibbs_exact_team.hpp: kXSpinLimit, XTeamCtl::abort).  A team needs all its workgroups on the GPU at
    once; the product guarantees that with a cooperative launch and a per-GPU lease, but another tenant's kernels can still hold the
    compute units.  Provoked for real here: 8 chains x 64 workgroups = 512 workgroups of 154 KB LDS each, launched NON-cooperatively
    (RSEM_GX_TEST_OVERSUBSCRIBE) on 256 compute units -- the resident half waits for the half that cannot start -- with the
    30-second limit lowered to 0.2 s (RSEM_GX_SPIN_LIMIT, ticks of 10 ns).  The run must (a) not hang, (b) start over with one
    workgroup per chain and return the SAME integers (the chain does not depend on the team size), telling the user on stderr;
    (c) with the fallback switched off (RSEM_GX_NO_FALLBACK) fail through the C ABI with a message that names the cause."""
    n_reads, chains = 150_000, 8
    M, (irp, isid, icp) = _synthetic_items(n_reads)
    init = np.zeros(M + 1, np.int32)
    N0, pseudoC = 777, 1.0
    eel, mw = np.full(M + 1, 700.0), np.ones(M + 1)
    grp = np.array([1, M + 1], np.int32)
    totc = (M + 1) * pseudoC + N0 + n_reads
    seeds = capi().gibbs_chain_seeds(31, chains)
    ns = [2] * chains
    ctx = capi().GibbsContext(M, irp, isid, icp, init, None, pseudoC, totc, N0, eel, mw, grp)
    base, acc0, _, p0 = ctx.run_chains(capi().GIBBS_EXACT, seeds, 2, ns, 1)
    assert p0.team > 1
    monkeypatch.setenv("RSEM_GX_TEAM", "64")
    monkeypatch.setenv("RSEM_GX_TEST_OVERSUBSCRIBE", "1")
    monkeypatch.setenv("RSEM_GX_SPIN_LIMIT", "20000000")
    capfd.readouterr()
    import time
    t0 = time.perf_counter()
    cvs, acc, _, p = ctx.run_chains(capi().GIBBS_EXACT, seeds, 2, ns, 1)
    took = time.perf_counter() - t0
    err = capfd.readouterr().err
    assert "gave up waiting at a team barrier" in err and "starting the run over with one workgroup per chain" in err
    assert p.team == 1 and took < 25.0
    for k in range(chains):
        assert np.array_equal(cvs[k], base[k]), k
    for a, b in zip(acc, acc0):
        assert np.array_equal(a, b)
    monkeypatch.setenv("RSEM_GX_NO_FALLBACK", "1")
    with pytest.raises(capi().RsemHipError) as ei:
        ctx.run_chains(capi().GIBBS_EXACT, seeds, 2, ns, 1)
    assert "gave up waiting at a team barrier" in str(ei.value) and "another program" in str(ei.value)
    # and the context is usable afterwards: the product's own launch again
    for v in ("RSEM_GX_TEAM", "RSEM_GX_TEST_OVERSUBSCRIBE", "RSEM_GX_SPIN_LIMIT", "RSEM_GX_NO_FALLBACK"):
        monkeypatch.delenv(v)
    again, _, _, p2 = ctx.run_chains(capi().GIBBS_EXACT, seeds, 2, ns, 1)
    assert p2.team == p0.team and all(np.array_equal(again[k], base[k]) for k in range(chains))
    ctx.close()


def test_chain_groups_reduce_over_local_comm():
    """Chains dealt to two groups (here: both on GPU 0, the LOCAL communicator; on a multi-GPU node the same calls run
    over RCCL): group sums meet in one reduce on rank 0 and equal the single-group run; count vectors stay with the
    group that ran the chain (file ownership: Gibbs.cpp:225-226)."""
    import threading
    d = _load("pe_q")
    seeds = capi().gibbs_chain_seeds(17, 4)
    ns = [5, 5, 4, 4]
    ctx = _ctx(d)
    cvs, acc, _, _ = ctx.run_chains(capi().GIBBS_EXACT, seeds, 4, ns, 1)
    ctx.close()
    comms = capi().Comm.create_local([0, 0])
    assert [c.rank for c in comms] == [0, 1] and comms[0].world == 2
    out = [None, None]

    def group(w):
        g = _ctx(d)
        g.set_comm(comms[w])
        mine = list(range(w, 4, 2))
        out[w] = g.run_chains(capi().GIBBS_EXACT, seeds[mine], 4, [ns[k] for k in mine], 1)
        g.close()

    ts = [threading.Thread(target=group, args=(w,)) for w in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in comms:
        c.close()
    assert out[0] is not None and out[1] is not None
    for a, b in zip(out[0][1], acc):  # rank 0 holds the totals
        assert np.allclose(a, b, rtol=1e-12, atol=1e-9)
    assert np.array_equal(out[0][0][0], cvs[0]) and np.array_equal(out[0][0][1], cvs[2])
    assert np.array_equal(out[1][0][0], cvs[1]) and np.array_equal(out[1][0][1], cvs[3])


def test_parallel_sampler_invariants_and_determinism():
    d = _load("se_q")
    ctx = _ctx(d)
    cv1, acc1, _ = ctx.run(capi().GIBBS_PARALLEL, 7, 10, 20, 1, thin=2)
    cv2, acc2, _ = ctx.run(capi().GIBBS_PARALLEL, 7, 10, 20, 1, thin=2)
    cv3, _, _ = ctx.run(capi().GIBBS_PARALLEL, 8, 10, 20, 1, thin=2)
    assert np.array_equal(cv1, cv2)              # fixed seed -> identical chain
    assert not np.array_equal(cv1, cv3)
    assert np.all(cv1.sum(1) == d["N0"] + d["N1"])  # every read assigned exactly once (+N0 in bin 0)
    assert cv1.min() >= 0
    ctx.close()


def test_parallel_sampler_reads_with_more_than_256_alignments():
    """The data-augmentation sweep walks reads that the sliced layout does not take (> 256 alignments) with a wave per read
    (k_sample_z_long: chunk scans instead of one thread's serial walk).  Here 40 such reads hit transcripts that no other read
    touches: every sample must put exactly those 40 reads on exactly those transcripts (their noise weight is negligible), every
    read is assigned once, the chain is a function of its seed, and the long reads' picks follow their weights -- half of a long
    read's weight sits on its first 10 alignments."""
    rng = np.random.default_rng(5)
    M, n_short, n_long, L = 3000, 20000, 40, 700
    lens = np.concatenate([rng.integers(1, 9, n_short), np.full(n_long, L)]) + 1   # + the noise item
    order = rng.permutation(n_short + n_long)
    is_long = (np.arange(n_short + n_long) >= n_short)[order]
    lens = lens[order]
    rp = np.zeros(len(lens) + 1, np.uint64)
    rp[1:] = np.cumsum(lens)
    sid = np.zeros(int(rp[-1]), np.int32)
    cp = np.zeros(int(rp[-1]))
    for i in range(len(lens)):
        a, b = int(rp[i]), int(rp[i + 1])
        cp[a] = 1e-12                                                  # noise
        if is_long[i]:
            sid[a + 1:b] = 2001 + np.sort(rng.choice(999, L, replace=False))   # ids 2001..2999: the long reads' own
            w = np.full(L, 0.5 / (L - 10))
            w[:10] = 0.05
            cp[a + 1:b] = w
        else:
            k = b - a - 1
            sid[a + 1:b] = 1 + np.sort(rng.choice(2000, k, replace=False))
            cp[a + 1:b] = 10.0 ** rng.uniform(-2, 0, k)
    N1 = len(lens)
    init = np.zeros(M + 1, np.int32)
    eel, mw, grp = np.full(M + 1, 500.0), np.ones(M + 1), np.array([1, M + 1], np.int32)
    ctx = capi().GibbsContext(M, rp, sid, cp, init, None, 1.0, (M + 1) + N1, 0, eel, mw, grp)
    cv1, _, _ = ctx.run(capi().GIBBS_PARALLEL, 11, 3, 30, 1, thin=1)
    cv2, _, _ = ctx.run(capi().GIBBS_PARALLEL, 11, 3, 30, 1, thin=1)
    ctx.close()
    assert np.array_equal(cv1, cv2)
    assert np.all(cv1.sum(1) == N1) and cv1.min() >= 0
    assert np.all(cv1[:, 2001:].sum(1) == n_long)                     # the long reads, and nobody else, on their own transcripts
    # (with the sampler's theta in play the first ten alignments keep about half of a read's weight: counts + 1 are nearly flat here)
    first_ids = np.zeros(M + 1, bool)
    for i in np.nonzero(is_long)[0]:
        first_ids[sid[int(rp[i]) + 1:int(rp[i]) + 11]] = True
    share = cv1[:, first_ids].sum() / (30.0 * n_long)
    assert 0.2 < share < 0.9, share


@pytest.mark.parametrize("name", ["se_q", "pe_q", "se_q_polya_rspd", "se_noq_rev_rspd_omit"])
def test_parallel_posterior_means_within_sampling_tolerance(name):
    """z-test of PARALLEL posterior mean counts against a long oracle (reference-equivalent) chain."""
    d = _load(name)
    M = d["M"]
    n_o = 4000
    ocv, oacc = orc.gibbs_chain(M, d["rp"], d["sid"], d["val"], d["init"], None, d["pseudoC"], d["totc"], d["N0"],
                                d["eel"], d["mw"], d["grp"], 99, 200, n_o, 1)
    ctx = _ctx(d)
    n_g = 4000
    cv, acc, _ = ctx.run(capi().GIBBS_PARALLEL, 1234, 200, n_g, 1, thin=4)
    mo, mg = oacc[0] / n_o, acc[0] / n_g
    vo = np.maximum(oacc[1] / n_o - mo * mo, 0)
    vg = np.maximum(acc[1] / n_g - mg * mg, 0)
    # posterior sd agrees (same target distribution) ...
    big = vo > 1.0
    assert np.all(np.abs(np.sqrt(vg[big]) / np.sqrt(vo[big]) - 1.0) < 0.35)
    # ... and means agree within Monte-Carlo error; autocorrelation is allowed for with an
    # effective sample size of n/50 for both chains
    se = np.sqrt(vo / (n_o / 50.0) + vg / (n_g / 50.0)) + 0.05
    z = np.abs(mo - mg) / se
    assert z.max() < 6.0, (z.max(), int(z.argmax()), mo[z.argmax()], mg[z.argmax()])
    # TPM means
    to, tg = oacc[2] / n_o, acc[2] / n_g
    assert np.corrcoef(to[1:], tg[1:])[0, 1] > 0.9995
    ctx.close()


def test_parallel_sampler_is_no_further_from_long_chains_than_the_reference_setup():
    """200 k reads x 5 k transcripts (tests/golden/make_gibbs_truth.py).  "Truth" = long collapsed chains (8 x (2000 + 500)
    sweeps of the oracle's restatement of Gibbs.cpp, bit-identical to the reference).  The reference's own configuration
    (BURNIN 200, 1000 samples over 64 chains) sits at a certain distance from them; the data-augmentation sampler with the
    program's default of 8 sweeps per round, same BURNIN / NSAMPLES / chains, must not sit further away.  And the exact
    mode, given the reference configuration's seeds, must reproduce its posterior mean counts EXACTLY (64 concurrent
    chains x 216 rounds x 200 k reads of the tile kernel against the sequential chain)."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_gibbs_truth", os.path.join(here, "golden", "make_gibbs_truth.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    T = np.load(os.path.join(here, "golden", "gibbs_truth", "truth.npz"))
    assert int(T["n_reads"]) == mk.N_READS and int(T["seed"]) == mk.SEED
    d = mk.items()
    ctx = capi().GibbsContext(d["M"], d["irp"], d["isid"], d["icp"], d["init"], None, d["pseudoC"], d["totc"], d["N0"], d["eel"], d["mw"], d["grp"])
    ns = [1000 // 64 + (1 if k < 1000 % 64 else 0) for k in range(64)]

    def dist(x):
        q = np.abs(x - T["long_mean"]) / (T["long_sd"] + 0.5)
        return float(np.sqrt((q ** 2).mean())), float(q.max())

    _, acc, _, prof = ctx.run_chains(capi().GIBBS_EXACT, orc.chain_seeds(5, 64), 200, ns, 1, want_vectors=False)
    assert np.array_equal(acc[0] / 1000.0, T["ref_mean"])  # the reference configuration, chain for chain
    _, accp, _, _ = ctx.run_chains(capi().GIBBS_PARALLEL, capi().gibbs_chain_seeds(5, 64), 200, ns, 1, thin=8, want_vectors=False)
    ctx.close()
    rms_ref, max_ref = dist(T["ref_mean"])
    rms_l2, max_l2 = dist(T["long2_mean"])
    rms_p, max_p = dist(accp[0] / 1000.0)
    print("distance to the long chains, |diff| / (sd + 0.5): second long set rms %.4f max %.3f | reference setup rms %.4f max %.3f | "
          "parallel sampler rms %.4f max %.3f | exact mode %.1f ms per round (64 chains)" % (rms_l2, max_l2, rms_ref, max_ref, rms_p, max_p, prof.sweep_ms))
    assert rms_p <= 1.10 * rms_ref and max_p <= 1.5 * max(max_ref, max_l2)
    assert abs(accp[0].sum() / 1000.0 - (d["N0"] + mk.N_READS)) < 1e-6


def test_exact_chain_with_prior_file_semantics():
    """--prior: per-transcript pseudo counts (Gibbs.cpp:171-194, 300-303): same integer draws as the oracle."""
    d = _load("pe_q")
    M = d["M"]
    rng = np.random.default_rng(3)
    alpha = np.concatenate([[0.0], rng.uniform(0.05, 3.0, M)])
    totc = 1 + alpha[1:].sum() + d["N0"] + d["N1"]
    ctx = capi().GibbsContext(M, d["rp"], d["sid"], d["val"], d["init"], alpha, 1.0, totc, d["N0"], d["eel"], d["mw"], d["grp"])
    cv, acc, _ = ctx.run(capi().GIBBS_EXACT, 777, 5, 6, 2)
    ocv, oacc = orc.gibbs_chain(M, d["rp"], d["sid"], d["val"], d["init"], alpha, 1.0, totc, d["N0"], d["eel"], d["mw"],
                                d["grp"], 777, 5, 6, 2)
    assert np.array_equal(cv, ocv)
    for a, b in zip(acc, oacc):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-9)
    ctx.close()
