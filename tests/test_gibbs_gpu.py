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
