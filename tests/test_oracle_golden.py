"""Pin oracle/rsem_oracle.c against outputs of the unmodified reference (tests/golden, CPU only).

The reference has no tests of its own; these fixtures were produced by running its
binaries (tests/golden/make_fixtures.py), so they are the known-answer vectors for the oracle.
"""
import os

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc


def _load(name):
    fx = rf.fixture(name)
    M, N0, rp_items, sid_items, val_items = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    raw, pol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    model = rf.read_model(os.path.join(fx, "stat", "s.model"))
    full, tot = rf.read_seq_lens(os.path.join(fx, "ref.seq"))
    grp = rf.read_grp(os.path.join(fx, "ref.grp"))
    return dict(fx=fx, M=M, N0=N0, items=(rp_items, sid_items, val_items), raw=raw, pol=pol, model=model,
                full=full, tot=tot, grp=grp, meta=rf.read_meta(fx))


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_estep_known_answer(name):
    """theta' (.theta line 2) + frozen conprb (.ofg) -> expected_count row of iso_res (EM.cpp:460-478)."""
    d = _load(name)
    rp, sid, cp, ncp = rf.split_noise(*d["items"])
    counts = orc.em_estep(d["M"], rp, sid, cp, ncp, d["raw"])
    expected = rf.per_target_rows(d["fx"], em_only=True)["count"]
    # rows dropped from .ofg (all entries < 1e-300) contribute nothing, exactly as in the reference
    assert np.allclose(counts[1:], expected, atol=0.00501)
    N0, N1, N2, Ntot = rf.read_cnt(os.path.join(d["fx"], "stat", "s.cnt"))
    c2, theta_new, s, b, t = orc.em_mstep(d["M"], N0, counts, d["raw"])
    # converged: the reference stopped because no theta >= 1e-7 moved by >= 1e-3
    assert t == 0 and b < 1e-3
    assert abs(s - (N0 + (len(rp) - 1))) < 1e-6


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_polish_and_expression(name):
    d = _load(name)
    eel = orc.calc_eel(d["M"], d["full"], d["tot"], d["model"]["gld"])
    res = rf.per_target_rows(d["fx"], em_only=True)
    assert np.allclose(eel[1:], res["eel"], atol=0.00501)
    pol = orc.polish_theta(d["M"], d["raw"], eel, d["model"]["mw"])
    assert np.allclose(pol, d["pol"], rtol=1e-9, atol=1e-300)
    tpm, fpkm = orc.calc_expression(d["M"], pol, eel)
    assert np.allclose(tpm[1:], res["tpm"], atol=0.00501)
    assert np.allclose(fpkm[1:], res["fpkm"], atol=0.00501)


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_gibbs_chain_bit_exact(name):
    """Same seeds -> the integer count vectors of the reference's chains, bit for bit (Gibbs.cpp:265-353)."""
    d = _load(name)
    rp, sid, val = d["items"]
    M, N0 = d["M"], d["N0"]
    N1 = len(rp) - 1
    burnin, nsamples, gap = d["meta"]["gibbs"]
    T = d["meta"]["gibbs_threads"]
    seeds = orc.chain_seeds(d["meta"]["gibbs_seed"], T)
    eel = orc.calc_eel(M, d["full"], d["tot"], d["model"]["gld"])
    mw = d["model"]["mw"]
    init_counts, pseudoC, totc = rf.gibbs_setup(d["fx"], M, N0, N1)
    m = len(d["grp"]) - 1
    tot = [np.zeros(M + 1) for _ in range(4)] + [np.zeros(m)]
    for k in range(T):
        ns = nsamples // T + (1 if k < nsamples % T else 0)
        cv, acc = orc.gibbs_chain(M, rp, sid, val, init_counts, None, pseudoC, totc, N0, eel, mw, d["grp"], seeds[k],
                                  burnin, ns, gap)
        gold = rf.read_countvectors(os.path.join(d["fx"], "temp", "s.countvectors%d" % k))
        assert np.array_equal(cv, gold)
        for a, b in zip(tot, acc):
            a += b
    pme_c = tot[0] / nsamples
    pve_c = np.maximum((tot[1] - nsamples * pme_c * pme_c) / (nsamples - 1), 0)
    res = rf.per_target_rows(d["fx"])
    assert np.allclose(pme_c[1:], res["pme_c"], atol=0.00501)
    assert np.allclose(np.sqrt(pve_c[1:]), res["sd"], atol=0.00501)
    assert np.allclose((tot[2] / nsamples)[1:], res["pme_tpm"], atol=0.00501)
    assert np.allclose((tot[3] / nsamples)[1:], res["pme_fpkm"], atol=0.00501)


def test_em_run_invariants():
    """Sum of counts = N0 + N1 each round; theta sums to 1 (SURVEY.md section 4 invariants)."""
    d = _load("se_q")
    rp, sid, cp, ncp = rf.split_noise(*d["items"])
    M = d["M"]
    N0, N1, N2, Ntot = rf.read_cnt(os.path.join(d["fx"], "stat", "s.cnt"))
    theta0 = np.full(M + 1, (1.0 - max(N0 / (Ntot - N2), 1e-8)) / M)
    theta0[0] = max(N0 / (Ntot - N2), 1e-8)
    theta, rounds, b, t = orc.em_run(M, rp, sid, cp, ncp, N0, theta0)
    assert rounds >= 20 and t == 0
    assert abs(theta.sum() - 1.0) < 1e-12
