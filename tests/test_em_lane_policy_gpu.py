"""Lanes per read that are not a power of two (rsem_em_set_option "lane_policy" 1; sell_layout.hpp, RSEM_GENERAL_G): a
prepared variant of the layout.  The product build has it compiled out and these tests skip; they run against a variant
library (tools/build_variants.sh g1 "-DRSEM_GENERAL_G=1"; RSEM_HIP_LIB=rsem_amd/librsem_hip_g1.so).  Results must not depend on
the policy: the same oracle comparisons as test_em_gpu.py, with fewer plane bytes."""
import os

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc
from tools.q32_ref import quantize_q32
from tools.synth_data import make_em_workload

pytestmark = pytest.mark.gpu


def _ctx(M, rp, sid, cp, ncp):
    from rsem_amd import capi
    ctx = capi.EmContext(M, rp, sid, cp, ncp)
    if ctx.info("general_g") != 1:
        ctx.close()
        pytest.skip("library built without RSEM_GENERAL_G")
    return ctx


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_lane_policy_step_and_run_on_fixtures(name):
    fx = rf.fixture(name)
    M, N0ofg, rpi, sidi, vali = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    rp, sid, cp, ncp = rf.split_noise(rpi, sidi, vali)
    raw, pol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    N0, N1, N2, Ntot = rf.read_cnt(os.path.join(fx, "stat", "s.cnt"))
    ctx = _ctx(M, rp, sid, cp, ncp)
    ctx.set_option("lane_policy", 1)
    counts, theta_new, s, b, t = ctx.step(raw, N0)
    oc = orc.em_estep(M, rp, sid, cp, ncp, raw)
    oc, oth, os_, ob, ot = orc.em_mstep(M, N0, oc, raw)
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-12) and np.allclose(theta_new, oth, rtol=1e-9, atol=1e-15) and t == ot
    th0 = max(N0 * 1.0 / (Ntot - N2), 1e-8)
    theta0 = np.full(M + 1, (1.0 - th0) / M)
    theta0[0] = th0
    out = ctx.run(theta0, N0)
    oth, orounds, ob, ot = orc.em_run(M, rp, sid, cp, ncp, N0, theta0)
    assert out["rounds"] == orounds and out["totNum"] == ot and np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12)
    ctx.close()


def test_lane_policy_every_row_length_every_loop_and_q32(monkeypatch):
    """Row lengths 1..300 (every shape incl. the CSR tail), random transcripts; then gene-structured data with both value
    formats and all loops."""
    rng = np.random.default_rng(5)
    M = 3000
    lens = np.concatenate([np.arange(1, 301), rng.integers(1, 70, 20000)]).astype(np.int64)
    rng.shuffle(lens)
    rp = np.zeros(len(lens) + 1, np.uint64)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    start = rng.integers(1, M - 300, len(lens))
    sid = (np.repeat(start, lens) + (np.arange(nnz) - np.repeat(rp[:-1].astype(np.int64), lens))).astype(np.int32)
    cp = np.power(10.0, rng.uniform(-30, -3, nnz))
    ncp = np.power(10.0, rng.uniform(-60, -30, len(lens)))
    theta = rng.random(M + 1)
    theta /= theta.sum()
    oc = orc.em_estep(M, rp, sid, cp, ncp, theta)
    oc[0] += 10.0
    ctx = _ctx(M, rp, sid, cp, ncp)
    b0 = ctx.info("value_plane_bytes")
    c0, *_ = ctx.step(theta, 10.0)
    ctx.set_option("lane_policy", 1)
    b1 = ctx.info("value_plane_bytes")
    c1, *_ = ctx.step(theta, 10.0)
    assert b1 < 0.97 * b0  # (0.94 by the policy table for this mix of lengths)
    assert np.allclose(c0, oc, rtol=1e-9, atol=1e-12) and np.allclose(c1, oc, rtol=1e-9, atol=1e-12)
    ctx.set_option("value_bits", 32)
    q, ok = quantize_q32(rp, cp, 8)
    assert ctx.info("reads_q32") == int(ok.sum())
    oq = orc.em_estep(M, rp, sid, q, ncp, theta)
    oq[0] += 10.0
    c2, *_ = ctx.step(theta, 10.0)
    assert np.allclose(c2, oq, rtol=1e-9, atol=1e-12)
    ctx.set_option("kernel", 2)  # the cross-check kernel needs power-of-two groups (and doubles)
    from rsem_amd import capi
    with pytest.raises(capi.RsemHipError):
        ctx.step(theta, 10.0)
    ctx.close()
    wl = make_em_workload("small", seed=41)
    oth, orounds, _, ot = orc.em_run(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], max_round=3000)
    for bits in (64, 32):
        ctx = _ctx(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
        ctx.set_option("lane_policy", 1)
        ctx.set_option("value_bits", bits)
        for mode in ("0", "1", "2"):
            monkeypatch.setenv("RSEM_EM_FUSED", mode)
            out = ctx.run(wl["theta0"], wl["N0"], max_round=3000)
            big = oth >= 1e-7
            assert out["rounds"] == orounds and out["totNum"] == ot, (bits, mode)
            assert np.max(np.abs(out["theta"][big] - oth[big]) / oth[big]) < 1e-6, (bits, mode)
        ctx.close()
