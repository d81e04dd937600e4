"""Q32 value planes of the theta-only E step (rsem_em_set_option "value_bits" 32; sell_layout.hpp).

Two claims, tested separately:
 1. EXACTNESS of the format: a mantissa times 2^e is exact in a double, so the Q32 kernel computes what the F64 kernel
    computes on the rounded values -- compared with the oracle run on tools/q32_ref.quantize_q32(values) at the
    per-step tolerance of the F64 tests (1e-9), and with the F64 kernel fed those rounded values (1e-11: only the
    order of the floating-point atomics differs).  Which reads were compressed must agree with the numpy rule exactly.
 2. ACCURACY of the rounding: a whole run on the ORIGINAL values stops at the oracle's ROUND and lands within the
    north-star bar (theta 1e-6 relative) of the oracle's theta on the original doubles.
"""
import os

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc
from tools.q32_ref import quantize_q32
from tools.synth_data import make_em_workload

pytestmark = pytest.mark.gpu


def capi():
    from rsem_amd import capi as c
    return c


def _fixture_csr(name):
    fx = rf.fixture(name)
    M, N0ofg, rpi, sidi, vali = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    rp, sid, cp, ncp = rf.split_noise(rpi, sidi, vali)
    raw, pol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    N0, N1, N2, Ntot = rf.read_cnt(os.path.join(fx, "stat", "s.cnt"))
    return dict(M=M, N0=N0, Ntot=Ntot, N2=N2, rp=rp, sid=sid, cp=cp, ncp=ncp, raw=raw)


def _q32_ctx(M, rp, sid, cp, ncp, range_bits=None):
    ctx = capi().EmContext(M, rp, sid, cp, ncp)
    if range_bits is not None:
        ctx.set_option("value_range_bits", range_bits)
    ctx.set_option("value_bits", 32)
    return ctx


def _check_step(M, rp, sid, cp, ncp, theta, N0, range_bits, atol=1e-12):
    q, ok = quantize_q32(rp, cp, range_bits)
    ctx = _q32_ctx(M, rp, sid, cp, ncp, range_bits)
    assert ctx.info("value_bits") == 32 and ctx.info("reads_q32") == int(ok.sum())
    counts, theta_new, s, b, t = ctx.step(theta, N0)
    ctx.close()
    oc = orc.em_estep(M, rp, sid, q, ncp, theta)
    oc, oth, os_, ob, ot = orc.em_mstep(M, N0, oc, theta)
    assert np.allclose(counts, oc, rtol=1e-9, atol=atol)
    assert np.allclose(theta_new, oth, rtol=1e-9, atol=1e-15) and t == ot
    # the F64 kernel on the rounded values: the same arithmetic
    ref = capi().EmContext(M, rp, sid, q, ncp)
    c64, th64, *_ = ref.step(theta, N0)
    ref.close()
    assert np.allclose(counts, c64, rtol=1e-11, atol=atol)  # two layouts: only the summation order differs
    return ok


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_q32_step_is_the_f64_step_on_rounded_values(name):
    d = _fixture_csr(name)
    ok = _check_step(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"], d["raw"], d["N0"], 8)
    assert ok.any()
    _check_step(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"], d["raw"], d["N0"], 24)


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_q32_full_run_same_rounds_theta_within_bar(name):
    d = _fixture_csr(name)
    M = d["M"]
    th0 = max(d["N0"] * 1.0 / (d["Ntot"] - d["N2"]), 1e-8)  # EM.cpp:343-346
    theta0 = np.full(M + 1, (1.0 - th0) / M)
    theta0[0] = th0
    ctx = _q32_ctx(M, d["rp"], d["sid"], d["cp"], d["ncp"])
    out = ctx.run(theta0, d["N0"])
    ctx.close()
    oth, orounds, ob, ot = orc.em_run(M, d["rp"], d["sid"], d["cp"], d["ncp"], d["N0"], theta0)  # the ORIGINAL doubles
    assert out["rounds"] == orounds and out["totNum"] == ot
    big = oth >= 1e-7
    assert np.max(np.abs(out["theta"][big] - oth[big]) / oth[big]) < 1e-6
    assert abs(out["theta"].sum() - 1.0) < 1e-12


def test_q32_mixed_formats_and_every_loop(monkeypatch):
    """Reads of both formats in one layout (a third of the reads get an alignment 2^-20 below their largest one), long
    rows in the CSR beside them, all three loops of rsem_em_run."""
    wl = make_em_workload("small", seed=31, long_row_every=50_000)
    M, rp, sid, ncp = wl["M"], wl["row_ptr"], wl["sid"], wl["ncp"]
    cp = wl["conprb"].copy()
    first = rp[:-1].astype(np.int64)
    lens = np.diff(rp.astype(np.int64))
    wide = (np.arange(len(lens)) % 3 == 0) & (lens >= 2)
    cp[first[wide]] *= 2.0 ** -20
    ok = _check_step(M, rp, sid, cp, ncp, wl["theta0"], wl["N0"], 8, atol=1e-9)
    assert 0.2 < ok.mean() < 0.8 and (lens > 256).any() and not ok[lens > 256].any()
    oth, orounds, _, ot = orc.em_run(M, rp, sid, cp, ncp, wl["N0"], wl["theta0"], max_round=400)
    ctx = _q32_ctx(M, rp, sid, cp, ncp)
    assert ctx.info("reads_long") == int((lens > 256).sum())
    res = {}
    for mode in ("0", "1", "2"):  # plain / fused / solo (long rows force plain: the request is then ignored)
        monkeypatch.setenv("RSEM_EM_FUSED", mode)
        res[mode] = ctx.run(wl["theta0"], wl["N0"], max_round=400)
    ctx.close()
    for mode, out in res.items():
        assert out["rounds"] == orounds and out["totNum"] == ot, mode
        big = oth >= 1e-7
        assert np.max(np.abs(out["theta"][big] - oth[big]) / oth[big]) < 1e-6, mode
    # without long rows the one-launch and the two-stream loops really run
    wl = make_em_workload("small", seed=32)
    M, rp, sid, cp, ncp = wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"]
    oth, orounds, _, ot = orc.em_run(M, rp, sid, cp, ncp, wl["N0"], wl["theta0"], max_round=3000)
    ctx = _q32_ctx(M, rp, sid, cp, ncp)
    assert ctx.info("reads_long") == 0 and ctx.info("reads_q32") > 0.9 * (len(rp) - 1)
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("RSEM_EM_FUSED", mode)
        out = ctx.run(wl["theta0"], wl["N0"], max_round=3000)
        assert out["rounds"] == orounds and out["totNum"] == ot, mode
        big = oth >= 1e-7
        assert np.max(np.abs(out["theta"][big] - oth[big]) / oth[big]) < 1e-6, mode
    ctx.close()


def test_q32_edge_cases():
    c = capi()
    M = 3
    # read 0: qualifies; read 1: single alignment (always qualifies); read 2: 1e-200 vs 1e-230 (range) stays F64;
    # read 3: contains an exact zero (zeros are exact in Q32); read 4: all zero -> F64; read 5: qualifies, and its
    # first term (2.5e-290 * theta 1e-15) falls under the 1e-300 clamp (EM.cpp:212) while the second does not;
    # read 6: values so small that 2^e would leave the exponent range the format admits -> F64
    rp = np.array([0, 2, 3, 5, 7, 8, 10, 12], np.uint64)
    sid = np.array([1, 2, 3, 1, 3, 2, 3, 1, 3, 2, 1, 2], np.int32)
    cp = np.array([1e-5, 2e-5, 3e-7, 1e-200, 1e-230, 0.0, 4e-9, 0.0, 2.5e-290, 1.5e-290, 1.5e-300, 2.5e-300], np.float64)
    ncp = np.array([1e-9, 0.0, 1e-250, 1e-12, 1e-30, 0.0, 0.0], np.float64)
    theta = np.array([0.1, 0.45, 0.45 - 1e-15, 1e-15])
    q, ok = quantize_q32(rp, cp, 8)
    assert list(ok) == [True, True, False, True, False, True, False]
    ctx = _q32_ctx(M, rp, sid, cp, ncp)
    assert ctx.info("reads_q32") == 4
    counts, th, s, b, t = ctx.step(theta, 2.0)
    oc = orc.em_estep(M, rp, sid, q, ncp, theta)
    oc[0] += 2.0
    assert np.allclose(counts, oc, rtol=1e-12, atol=0)
    # new values: the formats are chosen again from them
    cp2 = cp.copy()
    cp2[3], cp2[4] = 1e-200, 3e-200
    cp2[0] = 1e-9
    q2, ok2 = quantize_q32(rp, cp2, 8)
    assert list(ok2) == [False, True, True, True, False, True, False]
    ctx.set_values(cp2, ncp)
    assert ctx.info("reads_q32") == 4
    counts2, *_ = ctx.step(theta, 2.0)
    oc2 = orc.em_estep(M, rp, sid, q2, ncp, theta)
    oc2[0] += 2.0
    assert np.allclose(counts2, oc2, rtol=1e-12, atol=0)
    # back to doubles
    ctx.set_option("value_bits", 64)
    assert ctx.info("reads_q32") == 0
    counts3, *_ = ctx.step(theta, 2.0)
    oc3 = orc.em_estep(M, rp, sid, cp2, ncp, theta)
    oc3[0] += 2.0
    assert np.allclose(counts3, oc3, rtol=1e-12, atol=0)
    # the weights always come from the doubles
    ctx.set_option("value_bits", 32)
    _, w, wn = ctx.expected_weights(theta, 2.0)
    _, ow, own = orc.em_estep(M, rp, sid, cp2, ncp, theta, want_weights=True)
    assert np.allclose(w, ow, rtol=1e-12, atol=0) and np.allclose(wn, own, rtol=1e-12, atol=0)
    ctx.close()
    # format requested before the values exist
    ctx = c.EmContext(M, rp, sid)
    ctx.set_option("value_bits", 32)
    assert ctx.info("reads_q32") == 0
    ctx.set_values(cp, ncp)
    assert ctx.info("reads_q32") == 4
    counts4, *_ = ctx.step(theta, 2.0)
    assert np.allclose(counts4, oc, rtol=1e-12, atol=0)
    ctx.close()
    # empty shard
    ctx = c.EmContext(2, np.array([0], np.uint64), np.zeros(0, np.int32), np.zeros(0), np.zeros(0))
    ctx.set_option("value_bits", 32)
    counts, th, s, b, t = ctx.step(np.array([0.2, 0.4, 0.4]), 5.0)
    assert counts[0] == 5.0 and s == 5.0
    ctx.close()


def test_q32_sharded_equals_single_context():
    """Row shards (EM.cpp:135-157) over the LOCAL communicator with Q32 planes on every shard."""
    import threading
    c = capi()
    wl = make_em_workload("small", seed=33)
    M, rp, sid, cp, ncp = wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"]
    single = _q32_ctx(M, rp, sid, cp, ncp)
    ref = single.run(wl["theta0"], wl["N0"], max_round=300)
    single.close()
    from rsem_amd import dist as rd
    world = 2
    bounds = c.em_shard_rows(rp, world)
    comms = c.Comm.create_local([0] * world)
    outs = [None] * world

    def work(k):
        srp, ssid, scp, sncp = rd.take_shard(rp, sid, cp, ncp, bounds[k], bounds[k + 1])
        ctx = _q32_ctx(M, srp, np.ascontiguousarray(ssid), np.ascontiguousarray(scp), np.ascontiguousarray(sncp))
        ctx.set_comm(comms[k])
        outs[k] = ctx.run(wl["theta0"], wl["N0"], max_round=300)
        ctx.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for cm in comms:
        cm.close()
    for k in range(world):
        assert outs[k] is not None and outs[k]["rounds"] == ref["rounds"]
        assert np.allclose(outs[k]["theta"], ref["theta"], rtol=1e-9, atol=1e-15)
    assert np.array_equal(outs[0]["theta"], outs[1]["theta"])


def test_q32_full_size_c2_step_and_run():
    """BASELINE configs[1] at full size: the compressed step against the oracle on the rounded values; 30 rounds against
    the F64 layout of the same context."""
    wl = make_em_workload("C2")
    M, rp, sid, cp, ncp = wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"]
    q, ok = quantize_q32(rp, cp, 8)
    ctx = capi().EmContext(M, rp, sid, cp, ncp)
    f64 = ctx.run(wl["theta0"], wl["N0"], min_round=30, max_round=30)
    bytes64 = ctx.info("value_plane_bytes")
    ctx.set_option("value_bits", 32)
    assert ctx.info("reads_q32") == int(ok.sum()) and ok.mean() > 0.95
    assert ctx.info("value_plane_bytes") < 0.56 * bytes64
    counts, theta_new, s, b, t = ctx.step(wl["theta0"], wl["N0"])
    oc = orc.em_estep(M, rp, sid, q, ncp, wl["theta0"])
    oc[0] += wl["N0"]
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-7)
    out = ctx.run(wl["theta0"], wl["N0"], min_round=30, max_round=30)
    ctx.close()
    assert out["rounds"] == 30 and abs(out["theta"].sum() - 1.0) < 1e-12
    big = f64["theta"] >= 1e-7
    assert np.max(np.abs(out["theta"][big] - f64["theta"][big]) / f64["theta"][big]) < 1e-6


@pytest.mark.parametrize("extra", [[], ["--ngpus", "2", "--devices", "0,0"]], ids=["1gpu", "2shards"])
@pytest.mark.parametrize("name", rf.FIXTURES[:4])
def test_rsem_run_em_value_bits_32_matches_reference(name, extra, tmp_path):
    """The program with --value-bits 32 (switches the format once the model rounds are over): the reference's ROUND count,
    theta within 1e-6 of the reference's golden .theta, the same .ofg."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = rf.fixture(name)
    dst = os.path.join(str(tmp_path), name)
    shutil.copytree(fx, dst)
    meta = rf.read_meta(fx)
    for f in ("stat/s.theta", "stat/s.model", "temp/s.ofg", "temp/s.iso_res", "temp/s.gene_res"):
        os.remove(os.path.join(dst, f))
    r = subprocess.run([os.path.join(root, "rsem_amd", "bin", "rsem-run-em"), os.path.join(dst, "ref"), str(meta["model_type"]),
                        os.path.join(dst, "s"), os.path.join(dst, "temp", "s"), os.path.join(dst, "stat", "s"), "-p", "1", "--gibbs-out",
                        "--value-bits", "32"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    ref_log = [l for l in open(os.path.join(fx, "em.log")).read().strip().split("\n") if l.startswith("ROUND")]
    my_log = [l for l in r.stdout.split("\n") if l.startswith("ROUND")]
    assert len(my_log) == len(ref_log) and my_log[-1].split(",")[0] == ref_log[-1].split(",")[0]
    assert my_log[-1].replace(",", "").split()[-1] == ref_log[-1].replace(",", "").split()[-1]  # totNum of the stopping round
    raw, pol = rf.read_theta(os.path.join(dst, "stat", "s.theta"))
    graw, gpol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    for a, b in ((raw, graw), (pol, gpol)):
        big = b >= 1e-7
        assert np.max(np.abs(a[big] - b[big]) / b[big]) < 1e-6 and np.allclose(a, b, rtol=1e-6, atol=1e-10)
    M, N0, rp, sid, val = rf.read_ofg(os.path.join(dst, "temp", "s.ofg"))
    gM, gN0, grp, gsid, gval = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    assert (M, N0) == (gM, gN0) and np.array_equal(rp, grp) and np.array_equal(sid, gsid) and np.allclose(val, gval, rtol=1e-6, atol=0)
