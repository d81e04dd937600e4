"""Parity of the HIP EM path (through the C ABI) against the oracle and the reference's golden outputs.

Tolerances: the E step sums each read's fractions in a different order than the reference and uses
f * (1/sum) instead of f / sum, so counts/theta agree to ~1e-15 relative per operation; the
north-star bar is 1e-6 relative on theta.  Tests use 1e-9 for single steps and 1e-6 for whole runs.
"""
import os

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc
from tools.synth_data import make_em_workload

pytestmark = pytest.mark.gpu

VARIANTS = [0, 3]  # AUTO (= LANE), LANE: the thread-per-read (1) and slice-at-a-time (2) kernels of rounds 1-2 were retired in round 6


def capi():
    from rsem_amd import capi as c
    return c


def _fixture_csr(name):
    fx = rf.fixture(name)
    M, N0ofg, rpi, sidi, vali = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    rp, sid, cp, ncp = rf.split_noise(rpi, sidi, vali)
    raw, pol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    N0, N1, N2, Ntot = rf.read_cnt(os.path.join(fx, "stat", "s.cnt"))
    return dict(fx=fx, M=M, N0=N0, Ntot=Ntot, N2=N2, rp=rp, sid=sid, cp=cp, ncp=ncp, raw=raw)


def _rel(a, b, floor=1e-300):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))


@pytest.mark.parametrize("name", rf.FIXTURES)
@pytest.mark.parametrize("variant", VARIANTS)
def test_step_matches_oracle_and_golden(name, variant):
    d = _fixture_csr(name)
    ctx = capi().EmContext(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"])
    ctx.set_option("kernel", variant)
    counts, theta_new, s, b, t = ctx.step(d["raw"], d["N0"])
    oc = orc.em_estep(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"], d["raw"])
    oc, oth, os_, ob, ot = orc.em_mstep(d["M"], d["N0"], oc, d["raw"])
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-12)
    assert np.allclose(theta_new, oth, rtol=1e-9, atol=1e-15)
    assert abs(s - os_) < 1e-9 * os_
    assert t == ot and abs(b - ob) <= 1e-9 * max(ob, 1e-12) + 1e-12  # bChange is a difference quotient of nearly equal numbers
    # golden: expected_count row of the reference's iso_res (printed %.2f)
    gold = rf.per_target_rows(d["fx"], em_only=True)["count"]
    assert np.allclose(counts[1:], gold, atol=0.00501)
    ctx.close()


@pytest.mark.parametrize("name", rf.FIXTURES)
def test_expected_weights(name):
    d = _fixture_csr(name)
    ctx = capi().EmContext(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"])
    counts, w, wn = ctx.expected_weights(d["raw"], d["N0"])
    oc, ow, own = orc.em_estep(d["M"], d["rp"], d["sid"], d["cp"], d["ncp"], d["raw"], want_weights=True)
    oc[0] += d["N0"]
    assert np.allclose(w, ow, rtol=1e-12, atol=0)
    assert np.allclose(wn, own, rtol=1e-12, atol=0)
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-12)
    # every read with a non-zero normaliser distributes exactly one unit of mass
    rows = np.repeat(np.arange(len(d["rp"]) - 1), np.diff(d["rp"].astype(np.int64)))
    tot = np.bincount(rows, weights=w, minlength=len(wn)) + wn
    assert np.all((np.abs(tot - 1.0) < 1e-12) | (tot == 0.0))
    ctx.close()


@pytest.mark.parametrize("name", rf.FIXTURES)
@pytest.mark.parametrize("variant", VARIANTS)
def test_full_run_same_rounds_and_theta(name, variant):
    """Device-resident loop with the reference's stop rule: same ROUND count and theta as the oracle."""
    d = _fixture_csr(name)
    M = d["M"]
    th0 = max(d["N0"] * 1.0 / (d["Ntot"] - d["N2"]), 1e-8)  # EM.cpp:343-346
    theta0 = np.full(M + 1, (1.0 - th0) / M)
    theta0[0] = th0
    ctx = capi().EmContext(M, d["rp"], d["sid"], d["cp"], d["ncp"])
    ctx.set_option("kernel", variant)
    ctx.set_option("check_every", 7)
    out = ctx.run(theta0, d["N0"])
    oth, orounds, ob, ot = orc.em_run(M, d["rp"], d["sid"], d["cp"], d["ncp"], d["N0"], theta0)
    assert out["rounds"] == orounds
    assert out["totNum"] == ot
    big = oth >= 1e-7
    assert _rel(out["theta"][big], oth[big]) < 1e-6
    assert np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12)
    assert abs(out["theta"].sum() - 1.0) < 1e-12
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_synthetic_with_long_rows(variant):
    wl = make_em_workload("tiny", seed=3, long_row_every=2500)
    assert np.diff(wl["row_ptr"].astype(np.int64)).max() > 512
    ctx = capi().EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    ctx.set_option("kernel", variant)
    counts, theta_new, s, b, t = ctx.step(wl["theta0"], wl["N0"])
    oc = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc, oth, os_, ob, ot = orc.em_mstep(wl["M"], wl["N0"], oc, wl["theta0"])
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9)
    assert np.allclose(theta_new, oth, rtol=1e-9, atol=1e-15)
    assert t == ot
    out = ctx.run(wl["theta0"], wl["N0"], max_round=60)
    oth, orounds, ob, ot = orc.em_run(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"],
                                      max_round=60)
    assert out["rounds"] == orounds
    assert np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12)
    ctx.close()


def test_variants_agree_medium():
    wl = make_em_workload("small", seed=5)
    res = []
    for v in VARIANTS:
        ctx = capi().EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
        ctx.set_option("kernel", v)
        res.append(ctx.step(wl["theta0"], wl["N0"]))
        ctx.close()
    oc = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    for r in res:
        assert np.allclose(r[0], oc, rtol=1e-9, atol=1e-9)
        assert r[4] == res[0][4]


def test_edge_cases():
    c = capi()
    # rows whose every term underflows the 1e-300 clamp contribute nothing (EM.cpp:212,219,223)
    M = 3
    rp = np.array([0, 2, 3, 5], np.uint64)
    sid = np.array([1, 2, 3, 1, 3], np.int32)
    cp = np.array([1e-5, 2e-5, 1e-300, 1e-200, 1e-200], np.float64)
    ncp = np.array([1e-9, 0.0, 1e-250], np.float64)
    theta = np.array([0.1, 0.3, 0.3, 0.3])
    for v in VARIANTS:
        ctx = c.EmContext(M, rp, sid, cp, ncp)
        ctx.set_option("kernel", v)
        counts, th, s, b, t = ctx.step(theta, 2.0)
        oc = orc.em_estep(M, rp, sid, cp, ncp, theta)
        oc[0] += 2.0
        assert np.allclose(counts, oc, rtol=1e-12, atol=0)
        assert abs(s - 4.0) < 1e-12  # row 2 (0.3e-299 < 1e-300) is dropped entirely
        ctx.close()
    # empty shard
    ctx = c.EmContext(2, np.array([0], np.uint64), np.zeros(0, np.int32), np.zeros(0), np.zeros(0))
    counts, th, s, b, t = ctx.step(np.array([0.2, 0.4, 0.4]), 5.0)
    assert counts[0] == 5.0 and counts[1:].sum() == 0.0 and s == 5.0
    ctx.close()
    # values supplied later, in file order
    ctx = c.EmContext(M, rp, sid)
    with pytest.raises(c.RsemHipError):
        ctx.step(theta, 0.0)
    ctx.set_values(cp, ncp)
    counts2, *_ = ctx.step(theta, 2.0)
    assert np.allclose(counts2, oc, rtol=1e-12)
    ctx.close()
    # bad sid is rejected
    with pytest.raises(c.RsemHipError) as e:
        c.EmContext(2, np.array([0, 1], np.uint64), np.array([3], np.int32), np.array([1.0]), np.array([0.0]))
    assert e.value.status == -1


def test_full_size_c2_properties():
    """BASELINE configs[1] at full size: one step against the oracle + size-independent invariants."""
    wl = make_em_workload("C2")
    N1 = len(wl["row_ptr"]) - 1
    ctx = capi().EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    counts, theta_new, s, b, t = ctx.step(wl["theta0"], wl["N0"])
    assert abs(s - (wl["N0"] + N1)) < 1e-6 * N1     # every read carries mass 1 (SUM line, EM.cpp:415)
    assert abs(theta_new.sum() - 1.0) < 1e-12 and counts.min() >= 0.0
    oc = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-7)
    out = ctx.run(wl["theta0"], wl["N0"], min_round=30, max_round=30)
    assert out["rounds"] == 30 and abs(out["theta"].sum() - 1.0) < 1e-12
    oth, orounds, _, _ = orc.em_run(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], min_round=30, max_round=30)
    assert orounds == 30 and np.allclose(out["theta"], oth, rtol=1e-9, atol=1e-18)
    ctx.close()


def test_retired_kernel_variants_are_refused():
    """The cross-check kernels of rounds 1-2 (thread per read over the CSR, a slice at a time) no longer ship: the option says so
    instead of silently running something else."""
    c = capi()
    wl = make_em_workload("tiny", seed=1)
    ctx = c.EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    for v in (1, 2):
        with pytest.raises(c.RsemHipError) as e:
            ctx.set_option("kernel", v)
        assert "retired" in str(e.value)
    ctx.set_option("kernel", 3)
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert counts.sum() > 0
    ctx.close()


def test_full_size_c3_step_vs_oracle():
    """BASELINE configs[2] (the north-star config: 50 M reads, 200 k transcripts, 570 M alignments) at full size: one E + M
    step against the oracle's single-thread restatement (EM.cpp:199-236, 385-413), 1e-9 per count."""
    wl = make_em_workload("C3")
    N1 = len(wl["row_ptr"]) - 1
    ctx = capi().EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    counts, theta_new, s, b, t = ctx.step(wl["theta0"], wl["N0"])
    oraw = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc, ot, os_, ob, otn = orc.em_mstep(wl["M"], wl["N0"], oraw, wl["theta0"])  # oc = counts incl. +N0 (EM.cpp:392)
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-6)
    assert abs(s - (wl["N0"] + N1)) < 1e-6 * N1 and abs(theta_new.sum() - 1.0) < 1e-12
    assert np.allclose(theta_new, ot, rtol=1e-9, atol=1e-18) and t == otn
    c2, _, _ = ctx.expected_weights(wl["theta0"], wl["N0"], want_weights=False)  # K5 counts = the E step's counts
    assert np.allclose(c2, oc, rtol=1e-9, atol=1e-6)
    ctx.close()


def test_unstructured_tuples_all_variants():
    """No gene structure at all (every read hits random transcripts): the lane kernel's tuple runs and LDS window give
    nothing, every count goes through the out-of-window path -- results must not depend on that."""
    wl = make_em_workload("tinyR", seed=4)
    M = wl["M"]
    oc = orc.em_estep(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    for v in VARIANTS:
        ctx = capi().EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
        ctx.set_option("kernel", v)
        counts, *_ = ctx.step(wl["theta0"], wl["N0"])
        assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9), v
        ctx.close()
    wl = make_em_workload("C2R", scale=0.05)  # 500 k reads over 50 k transcripts: windows of 2048 ids cover 4 % of a unit's hits
    ctx = capi().EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    out = ctx.run(wl["theta0"], wl["N0"], max_round=200)
    oth, orounds, _, _ = orc.em_run(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], max_round=200)
    assert out["rounds"] == orounds and np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12)
    # Such reads are laid out as SPLIT rows (sell_layout.hpp: the alignments inside the read's window in the planes, the
    # others as far entries summed per read before and per transcript after the lane kernel -- no global atomic per
    # alignment): most reads here, and nearly all alignments are far entries.  Whole rows give the same numbers.
    N1, nnz = len(wl["row_ptr"]) - 1, len(wl["sid"])
    assert ctx.info("split_rows") > 0.5 * N1 and 0.5 * nnz < ctx.info("far_entries") < nnz
    oc = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    c_split, *_ = ctx.step(wl["theta0"], wl["N0"])
    ctx.set_option("split_rows", 0)
    assert ctx.info("split_rows") == 0 and ctx.info("far_entries") == 0
    c_whole, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert np.allclose(c_split, oc, rtol=1e-9, atol=1e-9) and np.allclose(c_whole, oc, rtol=1e-9, atol=1e-9)
    out2 = ctx.run(wl["theta0"], wl["N0"], max_round=200)
    assert out2["rounds"] == orounds and np.allclose(out2["theta"], oth, rtol=1e-6, atol=1e-12)
    ctx.close()


@pytest.mark.skipif(not os.environ.get("RSEM_TEST_XL"), reason="needs ~110 GB of host memory and 60 GB of HBM: set RSEM_TEST_XL=1")
def test_more_than_2_to_32_alignments():
    """4.4 G alignments in ONE shard (BASELINE configs[4] scale): every index past 2^32 (plane offsets of the sliced layout,
    CSR offsets of the weights pass) must be 64-bit.  108 M reads x 41 alignments over 500 k transcripts with a cheap
    closed-form structure; one step against the oracle."""
    N1, L, M = 108_000_000, 41, 500_000
    rp = (np.arange(N1 + 1, dtype=np.uint64) * np.uint64(L))
    nnz = N1 * L
    assert nnz > 2 ** 32
    base = (np.arange(N1, dtype=np.int64) * 7919) % (M - L)
    sid = (np.repeat(base, L) + np.tile(np.arange(L, dtype=np.int64), N1) + 1).astype(np.int32)
    del base
    rng = np.random.default_rng(1)
    cp = rng.random(nnz) * 1e-20
    ncp = np.full(N1, 1e-60)
    theta = np.full(M + 1, 0.95 / M)
    theta[0] = 0.05
    ctx = capi().EmContext(M, rp, sid, cp, ncp)
    counts, theta_new, s, b, t = ctx.step(theta, 1000.0)
    ctx.close()
    oc = orc.em_estep(M, rp, sid, cp, ncp, theta)
    oc[0] += 1000.0
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-6)
    assert abs(s - (1000.0 + N1)) < 1.0


@pytest.mark.parametrize("loop", ["solo", "fused"])
def test_device_loops_equal_kernel_per_step_loop(loop, monkeypatch):
    """rsem_em_run's loops that never materialise theta between rounds -- the E step reads theta out of the previous
    round's counts -- against the E-step -> M-step kernel sequence (RSEM_EM_FUSED=0).  "solo" (the default): a round is
    ONE launch, whose workgroups also close the previous round (statistics, stop rule, ROUND line); "fused" (sharded
    runs on large matrices): statistics on a second stream.  theta_i is the same expression evaluated in another kernel,
    so the runs agree to the noise of the floating-point atomics; same ROUND count, same final statistics, every ROUND
    line present."""
    wl = make_em_workload("small", seed=21)
    M = wl["M"]
    ctx = capi().EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    lines = []
    ctx.set_progress(lambda r, s, b, t: lines.append((r, s, b, t)))
    if loop == "fused":
        monkeypatch.setenv("RSEM_EM_FUSED", "1")
    else:
        monkeypatch.delenv("RSEM_EM_FUSED", raising=False)
    dev = ctx.run(wl["theta0"], wl["N0"], max_round=3000)
    assert [l[0] for l in lines] == list(range(1, dev["rounds"] + 1))
    assert abs(lines[-1][1] - (wl["N0"] + len(wl["row_ptr"]) - 1)) < 1e-6 and lines[-1][3] == dev["totNum"]
    monkeypatch.setenv("RSEM_EM_FUSED", "0")
    ctx.set_progress(None)
    plain = ctx.run(wl["theta0"], wl["N0"], max_round=3000)
    assert dev["rounds"] == plain["rounds"] and dev["totNum"] == plain["totNum"]
    assert np.allclose(dev["theta"], plain["theta"], rtol=1e-10, atol=1e-18)
    assert np.allclose(dev["counts"], plain["counts"], rtol=1e-10, atol=1e-9)
    assert abs(dev["bChange"] - plain["bChange"]) <= 1e-6 * max(1e-30, abs(plain["bChange"]))
    oth, orounds, _, _ = orc.em_run(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], max_round=3000)
    assert dev["rounds"] == orounds and np.allclose(dev["theta"], oth, rtol=1e-6, atol=1e-12)
    # a start in the middle (round0 > 0, as rsem-run-em does after its model rounds), a hard stop at max_round, one round only
    if loop == "fused":
        monkeypatch.setenv("RSEM_EM_FUSED", "1")
    else:
        monkeypatch.delenv("RSEM_EM_FUSED")
    part = ctx.run(wl["theta0"], wl["N0"], round0=11, min_round=20, max_round=37)
    assert part["rounds"] == 37
    monkeypatch.setenv("RSEM_EM_FUSED", "0")
    part0 = ctx.run(wl["theta0"], wl["N0"], round0=11, min_round=20, max_round=37)
    assert part0["rounds"] == 37 and part["totNum"] == part0["totNum"] and np.allclose(part["theta"], part0["theta"], rtol=1e-10, atol=1e-18)
    one0 = ctx.run(wl["theta0"], wl["N0"], round0=0, min_round=1, max_round=1)
    if loop == "fused":
        monkeypatch.setenv("RSEM_EM_FUSED", "1")
    else:
        monkeypatch.delenv("RSEM_EM_FUSED")
    one = ctx.run(wl["theta0"], wl["N0"], round0=0, min_round=1, max_round=1)
    assert one["rounds"] == one0["rounds"] == 1 and one["totNum"] == one0["totNum"]
    assert np.allclose(one["theta"], one0["theta"], rtol=1e-12, atol=1e-18) and np.allclose(one["counts"], one0["counts"], rtol=1e-12, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("config,scale", [("C3X", 0.01), ("smallX", 1.0)])
def test_reads_that_also_hit_another_gene(config, scale, monkeypatch):
    """10-30 % of the reads also hit 1-3 transcripts of a far-away gene (paralogs, cross-gene multi-mappers): their layout key
    is the anchor id (sell_layout.hpp row_key_of), the foreign ids take the out-of-window path.  Step and whole runs (all
    three loops) against the oracle, F64 and Q32 planes."""
    wl = make_em_workload(config, scale=scale, seed=11)
    M = wl["M"]
    oc = orc.em_estep(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    ctx = capi().EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9)
    oth, orounds, _, _ = orc.em_run(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], max_round=300)
    for loop in ("0", "1", "2"):  # kernel sequence, statistics on a second stream, one launch per round
        monkeypatch.setenv("RSEM_EM_FUSED", loop)
        out = ctx.run(wl["theta0"], wl["N0"], max_round=300)
        assert out["rounds"] == orounds and np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12), loop
    monkeypatch.delenv("RSEM_EM_FUSED")
    from tools.q32_ref import quantize_q32
    ctx.set_option("value_bits", 32)
    vq = quantize_q32(wl["row_ptr"], wl["conprb"], 8)[0]
    ocq = orc.em_estep(M, wl["row_ptr"], wl["sid"], vq, wl["ncp"], wl["theta0"])
    ocq[0] += wl["N0"]
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert np.allclose(counts, ocq, rtol=1e-9, atol=1e-9)
    ctx.set_option("value_bits", 64)
    # the layout: far-reaching reads sort behind the others of their shape and the few compact reads whose foreign id lies
    # within a window's width join them in a second pass (sell_build_refined), so that most units keep every id inside their
    # LDS window; with the knob off they are mixed in and (nearly) every unit has ids outside.  Same counts either way.
    far, units, strays = ctx.info("far_units"), ctx.info("units"), ctx.info("stray_reads")
    ctx.close()
    monkeypatch.setenv("RSEM_HIP_APART", "0")
    ctx = capi().EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9)
    far_mixed = ctx.info("far_units")
    print("%s: units with ids outside their window %d of %d (reads sorted apart in the second pass: %d); all reads in one sequence: %d" % (
        config, far, units, strays, far_mixed))
    assert far < far_mixed and ctx.info("stray_reads") == 0
    ctx.close()


@pytest.mark.parametrize("config,scale", [("smallX", 1.0), ("C3X30", 0.02)])
def test_every_read_with_a_foreign_id_split(config, scale):
    """Option "split_policy" 2 (rsem_hip.h): EVERY read with an id outside the window of its own gene is laid out as a split row (its
    in-window alignments in the planes, the others as far entries of the two side passes), not only the reads that are mostly outside;
    a split row's window ends where its unit's does (sell_refine_split_windows); the split rows' chain of kernels runs on a stream of
    its own beside the compact reads ("split_overlap").  Step and whole run against the oracle, with and without the second stream,
    and back to the default layout."""
    wl = make_em_workload(config, scale=scale, seed=11)
    M = wl["M"]
    oc = orc.em_estep(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    oth, orounds, _, _ = orc.em_run(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"], max_round=200)
    ctx = capi().EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    split_default = ctx.info("split_rows")
    ctx.set_option("split_policy", 2)
    N1 = len(wl["row_ptr"]) - 1
    assert ctx.info("split_rows") > max(split_default, 0.05 * N1) and ctx.info("far_entries") > 0
    for overlap in (1, 0):
        ctx.set_option("split_overlap", overlap)
        counts, *_ = ctx.step(wl["theta0"], wl["N0"])
        assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9), overlap
        out = ctx.run(wl["theta0"], wl["N0"], max_round=200)
        assert out["rounds"] == orounds and np.allclose(out["theta"], oth, rtol=1e-6, atol=1e-12), overlap
    c2, w, wn = ctx.expected_weights(wl["theta0"], wl["N0"])  # the weights pass walks the caller-order CSR: whole rows whatever the layout
    assert np.allclose(c2, oc, rtol=1e-9, atol=1e-9)
    ctx.set_option("split_policy", 1)
    assert ctx.info("split_rows") == split_default
    counts, *_ = ctx.step(wl["theta0"], wl["N0"])
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-9)
    ctx.close()


def test_release_csr_keeps_every_result_and_gives_the_values_back_bit_for_bit():
    """Option "release_csr" (rsem_hip.h): the caller-order ids and values -- 12 of a context's ~25 bytes per alignment -- are freed while
    the theta-only rounds run and read back from the value planes by whatever needs them next.  The rounds give the same theta, the
    values come back as the very doubles that went in, the weights pass gives the same weights, and the option is refused where the
    planes do not hold everything (Q32 planes here)."""
    from rsem_amd import capi
    from tools.synth_data import make_em_workload
    wl = make_em_workload("small", seed=21)
    M = wl["M"]
    ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    ref = ctx.run(wl["theta0"], wl["N0"], max_round=60)
    c0, w0, wn0 = ctx.expected_weights(ref["theta"], wl["N0"])
    assert ctx.info("csr_released") == 0 and ctx.info("csr_bytes") == 12 * len(wl["sid"])
    ctx.set_option("release_csr", 1)
    assert ctx.info("csr_released") == 1
    again = ctx.run(wl["theta0"], wl["N0"], max_round=60)   # the loop streams the sliced layout alone
    assert again["rounds"] == ref["rounds"] and np.allclose(again["theta"], ref["theta"], rtol=1e-12, atol=0)
    assert ctx.info("csr_released") == 1
    cp, ncp = ctx.get_values()                              # read back from the planes
    assert ctx.info("csr_released") == 0
    assert np.array_equal(cp, wl["conprb"]) and np.array_equal(ncp, wl["ncp"])
    ctx.set_option("release_csr", 1)
    c1, w1, wn1 = ctx.expected_weights(ref["theta"], wl["N0"])  # the weights pass walks the caller-order CSR: restored first
    assert ctx.info("csr_released") == 0
    assert np.allclose(c1, c0, rtol=1e-12, atol=1e-9) and np.array_equal(w1, w0) and np.array_equal(wn1, wn0)
    ctx.set_option("release_csr", 1)
    ctx.set_values(wl["conprb"] * 0.5, wl["ncp"])           # new values: restored, overwritten, planes refilled
    half = ctx.step(wl["theta0"], wl["N0"])
    assert ctx.info("csr_released") == 0 and np.isfinite(half[0]).all()
    ctx.set_option("value_bits", 32)
    if ctx.info("reads_q32") > 0:
        with pytest.raises(Exception):
            ctx.set_option("release_csr", 1)                # Q32 planes hold rounded values
        assert ctx.info("csr_released") == 0
    ctx.close()
