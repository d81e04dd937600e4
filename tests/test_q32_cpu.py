"""The Q32 value-plane rule (tools/q32_ref.py = the numpy statement of sell_layout.hpp's q32_scale_of / q32_mantissa)
and what the rounding does to a whole EM run, measured with the oracle: CPU only."""
import numpy as np

from oracle import pyoracle as orc
from tools.q32_ref import quantize_q32
from tools.synth_data import make_em_workload


def test_rule_error_bound_and_idempotence():
    wl = make_em_workload("tiny", seed=2)
    rp, cp = wl["row_ptr"], wl["conprb"]
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp.astype(np.int64)))
    mx = np.zeros(len(rp) - 1)
    np.maximum.at(mx, rows, cp)
    for D in (0, 4, 8, 16, 24):
        q, ok = quantize_q32(rp, cp, D)
        okj = ok[rows]
        assert np.array_equal(q[~okj], cp[~okj])                       # reads that do not qualify are untouched
        # absolute error <= half a unit of the read's scale = 2^-32 of its largest value
        assert np.all(np.abs(q - cp)[okj] <= mx[rows][okj] * 2.0 ** -32)
        # every compressed value keeps 32 - D significant bits
        assert np.all(np.abs(q - cp)[okj] <= cp[okj] * 2.0 ** -(32 - D))
        # the mantissas are integers below 2^32 and the largest one of a read has its top bit set
        _, ex = np.frexp(mx)
        m = np.ldexp(q, -(ex[rows] - 32))
        assert np.all(m[okj] == np.rint(m[okj])) and m[okj].max() < 2.0 ** 32
        top = np.zeros(len(rp) - 1)
        np.maximum.at(top, rows, m)
        assert np.all(top[ok] >= 2.0 ** 31)
        # a second pass changes nothing (exactly representable values stay put)
        q2, ok2 = quantize_q32(rp, q, D)
        assert np.array_equal(q2[okj], q[okj]) and np.all(ok2[ok])
    frac = [quantize_q32(rp, cp, D)[1].mean() for D in (0, 4, 8, 16)]
    assert frac == sorted(frac) and frac[-1] == 1.0


def test_rule_special_rows():
    rp = np.array([0, 0, 1, 3, 5, 7, 9], np.uint64)  # empty, single, zero + positive, all zero, negative, tiny
    cp = np.array([3.0, 0.0, 2.0, 0.0, 0.0, -1.0, 1.0, 1e-305, 2e-305])
    q, ok = quantize_q32(rp, cp, 8)
    assert list(ok) == [False, True, True, False, False, False]
    assert np.array_equal(q, cp)  # 3.0, 2.0 and 0.0 are exact; the others were not compressed
    rp = np.array([0, 300], np.uint64)
    assert not quantize_q32(rp, np.ones(300), 8)[1][0]  # > 256 alignments: stays in the CSR


def test_oracle_run_on_rounded_values_meets_the_bar():
    """What the format costs in accuracy: the oracle's EM on the rounded values against the oracle's EM on the doubles --
    same ROUND, theta within the north-star's 1e-6 (observed 1e-7 .. 1e-8 at 400 k reads; a 24-bit mantissa would not
    pass: 3e-5)."""
    for seed in (1, 2):
        wl = make_em_workload("tiny", seed=seed)
        M = wl["M"]
        th, r, _, tn = orc.em_run(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["N0"], wl["theta0"])
        for D in (8, 24):
            q, ok = quantize_q32(wl["row_ptr"], wl["conprb"], D)
            assert ok.mean() > 0.9
            th2, r2, _, tn2 = orc.em_run(M, wl["row_ptr"], wl["sid"], q, wl["ncp"], wl["N0"], wl["theta0"])
            big = th >= 1e-7
            assert r2 == r and tn2 == tn
            assert np.max(np.abs(th2 - th)[big] / th[big]) < 1e-6


def test_numpy_rule_is_the_rule_of_the_device_code(tmp_path):
    """tools/q32_ref.quantize_q32 against sell_layout.hpp's own q32_scale_of / q32_mantissa (host-callable; the kernels
    call the same functions): which reads qualify, exponents, every mantissa -- on values spread over the whole exponent
    range, with zeros, ties, negatives and NaN mixed in."""
    import os
    import shutil
    import subprocess
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        import pytest
        pytest.skip("needs hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "q32_rule_check")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value",
                           os.path.join(root, "tests", "q32_rule_check.cpp"), "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rng = np.random.default_rng(11)
    n, L = 20000, 6
    base = np.power(2.0, rng.uniform(-1060, 1000, n))[:, None]           # from subnormal to huge
    vals = base * np.power(2.0, rng.uniform(-14, 0, (n, L)))             # spans below and above 2^-8
    vals[rng.random((n, L)) < 0.05] = 0.0
    vals[::97, 0] = -1.0
    vals[::101, 1] = np.nan
    vals[::7] = np.ldexp(rng.integers(1, 2 ** 33, (len(vals[::7]), L)).astype(np.float64) + 0.5, -40)  # exact .5 ties after scaling
    vals = np.ascontiguousarray(vals)
    inp, outp = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    vals.tofile(inp)
    rp = (np.arange(n + 1) * L).astype(np.uint64)
    for D in (0, 8, 24):
        subprocess.check_call([exe, inp, outp, str(n), str(L), str(D)])
        out = np.fromfile(outp, np.int64).reshape(n, 2 + L)
        with np.errstate(invalid="ignore"):
            q, ok = quantize_q32(rp, vals.reshape(-1), D)
        assert np.array_equal(out[:, 0].astype(bool), ok)
        assert 0.02 < ok.mean() < 0.98 or D == 0
        e = out[:, 1][ok]
        m = out[:, 2:][ok].astype(np.float64)
        assert np.array_equal(np.ldexp(m, e[:, None]), q.reshape(n, L)[ok])   # every mantissa, every exponent
