"""The per-wave body of the Gibbs PARALLEL sweep kernel (rsem_amd/csrc/gibbs_block.hpp -- the file gibbs.hip compiles for the
GPU) run on the CPU by tests/gibbs_emu.cpp (thread per lane, tests/simt_emu.hpp).  z_i | g must pick alignment j of read i
with probability g_j * conprb_j / (g_0 * ncp_i + sum_k g_k * conprb_k): every sweep assigns every read once, and the picks
per transcript over a few sweeps sit where the binomial says (z-scores: rms ~1), whatever the block length and the size
of the LDS window.  No GPU involved."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import test_estep_emu_cpu as te

ROOT = te.ROOT
pytestmark = pytest.mark.skipif(not os.path.exists(te.CC), reason="needs hipcc (host compilation of the HIP headers)")

BUILDS = {"product": [], "tsan": ["-fsanitize=thread", "-g"]}  # (tsan: None where the toolchain cannot build it)


@pytest.fixture(scope="module")
def emulators(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("gibbs_emu"))
    procs = {}
    for name, defs in BUILDS.items():
        exe = os.path.join(d, "gibbs_emu_" + name)
        procs[name] = (exe, subprocess.Popen([te.CC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-DRSEM_EMU", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("RSEM_EMU_FLAGS", "").split() + defs +
                                             [os.path.join(ROOT, "tests", "gibbs_emu.cpp"), "-o", exe, "-lpthread"],
                                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    out = {}
    for name, (exe, p) in procs.items():
        err = p.communicate()[1]
        assert p.returncode == 0 or name == "tsan", err[-3000:]
        if p.returncode == 0 and name != "tsan":
            out[name] = exe
        if name == "tsan":
            out["_tsan"] = exe if p.returncode == 0 else None
    return out


def _run(exe, M, rp, sid, cp, ncp, g, T, sweeps, seed, window=0, half_units=0):
    d = tempfile.mkdtemp()
    try:
        inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([M, len(rp) - 1, T, sweeps, seed, window, half_units, 0], np.int32).tobytes())
            for a, t in ((rp, np.uint64), (sid, np.int32), (cp, np.float64), (ncp, np.float64), (g, np.float64)):
                f.write(np.ascontiguousarray(a, t).tobytes())
        subprocess.check_call([exe, inp, outp], timeout=900)
        return np.fromfile(outp, np.int32).reshape(sweeps, M + 1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_sweep_body_distribution_and_variants(emulators):
    M, rp, sid, cp, ncp, _ = te._data(1, n=300)   # every read length 1..256 once, then 300 short reads
    rng = np.random.default_rng(9)
    g = rng.gamma(0.5, 1.0, M + 1) + 1e-3
    g[0] = 0.02
    N1 = len(rp) - 1
    rows = np.repeat(np.arange(N1), np.diff(rp.astype(np.int64)))
    w = g[sid] * cp
    tot = np.bincount(rows, weights=w, minlength=N1) + g[0] * ncp
    p = w / tot[rows]
    p0 = g[0] * ncp / tot
    exp = np.bincount(sid, weights=p, minlength=M + 1)
    var = np.bincount(sid, weights=p * (1 - p), minlength=M + 1)
    exp[0], var[0] = p0.sum(), (p0 * (1 - p0)).sum()
    S = 8
    res = {name: _run(exe, M, rp, sid, cp, ncp, g, T=4, sweeps=S, seed=5, window=0) for name, exe in emulators.items() if not name.startswith("_")}
    for name, c in res.items():
        assert np.all(c.sum(1) == N1), name                      # every read picks exactly one of its items in every sweep
        big = S * var > 5
        z = (c.sum(0) - S * exp)[big] / np.sqrt(S * var[big])
        assert big.sum() > 50 and np.abs(z).max() < 5.0 and 0.7 < np.sqrt((z ** 2).mean()) < 1.3, (name, np.abs(z).max(), np.sqrt((z ** 2).mean()))
        assert np.all(c[:, exp == 0] == 0)                        # nothing lands where no alignment points
    small = _run(emulators["product"], M, rp, sid, cp, ncp, g, T=3, sweeps=S, seed=5, window=64)  # odd block length, tiny LDS window
    assert np.array_equal(small, res["product"])                  # the picks depend on the keys only, not on the layout's geometry
    halves = _run(emulators["product"], M, rp, sid, cp, ncp, g, T=7, sweeps=S, seed=5, half_units=1)  # waves that cross from one block into the next
    assert np.array_equal(halves, res["product"])
    assert not np.array_equal(res["product"][0], res["product"][1])  # sweeps differ from one another


def test_no_unordered_accesses_between_lanes(emulators, monkeypatch):
    """The sweep's body under ThreadSanitizer (one OS thread per lane, pthread barriers for the kernel's barriers): a report -- an LDS
    or global access of two lanes that no barrier orders -- makes the emulator exit with 66.  Same picks as the plain build."""
    if emulators["_tsan"] is None:
        pytest.skip("no ThreadSanitizer build with this toolchain")
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=0 exitcode=66")
    M, rp, sid, cp, ncp, _ = te._data(1, n=300)
    g = np.random.default_rng(9).gamma(0.5, 1.0, M + 1) + 1e-3
    a = _run(emulators["_tsan"], M, rp, sid, cp, ncp, g, T=3, sweeps=2, seed=5, window=64)
    b = _run(emulators["product"], M, rp, sid, cp, ncp, g, T=3, sweeps=2, seed=5, window=64)
    assert np.array_equal(a, b)
