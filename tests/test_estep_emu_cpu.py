"""The per-wave body of the LANE E-step kernel (rsem_amd/csrc/estep_block.hpp -- the file em.hip compiles for the GPU) run on
the CPU by tests/estep_emu.cpp: one OS thread per lane, cross-lane intrinsics as exchanges through memory, the layout
rebuilt on the host with sell_layout.hpp's own index helpers.  Checked against the oracle's E step (EM.cpp:199-236) on reads
of EVERY length 1..256, with clamped terms (theta * conprb < 1e-300), a noise term that matters, tiny LDS windows (the
out-of-window path), theta taken from raw counts (the one-launch round), Q32 planes (against the oracle on the rounded
values: the format is exact), and for other prefetch depths (compile-time constants).
No GPU involved: this is how kernel edits are checked before GPU minutes are spent on them."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as orc
from tools.q32_ref import quantize_q32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(CC), reason="needs hipcc (host compilation of the HIP headers)")

BUILDS = {
    "product": [],
    # (... and a far queue of 260 places per wave, 4 more than one slice can append: it is emptied before nearly every slice that appends)
    "variants": ["-DRSEM_F64_DEPTHS=4,3,3,2", "-DRSEM_Q32_DEPTHS=5,4,3,2", "-DRSEM_FARQ_CAP=260"],
    # the product's body under ThreadSanitizer: one OS thread per lane and pthread barriers for the kernel's barriers, so a report is an
    # LDS / global access of two lanes that no barrier of the kernel orders (None where the toolchain cannot build it)
    "tsan": ["-fsanitize=thread", "-g"],
}


@pytest.fixture(scope="module")
def emulators(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("estep_emu"))
    procs = {}
    for name, defs in BUILDS.items():
        exe = os.path.join(d, "estep_emu_" + name)
        procs[name] = (exe, subprocess.Popen([CC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-DRSEM_EMU", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("RSEM_EMU_FLAGS", "").split() + defs +
                                             [os.path.join(ROOT, "tests", "estep_emu.cpp"), "-o", exe, "-lpthread"],
                                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    out = {}
    for name, (exe, p) in procs.items():
        err = p.communicate()[1]
        assert p.returncode == 0 or name == "tsan", err[-3000:]
        out[name] = exe if p.returncode == 0 else None
    return out


def _data(seed, M=500, n=1200, maxlen=60):
    rng = np.random.default_rng(seed)
    lens = np.concatenate([np.arange(1, 257), rng.integers(1, maxlen, n)]).astype(np.int64)  # every length once, then a bulk
    rng.shuffle(lens)
    rp = np.zeros(len(lens) + 1, np.uint64)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    start = (rng.integers(1, M - 256, len(lens)) // 40) * 40 + 1   # few distinct tuples per length: runs of identical tuples
    rows = np.repeat(np.arange(len(lens)), lens)
    sid = (start[rows] + (np.arange(nnz) - rp[:-1].astype(np.int64)[rows])).astype(np.int32)
    cp = np.power(10.0, rng.uniform(-30, -3, len(lens)))[rows] * np.power(2.0, rng.uniform(-10, 0, nnz))
    ncp = np.power(10.0, rng.uniform(-20, -3, len(lens)))
    theta = rng.random(M + 1)
    theta[rng.random(M + 1) < 0.1] = 1e-310   # theta * conprb under the 1e-300 clamp
    theta[0] = 0.3
    theta /= theta.sum()
    return M, rp, sid, cp, ncp, theta


def _run(exe, M, rp, sid, cp, ncp, theta_in, N0=0.0, T=4, policy=0, q32=0, range_bits=8, from_counts=0, window=0):
    d = tempfile.mkdtemp()
    try:
        inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([M, len(rp) - 1, T, policy, q32, range_bits, from_counts, window], np.int32).tobytes())
            f.write(np.array([N0], np.float64).tobytes())
            for a, t in ((rp, np.uint64), (sid, np.int32), (cp, np.float64), (ncp, np.float64), (theta_in, np.float64)):
                f.write(np.ascontiguousarray(a, t).tobytes())
        subprocess.check_call([exe, inp, outp], timeout=600)
        out = np.fromfile(outp, np.float64)
        return out[:M + 1], out[M + 1], out[M + 2]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _check(exe, seed=1, **kw):
    M, rp, sid, cp, ncp, theta = _data(seed)
    vals = quantize_q32(rp, cp, kw.get("range_bits", 8))[0] if kw.get("q32") else cp
    N0, theta_in = 0.0, theta
    if kw.get("from_counts"):  # theta_i = (c_i + [i = 0] (noise + N0)) / (N0 + reads with a non-zero normaliser), EM.cpp:392-398
        N0 = 37.0
        raw = theta * 1000.0
        tot = np.zeros(128)
        tot[:64] = 0.25 / 64 * 5
        tot[64:] = (1000.0 - N0) / 64
        theta = raw.copy()
        theta[0] += tot[:64].sum() + N0
        theta = theta / (tot[64:].sum() + N0)
        theta_in = np.concatenate([raw, tot])
    oc = orc.em_estep(M, rp, sid, vals, ncp, theta)
    counts, noise, neff = _run(exe, M, rp, sid, cp, ncp, theta_in, N0=N0, **kw)
    assert neff == len(rp) - 1
    assert np.allclose(counts[1:], oc[1:], rtol=1e-12, atol=0.0), kw
    assert abs(noise - oc[0]) <= 1e-12 * oc[0], kw


CASES = [dict(), dict(from_counts=1), dict(window=64), dict(q32=1), dict(q32=1, range_bits=24, T=7, seed=2), dict(T=1, window=16, seed=3),
         # policy=1: the sort key without its apart bit (reads that reach beyond the window stay among the others)
         dict(policy=1, window=64), dict(policy=1, window=16, from_counts=1, T=3, seed=2),
         # policy=2: split rows -- a read that reaches beyond the window keeps its in-window alignments in the planes (F64X shapes,
         # an extra term in its normaliser, its reciprocal handed on), the others are far entries summed before / after the blocks
         dict(policy=2, window=64), dict(policy=2, window=16, T=3, seed=2), dict(policy=2, window=128, T=1, seed=3),
         # policy=3: EVERY read that reaches beyond the window splits (not only those that are mostly outside)
         dict(policy=3, window=64), dict(policy=3, window=16, T=3, seed=2), dict(policy=3, window=256, T=8, seed=4)]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join("%s%s" % kv for kv in sorted(k.items())) or "plain")
def test_kernel_body_as_built_for_the_product(emulators, kw):
    _check(emulators["product"], **kw)


@pytest.mark.parametrize("kw", CASES[:4] + [dict(q32=1, from_counts=1, T=5, seed=2), dict(T=1, window=16, seed=3), dict(policy=1, window=16, from_counts=1, T=3, seed=2)],
                         ids=lambda k: "-".join("%s%s" % kv for kv in sorted(k.items())) or "plain")
def test_other_prefetch_depths(emulators, kw):
    _check(emulators["variants"], **kw)


@pytest.mark.parametrize("kw", [dict(window=64), dict(policy=1, window=16, from_counts=1, T=3, seed=2), dict(q32=1, window=64)],
                         ids=lambda k: "-".join("%s%s" % kv for kv in sorted(k.items())))
def test_far_units_without_the_queue(emulators, kw, monkeypatch):
    """The units with ids outside their window take the far-queue instantiation (a lane's partial counts for such ids wait in LDS and
    leave by atomics the wave then waits for, estep_block.hpp FarQueue) -- every case above with `window`; the loop that issues a global
    atomic per such count (option far_queue 0; the units of split rows) must give the same counts."""
    monkeypatch.setenv("ESTEP_EMU_FAR_QUEUE", "0")
    _check(emulators["product"], **kw)


@pytest.mark.parametrize("kw", [dict(policy=2, window=64), dict(T=1, window=16, seed=3)] + ([dict(from_counts=1), dict(q32=1)] if os.environ.get("RSEM_TSAN_ALL") else []), ids=lambda k: "-".join("%s%s" % kv for kv in sorted(k.items())))
def test_no_unordered_accesses_between_lanes(emulators, kw, monkeypatch):
    """The kernel body under ThreadSanitizer (a report makes the emulator exit with 66).  The one store that overlaps on purpose --
    every lane of a split read stores the same reciprocal -- goes through RSEM_STORE_SAME (simt_macros.hpp)."""
    if emulators["tsan"] is None:
        pytest.skip("no ThreadSanitizer build with this toolchain")
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=0 exitcode=66")
    _check(emulators["tsan"], **kw)
