"""The exact Gibbs sampler's sweep with ONE workgroup per chain (rsem_amd/csrc/gibbs_exact_team.hpp over gibbs_exact_wg.hpp -- the
files gibbs.hip compiles for the GPU -- with a team of 1; teams of several workgroups: tests/test_gibbs_exact_team_emu_cpu.py)
run on the CPU by tests/gibbs_exact_team_emu.cpp: one OS thread per lane, four waves, the phases between workgroup barriers, the
fixed-point rounds inside a tile.  Its count vectors after every sweep must be the oracle chain's (Gibbs.cpp:265-311 with
MT19937 and sampling.h's sample()) BIT FOR BIT -- on data built to collide: few transcripts, so that most reads of a tile
share transcripts with earlier reads of the same tile, reads moving to and from the noise transcript, tiles cut short
by the item capacity, a read longer than a tile holds.  No GPU involved."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(CXX is None, reason="needs g++")


def _build(tmp_path_factory, name, defs):
    exe = os.path.join(str(tmp_path_factory.mktemp(name)), name)
    subprocess.check_call([CXX, "-O1", "-std=c++17", "-pthread"] + defs + os.environ.get("RSEM_EMU_DEFS", "").split()  # (variant builds by hand)
                          + [os.path.join(ROOT, "tests", "gibbs_exact_team_emu.cpp"), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    return _build(tmp_path_factory, "gibbs_exact_emu", [])


@pytest.fixture(scope="module")
def emulator_512(tmp_path_factory):
    """the variant with 512 threads per chain (gibbs_exact_wg.hpp: RSEM_GX_THREADS): four more waves that own no read"""
    return _build(tmp_path_factory, "gibbs_exact_emu_512", ["-DRSEM_GX_THREADS=512"])


def _items(seed, M, N1, maxlen, noise_scale, long_read=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, maxlen + 1, N1).astype(np.int64)
    if long_read:
        lens[N1 // 3] = long_read
    lens += 1  # the noise item comes first in every read (.ofg)
    rp = np.zeros(N1 + 1, np.uint64)
    rp[1:] = np.cumsum(lens)
    n = int(rp[-1])
    sid = np.zeros(n, np.int32)
    cp = np.zeros(n)
    for i in range(N1):
        a, b = int(rp[i]), int(rp[i + 1])
        k = b - a - 1
        if k <= M:
            start = int(rng.integers(1, M + 1))
            ids = (start - 1 + np.arange(k)) % M + 1      # neighbours in id space (isoforms of a gene), distinct
        else:
            ids = rng.integers(1, M + 1, k)               # (a read longer than the transcriptome: ids repeat)
        sid[a + 1:b] = ids
        cp[a] = noise_scale * 10.0 ** rng.uniform(-3, 0)
        cp[a + 1:b] = 10.0 ** rng.uniform(-3, 0, k)
    return rp, sid, cp


def _run(exe, M, rp, sid, cp, init, rounds, seed, N0, pseudoC, tile_items=0, alpha=None, W=1):
    d = tempfile.mkdtemp()
    try:
        inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([M, len(rp) - 1, rounds, seed, N0, 0 if alpha is None else 1, 0, 0], np.int32).tobytes())
            f.write(np.array([pseudoC], np.float64).tobytes())
            for a, t in ((rp, np.uint64), (sid, np.int32), (cp, np.float64), (init, np.int32)):
                f.write(np.ascontiguousarray(a, t).tobytes())
            if alpha is not None:
                f.write(np.ascontiguousarray(alpha, np.float64).tobytes())
        subprocess.check_call([exe, inp, outp, str(W)], timeout=1200, stderr=subprocess.DEVNULL)
        return np.fromfile(outp, np.int32).reshape(rounds, M + 1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _oracle(M, rp, sid, cp, init, rounds, seed, N0, pseudoC, alpha=None):
    eel, mw, grp = np.full(M + 1, 500.0), np.ones(M + 1), np.array([1, M + 1], np.int32)
    totc = (M + 1) * pseudoC + N0 + (len(rp) - 1)
    cv, _ = orc.gibbs_chain(M, rp, sid, cp, init, alpha, pseudoC, totc, N0, eel, mw, grp, seed, 0, rounds, 1)
    return cv


CASES = [
    dict(seed=1, M=12, N1=1500, maxlen=6, noise_scale=0.3, rounds=4, N0=40, pseudoC=1.0),                     # everybody collides; noise moves
    dict(seed=2, M=300, N1=2500, maxlen=20, noise_scale=1e-3, rounds=3, N0=5, pseudoC=1.0),                   # gene-like, 64-read tiles
    dict(seed=3, M=80, N1=1200, maxlen=60, noise_scale=0.05, rounds=3, N0=0, pseudoC=0.1),                    # tiles cut by the item capacity
    dict(seed=8, M=150, N1=500, maxlen=120, noise_scale=0.05, rounds=2, N0=3, pseudoC=1.0),                   # ~ 70 reads per tile
    dict(seed=9, M=100, N1=300, maxlen=250, noise_scale=0.05, rounds=2, N0=3, pseudoC=1.0),                   # ids repeat inside a read
    dict(seed=4, M=60, N1=700, maxlen=12, noise_scale=0.01, rounds=3, N0=7, pseudoC=1.0, long_read=4500),     # one read longer than a tile
    dict(seed=5, M=5, N1=130, maxlen=4, noise_scale=1.0, rounds=6, N0=3, pseudoC=1.0),                        # a last tile with few reads
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_workgroup_chain_is_the_reference_chain(emulator, case):
    c = dict(case)
    rp, sid, cp = _items(c["seed"], c["M"], c["N1"], c["maxlen"], c["noise_scale"], c.get("long_read", 0))
    init = np.zeros(c["M"] + 1, np.int32)
    got = _run(emulator, c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"], c.get("tile_items", 0))
    want = _oracle(c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"])
    assert got.sum(1).tolist() == [c["N0"] + c["N1"]] * c["rounds"]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5], CASES[6]], ids=lambda c: "seed%d" % c["seed"])
def test_workgroup_chain_with_512_threads_is_the_reference_chain(emulator_512, case):
    test_workgroup_chain_is_the_reference_chain(emulator_512, case)


@pytest.mark.parametrize("seed", [1, 2, 4, 5, 6])
def test_a_predecessor_that_moves_back(emulator, seed):
    """Three reads A {1,2}, B {2,3}, C {3,4} in one tile, counts of 0 / 1 so that one unit flips draws: B's speculative move
    is undone by A's move while C has already taken B's move into account -- C must drop that delta although none of ITS
    predecessors moves any more.  (With that rule removed from the kernel body these seeds give wrong chains.)"""
    M = 4
    rp = np.array([0, 3, 6, 9], np.uint64)
    sid = np.array([0, 1, 2, 0, 2, 3, 0, 3, 4], np.int32)
    cp = np.array([1e-9, 1.0, 1.0, 1e-9, 1.0, 1.0, 1e-9, 1.0, 1.0])
    init = np.zeros(M + 1, np.int32)
    got = _run(emulator, M, rp, sid, cp, init, 40, seed, 0, 0.05)
    assert np.array_equal(got, _oracle(M, rp, sid, cp, init, 40, seed, 0, 0.05))


@pytest.fixture(scope="module")
def emulator_prior(tmp_path_factory):
    """the --prior pass of the two headers (RSEM_GX_PRIOR: per-transcript pseudo counts, tiles of 3072 items)"""
    return _build(tmp_path_factory, "gibbs_exact_emu_prior", ["-DRSEM_GX_PRIOR=1"])


@pytest.mark.parametrize("W", [1, 3])
@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5]], ids=lambda c: "seed%d" % c["seed"])
def test_chain_with_per_transcript_pseudo_counts(emulator_prior, case, W):
    """--prior (Gibbs.cpp:171-194, 300-303): the weight of an item is (count + pseudo_counts[sid]) * conprb; same draws as the
    oracle's chain with that alpha, with one workgroup per chain and with a team of three"""
    c = dict(case)
    rp, sid, cp = _items(c["seed"], c["M"], c["N1"], c["maxlen"], c["noise_scale"], c.get("long_read", 0))
    init = np.zeros(c["M"] + 1, np.int32)
    alpha = np.random.default_rng(100 + c["seed"]).uniform(0.05, 3.0, c["M"] + 1)
    alpha[0] = 1.0
    got = _run(emulator_prior, c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"], alpha=alpha, W=W)
    want = _oracle(c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"], alpha=alpha)
    assert np.array_equal(got, want)
