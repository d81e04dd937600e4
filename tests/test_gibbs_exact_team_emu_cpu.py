"""The reference's Gibbs chain advanced by a TEAM of workgroups (rsem_amd/csrc/gibbs_exact_team.hpp -- the file gibbs.hip compiles
for the GPU as k_gibbs_exact_team) run on the CPU by tests/gibbs_exact_team_emu.cpp: W workgroups of one OS thread per lane side
by side, each taking one tile of a window of W tiles, the moves of earlier tiles published through the team's tables, the team
barrier as the very spin loop the GPU runs.  The count vectors after every sweep must be the oracle chain's (Gibbs.cpp:265-311
with MT19937 and sampling.h's sample()) BIT FOR BIT whatever W is, on the collision-heavy data of test_gibbs_exact_emu_cpu.py
(12 transcripts: every tile's draws depend on every earlier tile's moves).  The emulator itself checks after every sweep that the
tables are back at bias / zero and that every arrival at a team barrier was counted.  No GPU involved."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from tests.test_gibbs_exact_emu_cpu import CASES, ROOT, CXX, _items, _oracle

pytestmark = pytest.mark.skipif(CXX is None, reason="needs g++")


def _build(tmp_path_factory, name, defs):
    exe = os.path.join(str(tmp_path_factory.mktemp(name)), name)
    subprocess.check_call([CXX, "-O1", "-std=c++17", "-pthread"] + defs + os.environ.get("RSEM_EMU_DEFS", "").split()
                          + [os.path.join(ROOT, "tests", "gibbs_exact_team_emu.cpp"), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def team_256(tmp_path_factory):
    return _build(tmp_path_factory, "gibbs_exact_team_emu_256", [])  # the product's workgroup: 256 threads


@pytest.fixture(scope="module")
def team_512(tmp_path_factory):
    return _build(tmp_path_factory, "gibbs_exact_team_emu_512", ["-DRSEM_GX_THREADS=512"])


def _run(exe, W, M, rp, sid, cp, init, rounds, seed, N0, pseudoC):
    d = tempfile.mkdtemp()
    try:
        inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([M, len(rp) - 1, rounds, seed, N0, 0, 0, 0], np.int32).tobytes())
            f.write(np.array([pseudoC], np.float64).tobytes())
            for a, t in ((rp, np.uint64), (sid, np.int32), (cp, np.float64), (init, np.int32)):
                f.write(np.ascontiguousarray(a, t).tobytes())
        r = subprocess.run([exe, inp, outp, str(W)], timeout=1200, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.fromfile(outp, np.int32).reshape(rounds, M + 1), r.stderr
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _check(exe, W, case):
    c = dict(case)
    rp, sid, cp = _items(c["seed"], c["M"], c["N1"], c["maxlen"], c["noise_scale"], c.get("long_read", 0))
    init = np.zeros(c["M"] + 1, np.int32)
    got, log = _run(exe, W, c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"])
    want = _oracle(c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"])
    assert np.array_equal(got, want)
    return log


@pytest.mark.parametrize("W", [1, 2, 3])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_team_chain_is_the_reference_chain(team_256, case, W):
    log = _check(team_256, W, case)
    if W > 1:
        assert "team barriers" in log and " 0 team barriers" not in log  # the windows really went through the team's protocol


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5]], ids=lambda c: "seed%d" % c["seed"])
def test_team_of_512_thread_workgroups(team_512, case):
    _check(team_512, 2, case)


@pytest.mark.parametrize("W", [8, 18])
def test_wide_teams_and_the_group_cells(team_256, W):
    """18 workgroups: workgroups 16 and 17 sum the first group's cell of gnet and their own group's cells of net"""
    _check(team_256, W, CASES[0])
    _check(team_256, W, CASES[1])




@pytest.mark.parametrize("seed", [1, 2, 4, 5, 6])
def test_a_predecessor_that_moves_back(team_256, seed):
    """the three reads of test_gibbs_exact_emu_cpu.py's test of the same name: a delta must be dropped although none of the
    read's predecessors moves any more -- here the round that drops it is a phase of the window's loop"""
    M = 4
    rp = np.array([0, 3, 6, 9], np.uint64)
    sid = np.array([0, 1, 2, 0, 2, 3, 0, 3, 4], np.int32)
    cp = np.array([1e-9, 1.0, 1.0, 1e-9, 1.0, 1.0, 1e-9, 1.0, 1.0])
    init = np.zeros(M + 1, np.int32)
    got, _ = _run(team_256, 1, M, rp, sid, cp, init, 40, seed, 0, 0.05)
    assert np.array_equal(got, _oracle(M, rp, sid, cp, init, 40, seed, 0, 0.05))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_single_read_tiles_across_a_team(team_256, seed):
    """reads of 3000 items: one read per tile, so every dependence between reads is a dependence between TILES -- the chain
    A -> B -> C of the test above played across workgroups (a published move taken back by a later phase)"""
    rng = np.random.default_rng(seed)
    M, N1, k = 6, 9, 3000
    rp = (np.arange(N1 + 1) * k).astype(np.uint64)
    sid = rng.integers(0, M + 1, N1 * k).astype(np.int32)
    cp = 10.0 ** rng.uniform(-2, 0, N1 * k)
    init = np.zeros(M + 1, np.int32)
    for W in (2, 4):
        got, _ = _run(team_256, W, M, rp, sid, cp, init, 12, seed, 0, 0.05)
        assert np.array_equal(got, _oracle(M, rp, sid, cp, init, 12, seed, 0, 0.05))


# ---- the same body under ThreadSanitizer -------------------------------------------------------------------------------------------
# One OS thread per lane, pthread barriers for __syncthreads() / the wave barrier, CPU atomics for the LDS and global atomics: what
# ThreadSanitizer then reports is an LDS or global access of two lanes that no barrier of the kernel orders -- the missing
# __syncthreads() that shows on the GPU once in many runs (or never on one compiler version).  The two accesses that overlap on purpose
# (a look at a hash slot others may be claiming; lanes storing the same zero) go through GX_LDS_PEEK32 / GX_LDS_STORE_SAME.  With the
# barrier behind the initial assignment's draw taken out, these cases report races (tried by hand; profiles/HISTORY.md).
def _tsan_build(tmp_path_factory, name, defs):
    exe = os.path.join(str(tmp_path_factory.mktemp(name)), name)
    r = subprocess.run([CXX, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread"] + defs
                       + [os.path.join(ROOT, "tests", "gibbs_exact_team_emu.cpp"), "-o", exe], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        pytest.skip("this g++ cannot build with -fsanitize=thread: " + r.stderr[-300:])
    probe = subprocess.run([exe], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)  # (no arguments: usage, exit 2 -- or the runtime refuses to start)
    if "ThreadSanitizer" in probe.stderr and "FATAL" in probe.stderr:
        pytest.skip("ThreadSanitizer does not start here: " + probe.stderr[-300:])
    return exe


@pytest.fixture(scope="module")
def team_tsan(tmp_path_factory):
    return _tsan_build(tmp_path_factory, "gibbs_exact_team_tsan", [])


@pytest.fixture(scope="module")
def team_tsan_prior(tmp_path_factory):
    return _tsan_build(tmp_path_factory, "gibbs_exact_team_tsan_prior", ["-DRSEM_GX_PRIOR=1"])


@pytest.mark.parametrize("ci,W", [(0, 3), (5, 1), (6, 8)] + ([(0, 1), (4, 1), (5, 3), (6, 2)] if os.environ.get("RSEM_TSAN_ALL") else []))  # (all of CASES x W = 1, 2, 3, 8 were run once by hand: clean)
def test_no_unordered_accesses_between_lanes(team_tsan, ci, W, monkeypatch):
    case = CASES[ci]
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=0 exitcode=66")
    log = _check(team_tsan, W, case)  # (a report makes the exit code 66: _run asserts 0 and shows the report)
    assert "ThreadSanitizer" not in log


def test_no_unordered_accesses_between_lanes_prior(team_tsan_prior, monkeypatch):
    from tests.test_gibbs_exact_emu_cpu import _run as run_alpha
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=0 exitcode=66")
    c = dict(CASES[2])
    rp, sid, cp = _items(c["seed"], c["M"], c["N1"], c["maxlen"], c["noise_scale"], 0)
    init = np.zeros(c["M"] + 1, np.int32)
    alpha = np.random.default_rng(100 + c["seed"]).uniform(0.05, 3.0, c["M"] + 1)
    alpha[0] = 1.0
    for W in (1, 3):
        got = run_alpha(team_tsan_prior, c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"], alpha=alpha, W=W)
        assert np.array_equal(got, _oracle(c["M"], rp, sid, cp, init, c["rounds"], 1000 + c["seed"], c["N0"], c["pseudoC"], alpha=alpha))


@pytest.mark.parametrize("W", [1, 2])
def test_tile_and_item_capacity_boundaries(team_256, W):
    """one read; reads of the noise transcript alone; 255 / 256 / 257 / 512 / 513 reads (a tile takes 256); a read of 4095 / 4096 / 4097
    items (a tile's LDS takes 4096: the last one is walked alone); items with conprb 0 inside reads -- for one workgroup and for a team
    of two (with fewer tiles than workgroups in most of these)"""
    rng = np.random.default_rng(3)

    def run(M, rp, sid, cp, rounds=3, seed=7, N0=2, pc=1.0):
        rp, sid, cp = np.asarray(rp, np.uint64), np.asarray(sid, np.int32), np.asarray(cp, float)
        init = np.zeros(M + 1, np.int32)
        got, _ = _run(team_256, W, M, rp, sid, cp, init, rounds, seed, N0, pc)
        assert np.array_equal(got, _oracle(M, rp, sid, cp, init, rounds, seed, N0, pc))

    run(3, [0, 3], [0, 1, 2], [0.1, 1, 1])
    run(3, [0, 1, 2, 3], [0, 0, 0], [1, 1, 1])
    for n in (255, 256, 257, 512, 513):
        k = 3
        run(5, np.arange(n + 1) * k, np.concatenate([[0] + list(rng.integers(1, 6, k - 1)) for _ in range(n)]), rng.uniform(0.01, 1, n * k))
    for k in (4095, 4096, 4097):
        sid = [0, 1] + [0] + list(rng.integers(1, 6, k - 1)) + [0, 2]
        run(5, [0, 2, 2 + k, 4 + k], sid, rng.uniform(0.01, 1, len(sid)))
    n, k = 700, 4
    cp = rng.uniform(0.01, 1, n * k)
    cp[rng.random(n * k) < 0.4] = 0.0
    cp[::k] = 0.3
    run(6, np.arange(n + 1) * k, np.concatenate([[0] + list(rng.integers(1, 7, k - 1)) for _ in range(n)]), cp)
