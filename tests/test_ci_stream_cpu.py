"""rsem-calculate-credibility-intervals --ci-stream reference without a GPU: csrc/host/ci_stream.hpp restates what fixes the
reference's draws (sampling.h:19-44 engine factory, boost 1.55's mt19937 / uniform_01 / exponential / gamma distributions as shipped
with the reference, calcCI.cpp:93-164) -- tests/ci_stream_check.cpp draws a fixture's TPM samples with it, the oracle's interval
arithmetic (calcCI.cpp:216-284) goes on top, and the result must be the rows the REFERENCE BINARY appended for the same --seed and -p
(tests/golden/*/ci_stat, made by tests/golden/make_ci_golden.py), value for value as printed with %.6g: per transcript, TPM and FPKM,
lower bound, upper bound and coefficient of quartile variation."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import rsem_files as rf
from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(CXX is None, reason="needs g++")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = os.path.join(str(tmp_path_factory.mktemp("ci_stream_check")), "ci_stream_check")
    subprocess.check_call([CXX, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "ci_stream_check.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("name", ["se_noq", "se_q", "pe_q", "se_q_polya_rspd"])
def test_reference_stream_gives_the_reference_rows(checker, name, tmp_path):
    fx = rf.fixture(name)
    meta = rf.read_meta(fx)
    cmd = open(os.path.join(fx, "ci_stat", "CMD")).read().split()
    conf, nCV, nSpC = float(cmd[4]), int(cmd[5]), int(cmd[6])
    threads = int(cmd[cmd.index("-p") + 1])
    seed = int(cmd[cmd.index("--seed") + 1])
    pc = float(cmd[cmd.index("--pseudo-count") + 1]) if "--pseudo-count" in cmd else 1.0
    model = rf.read_model(os.path.join(fx, "stat", "s.model"))
    full, tot = rf.read_seq_lens(os.path.join(fx, "ref.seq"))
    M = len(full) - 1
    eel = orc.calc_eel(M, full, tot, model["gld"])
    nfiles = min(threads, nCV)
    parts = [rf.read_countvectors(os.path.join(fx, "temp", "s.countvectors%d" % k)) for k in range(nfiles)]
    assert sum(len(p) for p in parts) == nCV
    inp, outp = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([M, nfiles, nSpC, seed], np.int32).tobytes())
        f.write(np.array([pc], np.float64).tobytes())
        f.write(np.ascontiguousarray(eel, np.float64).tobytes())
        f.write(np.ascontiguousarray(model["mw"], np.float64).tobytes())
        for p in parts:
            f.write(np.array([len(p)], np.int32).tobytes())
            f.write(np.ascontiguousarray(p, np.int32).tobytes())
    subprocess.check_call([checker, inp, outp])
    nS = nCV * nSpC
    raw = np.fromfile(outp, np.float32)
    tpm, lbar = raw[:M * nS].reshape(M, nS), raw[M * nS:]
    gold = np.array([[float(x) for x in l.split("\t")] for l in open(os.path.join(fx, "ci_stat", "iso_res.txt")).read().strip().split("\n")])
    gold_s = [l.split("\t") for l in open(os.path.join(fx, "ci_stat", "iso_res.txt")).read().strip().split("\n")]
    assert gold.shape == (6, M)
    n_same = 0
    for j in range(M):
        row_t = tpm[j]
        row_f = (1e3 / lbar.astype(np.float64) * row_t.astype(np.float64)).astype(np.float32)  # calcCI.cpp:345
        got = list(orc.calc_ci(row_t, conf)) + list(orc.calc_ci(row_f, conf))
        for k in range(6):
            assert abs(float(got[k]) - gold[k, j]) <= 2e-5 * abs(gold[k, j]) + 1e-9, (j, k, got[k], gold[k, j])
            n_same += ("%.6g" % float(got[k])) == gold_s[k][j]
    assert n_same >= 0.995 * 6 * M, (n_same, 6 * M)
