"""Credibility intervals: the oracle's restatement of calcCI (calcCI.cpp:216-284) pinned on the reference's own sample
matrices (tests/golden/<fx>/ci_pin, made by tests/golden/make_ci_golden.py with the reference binary)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from tests import rsem_files as rf

CI_FIXTURES = ["se_noq", "se_q", "pe_q", "se_q_polya_rspd", "se_q_allele"]


def fmt(v):
    return "%.6g" % float(v)


def load_pin(name):
    fx = rf.fixture(name)
    d = os.path.join(fx, "ci_pin")
    full, tot = rf.read_seq_lens(os.path.join(fx, "ref.seq"))
    M = len(full) - 1
    S = np.fromfile(os.path.join(d, "s.tmp"), np.float32).reshape(M, -1)
    rows = {f[:-4]: [l.split("\t") for l in open(os.path.join(d, f)).read().strip().split("\n")] for f in os.listdir(d) if f.endswith("_res.txt")}
    return fx, M, S, rows


@pytest.mark.parametrize("name", CI_FIXTURES)
def test_oracle_calc_ci_reproduces_reference_tpm_rows(name):
    fx, M, S, rows = load_pin(name)
    per_target = rows["allele_res"] if "allele_res" in rows else rows["iso_res"]
    assert len(per_target) == 6 and len(per_target[0]) == M
    got = [orc.calc_ci(S[j], 0.95) for j in range(M)]
    for r in range(3):
        assert [fmt(g[r]) for g in got] == per_target[r], "TPM row %d" % r
    # gene level: float sums over the gene's transcripts in order, single-isoform genes copy (calcCI.cpp:318-372)
    grp = rf.read_grp(os.path.join(fx, "ref.grp"))
    gene = rows["gene_res"]
    for g in range(len(grp) - 1):
        b, e = grp[g], grp[g + 1]
        if e - b > 1:
            acc = np.zeros(S.shape[1], np.float32)
            for j in range(b, e):
                acc = (acc + S[j - 1]).astype(np.float32)
            ci = orc.calc_ci(acc, 0.95)
        else:
            ci = got[b - 1]
        for r in range(3):
            assert fmt(ci[r]) == gene[r][g], (g, r)
    if "allele_res" in rows:
        # Isoform level of an allele-specific reference (calcCI.cpp:325-341, 376-385).  Reference quirk: the
        # per-transcript accumulator `itsamples` is zeroed once per THREAD (calcCI.cpp:312-316), never between
        # transcripts, and calcCI sorts it in place -- so every transcript after a thread's first inherits the
        # (possibly sorted) sums of its predecessors, and the rows depend on -p.  Emulated here to show the golden
        # rows are understood; the drop-in resets per transcript ("known reference quirks").
        ta = rf.read_grp(os.path.join(fx, "ref.ta"))
        tid_of = np.zeros(M + 2, np.int64)
        for t in range(len(ta) - 1):
            tid_of[ta[t]:ta[t + 1]] = t
        iso = rows["iso_res"]
        m, nt = len(grp) - 1, int(rf.read_meta(fx)["gibbs_threads"])
        quotient = max(M // nt, 1)
        cur, ranges = 0, []
        for i in range(nt):  # calcCI.cpp:405-420
            start, niso = cur, 0
            while (m - cur > nt - i - 1) and (i == nt - 1 or niso < quotient):
                niso += grp[cur + 1] - grp[cur]
                cur += 1
            ranges.append((start, cur))
        res = {}
        for (g0, g1) in ranges:
            it = np.zeros(S.shape[1], np.float32)
            curtid, curaid = -1, -1

            def close(upto):
                nonlocal it
                if upto - curaid > 1:
                    srt = np.sort(it)
                    res[curtid] = orc.calc_ci(it, 0.95)
                    it = srt  # sorted in place by calcCI
                else:
                    res[curtid] = got[curaid - 1]
            for g in range(g0, g1):
                for j in range(grp[g], grp[g + 1]):
                    if curtid != tid_of[j]:
                        if curtid >= 0:
                            close(j)
                        curtid, curaid = int(tid_of[j]), j
                    it = (it + S[j - 1]).astype(np.float32)
            if curtid >= 0:
                close(grp[g1])
        for t in range(len(ta) - 1):
            for r in range(3):
                assert fmt(res[t][r]) == iso[r][t], (t, r)


@pytest.mark.parametrize("name", CI_FIXTURES)
def test_oracle_fpkm_rows_from_recomputed_lbar(name):
    """FPKM samples are 1e3 / l_bar[k] * tpm[k] (calcCI.cpp:345); l_bar is not in s.tmp, but it is
    sum_j tpm_j/1e6 * eel_j, so the reference's FPKM rows follow to float accuracy."""
    fx, M, S, rows = load_pin(name)
    full, tot = rf.read_seq_lens(os.path.join(fx, "ref.seq"))
    model = rf.read_model(os.path.join(fx, "stat", "s.model"))
    lb, ub, span, pdf, cdf = model["gld"]
    eel = orc.calc_eel(M, full, tot, model["gld"])
    lbar = (S.astype(np.float64) / 1e6 * eel[1:, None]).sum(axis=0).astype(np.float32)
    per_target = rows["allele_res"] if "allele_res" in rows else rows["iso_res"]
    for j in range(M):
        f = (1e3 / lbar.astype(np.float64) * S[j]).astype(np.float32)
        ci = orc.calc_ci(f, 0.95)
        for r in range(3):
            ref = float(per_target[3 + r][j])
            assert abs(float(ci[r]) - ref) <= 2e-5 * max(abs(ref), 1e-30) + 1e-30, (j, r, ci[r], ref)


def test_calc_ci_edge_cases():
    z = np.zeros(37, np.float32)
    assert orc.calc_ci(z, 0.95) == (0.0, 0.0, 0.0)
    a = np.array([5.0], np.float32)
    assert orc.calc_ci(a, 0.95)[:2] == (5.0, 5.0)
    # heavy ties: half zeros
    x = np.concatenate([np.zeros(50, np.float32), np.arange(1, 51, dtype=np.float32)])
    lb, ub, cqv = orc.calc_ci(x, 0.9)
    assert lb == 0.0 and ub == 40.0
    for n in (4, 5, 6, 7, 8):  # the four residue classes of Tukey's hinges
        y = np.arange(1, n + 1, dtype=np.float32)
        q = n // 4
        r = n % 4
        if r == 0:
            q1, q3 = (y[q - 1] + y[q]) / 2, (y[3 * q - 1] + y[3 * q]) / 2
        elif r == 3:
            q1, q3 = (y[q] + y[q + 1]) / 2, (y[3 * q + 1] + y[3 * q + 2]) / 2
        else:
            q1, q3 = y[q], y[3 * q]
        assert abs(orc.calc_ci(y, 0.95)[2] - (q3 - q1) / (q3 + q1)) < 1e-6


def test_philox_known_answers(tmp_path):
    """rsem_amd/csrc/rng.hpp's Philox4x32-10 (Gibbs PARALLEL sampler, credibility intervals) and Philox2x32-10 against the
    known-answer vectors of Random123, through the header's own (host-callable) functions: tests/rng_kat_check.cpp."""
    import shutil
    import subprocess
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("needs hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "rng_kat_check")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(root, "tests", "rng_kat_check.cpp"), "-o", exe],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "bad=0" in r.stdout, r.stdout
