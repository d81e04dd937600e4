"""numpy readers for the reference's intermediate/result files.

Test-side only: deliberately independent of the product's C++ readers so the two cross-check.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["se_noq", "se_q", "se_q_polya_rspd", "pe_noq", "pe_q", "pe_q_polya_rspd", "se_q_fragmean", "se_noq_rev_rspd_omit", "se_q_allele"]


def fixture(name):
    return os.path.join(GOLDEN, name)


def read_meta(fx):
    d = {}
    with open(os.path.join(fx, "META")) as f:
        for line in f:
            k, *v = line.split()
            d[k] = [int(x) for x in v] if len(v) > 1 else int(v[0])
    return d


def read_cnt(path):
    with open(path) as f:
        return [int(x) for x in f.readline().split()]  # N0 N1 N2 N_tot


def read_ofg(path):
    """.ofg (EM.cpp:435-457): 'M N0' then one line per read: sid conprb pairs, noise column sid 0 first."""
    with open(path) as f:
        M, N0 = [int(x) for x in f.readline().split()]
        row_ptr, sid, val = [0], [], []
        for line in f:
            t = line.split()
            sid.extend(int(x) for x in t[0::2])
            val.extend(float(x) for x in t[1::2])
            row_ptr.append(len(sid))
    return M, N0, np.array(row_ptr, np.uint64), np.array(sid, np.int32), np.array(val, np.float64)


def split_noise(row_ptr, sid, val):
    """(items CSR incl. noise column) -> (CSR without noise, ncp per row) as rsem-run-em holds it."""
    N1 = len(row_ptr) - 1
    ncp = np.zeros(N1)
    keep = sid != 0
    rows = np.repeat(np.arange(N1), np.diff(row_ptr).astype(np.int64))
    ncp[rows[~keep]] = val[~keep]
    lens = np.bincount(rows[keep], minlength=N1)
    rp = np.zeros(N1 + 1, np.uint64)
    rp[1:] = np.cumsum(lens)
    return rp, np.ascontiguousarray(sid[keep]), np.ascontiguousarray(val[keep]), ncp


def read_theta(path):
    with open(path) as f:
        n = int(f.readline())
        raw = np.array(f.readline().split(), np.float64)
        pol = np.array(f.readline().split(), np.float64)
    assert len(raw) == n and len(pol) == n
    return raw, pol


def read_seq_lens(path):
    """ref.seq (RefSeq.h:108-138) -> fullLen[M+1], totLen[M+1] (index 0 unused)."""
    full, tot = [0], [0]
    with open(path) as f:
        lines = f.read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        a, b = lines[i].split()
        full.append(int(a))
        tot.append(int(b))
    return np.array(full, np.int32), np.array(tot, np.int32)


def read_grp(path):
    with open(path) as f:
        return np.array(f.read().split(), np.int32)


def read_model(path):
    """stat.model (SingleQModel.h:383-411 etc.): returns dict with type, probF, gld, mld, rspd, mw and raw tables."""
    with open(path) as f:
        tok = f.read().split()
    p = 0

    def nxt(n=1, conv=float):
        nonlocal p
        v = [conv(x) for x in tok[p:p + n]]
        p += n
        return v

    def lendist():
        lb, ub, span = nxt(3, int)
        pdf = np.zeros(span + 1)
        pdf[1:] = nxt(span)
        cdf = np.zeros(span + 1)
        for i in range(1, span + 1):  # LenDist.h read(): cdf[i] = cdf[i-1] + pdf[i]
            cdf[i] = cdf[i - 1] + pdf[i]
        return (lb, ub, span, pdf, cdf)

    out = {}
    t = nxt(1, int)[0]
    out["type"] = t
    out["probF"] = nxt(1)[0]
    out["gld"] = lendist()
    if t < 2:
        has = nxt(1, int)[0]
        out["mld"] = lendist() if has else None
    else:
        out["mld"] = lendist()
    est = nxt(1, int)[0]
    if est:
        B = nxt(1, int)[0]
        out["rspd"] = np.array(nxt(B))
    else:
        out["rspd"] = None
    if t in (1, 3):
        size = nxt(1, int)[0]
        out["qd_init"] = np.array(nxt(size))
        out["qd_tran"] = np.array(nxt(size * size)).reshape(size, size)
        s2, nc = nxt(2, int)
        out["qpro"] = np.array(nxt(s2 * nc * nc)).reshape(s2, nc, nc)
        s3, nc = nxt(2, int)
        out["nqpro"] = np.array(nxt(s3 * nc)).reshape(s3, nc)
    else:
        L, nc = nxt(2, int)
        out["pro"] = np.array(nxt(L * nc * nc)).reshape(L, nc, nc)
        nc = nxt(1, int)[0]
        out["npro"] = np.array(nxt(nc))
    if p < len(tok):
        M = nxt(1, int)[0]
        out["M"] = M
        out["mw"] = np.array(nxt(M + 1))
    return out


def read_res(path):
    """iso_res / gene_res: row-major tab-separated lines (WriteResults.h:223-352)."""
    with open(path) as f:
        return [line.rstrip("\n").split("\t") for line in f]


def read_countvectors(path):
    return np.loadtxt(path, dtype=np.int32, ndmin=2)


def gibbs_setup(fx, M, N0, N1):
    """init_counts (0 / -1 for omitted transcripts), pseudo count and totc as rsem-run-gibbs derives them
    (Gibbs.cpp:152-167)."""
    meta = read_meta(fx)
    pseudoC = meta.get("pseudo_count_x1000", 1000) / 1000.0
    init = np.zeros(M + 1, np.int32)
    with open(os.path.join(fx, "temp", "s.omit")) as f:
        for tok in f.read().split():
            init[int(tok)] = -1
    totc = (M + 1 - int((init < 0).sum())) * pseudoC + N0 + N1
    return init, pseudoC, totc


def per_target_rows(fx, em_only=False):
    """Per-reference-sequence rows (one entry per internal sid 1..M) of the reference's result file: iso_res, or
    allele_res for allele-specific references (WriteResults.h:262-290 vs 223-260)."""
    allele = os.path.exists(os.path.join(fx, "ref.ta"))
    name = "s.allele_res" if allele else "s.iso_res"
    res = read_res(os.path.join(fx, "temp", name + (".em" if em_only else "")))
    o = 1 if allele else 0       # allele_res has one more leading id row
    g = 2 if allele else 0       # ... and one more percentage row before the Gibbs rows
    out = dict(eel=np.array(res[3 + o], float), count=np.array(res[4 + o], float), tpm=np.array(res[5 + o], float),
               fpkm=np.array(res[6 + o], float))
    if len(res) > 8 + g:
        out.update(pme_c=np.array(res[8 + g], float), sd=np.array(res[9 + g], float), pme_tpm=np.array(res[10 + g], float),
                   pme_fpkm=np.array(res[11 + g], float))
    return out
