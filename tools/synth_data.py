"""Seeded synthetic EM / Gibbs workloads shaped like BASELINE.json's configs.

Gene-family structure so multi-mapping is realistic: genes own contiguous transcript ids (.ti order is
gene-sorted, Transcript.h:53-55); every gene has ~2k "segments" (exon-combination classes), each
compatible with a fixed subset of the gene's isoforms; a read picks a transcript by expression, then
one of the segments containing it, and aligns to exactly that segment's isoform set.  True theta ~
lognormal(0, 2) with 30 % zeros, 5 % noise reads.  conprb = 10^U(-60,-3) per read with a per-hit
jitter 10^N(0,0.5); ncp = 10^U(-130,-40).  Everything is a pure function of (config, seed).
"""
import json
import os

import numpy as np

CONFIGS = {
    # name: (N1, M, mean alignments per read, gene size range)
    "C1": (100_000, 1_000, 2.5, (1, 4)),
    "C2": (10_000_000, 50_000, 5.0, (4, 12)),
    "C3": (50_000_000, 200_000, 10.0, (8, 24)),
    "C5": (100_000_000, 500_000, 40.0, (32, 64)),
    # worst case for the sliced layout: every read hits its own random transcripts all over the id space (no gene
    # structure), so lanes almost never see the same tuple twice and a workgroup's LDS window covers nothing
    "C2R": (10_000_000, 50_000, 5.0, None),
    "tinyR": (20_000, 400, 5.0, None),
    "tiny": (20_000, 400, 5.0, (4, 12)),
    "small": (400_000, 5_000, 5.0, (4, 12)),
}


_ARRAYS = ("row_ptr", "sid", "conprb", "ncp", "theta0")


# configs built by the threaded C++ generator (tools/gen_workload.cpp -> tools/bin/libgenwl.so): name: (N1, M, mean alignments
# per read, gene size range, fraction of reads that also hit 1..3 transcripts of ANOTHER gene)
FAST_CONFIGS = {
    "C5": (100_000_000, 500_000, 40.0, (32, 64), 0.0),
    # configs[2] with cross-gene multi-mappers (paralogs): between C3 (no read leaves its gene) and C2R (no genes at all)
    "C3X": (50_000_000, 200_000, 10.0, (8, 24), 0.10),
    "C3X30": (50_000_000, 200_000, 10.0, (8, 24), 0.30),  # ... and with 30 % such reads
    "tinyX": (20_000, 400, 5.0, (4, 12), 0.25),
    "smallX": (300_000, 20_000, 6.0, (4, 12), 0.30),  # more ids than an LDS window holds: cross-gene hits really leave it
}
_genwl = None


def _genwl_lib():
    global _genwl
    if _genwl is None:
        import ctypes as C
        here = os.path.dirname(os.path.abspath(__file__))
        so, src = os.path.join(here, "bin", "libgenwl.so"), os.path.join(here, "gen_workload.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            import subprocess
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", so, src])
        L = C.CDLL(so)
        L.wl_plan.restype = C.c_void_p
        L.wl_plan.argtypes = [C.c_int64, C.c_int32, C.c_double, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
        L.wl_fill.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.wl_free.argtypes = [C.c_void_p]
        _genwl = L
    return _genwl


def _make_fast(config, seed, scale, shard):
    import ctypes as C
    N1, M, mean_hits, (kmin, kmax), cross = FAST_CONFIGS[config]
    N1 = max(1, int(N1 * scale))
    L = _genwl_lib()
    nnz = C.c_uint64(0)
    h = L.wl_plan(N1, M, mean_hits, kmin, kmax, cross, seed, shard, 0, C.byref(nnz))
    if not h:
        raise ValueError("gen_workload: bad parameters")
    row_ptr, sid = np.empty(N1 + 1, np.uint64), np.empty(nnz.value, np.int32)
    conprb, ncp = np.empty(nnz.value, np.float64), np.empty(N1, np.float64)
    L.wl_fill(h, row_ptr.ctypes.data, sid.ctypes.data, conprb.ctypes.data, ncp.ctypes.data)
    L.wl_free(h)
    theta0 = np.full(M + 1, (1.0 - 0.05) / M)
    theta0[0] = 0.05
    return dict(M=M, N0=int(round(0.05 * N1 / 0.95)), row_ptr=row_ptr, sid=sid, conprb=conprb, ncp=ncp, theta0=theta0, config=config, seed=seed)


def make_em_workload(config="C2", seed=20250925, scale=1.0, long_row_every=0, shard=0):
    """Returns dict(M, N0, row_ptr u64, sid i32, conprb f64, ncp f64, theta0 f64).
    `shard` re-draws the reads (not the transcriptome): shard r of a weak-scaling job.
    RSEM_WL_CACHE=<dir> (e.g. on /dev/shm): the arrays are kept there as .npy files, so that GPU calls which start
    many processes on the same workload (profiling passes, library variants) generate it once."""
    cache = os.environ.get("RSEM_WL_CACHE")
    if not cache:
        return _make_em_workload(config, seed, scale, long_row_every, shard)
    d = os.path.join(cache, "%s_%d_%g_%d_%d" % (config, seed, scale, long_row_every, shard))
    meta = os.path.join(d, "meta.json")
    if os.path.exists(meta):
        with open(meta) as f:
            wl = json.load(f)
        for k in _ARRAYS:
            wl[k] = np.load(os.path.join(d, k + ".npy"))
        return wl
    wl = _make_em_workload(config, seed, scale, long_row_every, shard)
    tmp = d + ".tmp%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    for k in _ARRAYS:
        np.save(os.path.join(tmp, k + ".npy"), wl[k])
    with open(os.path.join(tmp, "meta.json"), "w") as f:
        json.dump({k: v for k, v in wl.items() if k not in _ARRAYS}, f)
    try:
        os.rename(tmp, d)
    except OSError:  # another process was faster
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return wl


def _make_em_workload(config, seed, scale, long_row_every, shard):
    if config in FAST_CONFIGS:
        return _make_fast(config, seed, scale, shard)
    N1, M, mean_hits, ksz = CONFIGS[config]
    N1 = max(1, int(N1 * scale))
    rng = np.random.default_rng(seed)
    if ksz is None:
        return _make_unstructured(config, seed, N1, M, mean_hits, shard)
    kmin, kmax = ksz
    # genes
    sizes = []
    tot = 0
    while tot < M:
        k = int(rng.integers(kmin, kmax + 1))
        k = min(k, M - tot)
        sizes.append(k)
        tot += k
    sizes = np.array(sizes, np.int64)
    gstart = np.concatenate([[0], np.cumsum(sizes)])  # 0-based transcript index
    n_genes = len(sizes)
    # inclusion probability so that E[row length] ~ mean_hits
    mean_k = sizes.mean()
    p_inc = float(np.clip((mean_hits - 1.0) / max(mean_k - 1.0, 1e-9), 0.05, 0.95))
    # segments: 2k per gene, each a subset of the gene's isoforms (non-empty)
    seg_gene = np.repeat(np.arange(n_genes), 2 * sizes)
    n_seg = len(seg_gene)
    seg_k = sizes[seg_gene]
    kcap = int(sizes.max())
    mask = rng.random((n_seg, kcap)) < p_inc
    mask &= np.arange(kcap)[None, :] < seg_k[:, None]
    forced = (rng.random(n_seg) * seg_k).astype(np.int64)  # guarantee one member
    mask[np.arange(n_seg), forced] = True
    seg_len = mask.sum(1)
    seg_ptr = np.concatenate([[0], np.cumsum(seg_len)])
    srow, scol = np.nonzero(mask)
    seg_sid = (gstart[seg_gene[srow]] + scol + 1).astype(np.int32)  # 1-based sids, ascending within a segment
    # transcript -> segments containing it
    order = np.argsort(seg_sid, kind="stable")
    t_sorted = seg_sid[order]
    t_ptr = np.searchsorted(t_sorted, np.arange(1, M + 2))
    t_segs = srow[order]
    # expression (still the structure stream); reads come from a per-shard stream
    theta_true = np.exp(rng.normal(0.0, 2.0, M))
    theta_true[rng.random(M) < 0.3] = 0.0
    has_seg = (t_ptr[1:] - t_ptr[:-1]) > 0
    theta_true[~has_seg] = 0.0
    cdf = np.cumsum(theta_true)
    cdf /= cdf[-1]
    if shard:
        rng = np.random.default_rng([seed, shard])
    t = np.searchsorted(cdf, rng.random(N1), side="right").astype(np.int64)  # 0-based transcript
    t = np.minimum(t, M - 1)
    nseg_t = (t_ptr[t + 1] - t_ptr[t])
    pick = t_ptr[t] + (rng.random(N1) * nseg_t).astype(np.int64)
    seg = t_segs[pick]
    lens = seg_len[seg]
    row_ptr = np.zeros(N1 + 1, np.uint64)
    row_ptr[1:] = np.cumsum(lens)
    nnz = int(row_ptr[-1])
    rows = np.repeat(np.arange(N1), lens)
    within = np.arange(nnz, dtype=np.int64) - row_ptr[:-1].astype(np.int64)[rows]
    sid = seg_sid[seg_ptr[seg][rows] + within]
    e_row = rng.uniform(-60.0, -3.0, N1)
    conprb = np.power(10.0, e_row[rows] + rng.normal(0.0, 0.5, nnz))
    ncp = np.power(10.0, rng.uniform(-130.0, -40.0, N1))
    if long_row_every:  # a few reads with > 512 alignments (exercises the long-row kernel)
        extra_rows = np.arange(0, N1, long_row_every)[:8]
        parts_sid, parts_cp, new_lens = [], [], lens.copy()
        for r in extra_rows:
            L = 513 + int(rng.integers(0, 300))
            parts_sid.append((r, rng.integers(1, M + 1, L).astype(np.int32)))
            parts_cp.append(np.power(10.0, rng.uniform(-30.0, -20.0, L)))
            new_lens[r] = L
        rp2 = np.zeros(N1 + 1, np.uint64)
        rp2[1:] = np.cumsum(new_lens)
        sid2 = np.empty(int(rp2[-1]), np.int32)
        cp2 = np.empty(int(rp2[-1]), np.float64)
        keep = np.ones(N1, bool)
        keep[extra_rows] = False
        krows = np.repeat(keep, lens)
        dst_rows = np.repeat(np.arange(N1), new_lens)
        dst_keep = keep[dst_rows]
        sid2[dst_keep] = sid[krows]
        cp2[dst_keep] = conprb[krows]
        for (r, s), c in zip(parts_sid, parts_cp):
            a, b = int(rp2[r]), int(rp2[r + 1])
            sid2[a:b] = s
            cp2[a:b] = c
        row_ptr, sid, conprb = rp2, sid2, cp2
    N0 = int(round(0.05 * N1 / 0.95))
    theta0 = np.full(M + 1, (1.0 - 0.05) / M)
    theta0[0] = 0.05
    return dict(M=M, N0=N0, row_ptr=row_ptr, sid=np.ascontiguousarray(sid, np.int32),
                conprb=np.ascontiguousarray(conprb), ncp=ncp, theta0=theta0, config=config, seed=seed)


def _make_unstructured(config, seed, N1, M, mean_hits, shard):
    rng = np.random.default_rng([seed, 77, shard])
    lens = np.clip(1 + rng.poisson(mean_hits - 1.0, N1), 1, 16).astype(np.int64)
    row_ptr = np.zeros(N1 + 1, np.uint64)
    row_ptr[1:] = np.cumsum(lens)
    nnz = int(row_ptr[-1])
    sid = rng.integers(1, M + 1, nnz, dtype=np.int32)
    rows = np.repeat(np.arange(N1), lens)
    e_row = rng.uniform(-60.0, -3.0, N1)
    conprb = np.power(10.0, e_row[rows] + rng.normal(0.0, 0.5, nnz))
    ncp = np.power(10.0, rng.uniform(-130.0, -40.0, N1))
    N0 = int(round(0.05 * N1 / 0.95))
    theta0 = np.full(M + 1, (1.0 - 0.05) / M)
    theta0[0] = 0.05
    return dict(M=M, N0=N0, row_ptr=row_ptr, sid=sid, conprb=np.ascontiguousarray(conprb), ncp=ncp, theta0=theta0, config=config, seed=seed)


def to_gibbs_items(wl):
    """EM CSR + ncp  ->  .ofg-style items CSR with the noise column (sid 0) first in every read."""
    rp, sid, cp, ncp = wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"]
    N1 = len(rp) - 1
    lens = np.diff(rp.astype(np.int64)) + 1
    irp = np.zeros(N1 + 1, np.uint64)
    irp[1:] = np.cumsum(lens)
    n = int(irp[-1])
    isid = np.zeros(n, np.int32)
    icp = np.zeros(n, np.float64)
    first = irp[:-1].astype(np.int64)
    is_noise = np.zeros(n, bool)
    is_noise[first] = True
    icp[first] = ncp
    isid[~is_noise] = sid
    icp[~is_noise] = cp
    return irp, isid, icp
