"""ctypes binding of librsem_hip.so (include/rsem_hip.h).  No fallbacks: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSEM_HIP_LIB: another build of the same library (tuning experiments build variants next to it: tools/build_variants.sh)
LIB_PATH = os.environ.get("RSEM_HIP_LIB") or os.path.join(_HERE, "librsem_hip.so")

KERNEL_AUTO, KERNEL_CSR, KERNEL_SELL, KERNEL_LANE = 0, 1, 2, 3
GIBBS_EXACT, GIBBS_PARALLEL = 0, 1

_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


class RsemHipError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("librsem_hip: status %d (%s): %s" % (status, _strerror(status), detail))
        self.status = status


ABI_VERSION = 4  # rsem_hip_abi_version() of the include/rsem_hip.h these bindings were written against


class CiProfile(C.Structure):
    _fields_ = [("sample_ms", C.c_double), ("sort_ms", C.c_double), ("interval_ms", C.c_double), ("total_ms", C.c_double),
                ("n_draws", C.c_uint64), ("n_keys_sorted", C.c_uint64)]


class GibbsProfile(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("sweep_ms", C.c_double), ("sweeps", C.c_int64), ("chains", C.c_int32), ("team", C.c_int32), ("reduce_ms", C.c_double)]


class EmProfile(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("estep_ms_sum", C.c_double), ("estep_launches", C.c_int32),
                ("rounds", C.c_int32), ("algorithmic_bytes_per_round", C.c_uint64)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -m rsem_amd.build` (hipcc, gfx950). "
                              "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, i32, u64, dbl, ci = C.c_void_p, C.c_int32, C.c_uint64, C.c_double, C.c_int
        L.rsem_hip_strerror.restype = C.c_char_p
        L.rsem_hip_strerror.argtypes = [ci]
        L.rsem_hip_last_error.restype = C.c_char_p
        L.rsem_hip_device_count.argtypes = [C.POINTER(ci)]
        L.rsem_hip_device_info.argtypes = [ci, C.c_char_p, C.POINTER(C.c_int64)]
        L.rsem_hip_abi_version.restype = ci
        if L.rsem_hip_abi_version() != ABI_VERSION:  # the ctypes structures below mirror ONE layout of include/rsem_hip.h
            raise ImportError("%s speaks ABI %d, rsem_amd/capi.py expects %d: rebuild it (`python -m rsem_amd.build --force`)"
                              % (LIB_PATH, L.rsem_hip_abi_version(), ABI_VERSION))
        L.rsem_hip_stream_probe.argtypes = [ci, u64, ci, C.POINTER(dbl), C.POINTER(dbl)]
        L.rsem_em_create.argtypes = [C.POINTER(vp), ci, i32, u64, u64, _u64p, vp, vp, vp]
        L.rsem_em_set_values.argtypes = [vp, _f64p, _f64p]
        L.rsem_em_get_values.argtypes = [vp, _f64p, _f64p]
        L.rsem_em_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.rsem_em_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
        L.rsem_em_destroy.argtypes = [vp]
        L.rsem_em_set_comm.argtypes = [vp, vp]
        L.rsem_em_set_progress.argtypes = [vp, vp, vp]
        L.rsem_em_shard_rows.argtypes = [u64, _u64p, ci, _u64p]
        L.rsem_em_step.argtypes = [vp, _f64p, dbl, vp, vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(i32)]
        L.rsem_em_run.argtypes = [vp, _f64p, dbl, ci, ci, ci, C.POINTER(ci), vp, C.POINTER(dbl), C.POINTER(i32), vp]
        L.rsem_em_expected_weights.argtypes = [vp, _f64p, dbl, vp, vp, vp]
        L.rsem_gibbs_create.argtypes = [C.POINTER(vp), ci, i32, u64, u64, _u64p, _i32p, _f64p, _i32p, vp, dbl, dbl, u64,
                                        _f64p, _f64p, i32, _i32p]
        L.rsem_gibbs_run.argtypes = [vp, ci, C.c_uint32, ci, ci, ci, ci, vp, _f64p, _f64p, _f64p, _f64p, _f64p, vp]
        L.rsem_gibbs_run_chains.argtypes = [vp, ci, ci, _u32p, ci, _i32p, ci, ci, vp, _f64p, _f64p, _f64p, _f64p, _f64p, vp, vp]
        L.rsem_gibbs_set_comm.argtypes = [vp, vp]
        L.rsem_gibbs_set_allele_groups.argtypes = [vp, i32, _i32p]
        L.rsem_gibbs_destroy.argtypes = [vp]
        L.rsem_comm_unique_id.argtypes = [C.c_char_p]
        L.rsem_comm_create.argtypes = [C.POINTER(vp), ci, ci, ci, C.c_char_p]
        L.rsem_comm_create_local.argtypes = [C.POINTER(vp), ci, C.POINTER(ci)]
        L.rsem_comm_rank.argtypes = [vp]
        L.rsem_comm_world.argtypes = [vp]
        L.rsem_comm_allreduce_f64.argtypes = [vp, vp, u64, vp]
        L.rsem_comm_destroy.argtypes = [vp]
        L.rsem_gibbs_chain_seeds.argtypes = [C.c_uint32, ci, _u32p]
        L.rsem_ci_calculate.argtypes = [ci, i32, i32, i32, _i32p, _f64p, _f64p, dbl, u64, dbl, i32, _i32p, i32, vp,
                                        _f32p, _f32p, _f32p, _f32p, vp, vp, vp]
        L.rsem_ci_sample.argtypes = [ci, i32, i32, i32, _i32p, _f64p, _f64p, dbl, u64, _f32p, _f32p]
        L.rsem_ci_intervals.argtypes = [ci, C.c_int64, i32, _f32p, dbl, _f32p, _f32p, _f32p]
        _lib = L
    return _lib


def _strerror(status):
    return lib().rsem_hip_strerror(status).decode()


def _check(status):
    if status != 0:
        raise RsemHipError(status, lib().rsem_hip_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    _check(lib().rsem_hip_device_count(C.byref(n)))
    return n.value


def stream_probe(device=0, nbytes=4 << 30, reps=5):
    """Measured device STREAM rates in GB/s: (read-only, copy with read + write bytes counted)."""
    r, c = C.c_double(), C.c_double()
    _check(lib().rsem_hip_stream_probe(device, int(nbytes), int(reps), C.byref(r), C.byref(c)))
    return r.value, c.value


class EmContext:
    """One GPU's EM shard (rsem_em_ctx).  Mirrors E_STEP<> / EM<> of EM.cpp for frozen CSR values."""

    def __init__(self, M, row_ptr, sid, conprb=None, ncp=None, device=0):
        self.M = int(M)
        self.N1 = len(row_ptr) - 1
        self.nnz = len(sid)
        row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
        sid = np.ascontiguousarray(sid, np.int32)
        if conprb is not None:
            conprb = np.ascontiguousarray(conprb, np.float64)
            ncp = np.ascontiguousarray(ncp, np.float64)
        self._h = C.c_void_p()
        _check(lib().rsem_em_create(C.byref(self._h), device, self.M, self.N1, self.nnz, row_ptr, _ptr(sid), _ptr(conprb),
                                    _ptr(ncp)))

    def close(self):
        if self._h:
            lib().rsem_em_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_values(self, conprb, ncp):
        _check(lib().rsem_em_set_values(self._h, np.ascontiguousarray(conprb, np.float64),
                                        np.ascontiguousarray(ncp, np.float64)))

    def set_option(self, key, value):
        _check(lib().rsem_em_set_option(self._h, key.encode(), int(value)))

    def info(self, key):
        v = C.c_int64()
        _check(lib().rsem_em_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def get_values(self):
        cp, ncp = np.zeros(self.nnz), np.zeros(self.N1)
        _check(lib().rsem_em_get_values(self._h, cp, ncp))
        return cp, ncp

    def set_comm(self, comm):
        _check(lib().rsem_em_set_comm(self._h, comm._h if comm is not None else None))

    def set_progress(self, fn):
        """fn(round, sum, bChange, totNum) for every round of run() (EM.cpp:415), or None."""
        self._cb = PROGRESS_FN(lambda r, s, b, t, u: fn(r, s, b, t)) if fn else None
        _check(lib().rsem_em_set_progress(self._h, C.cast(self._cb, C.c_void_p) if fn else None, None))

    def step(self, theta, N0):
        theta = np.ascontiguousarray(theta, np.float64)
        counts, theta_new = np.zeros(self.M + 1), np.zeros(self.M + 1)
        s, b, t = C.c_double(), C.c_double(), C.c_int32()
        _check(lib().rsem_em_step(self._h, theta, float(N0), _ptr(counts), _ptr(theta_new), C.byref(s), C.byref(b),
                                  C.byref(t)))
        return counts, theta_new, s.value, b.value, t.value

    def run(self, theta, N0, round0=0, min_round=20, max_round=10000, profile=False):
        theta = np.array(theta, np.float64)
        counts = np.zeros(self.M + 1)
        r, b, t = C.c_int(), C.c_double(), C.c_int32()
        prof = EmProfile() if profile else None
        _check(lib().rsem_em_run(self._h, theta, float(N0), round0, min_round, max_round, C.byref(r), _ptr(counts),
                                 C.byref(b), C.byref(t), C.cast(C.pointer(prof), C.c_void_p) if profile else None))
        out = dict(theta=theta, rounds=r.value, counts=counts, bChange=b.value, totNum=t.value)
        if profile:
            out["profile"] = prof
        return out

    def expected_weights(self, theta, N0, want_weights=True):
        theta = np.ascontiguousarray(theta, np.float64)
        counts = np.zeros(self.M + 1)
        w = np.zeros(self.nnz) if want_weights else None
        wn = np.zeros(self.N1) if want_weights else None
        _check(lib().rsem_em_expected_weights(self._h, theta, float(N0), _ptr(counts), _ptr(w), _ptr(wn)))
        return counts, w, wn


class GibbsContext:
    """One GPU's Gibbs sampler state (rsem_gibbs_ctx).  Mirrors Gibbs() of Gibbs.cpp."""

    def __init__(self, M, row_ptr, sid, conprb, init_counts, alpha, pseudoC, totc, N0, eel, mw, grp, device=0):
        self.M = int(M)
        self.m = len(grp) - 1
        self._h = C.c_void_p()
        alpha = None if alpha is None else np.ascontiguousarray(alpha, np.float64)
        _check(lib().rsem_gibbs_create(C.byref(self._h), device, self.M, len(row_ptr) - 1, len(sid),
                                       np.ascontiguousarray(row_ptr, np.uint64), np.ascontiguousarray(sid, np.int32),
                                       np.ascontiguousarray(conprb, np.float64),
                                       np.ascontiguousarray(init_counts, np.int32), _ptr(alpha), float(pseudoC),
                                       float(totc), int(N0), np.ascontiguousarray(eel, np.float64),
                                       np.ascontiguousarray(mw, np.float64), self.m, np.ascontiguousarray(grp, np.int32)))

    def close(self):
        if self._h:
            lib().rsem_gibbs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, mode, seed, burnin, nsamples, gap, thin=1, want_vectors=True):
        cv = np.zeros((nsamples, self.M + 1), np.int32) if want_vectors else None
        acc = [np.zeros(self.M + 1) for _ in range(4)] + [np.zeros(self.m)]
        ms = C.c_double()
        _check(lib().rsem_gibbs_run(self._h, mode, int(seed), burnin, nsamples, gap, thin, _ptr(cv), *acc,
                                    C.cast(C.pointer(ms), C.c_void_p)))
        return cv, acc, ms.value

    def set_comm(self, comm):
        _check(lib().rsem_gibbs_set_comm(self._h, comm._h if comm is not None else None))

    def set_allele_groups(self, ta):
        ta = np.ascontiguousarray(ta, np.int32)
        self.m_trans = len(ta) - 1
        _check(lib().rsem_gibbs_set_allele_groups(self._h, self.m_trans, ta))

    def run_chains(self, mode, seeds, burnin, nsamples, gap, thin=1, want_vectors=True):
        """rsem_gibbs_run_chains: all chains of this GPU in one call.  Returns (list of count-vector arrays or None,
        [pme_c, pve_c, pme_tpm, pme_fpkm, pve_c_genes] summed over the chains, pve_c_trans or None, GibbsProfile)."""
        seeds = np.ascontiguousarray(seeds, np.uint32)
        ns = np.ascontiguousarray(nsamples, np.int32)
        n = len(seeds)
        assert len(ns) == n
        cvs, ptrs = None, None
        if want_vectors:
            cvs = [np.zeros((int(k), self.M + 1), np.int32) for k in ns]
            ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in cvs])
        acc = [np.zeros(self.M + 1) for _ in range(4)] + [np.zeros(self.m)]
        mt = getattr(self, "m_trans", 0)
        trans = np.zeros(mt) if mt else None
        prof = GibbsProfile()
        _check(lib().rsem_gibbs_run_chains(self._h, mode, n, seeds, burnin, ns, gap, thin, ptrs, *acc, _ptr(trans),
                                           C.cast(C.pointer(prof), C.c_void_p)))
        return cvs, acc, trans, prof


COMM_ID_BYTES = 128


class Comm:
    """rsem_comm: one rank of a communicator (RCCL, or the same-process LOCAL kind)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib().rsem_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def create(cls, device, rank, world, uid):
        h = C.c_void_p()
        _check(lib().rsem_comm_create(C.byref(h), device, rank, world, uid))
        return cls(h)

    @classmethod
    def create_local(cls, devices):
        n = len(devices)
        hs = (C.c_void_p * n)()
        _check(lib().rsem_comm_create_local(hs, n, (C.c_int * n)(*devices)))
        return [cls(C.c_void_p(h)) for h in hs]

    @property
    def rank(self):
        return lib().rsem_comm_rank(self._h)

    @property
    def world(self):
        return lib().rsem_comm_world(self._h)

    def allreduce(self, d_ptr, n, stream=0):
        _check(lib().rsem_comm_allreduce_f64(self._h, d_ptr, n, stream))

    def close(self):
        if self._h:
            lib().rsem_comm_destroy(self._h)
            self._h = C.c_void_p()


def em_shard_rows(row_ptr, world):
    """rsem_em_shard_rows: the reference's split of the reads over `world` workers (EM.cpp:135-157)."""
    rp = np.ascontiguousarray(row_ptr, np.uint64)
    out = np.zeros(world + 1, np.uint64)
    _check(lib().rsem_em_shard_rows(len(rp) - 1, rp, world, out))
    return [int(x) for x in out]


def gibbs_chain_seeds(seed, n):
    out = np.zeros(n, np.uint32)
    _check(lib().rsem_gibbs_chain_seeds(int(seed), n, out))
    return out


def ci_intervals(rows, confidence, device=0):
    """calcCI (calcCI.cpp:216-284) for every row of a (nrows, nSamples) float32 array -> lb, ub, cqv."""
    rows = np.ascontiguousarray(rows, np.float32)
    n, ns = rows.shape
    lb, ub, cqv = (np.zeros(n, np.float32) for _ in range(3))
    _check(lib().rsem_ci_intervals(device, n, ns, rows, float(confidence), lb, ub, cqv))
    return lb, ub, cqv


def ci_sample(cvecs, nSpC, eel, mw, pseudoC=1.0, seed=0, device=0):
    """Phase I of calcCI.cpp alone: (M, nCV*nSpC) float32 TPM samples and l_bars."""
    cvecs = np.ascontiguousarray(cvecs, np.int32)
    nCV, M1 = cvecs.shape
    M = M1 - 1
    tpm = np.zeros((M, nCV * nSpC), np.float32)
    lbar = np.zeros(nCV * nSpC, np.float32)
    _check(lib().rsem_ci_sample(device, M, nCV, nSpC, cvecs, np.ascontiguousarray(eel, np.float64),
                                np.ascontiguousarray(mw, np.float64), float(pseudoC), int(seed), tpm, lbar))
    return tpm, lbar


def ci_calculate(cvecs, nSpC, eel, mw, gene_starts, confidence=0.95, pseudoC=1.0, seed=0, trans_starts=None, device=0):
    """rsem-calculate-credibility-intervals in one call.  Returns a dict of (3, n) float32 arrays [lb, ub, cqv]:
    tpm, fpkm (n = M), gene_tpm, gene_fpkm (n = m) and, with trans_starts, iso_tpm, iso_fpkm; plus 'profile'."""
    cvecs = np.ascontiguousarray(cvecs, np.int32)
    nCV, M1 = cvecs.shape
    M = M1 - 1
    gs = np.ascontiguousarray(gene_starts, np.int32)
    m = len(gs) - 1
    out = {k: np.zeros((3, n), np.float32) for k, n in (("tpm", M), ("fpkm", M), ("gene_tpm", m), ("gene_fpkm", m))}
    ts = None
    mt = 0
    if trans_starts is not None:
        ts = np.ascontiguousarray(trans_starts, np.int32)
        mt = len(ts) - 1
        out["iso_tpm"] = np.zeros((3, mt), np.float32)
        out["iso_fpkm"] = np.zeros((3, mt), np.float32)
    prof = CiProfile()
    _check(lib().rsem_ci_calculate(device, M, nCV, nSpC, cvecs, np.ascontiguousarray(eel, np.float64),
                                   np.ascontiguousarray(mw, np.float64), float(pseudoC), int(seed), float(confidence), m, gs, mt,
                                   _ptr(ts), out["tpm"], out["fpkm"], out["gene_tpm"], out["gene_fpkm"], _ptr(out.get("iso_tpm")),
                                   _ptr(out.get("iso_fpkm")), C.addressof(prof)))
    out["profile"] = prof
    return out
