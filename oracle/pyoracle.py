"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "lib"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "rsem_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build()
        L = C.CDLL(so)
        L.orc_em_estep.argtypes = [C.c_int32, C.c_uint64, _u64p, _i32p, _f64p, _f64p, _f64p, _f64p, C.c_void_p, C.c_void_p]
        L.orc_em_estep.restype = None
        L.orc_em_mstep.argtypes = [C.c_int32, C.c_double, _f64p, _f64p, _f64p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.orc_em_mstep.restype = None
        L.orc_em_run.argtypes = [C.c_int32, C.c_uint64, _u64p, _i32p, _f64p, _f64p, C.c_double, _f64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.orc_em_run.restype = C.c_int
        L.orc_calc_eel.argtypes = [C.c_int32, _i32p, _i32p, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        L.orc_calc_eel.restype = None
        L.orc_polish_theta.argtypes = [C.c_int32, _f64p, _f64p, _f64p]
        L.orc_polish_theta.restype = C.c_int
        L.orc_calc_expression.argtypes = [C.c_int32, _f64p, _f64p, _f64p, _f64p]
        L.orc_calc_expression.restype = None
        L.orc_chain_seeds.argtypes = [C.c_uint32, C.c_int, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
        L.orc_chain_seeds.restype = None
        L.orc_gibbs_chain.argtypes = [C.c_int32, C.c_uint64, _u64p, _i32p, _f64p, _i32p, C.c_void_p, C.c_double, C.c_double,
                                      C.c_uint64, _f64p, _f64p, C.c_int32, _i32p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, _f64p, _f64p, _f64p, _f64p, _f64p]
        L.orc_gibbs_chain.restype = None
        L.orc_gibbs_da_chain.argtypes = [C.c_int32, C.c_uint64, _u64p, _i32p, _f64p, _i32p, C.c_double, C.c_uint64, C.c_uint32,
                                         C.c_int, C.c_int, C.c_int, C.c_int, _f64p, _f64p]
        L.orc_gibbs_da_chain.restype = None
        _f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.orc_calc_ci.argtypes = [C.c_int, _f32p, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_calc_ci.restype = None
        L.orc_ci_transform.argtypes = [C.c_int32, _f64p, _i32p, _f64p, _f64p, _f32p]
        L.orc_ci_transform.restype = C.c_float
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def em_estep(M, row_ptr, sid, conprb, ncp, theta, want_weights=False):
    N1 = len(row_ptr) - 1
    counts = np.zeros(M + 1)
    w = np.zeros(len(sid)) if want_weights else None
    wn = np.zeros(N1) if want_weights else None
    lib().orc_em_estep(M, N1, row_ptr, sid, conprb, ncp, theta, counts, _ptr(w), _ptr(wn))
    return (counts, w, wn) if want_weights else counts


def em_mstep(M, N0, counts, theta_old):
    theta_new = np.zeros(M + 1)
    s, b, t = C.c_double(), C.c_double(), C.c_int32()
    counts = counts.copy()
    lib().orc_em_mstep(M, float(N0), counts, theta_old, theta_new, C.byref(s), C.byref(b), C.byref(t))
    return counts, theta_new, s.value, b.value, t.value


def em_run(M, row_ptr, sid, conprb, ncp, N0, theta, round0=0, min_round=20, max_round=10000):
    theta = theta.copy()
    b, t = C.c_double(), C.c_int32()
    r = lib().orc_em_run(M, len(row_ptr) - 1, row_ptr, sid, conprb, ncp, float(N0), theta, round0, min_round, max_round,
                         C.byref(b), C.byref(t))
    return theta, r, b.value, t.value


def calc_eel(M, fullLen, totLen, gld):
    lb, ub, span, pdf, cdf = gld
    eel = np.zeros(M + 1)
    lib().orc_calc_eel(M, fullLen, totLen, lb, ub, span, pdf, cdf, eel)
    return eel


def polish_theta(M, theta, eel, mw):
    t = theta.copy()
    rc = lib().orc_polish_theta(M, t, eel, mw)
    assert rc == 0
    return t


def calc_expression(M, theta, eel):
    tpm, fpkm = np.zeros(M + 1), np.zeros(M + 1)
    lib().orc_calc_expression(M, theta, eel, tpm, fpkm)
    return tpm, fpkm


def chain_seeds(seed, n):
    out = np.zeros(n, np.uint32)
    lib().orc_chain_seeds(seed, n, out)
    return out


def gibbs_chain(M, row_ptr, sid, conprb, init_counts, alpha, pseudoC, totc, N0, eel, mw, grp, mt_seed,
                burnin, nsamples, gap):
    m = len(grp) - 1
    cv = np.zeros((nsamples, M + 1), np.int32)
    acc = [np.zeros(M + 1) for _ in range(4)] + [np.zeros(m)]
    lib().orc_gibbs_chain(M, len(row_ptr) - 1, row_ptr, sid, conprb, init_counts, _ptr(alpha), float(pseudoC), float(totc),
                          int(N0), eel, mw, m, grp, int(mt_seed), burnin, nsamples, gap, _ptr(cv), *acc)
    return cv, acc


def calc_ci(samples, confidence):
    """calcCI.cpp:216-284 on a copy of `samples` (float32) -> (lb, ub, cqv) as float32."""
    a = np.ascontiguousarray(samples, np.float32).copy()
    lb, ub, cqv = C.c_float(), C.c_float(), C.c_float()
    lib().orc_calc_ci(len(a), a, float(confidence), C.byref(lb), C.byref(ub), C.byref(cqv))
    return np.float32(lb.value), np.float32(ub.value), np.float32(cqv.value)


def ci_transform(gam, cvec, eel, mw):
    """calcCI.cpp:129-149 for one vector of gamma variates -> (tpm float32[M+1], l_bar)."""
    M = len(gam) - 1
    tpm = np.zeros(M + 1, np.float32)
    lbar = lib().orc_ci_transform(M, np.ascontiguousarray(gam, np.float64), np.ascontiguousarray(cvec, np.int32),
                                  np.ascontiguousarray(eel, np.float64), np.ascontiguousarray(mw, np.float64), tpm)
    return tpm, np.float32(lbar)


def gibbs_da_chain(M, row_ptr, sid, conprb, init_counts, pseudoC, N0, mt_seed, burnin, nsamples, gap, thin):
    """CPU model of the drop-in's PARALLEL sampler (see rsem_oracle.h) -> (sum of counts, sum of counts^2) over the samples."""
    a, b = np.zeros(M + 1), np.zeros(M + 1)
    lib().orc_gibbs_da_chain(M, len(row_ptr) - 1, row_ptr, sid, conprb, init_counts, float(pseudoC), int(N0), int(mt_seed),
                             burnin, nsamples, gap, thin, a, b)
    return a, b
