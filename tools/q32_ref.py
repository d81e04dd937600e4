"""numpy statement of the Q32 value-plane rule of rsem_amd/csrc/sell_layout.hpp (q32_scale_of / q32_mantissa).

Test / bench infrastructure: the product never imports this.  A read (CSR row) qualifies when it has 1..256
alignments, its largest value mx is in (0, 1e300), every non-zero value is >= mx * 2^-range_bits and the exponent
e = frexp(mx).exp - 32 lies in [-1000, 900]; its values then become rint(v * 2^-e) * 2^e (mantissa capped at
2^32 - 1).  Everything else keeps its doubles.
"""
import numpy as np


def quantize_q32(row_ptr, conprb, range_bits=8, max_row=256):
    """-> (values as the Q32 E step sees them, bool[N1] which reads were compressed)."""
    rp = row_ptr.astype(np.int64)
    N1 = len(rp) - 1
    lens = np.diff(rp)
    rows = np.repeat(np.arange(N1), lens)
    cp = np.asarray(conprb, np.float64)
    ne = lens > 0                      # reduceat over the non-empty rows
    st = rp[:-1][ne]
    mx, mn, bad = np.zeros(N1), np.full(N1, np.inf), np.zeros(N1, bool)
    if len(cp):
        mx[ne] = np.maximum.reduceat(np.where(cp >= 0, cp, np.inf), st)
        mn[ne] = np.minimum.reduceat(np.where(cp > 0, cp, np.inf), st)
        bad[ne] = np.logical_or.reduceat(~(cp >= 0), st)
    _, ex = np.frexp(mx)
    e = ex.astype(np.int64) - 32
    with np.errstate(over="ignore", under="ignore"):
        ok = (mx > 0) & (mx < 1e300) & (e >= -1000) & (e <= 900) & (mn >= np.ldexp(mx, -range_bits)) & ~bad
        ok &= (lens >= 1) & (lens <= max_row)
        m = np.rint(np.ldexp(cp, -e[rows]))
        m = np.minimum(m, 4294967295.0)
        q = np.ldexp(m, e[rows])
    return np.where(ok[rows], q, cp), ok
