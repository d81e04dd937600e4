"""rsem_amd -- MI355X-native EM / Gibbs hot path of RSEM (rsem-run-em, rsem-run-gibbs).

The product is rsem_amd/librsem_hip.so (hand-written HIP kernels behind the C ABI of
include/rsem_hip.h) plus the C++ command line programs in rsem_amd/bin.  This Python package is
only the thin ctypes binding used by the tests and bench.py; there is no CPU fallback: importing
rsem_amd.capi fails loudly when the library is missing.
"""
__all__ = ["capi", "build"]
