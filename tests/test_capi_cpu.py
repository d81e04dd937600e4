"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rsem_hip.h declares,
and refuses to run without a GPU (there is no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "rsem_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsem_[a-z_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from rsem_amd import build, capi
    build.build()
    return capi.lib()


def test_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "librsem_hip.so does not export %s" % n


def test_abi_version_and_strerror(lib):
    assert lib.rsem_hip_abi_version() == 1
    assert lib.rsem_hip_strerror(0) == b"ok"
    assert b"gfx950" in lib.rsem_hip_strerror(-4)


def test_no_cpu_fallback(lib):
    """Without a GPU every create fails loudly with RSEM_ERR_NODEVICE (never a silent host path)."""
    from rsem_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    rp = np.array([0, 1], np.uint64)
    sid = np.array([1], np.int32)
    with pytest.raises(capi.RsemHipError) as e:
        capi.EmContext(1, rp, sid, np.array([1e-5]), np.array([1e-9]))
    assert e.value.status == -4


def test_argument_validation_precedes_device_use(lib):
    from rsem_amd import capi
    rp = np.array([0, 2], np.uint64)  # row_ptr[N1] != nnz
    sid = np.array([1], np.int32)
    with pytest.raises(capi.RsemHipError) as e:
        capi.EmContext(1, rp, sid, np.array([1e-5]), np.array([1e-9]))
    assert e.value.status == -1


def test_chain_seeds_match_oracle(lib):
    from oracle import pyoracle as orc
    from rsem_amd import capi
    for seed in (0, 1, 12345, 4294967295):
        assert np.array_equal(capi.gibbs_chain_seeds(seed, 8), orc.chain_seeds(seed, 8))


def test_product_never_imports_oracle():
    """Scope rule: nothing under rsem_amd/ may reference oracle/."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "rsem_amd")):
        if "build" in dp.split(os.sep)[-1:]:
            continue
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    t = f.read()
                if re.search(r"(from|import)\s+oracle|#include\s*[<\"][^>\"]*oracle|liboracle|pyoracle|rsem_oracle|orc_[a-z_]+\(", t):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
