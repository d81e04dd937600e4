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


def test_gibbs_auto_mode_budget(tmp_path):
    """rsem-run-gibbs --gibbs-mode auto (the default) may only pick the exact single-wave chain when it is cheap:
    <= 2.5e7 read-rounds per GPU, counting that a GPU's chains run one after the other.  --dry-run prints the choice
    without touching a device, so the rule is checked here on CPU.  (The first version of the rule budgeted one chain's
    rows and ignored how many chains share the GPU.)  An explicit --gibbs-mode exact is honoured whatever it costs, with
    a warning and a time estimate on stderr -- forcing it for 64 chains on 190 k reads once ran for 20 minutes."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rsem_amd", "bin", "rsem-run-gibbs")
    assert os.path.exists(exe), "rsem-run-gibbs was not built"
    d = str(tmp_path / "fx")
    shutil.copytree(os.path.join(root, "tests", "golden", "se_q"), d)  # N1 = 1410 alignable reads with hits in .ofg

    def choice(*args):
        r = subprocess.run([exe, d + "/ref", d + "/temp/s", d + "/stat/s"] + list(args) + ["-q", "--dry-run"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        line = r.stdout.strip().split("\n")[-1]
        assert line.startswith("dry run: ")
        return line.split()[2], int(line.split("chain(s), ")[1].split()[0])

    n1 = 1410
    assert choice("20", "40", "1", "-p", "2") == ("exact", 1)                      # the fixtures' own settings
    assert choice("200", "1000", "1", "-p", "8") == ("exact", 1)                   # 1410 * 326 * 8 = 3.7e6
    assert n1 * (200 + 1 + 16) * 64 <= 2.5e7 and choice("200", "1000", "1", "-p", "64") == ("exact", 1)
    assert n1 * (2200 + 1 + 16) * 8 > 2.5e7 and choice("2200", "1000", "1", "-p", "8") == ("parallel", 8)   # per-chain rows alone: 3.1e6
    assert n1 * (2200 + 1 + 125) * 8 > 2.5e7 and n1 * (2200 + 1 + 1000) * 1 <= 2.5e7
    assert choice("2200", "1000", "1", "-p", "1") == ("exact", 1)                  # same burn-in, one chain: affordable
    assert choice("200", "1000", "1", "-p", "8", "--gibbs-mode", "parallel") == ("parallel", 8)
    assert choice("200", "1000", "1", "-p", "8", "--gibbs-mode", "parallel", "--gibbs-thin", "3") == ("parallel", 3)
    assert choice("20000", "1000", "1", "-p", "8", "--gibbs-mode", "exact") == ("exact", 1)  # explicit request is honoured ...
    r = subprocess.run([exe, d + "/ref", d + "/temp/s", d + "/stat/s", "20000", "1000", "1", "-p", "8", "--gibbs-mode", "exact", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "Warning" in r.stderr and "exact" in r.stderr and "parallel" in r.stderr  # ... but not silently
    r = subprocess.run([exe, d + "/ref", d + "/temp/s", d + "/stat/s", "20", "40", "1", "-p", "2", "--gibbs-mode", "exact", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "Warning" not in r.stderr
