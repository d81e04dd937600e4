"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rsem_hip.h declares,
and refuses to run without a GPU (there is no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

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
    from rsem_amd import capi
    assert lib.rsem_hip_abi_version() == capi.ABI_VERSION == 4
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


def test_gibbs_sampler_choice(tmp_path):
    """rsem-run-gibbs's sampler choice, checked with --dry-run (no device needed): the default (auto) is the reference's
    own chain (exact: bit-identical count vectors; every chain is one wave and all chains of a GPU advance together, so
    its cost no longer grows with -p); the data-augmentation sampler is only taken on request and says so on stderr even
    with -q; an unknown --gibbs-mode is an error, not a silent fallback; --devices deals the chains to GPU groups."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rsem_amd", "bin", "rsem-run-gibbs")
    assert os.path.exists(exe), "rsem-run-gibbs was not built"
    d = str(tmp_path / "fx")
    shutil.copytree(os.path.join(root, "tests", "golden", "se_q"), d)

    def run(*args):
        return subprocess.run([exe, d + "/ref", d + "/temp/s", d + "/stat/s"] + list(args) + ["-q", "--dry-run"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def choice(*args):
        r = run(*args)
        assert r.returncode == 0, r.stderr
        line = r.stdout.strip().split("\n")[-1]
        assert line.startswith("dry run: ")
        return line.split()[2], int(line.split("GPU group(s), ")[1].split()[0]), int(line.split("on ")[1].split()[0])

    assert choice("20", "40", "1", "-p", "2") == ("exact", 1, 1)
    assert choice("200", "1000", "1", "-p", "64") == ("exact", 1, 1)
    assert choice("20000", "1000", "1", "-p", "8", "--gibbs-mode", "exact") == ("exact", 1, 1)
    assert choice("200", "1000", "1", "-p", "8", "--gibbs-mode", "parallel") == ("parallel", 8, 1)
    assert choice("200", "1000", "1", "-p", "8", "--gibbs-mode", "parallel", "--gibbs-thin", "3") == ("parallel", 3, 1)
    assert choice("200", "1000", "1", "-p", "8", "--devices", "0,0") == ("exact", 1, 2)
    assert choice("200", "1000", "1", "-p", "1", "--devices", "0,0") == ("exact", 1, 1)  # never more groups than chains
    r = run("200", "1000", "1", "-p", "8", "--gibbs-mode", "parallel")
    assert "data-augmentation" in r.stderr
    r = run("200", "1000", "1", "-p", "8")
    assert "data-augmentation" not in r.stderr
    r = run("200", "1000", "1", "-p", "8", "--gibbs-mode", "Exact")
    assert r.returncode != 0 and "unknown --gibbs-mode" in r.stderr


def test_run_em_rejects_inconsistent_alignment_coordinates(tmp_path):
    """getConPrb's coordinate assertions (SingleQModel.h:114-120, PairedEndQModel.h:109-115): an alignment that hangs over
    the end of its transcript, or starts before it, stops rsem-run-em with the reference's message -- before any device
    work (the kernels index reference sequences with these coordinates), so the check is visible without a GPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rsem_amd", "bin", "rsem-run-em")

    def run(name, read_type, edit):
        d = str(tmp_path / (name + "_" + edit.__name__))
        shutil.copytree(os.path.join(root, "tests", "golden", name), d)
        dat = os.path.join(d, "temp", "s.dat")
        lines = open(dat).read().split("\n")
        lines[1] = edit(lines[1].split())
        open(dat, "w").write("\n".join(lines))
        return subprocess.run([exe, d + "/ref", str(read_type), d + "/s", d + "/temp/s", d + "/stat/s", "-q"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True)

    def hang_over(f):       # first alignment of the first read: push it past the end of the transcript
        f[2] = "1000000"
        return " ".join(f)

    def negative_start(f):  # reverse-strand coordinate that maps before position 0
        f[1] = "-" + f[1].lstrip("-")
        f[2] = "1000000"
        return " ".join(f)

    r = run("se_q", 1, hang_over)
    assert r.returncode != 0 and ("is hung over the end of transcript" in r.stderr or "starts at" in r.stderr), r.stderr
    assert "different read lengths" in r.stderr
    r = run("se_q", 1, negative_start)
    assert r.returncode != 0 and "starts at" in r.stderr and "non-negative" in r.stderr, r.stderr
    r = run("pe_q", 3, hang_over)
    assert r.returncode != 0 and "ragment" in r.stderr, r.stderr
    # untouched input passes the check (and then stops for the missing GPU here, or runs on a GPU box)
    d = str(tmp_path / "ok")
    shutil.copytree(os.path.join(root, "tests", "golden", "se_q"), d)
    r = subprocess.run([exe, d + "/ref", "1", d + "/s", d + "/temp/s", d + "/stat/s", "-q"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert "hung over" not in r.stderr and "starts at" not in r.stderr


def test_run_em_rejects_unknown_value_format():
    """--value-bits takes 64 (default: the doubles as given) or 32 (Q32 planes); anything else ends the program with the
    reference's error convention (message, status -1) before any file or device is touched."""
    exe = os.path.join(ROOT, "rsem_amd", "bin", "rsem-run-em")
    if not os.path.exists(exe):
        pytest.skip("programs not built")
    r = subprocess.run([exe, "ref", "1", "s", "imd", "stat", "--value-bits", "16"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 255 and "--value-bits must be 64 or 32" in r.stdout


def test_null_arguments_of_the_info_and_option_calls(lib):
    import ctypes as C
    v = C.c_int64()
    assert lib.rsem_em_get_info(None, b"units", C.byref(v)) == -1
    assert lib.rsem_em_set_option(None, b"value_bits", 32) == -1


def test_posterior_moments_from_sums(tmp_path):
    """host/posterior_moments.hpp (the end of rsem-run-gibbs: means and unbiased variances from the device's sums) against
    the formulas of Gibbs.cpp:389-423 in numpy, bit for bit, incl. the floor at zero."""
    exe = os.path.join(str(tmp_path), "pm_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "posterior_moments_check.cpp"), "-o", exe])
    rng = np.random.default_rng(3)
    n, M1, m = 1000, 41, 9
    samples = rng.poisson(rng.gamma(0.7, 30.0, M1), size=(n, M1)).astype(np.float64)
    samples[:, 5] = 7.0          # a constant column: variance exactly 0 or a negative rounding residue -> floored
    samples[:, 6] = 1e8 + 0.1    # cancellation
    starts = np.sort(np.concatenate([[1, M1], rng.choice(np.arange(2, M1), m - 1, replace=False)])).astype(np.int64)
    s1, s2 = samples.sum(0), (samples ** 2).sum(0)
    g2 = np.array([(samples[:, a:b].sum(1) ** 2).sum() for a, b in zip(starts[:-1], starts[1:])])
    text = "%d %d %d\n" % (n, M1, m) + "\n".join("%.17g" % v for v in np.concatenate([s1, s2])) + "\n" + " ".join(map(str, starts)) + "\n" + "\n".join("%.17g" % v for v in g2) + "\n"
    out = np.array(subprocess.run([exe], input=text, stdout=subprocess.PIPE, text=True, check=True).stdout.split(), np.float64)
    mean = s1 / n
    var = np.maximum((s2 - float(n) * mean * mean) / float(n - 1), 0.0)
    gmean = np.array([mean[a:b].sum() for a, b in zip(starts[:-1], starts[1:])])
    # (the group mean is summed left to right in the header; numpy's pairwise sum may differ in the last bit for long groups)
    gvar = np.maximum((g2 - float(n) * gmean * gmean) / float(n - 1), 0.0)
    assert np.array_equal(out[:M1], mean) and np.array_equal(out[M1:2 * M1], var)
    assert np.allclose(out[2 * M1:], gvar, rtol=1e-12, atol=1e-9) and out[M1 + 5] == 0.0 and np.all(out[M1:] >= 0.0)


def test_integration_binding_compiles_against_the_reference_headers():
    """INTEGRATION.md section B as code (tests/integration_sketch.cpp): HitContainer<SingleHit / PairedEndHit> flattened into
    the CSR of rsem_em_create, rsem_em_run, rsem_em_expected_weights -- compiled (syntax only) against the reference's own
    headers in the reference's own dialect (its Makefile: gnu++98), with include/rsem_hip.h as the only thing from here."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "HitContainer.h")):
        pytest.skip("the reference sources are not on this machine")
    r = subprocess.run(["g++", "-std=gnu++98", "-fsyntax-only", "-w", "-I" + ref, "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "integration_sketch.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_model_file_reader_and_writer_reproduce_the_reference_files(tmp_path):
    """host/model_host.hpp: read each golden s.model (written by the reference: SingleModel / SingleQModel / PairedEndModel /
    PairedEndQModel ::write with their parts) and write it again -> the same bytes (layout of every table, %.10g / %.15g)."""
    import filecmp
    exe = os.path.join(str(tmp_path), "model_rt")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "model_roundtrip.cpp"), "-o", exe, "-lz", "-lpthread"])
    gold = os.path.join(ROOT, "tests", "golden")
    n = 0
    for name in sorted(os.listdir(gold)):
        src = os.path.join(gold, name, "stat", "s.model")
        if not os.path.exists(src):
            continue
        out = os.path.join(str(tmp_path), name + ".model")
        subprocess.check_call([exe, src, out])
        assert filecmp.cmp(src, out, shallow=False), name
        n += 1
    assert n >= 9


def test_gibbs_binary_handoff_equals_the_text_file(tmp_path):
    """host/ofb.hpp (rsem-run-em --gibbs-out -> rsem-run-gibbs as arrays): the arrays are what a reader of imd.ofg gets --
    values through the 15-digit text form -- on every fixture; a newer text file wins over an older directory."""
    exe = os.path.join(str(tmp_path), "ofb_rt")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "ofb_roundtrip.cpp"), "-o", exe, "-lz", "-lpthread"])
    gold = os.path.join(ROOT, "tests", "golden")
    n = 0
    for name in sorted(os.listdir(gold)):
        src = os.path.join(gold, name, "temp", "s.ofg")
        if not os.path.exists(src):
            continue
        r = subprocess.run([exe, src, os.path.join(str(tmp_path), name)], stdout=subprocess.PIPE, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok"), (name, r.stdout)
        n += 1
    assert n >= 9
