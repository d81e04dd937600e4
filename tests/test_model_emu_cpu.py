"""The group-per-read body of the model rounds' kernel (rsem_amd/csrc/model_block.hpp -- the file model.hip compiles for the
GPU: alignment probabilities, posterior weights and the model's statistics of a round in one pass over the reads) run on the
CPU by tests/model_emu.cpp: one OS thread per lane, the cross-lane intrinsics as exchanges through memory.  Checked against a
thread-per-alignment restatement of getConPrb / getNoiseConPrb / EM.cpp:199-244 / update written with the same scalar
helpers, on seeded synthetic data that holds the cases the lane mapping has to get right: reads with more than 16 alignments
(several chunks; runs of identical windows continuing across a chunk boundary), reads longer than 128 bases (two passes of the
positional split), low-quality reads, masked start positions, transcripts with mw = 0, all four model types, RSPD on / off,
the single-end model with a mate length distribution.  No GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(CC), reason="needs hipcc (host compilation of the HIP headers)")


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    exe = os.path.join(str(tmp_path_factory.mktemp("model_emu")), "model_emu")
    r = subprocess.run([CC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-DRSEM_EMU", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("RSEM_EMU_FLAGS", "").split() + [
                        os.path.join(ROOT, "tests", "model_emu.cpp"), "-o", exe, "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("model_type,seed,est_rspd,has_mld", [(3, 1, 0, 0), (3, 2, 1, 0), (1, 3, 0, 0), (1, 4, 1, 1), (0, 5, 1, 0), (2, 6, 0, 0), (0, 7, 0, 1)])
def test_group_kernel_body_equals_the_per_alignment_restatement(emulator, model_type, seed, est_rspd, has_mld):
    r = subprocess.run([emulator, str(model_type), str(seed), str(est_rspd), str(has_mld)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "MISMATCH" not in r.stdout


@pytest.fixture(scope="module")
def emulator_tsan(tmp_path_factory):
    exe = os.path.join(str(tmp_path_factory.mktemp("model_emu_tsan")), "model_emu_tsan")
    r = subprocess.run([CC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-DRSEM_EMU", "-fsanitize=thread", "-Wno-unused-result", "-Wno-unused-value",
                        os.path.join(ROOT, "tests", "model_emu.cpp"), "-o", exe, "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer build with this toolchain: " + r.stderr[-300:])
    return exe


@pytest.mark.skipif(not os.environ.get("RSEM_TSAN_ALL"), reason="a minute of CPU time: run with RSEM_TSAN_ALL=1 (all seven cases were run by hand: clean)")
@pytest.mark.parametrize("model_type,seed,est_rspd,has_mld", [(3, 2, 1, 0), (0, 7, 0, 1)])
def test_no_unordered_accesses_between_lanes(emulator_tsan, model_type, seed, est_rspd, has_mld, monkeypatch):
    """k_model_group's body under ThreadSanitizer (one OS thread per lane, pthread barriers for the kernel's barriers): an LDS or
    global access of two lanes that no barrier orders is reported and makes the emulator exit with 66."""
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=0 exitcode=66")
    test_group_kernel_body_equals_the_per_alignment_restatement(emulator_tsan, model_type, seed, est_rspd, has_mld)
