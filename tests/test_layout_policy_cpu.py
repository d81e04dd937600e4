"""Host-only check of the index helpers and the lanes-per-read policy table of rsem_amd/csrc/sell_layout.hpp as built with
-DRSEM_GENERAL_G=1 (a prepared variant, off in the product build): tests/layout_policy_check.cpp calls the header's own
__host__ functions -- no device needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_general_g_policy_and_index_helpers(tmp_path):
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(str(tmp_path), "layout_policy_check")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-DRSEM_GENERAL_G=1", "-Wno-unused-result", "-Wno-unused-value",
                           os.path.join(ROOT, "tests", "layout_policy_check.cpp"), "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    last = r.stdout.strip().split("\n")[-1]
    assert "bad=0" in last
    p2, mb = float(last.split("pow2")[1].split(",")[0]), float(last.split("min-bytes")[1].split(";")[0])
    assert mb < p2
