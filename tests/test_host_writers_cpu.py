"""Host-side writers of the programs (csrc/host/files.hpp): a row of 200 k numbers is formatted in pieces on the host's threads
(write_cells_line); the bytes must be those of the plain fprintf loop the reference's writers are (EM.cpp:484-500, WriteResults.h)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = shutil.which("g++")


@pytest.mark.skipif(CXX is None, reason="needs g++")
@pytest.mark.parametrize("n", [7, 49999, 50000, 200001])
def test_long_rows_are_the_same_bytes_as_a_printf_loop(n, tmp_path):
    exe = os.path.join(str(tmp_path), "write_cells_check")
    subprocess.check_call([CXX, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "write_cells_check.cpp"), "-o", exe, "-lpthread"])
    a, b = os.path.join(str(tmp_path), "a.txt"), os.path.join(str(tmp_path), "b.txt")
    subprocess.check_call([exe, str(n), a, b])
    assert open(a, "rb").read() == open(b, "rb").read()
