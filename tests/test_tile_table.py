"""CPU: the mixed-class tile kernel's compact matrix table (x266_tables.hpp: TileTab) and the lane-level image construction the
kernel performs from it, restated in C++ (tests/cpp/tile_table_check.cpp), reproduce the per-class operand images of every
(type, size) class -- built-in matrices and random int8 ones.  g++ only: no GPU, no HIP."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tile_table_reproduces_the_per_class_images(tmp_path):
    exe = str(tmp_path / "tile_table_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "x266_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "tile_table_check.cpp")], check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "reproduce the per-class images" in out.stdout
