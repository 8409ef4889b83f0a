"""CPU: host-side simulations of the LDS images the round-3 kernels read with ds_read_b64_tr_b16 — gemm_tn.hip's operand
stage (tests/csrc/gemm_tn_layout_sim.cpp against cambrian_amd/csrc/gemm_tn_layout.h) and sva_absorbed.hip's token window /
operand stage (tests/csrc/sva_abs_layout_sim.cpp against sva_abs_layout.h) and the round-5 flash kernels' operand tile
(tests/csrc/flash_layout_sim.cpp against flash_layout.h), the very headers the kernels include: every
lane receives the (column, contraction rows) its MFMA operand slot stands for, the contraction is complete, and the reads
spread over the banks as the kernels' comments claim."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["gemm_tn_layout_sim", "sva_abs_layout_sim", "flash_layout_sim", "vit_layout_sim"])
def test_layout_simulation(name):
    src = os.path.join(ROOT, "tests", "csrc", name + ".cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sim")
        subprocess.run(["g++", "-O1", "-std=c++17", src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
