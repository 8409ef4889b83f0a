"""CPU: host-side simulation of the GEMM's LDS-DMA staging, swizzled fragment reads (bank-conflict freedom
under the gfx950 ds_read_b128 lane grouping), accumulator map and XCD remap — tests/csrc/gemm_layout_sim.cpp
compiled with g++ against cambrian_amd/csrc/gemm_layout.h (the very header the kernel includes)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_layout_simulation():
    src = os.path.join(ROOT, "tests", "csrc", "gemm_layout_sim.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sim")
        subprocess.run(["g++", "-O1", "-std=c++17", src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
