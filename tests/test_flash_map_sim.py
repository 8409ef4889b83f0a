"""CPU: enumeration of the flash attention kernels' block -> work mapping — tests/csrc/flash_map_sim.cpp compiled with g++
against cambrian_amd/csrc/flash_map.h (the very header the kernels include): exact coverage of every (batch, head, block)
for odd / even block counts, causal pairs, group-per-XCD placement and XCD load balance."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flash_block_mapping_enumeration():
    src = os.path.join(ROOT, "tests", "csrc", "flash_map_sim.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sim")
        subprocess.run(["g++", "-O1", "-std=c++17", src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
