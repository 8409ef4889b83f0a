"""CPU (hipcc cross-compiles without a GPU): the tile loops of the LDS-DMA attention kernels (cambrian_amd/csrc/flash2.hip) carry
no compiler-inserted ``s_waitcnt vmcnt``.

The fills of the next K / V tile are ``global_load_lds`` instructions inside inline asm: the compiler's wait-count bookkeeping does
not see them.  A wait it inserts inside the loop for an ordinary load of the PROLOGUE (the Q / dO fragments) therefore carries a
count that is too small by the fills in flight and awaits them in the middle of the tile — the prefetch then runs serialised with
the products (round 5: forward -3..5 % once found, profiles/r05_lab.md).  The kernels consume the prologue's registers in front of
the loop (``asm volatile("" : "+v"(...))``); this test keeps it that way: in every basic block of ``flash_fwd2_kernel`` /
``flash_dq2_kernel`` that holds eight or more MFMAs, each ``vmcnt`` wait sits between ``#ASMSTART`` and ``#ASMEND`` (the
kernel's own, at the top of a tile).  Key-padding (MASKED) instantiations load the next tile's validity bytes inside the loop and
are exempt."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cambrian_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_compiler_vmcnt_waits_inside_the_lds_dma_tile_loops(tmp_path):
    out = tmp_path / "flash2.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-inline-asm", "-S",
           "--cuda-device-only", os.path.join(CSRC, "flash2.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=CSRC, capture_output=True, timeout=900)
    txt = out.read_text()
    checked = 0
    for m in re.finditer(r"^(_Z\S*flash_(?:fwd2|dq2)_kernelILb[01]ELb0E\S*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        blocks, cur = [], []
        for line in body.split("\n"):
            if line.startswith(".LBB") or line.startswith("; %bb."):
                blocks.append(cur)
                cur = []
            else:
                cur.append(line)
        blocks.append(cur)
        hot = [b for b in blocks if sum("v_mfma" in ln for ln in b) >= 8]
        assert hot, name
        for b in hot:
            inside = False
            for ln in b:
                if "#ASMSTART" in ln:
                    inside = True
                elif "#ASMEND" in ln:
                    inside = False
                elif "vmcnt" in ln:
                    assert inside, f"{name}: compiler-inserted wait inside an MFMA block of the tile loop: {ln.strip()}"
        checked += 1
    assert checked == 4, checked   # {fwd2, dq2} x {causal, bidirectional}, unmasked
