#!/usr/bin/env python3
"""Build-time guard for gemm_p5.hip.  gemm_nt_p5_kernel issues its LDS-DMA pieces as
    s_add_u32 m0, ...; s_nop 0; global_load_lds_dwordx4 v, s[base:base+1]
i.e. with the one wait state M0 needs and none for the SGPR base.  That is only safe while the base was not written by
a VALU instruction (v_readlane_b32 of an SGPR spill slot, v_readfirstlane_b32) in the 5 wait states before the DMA:
hipcc pads its own instructions for that hazard, not inline asm.  This script walks the kernel's assembly backwards
from every piece and fails the build when such a write (or a label it cannot see across) is closer than 5 wait states.

    python3 check_spills.py build/gemm_p5-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

NEED = 5


def wait_states(ins: str) -> int:
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def main(path: str) -> int:
    lines = open(path).read().split("\n")
    bad, pieces, kernels = [], 0, 0
    i = 0
    while i < len(lines):
        m = re.match(r"(_ZN\S*gemm_nt_p5_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        kernels += 1
        name = m.group(1)
        body = []
        i += 1
        while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
            t = lines[i].split(";")[0].strip()
            if t and not t.startswith("."):
                body.append(t)          # instruction
            elif re.match(r"\.LBB\S+:", t):
                body.append("LABEL")
            i += 1
        for k, ins in enumerate(body):
            g = re.match(r"global_load_lds_dwordx4\s+v\d+,\s*s\[(\d+):(\d+)\]", ins)
            if not g:
                continue
            pieces += 1
            regs = {int(g.group(1)), int(g.group(2))}
            ws, j = 0, k - 1
            while j >= 0 and ws < NEED:
                p = body[j]
                if p == "LABEL":
                    bad.append((name, k, "label %d wait states before the piece" % ws))
                    break
                w = re.match(r"v_(?:readlane|readfirstlane)_b32\s+s(\d+)", p)
                if w and int(w.group(1)) in regs:
                    bad.append((name, k, "%s only %d wait states before the piece" % (p, ws)))
                    break
                ws += wait_states(p)
                j -= 1
    if not kernels or not pieces:
        print("check_spills: no gemm_nt_p5_kernel pieces found in", path)
        return 1
    for b in bad[:20]:
        print("check_spills: %s instruction %d: %s" % b)
    print("check_spills: %d kernels, %d LDS-DMA pieces, %d violations" % (kernels, pieces, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
