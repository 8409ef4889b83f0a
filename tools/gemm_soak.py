"""Soak test of one GEMM tile configuration: many launches over the path's shapes (with and without epilogue terms, a
background stream loading the chip), every result compared bit for bit with the first launch of its shape.  A counted-wait
or hand-over bug in a persistent kernel shows up here as a mismatch or a memory fault long before it shows in a step.
Usage: python tools/gemm_soak.py TILE [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd import ops  # noqa: E402

tile = int(sys.argv[1])
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
dev = torch.device("cuda:0")
SHAPES = [(8192, 8192, 8192, 0), (32768, 6144, 1536, 1), (65536, 1536, 6144, 2), (147456, 2048, 1024, 0), (11680, 8192, 1536, 1),
          (4616, 1024, 1024, 2), (1000, 2048, 1024, 1), (9216, 4096, 1024, 1), (2300, 768, 320, 2), (65536, 1536, 384, 1)]
g = torch.Generator().manual_seed(1)
cases = []
for M, N, K, mode in SHAPES:
    a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev, torch.bfloat16)
    kw = {}
    if mode >= 1:
        kw["bias"] = torch.randn(N, generator=g).to(dev)
        kw["act"] = 1
    if mode == 2:
        kw = {"residual": torch.randn(M, N, generator=g).to(dev, torch.bfloat16), "colscale": torch.randn(N, generator=g).to(dev)}
    first = ops.k_gemm(a, w, tile=tile, **kw)
    ref = ops.k_gemm(a, w, tile=128, **kw)
    assert ((first.float() - ref.float()).abs().max() / ref.float().abs().max()).item() < 1e-2, (M, N, K)
    cases.append((a, w, kw, first))
side = torch.randn(4096, 4096, device=dev)
s2 = torch.cuda.Stream()
t0, n = time.time(), 0
while time.time() - t0 < budget:
    for a, w, kw, first in cases:
        with torch.cuda.stream(s2):
            side = side @ side * 1e-4
        out = ops.k_gemm(a, w, tile=tile, **kw)
        if not torch.equal(out, first):
            raise SystemExit(f"MISMATCH tile {tile} shape {tuple(a.shape)} x {tuple(w.shape)} after {n} launches")
        n += 1
torch.cuda.synchronize()
print(f"tile {tile}: {n} launches, all bit-identical to the first of their shape")
