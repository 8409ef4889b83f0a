"""Launch ONE GEMM shape a few times (for rocprofv3 --pmc passes: a small, known dispatch list).
Usage: python tools/bench_one_gemm.py M N K [tile] [iters] [act]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd import ops, lib as L  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
tile = (sys.argv[4] if sys.argv[4] == "blas" else int(sys.argv[4])) if len(sys.argv) > 4 else 256
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
act = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev) if act else None   # (an activation GEMM of the path always carries its bias)
def run():
    if tile == "blas":
        torch.matmul(a, w.T, out=out)
    else:
        ops.k_gemm(a, w, out=out, tile=tile, act=act, bias=bias)


for _ in range(iters):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
print(f"M={M} N={N} K={K} tile={tile} {t * 1e6:.1f} us {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s")
