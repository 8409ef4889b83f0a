"""fp8 vs bf16 projection GEMM on the release shapes (B = 16): aux projector of the ConvNeXt tower
(M = 16*9216, N = 1024, K = 5760), SVA K/V projection (M = 16*9216, N = 1024, K = 1024), mm_projector_aux of DINOv2
(M = 16*576 ...).  Prints one JSON line per shape: quantise time, fp8 GEMM time / TFLOP/s, bf16 GEMM time / TFLOP/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd import ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    for M, N, K in [(147456, 1024, 5760), (147456, 1024, 1024), (147456, 2048, 1024), (9216, 1024, 1536), (9216, 4096, 1024)]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        xq, xi = ops.k_quantize_fp8_rows(x)
        wq, wi = ops.k_quantize_fp8_rows(w)
        t_q = timed(lambda: ops.k_quantize_fp8_rows(x))
        t8 = timed(lambda: ops.k_gemm_fp8(xq, xi, wq, wi))
        t16 = timed(lambda: ops.k_gemm(x, w))
        t16_128 = timed(lambda: ops.k_gemm(x, w, tile=128))
        fl = 2.0 * M * N * K
        print(json.dumps({"M": M, "N": N, "K": K, "quantize_x_ms": round(t_q, 4),
                          "quantize_GBps": round((M * K * 3 + M * 4) / t_q / 1e6, 1),
                          "fp8_ms": round(t8, 4), "fp8_TFLOPs": round(fl / t8 / 1e9, 1),
                          "fp8_plus_quant_TFLOPs": round(fl / (t8 + t_q) / 1e9, 1),
                          "bf16_ms": round(t16, 4), "bf16_TFLOPs": round(fl / t16 / 1e9, 1),
                          "bf16_tile128_TFLOPs": round(fl / t16_128 / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
