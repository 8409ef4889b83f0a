"""Microbenchmarks of the individual kernels on one MI355X (HIP events on the launch stream).
Usage: python tools/bench_kernels.py [gemm|ln|sva|attn|all]"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd import ops, lib as L  # noqa: E402
from cambrian_amd.model.multimodal_encoder import vit_ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    shapes = [(8 * 10944, 2048, 1024), (8 * 9216, 1024, 5760), (8 * 576, 4096, 1024), (8 * 577, 4096, 1024),
              (8 * 730, 8192, 1536), (4096, 4096, 4096), (8192, 8192, 8192), (8 * 65536, 1536, 384),
              (8 * 4096, 6144, 1536), (8 * 4096, 1536, 6144)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.k_gemm(a, w, out=out))
        t128 = timeit(lambda: ops.k_gemm(a, w, out=out, tile=128))
        t256 = timeit(lambda: ops.k_gemm(a, w, out=out, tile=2560))
        t256b = timeit(lambda: ops.k_gemm(a, w, out=out, tile=2561))
        t_ref = timeit(lambda: torch.matmul(a, w.T, out=out))
        fl = 2.0 * M * N * K
        print(json.dumps({"kernel": "gemm_bf16", "M": M, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": fl / t / 1e12,
                          "tile128_TFLOPs": fl / t128 / 1e12, "tile256s0_TFLOPs": fl / t256 / 1e12,
                          "tile256s1_TFLOPs": fl / t256b / 1e12, "hipblaslt_TFLOPs": fl / t_ref / 1e12}), flush=True)
    for M, N, K in [(8 * 4096, 6144, 1536), (8 * 65536, 1536, 384)]:  # ConvNeXt fc1: bias + exact-erf GELU epilogue
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        r = {}
        for tile in (128, 2560, 2561):
            r[tile] = timeit(lambda: ops.k_gemm(a, w, out=out, bias=b, act=L.ACT_GELU_ERF, tile=tile))
        fl = 2.0 * M * N * K
        print(json.dumps({"kernel": "gemm_bf16_bias_gelu", "M": M, "N": N, "K": K,
                          "tile128_TFLOPs": fl / r[128] / 1e12, "tile256s0_TFLOPs": fl / r[2560] / 1e12,
                          "tile256s1_TFLOPs": fl / r[2561] / 1e12}), flush=True)
    a = torch.randn(4096, 4096, device=dev)
    w = torch.randn(4096, 4096, device=dev)
    t = timeit(lambda: ops.k_gemm(a, w), iters=5)
    print(json.dumps({"kernel": "gemm_fp32", "M": 4096, "N": 4096, "K": 4096, "ms": t * 1e3,
                      "TFLOPs": 2 * 4096 ** 3 / t / 1e12}), flush=True)


def bench_ln():
    rows, D = 8 * 9216, 1024
    x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
    add = torch.randn(16, D, device=dev)
    t = timeit(lambda: ops.k_layernorm_fwd(x, None, None, 1e-5, add=add, side=96, grid_r=4))
    print(json.dumps({"kernel": "sva_norm_fwd", "rows": rows, "D": D, "ms": t * 1e3,
                      "GBps": rows * D * 4 / t / 1e9}), flush=True)
    y, mean, rstd = ops.k_layernorm_fwd(x, None, None, 1e-5, add=add, side=96, grid_r=4)
    acc = torch.zeros(rows, D, device=dev)
    t = timeit(lambda: ops.k_layernorm_bwd(y, x, mean, rstd, add=add, side=96, grid_r=4, dx_acc=acc))
    print(json.dumps({"kernel": "sva_norm_bwd_acc", "rows": rows, "D": D, "ms": t * 1e3,
                      "GBps": rows * D * (2 + 2 + 8) / t / 1e9}), flush=True)
    x = torch.randn(8 * 2048, 4096, device=dev).to(torch.bfloat16)
    w = torch.ones(4096, device=dev)
    t = timeit(lambda: ops.rmsnorm(x, w, 1e-5))
    print(json.dumps({"kernel": "rmsnorm_fwd", "rows": x.shape[0], "D": 4096, "ms": t * 1e3,
                      "GBps": x.numel() * 4 / t / 1e9}), flush=True)
    tr = torch.randn(8 * 10944, 2048, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.k_transpose(tr, tr.shape[0]))
    print(json.dumps({"kernel": "transpose", "R": tr.shape[0], "C": 2048, "ms": t * 1e3,
                      "GBps": tr.numel() * 4 / t / 1e9}), flush=True)


def bench_sva():
    B, qside, r_list, C = 8, 24, [1, 1, 1, 4], 1024
    q = torch.randn(B * 576, C, device=dev).to(torch.bfloat16)
    kvs = [torch.randn(B * (qside * r) ** 2, 2 * C, device=dev).to(torch.bfloat16) for r in r_list]
    t = timeit(lambda: ops.k_sva_attn_fwd(q, kvs, None, r_list, B, qside, 16, 64))
    byts = (sum(k.numel() for k in kvs) + 2 * q.numel()) * 2
    print(json.dumps({"kernel": "sva_attn_fwd", "B": B, "ms": t * 1e3, "GBps": byts / t / 1e9}), flush=True)
    out, lse = ops.k_sva_attn_fwd(q, kvs, None, r_list, B, qside, 16, 64)
    t = timeit(lambda: ops.k_sva_attn_bwd(out, q, kvs, None, r_list, out, lse, B, qside, 16, 64))
    byts = (2 * sum(k.numel() for k in kvs) + 4 * q.numel()) * 2
    print(json.dumps({"kernel": "sva_attn_bwd", "B": B, "ms": t * 1e3, "GBps": byts / t / 1e9}), flush=True)


def bench_attn():
    for N, heads, hd in [(577, 16, 64), (730, 24, 64), (729, 16, 96)]:
        B = 8
        qkv = torch.randn(B * N, 3 * heads * hd, device=dev).to(torch.bfloat16)
        t = timeit(lambda: vit_ops.k_vit_attn(qkv, B, N, heads, hd, 1 / math.sqrt(hd)))
        fl = 4.0 * B * heads * N * N * hd
        q, k, v = [x.view(B, N, heads, hd).transpose(1, 2) for x in qkv.chunk(3, -1)]
        t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
        print(json.dumps({"kernel": "vit_attn", "N": N, "heads": heads, "hd": hd, "ms": t * 1e3,
                          "TFLOPs": fl / t / 1e12, "torch_sdpa_TFLOPs": fl / t_ref / 1e12}), flush=True)
    x = torch.randn(8, 64, 64, 1536, device=dev).to(torch.bfloat16)
    w, b = torch.randn(49, 1536, device=dev), torch.randn(1536, device=dev)
    t = timeit(lambda: vit_ops.k_dwconv7x7(x, w, b))
    print(json.dumps({"kernel": "dwconv7x7", "shape": list(x.shape), "ms": t * 1e3,
                      "GFLOPs": x.numel() * 98 / t / 1e9, "GBps": x.numel() * 4 / t / 1e9}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for name, fn in [("gemm", bench_gemm), ("ln", bench_ln), ("sva", bench_sva), ("attn", bench_attn)]:
        if which in (name, "all"):
            try:
                fn()
            except Exception as e:  # keep going: one failing kernel must not hide the others' numbers
                print(json.dumps({"kernel": name, "error": repr(e)}), flush=True)
