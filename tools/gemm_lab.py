"""GEMM variant lab (one MI355X): every tile configuration of cmb_gemm against hipBLASLt on the path's hot shapes,
interleaved rounds in one process (cdna_hip_programming.md §5.4 rule 24), random operands (rule 25).
Usage: python tools/gemm_lab.py [--rounds R] [--quick]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd import ops, lib as L  # noqa: E402

dev = torch.device("cuda:0")
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--variants", default="128,2560,2590,blas")
ap.add_argument("--shapes", default="")
args = ap.parse_args()

# (M, N, K, act, bias)  — the in-step shapes of bench.py at 16 img/GPU, by time (gpurun_out/gemm_shapes_b16.json)
SHAPES = [
    (65536, 6144, 1536, 1), (65536, 1536, 6144, 0), (11680, 8192, 1536, 0), (147456, 2048, 1024, 0),
    (1048576, 1536, 384, 1), (262144, 3072, 768, 1), (11680, 4608, 1536, 0), (147456, 1024, 2048, 0),
    (1048576, 384, 1536, 0), (11664, 4352, 1152, 1), (262144, 768, 3072, 0), (16384, 12288, 3072, 1),
    (147456, 1024, 5760, 1), (4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (32768, 6144, 1536, 0),
]
if args.quick:
    SHAPES = SHAPES[:4] + SHAPES[13:14]
variants = args.variants.split(",")
if args.shapes:
    SHAPES = [tuple(int(x) for x in t.split("x")) for t in args.shapes.split(",")]


def run(v, a, w, out, bias, act):
    if v == "blas":
        torch.matmul(a, w.T, out=out)       # no epilogue: the bare vendor GEMM as the bar
    else:
        ops.k_gemm(a, w, out=out, bias=bias, act=act, tile=int(v))


for M, N, K, act in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if act else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    # correctness of every variant against the 128 tile on a row sample
    ref = None
    errs = {}
    for v in variants:
        if v == "blas":
            continue
        out.zero_()
        run(v, a, w, out, bias, act)
        torch.cuda.synchronize()
        sample = torch.cat([out[:300], out[M // 2:M // 2 + 300], out[-300:]]).float()
        if ref is None:
            ref = sample
        errs[v] = ((sample - ref).abs().max() / ref.abs().max()).item()
    times = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            run(v, a, w, out, bias, act)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run(v, a, w, out, bias, act)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.iters * 1e-3)
    fl = 2.0 * M * N * K
    rec = {"M": M, "N": N, "K": K, "act": act}
    for v in variants:
        ts = sorted(times[v])
        rec[f"tf_{v}"] = round(fl / ts[len(ts) // 2] / 1e12, 1)
        rec[f"tfmax_{v}"] = round(fl / ts[0] / 1e12, 1)
    rec["err_vs_first"] = {k: float(f"{e:.2e}") for k, e in errs.items()}
    print(json.dumps(rec), flush=True)
    del a, w, out
