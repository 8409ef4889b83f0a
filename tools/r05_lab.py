#!/usr/bin/env python
"""Round-5 kernel lab (GPU): every new variant against the kernel it replaces, in one process, interleaved, with a
correctness check.  `python tools/r05_lab.py flash [--B 8]` ; `... gemm` ; `... lnmulti` ; `... all`.
Prints one JSON object per case (also appended to gpurun_out/r05/lab.jsonl)."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cambrian_amd import lib as L  # noqa: E402
from cambrian_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def emit(obj):
    print(json.dumps(obj), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r05"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05", "lab.jsonl"), "a") as f:
        f.write(json.dumps(obj) + "\n")


def time_variants(variants: dict, iters=10, rounds=3):
    """{name: fn} -> {name: median us}; variants interleaved round-robin, an event pair per call."""
    for f in variants.values():
        f()
    torch.cuda.synchronize()
    evs = {k: [] for k in variants}
    for _ in range(rounds):
        for k, f in variants.items():
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                f()
                e1.record()
                evs[k].append((e0, e1))
    torch.cuda.synchronize()
    out = {}
    for k, lst in evs.items():
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in lst)
        out[k] = round(us[len(us) // 2], 1)
    return out


def flash_case(B, S=2048, H=32, HKV=8, masked=False, causal=True):
    D = 128
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn((B, S, H, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2)
    k = torch.randn((B, S, HKV, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2)
    v = torch.randn((B, S, HKV, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2)
    do = torch.randn((B, S, H, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2)
    kvalid = None
    if masked:
        kvalid = torch.ones((B, S), dtype=torch.uint8, device=dev)
        for b in range(B):
            lo = 91 + 25 * ((b * 5) % 20)
            kvalid[b, lo:lo + 100 + 25 * (b % 4)] = 0
            kvalid[b, S - 64 * (b % 5):] = 0
    kv_len = None if causal else S - 37
    res = {}

    def run(knob):
        L.knob_set(L.KNOB_FLASH, knob)
        qq, kk, vv = (t.detach().clone().requires_grad_() for t in (q, k, v))
        o = ops.CausalAttnFn.apply(qq, kk, vv, None, kv_len, kvalid)
        o.backward(do)
        return o.detach(), qq.grad, kk.grad, vv.grad

    a, b_ = run(0), run(23)
    c_ = run(22)   # round-4 forward + the round-5 backward kernels: dq / dk / dv must be bit-identical to knob 0
    names = ["out", "dq", "dk", "dv"]
    eq = {n: bool(torch.equal(x, y)) for n, x, y in zip(names, a, b_)}
    err = {n: float((x.float() - y.float()).abs().max() / x.float().abs().max()) for n, x, y in zip(names, a, b_)}
    eq_bwd = {n: bool(torch.equal(x, y)) for n, x, y in zip(names, a, c_)}
    # timing: forward and backward separately
    qq, kk, vv = (t.detach().clone().requires_grad_() for t in (q, k, v))

    def fwd(knob):
        def f():
            L.knob_set(L.KNOB_FLASH, knob)
            with torch.no_grad():
                ops.CausalAttnFn.apply(qq, kk, vv, None, kv_len, kvalid)
        return f

    outs = {}
    for knob in (0, 6, 22):
        L.knob_set(L.KNOB_FLASH, knob)
        outs[knob] = ops.CausalAttnFn.apply(qq, kk, vv, None, kv_len, kvalid)

    def bwd(knob):
        def f():
            L.knob_set(L.KNOB_FLASH, knob)
            torch.autograd.grad(outs[knob], (qq, kk, vv), do, retain_graph=True)
        return f

    t = time_variants({"fwd_old": fwd(0), "fwd_new": fwd(1), "bwd_old": bwd(0), "bwd_dq2_dkdvpipe": bwd(6),
                       "bwd_new": bwd(22)}, iters=5, rounds=3)
    L.knob_set(L.KNOB_FLASH, 7)
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    emit({"case": "flash", "B": B, "S": S, "H": H, "HKV": HKV, "causal": causal, "masked": masked, "bit_equal_all_new": eq, "bit_equal_bwd_only_new": eq_bwd, "max_rel": err,
          "us": t, "fwd_TFLOPs": {k_: round(fl / t[k_] / 1e6, 1) for k_ in ("fwd_old", "fwd_new")},
          "bwd_TFLOPs_effective_2.5x": {k_: round(2.5 * fl / t[k_] / 1e6, 1) for k_ in ("bwd_old", "bwd_new")}})


def flash_only(B, knob, iters=3, what="fwd"):
    """a few launches of one flash kernel variant (the workload of the rocprofv3 --pmc passes: tools/gpu_r05.sh pmcflash)"""
    D, S, H, HKV = 128, 2048, 32, 8
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn((B, S, H, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2).requires_grad_()
    k = torch.randn((B, S, HKV, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2).requires_grad_()
    v = torch.randn((B, S, HKV, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2).requires_grad_()
    do = torch.randn((B, S, H, D), device=dev, dtype=torch.bfloat16, generator=g).transpose(1, 2)
    L.knob_set(L.KNOB_FLASH, knob)
    for _ in range(iters):
        o = ops.CausalAttnFn.apply(q, k, v, None, None, None)
        if what == "bwd":
            torch.autograd.grad(o, (q, k, v), do)
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="+")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--knob", type=int, default=0)
    a = ap.parse_args()
    if "flashonly" in a.what:
        flash_only(a.B, a.knob, what="fwd")
    if "flashbwdonly" in a.what:
        flash_only(a.B, a.knob, what="bwd")
    if "flash" in a.what or "all" in a.what:
        flash_case(a.B)
        flash_case(a.B, masked=True)
        flash_case(4, S=768, H=16, HKV=16, causal=False)
        flash_case(24)


if __name__ == "__main__":
    main()
