"""Round-4 kernel lab: every cmb_knob_set candidate against the kernel it would replace, same process, same buffers, at the
24-image release shapes (the driver's batch).  Prints one JSON line per (kernel, shape, variant) and checks that the
variants agree (bit-exact where they are meant to be).

    python tools/r04_lab.py [--only ln,dw,gelu,attn,flash] [--iters 20] [--out gpurun_out/r04_lab.jsonl]
    CAMBRIAN_AMD_LIB=cambrian_amd/csrc/libcambrian_amd_novf.so python tools/r04_lab.py --only attn,flash --tag novf
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cambrian_amd import lib as L  # noqa: E402
from cambrian_amd import ops  # noqa: E402
from cambrian_amd.model.multimodal_encoder import vit_ops as V  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="ln,dw,gelu,attn,g128")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--images", type=int, default=24)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_lab.jsonl"))
ap.add_argument("--tag", default="")
args = ap.parse_args()
B = args.images
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32
os.makedirs(os.path.dirname(args.out), exist_ok=True)
fout = open(args.out, "a")


def rn(*shape, dtype=bf, scale=1.0):
    return (torch.randn(*shape, device=dev, dtype=torch.float32) * scale).to(dtype)


def timeit(fn, iters=None):
    iters = iters or args.iters
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def emit(**kw):
    kw["tag"] = args.tag
    line = json.dumps(kw)
    print(line, flush=True)
    fout.write(line + "\n")
    fout.flush()


only = set(args.only.split(","))

if "ln" in only:
    shapes = [("ConvNeXt s3", B * 4096, 1536), ("ConvNeXt s1", B * 65536, 384), ("ConvNeXt s2", B * 16384, 768),
              ("ConvNeXt s4", B * 1024, 3072), ("DINOv2", B * 730, 1536), ("CLIP", B * 577, 1024), ("SigLIP", B * 729, 1152)]
    for name, rows, D in shapes:
        x, g, b = rn(rows, D), rn(D, dtype=f32), rn(D, dtype=f32)
        ref = None
        for var in (0, 1):
            L.knob_set(L.KNOB_LN_FWD, var)
            y, _, _ = ops.k_layernorm_fwd(x, g, b, 1e-6, want_stats=False)
            if ref is None:
                ref = y
            same = bool(torch.equal(y, ref))
            us = timeit(lambda: ops.k_layernorm_fwd(x, g, b, 1e-6, want_stats=False))
            nbytes = rows * D * 4
            emit(kernel="layernorm_fwd", shape=f"{name} {rows}x{D}", variant=var, us=round(us, 1),
                 tbps=round(nbytes / us / 1e6, 2), bit_equal_v0=same)
        L.knob_set(L.KNOB_LN_FWD, 1)
        del x, ref, y

if "dw" in only:
    for name, side, C in (("s1", 256, 384), ("s2", 128, 768), ("s3", 64, 1536), ("s4", 32, 3072)):
        x = rn(B, side, side, C)
        w, b = rn(49, C, dtype=f32, scale=0.2), rn(C, dtype=f32)
        ref = None
        for var in (0, 1, 32, 64):
            L.knob_set(L.KNOB_DWCONV, var)
            y = V.k_dwconv7x7(x, w, b)
            if ref is None:
                ref = y
            same = bool(torch.equal(y, ref))
            us = timeit(lambda: V.k_dwconv7x7(x, w, b))
            n_out = B * side * side * C
            emit(kernel="dwconv7x7", shape=f"{name} {B}x{side}^2x{C}", variant=var, us=round(us, 1),
                 tbps=round(n_out * 4 / us / 1e6, 2), gflops=round(n_out * 98 / us / 1e3, 0), bit_equal_v0=same)
        L.knob_set(L.KNOB_DWCONV, 1)
        del x, ref, y

if "gelu" in only:
    for name, M, N, K, act in (("ConvNeXt s3 fc1", B * 4096, 6144, 1536, "gelu_erf"), ("ConvNeXt s3 fc1 plain", B * 4096, 6144, 1536, "none"),
                               ("ConvNeXt s3 fc2 plain", B * 4096, 1536, 6144, "none"), ("ConvNeXt s2 fc1", B * 16384, 3072, 768, "gelu_erf"),
                               ("aux fc1", B * 9216, 1024, 3072, "gelu_erf")):
        a, w, bias = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, dtype=f32)
        out = torch.empty(M, N, device=dev, dtype=bf)
        ops.k_gemm(a, w, bias=bias, act=L.ACT_CODES[act], out=out, tile=2590)
        us = timeit(lambda: ops.k_gemm(a, w, bias=bias, act=L.ACT_CODES[act], out=out, tile=2590), iters=10)
        emit(kernel="gemm_p5", shape=f"{name} {M}x{N}x{K} {act}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))
        if act != "none":
            o8 = ops.k_gemm(a[:8192], w, bias=bias, act=L.ACT_CODES[act], tile=2560)
            emit(kernel="gemm_p5", shape=name, check="p5 == 8-wave kernel, bit for bit (first 8192 rows)",
                 equal=bool(torch.equal(o8, out[:8192])))
        if act != "none":
            want = torch.nn.functional.gelu(a[:4096].float() @ w.float().T + bias)
            err = float((out[:4096].float() - want).abs().max() / want.abs().max())
            emit(kernel="gemm_p5", shape=f"{name}", check="v1 vs fp32 torch gelu (first 4096 rows), max-abs / max", rel=err)
        del a, w, out

if "attn" in only:
    for name, N, heads, hd in (("CLIP", 577, 16, 64), ("DINOv2", 730, 24, 64), ("SigLIP", 729, 16, 96)):
        qkv = rn(B * N, 3 * heads * hd)
        ref = None
        for var in (0, 1):
            L.knob_set(L.KNOB_VIT_ATTN, var)
            o = V.k_vit_attn(qkv, B, N, heads, hd, hd ** -0.5)
            if ref is None:
                ref = o
            diff = float((o.float() - ref.float()).abs().max())
            us = timeit(lambda: V.k_vit_attn(qkv, B, N, heads, hd, hd ** -0.5))
            emit(kernel="vit_attn", shape=f"{name} {B}x{heads}x{N}x{hd}", variant=var, us=round(us, 1),
                 tflops=round(4.0 * B * heads * N * N * hd / us / 1e6, 1), max_abs_diff_v0=diff)
        if hd == 64:
            want = V.k_vit_attn(qkv[: 2 * N].float().contiguous(), 2, N, heads, hd, hd ** -0.5)
            emit(kernel="vit_attn", shape=name, check="variant 1 vs fp32 simple kernel (2 images)",
                 rel=float((o[: 2 * N].float() - want).abs().max() / want.abs().max()))
        L.knob_set(L.KNOB_VIT_ATTN, 1)

if "flash" in only:
    Bq, S, nh, nkv, hd = 8, 2048, 32, 8, 128
    q = rn(Bq, nh, S, hd).requires_grad_(True)
    k = rn(Bq, nkv, S, hd).requires_grad_(True)
    v = rn(Bq, nkv, S, hd).requires_grad_(True)
    o = ops.causal_attention(q, k, v)
    g = rn(*o.shape)
    us_f = timeit(lambda: ops.causal_attention(q, k, v), iters=10)

    def fb():
        oo = ops.causal_attention(q, k, v)
        oo.backward(g)
        q.grad = k.grad = v.grad = None
    us_fb = timeit(fb, iters=10)
    emit(kernel="flash (decoder, causal GQA)", shape=f"{Bq}x{S}x{nh}/{nkv}x{hd}", fwd_us=round(us_f, 1), fwd_bwd_us=round(us_fb, 1))

if "g128" in only:
    # kernels built with / without -amdgpu-mfma-vgpr-form (run once per library: CAMBRIAN_AMD_LIB)
    for name, M, N, K in (("DINOv2 proj", B * 730, 1536, 1536), ("DINOv2 fc2", B * 730, 1536, 4096), ("SVA q-side", B * 576, 1024, 1024)):
        a, w = rn(M, K), rn(N, K, scale=K ** -0.5)
        us = timeit(lambda: ops.k_gemm(a, w, tile=128))
        emit(kernel="gemm_nt_kernel<128>", shape=f"{name} {M}x{N}x{K}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))
    for name, rows, n_out, k_in in (("SVA proj wgrad", B * 576, 1024, 1024), ("aux proj wgrad", B * 576, 1024, 1536)):
        g, x = rn(rows, n_out), rn(rows, k_in)
        sk = ops._tn_splits(n_out, k_in, rows)
        us = timeit(lambda: ops.k_gemm_tn(g, x, split_k=sk))
        emit(kernel="gemm_tn_kernel", shape=f"{name} {rows} -> {n_out}x{k_in} split {sk}", us=round(us, 1),
             tflops=round(2.0 * rows * n_out * k_in / us / 1e6, 1))

if "epi" in only:
    # gemm_nt_p5_kernel per epilogue form (bias / LayerScale / residual) on the path's shapes
    shapes = [("ConvNeXt s3 fc2 +b+ls+res", B * 4096, 1536, 6144, "bcr"), ("ConvNeXt s2 fc2 +b+ls+res", B * 16384, 768, 3072, "bcr"),
              ("DINOv2 fc1 +b", B * 730, 8192, 1536, "b"), ("DINOv2 qkv +b", B * 730, 4608, 1536, "b"),
              ("DINOv2 fc2 +b+ls+res", B * 730, 1536, 4096, "bcr"), ("CLIP fc2 +b+res", B * 577, 1024, 4096, "br"),
              ("ConvNeXt s1 fc1 +b gelu", B * 65536, 1536, 384, "bg"), ("K|V +b", B * 576, 2048, 1024, "b"),
              ("plain", B * 4096, 6144, 1536, "")]
    for name, M, N, K, mode in shapes:
        a, w = rn(M, K), rn(N, K, scale=K ** -0.5)
        kw = {}
        if "b" in mode:
            kw["bias"] = rn(N, dtype=f32)
        if "c" in mode:
            kw["colscale"] = rn(N, dtype=f32)
        if "r" in mode:
            kw["residual"] = rn(M, N)
        if "g" in mode:
            kw["act"] = L.ACT_GELU_ERF
        out = torch.empty(M, N, device=dev, dtype=bf)
        ops.k_gemm(a, w, out=out, tile=2590, **kw)
        us = timeit(lambda: ops.k_gemm(a, w, out=out, tile=2590, **kw), iters=10)
        emit(kernel="gemm_p5 epilogue", shape=f"{name} {M}x{N}x{K}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))
        o8 = ops.k_gemm(a[:4096], w, tile=2560, **{k: (v[:4096] if k == "residual" else v) for k, v in kw.items()})
        emit(kernel="gemm_p5 epilogue", shape=name, check="p5 == 8-wave kernel, bit for bit (first 4096 rows)",
             equal=bool(torch.equal(o8, out[:4096])))
        del a, w, out, kw

if "lnmulti" in only:
    # the 13 SVA layers' LayerNorm backwards of the 9216-token tower: layer by layer vs cmb_layernorm_bwd_multi
    rows, D, side, r, Ln = B * 9216, 1024, 96, 4, 13
    x = rn(rows, D)
    items, dys = [], []
    for l in range(Ln):
        pos = rn(r * r, D, dtype=f32)
        _, mean, rstd = ops.k_layernorm_fwd(x, None, None, 1e-5, add=pos, side=side, grid_r=r)
        dys.append(rn(rows, D))
        items.append((dys[-1], mean, rstd, pos, l))
    acc = torch.zeros(rows, D, device=dev, dtype=f32)

    def seq():
        for dn, mean, rstd, pos, _ in items:
            ops.k_layernorm_bwd(dn, x, mean, rstd, add=pos, side=side, grid_r=r, dx_acc=acc, want_dadd=True)
    us = timeit(seq, iters=5)
    emit(kernel="layernorm_bwd x13 (layer by layer, fp32 accumulator)", shape=f"{rows}x{D}", us=round(us, 1),
         tbps=round(rows * D * 12 * Ln / us / 1e6, 2))
    dadd = [torch.zeros(r * r, D, device=dev) for _ in range(Ln)]
    for chunk in (7, 4):
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, chunk)
        us = timeit(lambda: ops.k_layernorm_bwd_multi(x, items, side, r, acc, False, dadd), iters=5)
        nl = -(-Ln // chunk)
        emit(kernel="layernorm_bwd_multi", shape=f"{rows}x{D} x{Ln} layers", chunk=chunk, us=round(us, 1),
             tbps=round(rows * D * (2 * Ln + 2 * nl + 8 * nl - 4) / us / 1e6, 2))
    L.knob_set(L.KNOB_LN_MULTI_CHUNK, 7)

if "epiforms" in only:
    # what each epilogue ingredient costs the 4-wave kernel on short-K and long-K shapes of the path; the forms are timed
    # round-robin (7 rounds of 10 launches each, minimum per form): one-after-the-other timing of 80 us kernels is dominated by
    # clock / cache state drift
    for name, M, N, K in (("DINOv2 proj", B * 730, 1536, 1536), ("DINOv2 fc2", B * 730, 1536, 4096), ("ConvNeXt s3 fc1", B * 4096, 6144, 1536),
                          ("ConvNeXt s3 fc2", B * 4096, 1536, 6144), ("CLIP proj", B * 577, 1024, 1024)):
        a, w = rn(M, K), rn(N, K, scale=K ** -0.5)
        bias, cs, res = rn(N, dtype=f32), rn(N, dtype=f32), rn(M, N)
        out = torch.empty(M, N, device=dev, dtype=bf)
        modes = ("", "b", "bc", "br", "bcr", "r", "bg")
        best = {m: 1e30 for m in modes}
        for rep in range(7):
            for mode in modes:
                kw = {}
                if "b" in mode:
                    kw["bias"] = bias
                if "c" in mode:
                    kw["colscale"] = cs
                if "r" in mode:
                    kw["residual"] = res
                if "g" in mode:
                    kw["act"] = L.ACT_GELU_ERF
                us = timeit(lambda: ops.k_gemm(a, w, out=out, tile=2590, **kw), iters=10)
                best[mode] = min(best[mode], us)
        for mode in modes:
            emit(kernel="gemm_p5 epilogue forms", shape=f"{name} {M}x{N}x{K}", mode=mode or "plain", us=round(best[mode], 1),
                 tflops=round(2.0 * M * N * K / best[mode] / 1e6, 1))
        del a, w, out, res
