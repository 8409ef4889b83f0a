"""Round-6 kernel lab (GPU): variants timed round-robin in one process, one JSON line per (case, variant).

    python tools/r06_lab.py gemmtiles          the region's laggard GEMM shapes on every tile configuration (2590 = 4-wave
                                                persistent, 2560 = 8-wave, 128 = 128 x 128 two workgroups per CU), real epilogues
    python tools/r06_lab.py attn               ViT attention: kernel variants (CMB_KNOB_VIT_ATTN) at the three towers' shapes
    python tools/r06_lab.py tn                 weight-gradient shapes: cmb_gemm_tn split-K sweep, batched form
    python tools/r06_lab.py lnmulti|casts|...  see the verbs below
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

ap = argparse.ArgumentParser()
ap.add_argument("verbs", nargs="+")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--images", type=int, default=24)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06", "lab.jsonl"))
ap.add_argument("--tag", default="")
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("--tower", default="DINOv2")
args = ap.parse_args()
B = args.images
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32
os.makedirs(os.path.dirname(args.out), exist_ok=True)
fout = open(args.out, "a")


def rn(*shape, dtype=bf, scale=1.0):
    return (torch.randn(*shape, device=dev, dtype=torch.float32) * scale).to(dtype)


def emit(**kw):
    kw["tag"] = args.tag
    line = json.dumps(kw)
    print(line, flush=True)
    fout.write(line + "\n")
    fout.flush()


def time_variants(fns: dict, iters=None, warm=2):
    """{name: fn} -> {name: median us}; variants interleaved round-robin, an event pair per launch."""
    iters = iters or args.iters
    for _ in range(warm):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    evs = {k: [] for k in fns}
    for _ in range(iters):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            evs[k].append((e0, e1))
    torch.cuda.synchronize()
    out = {}
    for k, lst in evs.items():
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in lst)
        out[k] = t[len(t) // 2]
    return out


verbs = set(args.verbs)

if "gemmtiles" in verbs:
    G = L.ACT_CODES["gelu_erf"]
    cases = [  # name, M, N, K, act, bias, colscale, residual
        ("ConvNeXt s1 fc1", B * 65536, 1536, 384, G, True, False, False),
        ("ConvNeXt s1 fc2", B * 65536, 384, 1536, 0, True, True, True),
        ("ConvNeXt s2 fc1", B * 16384, 3072, 768, G, True, False, False),
        ("ConvNeXt s2 fc2", B * 16384, 768, 3072, 0, True, True, True),
        ("DINOv2 proj", B * 730, 1536, 1536, 0, True, True, True),
        ("DINOv2 fc2", B * 730, 1536, 4096, 0, True, True, True),
        ("DINOv2 qkv", B * 730, 4608, 1536, 0, True, False, False),
        ("SigLIP proj", B * 729, 1152, 1536, 0, True, False, True),
        ("SigLIP fc2", B * 729, 1152, 4352, 0, True, False, True),
        ("SVA q-side 1024", B * 576, 1024, 1024, 0, False, False, False),
        ("SVA q-side 2048", B * 576, 2048, 1024, 0, False, False, False),
    ]
    for name, M, N, K, act, hb, hcs, hr in cases:
        a, w = rn(M, K), rn(N, K, scale=K ** -0.5)
        bias = rn(N, dtype=f32) if hb else None
        cs = rn(N, dtype=f32) if hcs else None
        res = rn(M, N) if hr else None
        out = torch.empty(M, N, device=dev, dtype=bf)
        fns = {}
        for tile in (2590, 2560, 128):
            fns[str(tile)] = (lambda tile=tile: ops.k_gemm(a, w, bias=bias, act=act, colscale=cs, residual=res, out=out, tile=tile))
        ref = None
        same = {}
        for k, f in fns.items():
            f()
            if ref is None:
                ref = out.clone()
            same[k] = bool(torch.equal(out, ref))
        us = time_variants(fns)
        emit(case="gemmtiles", shape=f"{name} {M}x{N}x{K} act{act} b{int(hb)}cs{int(hcs)}r{int(hr)}",
             us={k: round(v, 1) for k, v in us.items()}, tflops={k: round(2.0 * M * N * K / v / 1e6) for k, v in us.items()},
             bit_equal=same)
        del a, w, out, res, ref

if "attn" in verbs:
    from cambrian_amd.model.multimodal_encoder import vit_ops as V
    for name, N, heads, hd in (("CLIP", 577, 16, 64), ("DINOv2", 730, 24, 64), ("SigLIP", 729, 16, 72)):
        hdp = 96 if hd == 72 else hd
        qkv = rn(B * N, 3 * heads * hdp, scale=1.0)
        fns, ref, same = {}, None, {}
        for var in (0, 1, 2, 3):
            try:
                L.knob_set(L.KNOB_VIT_ATTN, var)
            except Exception:
                continue
            fns[str(var)] = (lambda var=var: (L.knob_set(L.KNOB_VIT_ATTN, var), V.k_vit_attn(qkv, B, N, heads, hdp, hd ** -0.5))[1])
            y = fns[str(var)]()
            if ref is None:
                ref = y
            same[str(var)] = float((y.float() - ref.float()).abs().max())
        us = time_variants(fns, iters=20)
        flops = 4.0 * B * heads * N * N * hd
        emit(case="vit_attn", shape=f"{name} N={N} heads={heads} hd={hd}", us={k: round(v, 1) for k, v in us.items()},
             useful_tflops={k: round(flops / v / 1e6) for k, v in us.items()}, max_abs_diff_vs_v0=same)
        L.knob_set(L.KNOB_VIT_ATTN, 1)
        del qkv

if "tn" in verbs:
    R = B * 576
    for name, n_out, k_in in (("1024x1024", 1024, 1024), ("2048x1024", 2048, 1024), ("1024x2048", 1024, 2048),
                              ("4096x1024", 4096, 1024), ("1024x4096", 1024, 4096), ("1024x5120", 1024, 5120)):
        g, x = rn(R, n_out), rn(R, k_in)
        fns = {}
        for sp in (1, 2, 4, 8, 16):
            fns[f"tn_split{sp}"] = (lambda sp=sp: ops.k_gemm_tn(g, x, split_k=sp))
        fns["auto"] = lambda: ops.k_gemm_tn(g, x, split_k=ops._tn_splits(n_out, k_in, R))
        us = time_variants(fns)
        emit(case="tn", shape=f"dW {name} from {R} rows", us={k: round(v, 1) for k, v in us.items()},
             tflops={k: round(2.0 * R * n_out * k_in / v / 1e6) for k, v in us.items()}, auto_split=ops._tn_splits(n_out, k_in, R))
        del g, x

if "attnone" in verbs:   # a few launches of ONE variant at ONE tower's shape: the workload of the rocprofv3 --pmc passes
    from cambrian_amd.model.multimodal_encoder import vit_ops as V
    name, N, heads, hd = {"CLIP": ("CLIP", 577, 16, 64), "DINOv2": ("DINOv2", 730, 24, 64), "SigLIP": ("SigLIP", 729, 16, 72)}[args.tower]
    hdp = 96 if hd == 72 else hd
    qkv = rn(B * N, 3 * heads * hdp)
    L.knob_set(L.KNOB_VIT_ATTN, args.variant)
    for _ in range(6):
        V.k_vit_attn(qkv, B, N, heads, hdp, hd ** -0.5)
    torch.cuda.synchronize()

if "lnfold" in verbs:   # LayerNorm + linear (round 5) against row statistics + the folded-LayerNorm linear, per kernel
    from cambrian_amd.model.multimodal_encoder import vit_ops as V
    cases = [("ConvNeXt s3 fc1", B * 4096, 6144, 1536, "gelu_erf"), ("ConvNeXt s1 fc1", B * 65536, 1536, 384, "gelu_erf"),
             ("ConvNeXt s2 fc1", B * 16384, 3072, 768, "gelu_erf"), ("DINOv2 qkv", B * 730, 4608, 1536, "none"),
             ("DINOv2 fc1", B * 730, 8192, 1536, "swiglu_pairs"), ("SigLIP fc1", B * 729, 4352, 1152, "gelu_erf"),
             ("CLIP qkv", B * 577, 3072, 1024, "none")]
    for name, M, N, K, act in cases:
        x = rn(M, K, scale=2.0)
        w = rn(N, K, dtype=f32, scale=K ** -0.5)
        b, gam, bet = rn(N, dtype=f32), 1 + 0.1 * rn(K, dtype=f32), 0.1 * rn(K, dtype=f32)
        wb = w.to(bf)
        w2, cs, b2 = V.fold_ln_into_linear(w, b, gam, bet, bf)
        code = L.ACT_CODES[act] if act in L.ACT_CODES else L.ACT_SWIGLU_PAIRS
        n_out = N // 2 if act == "swiglu_pairs" else N
        out = torch.empty(M, n_out, device=dev, dtype=bf)
        st = ops.k_row_stats(x, 1e-6)
        xn, _, _ = ops.k_layernorm_fwd(x, gam, bet, 1e-6, want_stats=False)
        fns = {"ln": lambda: ops.k_layernorm_fwd(x, gam, bet, 1e-6, want_stats=False),
               "gemm": lambda: ops.k_gemm(xn, wb, bias=b, act=code, out=out),
               "stats": lambda: ops.k_row_stats(x, 1e-6),
               "gemm_rs": lambda: ops.k_gemm(x, w2, bias=b2, act=code, out=out, row_stats=st, col_sum=cs),
               "gemm_on_raw_x": lambda: ops.k_gemm(x, wb, bias=b, act=code, out=out)}
        us = time_variants(fns)
        emit(case="lnfold", shape=f"{name} {M}x{N}x{K} {act}", us={k: round(v, 1) for k, v in us.items()},
             old=round(us["ln"] + us["gemm"], 1), new=round(us["stats"] + us["gemm_rs"], 1))
        del x, xn, out

if "expand" in verbs:   # the per-head K = 64 "expand" product of the absorbed SVA path: gemm_k64 vs the 128-tile kernel (CMB_GEMM_K64=0 build env)
    Bq, heads, hd, Cin = B * 576, 16, 64, 1024
    x, wt = rn(Bq, heads * hd), rn(Cin, heads * hd, scale=0.05)
    U = torch.empty(Bq, heads, Cin, device=dev, dtype=bf)
    f = lambda: ops.k_gemm_batched(x, wt, U, batch=heads, M=Bq, N=Cin, K=hd, lda=heads * hd, ldb=heads * hd, ldc=heads * Cin, a_bs=hd, b_bs=hd, c_bs=Cin)  # noqa: E731
    f()
    want = torch.einsum("qhj,chj->qhc", x[:2048].float().view(-1, heads, hd), wt.float().view(Cin, heads, hd))
    err = float((U[:2048].float() - want).abs().max() / want.abs().max())
    us = time_variants({"k": f}, iters=20)["k"]
    emit(case="expand", shape=f"{Bq}x{heads}x{Cin} K={hd}", kernel=L.load().cmb_gemm_last_kernel(), us=round(us, 1),
         write_tbps=round(U.numel() * 2 / us / 1e6, 2), rel_err=err)

if "pair" in verbs:   # cmb_gemm_pair: DINOv2's and SigLIP's same-position residual linears as one launch vs two (bit-equal results)
    lib = L.load()
    cases = [("proj", (B * 730, 1536, 1536, True), (B * 729, 1152, 1152, False)),
             ("fc2", (B * 730, 1536, 4096, True), (B * 729, 1152, 4304 if False else 4352, False))]
    for name, (M0, N0, K0, ls0), (M1, N1, K1, ls1) in cases:
        def mk(M, N, K, ls):
            return dict(a=rn(M, K), w=rn(N, K, scale=K ** -0.5), bias=rn(N, dtype=f32), colscale=rn(N, dtype=f32) if ls else None,
                        residual=rn(M, N), out=torch.empty(M, N, device=dev, dtype=bf))
        k0, k1 = mk(M0, N0, K0, ls0), mk(M1, N1, K1, ls1)
        ops.k_gemm(**k0); ops.k_gemm(**k1)
        r0, r1 = k0["out"].clone(), k1["out"].clone()
        k0["out"].zero_(); k1["out"].zero_()
        ops.k_gemm_pair(k0, k1)
        paired = int(lib.cmb_gemm_pair_last())
        same = bool(torch.equal(k0["out"], r0) and torch.equal(k1["out"], r1))
        us = time_variants({"two": lambda: (ops.k_gemm(**k0), ops.k_gemm(**k1)), "pair": lambda: ops.k_gemm_pair(k0, k1),
                            "first": lambda: ops.k_gemm(**k0), "second": lambda: ops.k_gemm(**k1)}, iters=20)
        fl = 2.0 * (M0 * N0 * K0 + M1 * N1 * K1)
        emit(case="pair", shape=f"{name} {M0}x{N0}x{K0} + {M1}x{N1}x{K1}", paired=paired, bit_equal=same,
             us={k: round(v, 1) for k, v in us.items()}, tflops={k: round(fl / us[k] / 1e6) for k in ("two", "pair")})
        del k0, k1, r0, r1

if "lnmulti" in verbs:   # the 13 SVA layers' LayerNorm backwards of the 9216-token tower: knob LN_MULTI_CHUNK 4 / 7 / 104 / 107
    rows, D, side, r, Ln = B * 9216, 1024, 96, 4, 13
    x = rn(rows, D)
    items = []
    for l in range(Ln):
        pos = rn(r * r, D, dtype=f32)
        _, mean, rstd = ops.k_layernorm_fwd(x, None, None, 1e-5, add=pos, side=side, grid_r=r)
        items.append((rn(rows, D), mean, rstd, pos, l))
    acc = torch.zeros(rows, D, device=dev, dtype=f32)
    ref = None
    for chunk in (4, 5, 6, 7):   # (104 / 107 = d(pos) sums by ds_add_f32 in LDS: built, 3.5x slower, removed — profiles/r06_lab.md)
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, chunk)
        dadd = [torch.zeros(r * r, D, device=dev) for _ in range(Ln)]
        ops.k_layernorm_bwd_multi(x, items, side, r, acc, False, dadd)
        got = (acc.clone(), torch.stack(dadd))
        if ref is None:
            ref = got
        err = (float((got[0] - ref[0]).abs().max() / ref[0].abs().max()), float((got[1] - ref[1]).abs().max() / ref[1].abs().max()))
        us4 = time_variants({"first4": lambda: ops.k_layernorm_bwd_multi(x, items[:4], side, r, acc, False, dadd),
                             "all13": lambda: ops.k_layernorm_bwd_multi(x, items, side, r, acc, False, dadd)}, iters=6)
        nl = -(-Ln // (chunk % 100))
        emit(case="lnmulti", chunk=chunk, us={k: round(v, 1) for k, v in us4.items()}, rel_diff_vs_chunk4=err,
             tbps_all13=round(rows * D * (2 * Ln + 2 * nl + 8 * nl - 4) / us4["all13"] / 1e6, 2),
             tbps_first4=round(rows * D * (2 * 4 + 2 + 4) / us4["first4"] / 1e6, 2))
    L.knob_set(L.KNOB_LN_MULTI_CHUNK, 5)

if "smallm" in verbs:
    a, w = rn(B, 1024), rn(1024, 1024, scale=1 / 32)
    us = time_variants({"auto": lambda: ops.k_gemm(a, w), "tile128": lambda: ops.k_gemm(a, w, tile=128)}, iters=30)
    emit(case="smallm", shape=f"{B}x1024x1024", us={k: round(v, 1) for k, v in us.items()})

if "contract" in verbs:   # the per-head N = 64 "contract" product of the absorbed SVA path: gemm_n64 vs the 128-tile kernel (CMB_GEMM_N64=0 env)
    Bq, heads, hd, Cin = B * 576, 16, 64, 1024
    xb, w = rn(Bq, heads, Cin), rn(heads * hd, Cin, scale=1 / 32)
    out = torch.empty(Bq, heads * hd, device=dev, dtype=bf)
    f = lambda: ops.k_gemm_batched(xb.view(Bq, heads * Cin), w, out, batch=heads, M=Bq, N=hd, K=Cin, lda=heads * Cin, ldb=Cin, ldc=heads * hd, a_bs=Cin, b_bs=hd * Cin, c_bs=hd)  # noqa: E731
    f()
    want = torch.einsum("qhc,hjc->qhj", xb[:2048].float(), w.float().view(heads, hd, Cin)).reshape(2048, heads * hd)
    err = float((out[:2048].float() - want).abs().max() / want.abs().max())
    us = time_variants({"k": f}, iters=20)["k"]
    emit(case="contract", shape=f"{Bq}x{heads}x{hd} K={Cin}", kernel=L.load().cmb_gemm_last_kernel(), us=round(us, 1),
         read_tbps=round(xb.numel() * 2 / us / 1e6, 2), rel_err=err)

if "plain" in verbs:   # plain (no epilogue work) launches of the persistent kernel on long-K and short-K shapes: the K loop's rate
    for name, M, N, K in (("s3 fc1", B * 4096, 6144, 1536), ("s1 fc1", B * 65536, 1536, 384), ("s2 fc1", B * 16384, 3072, 768), ("8192^3/8", 8192, 8192, 4096)):
        a, w = rn(M, K), rn(N, K, scale=K ** -0.5)
        out = torch.empty(M, N, device=dev, dtype=bf)
        us = time_variants({"p5": lambda: ops.k_gemm(a, w, out=out, tile=2590)}, iters=8)["p5"]
        emit(case="plain", shape=f"{name} {M}x{N}x{K}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6))
        del a, w, out

if "colsum" in verbs:   # column sums at the step's shapes: knob COLSUM_WGS 0 (256 rows per workgroup) against workgroup targets
    for (R, Cc, scaled) in ((B * 576, 1024, False), (B * 576, 2048, False), (B * 576, 4096, False), (B * 576, 1024, True),
                            (B * 9216, 2048, False)):
        x = rn(R, Cc)
        sc = rn(R, Cc // 64, dtype=f32) if scaled else None
        out = torch.zeros(Cc, device=dev)
        ref = x.float().mul(sc.repeat_interleave(64, 1)).sum(0) if scaled else x.float().sum(0)
        fns, errs = {}, {}
        for wgs in (0, 256, 512, 1024, 2048):
            def f(wgs=wgs):
                L.knob_set(L.KNOB_COLSUM_WGS, wgs)
                ops.k_colsum(x, out, row_scale=sc, group=64 if scaled else 0)
            out.zero_()
            f()
            errs[str(wgs)] = float((out - ref).abs().max() / ref.abs().max())
            fns[str(wgs)] = f
        us = time_variants(fns)
        emit(case="colsum", R=R, C=Cc, scaled=scaled, us={k: round(v, 1) for k, v in us.items()}, rel_err=errs)
    L.knob_set(L.KNOB_COLSUM_WGS, 768)

if "lnbwd" in verbs:   # LayerNorm backward of the SVA query-side inputs (13 824 x 1024, affine): rows per workgroup
    rows, D = B * 576, 1024
    x, dy, gamma = rn(rows, D), rn(rows, D), rn(D, dtype=f32)
    _, mean, rstd = ops.k_layernorm_fwd(x, gamma, torch.zeros(D, device=dev), 1e-5)
    fns, ref, errs = {}, None, {}
    for rpb in (8, 16, 32, 64, 128):
        def f(rpb=rpb):
            L.knob_set(L.KNOB_LN_BWD_ROWS, rpb)
            return ops.k_layernorm_bwd(dy, x, mean, rstd, gamma=gamma)
        got = f()
        if ref is None:
            ref = got
        errs[str(rpb)] = [float((g.float() - r_.float()).abs().max() / r_.float().abs().max()) for g, r_ in zip(got[:3], ref[:3])]
        fns[str(rpb)] = f
    us = time_variants(fns)
    emit(case="lnbwd", rows=rows, D=D, us={k: round(v, 1) for k, v in us.items()}, rel_diff_vs_8=errs)
    L.knob_set(L.KNOB_LN_BWD_ROWS, 32)
