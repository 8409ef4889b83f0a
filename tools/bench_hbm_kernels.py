"""HBM-bound kernels of the hot path at their release shapes (16 images per GPU): algorithmic bytes per launch, HIP-event
time, GB/s, fraction of the 8 TB/s HBM3E peak and of the 6.3 TB/s a float4 copy reaches on this chip
(MI355X_MICROARCH.md §HBM).  The table VERDICT r2 "missing #6" / next #5 asks for (-> profiles/r03_hbm_kernels.md).

    python tools/bench_hbm_kernels.py [--iters 20] [--only NAME] [--md gpurun_out/hbm.md] [--json gpurun_out/hbm.json]

Algorithmic bytes = every logical tensor the kernel must read or write, counted once, in the step's
dtype (bf16 activations, fp32 statistics / accumulators / parameters).  Buffers are larger than the 256 MiB Infinity Cache
or are rotated, so launches do not re-hit the previous launch's lines.  With --only the script runs ONE kernel (the
workload of the rocprofv3 --pmc passes: FETCH_SIZE x 2 / WRITE_SIZE per launch, tools/pmc_hbm.sh)."""
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

B = int(os.environ.get("CAMBRIAN_HBM_BENCH_IMAGES", "24"))   # images per GPU (24 = the driver's batch since round 3; r03 tables: 16)
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32


def rn(*shape, dtype=bf):
    return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)


def cases():
    """(name, kernel name in the rocprof table, shape text, algorithmic bytes, callable) — built lazily."""
    out = []

    def add(name, kern, shape, nbytes, make):
        out.append((name, kern, shape, nbytes, make))

    # ---- LayerNorm forward (affine, bf16): ConvNeXt stage-3 block LN and a ViT block LN
    for rows, D, tag in ((B * 4096, 1536, "ConvNeXt stage 3"), (B * 730, 1536, "DINOv2 block"), (B * 65536, 384, "ConvNeXt stage 1")):
        def mk(rows=rows, D=D):
            x, g, b = rn(rows, D), rn(D, dtype=f32), rn(D, dtype=f32)
            return lambda: ops.k_layernorm_fwd(x, g, b, 1e-6, want_stats=False)
        add(f"layernorm_fwd [{tag}]", "layernorm_fwd_lds_kernel", f"{rows}x{D}", rows * D * 2 * 2, mk)
    # ---- SVA shared-statistics normalisation (x + pos -> xhat, statistics kept): the ConvNeXt tower's 9216 tokens / image
    rows, D = B * 9216, 1024

    def mk_svan(rows=rows, D=D):
        x, pos = rn(rows, D), rn(16, D, dtype=f32)
        return lambda: ops.k_layernorm_fwd(x, None, None, 1e-5, add=pos, side=96, grid_r=4)
    add("layernorm_fwd [SVA xhat, ConvNeXt tower]", "layernorm_fwd_kernel", f"{rows}x{D}", rows * D * 2 * 2 + rows * 8, mk_svan)

    def mk_svan_multi(rows=rows, D=D):
        x = rn(rows, D)
        adds = [rn(16, D, dtype=f32) for _ in range(13)]
        return lambda: ops.k_layernorm_fwd_multi(x, adds, 1e-5, 96, 4)
    # x once, 13 normalised outputs + their statistics
    add("layernorm_fwd_multi [the 13 SVA layers' xhat of the ConvNeXt tower, one pass]", "layernorm_fwd_multi_kernel",
        f"{rows}x{D} x 13 layers", rows * D * 2 * 14 + 13 * rows * 8, mk_svan_multi)

    def mk_svanb(rows=rows, D=D):
        x, dn, pos = rn(rows, D), rn(rows, D), rn(16, D, dtype=f32)
        _, mean, rstd = ops.k_layernorm_fwd(x, None, None, 1e-5, add=pos, side=96, grid_r=4)
        acc = torch.zeros(rows, D, device=dev, dtype=f32)
        return lambda: ops.k_layernorm_bwd(dn, x, mean, rstd, add=pos, side=96, grid_r=4, dx_acc=acc, want_dadd=True)
    # reads dn, x (bf16), fp32 accumulator read + write, statistics
    add("layernorm_bwd [SVA, fp32 accumulate, ConvNeXt tower]", "layernorm_bwd_kernel<.., true>", f"{rows}x{D}",
        rows * D * (2 + 2 + 4 + 4) + rows * 8, mk_svanb)

    def mk_multi(rows=rows, D=D):
        x = rn(rows, D)
        items, dadd = [], []
        for l in range(13):
            pos = rn(16, D, dtype=f32)
            _, mean, rstd = ops.k_layernorm_fwd(x, None, None, 1e-5, add=pos, side=96, grid_r=4)
            items.append((rn(rows, D), mean, rstd, pos, l))
            dadd.append(torch.zeros(16, D, device=dev, dtype=f32))
        acc = torch.empty(rows, D, device=dev, dtype=f32)
        return lambda: ops.k_layernorm_bwd_multi(x, items, 96, 4, acc, False, dadd)
    # 13 layers in launches of 5 + 5 + 3 (round 6; 4 + 4 + 4 + 1 before): x read per launch (3 x 2), every gradient once (13 x 2),
    # the fp32 sum written by each launch and re-read by the next two (3 x 4 + 2 x 4)
    add("layernorm_bwd_multi [the 13 SVA layers of the ConvNeXt tower, one deferred pass]", "layernorm_bwd_multi_kernel",
        f"{rows}x{D} x 13 layers", rows * D * (3 * 2 + 13 * 2 + 3 * 4 + 2 * 4), mk_multi)

    def mk_lnb(D=D):
        r2 = B * 576
        x, dy, g = rn(r2, D), rn(r2, D), rn(D, dtype=f32)
        _, mean, rstd = ops.k_layernorm_fwd(x, g, g, 1e-5)
        return lambda: ops.k_layernorm_bwd(dy, x, mean, rstd, gamma=g)
    add("layernorm_bwd [SVA query side]", "layernorm_bwd_kernel", f"{B * 576}x{D}", B * 576 * D * 2 * 3, mk_lnb)
    # ---- SVA attention forward / backward at the release geometry
    heads, hd, qside, r_list = 16, 64, 24, [1, 1, 1, 4]

    def sva_inputs():
        q = rn(B * 576, heads * hd)
        kvs = [rn(B * (qside * r) ** 2, 2 * heads * hd) for r in r_list]
        return q, kvs
    kv_bytes = sum(B * (qside * r) ** 2 * 2 * heads * hd * 2 for r in r_list)
    q_bytes = B * 576 * heads * hd * 2

    def mk_svaf():
        q, kvs = sva_inputs()
        return lambda: ops.k_sva_attn_fwd(q, kvs, None, r_list, B, qside, heads, hd)
    add("sva_fwd", "sva_fwd_kernel", f"{B}x576 q, {B}x10944 kv rows x 2048", kv_bytes + 2 * q_bytes + B * 576 * heads * 4, mk_svaf)

    def mk_svab():
        q, kvs = sva_inputs()
        o, lse = ops.k_sva_attn_fwd(q, kvs, None, r_list, B, qside, heads, hd)
        do = rn(B * 576, heads * hd)
        return lambda: ops.k_sva_attn_bwd(do, q, kvs, None, r_list, o, lse, B, qside, heads, hd)
    add("sva_bwd", "sva_bwd_kernel", "same", 2 * kv_bytes + 5 * q_bytes, mk_svab)
    # ---- absorbed SVA attention core (round 3 default): window tower's tokens + U read, Xb written (forward); x, U, dXb read,
    #      dU, dx written (backward); plus the one-key towers' K|V rows and q / out rows
    def abs_desc():
        import ctypes as C
        Bq = B * 576
        t = dict(q=rn(Bq, 1024), kvs=[rn(Bq, 2048) for _ in range(3)], xhat=rn(B * 9216, 1024), U=rn(Bq, 16, 1024) * 0.05,
                 bk=rn(1024, dtype=f32), bv=rn(1024, dtype=f32), out=rn(Bq, 1024), xbar=rn(Bq, 16, 1024),
                 m3=torch.empty(Bq, 16, device=dev), P=torch.empty(Bq, 16, 20, device=dev), dout=rn(Bq, 1024),
                 dxbar=rn(Bq, 16, 1024), dq=rn(Bq, 1024), dU=rn(Bq, 16, 1024), dcb=torch.empty(Bq, 16, device=dev),
                 dxhat=rn(B * 9216, 1024))
        t["dkvs"] = [torch.empty_like(k) for k in t["kvs"]]
        d = L.SvaAbsDesc()
        d.B, d.qside, d.heads, d.hd, d.ntowers, d.window_major, d.ra = B, 24, 16, 64, 3, 0, 4
        d.q, d.ldq = t["q"].data_ptr(), 1024
        for i, kv in enumerate(t["kvs"]):
            d.r[i] = 1
            d.kv[i], d.ldkv[i], d.mask[i], d.dkv[i] = kv.data_ptr(), 2048, None, t["dkvs"][i].data_ptr()
        d.xhat, d.ldx, d.mask_a, d.U, d.bk, d.bv = t["xhat"].data_ptr(), 1024, None, t["U"].data_ptr(), t["bk"].data_ptr(), t["bv"].data_ptr()
        d.out, d.ldo, d.xbar, d.m3, d.P = t["out"].data_ptr(), 1024, t["xbar"].data_ptr(), t["m3"].data_ptr(), t["P"].data_ptr()
        d.dout, d.lddo, d.dxbar = t["dout"].data_ptr(), 1024, t["dxbar"].data_ptr()
        d.dq, d.lddq, d.dU, d.dcb, d.dxhat, d.lddx = t["dq"].data_ptr(), 1024, t["dU"].data_ptr(), t["dcb"].data_ptr(), t["dxhat"].data_ptr(), 1024
        return d, t, C
    big = B * 576 * 16 * 1024 * 2          # one [Bq, 16, 1024] bf16 tensor = the window tower's tokens = 302 MB at 16 images
    small = B * 576 * 1024 * 2             # one [Bq, 1024] row set

    def mk_absf():
        d, t, C = abs_desc()
        lib = L.load()
        return lambda: (L.check(lib.cmb_sva_abs_fwd(C.byref(d), L.stream_ptr(dev)), "abs_fwd"), t)[0]
    add("sva_abs_fwd", "sva_abs_fwd_kernel", f"{B}x576 q, 4x4 window, 3 one-key towers", 3 * big + 8 * small + B * 576 * 16 * 84, mk_absf)

    def mk_absb():
        d, t, C = abs_desc()
        lib = L.load()
        L.check(lib.cmb_sva_abs_fwd(C.byref(d), L.stream_ptr(dev)), "abs_fwd")
        return lambda: (L.check(lib.cmb_sva_abs_bwd(C.byref(d), L.stream_ptr(dev)), "abs_bwd"), t)[0]
    add("sva_abs_bwd", "sva_abs_bwd_kernel", "same", 5 * big + 15 * small + B * 576 * 16 * 84, mk_absb)
    # ---- RMSNorm family (decoder side, 16 x 2048 tokens x 4096)
    rows, D = B * 2048, 4096

    def mk_rms(rows=rows, D=D):
        x, w = rn(rows, D), rn(D, dtype=f32)
        y, rstd = torch.empty_like(x), torch.empty(rows, device=dev, dtype=f32)
        lib = L.load()
        return lambda: L.check(lib.cmb_rmsnorm_fwd(L.BF16, x.data_ptr(), rows, D, w.data_ptr(), 1e-5, y.data_ptr(), rstd.data_ptr(),
                                                   L.stream_ptr(dev)), "rmsnorm_fwd")
    add("rmsnorm_fwd", "rmsnorm_fwd_kernel", f"{rows}x{D}", rows * D * 2 * 2, mk_rms)

    def mk_addrms(rows=rows, D=D):
        x, d_, w = rn(rows, D), rn(rows, D), rn(D, dtype=f32)
        s, y, rstd = torch.empty_like(x), torch.empty_like(x), torch.empty(rows, device=dev, dtype=f32)
        lib = L.load()
        return lambda: L.check(lib.cmb_add_rmsnorm_fwd(L.BF16, x.data_ptr(), d_.data_ptr(), rows, D, w.data_ptr(), 1e-5, s.data_ptr(),
                                                       y.data_ptr(), rstd.data_ptr(), L.stream_ptr(dev)), "add_rmsnorm")
    add("add_rmsnorm_fwd", "add_rmsnorm_fwd_kernel", f"{rows}x{D}", rows * D * 2 * 4, mk_addrms)

    def mk_rmsb(rows=rows, D=D):
        x, dy, gs, w = rn(rows, D), rn(rows, D), rn(rows, D), rn(D, dtype=f32)
        rstd, dx = torch.rand(rows, device=dev, dtype=f32), torch.empty_like(x)
        lib = L.load()
        return lambda: L.check(lib.cmb_rmsnorm_bwd_add(L.BF16, dy.data_ptr(), x.data_ptr(), gs.data_ptr(), rows, D, w.data_ptr(),
                                                       rstd.data_ptr(), dx.data_ptr(), L.stream_ptr(dev)), "rmsnorm_bwd_add")
    add("rmsnorm_bwd_add", "rmsnorm_bwd_add_kernel", f"{rows}x{D}", rows * D * 2 * 4, mk_rmsb)
    # ---- ConvNeXt multi-stage resample: stage maps -> 96 x 96 grid of the 5760-channel buffer (one launch per stage).
    # Algorithmic bytes = the source pixels the bilinear taps actually TOUCH (a 256 -> 96 downscale reads 2 of every 2.67
    # rows and columns: 56 % of the map; VERDICT r3 #8: charging the whole map gave an impossible 8.0 TB/s) + the output
    def touched(side, out=96):
        import numpy as np
        o = np.arange(out)
        src = np.clip((o + 0.5) * (side / out) - 0.5, 0, side - 1)
        i0 = np.floor(src).astype(int)
        return len(set(i0.tolist()) | set(np.minimum(i0 + 1, side - 1).tolist()))
    for side, C, tag in ((256, 384, "stage 1"), (64, 1536, "stage 3")):
        def mk(side=side, C=C):
            x = rn(B, side * side, C)
            o = torch.empty(B, 9216, 5760, device=dev, dtype=bf)
            return lambda: V.k_resample(x, side, side, o, 96, 96, 0)
        u = touched(side)
        add(f"resample_bilinear [{tag} -> 96x96]", "resample_kernel", f"{B}x{side}^2x{C} ({u}^2 source pixels touched)",
            B * (u * u + 9216) * C * 2, mk)
    # ---- depthwise 7x7 (NHWC), stage 3 and stage 1
    for side, C, tag in ((64, 1536, "stage 3"), (256, 384, "stage 1")):
        def mk(side=side, C=C):
            x, w, b = rn(B, side, side, C), rn(49, C, dtype=f32), rn(C, dtype=f32)
            return lambda: V.k_dwconv7x7(x, w, b)
        add(f"dwconv7x7 [{tag}] (VALU-bound: 98 FLOP / element)", "dwconv7x7_col_kernel", f"{B}x{side}^2x{C}",
            B * side * side * C * 2 * 2, mk)
    # ---- embedding splice, column sum (bias gradient), token mean, row gather, transpose, cast
    S, H, V_ = 2048, 4096, 128256

    def mk_splice():
        ids = torch.randint(1000, 30000, (B, S), device=dev)
        ids[:, 91] = -200
        tab, feat, nl = rn(V_, H), rn(B, 576, H), rn(H)
        return lambda: ops.embed_splice(ids, tab, feat, nl, 24)
    add("embed_splice_fwd", "embed_splice_fwd_kernel", f"{B}x{S}x{H}", B * S * H * 2 * 2, mk_splice)

    def mk_colsum():
        x = rn(B * 9216, 2048)
        o = torch.zeros(2048, device=dev, dtype=f32)
        return lambda: ops.k_colsum(x, o)
    add("colsum", "colsum_kernel", f"{B * 9216}x2048", B * 9216 * 2048 * 2, mk_colsum)

    def mk_tm():
        x = rn(B, 576, 1024)
        return lambda: ops.k_token_mean(x)
    add("token_mean_fwd", "token_mean_kernel", f"{B}x576x1024", B * 576 * 1024 * 2, mk_tm)

    def mk_rows():
        hid = rn(B, S, H)
        o = torch.empty(B * 576, H, device=dev, dtype=bf)
        return lambda: ops.k_copy_rows(hid.view(-1)[91 * H:], ops.hook_row_map(S, H, 24), o, L.identity_map(H), B * 576, H)
    add("copy_rows [hook gather]", "copy_rows_kernel", f"{B * 576}x{H}", B * 576 * H * 2 * 2, mk_rows)

    def mk_tr():
        x = rn(B * 9216, 2048)
        return lambda: ops.k_transpose(x)
    add("transpose [dKV for the wgrad GEMM]", "transpose_kernel", f"{B * 9216}x2048", B * 9216 * 2048 * 2 * 2, mk_tr)

    def mk_cast():
        x = rn(4096, 4096, dtype=f32)
        return lambda: ops.k_cast(x, bf)
    add("cast fp32 -> bf16 [a 4096^2 master weight]", "cast_kernel", "4096x4096", 4096 * 4096 * 6, mk_cast)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--md", default=None)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    rows = []
    for name, kern, shape, nbytes, make in cases():
        if args.only and args.only not in name:
            continue
        f = make()
        for _ in range(3):
            f()
        evs = []
        for _ in range(args.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        avg = sum(us) / len(us)
        gbs = nbytes / avg / 1e3
        rows.append(dict(name=name, kernel=kern, shape=shape, algorithmic_bytes=nbytes, avg_us=avg, median_us=us[len(us) // 2],
                         GBps=gbs, frac_of_8TBps=gbs / 8000.0, frac_of_6p3TBps=gbs / 6300.0))
        del f
        torch.cuda.empty_cache()
    lines = ["| kernel (case) | shape | algorithmic MB / launch | avg us | GB/s | of 8 TB/s | of 6.3 TB/s |", "|---|---|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append(f"| `{r['kernel']}` — {r['name']} | {r['shape']} | {r['algorithmic_bytes'] / 1e6:.1f} | {r['avg_us']:.1f} | "
                     f"{r['GBps']:.0f} | {r['frac_of_8TBps']:.2f} | {r['frac_of_6p3TBps']:.2f} |")
    text = "\n".join(lines)
    print(text)
    if args.md:
        open(args.md, "w").write(text + "\n")
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
