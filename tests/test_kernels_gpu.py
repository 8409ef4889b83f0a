"""GPU parity tests of every C-ABI kernel against a plain fp32 CPU restatement of the same arithmetic
(`-m gpu`).  fp32 dtype exercises the exact-fp32 kernels (tight tolerance), bf16 the production path."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu

DTYPES = [("fp32", torch.float32), ("bf16", torch.bfloat16)]


def _ops():
    from cambrian_amd import ops, lib
    return ops, lib


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


# ------------------------------------------------------------------------------------------------ gemm
@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (577, 1152, 256), (256, 4304, 64),
                                   (33, 8, 64), (1000, 2048, 1024), (130, 136, 192)])
def test_gemm_plain(dev, name, dt, M, N, K):
    ops, L = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, w = _rand(g, M, K), _rand(g, N, K)
    ref = a.to(dt).float() @ w.to(dt).float().T
    out = ops.k_gemm(a.to(dev, dt), w.to(dev, dt))
    assert out.shape == (M, N)
    assert rel_err(out, ref) < (TOL[name] if name == "fp32" else 1e-2)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("act", ["none", "gelu_erf", "gelu_tanh", "quick_gelu", "silu"])
def test_gemm_epilogue(dev, name, dt, act):
    ops, L = _ops()
    g = torch.Generator().manual_seed(11)
    M, N, K = 300, 264, 128
    a, w = _rand(g, M, K), _rand(g, N, K, scale=0.2)
    bias, cs, res = _rand(g, N), _rand(g, N), _rand(g, M, N)
    a_, w_, res_ = a.to(dt).float(), w.to(dt).float(), res.to(dt).float()
    pre = a_ @ w_.T * 0.5 + bias
    fn = {"none": lambda x: x, "gelu_erf": F.gelu, "gelu_tanh": lambda x: F.gelu(x, approximate="tanh"),
          "quick_gelu": lambda x: x * torch.sigmoid(1.702 * x), "silu": F.silu}[act]
    ref = fn(pre) * cs + res_
    pre_out = torch.empty(M, N, dtype=dt, device=dev)
    out = ops.k_gemm(a.to(dev, dt), w.to(dev, dt), bias=bias.to(dev), act=L.ACT_CODES[act], colscale=cs.to(dev),
                     residual=res.to(dev, dt), pre_out=pre_out, alpha=0.5)
    tol = TOL[name] * (5 if name == "fp32" else 1)
    assert rel_err(out, ref) < tol
    assert rel_err(pre_out, pre) < tol


@pytest.mark.parametrize("name,dt", DTYPES)
def test_gemm_rowmaps_and_inplace_slice(dev, name, dt):
    """A rows gathered from hidden[:, p:p+600].view(B,24,25,H)[:, :, :24]; C scattered back to the same rows
    (the in-LLM SVA hook, cambrian_llama.py:181-207), here with side 6 instead of 24."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(5)
    B, S, H, side, p0 = 3, 80, 128, 6, 7
    hidden = _rand(g, B, S, H).to(dt)
    w = _rand(g, H, H, scale=0.1).to(dt)
    rows = hidden[:, p0:p0 + side * (side + 1)].reshape(B, side, side + 1, H)[:, :, :side].reshape(-1, H).float()
    ref_rows = rows @ w.float().T + rows
    ref = hidden.float().clone()
    ref[:, p0:p0 + side * (side + 1)].view(B, side, side + 1, H)[:, :, :side] = ref_rows.view(B, side, side, H)
    hd, src = hidden.to(dev), hidden.to(dev).clone()
    amap = L.make_map(side * side, side, S * H, (side + 1) * H, H)
    base, sbase = hd.view(-1)[p0 * H:], src.view(-1)[p0 * H:]
    # A is gathered from a separate buffer; residual and C alias (each element is read and written by the
    # same thread in the epilogue, which is what makes the in-place write-back of the hook legal)
    ops.k_gemm(sbase, w.to(dev), M=B * side * side, a_map=amap, residual=base, r_map=amap, out=base, c_map=amap)
    assert rel_err(hd, ref) < TOL[name]
    # untouched rows (text tokens and the newline column) must be bit-identical
    keep = torch.ones(B, S, dtype=torch.bool)
    keep[:, p0:p0 + side * (side + 1)].view(B, side, side + 1)[:, :, :side] = False
    assert torch.equal(hd.cpu()[keep], hidden[keep])


@pytest.mark.parametrize("name,dt", DTYPES)
def test_gemm_splitk_fp32_accumulate(dev, name, dt):
    ops, L = _ops()
    g = torch.Generator().manual_seed(9)
    M, N, K = 256, 136, 64 * 37
    a, w = _rand(g, M, K).to(dt), _rand(g, N, K).to(dt)
    old = _rand(g, M, N)
    ref = a.float() @ w.float().T
    out = ops.k_gemm(a.to(dev), w.to(dev), out_dtype=torch.float32, split_k=8)
    assert out.dtype == torch.float32 and rel_err(out, ref) < 1e-5 * (1 if name == "fp32" else 1)
    acc = old.to(dev).clone()
    ops.k_gemm(a.to(dev), w.to(dev), out=acc, beta=1.0)
    assert rel_err(acc, ref + old) < 1e-5
    acc2 = old.to(dev).clone()
    ops.k_gemm(a.to(dev), w.to(dev), out=acc2, beta=1.0, split_k=4, alpha=2.0)
    assert rel_err(acc2, 2 * ref + old) < 1e-5


def test_gemm_rejects_bad_shapes(dev):
    ops, L = _ops()
    a = torch.zeros(16, 40, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(16, 40, device=dev, dtype=torch.bfloat16)
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm(a, w)  # K not a multiple of 64
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm(torch.zeros(4, 64), torch.zeros(8, 64))  # CPU tensors: no fallback


# ------------------------------------------------------------------------------------- data movement
@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("R,C", [(64, 64), (577, 1024), (100, 72), (5, 8)])
def test_transpose(dev, name, dt, R, C):
    ops, L = _ops()
    x = _rand(torch.Generator().manual_seed(R + C), R, C).to(dt)
    rp = (R + 63) // 64 * 64
    out = ops.k_transpose(x.to(dev), rp).cpu()
    assert torch.equal(out[:, :R], x.T.contiguous())
    assert (out[:, R:] == 0).all()


@pytest.mark.parametrize("name,dt", DTYPES)
def test_colsum_cast_tokenmean(dev, name, dt):
    ops, L = _ops()
    g = torch.Generator().manual_seed(3)
    x = _rand(g, 1000, 1032).to(dt)
    assert rel_err(ops.k_colsum(x.to(dev)), x.float().sum(0)) < 1e-5
    y = _rand(g, 333, 40)
    assert torch.equal(ops.k_cast(y.to(dev), torch.bfloat16).cpu(), y.to(torch.bfloat16))
    z = _rand(g, 3, 576, 1024).to(dt)
    assert rel_err(ops.k_token_mean(z.to(dev)), z.float().mean(1)) < TOL[name]
    assert rel_err(ops.k_segment_sum(x.to(dev)[:960], 96), x.float()[:960].view(10, 96, -1).sum(1)) < TOL[name]


# ------------------------------------------------------------------------------------------ layernorm
def _window_pos(rows, side, r):
    t = torch.arange(rows) % (side * side)
    y, x = t // side, t % side
    return (y % r) * r + (x % r)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("D,affine,with_add", [(1024, True, False), (1024, False, True), (1152, True, False),
                                               (1536, False, False), (384, True, False), (4096, True, False),
                                               (3072, True, False), (3072, False, True), (8192, True, False)])
def test_layernorm_fwd_bwd(dev, name, dt, D, affine, with_add):
    ops, L = _ops()
    g = torch.Generator().manual_seed(D)
    side, r, B = 8, 4, 3
    rows = B * side * side
    x = _rand(g, rows, D).to(dt)
    gamma, beta = 1 + 0.1 * _rand(g, D), 0.1 * _rand(g, D)
    add = _rand(g, r * r, D) if with_add else None
    xr = x.float().clone().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ar = add.clone().requires_grad_() if with_add else None
    xin = xr + ar[_window_pos(rows, side, r)] if with_add else xr
    ref = F.layer_norm(xin, (D,), gr if affine else None, br if affine else None, 1e-5)
    dy = _rand(g, rows, D).to(dt)
    ref.backward(dy.float())
    y, mean, rstd = ops.k_layernorm_fwd(x.to(dev), gamma.to(dev) if affine else None, beta.to(dev) if affine else None,
                                        1e-5, add=add.to(dev) if with_add else None, side=side, grid_r=r if with_add else 1)
    assert rel_err(y, ref) < TOL[name]
    if D > 4096:      # the backward holds a row in registers: D <= 4096 (ConvNeXt-XXL stage 4 trains at 3072)
        return
    dx, dg, db, da = ops.k_layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma=gamma.to(dev) if affine else None,
                                         add=add.to(dev) if with_add else None, side=side, grid_r=r if with_add else 1)
    assert rel_err(dx, xr.grad) < TOL[name]
    if affine:
        assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    if with_add:
        assert rel_err(da, ar.grad) < 1e-4
    # fp32 accumulate variant
    acc = torch.full((rows, D), 0.5, device=dev)
    ops.k_layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma=gamma.to(dev) if affine else None,
                        add=add.to(dev) if with_add else None, side=side, grid_r=r if with_add else 1, dx_acc=acc)
    assert rel_err(acc, xr.grad + 0.5) < (1e-5 if name == "fp32" else 1e-2)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("D", [4096, 5120, 7168, 3072])
def test_rmsnorm(dev, name, dt, D):
    ops, L = _ops()
    g = torch.Generator().manual_seed(D)
    x = _rand(g, 70, D).to(dt)
    w = 1 + 0.1 * _rand(g, D)
    xr, wr = x.float().clone().requires_grad_(), w.clone().requires_grad_()
    ref = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))  # train_fsdp.py:1429-1438
    dy = _rand(g, 70, D).to(dt)
    ref.backward(dy.float())
    xd, wd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    y = ops.rmsnorm(xd, wd, 1e-5)
    y.backward(dy.to(dev))
    assert rel_err(y, ref) < TOL[name]
    assert rel_err(xd.grad, xr.grad) < TOL[name]
    assert rel_err(wd.grad, wr.grad) < 1e-4


@pytest.mark.parametrize("name,dt", DTYPES)
def test_rope(dev, name, dt):
    ops, L = _ops()
    g = torch.Generator().manual_seed(2)
    B, S, H, Dh, base = 2, 50, 8, 128, 500000.0
    x = _rand(g, B * S, H, Dh).to(dt)
    pos = torch.randint(0, 2048, (B, S), generator=g)
    inv = 1.0 / (base ** (torch.arange(0, Dh, 2).float() / Dh))  # phi3/modeling_phi3.py:127-131
    fr = pos.reshape(-1, 1).float() * inv[None, :]
    cos, sin = torch.cat([fr, fr], -1).cos()[:, None, :], torch.cat([fr, fr], -1).sin()[:, None, :]
    xf = x.float()
    rot = torch.cat([-xf[..., Dh // 2:], xf[..., :Dh // 2]], -1)
    ref = xf * cos + rot * sin
    c, s = ops.rope_table(pos.to(dev), Dh, base)
    xd = x.to(dev).requires_grad_()
    y = ops.rope(xd, c, s)
    assert rel_err(y, ref) < (1e-4 if name == "fp32" else TOL[name])
    y.backward(y.detach())  # R^T R x = x
    assert rel_err(xd.grad, xf) < (1e-4 if name == "fp32" else 2e-2)


# --------------------------------------------------------------------------------------- SVA attention
def _sva_ref(q, kvs_win, masks, heads, hd):
    """q [Bq,C]; kvs_win[i] [Bq, s2, 2C] window-major; masks[i] bool [Bq,s2]."""
    Bq, C = q.shape
    k = torch.cat([kv[..., :C] for kv in kvs_win], 1).view(Bq, -1, heads, hd).transpose(1, 2)
    v = torch.cat([kv[..., C:] for kv in kvs_win], 1).view(Bq, -1, heads, hd).transpose(1, 2)
    m = torch.cat(masks, 1)[:, None, None, :]
    qh = q.view(Bq, 1, heads, hd).transpose(1, 2)
    s = (qh @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~m, float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(Bq, C)


def _to_window_major(x, B, qside, r):
    C = x.shape[-1]
    return x.view(B, qside, r, qside, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B * qside * qside, r * r, C)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("window_major", [False, True])
def test_sva_attention_fwd_bwd(dev, name, dt, window_major):
    ops, L = _ops()
    g = torch.Generator().manual_seed(17)
    B, qside, heads, hd, r_list = 2, 6, 16, 64, [1, 1, 1, 4]
    C, Bq = heads * hd, B * qside * qside
    q = _rand(g, Bq, C).to(dt)
    kv_tm = [_rand(g, B * (qside * r) ** 2, 2 * C).to(dt) for r in r_list]  # tower-token-major
    masks = [torch.rand(Bq, r * r, generator=g) > 0.3 for r in r_list]
    for m in masks:
        m[m.sum(1) == 0] = True  # the collator guarantee (train_fsdp.py:1136)
    dout = _rand(g, Bq, C).to(dt)
    qr = q.float().clone().requires_grad_()
    kvr = [t.float().clone().requires_grad_() for t in kv_tm]
    ref = _sva_ref(qr, [_to_window_major(t, B, qside, r) for t, r in zip(kvr, r_list)], masks, heads, hd)
    ref.backward(dout.float())
    if window_major:
        kv_in = [_to_window_major(t, B, qside, r).reshape(-1, 2 * C).contiguous() for t, r in zip(kv_tm, r_list)]
        dref = [_to_window_major(t.grad, B, qside, r).reshape(-1, 2 * C) for t, r in zip(kvr, r_list)]
    else:
        kv_in, dref = kv_tm, [t.grad for t in kvr]
    qd = q.to(dev).requires_grad_()
    kvd = [t.to(dev).requires_grad_() for t in kv_in]
    mu8 = [m.to(torch.uint8).to(dev).contiguous() for m in masks]
    out = ops.sva_attention(qd, kvd, mu8, r_list, B, qside, heads, hd, window_major=window_major)
    out.backward(dout.to(dev))
    tol = TOL[name] * (5 if name == "fp32" else 1)
    assert rel_err(out, ref) < tol
    assert rel_err(qd.grad, qr.grad) < tol
    for a, b in zip(kvd, dref):
        assert rel_err(a.grad, b) < tol


# ------------------------------------------------------------------------------------- embedding splice
@pytest.mark.parametrize("name,dt", DTYPES)
def test_embed_splice_bit_exact(dev, name, dt):
    ops, L = _ops()
    g = torch.Generator().manual_seed(23)
    B, S, H, V, side = 3, 100, 128, 50, 4
    span = side * (side + 1)
    ids = torch.randint(1, V, (B, S), generator=g)
    ppos = [5, 60, -1]
    for b, p in enumerate(ppos):
        if p >= 0:
            ids[b, p] = -200
            ids[b, p + 1:p + span] = 0
    table, feat, nl = _rand(g, V, H).to(dt), _rand(g, B, side * side, H).to(dt), _rand(g, H).to(dt)
    # reference arithmetic: cambrian_arch.py:413-420 + :457-490
    vis = torch.cat([feat.view(B, side, side, H), nl.view(1, 1, 1, H).expand(B, side, 1, H)], 2).flatten(1, 2)
    emb = table[torch.where(ids == -200, 0, ids)]
    ref = []
    for b, p in enumerate(ppos):
        ref.append(emb[b] if p < 0 else torch.cat([emb[b, :p], vis[b], emb[b, p + span:]]))
    ref = torch.stack(ref)
    fd, nd = feat.to(dev).requires_grad_(), nl.to(dev).float().requires_grad_()
    out, pos = ops.embed_splice(ids.to(dev), table.to(dev), fd, nd, side)
    assert pos.cpu().tolist() == ppos
    assert torch.equal(out.cpu(), ref)  # copies only: bit-exact
    dout = _rand(g, B, S, H).to(dt)
    out.backward(dout.to(dev))
    dfeat_ref = torch.zeros(B, side * side, H)
    dnl_ref = torch.zeros(H)
    for b, p in enumerate(ppos):
        if p < 0:
            continue
        blk = dout[b, p:p + span].float().view(side, side + 1, H)
        dfeat_ref[b] = blk[:, :side].reshape(side * side, H)
        dnl_ref += blk[:, side].sum(0)
    assert torch.equal(fd.grad.cpu().float(), dfeat_ref.to(dt).float())
    assert rel_err(nd.grad, dnl_ref) < 1e-5


# ------------------------------------------------------------------------------------------ ViT attention
@pytest.mark.parametrize("N,heads,hd", [(577, 4, 64), (730, 3, 64), (729, 2, 96), (197, 12, 64), (64, 1, 64)])
def test_vit_attention(dev, N, heads, hd):
    ops, L = _ops()
    from cambrian_amd.model.multimodal_encoder import vit_ops
    g = torch.Generator().manual_seed(N)
    B = 2
    qkv = _rand(g, B * N, 3 * heads * hd)
    scale = 1 / math.sqrt(72 if hd == 96 else hd)
    q, k, v = [t.view(B, N, heads, hd).transpose(1, 2) for t in qkv.to(torch.bfloat16).float().chunk(3, -1)]
    ref = F.scaled_dot_product_attention(q, k, v, scale=scale).transpose(1, 2).reshape(B * N, heads * hd)
    out_bf = vit_ops.k_vit_attn(qkv.to(dev, torch.bfloat16), B, N, heads, hd, scale)
    out_simple = vit_ops.k_vit_attn(qkv.to(dev, torch.bfloat16), B, N, heads, hd, scale, force_simple=True)
    assert rel_err(out_bf, ref) < TOL["bf16"]
    assert rel_err(out_simple, ref) < TOL["bf16"]
    q, k, v = [t.view(B, N, heads, hd).transpose(1, 2) for t in qkv.chunk(3, -1)]
    ref32 = F.scaled_dot_product_attention(q, k, v, scale=scale).transpose(1, 2).reshape(B * N, heads * hd)
    out32 = vit_ops.k_vit_attn(qkv.to(dev), B, N, heads, hd, scale)
    assert rel_err(out32, ref32) < 1e-4


# ---------------------------------------------------------------------------------- conv-side kernels
@pytest.mark.parametrize("name,dt", DTYPES)
def test_patchify_and_conv_equivalence(dev, name, dt):
    ops, L = _ops()
    from cambrian_amd.model.multimodal_encoder import vit_ops
    g = torch.Generator().manual_seed(31)
    B, C, Hh, p, D = 2, 3, 56, 14, 64
    img = _rand(g, B, C, Hh, Hh)
    w, b = _rand(g, D, C, p, p, scale=0.05), _rand(g, D)
    ref = F.conv2d(img.to(dt).float(), w.to(dt).float(), b, stride=p).flatten(2).transpose(1, 2).reshape(-1, D)
    K = C * p * p
    Kpad = (K + 63) // 64 * 64
    cols = vit_ops.k_patchify(img.to(dev), p, Kpad, dt)
    wp = torch.zeros(D, Kpad)
    wp[:, :K] = w.reshape(D, K)
    out = ops.k_gemm(cols, wp.to(dev, dt), bias=b.to(dev))
    assert rel_err(out, ref) < TOL[name]
    # 2x2/2 NHWC patch gather == stride-2 conv (ConvNeXt downsample)
    x = _rand(g, B, 8, 8, 64).to(dt)
    w2 = _rand(g, 128, 64, 2, 2, scale=0.1).to(dt)
    ref2 = F.conv2d(x.float().permute(0, 3, 1, 2), w2.float(), stride=2).permute(0, 2, 3, 1).reshape(-1, 128)
    cols2 = vit_ops.k_patchify2x2(x.to(dev))
    out2 = ops.k_gemm(cols2, w2.permute(0, 2, 3, 1).reshape(128, 256).contiguous().to(dev))
    assert rel_err(out2, ref2) < TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 13, 10, 72),     # C % 64 != 0: direct kernel
                                       (2, 13, 10, 128),    # LDS-tiled kernel, ragged tile edges in x and y
                                       (1, 32, 48, 192),    # several full tiles
                                       (3, 7, 5, 64)])      # image smaller than one tile
def test_dwconv7x7(dev, name, dt, B, Hh, Ww, C):
    from cambrian_amd.model.multimodal_encoder import vit_ops
    g = torch.Generator().manual_seed(37 + C + Hh)
    x = _rand(g, B, Hh, Ww, C).to(dt)
    w, b = _rand(g, C, 1, 7, 7, scale=0.1), _rand(g, C)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    out = vit_ops.k_dwconv7x7(x.to(dev), w.view(C, 49).T.contiguous().to(dev), b.to(dev))
    assert rel_err(out, ref) < TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("hi,ho", [(27, 24), (16, 6), (8, 12), (24, 24)])
def test_resample_bilinear(dev, name, dt, hi, ho):
    from cambrian_amd.model.multimodal_encoder import vit_ops
    g = torch.Generator().manual_seed(hi * ho)
    B, C = 2, 72
    x = _rand(g, B, hi * hi, C).to(dt)
    # clip_encoder.py:70-96: NHWC -> NCHW, fp32 bilinear (align_corners=False), back
    ref = F.interpolate(x.float().view(B, hi, hi, C).permute(0, 3, 1, 2), size=(ho, ho), mode="bilinear",
                        align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    out = torch.zeros(B, ho * ho, C + 16, dtype=dt, device=dev)
    vit_ops.k_resample(x.to(dev), hi, hi, out, ho, ho, col_offset=8)
    assert rel_err(out[:, :, 8:8 + C], ref) < (1e-5 if name == "fp32" else 1e-2)
    assert (out[:, :, :8] == 0).all() and (out[:, :, 8 + C:] == 0).all()


@pytest.mark.parametrize("name,dt", DTYPES)
def test_act_kernels(dev, name, dt):
    ops, L = _ops()
    from cambrian_amd.model.multimodal_encoder import vit_ops
    g = torch.Generator().manual_seed(41)
    h = _rand(g, 50, 256).to(dt)
    ref = F.silu(h.float()[:, :128]) * h.float()[:, 128:]  # Dinov2SwiGLUFFN
    out = vit_ops.k_act_mul(h.to(dev)[:, :128], h.to(dev)[:, 128:], L.ACT_SILU)
    assert rel_err(out, ref) < TOL[name]
    pre = h.float().clone().requires_grad_()
    F.gelu(pre).backward(torch.ones_like(pre))
    hd, ones = h.to(dev), torch.ones_like(h, device=dev)  # keep the device buffers alive across the raw call
    dx = torch.empty_like(hd)
    rc = L.load().cmb_act_bwd(L.dtype_code(dt), L.ACT_GELU_ERF, ones.data_ptr(), hd.data_ptr(), hd.numel(),
                              dx.data_ptr(), L.stream_ptr(dev))
    assert rc == 0 and rel_err(dx, pre.grad) < TOL[name]


# ------------------------------------------------------------------------------------------------ LLM-side kernels
@pytest.mark.parametrize("name,dt", DTYPES)
def test_fused_cross_entropy_matches_torch(dev, name, dt):
    """cambrian_llama.py:409-422: logits.float() -> shift -> CrossEntropyLoss(ignore_index=-100), forward value and
    d loss / d logits, against torch on the same (dtype-rounded) logits."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(21)
    B, S, V = 3, 37, 1000
    logits = (_rand(g, B, S, V) * 3).to(dt)
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[:, :5] = -100
    labels[1, 20:25] = -100
    ref_in = logits.float().clone().requires_grad_()
    ref = F.cross_entropy(ref_in[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    (ref * 1.7).backward()
    x = logits.detach().to(dev).requires_grad_()
    shift = torch.full_like(labels, -100)
    shift[:, :-1] = labels[:, 1:]
    for inplace in (False, True):
        xin = x.detach().clone().requires_grad_()
        y = xin * 1.0  # non-leaf, so the in-place backward may overwrite it
        loss = ops.cross_entropy(y.view(B * S, V), shift.view(-1).to(dev), -100, inplace=inplace)
        assert abs(loss.item() - ref.item()) < (1e-5 if name == "fp32" else 1e-4) * max(1.0, abs(ref.item()))
        (loss * 1.7).backward()
        assert rel_err(xin.grad.view(B, S, V), ref_in.grad) < (TOL[name] if name == "fp32" else 1e-2)
        assert torch.count_nonzero(xin.grad[:, -1]) == 0  # last position of every sequence is never scored


@pytest.mark.parametrize("name,dt", DTYPES)
def test_swiglu_fwd_bwd(dev, name, dt):
    ops, L = _ops()
    g_ = torch.Generator().manual_seed(22)
    a, b = (_rand(g_, 5, 33, 64) * 2).to(dt), _rand(g_, 5, 33, 64).to(dt)
    ar, br = a.float().clone().requires_grad_(), b.float().clone().requires_grad_()
    ref = F.silu(ar) * br
    w = _rand(g_, 5, 33, 64)
    (ref * w).sum().backward()
    ad, bd = a.detach().to(dev).requires_grad_(), b.detach().to(dev).requires_grad_()
    out = ops.swiglu(ad, bd)
    (out.float() * w.to(dev)).sum().backward()
    tol = TOL[name]
    assert rel_err(out, ref) < tol
    assert rel_err(ad.grad, ar.grad) < tol
    assert rel_err(bd.grad, br.grad) < tol


@pytest.mark.parametrize("name,dt", DTYPES)
def test_qkv_rope_split_and_merge(dev, name, dt):
    """Packed QKV -> (q, k, v) head-major with RoPE == separate views + rotate-half RoPE (phi3/modeling_phi3.py:257-281)
    + transposes; the backward is checked through autograd against the same torch graph."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(31)
    B, S, nh, nkv, hd = 2, 19, 4, 2, 32
    packed = _rand(g, B, S, (nh + 2 * nkv) * hd).to(dt)
    pos = torch.randint(0, 500, (B, S), generator=g)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = pos.reshape(-1, 1).float() * inv[None]
    cos_r, sin_r = torch.cos(ang), torch.sin(ang)                     # [B*S, hd/2]

    def rot(x):                                                        # x [B,S,H,hd]
        x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
        c, s_ = cos_r.view(B, S, 1, hd // 2), sin_r.view(B, S, 1, hd // 2)
        return torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_], dim=-1)

    pr = packed.float().clone().requires_grad_()
    qr = rot(pr[..., : nh * hd].view(B, S, nh, hd)).transpose(1, 2)
    kr = rot(pr[..., nh * hd:(nh + nkv) * hd].view(B, S, nkv, hd)).transpose(1, 2)
    vr = pr[..., (nh + nkv) * hd:].view(B, S, nkv, hd).transpose(1, 2)
    wq, wk, wv = _rand(g, B, nh, S, hd), _rand(g, B, nkv, S, hd), _rand(g, B, nkv, S, hd)
    ((qr * wq).sum() + (kr * wk).sum() + (vr * wv).sum()).backward()

    cos, sin = ops.rope_table(pos.to(dev), hd, 10000.0)
    pd = packed.detach().to(dev).requires_grad_()
    q, k, v = ops.qkv_rope(pd, cos, sin, nh, nkv, hd)
    assert q.shape == (B, nh, S, hd) and k.shape == (B, nkv, S, hd) and v.transpose(1, 2).is_contiguous()
    ((q.float() * wq.to(dev)).sum() + (k.float() * wk.to(dev)).sum() + (v.float() * wv.to(dev)).sum()).backward()
    tol = TOL[name] * (5 if name == "fp32" else 1)
    assert rel_err(q, qr) < tol and rel_err(k, kr) < tol
    assert torch.equal(v.cpu().float(), vr.detach().to(dt).float())    # v is a pure copy: bit-exact
    assert rel_err(pd.grad, pr.grad) < tol


@pytest.mark.parametrize("name,dt", DTYPES)
def test_swiglu_packed(dev, name, dt):
    ops, L = _ops()
    g_ = torch.Generator().manual_seed(23)
    gu = (_rand(g_, 3, 17, 128) * 2).to(dt)
    r = gu.float().clone().requires_grad_()
    ref = F.silu(r[..., :64]) * r[..., 64:]
    w = _rand(g_, 3, 17, 64)
    (ref * w).sum().backward()
    x = gu.detach().to(dev).requires_grad_()
    out = ops.swiglu_packed(x)
    (out.float() * w.to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL[name] and rel_err(x.grad, r.grad) < TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("frozen", [True, False])
def test_add_rmsnorm_fused(dev, name, dt, frozen):
    """(x + d, rmsnorm(x + d) * w) and its backward vs the unfused torch graph (the sum is rounded to the storage type
    before it is normalised, exactly as the separate add + norm do)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(41)
    rows, D = 37, 4096
    x, d = _rand(g, 3, rows, D).to(dt), _rand(g, 3, rows, D).to(dt)
    w = (1 + 0.1 * _rand(g, D))
    ws, wy = _rand(g, 3, rows, D), _rand(g, 3, rows, D)
    xr, dr = x.float().clone().requires_grad_(), d.float().clone().requires_grad_()
    wr = w.clone().requires_grad_(not frozen)
    sr = (xr + dr).to(dt).float() if dt != torch.float32 else xr + dr
    sr_g = xr + dr  # gradient path of the sum (the rounding has zero derivative almost everywhere: straight-through)
    sr = sr_g + (sr - sr_g).detach()
    yr = wr * (sr * torch.rsqrt(sr.pow(2).mean(-1, keepdim=True) + 1e-5))
    ((sr * ws).sum() + (yr * wy).sum()).backward()
    xd, dd = x.detach().to(dev).requires_grad_(), d.detach().to(dev).requires_grad_()
    wd = w.to(dev).requires_grad_(not frozen)
    s, y = ops.add_rmsnorm(xd, dd, wd, 1e-5)
    ((s.float() * ws.to(dev)).sum() + (y.float() * wy.to(dev)).sum()).backward()
    tol = TOL[name]
    assert rel_err(s, sr) < tol and rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol and torch.equal(xd.grad, dd.grad)
    if not frozen:
        assert rel_err(wd.grad, wr.grad) < (tol if name == "fp32" else 3e-2)


# ------------------------------------------------------------------------------------------------ kernel-selection knobs
# cmb_knob_set picks between kernels that compute the SAME function (round 4): every pair is checked against each other here
# and — through the tests above, which run with the library's defaults — against the fp32 restatement.
@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("rows,D", [(1, 384), (7, 1536), (4099, 1536), (1000, 3072), (513, 1152), (2049, 4096), (300, 8192)])
def test_layernorm_fwd_variants_bit_equal(dev, name, dt, rows, D):
    """LDS-parameter / prefetching LayerNorm forward == the one-row-at-a-time kernel, bit for bit (same arithmetic and
    summation order), statistics included, for row strides larger than D and row counts that leave ragged waves."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(rows + D)
    x = _rand(g, rows, D + 8).to(dt).to(dev)[:, :D]
    gamma, beta = (1 + 0.1 * _rand(g, D)).to(dev), (0.1 * _rand(g, D)).to(dev)
    outs = []
    try:
        for var in (0, 1, 3):
            L.knob_set(L.KNOB_LN_FWD, var)
            outs.append(ops.k_layernorm_fwd(x, gamma, beta, 1e-6))
    finally:
        L.knob_set(L.KNOB_LN_FWD, 1)
    for y, mean, rstd in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(mean, outs[0][1]) and torch.equal(rstd, outs[0][2])
    ref = F.layer_norm(x.float().cpu(), (D,), gamma.cpu(), beta.cpu(), 1e-6)
    assert rel_err(outs[1][0], ref) < TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 13, 10, 128), (1, 32, 48, 192), (3, 7, 5, 64), (1, 70, 33, 64), (2, 64, 64, 128)])
def test_dwconv7x7_variants_bit_equal(dev, name, dt, B, Hh, Ww, C):
    """Column-walking depthwise conv (register-resident taps, rows in chunks of 32 / 64 / 5) == the LDS-tiled kernel, bit for
    bit: same tap order per output element; ragged widths (partial strips, strips beyond the map), maps shorter than a chunk,
    chunk boundaries inside the map."""
    from cambrian_amd.model.multimodal_encoder import vit_ops
    ops, L = _ops()
    g = torch.Generator().manual_seed(41 + C + Hh)
    x = _rand(g, B, Hh, Ww, C).to(dt).to(dev)
    w, b = _rand(g, C, 1, 7, 7, scale=0.1), _rand(g, C)
    w49 = w.view(C, 49).T.contiguous().to(dev)
    outs = []
    try:
        for var in (0, 1, 64, 5):
            L.knob_set(L.KNOB_DWCONV, var)
            outs.append(vit_ops.k_dwconv7x7(x, w49, b.to(dev)))
    finally:
        L.knob_set(L.KNOB_DWCONV, 1)
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    assert rel_err(outs[1], ref) < TOL[name]


VIT_ATTN_DEFAULT = 2   # the library's default (include/cambrian_amd.h CMB_KNOB_DEFAULTS)


@pytest.mark.parametrize("N,heads,hd", [(577, 16, 64), (730, 24, 64), (729, 16, 96), (64, 2, 64), (65, 1, 96), (128, 3, 64), (1, 2, 64)])
def test_vit_attn_variants(dev, N, heads, hd):
    """One-barrier (double-buffered) tower attention == the two-barrier kernel bit for bit (same instruction stream per tile),
    and both against the fp32 one-wave-per-query kernel."""
    from cambrian_amd.model.multimodal_encoder import vit_ops
    ops, L = _ops()
    g = torch.Generator().manual_seed(N + hd)
    B = 2
    qkv = _rand(g, B * N, 3 * heads * hd).to(torch.bfloat16).to(dev)
    outs = []
    try:
        for var in (0, 1, 2, 3):
            L.knob_set(L.KNOB_VIT_ATTN, var)
            outs.append(vit_ops.k_vit_attn(qkv, B, N, heads, hd, hd ** -0.5))
    finally:
        L.knob_set(L.KNOB_VIT_ATTN, VIT_ATTN_DEFAULT)
    assert torch.equal(outs[0], outs[1])
    want = vit_ops.k_vit_attn(qkv.float(), B, N, heads, hd, hd ** -0.5)
    assert rel_err(outs[1], want) < 1e-2
    # round 6: the LDS-DMA / transposing-read kernel — same products and softmax, the row sum in two partial sums and the
    # exponent argument as one fused multiply-add: equal to the register-staged kernel to bf16 rounding, not bit for bit
    for o in outs[2:]:   # 2: 128 queries per workgroup, 3: 256
        assert rel_err(o, want) < 1e-2
        assert rel_err(o, outs[1]) < 8e-3
        assert torch.isfinite(o.float()).all()


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("L_,D,side,r,with_acc", [(13, 1024, 8, 4, True), (3, 1024, 6, 2, False), (17, 1024, 4, 1, False),
                                                    (5, 512, 8, 4, True), (2, 384, 4, 2, False)])
@pytest.mark.parametrize("chunk", [4, 5, 6, 7])
def test_layernorm_bwd_multi_equals_layer_by_layer(dev, name, dt, L_, D, side, r, with_acc, chunk):
    """cmb_layernorm_bwd_multi (one pass over x for all layers: the SVA layers' deferred LayerNorm backwards) against
    cmb_layernorm_bwd run layer by layer into the fp32 accumulator, and against autograd of F.layer_norm: d(x) and every
    position table's gradient; grid_r = 1 layers carry no table; more layers than one launch holds (17 > 16)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(L_ * D + side)
    B = 3
    rows = B * side * side
    x = _rand(g, rows, D).to(dt)
    adds = [(_rand(g, r * r, D) if r > 1 else None) for _ in range(L_)]
    dys = [_rand(g, rows, D).to(dt) for _ in range(L_)]
    acc0 = _rand(g, rows, D) if with_acc else None
    # autograd reference
    xr = x.float().clone().requires_grad_()
    ars = [None if a is None else a.clone().requires_grad_() for a in adds]
    wp = _window_pos(rows, side, r)
    tot = 0.0
    for a, dy in zip(ars, dys):
        xin = xr if a is None else xr + a[wp]
        tot = tot + (F.layer_norm(xin, (D,), None, None, 1e-5) * dy.float()).sum()
    tot.backward()
    ref_dx = xr.grad + (acc0 if with_acc else 0.0)
    xd = x.to(dev)
    items, seq_acc = [], (acc0.clone().to(dev) if with_acc else torch.zeros(rows, D, device=dev))
    dadd = []
    for a, dy in zip(adds, dys):
        ad = None if a is None else a.to(dev)
        _, mean, rstd = ops.k_layernorm_fwd(xd, None, None, 1e-5, add=ad, side=side, grid_r=r if r > 1 else 1)
        slot = -1
        if ad is not None:
            slot = len(dadd)
            dadd.append(torch.zeros(r * r, D, device=dev))
        items.append((dy.to(dev), mean, rstd, ad, slot))
        ops.k_layernorm_bwd(dy.to(dev), xd, mean, rstd, add=ad, side=side, grid_r=r if r > 1 else 1, dx_acc=seq_acc)
    dx = acc0.clone().to(dev) if with_acc else torch.empty(rows, D, device=dev)
    saved = L.knob_get(L.KNOB_LN_MULTI_CHUNK)
    try:
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, chunk)          # layers per launch: every instantiation (2 / 4 / 5 / 6 / 7 sets of sums)
        ops.k_layernorm_bwd_multi(xd, items, side, r if r > 1 else 1, dx, with_acc, dadd)
    finally:
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, saved)
    assert rel_err(dx, ref_dx) < TOL[name]
    assert rel_err(dx, seq_acc) < 2e-6                       # same arithmetic, different summation order across layers
    # dx_out: the call's last launch writes the finished sum in x's dtype instead of updating the fp32 accumulator
    dx2 = acc0.clone().to(dev) if with_acc else torch.empty(rows, D, device=dev)
    out16 = torch.empty(rows, D, device=dev, dtype=dt)
    dadd2 = [torch.zeros_like(t) for t in dadd]
    try:
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, chunk)
        ops.k_layernorm_bwd_multi(xd, items, side, r if r > 1 else 1, dx2, with_acc, dadd2, dx_out=out16)
    finally:
        L.knob_set(L.KNOB_LN_MULTI_CHUNK, saved)
    assert torch.equal(out16, dx.to(dt))                     # the same fp32 sum, rounded once
    for a, b in zip(dadd2, dadd):
        assert rel_err(a, b) < 2e-6
    k = 0
    for a in ars:
        if a is not None:
            assert rel_err(dadd[k], a.grad) < TOL[name]
            k += 1


def test_colsum_scaled_matches_broadcast_product(dev):
    """cmb_colsum_scaled: sum_r scale[r, c // 64] * x[r, c] — the bias gradients of the absorbed SVA projections (ops.py
    SvaAbsorbedFn.backward) in one pass."""
    import torch
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(3)
    for R in (1, 777, 13824):
        x = torch.randn(R, 1024, generator=g).to(dev, torch.bfloat16)
        sc = torch.randn(R, 16, generator=g).to(dev)
        got = ops.k_colsum(x, row_scale=sc, group=64)
        want = (sc.double()[:, :, None] * x.double().view(R, 16, 64)).sum(0).reshape(1024)
        assert ((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-6)).item() < 1e-5


@pytest.mark.parametrize("wgs", [0, 64, 512, 4096])
def test_colsum_workgroup_targets(dev, wgs):
    """CMB_KNOB_COLSUM_WGS only changes how the rows are cut into workgroups: ragged row counts, column counts that are not a
    multiple of 512, strided inputs, plain and scaled sums against fp64."""
    import torch
    from cambrian_amd import lib as L, ops
    g = torch.Generator().manual_seed(11)
    saved = L.knob_get(L.KNOB_COLSUM_WGS)
    try:
        L.knob_set(L.KNOB_COLSUM_WGS, wgs)
        for R, C_ in ((1, 64), (19, 1032), (13824, 1024), (5000, 2048)):
            wide = torch.randn(R, C_ + 64, generator=g).to(dev, torch.bfloat16)
            x = wide[:, :C_]                                   # row stride C + 64
            want = x.double().sum(0)
            got = ops.k_colsum(x)
            assert ((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-6)).item() < 2e-5
            if C_ % 64 == 0:
                sc = torch.randn(R, C_ // 64, generator=g).to(dev)
                want = (sc.double()[:, :, None] * x.double().reshape(R, C_ // 64, 64)).sum(0).reshape(C_)
                got = ops.k_colsum(x, row_scale=sc, group=64)
                assert ((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-6)).item() < 2e-5
    finally:
        L.knob_set(L.KNOB_COLSUM_WGS, saved)


@pytest.mark.parametrize("rpb", [4, 16, 64, 256])
def test_layernorm_bwd_rows_per_workgroup(dev, rpb):
    """CMB_KNOB_LN_BWD_ROWS: the same gradients whatever the number of rows a workgroup sums before its atomics."""
    import torch
    from cambrian_amd import lib as L, ops
    g = torch.Generator().manual_seed(12)
    rows, D = 1357, 1024
    x = torch.randn(rows, D, generator=g).to(dev, torch.bfloat16)
    dy = torch.randn(rows, D, generator=g).to(dev, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = torch.zeros(D, device=dev, requires_grad=True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy.float())
    _, mean, rstd = ops.k_layernorm_fwd(x, gamma, torch.zeros(D, device=dev), 1e-5)
    saved = L.knob_get(L.KNOB_LN_BWD_ROWS)
    try:
        L.knob_set(L.KNOB_LN_BWD_ROWS, rpb)
        dx, dgamma, dbeta, _ = ops.k_layernorm_bwd(dy, x, mean, rstd, gamma=gamma)
    finally:
        L.knob_set(L.KNOB_LN_BWD_ROWS, saved)
    assert rel_err(dx.float().cpu(), xr.grad.cpu()) < 2e-2
    assert rel_err(dgamma.cpu(), gr.grad.cpu()) < 1e-4
    assert rel_err(dbeta.cpu(), br.grad.cpu()) < 1e-4
