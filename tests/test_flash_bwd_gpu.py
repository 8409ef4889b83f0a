"""`-m gpu`: cmb_flash_attn_bwd (flash_bwd.hip) — gradients of causal grouped-query attention against an fp32
restatement of softmax(QK^T/sqrt(d) + causal)V differentiated by autograd, at a small and at the LLM's shape; and the
log-sum-exp convention of the stock forward it consumes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, g):
    """fp32 math on the same bf16-rounded inputs; q [B,H,S,D], k / v [B,HKV,S,D]."""
    qf, kf, vf = q.float(), k.float().repeat_interleave(g, 1), v.float().repeat_interleave(g, 1)
    S = q.shape[2]
    s = qf @ kf.transpose(-1, -2) / math.sqrt(q.shape[-1])
    s = s.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=q.device).tril_(), float("-inf"))
    return torch.softmax(s, -1) @ vf, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,S,H,HKV", [(2, 256, 8, 2), (1, 384, 4, 4), (2, 2048, 32, 8)])
def test_flash_bwd_matches_fp32_autograd(dev, B, S, H, HKV):
    from cambrian_amd import ops
    g_ = torch.Generator().manual_seed(S + H)
    D = 128
    # token-major storage, as ops.qkv_rope hands the tensors to the attention
    qs = torch.randn(B, S, H, D, generator=g_).to(torch.bfloat16).to(dev)
    ks = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    vs = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    w = torch.randn(B, H, S, D, generator=g_).to(dev)
    q, k, v = (t.transpose(1, 2).detach().requires_grad_() for t in (qs, ks, vs))
    assert ops.causal_attention_supported(q, k)
    out = ops.causal_attention(q, k, v)
    qr, kr, vr = (t.transpose(1, 2).detach().float().requires_grad_() for t in (qs, ks, vs))
    ref, ref_lse = _ref(qr, kr, vr, H // HKV)
    assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 2e-2
    # forward log-sum-exp: natural log of sum exp(scale * q.k), and interchangeable with the stock flash kernel's
    lse = out.grad_fn.saved_tensors[4]
    assert lse.shape == (B, H, S) and (lse - ref_lse).abs().max().item() < 2e-2
    ke, ve = k.repeat_interleave(H // HKV, 1), v.repeat_interleave(H // HKV, 1)
    res = torch.ops.aten._scaled_dot_product_flash_attention(q, ke, ve, 0.0, True, False)
    assert (res[1] - lse).abs().max().item() < 2e-2
    assert ((res[0].float() - out.float()).abs().max() / ref.abs().max()).item() < 2e-2
    (out.float() * w).sum().backward()
    (ref * w).sum().backward()
    for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        err = ((a.float() - b).abs().max() / b.abs().max()).item()
        assert err < 3e-2, f"{name}: rel err {err}"
    # bit-reproducible (no atomics)
    q2, k2, v2 = (t.detach().clone().requires_grad_() for t in (q, k, v))
    (ops.causal_attention(q2, k2, v2).float() * w).sum().backward()
    assert torch.equal(q2.grad, q.grad) and torch.equal(k2.grad, k.grad) and torch.equal(v2.grad, v.grad)


def _collator_key_mask(B, S, image_sizes, position=91):
    """attention_mask of real collator batches (train_fsdp.py:1089-1165): padded rows / columns of the 24 x 25 visual span
    switched off, a padded tail on the last sample."""
    from cambrian_amd.train.data_layout import prepare_image_info
    m = torch.ones(B, S, dtype=torch.bool)
    for b, size in enumerate(image_sizes):
        vis, _ = prepare_image_info(size, 576, newline=True)          # bool [600]
        if position + 600 <= S:
            m[b, position:position + 600] = vis
    m[B - 1, S - 137:] = False
    return m


@pytest.mark.parametrize("B,S,H,HKV", [(2, 1024, 8, 2), (3, 2048, 32, 8)])
def test_flash_key_padding_mask_matches_fp32_autograd(dev, B, S, H, HKV):
    """causal AND key-padding (diagonal open), forward + backward, against fp32 SDPA-by-hand with the dense mask the
    reference's HF decoder materialises (cambrian_llama.py:142-166); masks from the collator's layout code for
    non-square images, plus a mask with whole 64-key tiles of padding (tile skipping) and an all-valid one."""
    from cambrian_amd import ops
    g_ = torch.Generator().manual_seed(S + H + 1)
    D, grp = 128, H // HKV
    sizes = [(336, 224), (224, 336), (1000, 90)][:B]
    masks = [_collator_key_mask(B, S, sizes), torch.ones(B, S, dtype=torch.bool)]
    holes = torch.ones(B, S, dtype=torch.bool)
    holes[0, 128:448] = False            # five whole key tiles of padding
    holes[1, :70] = False                # the sequence starts with padding (rows that see only themselves)
    masks.append(holes)
    qs = torch.randn(B, S, H, D, generator=g_).to(torch.bfloat16).to(dev)
    ks = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    vs = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    w = torch.randn(B, H, S, D, generator=g_).to(dev)
    for kvmask in masks:
        q, k, v = (t.transpose(1, 2).detach().requires_grad_() for t in (qs, ks, vs))
        out = ops.causal_attention(q, k, v, kvmask.to(dev))
        qr, kr, vr = (t.transpose(1, 2).detach().float().requires_grad_() for t in (qs, ks, vs))
        dense = (torch.ones(S, S, dtype=torch.bool, device=dev).tril_()[None, None] & kvmask.to(dev)[:, None, None, :]) \
            | torch.eye(S, dtype=torch.bool, device=dev)[None, None]
        sc = qr @ kr.repeat_interleave(grp, 1).transpose(-1, -2) / math.sqrt(D)
        ref = torch.softmax(sc.masked_fill(~dense, float("-inf")), -1) @ vr.repeat_interleave(grp, 1)
        assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 2e-2
        (out.float() * w).sum().backward()
        (ref * w).sum().backward()
        for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
            err = ((a.float() - b).abs().max() / b.abs().max()).item()
            assert err < 3e-2, f"{name}: rel err {err}"
        if not kvmask.all():             # a padded key receives gradient only from its own (ignored) query row
            pad = ~kvmask.to(dev)
            assert torch.isfinite(k.grad.float()).all() and torch.isfinite(out.float()).all()
            only_diag = (kr.grad.transpose(1, 2)[pad].abs().max() / kr.grad.abs().max()).item()
            assert ((k.grad.float().transpose(1, 2)[pad] - kr.grad.transpose(1, 2)[pad]).abs().max().item()
                    <= 3e-2 * kr.grad.abs().max().item()), only_diag


@pytest.mark.parametrize("request_frozen", [True, False])
def test_decoder_uses_hip_attention_for_collator_masks(dev, monkeypatch, request_frozen):
    """a real collator batch (non-trivial attention_mask) must stay on flash_bwd.hip in training mode —
    F.scaled_dot_product_attention is never reached."""
    import torch.nn.functional as F
    from cambrian_amd.model.language_model import cambrian_llama as CL
    cfg = CL.CambrianConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                            num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, max_position_embeddings=2048,
                            rope_theta=500000.0)
    torch.manual_seed(0)
    model = CL.LlamaBackbone(cfg, dev, torch.bfloat16)
    B, S = 2, 1024
    frozen = request_frozen
    for p_ in model.parameters():      # frozen = the pre-training stage (fused q|k|v GEMM); trainable = the finetune stage
        p_.requires_grad_(not frozen)
    x = torch.randn(B, S, 512, device=dev, dtype=torch.bfloat16, requires_grad=True)
    mask = _collator_key_mask(B, S, [(336, 224), (224, 336)]).to(dev)

    def boom(*a, **k):
        raise AssertionError("stock SDPA reached with a key-padding mask")

    cos, sin = __import__("cambrian_amd").ops.rope_table(torch.arange(S, device=dev)[None].expand(B, S), 128, 500000.0)
    ref = model.layers[0].self_attn(x, cos, sin, CL.KeyPadding(mask).dense())          # stock path, dense mask
    monkeypatch.setattr(F, "scaled_dot_product_attention", boom)
    out = model.layers[0].self_attn(x, cos, sin, CL.KeyPadding(mask))
    assert ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item() < 2e-2
    out.float().sum().backward()
    assert torch.isfinite(x.grad.float()).all()


def test_flash_bwd_speed_vs_stock(dev):
    """Not an assertion on speed (printed for the log): backward time of the HIP kernels vs PyTorch's SDPA backward."""
    from cambrian_amd import ops
    import torch.nn.functional as F
    B, S, H, HKV, D = 16, 2048, 32, 8, 128
    qs = torch.randn(B, S, H, D, device=dev).to(torch.bfloat16)
    ks = torch.randn(B, S, HKV, D, device=dev).to(torch.bfloat16)
    vs = torch.randn(B, S, HKV, D, device=dev).to(torch.bfloat16)
    q, k, v = (t.transpose(1, 2).requires_grad_() for t in (qs, ks, vs))

    def timed(fn):
        o = fn()
        gr = torch.ones_like(o)
        for _ in range(2):
            o.backward(gr, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o.backward(gr, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5

    t_hip = timed(lambda: ops.causal_attention(q, k, v))
    t_ref = timed(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True))
    print(f"\nflash bwd B={B}: HIP {t_hip:.3f} ms vs stock SDPA backward {t_ref:.3f} ms")

    def timed_fwd(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5

    with torch.no_grad():
        f_hip = timed_fwd(lambda: ops.causal_attention(q, k, v))
        f_ref = timed_fwd(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True))
    print(f"flash fwd B={B}: HIP {f_hip:.3f} ms vs stock SDPA forward (incl. K/V head expansion) {f_ref:.3f} ms")


@pytest.mark.parametrize("B,N,H,hd", [(2, 17, 2, 64), (2, 577, 4, 64), (1, 729, 3, 72), (2, 730, 2, 64), (1, 256, 2, 128)])
def test_vit_attention_bidirectional_padded(dev, B, N, H, hd):
    """ops.vit_attention (trainable towers): the decoder's flash kernels in their non-causal form, any
    token count (577 / 729 / 730) and head_dim (64 / 72) through zero padding + kv_len masking, forward and backward
    against fp32 autograd."""
    from cambrian_amd import ops
    g_ = torch.Generator().manual_seed(N + hd)
    q, k, v = (torch.randn(B, H, N, hd, generator=g_).to(torch.bfloat16).to(dev).requires_grad_() for _ in range(3))
    w = torch.randn(B, H, N, hd, generator=g_).to(dev)
    scale = hd ** -0.5
    out = ops.vit_attention(q, k, v, scale)
    assert out.shape == (B, H, N, hd)
    qr, kr, vr = (t.detach().float().requires_grad_() for t in (q, k, v))
    ref = torch.softmax(qr @ kr.transpose(-1, -2) * scale, -1) @ vr
    assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 2e-2
    (out.float() * w).sum().backward()
    (ref * w).sum().backward()
    for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        assert a.shape == b.shape
        err = ((a.float() - b).abs().max() / b.abs().max()).item()
        assert err < 3e-2, f"{name}: rel err {err}"


@pytest.mark.parametrize("masked", [False, True])
def test_every_shipped_flash_variant(dev, masked):
    """CMB_KNOB_FLASH: every variant the library ships — 0 (round-4 kernels), 4 (four-phase dK/dV body), 7
    (+ forward and dQ on LDS-DMA tiles), 23 (default: + transposing reads in dK/dV), and the single bits 1, 2, 16, 20 — against
    fp32 autograd of the plain formula, with dQ / dK / dV BIT-IDENTICAL across variants and the forward equal to fp32 rounding,
    as include/cambrian_amd.h claims; causal and causal + key-padding (the collator's mask)."""
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    B, S, H, HKV, D = 2, 1024, 8, 2, 128
    g_ = torch.Generator().manual_seed(11)
    qs = torch.randn(B, S, H, D, generator=g_).to(torch.bfloat16).to(dev)
    ks = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    vs = torch.randn(B, S, HKV, D, generator=g_).to(torch.bfloat16).to(dev)
    w = torch.randn(B, H, S, D, generator=g_).to(dev)
    key_valid = _collator_key_mask(B, S, [(336, 200), (224, 336)]).to(dev) if masked else None
    # fp32 reference
    qr, kr, vr = (t.transpose(1, 2).detach().float().requires_grad_() for t in (qs, ks, vs))
    s = qr @ kr.repeat_interleave(H // HKV, 1).transpose(-1, -2) / math.sqrt(D)
    allow = torch.ones(S, S, dtype=torch.bool, device=dev).tril_()[None, None]
    if masked:
        allow = allow & (key_valid[:, None, None, :] | torch.eye(S, dtype=torch.bool, device=dev)[None, None])
    ref = torch.softmax(s.masked_fill(~allow, float("-inf")), -1) @ vr.repeat_interleave(H // HKV, 1)
    (ref * w).sum().backward()
    results = {}
    default = L.load().cmb_knob_get(L.KNOB_FLASH)
    try:
        for knob in (0, 4, 7, 23, 1, 2, 16, 20):
            L.knob_set(L.KNOB_FLASH, knob)
            q, k, v = (t.transpose(1, 2).detach().requires_grad_() for t in (qs, ks, vs))
            out = ops.causal_attention(q, k, v, key_valid=key_valid)
            (out.float() * w).sum().backward()
            results[knob] = (out.detach(), q.grad, k.grad, v.grad)
            assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 2e-2, f"knob {knob}: forward"
            for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
                err = ((a.float() - b).abs().max() / b.abs().max()).item()
                assert err < 3e-2, f"knob {knob} {name}: rel err {err}"
        with pytest.raises(L.CambrianAmdError):
            L.knob_set(L.KNOB_FLASH, 8 | 2)      # the removed dK/dV variant is rejected, not silently mapped
    finally:
        L.knob_set(L.KNOB_FLASH, default)
    base = results[0]
    for knob, r in results.items():
        fwd_new = bool(knob & 1)
        if not fwd_new:
            assert torch.equal(r[0], base[0]), f"knob {knob}: forward differs from the round-4 kernel"
        else:   # two partial row sums: one bf16 ulp at most on an output
            assert (r[0].float() - base[0].float()).abs().max().item() <= 2.0 ** -7 * base[0].float().abs().max().item()
    # the backward kernels are bit-identical to each other GIVEN the same forward outputs (o, lse): compare within each forward
    for group in ((0, 4, 2, 16, 20), (7, 23, 1)):
        a = results[group[0]]
        for knob in group[1:]:
            for i, name in ((1, "dq"), (2, "dk"), (3, "dv")):
                assert torch.equal(results[knob][i], a[i]), f"knob {knob}: {name} differs from knob {group[0]}"
