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
