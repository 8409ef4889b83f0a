"""CPU: the absorbed form of the SVA attention core (cambrian_amd/csrc/sva_absorbed.hip, composed in
cambrian_amd/model/vision_sampler.py) is the reference's computation (vision_sampler.py:187-230) re-associated:
    score = xh_t . (W_k,h^T q_h) + b_k,h . q_h,      o_h = W_v,h (sum_t p_t xh_t) + (sum_t p_t) b_v,h.
Outputs and every gradient agree with the direct form (K and V projected per token) in fp32, with masks on both kinds of
key, for 2 x 2 and 4 x 4 windows."""
import pytest
import torch

from absorbed_ref import absorbed, direct


@pytest.mark.parametrize("T,nsmall", [(16, 3), (4, 1), (16, 0)])
def test_absorbed_equals_direct(T, nsmall):
    g = torch.Generator().manual_seed(T + nsmall)
    Bq, H, hd, Cin = 12, 16, 8, 40
    C = H * hd
    def rn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float64)
    qh = rn(Bq, C).requires_grad_()
    kv_small = [rn(Bq, 2 * C).requires_grad_() for _ in range(nsmall)]
    xhat = rn(Bq, T, Cin).requires_grad_()
    wk, wv = (0.3 * rn(C, Cin)).requires_grad_(), (0.3 * rn(C, Cin)).requires_grad_()
    bk, bv = rn(C).requires_grad_(), rn(C).requires_grad_()
    masks_small = [torch.rand(Bq, generator=g) > 0.2 for _ in range(nsmall)]
    if nsmall:
        masks_small[0] = None
    mask_a = torch.rand(Bq, T, generator=g) > 0.3
    mask_a[:, 0] = True                                    # a query always keeps one key (train_fsdp.py:1133-1137)
    leaves = [qh, xhat, wk, bk, wv, bv] + kv_small
    w = rn(Bq, C)
    outs, grads = [], []
    for f in (direct, absorbed):
        for t in leaves:
            t.grad = None
        o = f(qh, kv_small, masks_small, xhat, mask_a, wk, bk, wv, bv, heads=H)
        (o * w).sum().backward()
        outs.append(o.detach())
        grads.append([t.grad.clone() for t in leaves])
    assert torch.allclose(outs[0], outs[1].to(outs[0].dtype), atol=1e-9, rtol=1e-7)
    for a, b in zip(*grads):
        assert torch.allclose(a, b.to(a.dtype), atol=1e-8, rtol=1e-6)
    # masked keys of the windowed tower get exactly zero gradient in both forms
    assert torch.count_nonzero(grads[1][1][~mask_a]) == 0


def test_absorbed_tower_selection(monkeypatch):
    """VisionCrossAttentionLayer._absorbed_tower: which configuration takes the absorbed path — bf16 or fp32, 1024-wide features,
    exactly one windowed tower (up to 4 x 4) beside at most four one-key towers, switch on; ``config.fp8_projections`` composes with it
    (round 5: the fp8 GEMMs are the per-token ones that remain)."""
    import cambrian_amd.model.vision_sampler as VS
    from cambrian_amd import ops

    def layer(sizes):
        return VS.VisionCrossAttentionLayer(1024, 1024, [1024] * len(sizes), sizes, 1024, 0)

    q16, q32 = torch.zeros(4, 1024, dtype=torch.bfloat16), torch.zeros(4, 1024)
    feats = lambda sizes, d=1024: [torch.zeros(4 * s * s, d, dtype=torch.bfloat16) for s in sizes]
    rel = [1, 1, 1, 4]
    monkeypatch.setattr(VS, "ABSORB_KV", True)
    assert layer(rel)._absorbed_tower(q16, feats(rel)) == 3                       # the release tower set
    assert layer([4, 1])._absorbed_tower(q16, feats([4, 1])) == 0
    assert layer(rel)._absorbed_tower(q32, feats(rel)) == 3                       # fp32: the exact instantiation (round 4)
    assert layer(rel)._absorbed_tower(q32.double(), feats(rel)) == -1             # any other dtype: per-token K|V
    assert layer([1, 1, 2, 4])._absorbed_tower(q16, feats([1, 1, 2, 4])) == -1    # two windowed towers
    assert layer([1, 1, 1, 1])._absorbed_tower(q16, feats([1, 1, 1, 1])) == -1    # none
    assert layer([1, 8])._absorbed_tower(q16, feats([1, 8])) == -1                # window larger than 4 x 4
    assert layer([1] * 5 + [4])._absorbed_tower(q16, feats([1] * 5 + [4])) == -1  # more than four one-key towers
    monkeypatch.setattr(ops, "_FP8_LINEAR", True, raising=False)
    assert layer(rel)._absorbed_tower(q16, feats(rel)) == 3                       # config.fp8_projections no longer switches it off
    monkeypatch.setattr(ops, "_FP8_LINEAR", False, raising=False)
    monkeypatch.setattr(VS, "ABSORB_KV", False)
    assert layer(rel)._absorbed_tower(q16, feats(rel)) == -1                      # CAMBRIAN_AMD_ABSORB_KV=0


def test_bf16_probabilities_in_the_token_mix_cost_less_than_a_bf16_ulp_of_the_output():
    """The MFMA token mix takes the probabilities as bf16 operands (as every MFMA attention kernel does).  On release-sized
    rows (1024 channels, 16 tokens, unit-variance activations) the output moves by < 3e-3 of its range — below the bf16
    rounding of the output itself (2^-8), which is why tests/test_sva_absorbed_gpu.py can keep the per-token path's bounds."""
    g = torch.Generator().manual_seed(2)
    Bq, H, hd, Cin, T = 64, 16, 64, 1024, 16
    C = H * hd
    rn = lambda *s: torch.randn(*s, generator=g)
    qh, xhat = rn(Bq, C), rn(Bq, T, Cin)
    kv_small = [rn(Bq, 2 * C) for _ in range(3)]
    wk, wv, bk, bv = 0.03 * rn(C, Cin), 0.03 * rn(C, Cin), 0.5 * rn(C), 0.5 * rn(C)
    exact = absorbed(qh, kv_small, [None] * 3, xhat, None, wk, bk, wv, bv)
    rounded = absorbed(qh, kv_small, [None] * 3, xhat, None, wk, bk, wv, bv, mix_dtype=torch.bfloat16)
    err = ((exact - rounded).abs().max() / exact.abs().max()).item()
    assert 0 < err < 3e-3, err


def test_absorbed_form_reproduces_the_oracle_attention():
    """One hop from the reference: oracle.sva.multi_kv_cross_attention restates vision_sampler.py:177-234 and is pinned by
    tests/test_oracle_golden.py to fixtures the REAL reference produced.  The absorbed form — LayerNorm affines folded into
    the projections (ops.fold_kv), the 4 x 4-window tower's K / V projections applied on the query side — gives its output
    on the same parameters and inputs (fp32, masks on both kinds of key)."""
    from oracle import sva as O
    g = torch.Generator().manual_seed(11)
    hidden, sizes, Bq = 1024, [1, 1, 4], 6
    p = O.init_sampler_params(hidden, hidden, [hidden] * 3, sizes, hidden, 1, g)
    pre = "layers.0.cross_attn."
    for k in list(p):                                   # non-trivial LayerNorm affines
        if ".0.weight" in k or ".0.bias" in k:
            p[k] = p[k] + 0.3 * torch.randn(p[k].shape, generator=g)
    x = torch.randn(Bq, 1, hidden, generator=g)
    kvs = [torch.randn(Bq, s * s, hidden, generator=g) for s in sizes]
    masks = [torch.rand(Bq, 1, 1, s * s, generator=g) > 0.3 for s in sizes]
    masks[0][:] = True
    masks[2][..., 0] = True
    want = O.multi_kv_cross_attention(p, pre, x, kvs, masks)

    def norm(t):
        return (t - t.mean(-1, keepdim=True)) * torch.rsqrt(t.var(-1, unbiased=False, keepdim=True) + 1e-5)

    def folded(i, kind):
        w, gam, bet = p[f"{pre}{kind}_proj_{i}.1.weight"], p[f"{pre}{kind}_proj_{i}.0.weight"], p[f"{pre}{kind}_proj_{i}.0.bias"]
        return w * gam[None, :], w @ bet
    qh = (O.layer_norm(x, p[pre + "q_proj.0.weight"], p[pre + "q_proj.0.bias"]) @ p[pre + "q_proj.1.weight"].T).view(Bq, hidden)
    kv_small = []
    for i in (0, 1):
        (wk, bk), (wv, bv) = folded(i, "k"), folded(i, "v")
        xh = norm(kvs[i]).view(Bq, hidden)
        kv_small.append(torch.cat([xh @ wk.T + bk, xh @ wv.T + bv], -1))
    (wk, bk), (wv, bv) = folded(2, "k"), folded(2, "v")
    got = absorbed(qh, kv_small, [m.view(Bq) for m in masks[:2]], norm(kvs[2]), masks[2].view(Bq, 16), wk, bk, wv, bv)
    got = got @ p[pre + "o_proj.weight"].T
    assert torch.allclose(got, want.view(Bq, hidden), atol=2e-5, rtol=1e-4), (got - want.view(Bq, hidden)).abs().max()
