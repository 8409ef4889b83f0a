"""LayerNorm folded into the linear that follows it (round 6: cmb_row_stats + cmb_gemm_desc.row_mean) against the separate
LayerNorm + linear it replaces, on every GEMM kernel and activation the frozen towers use."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods():
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    from cambrian_amd.model.multimodal_encoder import vit_ops
    return ops, L, vit_ops


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


@pytest.mark.parametrize("rows,D", [(1000, 384), (577, 1024), (730, 1536), (96, 3072), (64, 1152), (3, 8)])
def test_row_stats_equal_the_layernorm_kernels(dev, rows, D):
    ops, L, _ = _mods()
    g = torch.Generator().manual_seed(rows + D)
    for dt in (torch.bfloat16, torch.float32):
        x = (torch.randn(rows, D, generator=g) * 3 + 1.5).to(dt).to(dev)
        mean, rstd = ops.k_row_stats(x, 1e-6)
        _, m2, r2 = ops.k_layernorm_fwd(x, torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-6)
        assert torch.equal(mean, m2) and torch.equal(rstd, r2)          # the same arithmetic, operation for operation
        xf = x.float()
        assert torch.allclose(mean, xf.mean(1), atol=1e-5, rtol=1e-5)
        assert torch.allclose(rstd, (xf.var(1, unbiased=False) + 1e-6).rsqrt(), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("tile", [2590, 2560, 128, 0])
@pytest.mark.parametrize("act", ["none", "gelu_erf", "gelu_tanh", "quick_gelu", "swiglu_pairs"])
def test_folded_layernorm_linear(dev, tile, act):
    """y = act(LN(x) W^T + b): fp32 math on the bf16-rounded operands (1e-2: the unfolded bf16 pipeline rounds the normalised
    rows to bf16, the folded one does not — it is the MORE accurate of the two), and against the round-5 pipeline (LayerNorm
    kernel + GEMM) at bf16 rounding; rows with a large common offset (what cancels against mean * colsum)."""
    ops, L, vit_ops = _mods()
    code = {"none": L.ACT_NONE, "gelu_erf": L.ACT_GELU_ERF, "gelu_tanh": L.ACT_GELU_TANH, "quick_gelu": L.ACT_QUICK_GELU,
            "swiglu_pairs": L.ACT_SWIGLU_PAIRS}[act]
    M, N, K = 20000 if tile in (2590, 0) else 1100, 768, 512
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(M, K, generator=g) * 2.0 + torch.randn(M, 1, generator=g) * 6.0).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) * 0.3
    gamma, beta = (1 + 0.2 * torch.randn(K, generator=g)).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
    w2, cs, b2 = vit_ops.fold_ln_into_linear(w, b, gamma, beta, torch.bfloat16)
    st = ops.k_row_stats(x, 1e-6)
    y = ops.k_gemm(x, w2, bias=b2, act=code, row_stats=st, col_sum=cs, tile=tile)

    def fin(z):
        if act == "swiglu_pairs":
            return F.silu(z[:, 0::2]) * z[:, 1::2]
        if act == "gelu_erf":
            return F.gelu(z)
        if act == "gelu_tanh":
            return F.gelu(z, approximate="tanh")
        if act == "quick_gelu":
            return z * torch.sigmoid(1.702 * z)
        return z
    want = fin(F.layer_norm(x.float(), (K,), gamma, beta, 1e-6) @ w.T + b)
    assert y.shape == want.shape
    assert rel(y, want) < 1e-2
    xn, _, _ = ops.k_layernorm_fwd(x, gamma, beta, 1e-6, want_stats=False)
    old = ops.k_gemm(xn, w.to(torch.bfloat16), bias=b, act=code, tile=tile)
    assert rel(y, old) < 2e-2
    assert rel(old, want) >= 0.5 * rel(y, want) or rel(y, want) < 4e-3    # folding does not cost accuracy


def test_folded_layernorm_rows_take_the_same_bits_on_every_kernel(dev):
    """a problem's rows may be split between the 256- and the 128-tile kernels (tail split): same roundings everywhere"""
    ops, L, vit_ops = _mods()
    g = torch.Generator().manual_seed(23)
    M, N, K = 1536, 512, 256
    x = (torch.randn(M, K, generator=g) * 2 + 3).to(torch.bfloat16).to(dev)
    w2, cs, b2 = vit_ops.fold_ln_into_linear(torch.randn(N, K, generator=g).to(dev) / 16, torch.randn(N, generator=g).to(dev),
                                             1 + 0.1 * torch.randn(K, generator=g).to(dev), 0.1 * torch.randn(K, generator=g).to(dev),
                                             torch.bfloat16)
    st = ops.k_row_stats(x, 1e-5)
    outs = [ops.k_gemm(x, w2, bias=b2, act=L.ACT_GELU_ERF, row_stats=st, col_sum=cs, tile=t) for t in (2590, 2560, 128)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm(x, w2, act=L.ACT_NONE, row_stats=st, col_sum=cs)      # no bias: rejected
