"""`-m gpu`: cmb_gemm_tn (csrc/gemm_tn.hip) — C = At^T Bt with both operands row-major over the contraction rows — against
fp32 torch on the same bf16 operands: ragged row counts (zero rows from the kernel's own zero source), ragged tiles,
strided operands, split-K, the batched per-head form, alpha / beta, and LinearFn's weight gradient through it.  Random
operands: any transposition or k-permutation mismatch between the two operands' LDS reads shows as O(1) error."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    return ops, L


@pytest.mark.parametrize("K,M,N,split", [(64, 128, 128, 1), (200, 128, 256, 1), (1000, 64, 1024, 1), (9216, 1024, 1024, 6),
                                         (577, 1152, 200, 3), (4096, 8, 24, 1), (130, 264, 136, 2), (3, 128, 128, 1)])
def test_gemm_tn_matches_torch(dev, K, M, N, split):
    ops, L = _ops()
    g = torch.Generator().manual_seed(K + M + N)
    at = torch.randn(K, M, generator=g).to(torch.bfloat16).to(dev)
    bt = torch.randn(K, N, generator=g).to(torch.bfloat16).to(dev)
    out = ops.k_gemm_tn(at, bt, split_k=split)
    ref = at.float().t() @ bt.float()
    assert out.dtype == torch.float32 and out.shape == (M, N)
    assert rel_err(out, ref) < 2e-5 * max(1, K) ** 0.5, rel_err(out, ref)
    assert L.load().cmb_gemm_last_kernel() == 1281


def test_gemm_tn_strided_operands_alpha_beta_and_bf16_result(dev):
    ops, L = _ops()
    g = torch.Generator().manual_seed(9)
    K, M, N = 777, 192, 320
    big_a = torch.randn(K, M + 64, generator=g).to(torch.bfloat16).to(dev)
    big_b = torch.randn(K, N + 40, generator=g).to(torch.bfloat16).to(dev)
    at, bt = big_a[:, 32:32 + M], big_b[:, 8:8 + N]                 # column windows of wider tensors (row strides M+64, N+40)
    c0 = torch.randn(M, N, generator=g).to(dev)
    out = c0.clone()
    ops.k_gemm_tn(at, bt, out=out, alpha=0.5, beta=2.0)
    ref = 0.5 * (at.float().t() @ bt.float()) + 2.0 * c0
    assert rel_err(out, ref) < 1e-4
    ob = ops.k_gemm_tn(at, bt, out_dtype=torch.bfloat16)
    assert ob.dtype == torch.bfloat16 and rel_err(ob, at.float().t() @ bt.float()) < 6e-3


def test_gemm_tn_batched_heads(dev):
    """The per-head weight gradients of the absorbed projections: problem h reads columns [64 h, 64 h + 64) of x and
    [1024 h, 1024 h + 1024) of dU."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(4)
    Bq, H, hd, Cin = 1000, 16, 64, 1024
    x = torch.randn(Bq, H * hd, generator=g).to(torch.bfloat16).to(dev)
    du = torch.randn(Bq, H * Cin, generator=g).to(torch.bfloat16).to(dev)
    dw = torch.empty(H * hd, Cin, dtype=torch.float32, device=dev)
    ref = torch.einsum("qhj,qhc->hjc", x.float().view(Bq, H, hd), du.float().view(Bq, H, Cin)).reshape(H * hd, Cin)
    for split in (1, 5):                                            # 5: [split][16 * 64][1024] slabs reduced as one matrix
        dw.fill_(float("nan"))
        ops.k_gemm_tn(x, du, out=dw, M=hd, N=Cin, batch=H, a_bs=hd, b_bs=Cin, c_bs=hd * Cin, ldc=Cin, split_k=split)
        assert rel_err(dw, ref) < 1e-4, split


def test_gemm_tn_rejects_what_it_does_not_do(dev):
    ops, L = _ops()
    at = torch.zeros(64, 128, dtype=torch.bfloat16, device=dev)
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm_tn(at[:, :100], at)                              # M not a multiple of 8
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm_tn(at.float(), at.float())                       # bf16 only


@pytest.mark.parametrize("M,K,N", [(1000, 1024, 1152), (9216, 1152, 1024), (333, 256, 64)])
def test_linear_weight_gradient_is_the_same_through_tn_and_transposes(dev, monkeypatch, M, K, N):
    """LinearFn.backward: dW from cmb_gemm_tn equals dW from transposed copies + the NT kernel (the round-2 path, still used
    for fp32) to fp32 summation-order noise."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    gy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    grads = []
    monkeypatch.setattr(ops, "_tn_wgrad_wins", lambda *a: True)    # (the size policy is not what is under test)
    for on in (True, False):
        monkeypatch.setattr(ops, "TN_WGRAD", on)
        w = (torch.randn(N, K, generator=torch.Generator().manual_seed(1)) * 0.02).to(dev).requires_grad_()
        b = torch.zeros(N, device=dev, requires_grad=True)
        y = ops.linear(x, w, b, act=L.ACT_GELU_ERF)
        y.backward(gy)
        grads.append(w.grad.clone())
    assert rel_err(grads[0], grads[1]) < 1e-5, rel_err(grads[0], grads[1])
