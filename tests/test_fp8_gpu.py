"""GPU: fp8 projection GEMMs (BASELINE configs[4]) — the row-wise e4m3fn quantiser bit-exact against the oracle
(oracle/fp8.py, PyTorch's OCP float8_e4m3fn cast), the v_mfma_f32_32x32x64_f8f6f4 GEMM against an fp32 matmul of the
de-quantised operands, and ops.linear / the whole model with fp8 forward GEMMs against the un-quantised fp32 oracle
within the stated fp8 tolerances."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,K", [(1, 128), (7, 1152), (300, 1024), (33, 5760), (5, 16384), (64, 16)])
def test_quantiser_bit_exact(dev, dt, rows, K):
    from cambrian_amd import ops
    from oracle import fp8
    g = torch.Generator().manual_seed(rows * 131 + K)
    x = (torch.randn(rows, K, generator=g) * torch.logspace(-3, 2, rows)[:, None]).to(dt)
    if rows > 2:
        x[2] = 0                                   # an all-zero row: scale 1, bytes 0
    q, inv = ops.k_quantize_fp8_rows(x.to(dev))
    q_ref, inv_ref = fp8.quantize_rows(x)
    assert torch.equal(inv.cpu(), inv_ref)
    assert torch.equal(q.cpu(), q_ref)
    # the largest element of every non-zero row maps to +-448 exactly (0x7E / 0xFE)
    top = q.cpu().view(torch.float8_e4m3fn).float().abs().amax(1)
    assert torch.equal(top[x.float().abs().amax(1) > 0], torch.full_like(top[x.float().abs().amax(1) > 0], 448.0))


def test_quantiser_strided_rows_and_errors(dev):
    from cambrian_amd import lib as L, ops
    from oracle import fp8
    base = torch.randn(40, 2048 + 256, device=dev, dtype=torch.bfloat16)
    view = base[:, 256:]                            # row stride 2304, 16-byte aligned start
    q, inv = ops.k_quantize_fp8_rows(view)
    q_ref, inv_ref = fp8.quantize_rows(view.cpu())
    assert torch.equal(q.cpu(), q_ref) and torch.equal(inv.cpu(), inv_ref)
    with pytest.raises(L.CambrianAmdError):
        ops.k_quantize_fp8_rows(torch.randn(4, 40, device=dev, dtype=torch.bfloat16))      # K % 16
    with pytest.raises(L.CambrianAmdError):
        ops.k_quantize_fp8_rows(torch.randn(4, 32768, device=dev, dtype=torch.bfloat16))   # K too large


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1000, 1024, 1152), (77, 264, 256), (4096, 1024, 5760)])
@pytest.mark.parametrize("out_dt", [torch.float32, torch.bfloat16])
def test_fp8_gemm_matches_dequantised_matmul(dev, M, N, K, out_dt):
    from cambrian_amd import ops
    from oracle import fp8
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * (1 + torch.arange(M)[:, None] % 7)     # rows of different magnitude
    w = torch.randn(N, K, generator=g) * (0.2 + (torch.arange(N)[:, None] % 5) * 0.3)
    xq, xi = ops.k_quantize_fp8_rows(x.to(dev))
    wq, wi = ops.k_quantize_fp8_rows(w.to(dev))
    y = ops.k_gemm_fp8(xq, xi, wq, wi, out_dtype=out_dt)
    ref = fp8.dequantize(xq.cpu(), xi.cpu()).double() @ fp8.dequantize(wq.cpu(), wi.cpu()).double().T
    # fp32 out: the products are exact, what differs is the f8f6f4 MFMA's internal summation (observed 2e-5 of the
    # largest output at K = 128); bf16 out: one more rounding
    tol = 1e-4 if out_dt == torch.float32 else 4e-3
    assert rel_err(y, ref.float()) < tol
    # against the un-quantised product: the fp8 tolerance of the mode
    assert rel_err(y, (x.double() @ w.double().T).float()) < 6e-2


def test_fp8_gemm_fused_epilogue(dev):
    from cambrian_amd import lib as L, ops
    from oracle import fp8, sva as O
    M, N, K = 600, 1024, 1536
    g = torch.Generator().manual_seed(5)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g).bfloat16()
    xq, xi = ops.k_quantize_fp8_rows(x.to(dev))
    wq, wi = ops.k_quantize_fp8_rows(w.to(dev))
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    y = ops.k_gemm_fp8(xq, xi, wq, wi, bias=bias.to(dev), act=L.ACT_GELU_ERF, residual=res.to(dev), pre_out=pre)
    lin = fp8.linear(x, w, bias)
    assert rel_err(pre, lin) < 4e-3
    assert rel_err(y, O.gelu_erf(lin) + res.float()) < 6e-3


def test_linear_fp8_forward_bf16_backward(dev):
    from cambrian_amd import ops
    M, N, K = 1152, 1024, 1152
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16).requires_grad_()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_()          # fp32 master
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    dy = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
    w = torch.nn.Parameter(w.detach())
    y_ref = ops.linear(x, w, b, heavy=True)         # context off: bf16
    y_ref.backward(dy)
    ref = (y_ref.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    with ops.fp8_projections(True):
        y = ops.linear(x, w, b, heavy=True)
        y_light = ops.linear(x, w, b)               # not marked heavy: stays bf16 inside the context
    assert not ops._FP8_LINEAR and torch.equal(y_light, y_ref)
    y.backward(dy)
    exact = x.detach().float() @ w.detach().T + b.detach()
    assert rel_err(y, exact) < 6e-2 and rel_err(y, exact) > 1e-3                       # it really ran in fp8
    assert rel_err(ref[0], exact) < 1e-2
    # no activation: the backward never sees the quantised forward -> identical to the bf16 path
    assert torch.equal(x.grad, ref[1]) and torch.equal(w.grad, ref[2])
    assert torch.allclose(b.grad, ref[3], rtol=1e-5, atol=1e-4)       # column sums accumulate with fp32 atomics
    # the quantised weight is cached and refreshed when the parameter changes
    n0 = len(ops._FP8_WEIGHT_CACHE)
    with ops.fp8_projections(True), torch.no_grad():
        y2 = ops.linear(x, w, b, heavy=True)
        assert len(ops._FP8_WEIGHT_CACHE) == n0 and torch.equal(y2, y)
        w.mul_(2.0)
        y3 = ops.linear(x, w, b, heavy=True)
        y4 = ops.linear(x, w.detach()[:, :], b, heavy=True)     # a view: quantised per call, never cached
        assert len(ops._FP8_WEIGHT_CACHE) == n0 and torch.equal(y4, y3)
    assert rel_err(y3, 2 * (exact - b.detach()) + b.detach()) < 6e-2


def test_fp8_projections_compose_with_the_absorbed_path(dev):
    """``fp8_projections`` must not select the slower per-token K|V algorithm.  A 1024-wide SVA layer with one
    one-key tower and one 4 x 4-window tower: under the fp8 context the windowed tower still takes the absorbed path (no K|V
    projection exists for it to quantise), the one-key tower's K|V GEMM runs in fp8; forward and feature gradients stay
    within fp8 tolerance of the bf16 run."""
    from cambrian_amd import ops
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    from oracle import sva as O
    gen = torch.Generator().manual_seed(0)
    kv_sizes, B, qside, hidden = [1, 4], 1, 4, 1024
    p = O.init_sampler_params(hidden, hidden, [hidden] * 2, kv_sizes, hidden, 1, gen)
    m = VisionTokenSampler(hidden, hidden, [hidden] * 2, kv_sizes, hidden, 1)
    m.load_state_dict(p, strict=True)
    m = m.to(dev)
    Bq = B * qside * qside
    q = torch.randn(Bq, hidden, generator=gen).to(dev, torch.bfloat16)
    ctx = torch.randn(B, hidden, generator=gen).to(dev, torch.bfloat16)
    feats = [torch.randn(B * (qside * s) ** 2, hidden, generator=gen) for s in kv_sizes]
    masks = [torch.ones(Bq, s * s, dtype=torch.uint8, device=dev) for s in kv_sizes]
    taken = []
    layer = m.layers[0]
    orig = layer._absorbed_tower
    layer._absorbed_tower = lambda qh, fs: taken.append(orig(qh, fs)) or taken[-1]
    res = {}
    for mode in (False, True):
        holders = [ops.GradAccumulator() for _ in kv_sizes]
        fd = [f.to(dev, torch.bfloat16).requires_grad_() for f in feats]
        shared = [ops.shared_grad(f, h, m.pos_tables(i)) for i, (f, h) in enumerate(zip(fd, holders))]
        with ops.fp8_projections(mode):
            out = m.forward_fused(q, ctx, shared, masks, holders, B, qside)
        out.float().sum().backward()
        res[mode] = (out.detach().float(), [f.grad.float() for f in fd])
    assert taken == [1, 1], taken                      # the windowed tower is absorbed in BOTH modes
    assert rel_err(res[True][0], res[False][0]) < 6e-2
    assert not torch.equal(res[True][0], res[False][0])    # the fp8 GEMM really ran
    for g8, g16 in zip(res[True][1], res[False][1]):
        assert rel_err(g8, g16) < 1e-1
