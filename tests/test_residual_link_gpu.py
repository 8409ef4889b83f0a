"""The two gradients of an SVA block's input (proj_in branch + the block's residual) meet inside the proj_in backward GEMM's
epilogue instead of an ATen add (ops.LinearFn link / role): same gradients as autograd's own sum."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3)])
def test_block_input_gradient_with_and_without_the_link(dev, dt, tol, monkeypatch):
    from cambrian_amd import ops
    import cambrian_amd.model.vision_sampler as VS
    torch.manual_seed(1)
    m = VS.VisionTokenSampler(1024, 1024, [1024] * 4, [1, 1, 1, 4], 1024, 2).to(dev)
    B, qside = 2, 4
    g = torch.Generator().manual_seed(4)
    q0 = torch.randn(B * qside * qside, 1024, generator=g).to(dev, dt)
    ctx = torch.randn(B, 1024, generator=g).to(dev, dt)
    feats = [torch.randn(B * (qside * s) ** 2, 1024, generator=g).to(dev, dt) for s in (1, 1, 1, 4)]
    masks = [None] * 4
    res = []
    adds = []
    real_gemm = ops.k_gemm
    for on in (True, False):
        monkeypatch.setattr(ops, "LINK_RESIDUAL_GRADS", on)
        with_res = []
        monkeypatch.setattr(ops, "k_gemm", lambda *a, **k: (with_res.append(k.get("residual") is not None), real_gemm(*a, **k))[1])
        m.zero_grad(set_to_none=True)
        q = q0.clone().requires_grad_()
        holders = [ops.GradAccumulator() for _ in range(4)]
        fd = [f.clone().requires_grad_() for f in feats]
        shared = [ops.shared_grad(f, h, m.pos_tables(i)) for i, (f, h) in enumerate(zip(fd, holders))]
        out = m.forward_fused(q, ctx, shared, masks, holders, B, qside)
        n_fwd = len(with_res)
        out.float().pow(2).mean().backward()
        res.append((out.detach().clone(), q.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
        adds.append(sum(with_res[n_fwd:]))          # backward GEMMs that carried a residual operand
        monkeypatch.setattr(ops, "k_gemm", real_gemm)
    assert torch.equal(res[0][0], res[1][0])                          # the forward is untouched
    assert rel_err(res[0][1].float(), res[1][1].float()) < tol        # d(q): one rounding instead of two in bf16
    for n in res[0][2]:
        assert rel_err(res[0][2][n].float(), res[1][2][n].float()) < max(tol, 1e-5), n
    assert adds == [2, 0]                                             # one linked d(x) GEMM per layer, none without the link
