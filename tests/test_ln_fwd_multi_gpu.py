"""cmb_layernorm_fwd_multi (round 6): the SVA layers' normalisations of one tower's tokens in ONE pass over x — against the
per-layer kernel bit for bit, and through the sampler (multi-layer pass on / off: same outputs, same gradients)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows_per_img,D,side,r,layers", [(64, 1024, 8, 4, 13), (36, 1024, 6, 2, 3), (16, 512, 4, 1, 5),
                                                          (144, 384, 12, 4, 17), (64, 1024, 8, 4, 1)])
def test_fwd_multi_equals_per_layer_kernel(dev, dt, rows_per_img, D, side, r, layers):
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(rows_per_img * layers + D)
    B = 3
    x = torch.randn(B * rows_per_img, D, generator=g).to(dev, dt)
    adds = [None if (r == 1 or l == 2) else torch.randn(r * r, D, generator=g).to(dev) for l in range(layers)]   # one table-less layer
    got = ops.k_layernorm_fwd_multi(x, adds, 1e-5, side, r)
    assert len(got) == layers
    for a, (y, mean, rstd) in zip(adds, got):
        wy, wm, wr = ops.k_layernorm_fwd(x, None, None, 1e-5, add=a, side=side, grid_r=r)
        assert torch.equal(y, wy) and torch.equal(mean, wm) and torch.equal(rstd, wr)
    # and against torch on one layer with a table
    k = next((i for i, a in enumerate(adds) if a is not None), None)
    if k is not None:
        rows = torch.arange(B * rows_per_img, device=dev) % (side * side)
        wp = (rows // side % r) * r + (rows % side) % r
        ref = torch.nn.functional.layer_norm(x.float() + adds[k][wp], (D,), None, None, 1e-5)
        assert rel_err(got[k][0].float(), ref) < (1e-2 if dt == torch.bfloat16 else 1e-5)


def test_fwd_multi_rejects_bad_arguments(dev):
    from cambrian_amd import lib as L, ops
    x = torch.randn(32, 2048, device=dev).bfloat16()
    with pytest.raises(L.CambrianAmdError):
        ops.k_layernorm_fwd_multi(x, [None], 1e-5)                       # D > 1024
    x = torch.randn(32, 1024, device=dev).bfloat16()
    with pytest.raises(L.CambrianAmdError):
        ops.k_layernorm_fwd_multi(x, [torch.zeros(4, 512, device=dev)], 1e-5, 4, 2)   # table of another width


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_sampler_with_and_without_the_multi_layer_pass(dev, dt, monkeypatch):
    """A 3-layer sampler over the release tower set with its position tables announced (as cambrian_arch.py does): the
    multi-layer forward pass + shared table-less normalisation give the per-layer launches' output bit for bit, gradients to
    the summation order of the deferred backward's atomics; the per-layer kernel is NOT launched for the windowed tower."""
    from cambrian_amd import ops
    import cambrian_amd.model.vision_sampler as VS
    torch.manual_seed(0)
    m = VS.VisionTokenSampler(1024, 1024, [1024] * 4, [1, 1, 1, 4], 1024, 3).to(dev)
    if dt == torch.bfloat16:
        pass   # fp32 masters, bf16 activations: the bench's configuration
    B, qside = 2, 6
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B * qside * qside, 1024, generator=g).to(dev, dt)
    ctx = torch.randn(B, 1024, generator=g).to(dev, dt)
    feats = [torch.randn(B * (qside * s) ** 2, 1024, generator=g).to(dev, dt) for s in (1, 1, 1, 4)]
    masks = [None, None, None, None]
    outs, grads, launches = [], [], []
    real = ops.k_layernorm_fwd
    for on in (True, False):
        monkeypatch.setattr(ops, "LN_FWD_MULTI", on)
        count = []
        monkeypatch.setattr(ops, "k_layernorm_fwd", lambda *a, **k: (count.append(k.get("add") is not None), real(*a, **k))[1])
        m.zero_grad(set_to_none=True)
        holders = [ops.GradAccumulator() for _ in range(4)]
        fd = [f.clone().requires_grad_() for f in feats]
        shared = [ops.shared_grad(f, h, m.pos_tables(i)) for i, (f, h) in enumerate(zip(fd, holders))]
        out = m.forward_fused(q, ctx, shared, masks, holders, B, qside)
        out.float().pow(2).mean().backward()
        outs.append(out.detach().clone())
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()} | {f"feat{i}": f.grad.clone() for i, f in enumerate(fd)})
        launches.append((sum(count), len(count)))
        monkeypatch.setattr(ops, "k_layernorm_fwd", real)
    assert torch.equal(outs[0], outs[1])
    for n in grads[0]:
        assert rel_err(grads[0][n].float(), grads[1][n].float()) < 1e-5, n
    with_table_on, total_on = launches[0]
    with_table_off, total_off = launches[1]
    assert with_table_on == 0 and with_table_off == 3          # the windowed tower's three normalisations left as one multi pass
    assert total_off - total_on >= 3 + 2 * 3                   # ... and each one-key tower is normalised once, not three times
