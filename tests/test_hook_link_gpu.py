"""GPU: the gather / scatter pair of the in-LLM SVA hook (ops.GatherQueryRowsFn / ScatterQueryRowsFn, the reference's
``hidden_states[:, a:b]`` read and in-place write-back, cambrian_llama.py:181-207) with the shared-gradient ``link`` against
plain torch indexing differentiated by autograd — for a ``hidden`` that is a fresh tensor, a VIEW of its producer's output
(autograd wraps the in-place scatter in CopySlices) and one with a THIRD consumer whose gradient is accumulated
(the link must not depend on which buffer autograd keeps)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S, H, SIDE, POS = 2, 40, 64, 3, 5


def _reference(leaf, w_rows, w_out, third):
    x = leaf * 1.5
    hidden = x.view(B, S, H)
    extra = (hidden * 0.25).sum() if third else 0.0
    n = SIDE * (SIDE + 1)
    blk = hidden[:, POS:POS + n].reshape(B, SIDE, SIDE + 1, H)
    q = blk[:, :, :SIDE].reshape(B * SIDE * SIDE, H)
    out = torch.tanh(q) * w_rows
    new_blk = torch.cat([out.view(B, SIDE, SIDE, H), blk[:, :, SIDE:]], dim=2).reshape(B, n, H)
    hidden2 = torch.cat([hidden[:, :POS], new_blk, hidden[:, POS + n:]], dim=1)
    return (hidden2 * w_out).sum() + extra


def _ours(ops, leaf, w_rows, w_out, as_view, third, use_link):
    x = leaf * 1.5
    hidden = x.view(B, S, H) if as_view else x.view(B, S, H).clone()
    extra = (hidden * 0.25).sum() if third else 0.0   # reads `hidden` BEFORE the in-place scatter
    link = {} if use_link else None
    q = ops.gather_query_rows(hidden, POS, SIDE, link)
    out = torch.tanh(q) * w_rows
    hidden2 = ops.scatter_query_rows(hidden, out, POS, SIDE, link)
    return (hidden2 * w_out).sum() + extra


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("as_view", [False, True])
@pytest.mark.parametrize("third", [False, True])
def test_hook_link_matches_plain_indexing(dev, dt, as_view, third):
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(11)
    leaf0 = torch.randn(B * S * H, generator=g)
    w_rows = torch.randn(B * SIDE * SIDE, H, generator=g).to(dev, dt)
    w_out = torch.randn(B, S, H, generator=g).to(dev, dt)
    grads = {}
    for name in ("ref", "link", "nolink"):
        leaf = leaf0.to(dev, dt).requires_grad_(True)
        if name == "ref":
            loss = _reference(leaf, w_rows, w_out, third)
        else:
            loss = _ours(ops, leaf, w_rows, w_out, as_view, third, use_link=(name == "link"))
        loss.backward()
        grads[name] = (loss.detach().float().cpu(), leaf.grad.float().cpu())
    tol = 1e-6 if dt == torch.float32 else 2e-2
    for name in ("link", "nolink"):
        assert torch.allclose(grads[name][0], grads["ref"][0], rtol=tol, atol=tol * 10), name
        d = (grads[name][1] - grads["ref"][1]).abs().max() / grads["ref"][1].abs().max()
        assert d < tol, (name, float(d))
    assert torch.equal(grads["link"][1], grads["nolink"][1]) or dt == torch.bfloat16 and third


def test_hook_link_hidden_without_grad(dev):
    """`hidden` needs no gradient (frozen producer): the scatter still returns d(rows); nothing is parked in the link."""
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn(B, S, H, generator=g).to(dev)
    w = torch.randn(B * SIDE * SIDE, H, generator=g).to(dev).requires_grad_(True)
    link = {}
    q = ops.gather_query_rows(hidden, POS, SIDE, link)
    out = torch.tanh(q) * w
    hidden2 = ops.scatter_query_rows(hidden, out, POS, SIDE, link)
    hidden2.sum().backward()
    assert "dh" not in link
    assert torch.allclose(w.grad, torch.tanh(q), atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("two_consumers", [False, True])
def test_scatter_backward_writes_into_an_exclusive_gradient_buffer(dev, dt, two_consumers):
    """Round 6: the gradient that RmsNormForkFn.backward allocates for the decoder layer's input is marked as handed to one
    receiver, and ScatterQueryRowsFn.backward zeroes the overwritten rows IN it instead of in a 400 MB copy.  The gradients must
    be those of plain indexing + a plain RMSNorm whether the mark arrives (one consumer) or autograd has summed two gradients
    into a new, unmarked tensor (two consumers)."""
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(5)
    leaf0 = torch.randn(B * S * H, generator=g)
    w_rows = torch.randn(B * SIDE * SIDE, H, generator=g).to(dev, dt)
    w_norm = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dev, dt)
    w_out = torch.randn(B, S, H, generator=g).to(dev, dt)
    n = SIDE * (SIDE + 1)

    def tail(hidden2, fused):
        if fused:
            skip, y = ops.rmsnorm_fork(hidden2, w_norm, 1e-6)
        else:
            skip, y = hidden2, hidden2 * torch.rsqrt(hidden2.float().pow(2).mean(-1, keepdim=True) + 1e-6).to(dt) * w_norm
        loss = (y * w_out).sum() + (skip * 0.5).sum()
        if two_consumers:
            loss = loss + (hidden2 * 0.125).sum()
        return loss

    grads = {}
    for name in ("ref", "ours"):
        leaf = leaf0.to(dev, dt).requires_grad_(True)
        hidden = (leaf * 1.5).view(B, S, H).clone()
        if name == "ref":
            blk = hidden[:, POS:POS + n].reshape(B, SIDE, SIDE + 1, H)
            out = torch.tanh(blk[:, :, :SIDE].reshape(B * SIDE * SIDE, H)) * w_rows
            new_blk = torch.cat([out.view(B, SIDE, SIDE, H), blk[:, :, SIDE:]], dim=2).reshape(B, n, H)
            loss = tail(torch.cat([hidden[:, :POS], new_blk, hidden[:, POS + n:]], dim=1), fused=False)
        else:
            link = {}
            out = torch.tanh(ops.gather_query_rows(hidden, POS, SIDE, link)) * w_rows
            loss = tail(ops.scatter_query_rows(hidden, out, POS, SIDE, link), fused=True)
        loss.backward()
        grads[name] = leaf.grad.float().cpu()
    tol = 2e-5 if dt == torch.float32 else 3e-2
    d = (grads["ours"] - grads["ref"]).abs().max() / grads["ref"].abs().max()
    assert d < tol, float(d)
