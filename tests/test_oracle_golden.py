"""CPU: the oracle (oracle/*.py) replayed against golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py ran /root/reference on seeded inputs in the build container)."""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.fail(f"golden fixture {name} is missing; run tests/golden/make_golden.py in the build container")
    return torch.load(path, weights_only=False)


def test_sva_oracle_matches_reference_forward_and_backward():
    from oracle import sva
    fx = _load("sva_small.pt")
    p = {k: v.clone().requires_grad_() for k, v in fx["state"].items()}
    q, ctx = fx["q"].clone().requires_grad_(), fx["ctx"].clone().requires_grad_()
    kvs = [k.clone().requires_grad_() for k in fx["kvs"]]
    out = sva.vision_token_sampler(p, q, ctx, kvs, fx["masks"])
    assert torch.allclose(out, fx["out"], atol=2e-6, rtol=1e-5)
    (out * fx["w"]).sum().backward()
    assert torch.allclose(q.grad, fx["dq"], atol=5e-6, rtol=1e-4)
    assert torch.allclose(ctx.grad, fx["dctx"], atol=5e-6, rtol=1e-4)
    for a, b in zip(kvs, fx["dkvs"]):
        assert torch.allclose(a.grad, b, atol=5e-6, rtol=1e-4)
    for name, g in fx["dparams"].items():
        assert torch.allclose(p[name].grad, g, atol=2e-5, rtol=1e-4), name


def test_sva_oracle_mask_size_check():
    from oracle import sva
    fx = _load("sva_small.pt")
    bad = [m.clone() for m in fx["masks"]]
    bad[2] = bad[2][:, :3]
    with pytest.raises(ValueError, match="Attention mask should be of size"):
        sva.vision_token_sampler(fx["state"], fx["q"], fx["ctx"], fx["kvs"], bad)
