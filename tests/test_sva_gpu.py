"""`-m gpu`: the HIP VisionTokenSampler (drop-in for cambrian/model/vision_sampler.py) against the CPU oracle
(oracle/sva.py, itself pinned to the reference by tests/golden/sva_small.pt) — forward, and every gradient."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

# stated tolerances (max-abs error / max-abs reference).  fp32: exact-fp32 MFMA + fp32 VALU kernels.
# bf16: bf16 storage / fp32 accumulate — what the reference's own TPU run computes in (fsdp_config.json:6).
FWD_TOL = {"fp32": 1e-4, "bf16": 3e-2}
BWD_TOL = {"fp32": 5e-4, "bf16": 6e-2}


def _build(dev, dt, q_dim, layers, kv_sizes, seed):
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    from oracle import sva as O
    gen = torch.Generator().manual_seed(seed)
    hidden = 1024
    p = O.init_sampler_params(q_dim, hidden, [hidden] * len(kv_sizes), kv_sizes, hidden, layers, gen)
    m = VisionTokenSampler(q_dim, hidden, [hidden] * len(kv_sizes), kv_sizes, hidden, layers)
    missing, unexpected = m.load_state_dict(p, strict=True), None
    m = m.to(dev)  # fp32 master parameters, as in the reference
    return m, p, gen


def _inputs(gen, B, qside, q_dim, kv_sizes, hidden=1024):
    Bq = B * qside * qside
    q = torch.randn(Bq, 1, q_dim, generator=gen)
    ctx_b = torch.randn(B, hidden, generator=gen)
    feats = [torch.randn(B, (qside * s) ** 2, hidden, generator=gen) for s in kv_sizes]  # tower-token-major
    masks = [torch.rand(Bq, s * s, generator=gen) > 0.25 for s in kv_sizes]
    for m in masks:
        m[m.sum(1) == 0] = True
    return q, ctx_b, feats, masks


def _window_major(f, B, qside, s):
    C = f.shape[-1]
    return f.view(B, qside, s, qside, s, C).permute(0, 1, 3, 2, 4, 5).contiguous().flatten(0, 2).flatten(1, 2)


def _oracle(p, q, ctx_b, feats, masks, B, qside, kv_sizes):
    from oracle import sva as O
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    qr, cr = q.clone().requires_grad_(), ctx_b.clone().requires_grad_()
    fr = [f.clone().requires_grad_() for f in feats]
    ctx = cr[:, None, None, :].expand(-1, qside * qside, 1, -1).flatten(0, 1)  # cambrian_arch.py:384
    kv = [_window_major(f, B, qside, s) for f, s in zip(fr, kv_sizes)]          # cambrian_arch.py:271-287
    out = O.vision_token_sampler(pr, qr, ctx, kv, masks)
    return out, pr, qr, cr, fr


@pytest.mark.parametrize("name,dt", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("q_dim,layers,fused", [(1024, 2, True), (1024, 1, False), (4096, 1, True)])
def test_sampler_matches_oracle(dev, name, dt, q_dim, layers, fused):
    from cambrian_amd import ops
    kv_sizes, B, qside = [1, 1, 1, 4], 2, 4
    m, p, gen = _build(dev, dt, q_dim, layers, kv_sizes, seed=100 + q_dim + layers)
    q, ctx_b, feats, masks = _inputs(gen, B, qside, q_dim, kv_sizes)
    w = torch.randn(B * qside * qside, 1, q_dim, generator=gen)
    ref, pr, qr, cr, fr = _oracle(p, q, ctx_b, feats, masks, B, qside, kv_sizes)
    (ref * w).sum().backward()

    qd = q.to(dev, dt).requires_grad_()
    cd = ctx_b.to(dev, dt).requires_grad_()
    fd = [f.to(dev, dt).requires_grad_() for f in feats]
    if fused:
        holders = [ops.GradAccumulator() for _ in kv_sizes]
        shared = [ops.shared_grad(f.view(-1, f.shape[-1]), h) for f, h in zip(fd, holders)]
        mu8 = [mm.to(torch.uint8).to(dev).contiguous() for mm in masks]
        out = m.forward_fused(qd.view(-1, q_dim), cd, shared, mu8, holders, B, qside).view(-1, 1, q_dim)
    else:
        ctx = cd[:, None, None, :].expand(-1, qside * qside, 1, -1).flatten(0, 1)
        kv = [_window_major(f, B, qside, s) for f, s in zip(fd, kv_sizes)]
        out = m(qd, ctx, *kv, *[mm.to(dev) for mm in masks])
    (out.float() * w.to(dev)).sum().backward()

    assert rel_err(out, ref) < FWD_TOL[name], f"forward rel err {rel_err(out, ref)}"
    assert rel_err(qd.grad, qr.grad) < BWD_TOL[name]
    assert rel_err(cd.grad, cr.grad) < BWD_TOL[name]
    for a, b in zip(fd, fr):
        assert rel_err(a.grad, b.grad) < BWD_TOL[name]
    worst = ("", 0.0)
    for n_, prm in m.named_parameters():
        assert prm.grad is not None, n_
        e = rel_err(prm.grad, pr[n_].grad)
        if e > worst[1]:
            worst = (n_, e)
    assert worst[1] < BWD_TOL[name], f"worst parameter gradient {worst}"


def test_mask_size_error_matches_reference(dev):
    """vision_sampler.py:202-206 raises ValueError on a mask/key length mismatch."""
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    m = VisionTokenSampler(1024, 1024, [1024], [2], 1024, 1).to(dev)
    q = torch.zeros(4, 1, 1024, device=dev)
    kv = torch.zeros(4, 4, 1024, device=dev)
    bad = torch.ones(4, 3, dtype=torch.bool, device=dev)
    with pytest.raises(ValueError, match="Attention mask should be of size"):
        m(q, q, kv, bad)


def test_cpu_tensor_has_no_fallback():
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    from cambrian_amd.lib import CambrianAmdError
    m = VisionTokenSampler(1024, 1024, [1024], [1], 1024, 1)
    q = torch.zeros(2, 1, 1024)
    with pytest.raises(CambrianAmdError):
        m(q, q, q, torch.ones(2, 1, dtype=torch.bool))


@pytest.mark.parametrize("H,K", [(1024, 1024), (64, 36), (130, 512)])
def test_fold_kv_matches_the_torch_expression(dev, H, K):
    """cmb_sva_fold_kv_fwd / _bwd (two launches) against the expression they replace (vision_sampler.py:173-174,188-189
    with the LayerNorm affines folded): [Wk*gk ; Wv*gv], [Wk@bk ; Wv@bv] and all six gradients, fp32."""
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(H + K)
    ts = [torch.randn(H, K, generator=g), torch.randn(K, generator=g), torch.randn(K, generator=g),
          torch.randn(H, K, generator=g), torch.randn(K, generator=g), torch.randn(K, generator=g)]
    a = [t.clone().to(dev).requires_grad_() for t in ts]
    b = [t.clone().double().requires_grad_() for t in ts]
    w, bias = ops.fold_kv(*a)
    wr = torch.cat([b[0] * b[1][None, :], b[3] * b[4][None, :]], 0)
    br = torch.cat([b[0] @ b[2], b[3] @ b[5]], 0)
    assert rel_err(w, wr.float()) < 1e-6 and rel_err(bias, br.float()) < 1e-5
    gw, gb = torch.randn(2 * H, K, generator=g), torch.randn(2 * H, generator=g)
    (w * gw.to(dev)).sum().backward(retain_graph=True)
    (bias * gb.to(dev)).sum().backward()
    ((wr * gw.double()).sum() + (br * gb.double()).sum()).backward()
    for x, y in zip(a, b):
        assert rel_err(x.grad, y.grad.float()) < 2e-5, (x.shape, rel_err(x.grad, y.grad.float()))
    # deterministic: a second backward gives the same bits
    a2 = [t.clone().to(dev).requires_grad_() for t in ts]
    w2, b2 = ops.fold_kv(*a2)
    ((w2 * gw.to(dev)).sum() + (b2 * gb.to(dev)).sum()).backward()
    w3, b3 = ops.fold_kv(*[t.clone().to(dev).requires_grad_() for t in ts])
    assert torch.equal(w2, w3) and torch.equal(b2, b3)
