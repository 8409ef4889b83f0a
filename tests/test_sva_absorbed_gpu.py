"""`-m gpu`: the absorbed form of the SVA attention core (csrc/sva_absorbed.hip + the batched per-head GEMMs of cmb_gemm)
against plain-torch restatements (tests/absorbed_ref.py; tests/test_absorbed_math.py shows on the CPU that the absorbed and
the direct restatement are the same function).  bf16 operands, fp32 references on the same rounded values."""
import pytest
import torch

from conftest import fit_err, rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    return ops, L


def test_batched_head_gemms_match_einsum(dev):
    """cmb_gemm `batch`: the two per-head projections and their gradients (K = 64, N = 64 and M = 64 problems)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(1)
    Bq, H, hd, Cin = 1000, 16, 64, 1024
    dt = torch.bfloat16
    x = torch.randn(Bq, H * hd, generator=g).to(dt).to(dev).requires_grad_()
    w = (torch.randn(H * hd, Cin, generator=g) * 0.05).to(dev).requires_grad_()          # fp32 master
    U = ops.HeadExpandFn.apply(x, w, H)
    wr = w.detach().to(dt).float()
    ref = torch.einsum("qhj,hjc->qhc", x.detach().float().view(Bq, H, hd), wr.view(H, hd, Cin))
    assert U.shape == (Bq, H, Cin) and rel_err(U, ref) < 1e-2
    gU = torch.randn(Bq, H, Cin, generator=g).to(dt).to(dev)
    U.backward(gU)
    dx_ref = torch.einsum("qhc,hjc->qhj", gU.float(), wr.view(H, hd, Cin)).reshape(Bq, H * hd)
    dw_ref = torch.einsum("qhj,qhc->hjc", x.detach().float().view(Bq, H, hd), gU.float()).reshape(H * hd, Cin)
    assert rel_err(x.grad, dx_ref) < 1e-2 and rel_err(w.grad, dw_ref) < 1e-2 and w.grad.dtype == torch.float32
    xb = torch.randn(Bq, H, Cin, generator=g).to(dt).to(dev).requires_grad_()
    w2 = (torch.randn(H * hd, Cin, generator=g) * 0.05).to(dev).requires_grad_()
    y = ops.HeadContractFn.apply(xb, w2, H)
    w2r = w2.detach().to(dt).float()
    yref = torch.einsum("qhc,hjc->qhj", xb.detach().float(), w2r.view(H, hd, Cin)).reshape(Bq, H * hd)
    assert rel_err(y, yref) < 1e-2
    gy = torch.randn(Bq, H * hd, generator=g).to(dt).to(dev)
    y.backward(gy)
    assert rel_err(xb.grad, torch.einsum("qhj,hjc->qhc", gy.float().view(Bq, H, hd), w2r.view(H, hd, Cin))) < 1e-2
    assert rel_err(w2.grad, torch.einsum("qhj,qhc->hjc", gy.float().view(Bq, H, hd), xb.detach().float()).reshape(H * hd, Cin)) < 1e-2
    # with an addend: y = addend + the per-head products inside the GEMM's epilogue (one rounding); its gradient is dy itself
    add = torch.randn(Bq, H * hd, generator=g).to(dt).to(dev).requires_grad_()
    xb2 = xb.detach().clone().requires_grad_()
    y2 = ops.HeadContractFn.apply(xb2, w2, H, add)
    assert rel_err(y2, yref + add.detach().float()) < 1e-2
    assert rel_err(y2.float(), y.detach().float() + add.detach().float()) < 8e-3      # against the two-step form: bf16 rounding
    y2.backward(gy)
    assert torch.equal(add.grad, gy) and torch.equal(xb2.grad, xb.grad)


def _case(dev, B, qside, ra, nsmall, window_major, seed, dt=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    C = 1024
    Bq, T = B * qside * qside, ra * ra
    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dt).to(dev)
    qh = rn(Bq, C)
    kvs = [rn(Bq, 2 * C) for _ in range(nsmall)]
    xh_win = rn(Bq, T, C)                                  # window-major truth
    if window_major:
        xhat = xh_win.reshape(Bq * T, C).clone()
    else:                                                  # tower-token-major: token (b, qy*ra+ry, qx*ra+rx)
        xhat = xh_win.view(B, qside, qside, ra, ra, C).permute(0, 1, 3, 2, 4, 5).reshape(B * (qside * ra) ** 2, C).contiguous()
    masks = [None] + [(torch.rand(Bq, generator=g) > 0.2).to(torch.uint8).to(dev) for _ in range(nsmall - 1)] if nsmall else []
    mask_a = (torch.rand(Bq, T, generator=g) > 0.3).to(torch.uint8)
    mask_a[:, 0] = 1
    mask_a = mask_a.to(dev)
    wk, wv = (torch.randn(C, C, generator=g) * 0.03).to(dev), (torch.randn(C, C, generator=g) * 0.03).to(dev)
    bk, bv = (torch.randn(C, generator=g) * 0.5).to(dev), (torch.randn(C, generator=g) * 0.5).to(dev)
    return dict(qh=qh, kvs=kvs, xh_win=xh_win, xhat=xhat, masks=masks, mask_a=mask_a, wk=wk, bk=bk, wv=wv, bv=bv, Bq=Bq, T=T)


@pytest.mark.parametrize("name,dt,tol_f,tol_g", [("bf16", torch.bfloat16, 1.5e-2, 3e-2), ("fp32", torch.float32, 1e-4, 5e-4)])
@pytest.mark.parametrize("B,qside,ra,nsmall,window_major", [(2, 3, 4, 3, False), (1, 5, 4, 3, True), (2, 4, 2, 1, False),
                                                          (3, 2, 4, 0, False), (1, 24, 4, 3, False)])
def test_absorbed_attention_matches_direct_reference(dev, B, qside, ra, nsmall, window_major, name, dt, tol_f, tol_g):
    """Forward and every gradient of ops.sva_absorbed_attention against the DIRECT reference (K and V projected per token,
    fp32, autograd) on the same operands; masks on both kinds of key; both token layouts.  bf16: the MFMA kernels (operands
    rounded to bf16 on both sides); fp32: the exact instantiation of the same algorithm (sva_abs_simple_*_kernel<float>, the
    per-head GEMMs on v_mfma_f32_32x32x2_f32) at the tolerance of every other fp32 kernel."""
    ops, L = _ops()
    from absorbed_ref import direct
    c = _case(dev, B, qside, ra, nsmall, window_major, 17 + B + qside + ra, dt)
    leaves_hip = dict(qh=c["qh"].clone().requires_grad_(), xhat=c["xhat"].clone().requires_grad_(),
                      wk=c["wk"].clone().requires_grad_(), bk=c["bk"].clone().requires_grad_(),
                      wv=c["wv"].clone().requires_grad_(), bv=c["bv"].clone().requires_grad_())
    kv_hip = [k.clone().requires_grad_() for k in c["kvs"]]
    out = ops.sva_absorbed_attention(leaves_hip["qh"], kv_hip, c["masks"], leaves_hip["xhat"], c["mask_a"], ra,
                                     leaves_hip["wk"], leaves_hip["bk"], leaves_hip["wv"], leaves_hip["bv"], B, qside,
                                     window_major=window_major)
    w = torch.randn(c["Bq"], 1024, generator=torch.Generator().manual_seed(5)).to(dev)
    (out.float() * w).sum().backward()
    # reference: fp32 on the bf16-rounded activations and bf16-rounded weights (the kernels cast the fp32 masters)
    r = dict(qh=c["qh"].float().requires_grad_(), xh=c["xh_win"].float().requires_grad_(),
             wk=c["wk"].to(dt).float().requires_grad_(), bk=c["bk"].clone().requires_grad_(),
             wv=c["wv"].to(dt).float().requires_grad_(), bv=c["bv"].clone().requires_grad_())
    kv_ref = [k.float().requires_grad_() for k in c["kvs"]]
    mref = [None if m is None else m.bool() for m in c["masks"]]
    ref = direct(r["qh"], kv_ref, mref, r["xh"], c["mask_a"].bool(), r["wk"], r["bk"], r["wv"], r["bv"])
    (ref * w).sum().backward()
    assert out.dtype == dt
    assert rel_err(out, ref.detach()) < tol_f, rel_err(out, ref.detach())
    assert fit_err(out, ref.detach())[0] < (5e-3 if name == "bf16" else 1e-5)
    T, Bq = c["T"], c["Bq"]
    if window_major:
        dxh = leaves_hip["xhat"].grad.view(Bq, T, 1024)
    else:
        dxh = leaves_hip["xhat"].grad.view(B, qside, ra, qside, ra, 1024).permute(0, 1, 3, 2, 4, 5).reshape(Bq, T, 1024)
    checks = [("qh", leaves_hip["qh"].grad, r["qh"].grad), ("xhat", dxh, r["xh"].grad), ("wk", leaves_hip["wk"].grad, r["wk"].grad),
              ("bk", leaves_hip["bk"].grad, r["bk"].grad), ("wv", leaves_hip["wv"].grad, r["wv"].grad),
              ("bv", leaves_hip["bv"].grad, r["bv"].grad)]
    checks += [(f"kv{i}", a.grad, b.grad) for i, (a, b) in enumerate(zip(kv_hip, kv_ref))]
    for name, a, b in checks:
        assert a is not None and b is not None, name
        if name == "bk" and nsmall == 0:
            # b_k . q_h shifts EVERY key's score of a (query, head) when no other tower is present: softmax is shift
            # invariant, the gradient is mathematically zero and both sides hold rounding noise only
            assert a.abs().max().item() < (1e-2 if name == "bf16" else 1e-4) * leaves_hip["bv"].grad.abs().max().item()
            continue
        e = rel_err(a, b)
        assert e < tol_g, (name, e)
    # masked tokens of the windowed tower: exactly zero gradient rows
    dead = ~c["mask_a"].bool()
    assert torch.count_nonzero(dxh[dead]) == 0


@pytest.mark.parametrize("B,qside,ra,nsmall,window_major", [(2, 3, 4, 3, False), (1, 6, 2, 2, True)])
def test_mfma_kernels_match_the_exact_instantiation(dev, B, qside, ra, nsmall, window_major):
    """cmb_sva_abs_fwd / _bwd on the SAME bf16 operands through the MFMA kernels and through the exact instantiation
    (CMB_KNOB_SVA_ABS = 1: plain fp32 arithmetic, probabilities never rounded): what the MFMA form costs is the bf16 rounding
    of P / dS in the token mixes and of the outputs — every output within 1e-2, the saved probabilities within 1e-5."""
    ops, L = _ops()
    c = _case(dev, B, qside, ra, nsmall, window_major, 99 + qside)
    U = ops.HeadExpandFn.apply(c["qh"], c["wk"], 16)
    res = []
    try:
        for knob in (0, 1):
            L.knob_set(L.KNOB_SVA_ABS, knob)
            leaves = [c["qh"].clone().requires_grad_(), U.detach().clone().requires_grad_(), c["xhat"].clone().requires_grad_()]
            kv = [k.clone().requires_grad_() for k in c["kvs"]]
            out, xbar, m3 = ops.SvaAbsorbedFn.apply(leaves[0], leaves[1], c["bk"], c["bv"], leaves[2], B, qside, ra,
                                                    list(c["masks"]), c["mask_a"], window_major, *kv)
            g = torch.Generator().manual_seed(7)
            w1 = torch.randn(out.shape, generator=g).to(dev)
            w2 = torch.randn(xbar.shape, generator=g).to(dev) * 0.1
            ((out.float() * w1).sum() + (xbar.float() * w2).sum()).backward()
            res.append([out.detach(), xbar.detach(), m3.detach()] + [t.grad for t in leaves + kv])
    finally:
        L.knob_set(L.KNOB_SVA_ABS, 0)
    names = ["out", "xbar", "m3", "dq", "dU", "dxhat"] + [f"dkv{i}" for i in range(nsmall)]
    for n, a, b in zip(names, res[0], res[1]):
        assert rel_err(a, b.float()) < (1e-5 if n == "m3" else 1.5e-2), (n, rel_err(a, b.float()))


def test_absorbed_path_is_what_the_layer_runs(dev, monkeypatch):
    """VisionCrossAttentionLayer takes the absorbed path for the release tower set in bf16 and gives the per-token path's
    result (CAMBRIAN_AMD_ABSORB_KV=0) within bf16 rounding, forward and parameter gradients."""
    from cambrian_amd import ops
    import cambrian_amd.model.vision_sampler as VS
    torch.manual_seed(0)
    m = VS.VisionTokenSampler(1024, 1024, [1024] * 4, [1, 1, 1, 4], 1024, 1).to(dev)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "pos_embed" in n:
                p.mul_(0.3)
    B, qside = 2, 6
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B * qside * qside, 1024, generator=g).to(dev, torch.bfloat16)
    ctx = torch.randn(B, 1024, generator=g).to(dev, torch.bfloat16)
    feats = [torch.randn(B * (qside * s) ** 2, 1024, generator=g).to(dev, torch.bfloat16) for s in (1, 1, 1, 4)]
    masks = [None, None, None, (torch.rand(B * qside * qside, 16, generator=g) > 0.25).to(torch.uint8).to(dev)]
    masks[3][:, 5] = 1
    outs, grads = [], []
    calls = []
    real = ops.sva_absorbed_attention
    monkeypatch.setattr(ops, "sva_absorbed_attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for on in (True, False):
        monkeypatch.setattr(VS, "ABSORB_KV", on)
        m.zero_grad(set_to_none=True)
        holders = [ops.GradAccumulator() for _ in range(4)]
        fd = [f.clone().requires_grad_() for f in feats]
        shared = [ops.shared_grad(f, h) for f, h in zip(fd, holders)]
        out = m.forward_fused(q, ctx, shared, masks, holders, B, qside)
        out.float().pow(2).mean().backward()
        outs.append(out.detach().float())
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()} | {f"feat{i}": f.grad.clone() for i, f in enumerate(fd)})
    assert len(calls) == 1                                   # the absorbed path ran exactly when it was on
    assert rel_err(outs[0], outs[1]) < 2e-2
    for n in grads[0]:
        assert rel_err(grads[0][n], grads[1][n].float()) < 6e-2, (n, rel_err(grads[0][n], grads[1][n].float()))
