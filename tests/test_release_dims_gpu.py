"""`-m gpu`: parity at the RELEASE dimensions (BASELINE.json configs[2], scripts/cambrian/pretrain_cambrian_8b.sh:15-27),
not only at toy sizes:

* the golden fixture the REAL reference produced at kernel-legal dimensions (tests/golden/sva_k1024.pt: hidden 1024,
  windows [1,1,1,4]) replayed straight through the HIP VisionTokenSampler — one hop, reference -> HIP;
* one full-size SVA layer — 576 queries x 10 944 keys of one image, q_dim 1024 (connector) and 4096 (in-LLM), with the
  collator's masks of a (336, 200) image — forward and every gradient against oracle/sva.py (0.5 s on the CPU);
* one real-dimension block of every tower against oracle/towers.py: CLIP-L/14 (577 x 1024), SigLIP-SO400M (729 x 1152,
  head_dim 72 -> padded 96, MLP 4304 -> padded 4352), DINOv2-g (730 x 1536, SwiGLU + LayerScale), ConvNeXt-XXL at 1024 px
  (stage 3 at 64^2 x 1536, the 5760-channel multi-stage resample), at reduced DEPTH only.

Tolerances are max-abs error / max-abs reference (conftest.rel_err), stated per test."""
import os

import pytest
import torch

from conftest import fit_err, rel_err

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = [("fp32", torch.float32), ("bf16", torch.bfloat16)]
FWD_TOL = {"fp32": 1e-4, "bf16": 3e-2}
BWD_TOL = {"fp32": 5e-4, "bf16": 6e-2}
# no systematic term (conftest.fit_err): least-squares slope against the oracle, relative L2 error
SLOPE_TOL = {"fp32": 1e-4, "bf16": 5e-3}
L2_TOL = {"fp32": 1e-4, "bf16": 1e-2}


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("q_dim", [1024, 4096])
def test_reference_golden_replays_through_hip(dev, name, dt, q_dim):
    """(the reference's own call signature: window-major kv tensors and one context row per query; the training path
    with tower-token-major features is the next test's)"""
    from golden_recipes import fill_state, grad_summary, sva_k1024_inputs
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    fx = torch.load(os.path.join(GOLD, "sva_k1024.pt"), weights_only=False)[q_dim]
    c = fx["cfg"]
    hidden, kv_sizes, qside, B = c["hidden"], c["kv_sizes"], c["qside"], c["B"]
    m = VisionTokenSampler(q_dim, hidden, [hidden] * 4, kv_sizes, hidden, 1)
    m.load_state_dict(fill_state(m.state_dict(), c["seed"]), strict=True)      # the reference's own key names
    m = m.to(dev)
    q, ctx, kvs, masks, w = sva_k1024_inputs(q_dim, hidden, kv_sizes, qside, B, c["seed"])
    qd = q.to(dev, dt).requires_grad_()
    cd = ctx.to(dev, dt).requires_grad_()
    kd = [k.to(dev, dt).requires_grad_() for k in kvs]
    out = m(qd, cd, *kd, *[mk.to(dev) for mk in masks])
    (out.float() * w.to(dev)).sum().backward()
    assert rel_err(out, fx["out"]) < FWD_TOL[name], rel_err(out, fx["out"])
    assert rel_err(qd.grad, fx["dq"]) < BWD_TOL[name]
    assert rel_err(cd.grad, fx["dctx"]) < BWD_TOL[name]
    for a, b in zip(kd, fx["dkvs"]):
        assert rel_err(a.grad, b) < BWD_TOL[name]
    for n_, prm in m.named_parameters():
        got, want = grad_summary(prm.grad, n_, c["seed"]), fx["dparams"][n_]
        scale = float(want[1]) + 1e-6                       # the gradient's l2 norm
        assert abs(float(got[1]) - float(want[1])) < BWD_TOL[name] * scale, n_
        # a unit-variance projection of an error vector e has magnitude ~|e|: allow 4 sigma
        assert (got[2:] - want[2:]).abs().max().item() < 4 * BWD_TOL[name] * scale, n_


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("q_dim", [1024, 4096])
def test_sva_layer_release_dims_matches_oracle(dev, name, dt, q_dim):
    """576 queries x (576 + 576 + 576 + 9216) keys, a letter-boxed image's masks from the collator layout code."""
    from cambrian_amd import ops
    from cambrian_amd.train.data_layout import sva_window_mask
    from test_sva_gpu import _build, _oracle
    kv_sizes, B, qside, hidden = [1, 1, 1, 4], 1, 24, 1024
    m, p, gen = _build(dev, dt, q_dim, 1, kv_sizes, seed=900 + q_dim)
    Bq = B * qside * qside
    q = torch.randn(Bq, 1, q_dim, generator=gen)
    ctx_b = torch.randn(B, hidden, generator=gen)
    feats = [torch.randn(B, (qside * s) ** 2, hidden, generator=gen) for s in kv_sizes]
    # a (336, 200) image: 19 of the 96 rows of the 4 x 4-window tower are padding on each side, i.e. windows whose
    # first rows only are masked (a (336, 224) image pads whole windows, which the collator re-opens: all True)
    masks = [sva_window_mask((336, 200), qside, qside * s) for s in kv_sizes]          # bool [576, s*s], window-major
    assert not masks[3].all() and all(mk.any(1).all() for mk in masks)
    w = torch.randn(Bq, 1, q_dim, generator=gen)
    ref, pr, qr, cr, fr = _oracle(p, q, ctx_b, feats, masks, B, qside, kv_sizes)
    (ref * w).sum().backward()

    qd = q.to(dev, dt).requires_grad_()
    cd = ctx_b.to(dev, dt).requires_grad_()
    fd = [f.to(dev, dt).requires_grad_() for f in feats]
    holders = [ops.GradAccumulator() for _ in kv_sizes]
    shared = [ops.shared_grad(f.view(-1, hidden), h) for f, h in zip(fd, holders)]
    mu8 = [mk.to(torch.uint8).to(dev).contiguous() for mk in masks]
    out = m.forward_fused(qd.view(-1, q_dim), cd, shared, mu8, holders, B, qside).view(-1, 1, q_dim)
    (out.float() * w.to(dev)).sum().backward()
    assert rel_err(out, ref) < FWD_TOL[name], rel_err(out, ref)
    assert fit_err(out, ref)[0] < SLOPE_TOL[name] and fit_err(out, ref)[1] < L2_TOL[name], fit_err(out, ref)
    assert rel_err(qd.grad, qr.grad) < BWD_TOL[name]
    assert rel_err(cd.grad, cr.grad) < BWD_TOL[name]
    for a, b in zip(fd, fr):
        assert rel_err(a.grad, b.grad) < BWD_TOL[name]
        assert fit_err(a.grad, b.grad)[0] < 2 * SLOPE_TOL[name], fit_err(a.grad, b.grad)
    worst = max(((n_, rel_err(prm.grad, pr[n_].grad)) for n_, prm in m.named_parameters()), key=lambda t: t[1])
    assert worst[1] < BWD_TOL[name], f"worst parameter gradient {worst}"
    # masked keys of the padded rows must carry exactly zero gradient
    dead = ~masks[3].view(qside, qside, 4, 4).permute(0, 2, 1, 3).reshape(-1)        # token-major [96*96]
    assert dead.any() and torch.count_nonzero(fd[3].grad[0][dead.to(dev)]) == 0


TOWER_TOL = {"fp32": 2e-4, "bf16": 4e-2}


def _release_vit(kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig
    if kind == "clip_l_336":      # openai/clip-vit-large-patch14-336: 577 tokens x 1024, 16 heads x 64, quick_gelu 4096
        return ViTConfig(image_size=336, patch_size=14, hidden_size=1024, num_layers=2, num_heads=16, mlp_dim=4096,
                         act="quick_gelu", ln_eps=1e-5, has_cls=True, pre_ln=True, final_ln=False, patch_bias=False,
                         run_layers=1)
    if kind == "so400m_384":      # SigLIP SO400M/14@384: 729 x 1152, 16 heads x 72, MLP 4304
        return ViTConfig(image_size=384, patch_size=14, hidden_size=1152, num_layers=1, num_heads=16, mlp_dim=4304,
                         act="gelu", ln_eps=1e-6, has_cls=False, final_ln=True)
    if kind == "dinov2_g_378":    # facebook/dinov2-giant @378: 730 x 1536, 24 heads x 64, SwiGLU 4096, LayerScale
        return ViTConfig(image_size=378, patch_size=14, hidden_size=1536, num_layers=1, num_heads=24, mlp_dim=4096,
                         act="swiglu", ln_eps=1e-6, has_cls=True, final_ln=True, layerscale=True)
    raise KeyError(kind)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("kind", ["clip_l_336", "so400m_384", "dinov2_g_378"])
def test_vit_block_release_dims_matches_oracle(dev, name, dt, kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk, resample_tokens
    from oracle import towers as O
    cfg = _release_vit(kind)
    gen = torch.Generator().manual_seed(sum(map(ord, kind)))
    p = ViTTrunk.random_canonical(cfg, gen)
    img = torch.randn(1, 3, cfg.image_size, cfg.image_size, generator=gen)
    ref = O.vit_forward(cfg, p, img)
    out = ViTTrunk(cfg, dt).load_canonical(p, dev)(img.to(dev))
    assert out.shape == ref.shape and out.shape[1] == {"clip_l_336": 576, "so400m_384": 729, "dinov2_g_378": 729}[kind]
    assert rel_err(out, ref) < TOWER_TOL[name], rel_err(out, ref)
    assert fit_err(out, ref)[0] < SLOPE_TOL[name] and fit_err(out, ref)[1] < 2 * L2_TOL[name], fit_err(out, ref)
    if out.shape[1] != 576:      # the 27^2 -> 24^2 bilinear token resize of the SigLIP / DINOv2 wrappers
        assert rel_err(resample_tokens(out, 576, force_copy=True), O.interpolate_tokens(ref, 576)) < TOWER_TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
def test_convnext_xxl_release_dims_matches_oracle(dev, name, dt):
    """ConvNeXt-XXL widths (384, 768, 1536, 3072) at 1024 px, one block per stage: stage maps 256^2 / 128^2 / 64^2 / 32^2 and
    the multi-stage resample to 96^2 x 5760 (clip_convnext_encoder.py:99-144)."""
    from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
    from oracle import towers as O
    cfg = ConvNeXtConfig(depths=(1, 1, 1, 1), dims=(384, 768, 1536, 3072), ln_eps=1e-5)
    gen = torch.Generator().manual_seed(1024)
    p = ConvNeXtTrunk.random_canonical(cfg, gen)
    img = torch.randn(1, 3, 1024, 1024, generator=gen)
    trunk = ConvNeXtTrunk(cfg, dt).load_canonical(p, dev)
    stages = trunk.forward_stages(img.to(dev))
    refs = O.convnext_stages(cfg, p, img)
    for a, b, side in zip(stages, refs, (256, 128, 64, 32)):
        assert a.shape[1] == side and rel_err(a.permute(0, 3, 1, 2), b) < TOWER_TOL[name]
    out = trunk(img.to(dev), 96, multi_stage=True)
    ref = O.convnext_forward(cfg, p, img, 96, multi_stage=True)
    assert out.shape == (1, 9216, 5760) and rel_err(out, ref) < TOWER_TOL[name]
