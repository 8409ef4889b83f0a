"""`-m gpu`: trainable vision towers (``--unfreeze_mm_vision_tower``): the autograd trunk
(vit_train.py) against the CPU oracle differentiated by autograd — outputs and the gradient of EVERY tower parameter —
plus the reference-protocol wrappers in unfrozen mode at release size."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err
from test_towers_gpu import _vit_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["clip", "siglip", "siglip_tanh", "dino"])
def test_trainable_vit_forward_and_all_parameter_gradients(dev, kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk
    from cambrian_amd.model.multimodal_encoder.vit_train import TrainableViT, resample_tokens_autograd
    from oracle import towers as O
    cfg = _vit_case(kind)
    gen = torch.Generator().manual_seed(sum(map(ord, kind)) + 1)
    canon = ViTTrunk.random_canonical(cfg, gen)
    img = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=gen)
    tgt = (cfg.grid - 2) ** 2
    w = torch.randn(2, tgt, cfg.hidden_size, generator=gen)
    # oracle: fp32 autograd through the CPU restatement (+ the wrappers' token-grid resize)
    p_ref = {k: v.clone().requires_grad_() for k, v in canon.items()}
    ref = O.interpolate_tokens(O.vit_forward(cfg, p_ref, img), tgt)
    (ref * w).sum().backward()
    tower = TrainableViT(cfg, canon, dev)
    out = resample_tokens_autograd(tower(img.to(dev)), tgt)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert rel_err(out, ref) < 5e-2
    (out.float() * w.to(dev)).sum().backward()
    nl = cfg.run_layers if cfg.run_layers is not None else cfg.num_layers
    checked, worst = 0, ("", 0.0)
    for name, t in p_ref.items():
        if name.startswith("layers.") and int(name.split(".")[1]) >= nl:
            assert tower.P(name).grad is None           # behind select_layer: a parameter (checkpointed), never run
            continue
        g = tower.P(name).grad
        assert g is not None, name
        assert t.grad is not None and t.grad.abs().max() > 0, name
        checked += 1
        if name.endswith(".k.bias"):
            # softmax is invariant to adding q.b_k to every score of a query: the true gradient is 0, both sides hold
            # rounding noise only — compare against the scale of the query bias gradient instead
            qb = p_ref[name.replace(".k.bias", ".q.bias")].grad.abs().max()
            assert t.grad.abs().max() < 1e-4 * qb and g.abs().max() < 5e-2 * qb, name
            continue
        e = rel_err(g, t.grad)
        if e > worst[1]:
            worst = (name, e)
    assert checked > 20 and checked == sum(1 for q in tower.p.values() if q.grad is not None)
    assert worst[1] < 1.2e-1, f"worst tower-parameter gradient {worst}"


def test_trainable_convnext_forward_and_all_parameter_gradients(dev):
    from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
    from cambrian_amd.model.multimodal_encoder.convnext_train import TrainableConvNeXt
    from oracle import towers as O
    cfg = ConvNeXtConfig(depths=(1, 1, 2, 1), dims=(64, 128, 256, 512), ln_eps=1e-5)
    gen = torch.Generator().manual_seed(78)
    canon = ConvNeXtTrunk.random_canonical(cfg, gen)
    img = torch.randn(2, 3, 160, 160, generator=gen)           # stage maps 40 / 20 / 10 / 5: ragged 8x8 wgrad tiles
    w = torch.randn(2, 144, 960, generator=gen)
    p_ref = {k: v.clone().requires_grad_() for k, v in canon.items()}
    ref = O.convnext_forward(cfg, p_ref, img, 12, multi_stage=True)
    (ref * w).sum().backward()
    tower = TrainableConvNeXt(cfg, canon, dev)
    out = tower(img.to(dev), 12, multi_stage=True)
    assert out.shape == ref.shape and rel_err(out, ref) < 5e-2
    (out.float() * w.to(dev)).sum().backward()
    worst = ("", 0.0)
    for name, t in p_ref.items():
        g = tower.P(name).grad
        assert g is not None and t.grad is not None and t.grad.abs().max() > 0, name
        e = rel_err(g, t.grad)
        if e > worst[1]:
            worst = (name, e)
    assert len(tower.p) == len(p_ref) > 50
    assert worst[1] < 1.2e-1, f"worst tower-parameter gradient {worst}"


def test_dwconv_backward_kernels(dev):
    """dX (forward kernel on dY with reversed taps), dW (cmb_dwconv7x7_wgrad + column sum) and d(bias) of the depthwise
    7x7 against torch's conv2d autograd in fp32, on a ragged map."""
    from cambrian_amd.model.multimodal_encoder.convnext_train import DwConv7x7Fn
    import torch.nn.functional as F
    B, H, W, C = 3, 19, 13, 128
    g_ = torch.Generator().manual_seed(4)
    x = torch.randn(B, H, W, C, generator=g_).to(dev).requires_grad_()
    w49 = (torch.randn(49, C, generator=g_) * 0.2).to(dev).requires_grad_()
    b = torch.randn(C, generator=g_).to(dev).requires_grad_()
    dy = torch.randn(B, H, W, C, generator=g_).to(dev)
    y = DwConv7x7Fn.apply(x, w49, b)
    y.backward(dy)
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w49, b))
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr.t().reshape(C, 1, 7, 7), br, padding=3, groups=C).permute(0, 2, 3, 1)
    yr.backward(dy)
    assert rel_err(y, yr.detach().cpu()) < 1e-5
    assert rel_err(x.grad, xr.grad.cpu()) < 1e-5
    assert rel_err(w49.grad, wr.grad.cpu()) < 1e-5
    assert rel_err(b.grad, br.grad.cpu()) < 1e-5


def test_trainable_vit_pos_fn_reaches_native_parameter(dev):
    """DINOv2: the 37x37 position grid is bicubically resized inside the forward; its gradient lands on the native rows."""
    from cambrian_amd.model.multimodal_encoder.dino_encoder import interpolate_pos_encoding
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig, ViTTrunk
    from cambrian_amd.model.multimodal_encoder.vit_train import TrainableViT
    from oracle import towers as O
    kw = dict(patch_size=14, hidden_size=128, num_layers=1, num_heads=2, mlp_dim=256, act="swiglu", ln_eps=1e-6,
              has_cls=True, final_ln=True, layerscale=True)
    native, run = ViTConfig(image_size=126, **kw), ViTConfig(image_size=98, **kw)      # 9x9 -> 7x7 grid
    gen = torch.Generator().manual_seed(3)
    canon = ViTTrunk.random_canonical(native, gen)
    img = torch.randn(2, 3, 98, 98, generator=gen)
    fn = lambda pos: interpolate_pos_encoding(pos, run.grid)  # noqa: E731
    p_ref = {k: v.clone().requires_grad_() for k, v in canon.items()}
    ref = O.vit_forward(run, {**p_ref, "pos": fn(p_ref["pos"])}, img)
    ref.square().sum().backward()
    tower = TrainableViT(run, canon, dev, pos_fn=fn)
    out = tower(img.to(dev))
    assert rel_err(out, ref) < 5e-2
    out.float().square().sum().backward()
    assert tower.P("pos").shape == (82, 128) and rel_err(tower.P("pos").grad, p_ref["pos"].grad) < 1.2e-1


def test_unfrozen_wrapper_equals_frozen_wrapper_and_trains(dev):
    """ClipVisionTower with unfreeze_mm_vision_tower (clip_encoder.py:103) at release size: same features as the frozen
    wrapper built from the same seed, is an nn.Module with parameters, back-propagates into all of them, and an
    optimizer group with mm_vision_tower_lr (cambrian_trainer.py:319-348) updates them."""
    from cambrian_amd.model.multimodal_encoder.clip_encoder import ClipVisionTower
    from cambrian_amd.model.multimodal_encoder.vit_train import tower_param_groups
    name = "openai/clip-vit-large-patch14-336"
    frozen = ClipVisionTower(name, SimpleNamespace(mm_vision_select_layer=-2, unfreeze_mm_vision_tower=False))
    train = ClipVisionTower(name, SimpleNamespace(mm_vision_select_layer=-2, unfreeze_mm_vision_tower=True))
    assert sum(p.numel() for p in frozen.parameters()) == 0
    n_par = sum(p.numel() for p in train.parameters())
    assert 3.0e8 < n_par < 3.1e8                                   # all of CLIP-L's 24 layers (the last never runs)
    img = torch.randn(2, 3, 336, 336, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16)
    f0 = frozen(img)
    f1 = train(img)
    assert f0.shape == f1.shape == (2, 576, 1024) and not f0.requires_grad and f1.requires_grad
    assert rel_err(f1, f0.float().cpu()) < 3e-2
    holder = torch.nn.Module()
    holder.vision_tower_aux_list = torch.nn.ModuleList([train])
    holder.other = torch.nn.Linear(4, 4).to(dev)
    groups = tower_param_groups(holder, base_lr=1e-3, tower_lr=1e-5)
    assert len(groups) == 2 and groups[1]["lr"] == 1e-5 and len(groups[1]["params"]) == len(list(train.parameters()))
    opt = torch.optim.AdamW(groups)
    before = train.vision_tower.P("layers.0.fc1.weight").detach().clone()
    f1.float().square().mean().backward()
    got = {n for n, p in train.named_parameters() if p.grad is not None}
    assert all(("layers__23__" in n) != (n in got) for n, _ in train.named_parameters())     # all but the unused last layer
    opt.step()
    assert not torch.equal(before, train.vision_tower.P("layers.0.fc1.weight").detach())
    with torch.no_grad():
        assert not train(img).requires_grad


def test_unfrozen_convnext_wrapper(dev):
    from cambrian_amd.model.multimodal_encoder.clip_convnext_encoder import CLIPConvNextTower
    from cambrian_amd.model.multimodal_encoder.convnext_train import TrainableConvNeXt
    t = CLIPConvNextTower("clip-convnext-L-multi-stage-res256-interp144", SimpleNamespace(unfreeze_mm_vision_tower=True))
    assert isinstance(t.vision_tower, TrainableConvNeXt) and sum(p.numel() for p in t.parameters()) > 1e8
    img = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)
    f = t(img)
    assert f.requires_grad and f.shape[:2] == (1, 144)
    f.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in t.parameters())


@pytest.mark.parametrize("which", ["vit", "convnext"])
def test_tower_block_recompute_is_the_same_function(dev, which):
    """Per-block activation re-computation inside a trainable tower (``recompute``: VERDICT r3 next #5 — the reference batch
    of 8 images with the towers unfrozen needs it): output bit-identical and every parameter gradient identical up to the
    summation order of the atomically reduced bias / LayerNorm gradients (the same kernels run twice)."""
    gen = torch.Generator().manual_seed(5)
    if which == "vit":
        from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk
        from cambrian_amd.model.multimodal_encoder.vit_train import TrainableViT
        cfg = _vit_case("dino")
        canon = ViTTrunk.random_canonical(cfg, gen)
        img = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=gen).to(dev)
        make = lambda: TrainableViT(cfg, canon, dev)  # noqa: E731
        run = lambda t: t(img)  # noqa: E731
    else:
        from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
        from cambrian_amd.model.multimodal_encoder.convnext_train import TrainableConvNeXt
        cfg = ConvNeXtConfig(depths=(2, 2, 2, 1), dims=(64, 64, 128, 128), ln_eps=1e-5)
        canon = ConvNeXtTrunk.random_canonical(cfg, gen)
        img = torch.randn(2, 3, 64, 64, generator=gen).to(dev)
        make = lambda: TrainableConvNeXt(cfg, canon, dev)  # noqa: E731
        run = lambda t: t(img, 8, multi_stage=True)  # noqa: E731
    res = []
    for rec in (False, True):
        tower = make()
        tower.recompute = rec
        torch.cuda.reset_peak_memory_stats()
        out = run(tower)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(dev)
        (out.float() * w).sum().backward()
        res.append((out.detach(), {k: v.grad for k, v in tower.p.items()}))
    assert torch.equal(res[0][0], res[1][0])
    for k, g in res[0][1].items():
        g2 = res[1][1][k]
        assert (g is None) == (g2 is None), k
        if g is not None:   # (bias / LayerNorm-parameter gradients are fp32 atomic sums: order-dependent in the last bits)
            assert rel_err(g2, g) < 2e-5, (k, rel_err(g2, g))
