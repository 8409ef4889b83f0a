"""`-m gpu`: the native tower trunks (HIP kernels) against oracle/towers.py on seeded weights, plus the
reference-protocol wrappers at the release-8B shapes."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-4, "bf16": 4e-2}
DTYPES = [("fp32", torch.float32), ("bf16", torch.bfloat16)]


def _vit_case(kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig
    if kind == "clip":      # pre-LN, CLS, quick_gelu, no patch bias, stop one layer early (select_layer -2)
        return ViTConfig(image_size=112, patch_size=14, hidden_size=256, num_layers=3, num_heads=4, mlp_dim=512,
                         act="quick_gelu", ln_eps=1e-5, has_cls=True, pre_ln=True, final_ln=False, patch_bias=False,
                         run_layers=2)
    if kind == "siglip":    # no CLS, head_dim 72 (-> padded to 96), MLP width not a multiple of 64 (-> padded)
        # image side not a multiple of the patch (SO400M: 384 px / 14 -> 27 patches, 6 px dropped)
        return ViTConfig(image_size=118, patch_size=14, hidden_size=576, num_layers=2, num_heads=8, mlp_dim=1080,
                         act="gelu", ln_eps=1e-6, has_cls=False, final_ln=True)
    if kind == "siglip_tanh":
        return ViTConfig(image_size=112, patch_size=14, hidden_size=256, num_layers=2, num_heads=4, mlp_dim=512,
                         act="gelu_tanh", ln_eps=1e-6, has_cls=False, final_ln=True)
    if kind == "dino":      # CLS, LayerScale, SwiGLU
        return ViTConfig(image_size=126, patch_size=14, hidden_size=384, num_layers=2, num_heads=6, mlp_dim=1024,
                         act="swiglu", ln_eps=1e-6, has_cls=True, final_ln=True, layerscale=True)
    raise KeyError(kind)


@pytest.mark.parametrize("name,dt", DTYPES)
@pytest.mark.parametrize("kind", ["clip", "siglip", "siglip_tanh", "dino"])
def test_vit_trunk_matches_oracle(dev, name, dt, kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk, resample_tokens
    from oracle import towers as O
    cfg = _vit_case(kind)
    gen = torch.Generator().manual_seed(sum(map(ord, kind)))
    p = ViTTrunk.random_canonical(cfg, gen)
    img = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=gen)
    ref = O.vit_forward(cfg, p, img)
    trunk = ViTTrunk(cfg, dt).load_canonical(p, dev)
    out = trunk(img.to(dev))
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[name], rel_err(out, ref)
    # token-grid resize of the wrappers (clip_encoder.py:70-96)
    tgt = (cfg.grid - 2) ** 2
    assert rel_err(resample_tokens(out, tgt, force_copy=True), O.interpolate_tokens(ref, tgt)) < TOL[name]


@pytest.mark.parametrize("name,dt", DTYPES)
def test_convnext_trunk_matches_oracle(dev, name, dt):
    from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
    from oracle import towers as O
    cfg = ConvNeXtConfig(depths=(1, 1, 2, 1), dims=(64, 128, 256, 512), ln_eps=1e-5)
    gen = torch.Generator().manual_seed(77)
    p = ConvNeXtTrunk.random_canonical(cfg, gen)
    img = torch.randn(2, 3, 128, 128, generator=gen)
    trunk = ConvNeXtTrunk(cfg, dt).load_canonical(p, dev)
    stages = trunk.forward_stages(img.to(dev))
    for a, b in zip(stages, O.convnext_stages(cfg, p, img)):
        assert rel_err(a.permute(0, 3, 1, 2), b) < TOL[name]
    out = trunk(img.to(dev), 12, multi_stage=True)          # clip_convnext_encoder.py:121-144
    ref = O.convnext_forward(cfg, p, img, 12, multi_stage=True)
    assert out.shape == (2, 144, 960) and rel_err(out, ref) < TOL[name]
    last = trunk(img.to(dev), None, multi_stage=False)
    assert rel_err(last, O.convnext_forward(cfg, p, img, None, multi_stage=False)) < TOL[name]


def test_release_8b_tower_shapes(dev):
    """The four production towers at the release shapes: [B,576,{1152,1024,1536}] and [B,9216,5760]
    (scripts/cambrian/pretrain_cambrian_8b.sh:15-27)."""
    from cambrian_amd.model.multimodal_encoder.builder import build_vision_tower_aux_list
    cfg = SimpleNamespace(mm_vision_tower_aux_list=["siglip/CLIP-ViT-SO400M-14-384", "openai/clip-vit-large-patch14-336",
                                                    "facebook/dinov2-giant-res378", "clip-convnext-XXL-multi-stage"],
                          mm_vision_tower_aux_token_len_list=[576, 576, 576, 9216], mm_vision_select_layer=-2,
                          mm_vision_select_feature="patch", unfreeze_mm_vision_tower=False)
    towers = build_vision_tower_aux_list(cfg)
    want = [(576, 1152, 384), (576, 1024, 336), (576, 1536, 378), (9216, 5760, 1024)]
    for t, (tok, hid, res) in zip(towers, want):
        assert t.is_loaded and t.num_patches == tok and t.hidden_size == hid and t.image_size == res
        x = torch.randn(1, 3, res, res, device=dev, dtype=torch.bfloat16)
        y = t(x)
        assert y.shape == (1, tok, hid) and y.dtype == torch.bfloat16 and y.is_contiguous()
        assert torch.isfinite(y.float()).all()
        assert not any(p.requires_grad for p in t.parameters())  # frozen, and absent from state_dict
        assert len(t.state_dict()) == 0


def test_unknown_tower_raises():
    from cambrian_amd.model.multimodal_encoder.builder import build_vision_tower
    with pytest.raises(ValueError, match="Unknown vision tower"):
        build_vision_tower(SimpleNamespace(mm_vision_tower="eva/clip-thing", mm_vision_select_layer=-2), delay_load=True)
