"""CPU: an unfrozen tower is checkpointed under the reference module's own key names (SURVEY.md §8f N4 / §8b:
``model.vision_tower_aux_list.{i}.vision_tower.<HF / timm keys>``) — canonical -> reference -> canonical round trips for
the four module types, the state_dict()/load_state_dict() hooks of the trainable trunks, and pass-through of reference
tensors the hot path never reads."""
import torch

from cambrian_amd.model.multimodal_encoder import weight_maps as WM
from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
from cambrian_amd.model.multimodal_encoder.vit import ViTConfig, ViTTrunk


def _vit(**kw):
    base = dict(image_size=56, patch_size=14, hidden_size=64, num_layers=2, num_heads=2, mlp_dim=128)
    base.update(kw)
    return ViTConfig(**base)


def _same(a, b):
    assert a.keys() == b.keys(), sorted(set(a) ^ set(b))
    for k in a:
        assert torch.equal(a[k].reshape(-1), b[k].reshape(-1)), k


def test_round_trips():
    g = torch.Generator().manual_seed(0)
    clip = _vit(act="quick_gelu", has_cls=True, pre_ln=True, final_ln=False, patch_bias=False)
    p = ViTTrunk.random_canonical(clip, g)
    ref = WM.canonical_to_hf_clip(p, 2)
    assert "vision_model.encoder.layers.1.self_attn.q_proj.weight" in ref and "vision_model.pre_layrnorm.bias" in ref
    _same(WM.hf_clip_to_canonical(ref, 2), p)
    sig = _vit(act="gelu", has_cls=False, final_ln=True)
    p = ViTTrunk.random_canonical(sig, g)
    ref = WM.canonical_to_timm_vit(p, 2)
    assert ref["blocks.0.attn.qkv.weight"].shape == (192, 64) and ref["pos_embed"].shape == (1, 16, 64)
    _same(WM.timm_vit_to_canonical(ref, 2), p)
    dino = _vit(act="swiglu", has_cls=True, final_ln=True, layerscale=True)
    p = ViTTrunk.random_canonical(dino, g)
    ref = WM.canonical_to_hf_dinov2(p, 2, True)
    assert ref["embeddings.cls_token"].shape == (1, 1, 64) and "encoder.layer.1.mlp.weights_in.weight" in ref
    _same(WM.hf_dinov2_to_canonical(ref, 2, True), p)
    cn = ConvNeXtConfig(depths=(1, 1, 1, 1), dims=(64, 64, 128, 128), ln_eps=1e-5)
    p = ConvNeXtTrunk.random_canonical(cn, g)
    ref = WM.canonical_to_timm_convnext(p, cn.depths)
    assert "stages.2.downsample.1.weight" in ref and "stages.0.blocks.0.conv_dw.weight" in ref
    _same(WM.timm_convnext_to_canonical(ref, cn.depths), p)


def test_state_dict_hooks_use_reference_keys_and_keep_extras():
    from cambrian_amd.model.multimodal_encoder.vit_train import TrainableViT
    cfg = _vit(act="quick_gelu", has_cls=True, pre_ln=True, final_ln=False, patch_bias=False, run_layers=1)
    g = torch.Generator().manual_seed(1)
    holder = torch.nn.Module()
    holder.vision_tower = TrainableViT(cfg, ViTTrunk.random_canonical(cfg, g), "cpu")
    WM.ReferenceKeys(lambda p: WM.canonical_to_hf_clip(p, 2), lambda sd: WM.hf_clip_to_canonical(sd, 2)).install(holder.vision_tower)
    sd = holder.state_dict()
    assert "vision_tower.vision_model.encoder.layers.1.mlp.fc2.bias" in sd          # also the layer behind select_layer
    assert not any(".p." in k for k in sd)
    # a reference checkpoint: other values + tensors the hot path never reads (CLIP's post_layernorm)
    ckpt = {k: torch.randn_like(v) for k, v in sd.items()}
    ckpt["vision_tower.vision_model.post_layernorm.weight"] = torch.randn(64)
    ckpt["vision_tower.vision_model.post_layernorm.bias"] = torch.randn(64)
    missing, unexpected = holder.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    assert torch.equal(holder.vision_tower.P("layers.0.q.weight"),
                       ckpt["vision_tower.vision_model.encoder.layers.0.self_attn.q_proj.weight"])
    out = holder.state_dict()
    assert out.keys() == ckpt.keys()
    for k in ckpt:
        assert torch.equal(out[k], ckpt[k]), k
    # native naming is accepted too
    native = {"vision_tower.p." + k.replace(".", "__"): v for k, v in holder.vision_tower.canonical_state().items()}
    holder.load_state_dict(native, strict=True)
