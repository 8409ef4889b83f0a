"""State-dict key maps from the third-party checkpoints the reference loads (HF ``CLIPVisionModel`` /
``Dinov2Model`` at transformers==4.37.0; timm ViT / ConvNeXt through open_clip; HF ``SiglipVisionModel`` /
``ConvNextModel`` as arithmetic stand-ins) onto the canonical names consumed by
``ViTTrunk.load_canonical`` / ``ConvNeXtTrunk.load_canonical``.  Used by the golden tests today and by
real-weight loading once checkpoints are reachable."""
from __future__ import annotations

from typing import Dict

import torch


def _strip(sd: Dict[str, torch.Tensor], prefixes=("vision_model.", "visual.trunk.", "trunk.")) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


def hf_clip_to_canonical(sd: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    sd = _strip(sd)
    p = {"patch.weight": sd["embeddings.patch_embedding.weight"], "cls": sd["embeddings.class_embedding"],
         "pos": sd["embeddings.position_embedding.weight"],
         "pre_ln.weight": sd["pre_layrnorm.weight"], "pre_ln.bias": sd["pre_layrnorm.bias"]}
    for l in range(num_layers):
        s, d = f"encoder.layers.{l}.", f"layers.{l}."
        for a, b in (("layer_norm1", "ln1"), ("layer_norm2", "ln2"), ("self_attn.q_proj", "q"), ("self_attn.k_proj", "k"),
                     ("self_attn.v_proj", "v"), ("self_attn.out_proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
            p[d + b + ".weight"] = sd[s + a + ".weight"]
            p[d + b + ".bias"] = sd[s + a + ".bias"]
    return p


def hf_siglip_to_canonical(sd: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    sd = _strip(sd)
    p = {"patch.weight": sd["embeddings.patch_embedding.weight"], "patch.bias": sd["embeddings.patch_embedding.bias"],
         "pos": sd["embeddings.position_embedding.weight"],
         "final_ln.weight": sd["post_layernorm.weight"], "final_ln.bias": sd["post_layernorm.bias"]}
    for l in range(num_layers):
        s, d = f"encoder.layers.{l}.", f"layers.{l}."
        for a, b in (("layer_norm1", "ln1"), ("layer_norm2", "ln2"), ("self_attn.q_proj", "q"), ("self_attn.k_proj", "k"),
                     ("self_attn.v_proj", "v"), ("self_attn.out_proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
            p[d + b + ".weight"] = sd[s + a + ".weight"]
            p[d + b + ".bias"] = sd[s + a + ".bias"]
    return p


def timm_vit_to_canonical(sd: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    """timm VisionTransformer (open_clip ``visual.trunk``) as used for SigLIP-SO400M: fused qkv, no CLS."""
    sd = _strip(sd)
    p = {"patch.weight": sd["patch_embed.proj.weight"], "patch.bias": sd["patch_embed.proj.bias"],
         "pos": sd["pos_embed"][0], "final_ln.weight": sd["norm.weight"], "final_ln.bias": sd["norm.bias"]}
    for l in range(num_layers):
        s, d = f"blocks.{l}.", f"layers.{l}."
        D = sd[s + "attn.qkv.weight"].shape[1]
        for i, n in enumerate("qkv"):
            p[d + n + ".weight"] = sd[s + "attn.qkv.weight"][i * D:(i + 1) * D]
            p[d + n + ".bias"] = sd[s + "attn.qkv.bias"][i * D:(i + 1) * D]
        for a, b in (("norm1", "ln1"), ("norm2", "ln2"), ("attn.proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
            p[d + b + ".weight"] = sd[s + a + ".weight"]
            p[d + b + ".bias"] = sd[s + a + ".bias"]
    return p


def hf_dinov2_to_canonical(sd: Dict[str, torch.Tensor], num_layers: int, swiglu: bool) -> Dict[str, torch.Tensor]:
    p = {"patch.weight": sd["embeddings.patch_embeddings.projection.weight"],
         "patch.bias": sd["embeddings.patch_embeddings.projection.bias"],
         "cls": sd["embeddings.cls_token"].reshape(-1), "pos": sd["embeddings.position_embeddings"][0],
         "final_ln.weight": sd["layernorm.weight"], "final_ln.bias": sd["layernorm.bias"]}
    for l in range(num_layers):
        s, d = f"encoder.layer.{l}.", f"layers.{l}."
        pairs = [("norm1", "ln1"), ("norm2", "ln2"), ("attention.attention.query", "q"), ("attention.attention.key", "k"),
                 ("attention.attention.value", "v"), ("attention.output.dense", "proj")]
        pairs += [("mlp.weights_in", "fc1"), ("mlp.weights_out", "fc2")] if swiglu else [("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")]
        for a, b in pairs:
            p[d + b + ".weight"] = sd[s + a + ".weight"]
            p[d + b + ".bias"] = sd[s + a + ".bias"]
        p[d + "ls1"] = sd[s + "layer_scale1.lambda1"]
        p[d + "ls2"] = sd[s + "layer_scale2.lambda1"]
    return p


def hf_convnext_to_canonical(sd: Dict[str, torch.Tensor], depths) -> Dict[str, torch.Tensor]:
    p = {"stem.conv.weight": sd["embeddings.patch_embeddings.weight"], "stem.conv.bias": sd["embeddings.patch_embeddings.bias"],
         "stem.ln.weight": sd["embeddings.layernorm.weight"], "stem.ln.bias": sd["embeddings.layernorm.bias"]}
    for s, depth in enumerate(depths):
        if s > 0:
            p[f"stages.{s}.down.ln.weight"] = sd[f"encoder.stages.{s}.downsampling_layer.0.weight"]
            p[f"stages.{s}.down.ln.bias"] = sd[f"encoder.stages.{s}.downsampling_layer.0.bias"]
            p[f"stages.{s}.down.conv.weight"] = sd[f"encoder.stages.{s}.downsampling_layer.1.weight"]
            p[f"stages.{s}.down.conv.bias"] = sd[f"encoder.stages.{s}.downsampling_layer.1.bias"]
        for b in range(depth):
            src, dst = f"encoder.stages.{s}.layers.{b}.", f"stages.{s}.blocks.{b}."
            for a, c in (("dwconv", "dw"), ("layernorm", "ln"), ("pwconv1", "fc1"), ("pwconv2", "fc2")):
                p[dst + c + ".weight"] = sd[src + a + ".weight"]
                p[dst + c + ".bias"] = sd[src + a + ".bias"]
            p[dst + "gamma"] = sd[src + "layer_scale_parameter"]
    return p


def timm_convnext_to_canonical(sd: Dict[str, torch.Tensor], depths) -> Dict[str, torch.Tensor]:
    """timm ConvNeXt (open_clip ``visual.trunk``): stem.0 conv / stem.1 LayerNorm2d; stages.{s}.downsample.{0,1};
    stages.{s}.blocks.{b}.{conv_dw, norm, mlp.fc1, mlp.fc2, gamma}."""
    sd = _strip(sd)
    p = {"stem.conv.weight": sd["stem.0.weight"], "stem.conv.bias": sd["stem.0.bias"],
         "stem.ln.weight": sd["stem.1.weight"], "stem.ln.bias": sd["stem.1.bias"]}
    for s, depth in enumerate(depths):
        if s > 0:
            p[f"stages.{s}.down.ln.weight"] = sd[f"stages.{s}.downsample.0.weight"]
            p[f"stages.{s}.down.ln.bias"] = sd[f"stages.{s}.downsample.0.bias"]
            p[f"stages.{s}.down.conv.weight"] = sd[f"stages.{s}.downsample.1.weight"]
            p[f"stages.{s}.down.conv.bias"] = sd[f"stages.{s}.downsample.1.bias"]
        for b in range(depth):
            src, dst = f"stages.{s}.blocks.{b}.", f"stages.{s}.blocks.{b}."
            for a, c in (("conv_dw", "dw"), ("norm", "ln"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
                p[dst + c + ".weight"] = sd[src + a + ".weight"]
                p[dst + c + ".bias"] = sd[src + a + ".bias"]
            p[dst + "gamma"] = sd[src + "gamma"]
    return p


# ------------------------------------------------------------------------------------------------------------------
# canonical -> reference key names: an UNFROZEN tower is a registered sub-module in the reference
# (cambrian_arch.py:125-126), so its weights travel in the model's checkpoints under the third-party module's own keys
# (``model.vision_tower_aux_list.{i}.vision_tower.<keys below>``).  These are the exact inverses of the maps above for
# the four module types the reference instantiates (HF CLIPVisionModel, timm ViT trunk, HF Dinov2Model, timm ConvNeXt).
# ------------------------------------------------------------------------------------------------------------------
_CLIP_PAIRS = (("layer_norm1", "ln1"), ("layer_norm2", "ln2"), ("self_attn.q_proj", "q"), ("self_attn.k_proj", "k"),
               ("self_attn.v_proj", "v"), ("self_attn.out_proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2"))


def canonical_to_hf_clip(p: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    sd = {"vision_model.embeddings.patch_embedding.weight": p["patch.weight"],
          "vision_model.embeddings.class_embedding": p["cls"],
          "vision_model.embeddings.position_embedding.weight": p["pos"],
          "vision_model.pre_layrnorm.weight": p["pre_ln.weight"], "vision_model.pre_layrnorm.bias": p["pre_ln.bias"]}
    for l in range(num_layers):
        s, d = f"vision_model.encoder.layers.{l}.", f"layers.{l}."
        for a, b in _CLIP_PAIRS:
            sd[s + a + ".weight"] = p[d + b + ".weight"]
            sd[s + a + ".bias"] = p[d + b + ".bias"]
    return sd


def canonical_to_timm_vit(p: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    sd = {"patch_embed.proj.weight": p["patch.weight"], "patch_embed.proj.bias": p["patch.bias"],
          "pos_embed": p["pos"][None], "norm.weight": p["final_ln.weight"], "norm.bias": p["final_ln.bias"]}
    for l in range(num_layers):
        s, d = f"blocks.{l}.", f"layers.{l}."
        sd[s + "attn.qkv.weight"] = torch.cat([p[d + n + ".weight"] for n in "qkv"], 0)
        sd[s + "attn.qkv.bias"] = torch.cat([p[d + n + ".bias"] for n in "qkv"], 0)
        for a, b in (("norm1", "ln1"), ("norm2", "ln2"), ("attn.proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
            sd[s + a + ".weight"] = p[d + b + ".weight"]
            sd[s + a + ".bias"] = p[d + b + ".bias"]
    return sd


def canonical_to_hf_dinov2(p: Dict[str, torch.Tensor], num_layers: int, swiglu: bool) -> Dict[str, torch.Tensor]:
    sd = {"embeddings.patch_embeddings.projection.weight": p["patch.weight"],
          "embeddings.patch_embeddings.projection.bias": p["patch.bias"],
          "embeddings.cls_token": p["cls"].reshape(1, 1, -1), "embeddings.position_embeddings": p["pos"][None],
          "layernorm.weight": p["final_ln.weight"], "layernorm.bias": p["final_ln.bias"]}
    for l in range(num_layers):
        s, d = f"encoder.layer.{l}.", f"layers.{l}."
        pairs = [("norm1", "ln1"), ("norm2", "ln2"), ("attention.attention.query", "q"), ("attention.attention.key", "k"),
                 ("attention.attention.value", "v"), ("attention.output.dense", "proj")]
        pairs += [("mlp.weights_in", "fc1"), ("mlp.weights_out", "fc2")] if swiglu else [("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")]
        for a, b in pairs:
            sd[s + a + ".weight"] = p[d + b + ".weight"]
            sd[s + a + ".bias"] = p[d + b + ".bias"]
        sd[s + "layer_scale1.lambda1"] = p[d + "ls1"]
        sd[s + "layer_scale2.lambda1"] = p[d + "ls2"]
    return sd


def canonical_to_timm_convnext(p: Dict[str, torch.Tensor], depths) -> Dict[str, torch.Tensor]:
    sd = {"stem.0.weight": p["stem.conv.weight"], "stem.0.bias": p["stem.conv.bias"],
          "stem.1.weight": p["stem.ln.weight"], "stem.1.bias": p["stem.ln.bias"]}
    for s, depth in enumerate(depths):
        if s > 0:
            sd[f"stages.{s}.downsample.0.weight"] = p[f"stages.{s}.down.ln.weight"]
            sd[f"stages.{s}.downsample.0.bias"] = p[f"stages.{s}.down.ln.bias"]
            sd[f"stages.{s}.downsample.1.weight"] = p[f"stages.{s}.down.conv.weight"]
            sd[f"stages.{s}.downsample.1.bias"] = p[f"stages.{s}.down.conv.bias"]
        for b in range(depth):
            src = f"stages.{s}.blocks.{b}."
            for a, c in (("conv_dw", "dw"), ("norm", "ln"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
                sd[src + a + ".weight"] = p[src + c + ".weight"]
                sd[src + a + ".bias"] = p[src + c + ".bias"]
            sd[src + "gamma"] = p[src + "gamma"]
    return sd


class ReferenceKeys:
    """state_dict()/load_state_dict() of a trainable trunk in the reference module's key names.  ``to_ref`` /
    ``from_ref`` are a pair of the maps in this file.  Tensors of the reference module that the hot path never reads
    (CLIP's post_layernorm, the timm SigLIP trunk's attention-pool head, ...) are kept verbatim in ``extras`` when a
    checkpoint is loaded and written back on save, so a load -> save round trip loses nothing."""

    def __init__(self, to_ref, from_ref):
        self.to_ref, self.from_ref = to_ref, from_ref
        self.extras: Dict[str, torch.Tensor] = {}

    def install(self, trunk) -> None:
        trunk._ref_keys = self
        trunk._register_state_dict_hook(self._save_hook)
        trunk._register_load_state_dict_pre_hook(self._load_hook, with_module=True)

    @staticmethod
    def _save_hook(module, state_dict, prefix, local_metadata):
        me = module._ref_keys
        own = {k[len(prefix) + 2:].replace("__", "."): state_dict.pop(k) for k in list(state_dict)
               if k.startswith(prefix + "p.")}
        for k, v in me.to_ref(own).items():
            state_dict[prefix + k] = v
        for k, v in me.extras.items():
            state_dict[prefix + k] = v
        return state_dict

    @staticmethod
    def _load_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        me = module._ref_keys
        mine = {k[len(prefix):]: state_dict.pop(k) for k in list(state_dict) if k.startswith(prefix)}
        if not mine or any(k.startswith("p.") for k in mine):       # already in native naming
            for k, v in mine.items():
                state_dict[prefix + k] = v
            return
        canon = me.from_ref(mine)
        produced = me.to_ref(canon)
        me.extras = {k: v for k, v in mine.items() if k not in produced}
        for k, v in canon.items():
            state_dict[prefix + "p." + k.replace(".", "__")] = v
