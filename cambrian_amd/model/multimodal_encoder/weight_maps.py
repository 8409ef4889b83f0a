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
