"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the vision towers' arithmetic.  The reference holds only thin wrappers
(clip_encoder.py:57-107, siglip_encoder.py:67-99, dino_encoder.py:115-165, clip_convnext_encoder.py:99-144);
the math itself is in third-party packages that are NOT under /root/reference:
  * transformers==4.37.0  (pyproject.toml:17): CLIPVisionModel (clip_encoder.py:47,104), Dinov2Model (dino_encoder.py:81,159)
  * timm==0.9.16 (pyproject.toml:21) + open_clip_torch (unpinned, :22): VisionTransformer.forward_features
    (siglip_encoder.py:53-56,97), ConvNeXt.stem/.stages (clip_convnext_encoder.py:84-90,133-136)
Their published algorithms are restated below on the canonical parameter names of
cambrian_amd/model/multimodal_encoder/{vit,convnext}.py.  Pinning: tests/golden/towers_*.pt hold outputs of
the installed HF modules (transformers 5.15: CLIPVisionModel, Dinov2Model, SiglipVisionModel, ConvNextModel —
the same op graphs) on seeded weights; tests/test_oracle_golden.py replays them.  timm itself is not
installable here, so the timm-specific choices (SO400M activation = nn.GELU erf under timm 0.9.16, ConvNeXt-XXL
LayerNorm eps 1e-5) are "parity unpinned" at that boundary and are config switches.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _act(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)          # HF QuickGELUActivation (CLIP)
    if name == "gelu":
        return F.gelu(x)
    if name == "gelu_tanh":
        return F.gelu(x, approximate="tanh")          # HF gelu_pytorch_tanh (SigLIP)
    raise ValueError(name)


def vit_forward(cfg, p: Dict[str, torch.Tensor], images: torch.Tensor) -> torch.Tensor:
    """cfg: ViTConfig.  Returns [B, num_patches, D]: CLS dropped; CLIP = hidden_states[run_layers]
    (no post-LN), SigLIP/DINOv2 = after the final LayerNorm."""
    B = images.shape[0]
    D, H, hd = cfg.hidden_size, cfg.num_heads, cfg.head_dim
    x = F.conv2d(images.float(), p["patch.weight"].float(), p.get("patch.bias"), stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)                                  # [B, T, D]
    if cfg.has_cls:
        x = torch.cat([p["cls"].float().view(1, 1, D).expand(B, -1, -1), x], 1)
    x = x + p["pos"].float()[None]
    if cfg.pre_ln:
        x = _ln(x, p["pre_ln.weight"], p["pre_ln.bias"], cfg.ln_eps)
    nl = cfg.run_layers if cfg.run_layers is not None else cfg.num_layers
    for l in range(nl):
        pre = f"layers.{l}."
        h = _ln(x, p[pre + "ln1.weight"], p[pre + "ln1.bias"], cfg.ln_eps)
        q = (h @ p[pre + "q.weight"].T + p[pre + "q.bias"]).view(B, -1, H, hd).transpose(1, 2)
        k = (h @ p[pre + "k.weight"].T + p[pre + "k.bias"]).view(B, -1, H, hd).transpose(1, 2)
        v = (h @ p[pre + "v.weight"].T + p[pre + "v.bias"]).view(B, -1, H, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), -1) @ v
        a = a.transpose(1, 2).reshape(B, -1, D) @ p[pre + "proj.weight"].T + p[pre + "proj.bias"]
        if cfg.layerscale:
            a = a * p[pre + "ls1"]
        x = x + a
        h = _ln(x, p[pre + "ln2.weight"], p[pre + "ln2.bias"], cfg.ln_eps)
        f = h @ p[pre + "fc1.weight"].T + p[pre + "fc1.bias"]
        if cfg.act == "swiglu":                                       # HF Dinov2SwiGLUFFN
            x1, x2 = f.chunk(2, dim=-1)
            f = F.silu(x1) * x2
        else:
            f = _act(cfg.act, f)
        f = f @ p[pre + "fc2.weight"].T + p[pre + "fc2.bias"]
        if cfg.layerscale:
            f = f * p[pre + "ls2"]
        x = x + f
    if cfg.final_ln:
        x = _ln(x, p["final_ln.weight"], p["final_ln.bias"], cfg.ln_eps)
    return x[:, 1:] if cfg.has_cls else x


def interpolate_tokens(x: torch.Tensor, target_tokens: int) -> torch.Tensor:
    """The wrappers' token-grid resize (clip_encoder.py:70-96 and its copies): NHWC->NCHW, fp32 bilinear,
    align_corners=False, back to [B, target, C]."""
    b, n, dim = x.shape
    if n == target_tokens:
        return x
    h = w = int(n ** 0.5)
    t = int(target_tokens ** 0.5)
    y = x.view(b, h, w, dim).permute(0, 3, 1, 2).contiguous()
    y = F.interpolate(y.to(torch.float32), size=(t, t), mode="bilinear", align_corners=False).to(x.dtype)
    return y.permute(0, 2, 3, 1).contiguous().flatten(1, 2)


def convnext_stages(cfg, p: Dict[str, torch.Tensor], images: torch.Tensor) -> List[torch.Tensor]:
    """timm ConvNeXt stem + stages (NCHW, like the original); returns the 4 stage maps [B, C_s, H_s, W_s]."""
    def ln2d(x, w, b):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, cfg.ln_eps).permute(0, 3, 1, 2)

    x = F.conv2d(images.float(), p["stem.conv.weight"].float(), p["stem.conv.bias"], stride=4)
    x = ln2d(x, p["stem.ln.weight"], p["stem.ln.bias"])
    outs = []
    for s, (depth, c) in enumerate(zip(cfg.depths, cfg.dims)):
        if s > 0:
            x = ln2d(x, p[f"stages.{s}.down.ln.weight"], p[f"stages.{s}.down.ln.bias"])
            x = F.conv2d(x, p[f"stages.{s}.down.conv.weight"].float(), p[f"stages.{s}.down.conv.bias"], stride=2)
        for b in range(depth):
            pre = f"stages.{s}.blocks.{b}."
            y = F.conv2d(x, p[pre + "dw.weight"].float(), p[pre + "dw.bias"], padding=3, groups=c)
            y = y.permute(0, 2, 3, 1)
            y = F.layer_norm(y, (c,), p[pre + "ln.weight"], p[pre + "ln.bias"], cfg.ln_eps)
            y = F.gelu(y @ p[pre + "fc1.weight"].T + p[pre + "fc1.bias"]) @ p[pre + "fc2.weight"].T + p[pre + "fc2.bias"]
            if cfg.layer_scale:
                y = y * p[pre + "gamma"]
            x = x + y.permute(0, 3, 1, 2)
        outs.append(x)
    return outs


def convnext_forward(cfg, p, images, out_side: Optional[int], multi_stage: bool = True) -> torch.Tensor:
    """clip_convnext_encoder.py:121-144 + :99-119."""
    stages = convnext_stages(cfg, p, images)
    if not multi_stage:
        stages = stages[-1:]
    feats = []
    for s in stages:
        if out_side is not None:
            s = F.interpolate(s.float(), size=(out_side, out_side), mode="bilinear", align_corners=False).to(s.dtype)
        feats.append(s.flatten(2, 3).permute(0, 2, 1).contiguous())
    return torch.cat(feats, -1)
