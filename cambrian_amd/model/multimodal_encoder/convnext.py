"""Native ConvNeXt trunk (CLIP ConvNeXt-XXL / -L towers), forward only, channels-last.

The reference runs timm's ``ConvNeXt.stem`` / ``.stages`` obtained through open_clip
(clip_convnext_encoder.py:84-90,133-136).  Re-stated on the HIP kernels with every activation kept NHWC
([B*H*W, C] rows), so that all 1x1 work is a plain MFMA GEMM and the multi-stage output lands directly in
the channel slices of one [B, 9216, 5760] buffer:

    stem        : 4x4/4 conv = patch gather + GEMM(+bias) -> LayerNorm(C)
    block       : depthwise 7x7 (+bias) -> LayerNorm(C) -> GEMM C->4C (+bias, GELU) -> GEMM 4C->C (+bias,
                  layer-scale gamma, +residual)                      [timm ConvNeXtBlock, conv_mlp=False]
    downsample  : LayerNorm(C) -> 2x2/2 conv = NHWC patch gather + GEMM(+bias)
    multi-stage : each stage map bilinearly resampled to side x side (fp32 lerp) and channel-concatenated
                  (clip_convnext_encoder.py:99-119,137-143).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from ... import lib as L
from ... import ops
from . import vit_ops


@dataclass
class ConvNeXtConfig:
    depths: Sequence[int] = (3, 4, 30, 3)
    dims: Sequence[int] = (384, 768, 1536, 3072)
    ln_eps: float = 1e-5          # timm convnext_xxlarge: norm_eps=1e-5 (HF ConvNextModel hard-codes 1e-6)
    num_channels: int = 3
    layer_scale: bool = True


class ConvNeXtTrunk(nn.Module):
    def __init__(self, cfg: ConvNeXtConfig, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.cfg = cfg
        self.compute_dtype = dtype
        self._packed = False

    # canonical names:
    #   stem.conv.weight [C0,3,4,4] .bias, stem.ln.weight/.bias
    #   stages.{s}.down.ln.weight/.bias, stages.{s}.down.conv.weight [Cs,Cs-1,2,2] .bias      (s >= 1)
    #   stages.{s}.blocks.{b}.dw.weight [C,1,7,7] .bias, .ln.weight/.bias, .fc1.weight [4C,C] .bias,
    #                          .fc2.weight [C,4C] .bias, .gamma [C]
    @staticmethod
    def random_canonical(cfg: ConvNeXtConfig, gen: torch.Generator) -> Dict[str, torch.Tensor]:
        def rn(*s, std=0.02):
            return torch.randn(*s, generator=gen, device=gen.device) * std

        p: Dict[str, torch.Tensor] = {}
        c0 = cfg.dims[0]
        p["stem.conv.weight"] = rn(c0, cfg.num_channels, 4, 4, std=0.1)
        p["stem.conv.bias"] = rn(c0)
        p["stem.ln.weight"] = 1 + rn(c0, std=0.1)
        p["stem.ln.bias"] = rn(c0, std=0.1)
        for s, (depth, c) in enumerate(zip(cfg.depths, cfg.dims)):
            if s > 0:
                cp = cfg.dims[s - 1]
                p[f"stages.{s}.down.ln.weight"] = 1 + rn(cp, std=0.1)
                p[f"stages.{s}.down.ln.bias"] = rn(cp, std=0.1)
                p[f"stages.{s}.down.conv.weight"] = rn(c, cp, 2, 2, std=1.0 / math.sqrt(4 * cp))
                p[f"stages.{s}.down.conv.bias"] = rn(c)
            for b in range(depth):
                pre = f"stages.{s}.blocks.{b}."
                p[pre + "dw.weight"] = rn(c, 1, 7, 7, std=0.1)
                p[pre + "dw.bias"] = rn(c)
                p[pre + "ln.weight"] = 1 + rn(c, std=0.1)
                p[pre + "ln.bias"] = rn(c, std=0.1)
                p[pre + "fc1.weight"] = rn(4 * c, c, std=1.0 / math.sqrt(c))
                p[pre + "fc1.bias"] = rn(4 * c)
                p[pre + "fc2.weight"] = rn(c, 4 * c, std=1.0 / math.sqrt(4 * c))
                p[pre + "fc2.bias"] = rn(c)
                if cfg.layer_scale:
                    p[pre + "gamma"] = 0.5 + rn(c, std=0.1)
        return p

    def load_canonical(self, p: Dict[str, torch.Tensor], device) -> "ConvNeXtTrunk":
        cfg, dt = self.cfg, self.compute_dtype
        ks = 64 if dt == torch.bfloat16 else 32

        def buf(name, t, dtype):
            self.register_buffer(name, t.to(dtype).contiguous().to(device), persistent=False)

        self._ln_fused = fuse = bool(vit_ops.LN_FUSE and dt == torch.bfloat16)   # block ln -> fc1 folded (vit_ops.fold_ln_into_linear)
        c0 = cfg.dims[0]
        K = cfg.num_channels * 16
        self.kpad = (K + ks - 1) // ks * ks
        w = torch.zeros(c0, self.kpad, device=p["stem.conv.weight"].device)
        w[:, :K] = p["stem.conv.weight"].float().reshape(c0, K)
        buf("stem_w", w, dt)
        buf("stem_b", p["stem.conv.bias"], torch.float32)
        buf("stem_ln_w", p["stem.ln.weight"], torch.float32)
        buf("stem_ln_b", p["stem.ln.bias"], torch.float32)
        for s, (depth, c) in enumerate(zip(cfg.depths, cfg.dims)):
            if s > 0:
                cp = cfg.dims[s - 1]
                buf(f"s{s}_down_ln_w", p[f"stages.{s}.down.ln.weight"], torch.float32)
                buf(f"s{s}_down_ln_b", p[f"stages.{s}.down.ln.bias"], torch.float32)
                # [Cout, Cin, 2, 2] -> [Cout, (dy, dx, cin)] to match the NHWC 2x2 patch gather
                wd = p[f"stages.{s}.down.conv.weight"].float().permute(0, 2, 3, 1).reshape(c, 4 * cp)
                buf(f"s{s}_down_w", wd, dt)
                buf(f"s{s}_down_b", p[f"stages.{s}.down.conv.bias"], torch.float32)
            for b in range(depth):
                pre = f"stages.{s}.blocks.{b}."
                n = f"s{s}_b{b}_"
                buf(n + "dw_w", p[pre + "dw.weight"].float().reshape(c, 49).T, torch.float32)  # [49, C]
                buf(n + "dw_b", p[pre + "dw.bias"], torch.float32)
                buf(n + "ln_w", p[pre + "ln.weight"], torch.float32)
                buf(n + "ln_b", p[pre + "ln.bias"], torch.float32)
                if fuse:
                    w2, cs, b2 = vit_ops.fold_ln_into_linear(p[pre + "fc1.weight"], p[pre + "fc1.bias"], p[pre + "ln.weight"],
                                                             p[pre + "ln.bias"], dt)
                    buf(n + "fc1_w", w2, dt)
                    buf(n + "fc1_b", b2, torch.float32)
                    buf(n + "fc1_cs", cs, torch.float32)
                else:
                    buf(n + "fc1_w", p[pre + "fc1.weight"], dt)
                    buf(n + "fc1_b", p[pre + "fc1.bias"], torch.float32)
                buf(n + "fc2_w", p[pre + "fc2.weight"], dt)
                buf(n + "fc2_b", p[pre + "fc2.bias"], torch.float32)
                if cfg.layer_scale:
                    buf(n + "gamma", p[pre + "gamma"], torch.float32)
        self._packed = True
        return self

    @torch.no_grad()
    def forward_stages(self, images: torch.Tensor) -> List[torch.Tensor]:
        """images [B,3,S,S] -> list of the 4 stage maps, each NHWC [B, H_s, W_s, C_s]."""
        if not self._packed:
            raise L.CambrianAmdError("ConvNeXtTrunk weights are not loaded")
        cfg, dt = self.cfg, self.compute_dtype
        B, _, S, _ = images.shape
        img = images if images.dtype in (torch.float32, torch.bfloat16) else images.float()
        H = S // 4
        cols = vit_ops.k_patchify(img, 4, self.kpad, dt)
        x = ops.k_gemm(cols, self.stem_w, bias=self.stem_b)
        x, _, _ = ops.k_layernorm_fwd(x, self.stem_ln_w, self.stem_ln_b, cfg.ln_eps, want_stats=False)
        outs = []
        for s, (depth, c) in enumerate(zip(cfg.depths, cfg.dims)):
            if s > 0:
                cp = cfg.dims[s - 1]
                g = lambda n: getattr(self, f"s{s}_down_{n}")  # noqa: E731
                xn, _, _ = ops.k_layernorm_fwd(x, g("ln_w"), g("ln_b"), cfg.ln_eps, want_stats=False)
                cols = vit_ops.k_patchify2x2(xn.view(B, H, H, cp))
                H //= 2
                x = ops.k_gemm(cols, g("w"), bias=g("b"))
            for b in range(depth):
                g = lambda n: getattr(self, f"s{s}_b{b}_{n}")  # noqa: E731
                y = vit_ops.k_dwconv7x7(x.view(B, H, H, c), g("dw_w"), g("dw_b")).view(-1, c)
                if self._ln_fused:   # row statistics + the LayerNorm inside fc1's epilogue: no normalised copy of the map
                    h = ops.k_gemm(y, g("fc1_w"), bias=g("fc1_b"), act=L.ACT_GELU_ERF, row_stats=ops.k_row_stats(y, cfg.ln_eps),
                                   col_sum=g("fc1_cs"))
                else:
                    yn, _, _ = ops.k_layernorm_fwd(y, g("ln_w"), g("ln_b"), cfg.ln_eps, want_stats=False)
                    h = ops.k_gemm(yn, g("fc1_w"), bias=g("fc1_b"), act=L.ACT_GELU_ERF)
                x = ops.k_gemm(h, g("fc2_w"), bias=g("fc2_b"), colscale=g("gamma") if cfg.layer_scale else None,
                               residual=x)
            outs.append(x.view(B, H, H, c))
        return outs

    @torch.no_grad()
    def forward(self, images: torch.Tensor, out_side: Optional[int], multi_stage: bool = True) -> torch.Tensor:
        """-> [B, out_side^2, sum(dims)] (multi-stage) or [B, out_side^2, dims[-1]]; out_side None keeps the
        last stage's own grid (clip_convnext_encoder.py:99-119: no resize when _interp_size is None)."""
        stages = self.forward_stages(images)
        if not multi_stage:
            stages = stages[-1:]
        B = images.shape[0]
        if out_side is None:
            if len(stages) != 1:
                raise L.CambrianAmdError("multi-stage output needs a common output grid (interp size)")
            s = stages[0]
            return s.view(B, -1, s.shape[-1])
        ctot = sum(s.shape[-1] for s in stages)
        out = torch.empty((B, out_side * out_side, ctot), dtype=self.compute_dtype, device=images.device)
        off = 0
        for s in stages:
            _, Hs, Ws, C = s.shape
            vit_ops.k_resample(s.view(B, Hs * Ws, C), Hs, Ws, out, out_side, out_side, col_offset=off)
            off += C
        return out
