"""Trainable ConvNeXt trunk — ``--unfreeze_mm_vision_tower`` for the CLIP-ConvNeXt tower (SURVEY.md §8f N4;
``clip_convnext_encoder.py:121-144`` under ``torch.set_grad_enabled(self.unfreeze_mm_vision_tower)``).

Same split as vit_train.py: fp32 master ``nn.Parameter``s under the canonical names of convnext.py, forward built
from operators that carry a backward:

  stem 4x4/4, downsample 2x2/2 : patch gather (``cmb_patchify_*``; the 2x2 gather is a permutation, its backward the
                                  inverse permutation) + ``ops.linear``
  depthwise 7x7                 : ``DwConv7x7Fn`` — forward ``cmb_dwconv7x7_nhwc``; dX = the same kernel on dY with the
                                  taps reversed; dW = ``cmb_dwconv7x7_wgrad`` (LDS-tiled, per-slot partials, column-summed);
                                  d(bias) = column sum of dY
  LayerNorm, fc1+GELU, fc2      : ``ops.layernorm`` / ``ops.linear``
  LayerScale gamma, residual, multi-stage bilinear resize + concat (clip_convnext_encoder.py:99-119,143): torch ops.
bf16 compute only.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ... import lib as L
from ... import ops
from . import vit_ops
from .convnext import ConvNeXtConfig

WGRAD_SLOTS = 128


class DwConv7x7Fn(torch.autograd.Function):
    """x NHWC [B,H,W,C], w49 fp32 [49,C] (tap-major), bias fp32 [C]."""

    @staticmethod
    def forward(ctx, x, w49, bias):
        x = x if x.is_contiguous() else x.contiguous()
        w49 = w49.contiguous()
        y = vit_ops.k_dwconv7x7(x, w49, bias.contiguous())
        ctx.save_for_backward(x, w49)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w49 = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = vit_ops.k_dwconv7x7(dy, w49.flip(0).contiguous(), torch.zeros(C, dtype=torch.float32, device=x.device))
        if ctx.needs_input_grad[1]:
            if C % 64 != 0:
                raise L.CambrianAmdError("depthwise weight gradient needs C % 64 == 0")
            slots = min(WGRAD_SLOTS, B * ((H + 7) // 8) * ((W + 7) // 8))
            part = torch.empty((slots, 49 * C), dtype=torch.float32, device=x.device)
            rc = L.load().cmb_dwconv7x7_wgrad(L.dtype_code(x.dtype), x.data_ptr(), dy.data_ptr(), B, H, W, C,
                                              part.data_ptr(), slots, L.stream_ptr(x.device))
            L.check(rc, "cmb_dwconv7x7_wgrad")
            dw = ops.k_colsum(part).view(49, C)
        if ctx.needs_input_grad[2]:
            db = ops.k_colsum(dy.view(-1, C))
        return dx, dw, db


class Patchify2x2Fn(torch.autograd.Function):
    """NHWC [B,H,W,C] -> [B*(H/2)*(W/2), 4C], column order (dy, dx, c); the backward is the inverse permutation."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return vit_ops.k_patchify2x2(x if x.is_contiguous() else x.contiguous())

    @staticmethod
    def backward(ctx, g):
        B, H, W, C = ctx.shape
        return g.view(B, H // 2, W // 2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def _key(name: str) -> str:
    return name.replace(".", "__")


class TrainableConvNeXt(nn.Module):
    def __init__(self, cfg: ConvNeXtConfig, canonical: Dict[str, torch.Tensor], device, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if dtype != torch.bfloat16:
            raise L.CambrianAmdError("trainable towers compute in bf16 (fp32 masters)")
        self.cfg, self.compute_dtype = cfg, dtype
        self.recompute = False   # per-block activation re-computation (bench.py --tower-recompute)
        self.p = nn.ParameterDict({_key(k): nn.Parameter(v.detach().to(device=device, dtype=torch.float32).clone())
                                   for k, v in canonical.items()})

    def P(self, name: str) -> torch.Tensor:
        return self.p[_key(name)]

    def canonical_state(self) -> Dict[str, torch.Tensor]:
        return {k.replace("__", "."): v.detach() for k, v in self.p.items()}

    def _block(self, s: int, b: int, x: torch.Tensor, B: int, H: int, c: int) -> torch.Tensor:
        """One ConvNeXt block on NHWC rows [B*H*H, c] (timm ConvNeXtBlock: dw 7x7 -> LN -> fc1 + GELU -> fc2 (* gamma) + x)."""
        cfg, dt = self.cfg, self.compute_dtype
        g = lambda n: self.P(f"stages.{s}.blocks.{b}.{n}")  # noqa: E731
        y = DwConv7x7Fn.apply(x.view(B, H, H, c), g("dw.weight").reshape(c, 49).t(), g("dw.bias")).view(-1, c)
        yn = ops.layernorm(y, g("ln.weight"), g("ln.bias"), cfg.ln_eps)
        h = ops.linear(yn, g("fc1.weight"), g("fc1.bias"), act=L.ACT_GELU_ERF)
        if cfg.layer_scale:
            return x + ops.linear(h, g("fc2.weight"), g("fc2.bias")) * g("gamma").to(dt)
        return ops.linear(h, g("fc2.weight"), g("fc2.bias"), residual=x)

    def forward_stages(self, images: torch.Tensor) -> List[torch.Tensor]:
        cfg, dt = self.cfg, self.compute_dtype
        B, _, S, _ = images.shape
        img = images if images.dtype in (torch.float32, torch.bfloat16) else images.float()
        H = S // 4
        c0 = cfg.dims[0]
        K = cfg.num_channels * 16
        kpad = (K + 63) // 64 * 64
        cols = vit_ops.k_patchify(img, 4, kpad, dt)                                   # pixels need no gradient
        x = ops.linear(cols, F.pad(self.P("stem.conv.weight").reshape(c0, K), (0, kpad - K)), self.P("stem.conv.bias"))
        x = ops.layernorm(x, self.P("stem.ln.weight"), self.P("stem.ln.bias"), cfg.ln_eps)
        outs = []
        for s, (depth, c) in enumerate(zip(cfg.depths, cfg.dims)):
            if s > 0:
                cp = cfg.dims[s - 1]
                g = lambda n: self.P(f"stages.{s}.down.{n}")  # noqa: E731
                xn = ops.layernorm(x, g("ln.weight"), g("ln.bias"), cfg.ln_eps)
                cols = Patchify2x2Fn.apply(xn.view(B, H, H, cp))
                H //= 2
                wd = g("conv.weight").permute(0, 2, 3, 1).reshape(c, 4 * cp)          # (dy, dx, cin) column order
                x = ops.linear(cols, wd, g("conv.bias"))
            for b in range(depth):
                if self.recompute and torch.is_grad_enabled():   # per-block activation re-computation (flag-controlled)
                    x = checkpoint(self._block, s, b, x, B, H, c, use_reentrant=False)
                else:
                    x = self._block(s, b, x, B, H, c)
            outs.append(x.view(B, H, H, c))
        return outs

    def forward(self, images: torch.Tensor, out_side: Optional[int], multi_stage: bool = True) -> torch.Tensor:
        """clip_convnext_encoder.py:99-144: every kept stage map bilinearly resized (fp32, align_corners=False) to
        out_side x out_side, flattened, channel-concatenated."""
        stages = self.forward_stages(images)
        if not multi_stage:
            stages = stages[-1:]
        B = images.shape[0]
        if out_side is None:
            if len(stages) != 1:
                raise L.CambrianAmdError("multi-stage output needs a common output grid (interp size)")
            return stages[0].reshape(B, -1, stages[0].shape[-1])
        outs = []
        for s in stages:
            y = s.permute(0, 3, 1, 2)
            y = F.interpolate(y.float(), size=(out_side, out_side), mode="bilinear", align_corners=False).to(s.dtype)
            outs.append(y.permute(0, 2, 3, 1).flatten(1, 2))
        return torch.cat(outs, -1)
