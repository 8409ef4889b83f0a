"""Trainable ViT trunk — the ``--unfreeze_mm_vision_tower`` mode of the reference (SURVEY.md §8f N4;
``train_fsdp.py:127,1691-1695``, ``cambrian_arch.py:125-126``, towers run under
``torch.set_grad_enabled(self.unfreeze_mm_vision_tower)``: ``clip_encoder.py:103``, ``siglip_encoder.py:96``,
``dino_encoder.py:158``).

The frozen trunk (vit.py) packs its weights once into padded / fused buffers and drives raw kernels with no autograd
graph.  This one keeps every canonical tensor as an fp32 master ``nn.Parameter`` and builds the same arithmetic from
the autograd-aware operators the SVA side already trains with:

  patch embed / QKV / proj / fc1 / fc2 : ``ops.linear``  (HIP GEMM forward, dX and split-K dW GEMMs, fused bias /
                                          activation / residual; activation backward ``cmb_act_bwd``)
  LayerNorm                             : ``ops.layernorm`` (HIP forward + backward)
  self-attention                        : ``ops.vit_attention`` — the decoder's flash forward / dQ / dK+dV kernels in
                                          their bidirectional form (tokens and head_dim zero-padded to 128-multiples)
  SwiGLU (DINOv2-g)                     : ``ops.swiglu`` (HIP forward + backward)
  CLS / position rows, LayerScale, token-grid resize : plain torch ops on the GPU (small, pointwise / tiny)

Weights that need padding for the GEMM's K-step (14x14x3 patches -> 640 columns, SigLIP's 4304-wide MLP -> 4352) are
padded on the fly from the masters, so gradients land on the un-padded parameters.  bf16 compute only.
Parameter names are the canonical ones of vit.py with '.' -> '__' (``layers__3__fc1__weight``);
``canonical_state()`` / ``reference_state()`` give them back under canonical / reference (HF, timm) key names.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ... import lib as L
from ... import ops
from . import vit_ops
from .vit import ViTConfig

_ACT = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF, "gelu_tanh": L.ACT_GELU_TANH, "swiglu": L.ACT_NONE}


def _key(name: str) -> str:
    return name.replace(".", "__")


def resample_tokens_autograd(x: torch.Tensor, target_tokens: int) -> torch.Tensor:
    """clip_encoder.py:70-96 / siglip_encoder.py:67-93 / dino_encoder.py:128-154 verbatim in torch (fp32 bilinear,
    align_corners=False) so that it back-propagates; the frozen path uses the HIP resample kernel instead."""
    B, T, C = x.shape
    if T == target_tokens:
        return x
    hi, ho = int(T ** 0.5), int(target_tokens ** 0.5)
    y = x.view(B, hi, hi, C).permute(0, 3, 1, 2).contiguous()
    y = F.interpolate(y.to(torch.float32), size=(ho, ho), mode="bilinear", align_corners=False).to(x.dtype)
    return y.permute(0, 2, 3, 1).contiguous().flatten(1, 2)


class TrainableViT(nn.Module):
    def __init__(self, cfg: ViTConfig, canonical: Dict[str, torch.Tensor], device, dtype: torch.dtype = torch.bfloat16,
                 pos_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
        """``pos_fn``: applied to the position-embedding master in every forward (DINOv2's bicubic 37x37 -> 27x27
        interpolation, dino_encoder.py / HF Dinov2Embeddings.interpolate_pos_encoding) so its gradient reaches the
        native-resolution parameter."""
        super().__init__()
        self.pos_fn = pos_fn
        self.recompute = False   # per-block activation re-computation (set by the tower wrapper / bench.py --tower-recompute)
        if dtype != torch.bfloat16:
            raise L.CambrianAmdError("trainable towers compute in bf16 (fp32 masters)")
        self.cfg = cfg
        self.compute_dtype = dtype
        self.nl = cfg.run_layers if cfg.run_layers is not None else cfg.num_layers
        # every tensor of the checkpoint stays a parameter — also the layers behind select_layer that never run
        # (clip_encoder.py:66): they receive no gradient, exactly as in the reference, and survive a save / load
        self.p = nn.ParameterDict({_key(name): nn.Parameter(t.detach().to(device=device, dtype=torch.float32).clone())
                                   for name, t in canonical.items()})

    # ---- state under other naming schemes -----------------------------------------------------------------------
    def canonical_state(self) -> Dict[str, torch.Tensor]:
        return {k.replace("__", "."): v.detach() for k, v in self.p.items()}

    def reference_state(self, to_reference: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]]):
        """Keys as the reference checkpoint stores an unfrozen tower (weight_maps.canonical_to_*)."""
        return to_reference(self.canonical_state())

    def P(self, name: str) -> torch.Tensor:
        return self.p[_key(name)]

    def has(self, name: str) -> bool:
        return _key(name) in self.p

    def _block(self, l: int, x: torch.Tensor, B: int, N: int) -> torch.Tensor:
        """One transformer block on token-major rows [B*N, D] (HF CLIPEncoderLayer / Dinov2Layer / timm Block)."""
        cfg, dt = self.cfg, self.compute_dtype
        D, H, hd = cfg.hidden_size, cfg.num_heads, cfg.head_dim
        act = _ACT[cfg.act]
        scale = 1.0 / math.sqrt(hd)
        Fm = cfg.mlp_dim
        Fp = (Fm + 63) // 64 * 64
        g = lambda n: self.P(f"layers.{l}.{n}")  # noqa: E731
        h = ops.layernorm(x, g("ln1.weight"), g("ln1.bias"), cfg.ln_eps)
        wqkv = torch.cat([g("q.weight"), g("k.weight"), g("v.weight")], 0)
        bqkv = torch.cat([g("q.bias"), g("k.bias"), g("v.bias")], 0)
        qkv = ops.linear(h, wqkv, bqkv).view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)     # [3,B,H,N,hd] views
        a = ops.vit_attention(qkv[0], qkv[1], qkv[2], scale)                            # [B,H,N,hd]
        a = a.transpose(1, 2).reshape(B * N, D)
        if cfg.layerscale:
            x = x + ops.linear(a, g("proj.weight"), g("proj.bias")) * g("ls1").to(dt)
        else:
            x = ops.linear(a, g("proj.weight"), g("proj.bias"), residual=x)
        h = ops.layernorm(x, g("ln2.weight"), g("ln2.bias"), cfg.ln_eps)
        if cfg.act == "swiglu":                   # Dinov2SwiGLUFFN: silu(x1) * x2, x1 | x2 = weights_in(x).chunk(2)
            f = ops.linear(h, g("fc1.weight"), g("fc1.bias"))
            f = ops.swiglu(f[:, :Fm], f[:, Fm:])
            w2 = g("fc2.weight")
        else:
            w1, b1, w2 = g("fc1.weight"), g("fc1.bias"), g("fc2.weight")
            if Fp != Fm:                           # zero rows / columns: act(0) = 0 for every activation used here
                w1, b1, w2 = F.pad(w1, (0, 0, 0, Fp - Fm)), F.pad(b1, (0, Fp - Fm)), F.pad(w2, (0, Fp - Fm))
            f = ops.linear(h, w1, b1, act=act)
        if cfg.layerscale:
            x = x + ops.linear(f, w2, g("fc2.bias")) * g("ls2").to(dt)
        else:
            x = ops.linear(f, w2, g("fc2.bias"), residual=x)
        return x

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        cfg, dt = self.cfg, self.compute_dtype
        B = images.shape[0]
        T, D, H, hd = cfg.num_patches, cfg.hidden_size, cfg.num_heads, cfg.head_dim
        K = cfg.num_channels * cfg.patch_size ** 2
        kpad = (K + 63) // 64 * 64
        img = images if images.dtype in (torch.float32, torch.bfloat16) else images.float()
        cols = vit_ops.k_patchify(img, cfg.patch_size, kpad, dt)                       # [B*T, kpad]; pixels need no grad
        w = F.pad(self.P("patch.weight").reshape(D, K), (0, kpad - K))
        x = ops.linear(cols, w, self.P("patch.bias") if cfg.patch_bias else None).view(B, T, D)
        pos = self.P("pos")
        pos = (self.pos_fn(pos) if self.pos_fn is not None else pos).to(dt)
        if cfg.has_cls:
            cls = (self.P("cls").to(dt) + pos[0]).view(1, 1, D).expand(B, 1, D)
            x = torch.cat([cls, x + pos[1:]], 1)
        else:
            x = x + pos
        N = x.shape[1]
        x = x.reshape(B * N, D)
        if cfg.pre_ln:
            x = ops.layernorm(x, self.P("pre_ln.weight"), self.P("pre_ln.bias"), cfg.ln_eps)
        for l in range(self.nl):
            if self.recompute and torch.is_grad_enabled():
                # per-block activation re-computation (flag-controlled: `recompute`): a block keeps only its input
                x = checkpoint(self._block, l, x, B, N, use_reentrant=False)
            else:
                x = self._block(l, x, B, N)
        if cfg.final_ln:
            x = ops.layernorm(x, self.P("final_ln.weight"), self.P("final_ln.bias"), cfg.ln_eps)
        x = x.view(B, N, D)
        return x[:, 1:] if cfg.has_cls else x


def tower_param_groups(model: nn.Module, base_lr: float, tower_lr: Optional[float], weight_decay: float = 0.0):
    """Optimizer groups of cambrian_trainer.py:319-348: parameters under ``vision_tower_aux_list`` get
    ``mm_vision_tower_lr`` when it is set, everything else the base learning rate."""
    tower, rest = [], []
    for n, p_ in model.named_parameters():
        if p_.requires_grad:
            (tower if "vision_tower_aux_list" in n else rest).append(p_)
    groups = [{"params": rest, "lr": base_lr, "weight_decay": weight_decay}]
    if tower:
        groups.append({"params": tower, "lr": tower_lr if tower_lr is not None else base_lr, "weight_decay": weight_decay})
    return groups
