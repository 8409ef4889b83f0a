"""Native ViT trunk for the three transformer towers of Cambrian-1, forward only.

The reference delegates the arithmetic to third-party modules — HF ``CLIPVisionModel`` (clip_encoder.py:47,104),
timm ``VisionTransformer.forward_features`` via open_clip (siglip_encoder.py:53-56,97) and HF ``Dinov2Model``
(dino_encoder.py:81,159).  This file re-states that published arithmetic on the HIP kernels:

    patch-embed conv (stride == kernel) = patch gather + MFMA GEMM, position embedding added in the GEMM
    epilogue (row-mapped residual), CLS row broadcast, optional pre-LN (CLIP), then per layer
    LN -> fused QKV GEMM(+bias) -> flash attention -> proj GEMM(+bias, LayerScale, +residual)
       -> LN -> fc1 GEMM(+bias, activation | SwiGLU) -> fc2 GEMM(+bias, LayerScale, +residual),
    optional final LN.

Weights are frozen (reference default, train_fsdp.py:1655-1659) and are packed once into the layouts the
kernels want: QKV fused to one [3*heads*hd_pad, D] matrix with head_dim zero-padded to a multiple of 32
(SigLIP-SO400M: 72 -> 96), MLP hidden zero-padded to a multiple of 64 (SO400M: 4304 -> 4352), patch kernel
flattened to [D, K_pad].  Zero padding does not change any result.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ... import lib as L
from ... import ops
from . import vit_ops


@dataclass
class ViTConfig:
    image_size: int
    patch_size: int
    hidden_size: int
    num_layers: int
    num_heads: int
    mlp_dim: int                  # hidden width of the MLP (SwiGLU: width of each half)
    act: str = "gelu"             # "quick_gelu" (CLIP) | "gelu" | "gelu_tanh" (SigLIP) | "swiglu" (DINOv2-g)
    ln_eps: float = 1e-5
    has_cls: bool = True
    pre_ln: bool = False          # CLIP pre_layrnorm
    final_ln: bool = True         # SigLIP norm / DINOv2 layernorm; CLIP hidden_states[-2] never sees post_layernorm
    layerscale: bool = False      # DINOv2
    patch_bias: bool = True       # CLIP's patch conv has no bias
    run_layers: Optional[int] = None  # layers actually needed (CLIP select_layer=-2 -> num_layers-1)
    num_channels: int = 3

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def head_dim_pad(self) -> int:
        return (self.head_dim + 31) // 32 * 32

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


def _pad_rows(w: torch.Tensor, rows: int) -> torch.Tensor:
    if w.shape[0] == rows:
        return w
    out = torch.zeros((rows,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[: w.shape[0]] = w
    return out


def _pad_cols(w: torch.Tensor, cols: int) -> torch.Tensor:
    if w.shape[1] == cols:
        return w
    out = torch.zeros((w.shape[0], cols), dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


class ViTTrunk(nn.Module):
    """Holds the packed, frozen weights as buffers and runs the forward on the HIP kernels.

    ``from_canonical`` takes a dict in a neutral naming (documented below) so that HF-CLIP, HF-DINOv2,
    HF-SigLIP and timm checkpoints can all be mapped onto it (``weight_maps.py``)."""

    def __init__(self, cfg: ViTConfig, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.cfg = cfg
        self.compute_dtype = dtype
        self._packed = False

    # ---- canonical (unpacked) parameter names ---------------------------------------------------
    # patch.weight [D,C,p,p]  patch.bias [D]?  cls [D]?  pos [T(+1), D]
    # pre_ln.weight/bias?     final_ln.weight/bias?
    # layers.{l}.ln1.weight/bias, q/k/v.weight [D,D] + .bias [D], proj.weight/bias, ls1 [D]?,
    # layers.{l}.ln2.weight/bias, fc1.weight [F or 2F, D] + bias, fc2.weight [D, F] + bias, ls2 [D]?
    @staticmethod
    def random_canonical(cfg: ViTConfig, gen: torch.Generator) -> Dict[str, torch.Tensor]:
        D, F = cfg.hidden_size, cfg.mlp_dim

        def rn(*s, std=0.02):
            return torch.randn(*s, generator=gen, device=gen.device) * std

        p: Dict[str, torch.Tensor] = {}
        p["patch.weight"] = rn(D, cfg.num_channels, cfg.patch_size, cfg.patch_size)
        if cfg.patch_bias:
            p["patch.bias"] = rn(D)
        T = cfg.num_patches + (1 if cfg.has_cls else 0)
        p["pos"] = rn(T, D)
        if cfg.has_cls:
            p["cls"] = rn(D)
        for name, on in (("pre_ln", cfg.pre_ln), ("final_ln", cfg.final_ln)):
            if on:
                p[f"{name}.weight"] = 1 + rn(D, std=0.1)
                p[f"{name}.bias"] = rn(D, std=0.1)
        for l in range(cfg.num_layers):
            pre = f"layers.{l}."
            for n in ("ln1", "ln2"):
                p[pre + n + ".weight"] = 1 + rn(D, std=0.1)
                p[pre + n + ".bias"] = rn(D, std=0.1)
            for n in ("q", "k", "v", "proj"):
                p[pre + n + ".weight"] = rn(D, D, std=1.0 / math.sqrt(D))
                p[pre + n + ".bias"] = rn(D)
            f_in = 2 * F if cfg.act == "swiglu" else F
            p[pre + "fc1.weight"] = rn(f_in, D, std=1.0 / math.sqrt(D))
            p[pre + "fc1.bias"] = rn(f_in)
            p[pre + "fc2.weight"] = rn(D, F, std=1.0 / math.sqrt(F))
            p[pre + "fc2.bias"] = rn(D)
            if cfg.layerscale:
                p[pre + "ls1"] = 0.5 + rn(D, std=0.1)
                p[pre + "ls2"] = 0.5 + rn(D, std=0.1)
        return p

    def load_canonical(self, p: Dict[str, torch.Tensor], device) -> "ViTTrunk":
        cfg, dt = self.cfg, self.compute_dtype
        D, H, hd, hdp = cfg.hidden_size, cfg.num_heads, cfg.head_dim, cfg.head_dim_pad
        ks = 64 if dt == torch.bfloat16 else 32
        K = cfg.num_channels * cfg.patch_size ** 2
        self.kpad = (K + ks - 1) // ks * ks
        Fp = (cfg.mlp_dim + 63) // 64 * 64
        self.fpad = Fp

        def buf(name, t, dtype):
            self.register_buffer(name.replace(".", "_"), t.to(dtype).contiguous().to(device), persistent=False)

        buf("patch_w", _pad_cols(p["patch.weight"].reshape(D, K).float(), self.kpad), dt)
        buf("patch_b", p["patch.bias"].float() if cfg.patch_bias else torch.zeros(D), torch.float32)
        pos = p["pos"].float()
        if cfg.has_cls:
            buf("cls_row", p["cls"].float() + pos[0], dt)
            buf("pos_patch", pos[1:], dt)
        else:
            buf("pos_patch", pos, dt)
        for name, on in (("pre_ln", cfg.pre_ln), ("final_ln", cfg.final_ln)):
            if on:
                buf(name + "_w", p[f"{name}.weight"], torch.float32)
                buf(name + "_b", p[f"{name}.bias"], torch.float32)
        nl = cfg.run_layers if cfg.run_layers is not None else cfg.num_layers
        self.nl = nl
        # ln1 -> qkv and ln2 -> fc1 with the LayerNorm folded into the GEMM (vit_ops.fold_ln_into_linear): bf16 towers
        self._ln_fused = fuse = bool(vit_ops.LN_FUSE and dt == torch.bfloat16)
        for l in range(nl):
            pre = f"layers.{l}."
            for n in ("ln1", "ln2"):
                buf(f"l{l}_{n}_w", p[pre + n + ".weight"], torch.float32)
                buf(f"l{l}_{n}_b", p[pre + n + ".bias"], torch.float32)
            # fused QKV with per-head zero padding hd -> hdp
            ws, bs = [], []
            for n in ("q", "k", "v"):
                w = p[pre + n + ".weight"].float().view(H, hd, D)
                b = p[pre + n + ".bias"].float().view(H, hd)
                wp = torch.zeros(H, hdp, D, device=w.device)
                bp = torch.zeros(H, hdp, device=w.device)
                wp[:, :hd], bp[:, :hd] = w, b
                ws.append(wp.reshape(H * hdp, D))
                bs.append(bp.reshape(H * hdp))
            if fuse:
                w2, cs, b2 = vit_ops.fold_ln_into_linear(torch.cat(ws, 0), torch.cat(bs, 0), p[pre + "ln1.weight"],
                                                         p[pre + "ln1.bias"], dt)
                buf(f"l{l}_qkv_w", w2, dt)
                buf(f"l{l}_qkv_b", b2, torch.float32)
                buf(f"l{l}_qkv_cs", cs, torch.float32)
            else:
                buf(f"l{l}_qkv_w", torch.cat(ws, 0), dt)
                buf(f"l{l}_qkv_b", torch.cat(bs, 0), torch.float32)
            pw = p[pre + "proj.weight"].float().view(D, H, hd)
            pwp = torch.zeros(D, H, hdp, device=pw.device)
            pwp[:, :, :hd] = pw
            buf(f"l{l}_proj_w", pwp.reshape(D, H * hdp), dt)
            buf(f"l{l}_proj_b", p[pre + "proj.bias"], torch.float32)
            F_ = cfg.mlp_dim
            if cfg.act == "swiglu":
                # Dinov2SwiGLUFFN: x1 | x2 = weights_in(x).chunk(2); silu(x1) * x2.  Rows INTERLEAVED (2j = gate_j, 2j + 1 =
                # up_j) so that the GEMM's epilogue forms the gated product itself (CMB_ACT_SWIGLU_PAIRS: the [tokens, 2 F]
                # intermediate is never written; round 4: 40 act_mul launches of 68 us per 24-image step gone)
                w1 = p[pre + "fc1.weight"].float()
                b1 = p[pre + "fc1.bias"].float()
                w1p = torch.stack([_pad_rows(w1[:F_], Fp), _pad_rows(w1[F_:], Fp)], 1).reshape(2 * Fp, -1)
                b1p = torch.stack([_pad_rows(b1[:F_, None], Fp)[:, 0], _pad_rows(b1[F_:, None], Fp)[:, 0]], 1).reshape(2 * Fp)
            else:
                w1p = _pad_rows(p[pre + "fc1.weight"].float(), Fp)
                b1p = _pad_rows(p[pre + "fc1.bias"].float()[:, None], Fp)[:, 0]
            if fuse:
                w2, cs, b2 = vit_ops.fold_ln_into_linear(w1p, b1p, p[pre + "ln2.weight"], p[pre + "ln2.bias"], dt)
                buf(f"l{l}_fc1_w", w2, dt)
                buf(f"l{l}_fc1_b", b2, torch.float32)
                buf(f"l{l}_fc1_cs", cs, torch.float32)
            else:
                buf(f"l{l}_fc1_w", w1p, dt)
                buf(f"l{l}_fc1_b", b1p, torch.float32)
            buf(f"l{l}_fc2_w", _pad_cols(p[pre + "fc2.weight"].float(), Fp), dt)
            buf(f"l{l}_fc2_b", p[pre + "fc2.bias"], torch.float32)
            if cfg.layerscale:
                buf(f"l{l}_ls1", p[pre + "ls1"], torch.float32)
                buf(f"l{l}_ls2", p[pre + "ls2"], torch.float32)
        self._packed = True
        return self

    # ---------------------------------------------------------------------------------------------
    def _b(self, name):
        return getattr(self, name)

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """images [B,3,S,S] (any float dtype) -> token features [B, num_patches, D] in the compute dtype
        (CLS dropped; CLIP: hidden_states[-2]; SigLIP/DINOv2: after the final LayerNorm)."""
        gen = self.forward_steps(images)
        try:
            req = next(gen)
            while True:
                req = gen.send(ops.k_gemm(**req[1]))
        except StopIteration as stop:
            return stop.value

    def forward_steps(self, images: torch.Tensor):
        """The forward as a generator: every residual linear of a block (the attention projection, fc2) is YIELDED as
        ``(kind, k_gemm keyword dict)`` and its result sent back in; everything else is launched as the generator advances.
        ``forward`` answers each request with ``ops.k_gemm``; ``forward_paired`` advances two frozen trunks in lock-step and
        answers same-kind requests of the two with ONE ``ops.k_gemm_pair`` launch (the towers are independent:
        cambrian_arch.py:271-278).  The generator's return value is ``forward``'s."""
        if not self._packed:
            raise L.CambrianAmdError("ViTTrunk weights are not loaded")
        cfg, dt = self.cfg, self.compute_dtype
        B = images.shape[0]
        T, D = cfg.num_patches, cfg.hidden_size
        N = T + (1 if cfg.has_cls else 0)
        dev = images.device
        img = images if images.dtype in (torch.float32, torch.bfloat16) else images.float()
        cols = vit_ops.k_patchify(img, cfg.patch_size, self.kpad, dt)                  # [B*T, Kpad]
        seq = torch.empty((B, N, D), dtype=dt, device=dev)
        off = D if cfg.has_cls else 0
        ops.k_gemm(cols, self.patch_w, bias=self.patch_b,
                   residual=self.pos_patch, r_map=L.make_map(T, T, 0, 0, D),
                   out=seq.view(-1)[off:], c_map=L.make_map(T, T, N * D, 0, D))
        if cfg.has_cls:
            vit_ops.k_bcast_rows(seq, N * D, B, self.cls_row)
        x = seq.view(B * N, D)
        if cfg.pre_ln:
            x, _, _ = ops.k_layernorm_fwd(x, self.pre_ln_w, self.pre_ln_b, cfg.ln_eps, want_stats=False)
        act = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF, "gelu_tanh": L.ACT_GELU_TANH,
               "swiglu": L.ACT_SWIGLU_PAIRS}[cfg.act]
        scale = 1.0 / math.sqrt(cfg.head_dim)
        for l in range(self.nl):
            g = lambda n: self._b(f"l{l}_{n}")  # noqa: E731
            if self._ln_fused:   # row statistics + the LayerNorm inside the GEMM's epilogue: no normalised copy
                qkv = ops.k_gemm(x, g("qkv_w"), bias=g("qkv_b"), row_stats=ops.k_row_stats(x, cfg.ln_eps), col_sum=g("qkv_cs"))
            else:
                h, _, _ = ops.k_layernorm_fwd(x, g("ln1_w"), g("ln1_b"), cfg.ln_eps, want_stats=False)
                qkv = ops.k_gemm(h, g("qkv_w"), bias=g("qkv_b"))
            a = vit_ops.k_vit_attn(qkv, B, N, cfg.num_heads, cfg.head_dim_pad, scale)
            x = yield ("proj", dict(a=a, w=g("proj_w"), bias=g("proj_b"), colscale=g("ls1") if cfg.layerscale else None, residual=x))
            if self._ln_fused:
                f = ops.k_gemm(x, g("fc1_w"), bias=g("fc1_b"), act=act, row_stats=ops.k_row_stats(x, cfg.ln_eps),
                               col_sum=g("fc1_cs"))
            else:
                h, _, _ = ops.k_layernorm_fwd(x, g("ln2_w"), g("ln2_b"), cfg.ln_eps, want_stats=False)
                f = ops.k_gemm(h, g("fc1_w"), bias=g("fc1_b"), act=act)  # (swiglu: [tokens, F] = silu(gate) * up already)
            x = yield ("fc2", dict(a=f, w=g("fc2_w"), bias=g("fc2_b"), colscale=g("ls2") if cfg.layerscale else None, residual=x))
        if cfg.final_ln:
            x, _, _ = ops.k_layernorm_fwd(x, self.final_ln_w, self.final_ln_b, cfg.ln_eps, want_stats=False)
        x = x.view(B, N, D)
        return x[:, 1:] if cfg.has_cls else x


@torch.no_grad()
def forward_paired(trunk_a: "ViTTrunk", images_a: torch.Tensor, trunk_b: "ViTTrunk", images_b: torch.Tensor):
    """Two frozen trunks advanced in lock-step on the current stream: while both have blocks left, block l of A and block l of B
    run side by side and their same-kind residual linears leave as one ``ops.k_gemm_pair`` call (one launch when the library's
    round arithmetic says it pays — DINOv2 beside SigLIP at 24 images: 414 + 345 tiles = 3.0 + 2.9 rounds on 138 + 118
    workgroups instead of 2 + 2 rounds on 256); the longer trunk finishes alone.  Returns (features_a, features_b), bit-identical
    to ``trunk_a(images_a), trunk_b(images_b)``."""
    ga, gb = trunk_a.forward_steps(images_a), trunk_b.forward_steps(images_b)
    out = [None, None]

    def start(gen, i):
        try:
            return next(gen)
        except StopIteration as stop:
            out[i] = stop.value
            return None

    def answer(gen, i, y):
        try:
            return gen.send(y)
        except StopIteration as stop:
            out[i] = stop.value
            return None

    ra, rb = start(ga, 0), start(gb, 1)
    while ra is not None and rb is not None:
        if ra[0] == rb[0]:
            ya, yb = ops.k_gemm_pair(ra[1], rb[1])
        else:
            ya, yb = ops.k_gemm(**ra[1]), ops.k_gemm(**rb[1])
        ra, rb = answer(ga, 0, ya), answer(gb, 1, yb)
    while ra is not None:
        ra = answer(ga, 0, ops.k_gemm(**ra[1]))
    while rb is not None:
        rb = answer(gb, 1, ops.k_gemm(**rb[1]))
    return out[0], out[1]


def resample_tokens(x: torch.Tensor, target_tokens: int, force_copy: bool = False) -> torch.Tensor:
    """Bilinear token-grid resize of the wrappers (clip_encoder.py:70-96, siglip_encoder.py:67-93,
    dino_encoder.py:128-154): [B, h*h, C] -> [B, target, C], fp32 lerp, align_corners=False.  With equal grids
    the lerp weights are exactly (1, 0), i.e. a bit-exact strided copy (used to drop the CLS row)."""
    B, T, C = x.shape
    if T == target_tokens and not (force_copy and not x.is_contiguous()):
        return x
    hi, ho = int(T ** 0.5), int(target_tokens ** 0.5)
    out = torch.empty((B, ho * ho, C), dtype=x.dtype, device=x.device)
    vit_ops.k_resample(x, hi, hi, out, ho, ho)
    return out
