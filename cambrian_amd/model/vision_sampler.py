"""Spatial Vision Aggregator on the MI355X kernels — drop-in for ``cambrian/model/vision_sampler.py``.

Same class names, constructor signatures, forward signatures and state-dict keys as the reference
(``VisionTokenSampler:407``, ``VisionCrossAttentionLayer:248``, ``MultiKVCrossAttention:155``, ``MLP:237``;
key list in SURVEY.md §8b); the module tree below only *holds* parameters, all arithmetic runs in the HIP
library through ``cambrian_amd.ops``.  There is no torch fallback: CPU tensors raise.

What is computed differently from the reference (same results within fp32 round-off):
  * ``cat([q, ctx']) @ Win^T`` is evaluated as ``q @ Win[:, :Dq]^T + ctx' @ Win[:, Dq:]^T``; on the fused path
    the second term is computed once per image ([B,1024]) and broadcast in the GEMM epilogue.
  * the K- and V-LayerNorms of one tower see the same input, so one normalised tensor is produced
    (``SvaNormFn``) and each LayerNorm's affine is folded into its projection:
    ``LN(x) @ W^T = xhat @ (W * gamma)^T + W @ beta``; K and V projections of a tower are one GEMM (N = 2048).
  * the window partition (``rearrange_vision_tower_features_train``, cambrian_arch.py:271-287) is not
    materialised on the fused path: the attention kernel walks the windows by index arithmetic.
  * the gradient of the aux features, shared by all layers, is accumulated in one fp32 buffer.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

import os

from .. import lib as L
from .. import ops

ABSORB_KV = os.environ.get("CAMBRIAN_AMD_ABSORB_KV", "1") != "0"



class MLP(nn.Module):
    """vision_sampler.py:237-245 (no biases, exact-erf GELU)."""

    def __init__(self, d_in, d_hidden, d_out):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d_hidden, bias=False)
        self.act = nn.GELU()
        self.linear_2 = nn.Linear(d_hidden, d_out, bias=False)

    def forward(self, x, residual=None):
        shape = x.shape
        h = ops.linear(x.reshape(-1, shape[-1]), self.linear_1.weight, act=L.ACT_GELU_ERF)
        res2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
        y = ops.linear(h, self.linear_2.weight, residual=res2)
        return y.view(*shape[:-1], y.shape[-1])


class MultiKVCrossAttention(nn.Module):
    """vision_sampler.py:155-234.  Parameter container + the fused K/V weight folding."""

    def __init__(self, q_dim, kv_dim_list, hidden_dim, num_heads, attention_bias=False):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_heads = num_heads
        self.head_dim = self.hidden_dim // self.num_heads
        if (self.head_dim * self.num_heads) != self.hidden_dim:
            raise ValueError(
                f"hidden_dim must be divisible by num_heads (got `hidden_dim`: {self.hidden_dim}"
                f" and `num_heads`: {self.num_heads})."
            )
        if attention_bias:
            raise NotImplementedError("attention_bias=True is never used by the reference (vision_sampler.py:263)")
        self.q_proj = nn.Sequential(nn.LayerNorm(q_dim), nn.Linear(q_dim, hidden_dim, bias=False))
        self.num_of_kvs = len(kv_dim_list)
        for i, kv_dim in enumerate(kv_dim_list):
            setattr(self, f"k_proj_{i}", nn.Sequential(nn.LayerNorm(kv_dim), nn.Linear(kv_dim, hidden_dim, bias=False)))
            setattr(self, f"v_proj_{i}", nn.Sequential(nn.LayerNorm(kv_dim), nn.Linear(kv_dim, hidden_dim, bias=False)))
        self.o_proj = nn.Linear(hidden_dim, q_dim, bias=False)

    def folded_kv(self, i: int):
        """[Wk*gk ; Wv*gv] ([2*hidden, kv_dim]) and [Wk@bk ; Wv@bv] ([2*hidden]) in fp32 (autograd-tracked)."""
        kp, vp = getattr(self, f"k_proj_{i}"), getattr(self, f"v_proj_{i}")
        return ops.fold_kv(kp[1].weight, kp[0].weight, kp[0].bias, vp[1].weight, vp[0].weight, vp[0].bias)

    def forward(self, queries, *vision_latents_attention_mask_list):
        raise RuntimeError("MultiKVCrossAttention is driven by VisionCrossAttentionLayer on the HIP path")


class VisionCrossAttentionLayer(nn.Module):
    """vision_sampler.py:248-327."""

    def __init__(self, q_dim, context_dim, kv_dim_list, kv_size_list, hidden_dim=1024, layer_idx=0):
        super().__init__()
        num_heads = 16
        self.num_of_kvs = len(kv_dim_list)
        self.q_dim, self.hidden_dim = q_dim, hidden_dim
        self.proj_context = nn.Linear(context_dim, hidden_dim, bias=False)
        self.proj_in = nn.Linear(q_dim + hidden_dim, hidden_dim, bias=False)
        self.proj_out = MLP(hidden_dim, hidden_dim, q_dim)
        self.norm = nn.LayerNorm(hidden_dim)
        self.cross_attn = MultiKVCrossAttention(hidden_dim, kv_dim_list, hidden_dim, num_heads)
        self.kv_size_list = kv_size_list
        for i, kv_size in enumerate(kv_size_list):
            if kv_size > 1:
                setattr(self, f"pos_embed_{i}", nn.Parameter(torch.randn(kv_size ** 2, hidden_dim)))

    def pos_tables(self, i: int):
        """The position table this layer adds to tower i's tokens ([] for a one-key tower): announced to ops.shared_grad so
        that the layers' LayerNorm backwards can run as one deferred pass (ops.GradAccumulator)."""
        return [getattr(self, f"pos_embed_{i}")] if self.kv_size_list[i] > 1 else []

    # ------------------------------------------------------------------------------------------
    def _run(self, q2: torch.Tensor, ctx2: torch.Tensor, ctx_rep: int, feats: Sequence[torch.Tensor],
             masks_u8: Sequence[Optional[torch.Tensor]], holders: Sequence[ops.GradAccumulator], B: int, qside: int,
             window_major: bool) -> torch.Tensor:
        """q2 [Bq,q_dim]; ctx2 [Bq,ctx] (ctx_rep == 0) or [B,ctx] (ctx_rep == queries per image);
        feats[i] 2-D [rows_i, kv_dim_i] in tower-token-major or window-major order."""
        ca = self.cross_attn
        Dq = self.q_dim
        # proj_context + proj_in (vision_sampler.py:279-292) without the concat
        c = ops.linear(ctx2, self.proj_context.weight)
        cb = ops.linear(c, self.proj_in.weight[:, Dq:])
        link = {}   # q2 is used twice (here and as the block's residual): the two gradients meet in this GEMM's backward (ops.LinearFn)
        x = ops.linear(q2, self.proj_in.weight[:, :Dq], residual=cb, res_rep=ctx_rep, link=link, role=1)
        # Q (vision_sampler.py:187)
        xn = ops.layernorm(x, ca.q_proj[0].weight, ca.q_proj[0].bias, ca.q_proj[0].eps)
        qh = ops.linear(xn, ca.q_proj[1].weight)
        # K|V per tower (vision_sampler.py:188-189,304-309).  The ONE windowed tower (every token of an s x s window is
        # seen by exactly one query) is not projected per token: its K / V projections are applied on the query side
        # (ops.sva_absorbed_attention, csrc/sva_absorbed.hip) — same result, 4.6x fewer FLOPs, no K|V / dK|dV for it.
        ai = self._absorbed_tower(qh, feats)
        kvs, absorbed = [], None
        for i, f in enumerate(feats):
            s = self.kv_size_list[i]
            pos = getattr(self, f"pos_embed_{i}") if s > 1 else None
            side = s if window_major else qside * s
            n = ops.sva_norm(f, pos, holders[i], side, s, ca.k_proj_0[0].eps)
            w, b = ca.folded_kv(i)
            if i == ai:
                absorbed = (n, w, b)
            else:
                kvs.append(ops.linear(n, w, b, heavy=True))   # K|V projection of every tower token
        if absorbed is not None:
            n, w, b = absorbed
            H = self.hidden_dim
            md = [m for i, m in enumerate(masks_u8) if i != ai]
            o = ops.sva_absorbed_attention(qh, kvs, md, n, masks_u8[ai], self.kv_size_list[ai], w[:H], b[:H], w[H:], b[H:],
                                           B, qside, window_major=window_major)
        else:
            o = ops.sva_attention(qh, kvs, list(masks_u8), list(self.kv_size_list), B, qside, ca.num_heads, ca.head_dim,
                                  window_major=window_major)
        y0 = ops.linear(o, ca.o_proj.weight, residual=x)                        # x + attn  (:319)
        y = ops.layernorm(y0, self.norm.weight, self.norm.bias, self.norm.eps)   # :321
        h = ops.linear(y, self.proj_out.linear_1.weight, act=L.ACT_GELU_ERF)     # :323
        return ops.linear(h, self.proj_out.linear_2.weight, residual=q2, link=link, role=2)   # + residual (:325)

    def _absorbed_tower(self, qh: torch.Tensor, feats) -> int:
        """Index of the tower whose K / V projections are absorbed into the query side, or -1: bf16 (MFMA kernels) or fp32
        (the exact instantiation of the same algorithm: the fp32 parity path runs what the bench line runs), 16 heads x 64 over
        1024-wide features, exactly one windowed tower (2 x 2 ... 4 x 4 keys) beside at most four one-key towers.
        ``config.fp8_projections`` COMPOSES with it (round 5): the windowed tower's K / V projections do not exist on this
        path, so the fp8 forward GEMMs are the ones that remain per token — the one-key towers' K|V projections and the aux
        projectors — and switching the mode on no longer selects the slower per-token algorithm.
        CAMBRIAN_AMD_ABSORB_KV=0 keeps the per-token projection (A/B runs)."""
        if not ABSORB_KV or qh.dtype not in (torch.bfloat16, torch.float32) or self.hidden_dim != 1024:
            return -1
        big = [i for i, s in enumerate(self.kv_size_list) if s > 1]
        if len(big) != 1 or self.kv_size_list[big[0]] > 4 or len(self.kv_size_list) - 1 > 4:
            return -1
        if feats[big[0]].shape[-1] != 1024:
            return -1
        return big[0]

    def forward(self, queries, context_feature, *vision_latents_attention_mask_list, _holders=None):
        """Reference signature (vision_sampler.py:270-275): queries [Bq,1,q_dim], context [Bq,1,ctx],
        then N window-major KV tensors [Bq, s_i^2, kv_dim] and N bool masks [Bq, s_i^2]."""
        n = self.num_of_kvs
        latents = vision_latents_attention_mask_list[:n]
        masks = vision_latents_attention_mask_list[n:]
        Bq, q_len, _ = queries.shape
        if q_len != 1:
            raise ValueError("the SVA kernels implement the reference's q_len == 1 case (one latent query per cell)")
        masks_u8 = []
        v_len = sum(int(l.shape[1]) for l in latents)
        got = sum(int(m.numel() // Bq) for m in masks)
        if len(masks) != n or got != v_len:  # vision_sampler.py:202-206
            raise ValueError(f"Attention mask should be of size {(Bq, 1, q_len, v_len)}, but is {(Bq, 1, q_len, got)}")
        for m in masks:
            m2 = m.reshape(Bq, -1).contiguous()
            masks_u8.append(m2.view(torch.uint8) if m2.dtype == torch.bool else m2.to(torch.uint8))
        holders = _holders if _holders is not None else [ops.GradAccumulator() for _ in range(n)]
        feats = []
        for i, lat in enumerate(latents):
            f = lat.reshape(-1, lat.shape[-1])
            if f.dtype != queries.dtype:
                f = f.to(queries.dtype)
            feats.append(ops.shared_grad(f, holders[i], self.pos_tables(i)) if _holders is None and f.requires_grad else f)
        out = self._run(queries.reshape(Bq, -1), context_feature.reshape(Bq, -1), 0, feats, masks_u8, holders,
                        B=Bq, qside=1, window_major=True)
        return out.view(Bq, 1, -1)


class VisionTokenSampler(nn.Module):
    """vision_sampler.py:407-419 (only the production "joint" layer type; "sep" is never constructed)."""

    def __init__(self, q_dim, context_dim, kv_dim_list, kv_size_list, vision_hidden_size, num_of_layers=1,
                 layer_type="joint"):
        super().__init__()
        assert layer_type in ["joint", "sep"]
        if layer_type != "joint":
            raise NotImplementedError('layer_type "sep" is an unused ablation in the reference (SURVEY.md §2 row 1)')
        self.layers = nn.ModuleList([
            VisionCrossAttentionLayer(q_dim, context_dim, kv_dim_list, kv_size_list, vision_hidden_size, idx)
            for idx in range(num_of_layers)])
        self.kv_size_list = list(kv_size_list)

    def forward(self, queries, context_feature, *vision_latents_attention_mask_list):
        """Reference calling convention (window-major, per-query context)."""
        n = len(self.kv_size_list)
        latents = list(vision_latents_attention_mask_list[:n])
        masks = vision_latents_attention_mask_list[n:]
        holders = [ops.GradAccumulator() for _ in range(n)]
        shared = []
        for i, lat in enumerate(latents):
            f = lat if lat.dtype == queries.dtype else lat.to(queries.dtype)
            shared.append(ops.shared_grad(f, holders[i], self.pos_tables(i)) if f.requires_grad else f)
        for layer in self.layers:
            queries = layer(queries, context_feature, *shared, *masks, _holders=holders)
        return queries

    def pos_tables(self, i: int):
        """Position tables of all this sampler's layers for tower i (VisionCrossAttentionLayer.pos_tables)."""
        return [t for layer in self.layers for t in layer.pos_tables(i)]

    def forward_fused(self, q2: torch.Tensor, ctx_b: torch.Tensor, feats: Sequence[torch.Tensor],
                      masks_u8: Sequence[Optional[torch.Tensor]], holders: Sequence[ops.GradAccumulator], B: int,
                      qside: int) -> torch.Tensor:
        """MI355X-first entry used by CambrianMetaForCausalLM: q2 [B*qside^2, q_dim] (one row per latent
        query), ctx_b [B, ctx] (one row per image), feats[i] [B*(qside*s_i)^2, C] in tower-token-major
        order (NOT window-rearranged), masks_u8[i] uint8 [B*qside^2, s_i^2] or None."""
        for layer in self.layers:
            q2 = layer._run(q2, ctx_b, qside * qside, feats, masks_u8, holders, B, qside, window_major=False)
        return q2
