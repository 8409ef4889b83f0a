"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in plain fp32 torch arithmetic, of the Spatial Vision Aggregator of the reference:
``cambrian/model/vision_sampler.py`` — MultiKVCrossAttention (:155-234), MLP (:237-245),
VisionCrossAttentionLayer (:248-327), VisionTokenSampler (:407-419).  Parameters are taken as a flat
state dict with the reference's own key names so that a fixture produced by the real
reference module can be fed in unchanged.

Pinned against the reference itself: ``tests/golden/make_golden.py`` imports
``/root/reference/cambrian/model/vision_sampler.py`` and stores its outputs and gradients on seeded
inputs in ``tests/golden/sva_*.pt``; ``tests/test_oracle_golden.py`` replays them through this file.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch

NUM_HEADS = 16  # hard-coded in the reference: vision_sampler.py:251


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm semantics (biased variance over the last dim, eps inside the sqrt)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (vision_sampler.py:241)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def multi_kv_cross_attention(p: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor,
                             kvs: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]) -> torch.Tensor:
    """vision_sampler.py:177-234.  x [Bq,1,hidden]; kvs[i] [Bq,s_i^2,kv_dim_i]; masks[i] bool [Bq,1,1,s_i^2]
    (True = may attend).  Softmax in fp32 with -inf on masked keys == torch SDPA with a bool mask."""
    n = len(kvs)
    bq, q_len, _ = x.shape
    q = layer_norm(x, p[prefix + "q_proj.0.weight"], p[prefix + "q_proj.0.bias"]) @ p[prefix + "q_proj.1.weight"].T
    ks, vs = [], []
    for i in range(n):
        kn = layer_norm(kvs[i], p[f"{prefix}k_proj_{i}.0.weight"], p[f"{prefix}k_proj_{i}.0.bias"])
        vn = layer_norm(kvs[i], p[f"{prefix}v_proj_{i}.0.weight"], p[f"{prefix}v_proj_{i}.0.bias"])
        ks.append(kn @ p[f"{prefix}k_proj_{i}.1.weight"].T)
        vs.append(vn @ p[f"{prefix}v_proj_{i}.1.weight"].T)
    k = torch.cat(ks, dim=1)
    v = torch.cat(vs, dim=1)
    v_len = k.shape[1]
    hidden = q.shape[-1]
    hd = hidden // NUM_HEADS
    mask = torch.cat(list(masks), dim=-1)
    if mask.shape != (bq, 1, q_len, v_len):  # :202-206
        raise ValueError(f"Attention mask should be of size {(bq, 1, q_len, v_len)}, but is {tuple(mask.shape)}")
    qh = q.view(bq, q_len, NUM_HEADS, hd).transpose(1, 2)
    kh = k.view(bq, v_len, NUM_HEADS, hd).transpose(1, 2)
    vh = v.view(bq, v_len, NUM_HEADS, hd).transpose(1, 2)
    scores = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
    scores = scores.masked_fill(~mask, float("-inf"))
    attn = torch.softmax(scores, dim=-1) @ vh
    attn = attn.transpose(1, 2).reshape(bq, q_len, hidden)
    return attn @ p[prefix + "o_proj.weight"].T


def cross_attention_layer(p: Dict[str, torch.Tensor], prefix: str, queries: torch.Tensor, context: torch.Tensor,
                          kvs: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]) -> torch.Tensor:
    """vision_sampler.py:270-327 (the "joint" layer)."""
    residual = queries
    ctx = context @ p[prefix + "proj_context.weight"].T                       # :279
    x = torch.cat([queries, ctx], dim=-1) @ p[prefix + "proj_in.weight"].T    # :281,292
    masks4 = [m.view(m.shape[0], 1, 1, -1).expand(-1, -1, x.shape[1], -1) for m in masks]  # :297-302
    kv_pos = []
    for i, kv in enumerate(kvs):                                              # :304-309
        if kv.shape[1] > 1:
            kv_pos.append(kv + p[f"{prefix}pos_embed_{i}"][None, :, :].to(kv.dtype))
        else:
            kv_pos.append(kv)
    a = multi_kv_cross_attention(p, prefix + "cross_attn.", x, kv_pos, masks4)
    x = layer_norm(x + a, p[prefix + "norm.weight"], p[prefix + "norm.bias"])  # :319-321
    h = gelu_erf(x @ p[prefix + "proj_out.linear_1.weight"].T) @ p[prefix + "proj_out.linear_2.weight"].T
    return h + residual                                                        # :325


def vision_token_sampler(p: Dict[str, torch.Tensor], queries: torch.Tensor, context: torch.Tensor,
                         kvs: Sequence[torch.Tensor], masks: Sequence[torch.Tensor], prefix: str = "") -> torch.Tensor:
    """vision_sampler.py:416-419: apply layers.{l} in order."""
    n_layers = 0
    while f"{prefix}layers.{n_layers}.proj_in.weight" in p:
        n_layers += 1
    for l in range(n_layers):
        queries = cross_attention_layer(p, f"{prefix}layers.{l}.", queries, context, kvs, masks)
    return queries


def init_sampler_params(q_dim: int, context_dim: int, kv_dim_list: Sequence[int], kv_size_list: Sequence[int],
                        hidden: int, num_layers: int, gen: torch.Generator) -> Dict[str, torch.Tensor]:
    """Seeded parameters with the reference's shapes/keys and PyTorch-default-like scales
    (kaiming-uniform linears, unit/zero LayerNorm perturbed so the affine is exercised, pos_embed ~ N(0,1))."""
    def lin(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound

    def ln(d, key, out):
        out[key + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=gen)
        out[key + ".bias"] = 0.1 * torch.randn(d, generator=gen)

    p: Dict[str, torch.Tensor] = {}
    for l in range(num_layers):
        pre = f"layers.{l}."
        p[pre + "proj_context.weight"] = lin(hidden, context_dim)
        p[pre + "proj_in.weight"] = lin(hidden, q_dim + hidden)
        p[pre + "proj_out.linear_1.weight"] = lin(hidden, hidden)
        p[pre + "proj_out.linear_2.weight"] = lin(q_dim, hidden)
        ln(hidden, pre + "norm", p)
        ln(hidden, pre + "cross_attn.q_proj.0", p)
        p[pre + "cross_attn.q_proj.1.weight"] = lin(hidden, hidden)
        for i, kd in enumerate(kv_dim_list):
            ln(kd, pre + f"cross_attn.k_proj_{i}.0", p)
            p[pre + f"cross_attn.k_proj_{i}.1.weight"] = lin(hidden, kd)
            ln(kd, pre + f"cross_attn.v_proj_{i}.0", p)
            p[pre + f"cross_attn.v_proj_{i}.1.weight"] = lin(hidden, kd)
        p[pre + "cross_attn.o_proj.weight"] = lin(hidden, hidden)
        for i, s in enumerate(kv_size_list):
            if s > 1:
                p[pre + f"pos_embed_{i}"] = torch.randn(s * s, hidden, generator=gen)
    return p
