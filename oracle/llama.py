"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the LLM-side pieces the north star names:
  * RMSNorm — the reference's patched ``LlamaRMSNorm.forward`` (train_fsdp.py:1429-1438; same arithmetic as
    phi3/modeling_phi3.py:83-97): fp32, weight multiplied before the down-cast;
  * RoPE — phi3/modeling_phi3.py:114-141 (inv_freq, cos/sin in fp32) and :257-281 (rotate_half form);
  * the in-LLM SVA hook, static branch — cambrian_llama.py:177-207;
  * fp32 logits + shifted cross-entropy — cambrian_llama.py:402-422;
  * a plain Llama decoder (HF parameter names) so that whole-model logits can be compared.
Pinned by tests/golden/llama_small.pt: the reference's own Phi3RMSNorm / Phi3RotaryEmbedding / apply_rotary_pos_emb
(imported from /root/reference), the hook lines 181-207 exec'd verbatim on seeded tensors, and the installed HF
``LlamaForCausalLM`` logits for the bare decoder.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import sva as O

IGNORE_INDEX = -100


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return (w.to(torch.float32) * h).to(dt)


def rope_cos_sin(position_ids: torch.Tensor, dim: int, base: float):
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    return torch.cat((-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]), dim=-1)


def apply_rope(q, k, cos, sin):
    """q,k [B, heads, S, hd]; cos/sin [B, S, hd]."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def sva_hook(hidden: torch.Tensor, sampler_params: Dict[str, torch.Tensor], prefix: str, image_position: int,
             image_token_len: int, ctx: torch.Tensor, kvs: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]):
    """cambrian_llama.py:177-207: slice 24x25 rows, drop the newline column, run the sampler, write back."""
    side = int(image_token_len ** 0.5)
    n_nl = image_token_len + side
    blk = hidden[:, image_position:image_position + n_nl, :].clone()
    bs = blk.shape[0]
    blk = blk.view(bs, side, side + 1, -1)
    q, nl = blk[:, :, :-1, :], blk[:, :, -1:, :]
    q = q.reshape(bs * image_token_len, 1, -1)
    q = O.vision_token_sampler(sampler_params, q, ctx, [k.to(q.dtype) for k in kvs], masks, prefix=prefix)
    q = q.view(bs, side, side, -1)
    out = hidden.clone()
    out[:, image_position:image_position + n_nl] = torch.cat([q, nl], 2).flatten(1, 2)
    return out


def decoder_forward(p: Dict[str, torch.Tensor], cfg, inputs_embeds: torch.Tensor, position_ids: torch.Tensor,
                    attention_mask: Optional[torch.Tensor] = None, hook=None) -> torch.Tensor:
    """Plain Llama decoder with HF key names (model.layers.{i}....); ``hook(i, hidden)`` runs after layer i.
    Phi-3 (phi3/modeling_phi3.py): same arithmetic with packed ``qkv_proj`` (q | k | v rows, :397-402) and
    ``gate_up_proj`` (gate | up rows, :303-308) weights, and, with ``cfg.sliding_window`` set, keys further than the
    window behind the query masked (eager mask of :1180-1186: visible iff 0 <= i - j <= window)."""
    B, S, H = inputs_embeds.shape
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    hd = H // nh
    cos, sin = rope_cos_sin(position_ids, hd, cfg.rope_theta)
    causal = torch.ones(S, S, dtype=torch.bool).tril_()
    window = getattr(cfg, "sliding_window", None)
    if window is not None:
        i = torch.arange(S)
        causal = causal & ((i[:, None] - i[None, :]) <= window)
    mask = causal[None, None]
    if attention_mask is not None:
        mask = (mask & attention_mask.bool()[:, None, None, :]) | torch.eye(S, dtype=torch.bool)[None, None]
    x = inputs_embeds
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        h = rms_norm(x, p[pre + "input_layernorm.weight"], cfg.rms_norm_eps)
        if pre + "self_attn.qkv_proj.weight" in p:
            wq, wk, wv = torch.split(p[pre + "self_attn.qkv_proj.weight"], [nh * hd, nkv * hd, nkv * hd], 0)
        else:
            wq, wk, wv = (p[pre + f"self_attn.{n}_proj.weight"] for n in "qkv")
        q = (h @ wq.T).view(B, S, nh, hd).transpose(1, 2)
        k = (h @ wk.T).view(B, S, nkv, hd).transpose(1, 2)
        v = (h @ wv.T).view(B, S, nkv, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        s = s.masked_fill(~mask, float("-inf"))
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, H)
        x = x + a @ p[pre + "self_attn.o_proj.weight"].T
        h = rms_norm(x, p[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        if pre + "mlp.gate_up_proj.weight" in p:
            wg, wu = p[pre + "mlp.gate_up_proj.weight"].chunk(2, 0)
        else:
            wg, wu = p[pre + "mlp.gate_proj.weight"], p[pre + "mlp.up_proj.weight"]
        x = x + (F.silu(h @ wg.T) * (h @ wu.T)) @ p[pre + "mlp.down_proj.weight"].T
        if hook is not None:
            x = hook(i, x)
    return rms_norm(x, p["model.norm.weight"], cfg.rms_norm_eps)


def lm_loss(hidden: torch.Tensor, lm_head: torch.Tensor, labels: torch.Tensor):
    """cambrian_llama.py:402-422."""
    logits = (hidden @ lm_head.T).float()
    shift_logits = logits[..., :-1, :].contiguous().view(-1, logits.shape[-1])
    shift_labels = labels[..., 1:].contiguous().view(-1)
    return F.cross_entropy(shift_logits, shift_labels, ignore_index=IGNORE_INDEX), logits


def sva_hook_dynamic(hidden: torch.Tensor, sampler_params: Dict[str, torch.Tensor], prefix: str, start: int, final_size,
                     ctx: torch.Tensor, kvs: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]):
    """cambrian_llama.py:209-253 (eval branch): per sample a cur_h x (cur_w + 1) block starting at ``start`` holds the
    latent queries + one newline column; all samples' queries are concatenated ([sum h*w, 1, H]) through the sampler
    layer (KV lists / masks / context are already concatenated the same way) and written back."""
    bs = len(final_size)
    qs, nls, nums = [], [], []
    for b in range(bs):
        h, w = final_size[b]
        blk = hidden[b:b + 1, start:start + h * (w + 1), :].clone().view(1, h, w + 1, -1)
        qs.append(blk[:, :, :-1, :].contiguous().view(h * w, 1, -1))
        nls.append(blk[:, :, -1:, :])
        nums.append(h * w)
    q = O.vision_token_sampler(sampler_params, torch.cat(qs, 0), ctx, [k.to(hidden.dtype) for k in kvs], masks, prefix=prefix)
    out = hidden.clone()
    for b, qb in enumerate(torch.split(q, nums, 0)):
        h, w = final_size[b]
        out[b:b + 1, start:start + h * (w + 1)] = torch.cat([qb.view(1, h, w, -1), nls[b]], 2).flatten(1, 2)
    return out
