"""Cambrian-Phi-3 on MI355X — counterpart of ``cambrian/model/language_model/cambrian_phi3.py`` (and of the vendored
``phi3/modeling_phi3.py`` it builds on).  BASELINE configs[0] is this wrapper around one CLIP tower and the
``mlp2x_gelu`` projector.

Phi-3 is the Llama block with packed parameters: ``self_attn.qkv_proj`` ([q | k | v] rows, modeling_phi3.py:397-402)
and ``mlp.gate_up_proj`` ([gate | up] rows, :303-308).  That is exactly the layout the Llama path here builds on the
fly for frozen weights, so the same kernels serve both: one GEMM -> ``cmb_qkv_rope`` (split + RoPE, token-major) ->
attention, and one GEMM -> ``cmb_act_mul`` (SiLU(gate) * up on the packed buffer).  RMSNorm / RoPE arithmetic is the
reference's (:83-97, :114-141, :257-281); the in-LLM SVA hook (:1221-1260) is line-for-line the Llama one
(cambrian_llama.py:168-207) and is inherited.  ``sliding_window`` follows the eager mask of :1180-1186 (keys more than
``window`` behind the query are masked) and only costs a mask when the sequence is longer than the window
(Phi-3-mini-4k: 2047 vs the 2048-token training sequence -> plain causal).  LongRoPE ('su'/'yarn' ``rope_scaling``,
:143-232) belongs to the 128k checkpoints the reference's scripts do not use and is rejected.  Dropouts
(resid_pdrop / embd_pdrop / attention_dropout) default to 0 in the reference config and are not modelled.

State-dict keys equal the reference's: ``model.layers.{i}.self_attn.{qkv_proj,o_proj}.weight``,
``model.layers.{i}.mlp.{gate_up_proj,down_proj}.weight``, ``…input_layernorm.weight``,
``…post_attention_layernorm.weight``, ``model.norm.weight``, ``model.embed_tokens.weight``, ``lm_head.weight``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

try:  # HF config class only
    from transformers import Phi3Config
except Exception:  # pragma: no cover
    from transformers import PretrainedConfig as Phi3Config

from ... import lib as L
from ... import ops
from .cambrian_llama import CambrianLlamaForCausalLM, CambrianLlamaModel, HipRMSNorm, _down_proj, _lin


class CambrianConfig(Phi3Config):
    model_type = "cambrian_phi3"
    debug = "debug"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not hasattr(self, "rope_theta"):  # transformers 5.x folds it into rope_parameters
            rp = getattr(self, "rope_parameters", None) or {}
            self.rope_theta = float(rp.get("rope_theta", kwargs.get("rope_theta", 10000.0)))


def _check_rope(cfg) -> None:
    scaling = getattr(cfg, "rope_scaling", None)
    kind = (scaling or {}).get("type", (scaling or {}).get("rope_type")) if isinstance(scaling, dict) else scaling
    if kind not in (None, "default"):
        raise L.CambrianAmdError(f"Phi-3 rope_scaling {kind!r} (LongRoPE, 128k checkpoints) is not supported")


class Phi3MLP(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        kw = dict(bias=False, device=device, dtype=dtype)
        self.gate_up_proj = nn.Linear(cfg.hidden_size, 2 * cfg.intermediate_size, **kw)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, **kw)

    def forward(self, x, residual=None):
        inner = ops.swiglu_packed(_lin(self.gate_up_proj, x))                    # up * silu(gate), :303-308
        return _down_proj(self.down_proj, inner, residual)                       # skip connection folded into the GEMM


class Phi3Attention(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        self.nh, self.nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.hd = cfg.hidden_size // cfg.num_attention_heads
        kw = dict(bias=False, device=device, dtype=dtype)
        self.qkv_proj = nn.Linear(cfg.hidden_size, (self.nh + 2 * self.nkv) * self.hd, **kw)
        self.o_proj = nn.Linear(self.nh * self.hd, cfg.hidden_size, **kw)

    def forward(self, x, cos, sin, attn_mask, kv_out: Optional[list] = None):
        B, S, _ = x.shape
        q, k, v = ops.qkv_rope(_lin(self.qkv_proj, x).contiguous(), cos, sin, self.nh, self.nkv, self.hd)
        if kv_out is not None:
            kv_out.append((k, v))
        from .cambrian_llama import KeyPadding
        key_valid = attn_mask.key_valid if isinstance(attn_mask, KeyPadding) else None
        if (attn_mask is None or key_valid is not None) and torch.is_grad_enabled() and q.requires_grad \
                and ops.causal_attention_supported(q, k):
            o = ops.causal_attention(q, k, v, key_valid)  # head_dim 128 geometries (Phi-3-medium); mini is 96 -> SDPA
        else:
            if key_valid is not None:
                attn_mask = attn_mask.dense()
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, is_causal=attn_mask is None,
                                               enable_gqa=self.nkv != self.nh)
        return _lin(self.o_proj, o.transpose(1, 2).reshape(B, S, self.nh * self.hd))

    def decode(self, x, cos, sin, kcache, vcache, t: int, key_mask):
        """One new token per sequence against the cache (see LlamaAttention.decode)."""
        B = x.shape[0]
        q, k, v = ops.qkv_rope(self.qkv_proj(x).contiguous(), cos, sin, self.nh, self.nkv, self.hd)   # [B,h,1,hd]
        kcache[:, :, t] = k[:, :, 0]
        vcache[:, :, t] = v[:, :, 0]
        o = F.scaled_dot_product_attention(q, kcache[:, :, :t + 1], vcache[:, :, :t + 1],
                                           attn_mask=key_mask[:, None, None, :], enable_gqa=self.nkv != self.nh)
        return self.o_proj(o.transpose(1, 2).reshape(B, 1, self.nh * self.hd))


class Phi3DecoderLayer(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        _check_rope(cfg)
        self.self_attn = Phi3Attention(cfg, device, dtype)
        self.mlp = Phi3MLP(cfg, device, dtype)
        self.input_layernorm = HipRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device, dtype)
        self.post_attention_layernorm = HipRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device, dtype)

    def forward(self, x, cos, sin, attn_mask, kv_out: Optional[list] = None):
        n1 = self.input_layernorm
        x, xn = ops.rmsnorm_fork(x, n1.weight, n1.variance_epsilon)   # (skip path, attention input): one backward node
        a = self.self_attn(xn, cos, sin, attn_mask, kv_out)
        n2 = self.post_attention_layernorm
        x, h = ops.add_rmsnorm(x, a, n2.weight, n2.variance_epsilon)              # :903-911 residual + norm, one pass
        return self.mlp(h, residual=x)

    def decode(self, x, cos, sin, kcache, vcache, t, key_mask):
        x = x + self.self_attn.decode(self.input_layernorm(x), cos, sin, kcache, vcache, t, key_mask)
        return x + self.mlp(self.post_attention_layernorm(x))


class CambrianPhi3Model(CambrianLlamaModel):
    """``CambrianMetaModel`` + Phi-3 backbone (cambrian_phi3.py:38-43)."""
    config_class = CambrianConfig
    layer_class = Phi3DecoderLayer


class CambrianPhi3ForCausalLM(CambrianLlamaForCausalLM):
    """cambrian_phi3.py:46-175: same forward contract as the Llama wrapper (prepare_inputs_labels_for_multimodal ->
    decoder with the SVA hook -> fp32 logits + shifted CE); ``generate`` is shared."""
    config_class = CambrianConfig
    model_class = CambrianPhi3Model


def phi3_mini_config(**overrides) -> CambrianConfig:
    """microsoft/Phi-3-mini-4k-instruct geometry (scripts' Phi-3 runs; BASELINE configs[0])."""
    kw = dict(vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32, num_attention_heads=32,
              num_key_value_heads=32, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096,
              original_max_position_embeddings=4096, sliding_window=2047, pad_token_id=32000)
    kw.update(overrides)
    return CambrianConfig(**kw)
