"""Cambrian-Llama on MI355X — counterpart of ``cambrian/model/language_model/cambrian_llama.py``.

What is on the hot path here:
  * the in-LLM SVA hook (:168-207, static branch): after decoder layer ``start + k*stride`` the 576 latent-query
    rows of ``hidden[:, image_position : image_position+600]`` (24x25 grid, newline column skipped) go through
    ``vision_sampler_layers[k]`` and are written back in place — one strided gather kernel, the fused SVA layer,
    one strided scatter kernel;
  * RMSNorm (the reference's patched fp32 version, train_fsdp.py:1429-1438) and RoPE as HIP kernels;
  * fp32 logits + shifted cross-entropy exactly as :402-422.
The decoder's own GEMMs / causal attention stay stock PyTorch-ROCm (hipBLASLt, SDPA) per the north star; the
reference file's decoder loop is written against the transformers==4.37 tuple API and does not run on the installed
transformers 5.x, so the loop is re-stated here with HF-compatible parameter names
(``model.layers.{i}.self_attn.q_proj.weight`` ...).
"""
from __future__ import annotations

import os

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint as _checkpoint

try:  # HF config class only (no HF modelling code is used)
    from transformers import LlamaConfig
    from transformers.modeling_outputs import CausalLMOutputWithPast
except Exception:  # pragma: no cover - transformers is part of the image
    LlamaConfig = object
    CausalLMOutputWithPast = None

from ... import ops
from ...constants import IGNORE_INDEX
from ..cambrian_arch import CambrianMetaForCausalLM, CambrianMetaModel, SvaContext


class CambrianConfig(LlamaConfig):
    model_type = "cambrian_llama"
    debug = "debug"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # transformers 5.x folds rope_theta into `rope_parameters`; keep the 4.37 attribute the reference reads
        if not hasattr(self, "rope_theta"):
            rp = getattr(self, "rope_parameters", None) or {}
            self.rope_theta = float(rp.get("rope_theta", kwargs.get("rope_theta", 10000.0)))


class SvaDynamic:
    """In-LLM SVA inputs of the eval / generate branch: the reference-format lists of cambrian_arch.py:422-451."""

    def __init__(self, feats, masks, final_size, ctx):
        self.feats, self.masks, self.final_size, self.ctx = list(feats), list(masks), list(final_size), ctx


class HipRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.rmsnorm(x, self.weight, self.variance_epsilon)


def _fused_frozen_weight(owner: nn.Module, key: str, mods) -> Optional[torch.Tensor]:
    """[W_0; W_1; ...] of several bias-free nn.Linear whose weights are FROZEN (the pre-training stage freezes the
    whole LLM, train_fsdp.py:1677-1685), cached on ``owner`` and rebuilt when a weight moves or changes.  One GEMM with
    N = sum(N_i) then replaces len(mods) GEMMs forward and, in the backward, len(mods) dX GEMMs plus the adds that
    accumulate them.  Trainable or biased projections return None (callers keep the separate-GEMM path); parameter
    names / state_dict keys stay HF's (q_proj.weight, ...)."""
    ws = [m.weight for m in mods]
    if any(w.requires_grad for w in ws) or any(m.bias is not None for m in mods):
        return None
    if getattr(owner, "_cmb_no_weight_cache", False):
        return None  # ZeRO-3 unit: a cached concatenation would keep the whole layer resident on every rank
    tag = tuple((w.data_ptr(), w._version, w.dtype) for w in ws)
    cached = owner.__dict__.get(key)
    if cached is None or cached[0] != tag:
        cached = (tag, torch.cat([w.detach() for w in ws], dim=0).contiguous())
        owner.__dict__[key] = cached
    return cached[1]


class FrozenLinearFn(torch.autograd.Function):
    """y = x @ W^T (+ residual) for a FROZEN W [N,K], with the backward dX = g @ W computed as ``F.linear(g, W_t)`` on a
    cached transposed copy W_t [K,N].  hipBLASLt is fastest when both operands are contiguous along the contraction
    dimension (tools/mm_layout_probe.py: 1.34-1.64 PFLOP/s for that form against 1.36-1.41 for ``g @ W`` at the decoder's
    shapes), which the forward already is; the copy gives the backward the same form (-12 % on every decoder dX GEMM) for
    one more resident copy of the frozen weights (15 GB of the 288 GB for Llama-3-8B)."""

    @staticmethod
    def forward(ctx, x2, w, w_t, res2, out_shape):
        """2-D operands, result written into a FRESH tensor of the caller's N-D ``out_shape`` (the GEMM's ``out=`` is its
        2-D view; no autograd is recorded in here).  The result is therefore NOT a view: the SVA hook's in-place scatter
        into a decoder layer's output stays a plain in-place op — on a ``y2d.view(B, S, H)`` made outside, autograd wraps
        it in CopySlices (a clone and a copy of [B, S, H] per hook in the backward, and the gradient tensor the hook's
        scatter returns is no longer the one the layer receives)."""
        ctx.save_for_backward(w_t)
        ctx.has_res = res2 is not None
        out = torch.empty(out_shape, dtype=x2.dtype, device=x2.device)
        o2 = out.view(-1, w.shape[0])
        if res2 is None:
            torch.mm(x2, w.t(), out=o2)
        else:
            torch.addmm(res2, x2, w.t(), out=o2)
        return out

    @staticmethod
    def backward(ctx, g):
        (w_t,) = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        dx = F.linear(g2, w_t) if ctx.needs_input_grad[0] else None
        return dx, None, None, (g2 if ctx.has_res else None), None


def _frozen_transposed(owner: nn.Module, key: str, w: torch.Tensor) -> torch.Tensor:
    """W^T (contiguous) of a frozen weight, cached on ``owner`` and rebuilt when the weight moves or changes."""
    tag = (w.data_ptr(), w._version, w.dtype)
    cached = owner.__dict__.get(key)
    if cached is None or cached[0] != tag:
        cached = (tag, w.detach().t().contiguous())
        owner.__dict__[key] = cached
    return cached[1]


def frozen_linear(owner: nn.Module, key: str, x: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None):
    """``F.linear(x, w)`` (+ residual); frozen bias-free GPU weights take FrozenLinearFn, anything else the stock path."""
    if w.requires_grad or not x.is_cuda or getattr(owner, "_cmb_no_weight_cache", False):
        # (ZeRO-3 units: no transposed copy either — the layer's storage is dropped between its uses, zero3.py)
        y = F.linear(x, w)
        return y if residual is None else residual + y
    x2 = x.reshape(-1, x.shape[-1])
    res2 = None if residual is None else residual.reshape(-1, w.shape[0])
    if torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)):
        return FrozenLinearFn.apply(x2, w, _frozen_transposed(owner, key, w), res2, (*x.shape[:-1], w.shape[0]))
    # eval / generate: nothing will be back-propagated, so no transposed copy is made or kept
    y = F.linear(x2, w) if res2 is None else torch.addmm(res2, x2, w.t())
    return y.view(*x.shape[:-1], w.shape[0])


def _lin(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear forward; frozen bias-free weights go through FrozenLinearFn (faster dX)."""
    if lin.bias is not None:
        return lin(x)
    return frozen_linear(lin, "_w_t", x, lin.weight)


def _down_proj(lin: nn.Linear, inner: torch.Tensor, residual: Optional[torch.Tensor]) -> torch.Tensor:
    if lin.bias is not None:
        y = lin(inner)
        return y if residual is None else residual + y
    return frozen_linear(lin, "_w_t", inner, lin.weight, residual)   # skip connection inside the GEMM (C = residual)


class LlamaMLP(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        kw = dict(bias=False, device=device, dtype=dtype)
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, **kw)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, **kw)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, **kw)

    def forward(self, x, residual=None):
        """``residual``: the layer's skip connection — added by the down-projection GEMM itself (C = residual, beta = 1)
        instead of a separate elementwise pass over [tokens, hidden]."""
        w_gu = _fused_frozen_weight(self, "_w_gate_up", (self.gate_proj, self.up_proj))
        if w_gu is not None and x.is_cuda:
            inner = ops.swiglu_packed(frozen_linear(self, "_w_gate_up_t", x, w_gu))
        else:
            inner = ops.swiglu(self.gate_proj(x), self.up_proj(x))
        return _down_proj(self.down_proj, inner, residual)


class KeyPadding:
    """The decoder's attention mask when it is "causal AND key-padding, diagonal open" (every collator batch without a
    sliding window): carried as the [B, S] key mask so the HIP flash kernels can apply it tile by tile; ``dense()`` builds
    the [B, 1, S, S] boolean mask for the stock kernel (what the reference's HF decoder materialises,
    cambrian_llama.py:142-166)."""

    def __init__(self, key_valid: torch.Tensor):
        self.key_valid = key_valid.to(torch.bool)
        self._dense = None

    def dense(self) -> torch.Tensor:
        if self._dense is None:
            B, S = self.key_valid.shape
            dev = self.key_valid.device
            m = torch.ones(S, S, dtype=torch.bool, device=dev).tril_()[None, None] & self.key_valid[:, None, None, :]
            # a fully masked query row would be NaN in SDPA; padded rows attend to themselves (their loss is ignored)
            self._dense = m | torch.eye(S, dtype=torch.bool, device=dev)[None, None]
        return self._dense


class LlamaAttention(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        self.nh, self.nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        kw = dict(bias=getattr(cfg, "attention_bias", False), device=device, dtype=dtype)
        self.q_proj = nn.Linear(cfg.hidden_size, self.nh * self.hd, **kw)
        self.k_proj = nn.Linear(cfg.hidden_size, self.nkv * self.hd, **kw)
        self.v_proj = nn.Linear(cfg.hidden_size, self.nkv * self.hd, **kw)
        self.o_proj = nn.Linear(self.nh * self.hd, cfg.hidden_size, **kw)

    def forward(self, x, cos, sin, attn_mask, kv_out: Optional[list] = None):
        """``kv_out``: if a list, (k, v) [B,nkv,S,hd] (post-RoPE) are appended — the prefill of ``generate()``."""
        B, S, _ = x.shape
        w_qkv = _fused_frozen_weight(self, "_w_qkv", (self.q_proj, self.k_proj, self.v_proj))
        if w_qkv is not None and x.is_cuda:
            q, k, v = ops.qkv_rope(frozen_linear(self, "_w_qkv_t", x, w_qkv), cos, sin, self.nh, self.nkv, self.hd)
            if kv_out is not None:
                kv_out.append((k, v))
            # attn_mask is None (plain causal) or a KeyPadding marker (causal AND the collator's key mask, diagonal open):
            # both run on flash_bwd.hip; only a dense mask (sliding window) falls through to the stock kernel
            key_valid = attn_mask.key_valid if isinstance(attn_mask, KeyPadding) else None
            if (attn_mask is None or key_valid is not None) and torch.is_grad_enabled() and q.requires_grad \
                    and ops.causal_attention_supported(q, k):
                o = ops.causal_attention(q, k, v, key_valid)
            else:
                if key_valid is not None:
                    attn_mask = attn_mask.dense()
                o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, is_causal=attn_mask is None,
                                                   enable_gqa=self.nkv != self.nh)
            return _lin(self.o_proj, o.transpose(1, 2).reshape(B, S, self.nh * self.hd))
        q = ops.rope(self.q_proj(x).view(B * S, self.nh, self.hd), cos, sin).view(B, S, self.nh, self.hd).transpose(1, 2)
        k = ops.rope(self.k_proj(x).view(B * S, self.nkv, self.hd), cos, sin).view(B, S, self.nkv, self.hd).transpose(1, 2)
        v = self.v_proj(x).view(B, S, self.nkv, self.hd).transpose(1, 2)
        if kv_out is not None:
            kv_out.append((k, v))
        key_valid = attn_mask.key_valid if isinstance(attn_mask, KeyPadding) else None
        if x.is_cuda and (attn_mask is None or key_valid is not None) and torch.is_grad_enabled() and q.requires_grad \
                and ops.causal_attention_supported(q, k):     # trainable projections (finetune stage): same HIP attention
            o = ops.causal_attention(q, k, v, key_valid)
            return self.o_proj(o.transpose(1, 2).reshape(B, S, self.nh * self.hd))
        if key_valid is not None:
            attn_mask = attn_mask.dense()
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, is_causal=attn_mask is None,
                                           enable_gqa=self.nkv != self.nh)
        return self.o_proj(o.transpose(1, 2).reshape(B, S, self.nh * self.hd))

    def decode(self, x, cos, sin, kcache, vcache, t: int, key_mask):
        """One new token per sequence against the cache (decode steps never see the SVA hook: cambrian_llama.py:174,
        ``prepare_inputs`` early-outs for length-1 inputs).  x [B,1,H]; k/v caches [B,nkv,Lmax,hd], slot t is filled here;
        key_mask bool [B, t+1]."""
        B = x.shape[0]
        q = ops.rope(self.q_proj(x).view(B, self.nh, self.hd), cos, sin).view(B, 1, self.nh, self.hd).transpose(1, 2)
        k = ops.rope(self.k_proj(x).view(B, self.nkv, self.hd), cos, sin)
        kcache[:, :, t] = k
        vcache[:, :, t] = self.v_proj(x).view(B, self.nkv, self.hd)
        o = F.scaled_dot_product_attention(q, kcache[:, :, :t + 1], vcache[:, :, :t + 1], attn_mask=key_mask[:, None, None, :],
                                           enable_gqa=self.nkv != self.nh)
        return self.o_proj(o.transpose(1, 2).reshape(B, 1, self.nh * self.hd))


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg, device, dtype):
        super().__init__()
        self.self_attn = LlamaAttention(cfg, device, dtype)
        self.mlp = LlamaMLP(cfg, device, dtype)
        self.input_layernorm = HipRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device, dtype)
        self.post_attention_layernorm = HipRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device, dtype)

    def forward(self, x, cos, sin, attn_mask, kv_out: Optional[list] = None):
        n1 = self.input_layernorm
        x, xn = ops.rmsnorm_fork(x, n1.weight, n1.variance_epsilon)   # (skip path, attention input): one backward node
        a = self.self_attn(xn, cos, sin, attn_mask, kv_out)
        n2 = self.post_attention_layernorm
        x, h = ops.add_rmsnorm(x, a, n2.weight, n2.variance_epsilon)  # x + a and norm(x + a) in one pass
        return self.mlp(h, residual=x)

    def decode(self, x, cos, sin, kcache, vcache, t, key_mask):
        x = x + self.self_attn.decode(self.input_layernorm(x), cos, sin, kcache, vcache, t, key_mask)
        return x + self.mlp(self.post_attention_layernorm(x))


class LlamaBackbone(nn.Module):
    """HF ``LlamaModel`` parameter layout (embed_tokens / layers / norm), minimal forward.  ``layer_class`` is the
    decoder block; cambrian_phi3.py swaps in the Phi-3 block (packed qkv_proj / gate_up_proj parameters)."""
    layer_class = None  # set below, once LlamaDecoderLayer's own class statement has run

    def __init__(self, config, device=None, llm_dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.llm_dtype = llm_dtype
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, device=device, dtype=llm_dtype)
        block = type(self).layer_class or LlamaDecoderLayer
        self.layers = nn.ModuleList([block(config, device, llm_dtype) for _ in range(config.num_hidden_layers)])
        self.norm = HipRMSNorm(config.hidden_size, config.rms_norm_eps, device, llm_dtype)

    @property
    def dtype(self):
        """dtype for NEW trainable parameters (vision_query, image_newline): fp32 masters, as in the reference
        run where parameters are fp32 and only the compute is bf16 (fsdp_config.json:6, train_fsdp.py:1267-1398)."""
        return torch.float32


class CambrianLlamaModel(CambrianMetaModel, LlamaBackbone):
    config_class = CambrianConfig

    def __init__(self, config, device=None, llm_dtype=torch.bfloat16):
        # CambrianMetaModel.__init__ -> super().__init__(config) -> LlamaBackbone.__init__(config)
        self._ctor_kw = (device, llm_dtype)
        LlamaBackbone.__init__(self, config, device, llm_dtype)
        if hasattr(config, "mm_vision_tower_aux_list"):
            # re-run the mixin's config-driven construction on the already initialised backbone
            _BackboneShim.attach(self, config)

    def forward(self, inputs_embeds: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, sva=None, kv_out: Optional[list] = None) -> torch.Tensor:
        cfg = self.config
        B, S, H = inputs_embeds.shape
        dev = inputs_embeds.device
        if position_ids is None:
            position_ids = torch.arange(S, device=dev).unsqueeze(0).expand(B, S)
        hd = self.layers[0].self_attn.hd
        cos, sin = ops.rope_table(position_ids, hd, float(getattr(cfg, "rope_theta", 10000.0)))
        attn_mask = None
        window = getattr(cfg, "sliding_window", None)
        if window is not None and S <= window + 1:
            window = None                      # every key is within reach: plain causal attention
        if attention_mask is not None and window is None:
            attn_mask = KeyPadding(attention_mask)
        elif attention_mask is not None or window is not None:
            causal = torch.ones(S, S, dtype=torch.bool, device=dev).tril_()
            if window is not None:             # Phi-3 eager mask (phi3/modeling_phi3.py:1180-1186): 0 <= i - j <= window
                causal = causal & ~torch.ones(S, S, dtype=torch.bool, device=dev).tril_(-(window + 1))
            attn_mask = causal[None, None]
            if attention_mask is not None:
                attn_mask = attn_mask & attention_mask.to(torch.bool)[:, None, None, :]
                # a fully masked query row would be NaN in SDPA; padded rows attend to themselves (their loss is ignored)
                attn_mask = attn_mask | torch.eye(S, dtype=torch.bool, device=dev)[None, None]
        hidden = inputs_embeds
        hook_layers = {}
        if sva is not None and not getattr(cfg, "connector_only", True):
            start, stride = cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers
            hook_layers = {start + k * stride: k for k in range(len(self.vision_sampler_layers))}  # :170-172
        # Activation re-computation (the reference: `--gradient_checkpointing True` in the finetune scripts,
        # scripts/cambrian/finetune_cambrian_8b.sh; FSDP wraps and checkpoints every decoder layer, fsdp_config.json:9, and
        # cambrian_llama.py:189-196 checkpoints the in-LLM SVA layers): a layer keeps only its input, its forward runs a second
        # time inside the backward.  Non-reentrant, so the custom autograd operators' saved tensors are the recomputed ones.
        ckpt = bool(getattr(cfg, "gradient_checkpointing", False)) and torch.is_grad_enabled() and kv_out is None
        for i, layer in enumerate(self.layers):
            if ckpt and hidden.requires_grad:
                hidden = _checkpoint(layer, hidden, cos, sin, attn_mask, None, use_reentrant=False)
            else:
                hidden = layer(hidden, cos, sin, attn_mask, kv_out)
            if i in hook_layers:
                if isinstance(sva, SvaDynamic):
                    hidden = self._sva_hook_dynamic(hidden, hook_layers[i], sva)
                else:
                    hidden = self._sva_hook(hidden, hook_layers[i], sva)
        return self.norm(hidden)

    def _sva_hook_dynamic(self, hidden: torch.Tensor, k: int, sva: "SvaDynamic") -> torch.Tensor:
        """cambrian_llama.py:209-253 (eval branch): every sample has its own cur_h x (cur_w + 1) block of latent
        queries + newline column starting at ``image_position``; the queries of all samples go through the sampler layer
        as one [sum h*w, 1, H] batch (KV lists / masks / context rows were concatenated the same way by
        rearrange_vision_tower_features_inference(unpad=True)) and are written back; newline rows stay."""
        p0 = self.config.image_position
        rows, nums = [], []
        for b, (h, w) in enumerate(sva.final_size):
            blk = hidden[b, p0:p0 + h * (w + 1)].view(h, w + 1, -1)
            rows.append(blk[:, :w].reshape(h * w, 1, -1))
            nums.append(h * w)
        q = torch.cat(rows, 0)
        feats = [f if f.dtype == q.dtype else f.to(q.dtype) for f in sva.feats]      # :186 (same cast in both branches)
        out = self.vision_sampler_layers[k](q, sva.ctx.to(q.dtype), *feats, *sva.masks)
        hidden = hidden.clone()
        for b, ob in enumerate(torch.split(out, nums, 0)):
            h, w = sva.final_size[b]
            hidden[b, p0:p0 + h * (w + 1)].view(h, w + 1, -1)[:, :w] = ob.view(h, w, -1)
        return hidden

    def _sva_hook(self, hidden: torch.Tensor, k: int, sva: SvaContext) -> torch.Tensor:
        """cambrian_llama.py:177-207 (static branch)."""
        cfg = self.config
        p0 = cfg.image_position
        side = int(cfg.image_token_len ** 0.5)
        hidden = hidden if hidden.is_contiguous() else hidden.contiguous()
        span = ops.region_begin("sva_in_llm")                                          # (bench.py roofline.region)
        link = None if os.environ.get("CAMBRIAN_AMD_NO_HOOK_LINK") else {}             # gather / scatter share d(hidden)
        q2 = ops.gather_query_rows(hidden, p0, side, link)                             # [B*576, H]
        q2 = ops.region_mark(q2, span, "b1")
        feats = [f if f.dtype == q2.dtype else f.to(q2.dtype) for f in sva.feats]      # :186
        sampler = self.vision_sampler_layers[k]
        if bool(getattr(cfg, "gradient_checkpointing", False)) and torch.is_grad_enabled() and q2.requires_grad:
            out = _checkpoint(sampler.forward_fused, q2, sva.ctx_b.to(q2.dtype), feats, sva.masks_u8, sva.holders, sva.B, side,
                              use_reentrant=False)                                     # cambrian_llama.py:189-196
        else:
            out = sampler.forward_fused(q2, sva.ctx_b.to(q2.dtype), feats, sva.masks_u8, sva.holders, sva.B, side)
        out = ops.region_mark(out, span, "b0")
        hidden = ops.scatter_query_rows(hidden, out, p0, side, link)
        ops.region_fwd_end(span)
        return hidden


class _BackboneShim:
    """CambrianMetaModel.__init__ calls ``super().__init__(config)`` (it is written as a mixin placed before the HF
    model class).  Our backbone is already constructed when we need the mixin's body, so run that body with a
    no-op super().__init__."""

    @staticmethod
    def attach(model: "CambrianLlamaModel", config):
        from ..cambrian_arch import _sva_modules, build_vision_projector, build_vision_tower_aux_list
        projector_type = getattr(config, "mm_projector_type", "linear")
        model.vision_tower_aux_list = build_vision_tower_aux_list(config, delay_load=True)
        if projector_type == "sva":
            _sva_modules(model, config, model.vision_tower_aux_list, config.hidden_size)
            model.vision_query = nn.Parameter(torch.randn((config.num_query_group, config.vision_hidden_size), dtype=model.dtype))
        else:
            config.mm_hidden_size = sum(t.hidden_size for t in model.vision_tower_aux_list)
            model.mm_projector = build_vision_projector(config)
        model.image_newline = nn.Parameter(torch.empty(config.hidden_size, dtype=model.dtype))


class CambrianLlamaForCausalLM(nn.Module, CambrianMetaForCausalLM):
    config_class = CambrianConfig
    model_class = CambrianLlamaModel

    def __init__(self, config, device=None, llm_dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.model = type(self).model_class(config, device, llm_dtype)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False, device=device, dtype=llm_dtype)

    def get_model(self):
        return self.model

    @property
    def device(self):
        return self.lm_head.weight.device

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def resize_token_embeddings(self, new_num_tokens: int):
        """HF ``PreTrainedModel.resize_token_embeddings`` for the two vocabulary-sized matrices (``load_pretrained_model``
        calls it after adding the image tokens, model/builder.py:158): old rows kept, new rows ~ N(0, initializer_range)."""
        old = self.model.embed_tokens.weight
        n_old, H = old.shape
        if new_num_tokens == n_old:
            return self.model.embed_tokens
        std = float(getattr(self.config, "initializer_range", 0.02))

        def grown(w):
            out = torch.empty((new_num_tokens, H), dtype=w.dtype, device=w.device).normal_(0.0, std)
            n = min(n_old, new_num_tokens)
            out[:n] = w.detach()[:n]
            return out

        emb = nn.Embedding(new_num_tokens, H, device=old.device, dtype=old.dtype)
        emb.weight = nn.Parameter(grown(old), requires_grad=old.requires_grad)
        self.model.embed_tokens = emb
        head = nn.Linear(H, new_num_tokens, bias=False, device=old.device, dtype=self.lm_head.weight.dtype)
        head.weight = nn.Parameter(grown(self.lm_head.weight), requires_grad=self.lm_head.weight.requires_grad)
        self.lm_head = head
        self.vocab_size = self.config.vocab_size = new_num_tokens
        return self.model.embed_tokens

    def forward(self, *args, **kwargs):
        """``config.fp8_projections`` (BASELINE configs[4]): the forward GEMMs of the SVA-side projections — aux
        projectors, connector and in-LLM SVA layers, mm_projector, i.e. everything that goes through ``ops.linear`` —
        run on the fp8 MFMA with row-wise e4m3 scaling; towers and decoder are untouched, the backward stays bf16."""
        try:
            with ops.fp8_projections(bool(getattr(self.config, "fp8_projections", False))):
                return self._forward(*args, **kwargs)
        finally:
            ops.weight_step_end()   # (the window prepare_inputs_labels_for_multimodal opened for this forward's linears)

    def _forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                 image_aux_attention_masks_list=None, image_sizes=None, return_dict=None, cache_position=None):
        sva = _masks = _final_size = _ctx = None
        if inputs_embeds is None:  # cambrian_llama.py:315-336
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels, sva, _masks, _final_size,
             _ctx) = self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                               labels, images, image_aux_attention_masks_list, image_sizes)
            if inputs_embeds is None:  # text-only early-out (cambrian_arch.py:346-347)
                inputs_embeds = self.model.embed_tokens(input_ids)
        if isinstance(sva, (list, tuple)):
            # reference-format lists (eval branch, or sva_fused = False): window-major KV, bool masks, per-sample (h, w),
            # one context row per query — consumed by the general per-sample hook (cambrian_llama.py:209-253)
            sva = SvaDynamic(sva, _masks, _final_size, _ctx)
        hidden = self.model(inputs_embeds.to(self.model.llm_dtype), position_ids, attention_mask, sva)
        if labels is not None and getattr(self.config, "fused_loss", False) == "scored_rows":
            # Opt-in (round 5, ``config.fused_loss = "scored_rows"``): lm_head and the cross-entropy only over the positions that
            # are scored — shifted label != IGNORE_INDEX; the collator masks the 600 visual slots, the prompt and the padding
            # (train_fsdp.py:1089-1165), a third of the synthetic batch and more of a real one.  Ignored rows contribute nothing to
            # the loss and receive a zero gradient, so loss and every gradient are those of the full computation
            # (tests/test_model_gpu.py::test_scored_rows_loss_equals_the_full_one); what changes is that ``logits`` is not
            # produced (None) — the reference's training_step reads ``loss`` only (cambrian_trainer.py:226-236).  Labels handed
            # over on the CPU (as the collator makes them) give the row list without a device synchronisation.
            Bq_, Sq_, Hd_ = hidden.shape
            shift_labels = torch.full_like(labels, IGNORE_INDEX)
            shift_labels[:, :-1] = labels[:, 1:]
            flat = shift_labels.reshape(-1)
            idx = torch.nonzero(flat != IGNORE_INDEX).squeeze(1)
            if idx.numel() > 0:
                sel = flat[idx].to(hidden.device, non_blocking=True)
                h_sel = hidden.reshape(Bq_ * Sq_, Hd_).index_select(0, idx.to(hidden.device, non_blocking=True))
                loss = ops.cross_entropy(_lin(self.lm_head, h_sel), sel, IGNORE_INDEX, inplace=True)
                if CausalLMOutputWithPast is not None:
                    return CausalLMOutputWithPast(loss=loss, logits=None)
                return {"loss": loss, "logits": None}
        logits = _lin(self.lm_head, hidden)                                          # :402-408
        loss = None
        if labels is not None and getattr(self.config, "fused_loss", False):
            # :409-422 without logits.float() / the shifted copy: labels are shifted instead of the logits (position
            # t is scored against labels[t+1], the last position of every sequence is ignored), fp32 log-sum-exp over
            # the bf16 logits in one pass, dlogits written in place of the logits in the backward.  ``logits`` is then
            # returned in the compute dtype (its .float() is the reference's tensor) and is consumed by backward().
            B_, S_, V_ = logits.shape
            shift_labels = torch.full_like(labels, IGNORE_INDEX)
            shift_labels[:, :-1] = labels[:, 1:]
            loss = ops.cross_entropy(logits.view(B_ * S_, V_), shift_labels.view(-1).to(logits.device), IGNORE_INDEX,
                                     inplace=True)
        else:
            logits = logits.float()                                                  # :409
            if labels is not None:                                                   # :411-422
                shift_logits = logits[..., :-1, :].contiguous().view(-1, self.vocab_size)
                shift_labels = labels[..., 1:].contiguous().view(-1).to(shift_logits.device)
                loss = F.cross_entropy(shift_logits, shift_labels, ignore_index=IGNORE_INDEX)
        if CausalLMOutputWithPast is not None:
            return CausalLMOutputWithPast(loss=loss, logits=logits)
        return {"loss": loss, "logits": logits}


# generate() keyword arguments that change nothing here: the K/V cache is always on, padding / bos ids are not needed by a
# loop that returns only the new tokens, and the listed output switches are accepted at their "off" value only
_GENERATE_ACCEPTED_NOOPS = {"use_cache": (True, None), "pad_token_id": None, "bos_token_id": None, "output_scores": (False, None),
                            "return_dict_in_generate": (False, None), "output_attentions": (False, None),
                            "output_hidden_states": (False, None), "num_return_sequences": (1, None),
                            "repetition_penalty": (1.0, None), "length_penalty": (1.0, None), "early_stopping": (False, None),
                            "stopping_criteria": (None,), "logits_processor": (None,), "streamer": (None,)}


def filter_logits_top_k_top_p(logits: torch.Tensor, top_k=None, top_p=None) -> torch.Tensor:
    """HF's TopKLogitsWarper then TopPLogitsWarper (the order GenerationMixin applies them in, after the temperature), on
    fp32 logits [B, V]: tokens outside the k largest, then tokens outside the smallest set whose probability mass reaches
    ``top_p`` (ascending sort, drop while the cumulative mass <= 1 - top_p, always keep the most probable), are set to -inf."""
    if top_k is not None and int(top_k) > 0:
        k = min(int(top_k), logits.shape[-1])
        kth = torch.topk(logits, k, dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and float(top_p) < 1.0:
        if not 0.0 < float(top_p):
            raise ValueError(f"`top_p` has to be a float > 0 and <= 1, but is {top_p}")
        srt, idx = torch.sort(logits, descending=False, dim=-1)
        cum = srt.softmax(-1).cumsum(-1)
        drop = cum <= (1.0 - float(top_p))
        drop[..., -1:] = False
        logits = logits.masked_fill(torch.zeros_like(drop).scatter(-1, idx, drop), float("-inf"))
    return logits


def _generate(self, inputs=None, images=None, image_sizes=None, max_new_tokens: int = 16, do_sample: bool = False,
              temperature: float = 1.0, eos_token_id=None, attention_mask=None, position_ids=None, top_p=None, top_k=None,
              num_beams: int = 1, **kwargs):
    """cambrian_llama.py:437-483: ``generate(input_ids, images=, image_sizes=)`` of the eval harness
    (eval/eval/*/…_eval.py, e.g. gqa_eval.py:108-117: do_sample / temperature / top_p / num_beams / max_new_tokens /
    use_cache).  Prefill = the eval branch of prepare_inputs_labels_for_multimodal + the decoder with the per-sample in-LLM
    SVA hook, K/V of every layer kept; decode steps = one token against the cache, no hook (cambrian_llama.py:174).
    Greedy, or sampling with temperature -> top-k -> nucleus top-p as HF's logits warpers define them; returns the NEW
    token ids [B, <= max_new_tokens] (like HF generate() called with inputs_embeds).  The reference inherits the loop from HF
    GenerationMixin; what this loop does not implement RAISES instead of being dropped: beam search (``num_beams > 1``),
    ``inputs_embeds`` (the reference raises too, cambrian_llama.py:447-448) and any keyword it does not know."""
    if "inputs_embeds" in kwargs:
        raise NotImplementedError("`inputs_embeds` is not supported")          # cambrian_llama.py:447-448
    if num_beams is not None and int(num_beams) > 1:
        raise NotImplementedError(f"generate(num_beams={num_beams}): beam search is not implemented in cambrian_amd "
                                  "(greedy / temperature / top-k / top-p sampling only); pass num_beams=1")
    if "max_length" in kwargs and kwargs["max_length"] is not None:
        raise NotImplementedError("generate(max_length=...): use max_new_tokens")
    kwargs.pop("max_length", None)
    for k, v in kwargs.items():
        ok = _GENERATE_ACCEPTED_NOOPS.get(k, ())
        if k not in _GENERATE_ACCEPTED_NOOPS or (ok is not None and v not in ok):
            raise ValueError(f"generate(): unsupported argument {k}={v!r} (cambrian_amd implements greedy / temperature / "
                             "top_k / top_p decoding with a K/V cache)")
    with torch.no_grad():
        model = self.model
        if images is not None:
            self._dynamic_path = True
            try:
                (_, position_ids, attention_mask, _, inputs_embeds, _, kv, masks, final_size, ctx) = \
                    self.prepare_inputs_labels_for_multimodal(inputs, position_ids, attention_mask, None, None, images,
                                                              image_sizes=image_sizes)
            finally:
                self._dynamic_path = False
            sva = None if kv is None else SvaDynamic(kv, masks, final_size, ctx)
        else:
            inputs_embeds, sva = model.embed_tokens(inputs), None
        B, L, _ = inputs_embeds.shape
        dev = inputs_embeds.device
        valid = torch.ones(B, L, dtype=torch.bool, device=dev) if attention_mask is None else attention_mask.bool()
        pos = (valid.long().cumsum(1) - 1).clamp_min(0) if position_ids is None else position_ids
        kv_out: list = []
        hidden = model(inputs_embeds.to(model.llm_dtype), pos, attention_mask, sva, kv_out=kv_out)
        last = valid.long().sum(1) - 1                                              # last real token of every row
        if not bool((valid[:, 1:] <= valid[:, :-1]).all()):
            last = torch.full_like(last, L - 1)                                      # left padding: the last column
        logits = self.lm_head(hidden[torch.arange(B, device=dev), last]).float()
        nh_kv, hd = model.layers[0].self_attn.nkv, model.layers[0].self_attn.hd
        Lmax = L + max_new_tokens
        caches = []
        for k, v in kv_out:
            kc = torch.zeros(B, nh_kv, Lmax, hd, dtype=k.dtype, device=dev)
            vc = torch.zeros_like(kc)
            kc[:, :, :L], vc[:, :, :L] = k, v
            caches.append((kc, vc))
        key_mask = torch.zeros(B, Lmax, dtype=torch.bool, device=dev)
        key_mask[:, :L] = valid
        n_tok = valid.long().sum(1)
        out_ids = []
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        theta = float(getattr(self.config, "rope_theta", 10000.0))
        for step in range(max_new_tokens):
            if do_sample and temperature > 0:
                warped = filter_logits_top_k_top_p(logits / temperature, top_k, top_p)
                nxt = torch.multinomial(torch.softmax(warped, -1), 1)[:, 0]
            else:
                nxt = logits.argmax(-1)
            if eos_token_id is not None:
                nxt = torch.where(done, torch.full_like(nxt, eos_token_id), nxt)
                done = done | (nxt == eos_token_id)
            out_ids.append(nxt)
            if step + 1 == max_new_tokens or (eos_token_id is not None and bool(done.all())):
                break
            t = L + step
            key_mask[:, t] = True
            cos, sin = ops.rope_table((n_tok + step)[:, None], hd, theta)
            x = model.embed_tokens(nxt)[:, None, :].to(model.llm_dtype)
            for layer, (kc, vc) in zip(model.layers, caches):
                x = layer.decode(x, cos, sin, kc, vc, t, key_mask[:, :t + 1])
            logits = self.lm_head(model.norm(x)[:, 0]).float()
        return torch.stack(out_ids, 1)


CambrianLlamaForCausalLM.generate = _generate


def llama3_8b_config(**overrides) -> CambrianConfig:
    """Meta-Llama-3-8B-Instruct geometry (the LLM of Cambrian-8B, scripts/cambrian/pretrain_cambrian_8b.sh:12)."""
    kw = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
              num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=8192)
    kw.update(overrides)
    return CambrianConfig(**kw)


def apply_release_8b_vision_config(cfg, towers: Optional[List[str]] = None, token_lens: Optional[List[int]] = None):
    """scripts/cambrian/pretrain_cambrian_8b.sh:15-27 copied onto the config as train_fsdp.py:1671-1709 does."""
    cfg.mm_vision_tower_aux_list = towers or ["siglip/CLIP-ViT-SO400M-14-384", "openai/clip-vit-large-patch14-336",
                                              "facebook/dinov2-giant-res378", "clip-convnext-XXL-multi-stage"]
    cfg.mm_vision_tower_aux_token_len_list = token_lens or [576, 576, 576, 9216]
    cfg.image_token_len = 576
    cfg.num_query_group = 1
    cfg.query_num_list = [576]
    cfg.connector_depth = 3
    cfg.vision_hidden_size = 1024
    cfg.connector_only = False
    cfg.num_of_vision_sampler_layers = 10
    cfg.start_of_vision_sampler_layers = 0
    cfg.stride_of_vision_sampler_layers = 3
    cfg.mm_projector_type = "sva"
    cfg.image_position = 91
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.unfreeze_mm_vision_tower = False
    cfg.sva_fused = True
    return cfg
