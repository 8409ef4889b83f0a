"""Multimodal glue on the MI355X kernels — drop-in for ``cambrian/model/cambrian_arch.py``.

Same mixins, method names, argument lists, 10-tuple return and state-dict keys as the reference
(``CambrianMetaModel:33``, ``initialize_vision_modules:99``, ``CambrianMetaForCausalLM:259``,
``rearrange_vision_tower_features_train:271``, ``encode_images:332``, ``prepare_inputs_labels_for_multimodal:340``).
The static ("XLA"/training) branch — the hot path of the north star — is implemented; it is selected by
``cambrian_amd.model.STATIC_PATH`` instead of ``torch_xla`` being importable (SURVEY.md §8b "path switch").
The dynamic eval/generate branch (:203-256,289-330,388-390,422-451,492-609) is ``_prepare_inputs_dynamic``
below (variable query counts per image, per-sample masks; ``rearrange_vision_tower_features_inference``, ``unpad_image``).

MI355X-first differences inside the same contract:
  * aux features stay in tower-token-major layout; the window partition of :271-287 is folded into the SVA
    attention kernel's index arithmetic (nothing is permuted or copied, bit-exact gather);
  * the global context is kept as one row per image ([B,1024]) and broadcast in a GEMM epilogue;
  * newline column + image-token splice + embedding lookup are one gather kernel (:413-420,457-490);
  * with ``config.sva_fused = True`` (default) the 7th..10th return values carry the compact fused-path
    tensors (``SvaContext``) consumed by ``cambrian_llama.py`` here; with ``False`` they are the reference's
    window-major lists, so an unmodified reference decoder wrapper can consume them.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import os

import torch
import torch.nn as nn

from .. import lib as L
from .. import ops
from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .multimodal_encoder.builder import build_vision_tower_aux_list
from .multimodal_projector.builder import HipSequential, build_vision_projector
from .vision_sampler import VisionTokenSampler

# static (fixed 2048-token layout, train) vs dynamic (eval/generate) path; replaces `IS_XLA_AVAILABLE`
# (cambrian/utils.py:17-22) which cannot be the switch on a machine without torch_xla.
STATIC_PATH = True
_TOWER_STREAMS = os.environ.get("CAMBRIAN_AMD_TOWER_STREAMS", "0")   # frozen towers on side HIP streams (encode_images)


_PAIR_TOWERS = os.environ.get("CAMBRIAN_AMD_PAIR_TOWERS", "1") != "0"   # two frozen ViT trunks in lock-step (encode_images)


def _pair_rounds_gain(ta, tb, batch: int, n_cu: int = 256) -> float:
    """Whole-round arithmetic of running trunks ``ta`` / ``tb`` (ViTTrunk) block by block side by side: K tiles + epilogue per
    256 x 256 item, rounds = ceil(items / workgroups), for the two residual linears of a block (the launches forward_paired
    hands to cmb_gemm_pair, which repeats this arithmetic per call).  Returns the modelled time saved per common block in
    item units (<= 0: do not pair)."""
    def items(t, n_out):
        cfg = t.cfg
        rows = batch * (cfg.num_patches + (1 if cfg.has_cls else 0))
        return -(-rows // 256) * -(-n_out // 256)

    def best_pair(i0, c0, i1, c1):
        best = float("inf")
        for g in range(8, n_cu - 7):
            best = min(best, max(-(-i0 // g) * c0, -(-i1 // (n_cu - g)) * c1))
        return best

    gain = 0.0
    for k_of in (lambda c: c.hidden_size, lambda c: c.mlp_dim):   # contraction depth of proj / fc2
        (ia, ca), (ib, cb) = [(items(t, t.cfg.hidden_size), k_of(t.cfg) / 64.0 + 6.0) for t in (ta, tb)]
        single = -(-ia // n_cu) * ca + -(-ib // n_cu) * cb
        gain += max(0.0, single - best_pair(ia, ca, ib, cb))
    return gain


def _tower_stream_plan(n: int):
    """CAMBRIAN_AMD_TOWER_STREAMS: "0" = every tower on the launch stream (None); "1" = one side stream per tower; else one
    character per tower in list order — equal characters share a side stream (and run one after the other on it), "0" keeps
    the tower on the launch stream, enqueued after the side streams' work."""
    v = _TOWER_STREAMS
    if v in ("", "0"):
        return None
    if v == "1":
        return [chr(ord("a") + i) for i in range(n)]
    if len(v) != n or set(v) == {"0"}:
        return None
    return list(v)


@dataclass
class SvaContext:
    """What the in-LLM SVA layers need (cambrian_llama.py:168-207), in the fused layout."""
    feats: List[torch.Tensor]                 # [B*T_i, C] tower-token-major, wrapped by shared_grad
    masks_u8: List[Optional[torch.Tensor]]    # uint8 [B*side*side, r_i*r_i]
    holders: List[ops.GradAccumulator]
    ctx_b: torch.Tensor                       # [B, C] global context, one row per image
    B: int
    side: int


def unmask_attention_mask(mask, original_size):
    """cambrian_arch.py:203-225 — zero the grid rows / columns that are padding of the squared (expand2square) image."""
    original_w, original_h = original_size
    cur_h, cur_w = mask.shape[1:3]
    if original_w / original_h > cur_w / cur_h:
        new_height = int(original_h * (cur_w / original_w))
        padding = (cur_h - new_height) // 2
        if padding > 0:
            mask[:, :padding, :] = 0
            mask[:, -padding:, :] = 0
    else:
        new_width = int(original_w * (cur_h / original_h))
        padding = (cur_w - new_width) // 2
        if padding > 0:
            mask[:, :, :padding] = 0
            mask[:, :, -padding:] = 0
    return mask


def unpad_image(tensor, original_size):
    """cambrian_arch.py:228-256 — crop dims (1, 2) of a grid tensor back to the original aspect ratio."""
    original_width, original_height = original_size
    current_height, current_width = tensor.shape[1:3]
    if original_width / original_height > current_width / current_height:
        new_height = int(original_height * (current_width / original_width))
        padding = (current_height - new_height) // 2
        return tensor[:, padding:current_height - padding, :]
    new_width = int(original_width * (current_height / original_height))
    padding = (current_width - new_width) // 2
    return tensor[:, :, padding:current_width - padding]


def _sva_modules(owner: nn.Module, config, vision_tower_aux_list, hidden_size: int):
    """Shared by __init__ (config-driven, cambrian_arch.py:41-79) and initialize_vision_modules (:142-169)."""
    vh = config.vision_hidden_size
    n_group = config.num_query_group
    query_num_list = config.query_num_list
    token_lens = config.mm_vision_tower_aux_token_len_list
    image_token_len = config.image_token_len
    owner.mm_projector = HipSequential(nn.Linear(vh * n_group, hidden_size), nn.GELU(), nn.Linear(hidden_size, hidden_size))
    for aux_i, tower in enumerate(vision_tower_aux_list):
        aux = HipSequential(nn.Linear(tower.hidden_size, vh), nn.GELU(), nn.Linear(vh, vh), nn.LayerNorm(vh))
        aux.fp8_heavy = True     # rows = every tower token: fp8 forward GEMMs under config.fp8_projections
        setattr(owner, f"mm_projector_aux_{aux_i}", aux)
    n_towers = len(vision_tower_aux_list)
    for g in range(n_group):
        sizes = [int(t ** 0.5) // int(query_num_list[g] ** 0.5) for t in token_lens]
        setattr(owner, f"vision_sampler_{g}", VisionTokenSampler(vh, vh, [vh] * n_towers, sizes, vh, config.connector_depth))
    if not config.connector_only:
        sizes = [int(t ** 0.5) // int(image_token_len ** 0.5) for t in token_lens]
        owner.vision_sampler_layers = nn.ModuleList(
            [VisionTokenSampler(hidden_size, vh, [vh] * n_towers, sizes, vh, 1)
             for _ in range(config.num_of_vision_sampler_layers)])


class CambrianMetaModel:
    """cambrian_arch.py:33-200."""

    def __init__(self, config):
        super(CambrianMetaModel, self).__init__(config)
        if hasattr(config, "mm_vision_tower_aux_list"):
            projector_type = getattr(config, "mm_projector_type", "linear")
            self.vision_tower_aux_list = build_vision_tower_aux_list(config, delay_load=True)
            if projector_type == "sva":
                _sva_modules(self, config, self.vision_tower_aux_list, config.hidden_size)
                self.vision_query = nn.Parameter(torch.randn((config.num_query_group, config.vision_hidden_size), dtype=self.dtype))
                self.image_newline = nn.Parameter(torch.empty(config.hidden_size, dtype=self.dtype))
            else:
                config.mm_hidden_size = sum(t.hidden_size for t in self.vision_tower_aux_list)
                self.mm_projector = build_vision_projector(config)
                self.image_newline = nn.Parameter(torch.empty(config.hidden_size, dtype=self.dtype))

    def get_vision_tower_aux_list(self):
        return getattr(self, "vision_tower_aux_list", None)

    def initialize_vision_modules(self, model_args, fsdp=None):
        cfg = self.config
        cfg.image_token_len = model_args.image_token_len
        cfg.num_query_group = model_args.num_query_group
        cfg.query_num_list = model_args.query_num_list
        assert model_args.num_query_group == len(model_args.query_num_list)
        cfg.connector_depth = model_args.connector_depth
        cfg.mm_vision_tower_aux_list = model_args.vision_tower_aux_list
        cfg.mm_vision_tower_aux_token_len_list = model_args.vision_tower_aux_token_len_list
        cfg.connector_only = model_args.connector_only

        if self.get_vision_tower_aux_list() is None:
            towers = build_vision_tower_aux_list(model_args)
            # frozen towers are deliberately NOT registered sub-modules (cambrian_arch.py:125-128)
            self.vision_tower_aux_list = nn.ModuleList(towers) if model_args.unfreeze_mm_vision_tower else towers
        else:
            towers = self.vision_tower_aux_list
            for t in towers:
                t.load_model()

        cfg.use_mm_proj = True
        cfg.mm_projector_type = getattr(model_args, "mm_projector_type", "linear")
        cfg.vision_hidden_size = model_args.vision_hidden_size
        cfg.mm_vision_select_layer = model_args.mm_vision_select_layer
        cfg.mm_vision_select_feature = model_args.mm_vision_select_feature

        if getattr(self, "mm_projector", None) is None:
            if cfg.mm_projector_type == "sva":
                if not cfg.connector_only:
                    cfg.num_of_vision_sampler_layers = model_args.num_of_vision_sampler_layers
                    cfg.start_of_vision_sampler_layers = model_args.start_of_vision_sampler_layers
                    cfg.stride_of_vision_sampler_layers = model_args.stride_of_vision_sampler_layers
                _sva_modules(self, cfg, towers, cfg.hidden_size)
                vision_embed_std = 1 / torch.sqrt(torch.tensor(cfg.vision_hidden_size, dtype=self.dtype))
                self.vision_query = nn.Parameter(
                    torch.randn((cfg.num_query_group, cfg.vision_hidden_size), dtype=self.dtype) * vision_embed_std)
            else:
                cfg.mm_hidden_size = sum(t.hidden_size for t in towers)
                self.mm_projector = build_vision_projector(cfg)
            embed_std = 1 / torch.sqrt(torch.tensor(cfg.hidden_size, dtype=self.dtype))
            self.image_newline = nn.Parameter(torch.randn(cfg.hidden_size, dtype=self.dtype) * embed_std)
        else:
            for p in self.mm_projector.parameters():  # in case it is frozen by LoRA
                p.requires_grad = True

        pretrain = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if pretrain is not None:
            weights = torch.load(pretrain, map_location="cpu")

            def get_w(w, keyword):
                return {k.split(keyword + ".")[1]: v for k, v in w.items() if keyword + "." in k}

            self.mm_projector.load_state_dict(get_w(weights, "mm_projector"), strict=True)
            if cfg.mm_projector_type == "sva":
                for aux_i in range(len(towers)):
                    getattr(self, f"mm_projector_aux_{aux_i}").load_state_dict(get_w(weights, f"mm_projector_aux_{aux_i}"), strict=True)
                for g in range(cfg.num_query_group):
                    getattr(self, f"vision_sampler_{g}").load_state_dict(get_w(weights, f"vision_sampler_{g}"), strict=True)
                if not cfg.connector_only:
                    self.vision_sampler_layers.load_state_dict(get_w(weights, "vision_sampler_layers"), strict=True)
                self.vision_query.data = weights["model.vision_query"]
            self.image_newline.data = weights["model.image_newline"]


class CambrianMetaForCausalLM(ABC):
    """cambrian_arch.py:259-653 (static path)."""

    @abstractmethod
    def get_model(self):
        pass

    def get_vision_tower_aux_list(self):
        return self.get_model().get_vision_tower_aux_list()

    def rearrange_vision_tower_features_train(self, vision_tower_aux_feature_list, vision_tower_aux_attention_masks_list,
                                              query_side_len):
        """cambrian_arch.py:271-287 — kept for API compatibility (``sva_fused=False``): window-major copies.
        The fused path never calls this; its consumers do the same gather by index arithmetic."""
        feats, masks = [], []
        bs = vision_tower_aux_feature_list[0].shape[0]
        for f, m in zip(vision_tower_aux_feature_list, vision_tower_aux_attention_masks_list):
            side = int(f.shape[1] ** 0.5)
            assert (side // query_side_len) * query_side_len == side
            r = side // query_side_len
            f = f.view(bs, query_side_len, r, query_side_len, r, -1).permute(0, 1, 3, 2, 4, 5).contiguous()
            feats.append(f.flatten(0, 2).flatten(1, 2))
            masks.append(m.view(bs * query_side_len * query_side_len, r * r))
        return feats, masks

    def rearrange_vision_tower_features_inference(self, vision_tower_aux_feature_list, query_side_len, image_sizes,
                                                  unpad=False):
        """cambrian_arch.py:289-330 (eval / generate branch): per-sample window-major KV lists and masks; with
        ``unpad`` the query grid is cropped to the image's aspect ratio, so samples contribute different numbers of
        queries.  Pure index work (views / copies), as in the reference."""
        feats_out, masks_out = [], []
        bs = vision_tower_aux_feature_list[0].shape[0]
        for f in vision_tower_aux_feature_list:
            side = int(f.shape[1] ** 0.5)
            assert (side // query_side_len) * query_side_len == side
            r = side // query_side_len
            fl, ml = [], []
            for b in range(bs):
                m = torch.ones((1, side, side), dtype=torch.bool, device=f.device)
                x = f[b].view(1, query_side_len, r, query_side_len, r, -1).permute(0, 1, 3, 2, 4, 5).contiguous()
                if unpad:
                    x = unpad_image(x, image_sizes[b])
                x = x.flatten(0, 2).flatten(1, 2)
                m = unmask_attention_mask(m, image_sizes[b])
                m = m.view(1, query_side_len, r, query_side_len, r).permute(0, 1, 3, 2, 4).contiguous()
                if unpad:
                    m = unpad_image(m, image_sizes[b])
                m = m.flatten(0, 2).flatten(1, 2)
                m[m.sum(-1) == 0] = True
                fl.append(x)
                ml.append(m)
            feats_out.append(torch.cat(fl, 0))
            masks_out.append(torch.cat(ml, 0))
        return feats_out, masks_out

    def encode_images(self, image_aux_list):
        """cambrian_arch.py:271-278.  Frozen towers are independent forward-only kernel chains; with
        CAMBRIAN_AMD_TOWER_STREAMS=1 each runs on its own HIP stream, so the last, partly filled round of one tower's GEMM
        tiles and its HBM-bound LayerNorm / depthwise-conv kernels can overlap another tower's MFMA work.  Measured
        (same-box A/B, 16 images): 1128.8 -> 1125.9 ms/step (-0.26 %) — the persistent GEMMs already occupy every CU, so
        little overlaps — while per-launch durations (and with them the per-kernel roofline) stop being comparable; it
        is therefore off by default."""
        towers = self.get_model().get_vision_tower_aux_list()
        frozen = all(not getattr(t, "unfreeze_mm_vision_tower", False) for t in towers)
        plan = _tower_stream_plan(len(towers))
        if (plan is None or len(towers) < 2 or not frozen or not image_aux_list[0].is_cuda
                or torch.is_grad_enabled() and any(x.requires_grad for x in image_aux_list)):
            return self._encode_images_one_stream(image_aux_list, towers, frozen)
        main = torch.cuda.current_stream()
        pool = getattr(self, "_tower_streams", None)
        if pool is None:
            pool = self._tower_streams = {}
        outs = [None] * len(towers)
        used = []
        for key in sorted(set(plan) - {"0"}):         # side streams first (each runs its towers in list order) ...
            st = pool.get(key)
            if st is None:
                st = pool[key] = torch.cuda.Stream()
            st.wait_stream(main)                      # the images (and whatever produced them) are ready
            with torch.cuda.stream(st):
                for i, k in enumerate(plan):
                    if k == key:
                        outs[i] = towers[i](image_aux_list[i])
                        outs[i].record_stream(main)   # allocated from st's pool, consumed on the main stream
            used.append(st)
        # ... then the towers that stay on the launch stream (the two ViT trunks that pair best among them in lock-step)
        self._encode_images_one_stream(image_aux_list, towers, frozen, only=[i for i, k in enumerate(plan) if k == "0"], outs=outs)
        for st in used:
            main.wait_stream(st)
        return outs

    def _encode_images_one_stream(self, image_aux_list, towers, frozen, only=None, outs=None):
        """Every tower on the launch stream.  Round 6: the two frozen ViT trunks whose blocks gain most from running side by
        side (DINOv2 and SigLIP at the release sizes: 1.62- and 1.35-round linears) advance in lock-step and hand their
        same-position residual linears to cmb_gemm_pair — one launch with the workgroups split between the two problems
        (vit.py::forward_paired).  The towers' own post-processing (interpolation, dtype) follows per tower as before;
        CAMBRIAN_AMD_PAIR_TOWERS=0 runs them one after the other."""
        from .multimodal_encoder.vit import ViTTrunk, forward_paired
        pair = None
        if (_PAIR_TOWERS and frozen and image_aux_list[0].is_cuda and not isinstance(image_aux_list[0], list)
                and image_aux_list[0].dtype == torch.bfloat16):
            cand = [i for i, t in enumerate(towers)
                    if (only is None or i in only) and isinstance(getattr(t, "vision_tower", None), ViTTrunk) and "trunk_out" in t._forward.__code__.co_varnames
                    and not isinstance(image_aux_list[i], list) and not t.vision_tower._ln_fused]
            best = 0.0
            for x in range(len(cand)):
                for y in range(x + 1, len(cand)):
                    i, j = cand[x], cand[y]
                    gain = _pair_rounds_gain(towers[i].vision_tower, towers[j].vision_tower, image_aux_list[i].shape[0])
                    if gain > best:
                        best, pair = gain, (i, j)
        outs = [None] * len(towers) if outs is None else outs
        for k, (image_aux, tower) in enumerate(zip(image_aux_list, towers)):
            if only is not None and k not in only:
                continue
            if pair is not None and k == pair[1]:
                continue                                  # produced together with pair[0]
            if pair is not None and k == pair[0]:
                i, j = pair
                ta, tb = towers[i], towers[j]
                xa, xb = image_aux_list[i].to(device=ta.device), image_aux_list[j].to(device=tb.device)
                sa, sb = forward_paired(ta.vision_tower, xa, tb.vision_tower, xb)
                outs[i] = ta._forward(image_aux_list[i], trunk_out=sa)
                outs[j] = tb._forward(image_aux_list[j], trunk_out=sb)
            else:
                outs[k] = tower(image_aux)
        return outs

    # ------------------------------------------------------------------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, image_aux_attention_masks_list=None, image_sizes=None):
        model = self.get_model()
        towers = model.get_vision_tower_aux_list()
        if towers is None or images is None or input_ids.shape[1] == 1:  # cambrian_arch.py:346-347
            return input_ids, position_ids, attention_mask, past_key_values, None, labels, None, None, None, None
        if not STATIC_PATH or getattr(self, "_dynamic_path", False):
            return self._prepare_inputs_dynamic(input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                                image_sizes)
        cfg = model.config
        if getattr(cfg, "tune_mm_mlp_adapter", False) and getattr(cfg, "mm_use_im_start_end", False):
            raise NotImplementedError  # cambrian_arch.py:454-455

        image_aux_list = images
        bs = image_aux_list[0].shape[0]
        dtype = image_aux_list[0].dtype
        image_token_len = cfg.image_token_len
        side = int(image_token_len ** 0.5)
        span = ops.region_begin("towers_connector")                                 # (bench.py roofline.region)
        if torch.is_grad_enabled():
            ops.weight_step_begin()    # bf16 copies of the trainable weights: one launch per step (closed by the model's forward)
        feats_raw = self.encode_images(image_aux_list)                              # :366

        sva_ctx = None
        if cfg.mm_projector_type == "sva":
            vh = cfg.vision_hidden_size
            feats, holders = [], []
            # every sampler that will normalise these features (connector groups, then the in-LLM layers): their position
            # tables are announced to shared_grad so that the 13 LayerNorm backwards of a tower run as one pass at the end
            samplers = [getattr(model, f"vision_sampler_{g}") for g in range(len(cfg.query_num_list))]
            samplers += list(getattr(model, "vision_sampler_layers", []) or [])
            for aux_i in range(len(towers)):                                         # :372-379
                f = feats_raw[aux_i]
                f = getattr(model, f"mm_projector_aux_{aux_i}")(f.to(dtype)).to(dtype)
                holders.append(ops.GradAccumulator())
                f2 = f.reshape(-1, vh)
                tables = [t for sm in samplers for t in sm.pos_tables(aux_i)]
                feats.append(ops.shared_grad(f2, holders[-1], tables) if f2.requires_grad else f2)
            T0 = feats_raw[0].shape[1]
            ctx_b = ops.token_mean(feats[0].view(bs, T0, vh), holders[0] if feats[0].requires_grad else None)  # [B, C] (:377)
            masks_u8 = self._masks_u8(image_aux_attention_masks_list, bs, side, feats)
            group_out = []
            for g, query_num in enumerate(cfg.query_num_list):                       # :382-402
                qside = int(query_num ** 0.5)
                # the reference re-views the collator's masks per group (`.view(bs * q * q, -1)`, :283), whatever q is
                masks_g = masks_u8 if qside == side else self._masks_u8(image_aux_attention_masks_list, bs, qside, feats)
                vq = model.vision_query[g].to(dtype)
                q2 = vq.view(1, vh).expand(bs * query_num, vh).contiguous()
                out = getattr(model, f"vision_sampler_{g}").forward_fused(q2, ctx_b, feats, masks_g, holders, bs, qside)
                if qside != side:                                                    # :395-401 (S5): fp32 bilinear to the
                    out = out.view(bs, qside, qside, -1).permute(0, 3, 1, 2)         # final grid, align_corners=False
                    out = torch.nn.functional.interpolate(out.float(), size=(side, side), mode="bilinear",
                                                          align_corners=False).to(dtype)
                    out = out.permute(0, 2, 3, 1).reshape(bs * side * side, -1)
                group_out.append(out)
            image_features = group_out[0] if len(group_out) == 1 else torch.cat(group_out, -1)
            sva_ctx = SvaContext(feats, masks_u8, holders, ctx_b, bs, side)
        else:
            image_features = torch.cat(feats_raw, -1).to(dtype)                      # :408-410
            image_features = image_features.reshape(-1, image_features.shape[-1])

        image_features = model.mm_projector(image_features).to(dtype)               # :411   [B*side*side, H]
        H = image_features.shape[-1]
        # newline column + splice into the token embeddings: one gather kernel (:413-420, :457-490)
        inputs_embeds, _pos = ops.embed_splice(input_ids, model.embed_tokens.weight, image_features.view(bs, side * side, H),
                                               model.image_newline, side, IMAGE_TOKEN_INDEX)
        inputs_embeds = ops.region_mark(inputs_embeds, span, "b0")  # its backward runs to the end of backward(): region_close
        ops.region_fwd_end(span)
        final_size = [(side, side)] * bs
        if sva_ctx is None:
            return None, position_ids, attention_mask, past_key_values, inputs_embeds, labels, None, None, final_size, None
        if getattr(cfg, "sva_fused", True):
            return (None, position_ids, attention_mask, past_key_values, inputs_embeds, labels,
                    sva_ctx, sva_ctx.masks_u8, final_size, sva_ctx.ctx_b)
        # reference-format lists (window-major) for an unmodified reference decoder wrapper (:404-406)
        f3 = [f.view(bs, -1, f.shape[-1]) for f in sva_ctx.feats]
        kv_final, m_final = self.rearrange_vision_tower_features_train(f3, image_aux_attention_masks_list, side)
        ctx_final = sva_ctx.ctx_b[:, None, None, :].expand(-1, side * side, 1, -1).flatten(0, 1)
        return None, position_ids, attention_mask, past_key_values, inputs_embeds, labels, kv_final, m_final, final_size, ctx_final

    # ------------------------------------------------------------------------------------------
    def _prepare_inputs_dynamic(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, image_sizes):
        """The eval / generate branch of prepare_inputs_labels_for_multimodal (``IS_XLA_AVAILABLE`` False in the reference;
        here ``STATIC_PATH = False`` or ``generate()``): cambrian_arch.py:366-402 with rearrange_..._inference, :422-451
        (per-sample unpad + newline; in-LLM KV lists with unpad=True), :492-609 (variable-length merge, padding).
        Returns the reference's 10-tuple with window-major KV lists / bool masks / concatenated context rows; the SVA
        layers then run through their reference calling convention (VisionCrossAttentionLayer.forward)."""
        model = self.get_model()
        cfg = model.config
        if getattr(cfg, "tune_mm_mlp_adapter", False) and getattr(cfg, "mm_use_im_start_end", False):
            raise NotImplementedError  # cambrian_arch.py:454-455
        towers = model.get_vision_tower_aux_list()
        bs = images[0].shape[0]
        dtype = images[0].dtype
        side = int(cfg.image_token_len ** 0.5)
        feats_raw = self.encode_images(images)
        kv_final = mask_final = ctx_final = None
        if cfg.mm_projector_type == "sva":
            feats = [getattr(model, f"mm_projector_aux_{i}")(feats_raw[i].to(dtype)).to(dtype) for i in range(len(towers))]
            ctx_b = ops.token_mean(feats[0])                                           # [B, C] (:377)
            ctx = ctx_b.view(bs, 1, 1, -1)
            group_out = []
            for g, query_num in enumerate(cfg.query_num_list):
                qside = int(query_num ** 0.5)
                q = model.vision_query[g].to(dtype).view(1, 1, 1, -1).expand(bs, query_num, -1, -1).flatten(0, 1)
                ctx_g = ctx.expand(-1, query_num, 1, -1).flatten(0, 1)
                kv, masks = self.rearrange_vision_tower_features_inference(feats, qside, image_sizes)
                out = getattr(model, f"vision_sampler_{g}")(q.contiguous(), ctx_g.contiguous(), *kv, *masks)
                out = out.view(bs, query_num, -1)
                if qside != side:                                                      # :395-401
                    out = out.permute(0, 2, 1).contiguous().view(bs, -1, qside, qside)
                    out = torch.nn.functional.interpolate(out.float(), size=(side, side), mode="bilinear",
                                                          align_corners=False).to(dtype)
                    out = out.permute(0, 2, 3, 1).contiguous().flatten(1, 2)
                group_out.append(out)
            image_features = torch.cat(group_out, -1)
            kv_final, mask_final = self.rearrange_vision_tower_features_inference(feats, side, image_sizes, unpad=True)
        else:
            image_features = torch.cat(feats_raw, -1).to(dtype)
        image_features = model.mm_projector(image_features).to(dtype).view(bs, side, side, -1)   # :411, :424
        vis, final_size, ctx_rows = [], [], []
        for b in range(bs):                                                                      # :431-447
            cur = unpad_image(image_features[b].unsqueeze(0), image_sizes[b])
            h, w = cur.shape[1:3]
            final_size.append((h, w))
            nl = model.image_newline.to(cur.dtype).view(1, 1, 1, -1).expand(1, h, 1, -1)
            vis.append(torch.cat((cur.reshape(1, h, w, -1), nl), dim=2).flatten(1, 2).squeeze(0))
            if kv_final is not None:
                ctx_rows.append(ctx[b].expand(h * w, 1, -1))
        if kv_final is not None:
            ctx_final = torch.cat(ctx_rows, 0)

        # ---- variable-length merge (:492-609) ----
        _labels, _position_ids, _attention_mask = labels, position_ids, attention_mask
        att = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
        if labels is None:
            labels = torch.full_like(input_ids, IGNORE_INDEX)
        embed = model.embed_tokens
        new_embeds, new_labels = [], []
        cur_image_idx = 0
        for b in range(bs):
            ids, lab = input_ids[b][att[b]], labels[b][att[b]]
            idx = torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist()
            if not idx:                                                                           # :519-526
                new_embeds.append(embed(ids).to(vis[0].dtype))
                new_labels.append(lab)
                cur_image_idx += 1
                continue
            bounds = [-1] + idx + [ids.shape[0]]
            pieces_e, pieces_l = [], []
            for i in range(len(bounds) - 1):
                seg = ids[bounds[i] + 1:bounds[i + 1]]
                pieces_e.append(embed(seg).to(vis[0].dtype))
                pieces_l.append(lab[bounds[i] + 1:bounds[i + 1]])
                if i < len(idx):
                    v = vis[cur_image_idx]
                    cur_image_idx += 1
                    pieces_e.append(v)
                    pieces_l.append(torch.full((v.shape[0],), IGNORE_INDEX, device=lab.device, dtype=lab.dtype))
            new_embeds.append(torch.cat(pieces_e))
            new_labels.append(torch.cat(pieces_l))
        max_model_len = getattr(cfg, "tokenizer_model_max_length", None)
        if max_model_len is not None:
            new_embeds = [x[:max_model_len] for x in new_embeds]
            new_labels = [x[:max_model_len] for x in new_labels]
        max_len = max(x.shape[0] for x in new_embeds)
        dev = new_embeds[0].device
        emb = torch.zeros((bs, max_len, new_embeds[0].shape[1]), dtype=new_embeds[0].dtype, device=dev)
        lab_pad = torch.full((bs, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype, device=dev)
        att_pad = torch.zeros((bs, max_len), dtype=torch.bool, device=dev)
        pos_pad = torch.zeros((bs, max_len), dtype=torch.long, device=dev)
        left = getattr(cfg, "tokenizer_padding_side", "right") == "left"
        for b, (e, l) in enumerate(zip(new_embeds, new_labels)):
            n = e.shape[0]
            sl = slice(max_len - n, max_len) if left else slice(0, n)
            emb[b, sl] = e
            if n > 0:
                lab_pad[b, sl] = l
                att_pad[b, sl] = True
                pos_pad[b, sl] = torch.arange(n, device=dev)
        out_labels = None if _labels is None else lab_pad
        out_att = None if _attention_mask is None else att_pad.to(_attention_mask.dtype)
        out_pos = None if _position_ids is None else pos_pad
        return (None, out_pos, out_att, past_key_values, emb, out_labels, kv_final, mask_final, final_size, ctx_final)

    @staticmethod
    def _masks_u8(mask_list, bs: int, side: int, feats: Sequence[torch.Tensor]):
        """Collator masks are bool [B, side*side, r*r] already window-major per query (train_fsdp.py:1127-1137,
        1164); the kernel reads them as uint8 [B*side*side, r*r].  None -> all keys visible."""
        if mask_list is None:
            return [None] * len(feats)
        out = []
        for m in mask_list:
            m2 = m.reshape(bs * side * side, -1).contiguous()
            out.append(m2.view(torch.uint8) if m2.dtype == torch.bool else m2.to(torch.uint8))
        return out

    def initialize_vision_tokenizer(self, model_args, tokenizer):
        """cambrian_arch.py:611-653.  The release configs use neither extra patch nor start/end tokens
        (scripts/cambrian/*.sh: --mm_use_im_start_end False --mm_use_im_patch_token False): nothing to resize."""
        if getattr(model_args, "mm_use_im_patch_token", False) or getattr(model_args, "mm_use_im_start_end", False):
            raise NotImplementedError("extra image tokens are not used by any Cambrian-1 release config")
