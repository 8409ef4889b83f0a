"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the static ("XLA"/training) branch of the multimodal glue:
``cambrian/model/cambrian_arch.py`` — rearrange_vision_tower_features_train (:271-287),
prepare_inputs_labels_for_multimodal (:340-490, IS_XLA_AVAILABLE branch) — and of the collator's layout
arithmetic ``cambrian/train/train_fsdp.py`` get_padding_offset (:1039-1055), prepare_image_info (:1057-1085),
prepare_multimodal_data (:1089-1165).

Pinned against the reference itself: ``tests/golden/make_golden.py`` runs the real cambrian_arch.py (stub-package
loader, fake towers) and the real collator functions (exec'd from their line range) on seeded inputs and stores
``tests/golden/arch_small.pt`` / ``collator_cases.pt``; ``tests/test_oracle_golden.py`` replays them here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import sva as O

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


# ------------------------------------------------------------------------------------------------ glue
def window_rearrange(feat: torch.Tensor, query_side: int) -> torch.Tensor:
    """cambrian_arch.py:276-281: [B,(q*r)^2,C] -> [B*q*q, r*r, C] (window-major)."""
    bs, n, c = feat.shape
    side = int(n ** 0.5)
    assert (side // query_side) * query_side == side
    r = side // query_side
    x = feat.view(bs, query_side, r, query_side, r, c).permute(0, 1, 3, 2, 4, 5).contiguous()
    return x.flatten(0, 2).flatten(1, 2)


def mlp_projector(p: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, layer_norm: bool) -> torch.Tensor:
    """nn.Sequential(Linear, GELU, Linear[, LayerNorm]) of cambrian_arch.py:49,56."""
    h = O.gelu_erf(x @ p[prefix + "0.weight"].T + p[prefix + "0.bias"])
    y = h @ p[prefix + "2.weight"].T + p[prefix + "2.bias"]
    if layer_norm:
        y = O.layer_norm(y, p[prefix + "3.weight"], p[prefix + "3.bias"])
    return y


def prepare_inputs_static(p: Dict[str, torch.Tensor], cfg, input_ids: torch.Tensor, tower_feats: Sequence[torch.Tensor],
                          aux_masks: Sequence[torch.Tensor], embed_table: torch.Tensor):
    """The SVA static branch of prepare_inputs_labels_for_multimodal.  ``p`` uses the model-level key names
    (mm_projector.*, mm_projector_aux_{i}.*, vision_sampler_{g}.*, vision_query, image_newline).
    tower_feats[i]: [B,T_i,h_i] tower outputs; aux_masks[i]: bool [B, side*side, r_i*r_i] from the collator.
    Returns (inputs_embeds [B,S,H], kv_final list (window-major), mask_final list, ctx_final)."""
    bs = tower_feats[0].shape[0]
    side = int(cfg.image_token_len ** 0.5)
    if getattr(cfg, "mm_projector_type", "sva") != "sva":
        # :407-411 non-SVA branch (BASELINE configs[0]): channel-concat of the tower features -> mlpNx_gelu projector
        # (multimodal_projector/builder.py:60-67, nn.Sequential keys 0, 2, ...), no in-LLM lists
        img = torch.cat(list(tower_feats), -1)
        depth = int(cfg.mm_projector_type[len("mlp")])
        for d in range(depth):
            if d:
                img = O.gelu_erf(img)
            img = img @ p[f"mm_projector.{2 * d}.weight"].T + p[f"mm_projector.{2 * d}.bias"]
        return _newline_and_splice(p, img, bs, side, input_ids, embed_table), None, None, None
    feats = [mlp_projector(p, f"mm_projector_aux_{i}.", f, True) for i, f in enumerate(tower_feats)]   # :372-379
    ctx = feats[0].mean(1).view(bs, 1, 1, -1)                                                        # :377
    finals = []
    for g, query_num in enumerate(cfg.query_num_list):                                               # :382-402
        q = p["vision_query"][g].view(1, 1, 1, -1).expand(bs, query_num, -1, -1).flatten(0, 1)
        ctx_g = ctx.expand(-1, query_num, 1, -1).flatten(0, 1)
        qs = int(query_num ** 0.5)
        kv = [window_rearrange(f, qs) for f in feats]
        masks = [m.view(bs * qs * qs, -1) for m in aux_masks]
        out = O.vision_token_sampler(p, q, ctx_g, kv, masks, prefix=f"vision_sampler_{g}.").view(bs, query_num, -1)
        if qs != side:
            out = out.permute(0, 2, 1).contiguous().view(bs, -1, qs, qs)
            out = F.interpolate(out.float(), size=(side, side), mode="bilinear", align_corners=False)
            out = out.permute(0, 2, 3, 1).contiguous().flatten(1, 2)
        finals.append(out)
    kv_final = [window_rearrange(f, side) for f in feats]                                            # :404-406
    mask_final = [m.view(bs * side * side, -1) for m in aux_masks]
    ctx_final = ctx.expand(-1, side * side, 1, -1).flatten(0, 1)
    img = torch.cat(finals, -1)                                                                      # :410
    img = mlp_projector(p, "mm_projector.", img, False)                                              # :411
    return _newline_and_splice(p, img, bs, side, input_ids, embed_table), kv_final, mask_final, ctx_final


def _newline_and_splice(p, img, bs, side, input_ids, embed_table):
    img = img.view(bs, side, side, -1)                                                               # :413-420
    img = torch.cat([img, p["image_newline"][None, None, None, :].expand(bs, side, 1, -1)], dim=2).flatten(1, 2)
    emb = embed_table[torch.where(input_ids == IMAGE_TOKEN_INDEX, 0, input_ids)]                    # :460-461
    rows = []
    for b in range(bs):                                                                              # :465-487
        idx = torch.where(input_ids[b] == IMAGE_TOKEN_INDEX)[0].tolist()
        if not idx:
            rows.append(emb[b])
            continue
        assert len(idx) == 1, "the collator guarantees exactly one image token per row (train_fsdp.py:1100-1101)"
        pos = idx[0]
        rows.append(torch.cat([emb[b, :pos], img[b], emb[b, pos + img.shape[1]:]]))
    return torch.stack(rows)


# -------------------------------------------------------------------------------------------- collator
def get_padding_offset(cur_size: Tuple[int, int], original_size: Tuple[int, int]):
    """train_fsdp.py:1039-1055: (left, right, top, bottom) padding in tokens of an image letter-boxed to a square."""
    cur_w, cur_h = cur_size
    ow, oh = original_size
    if ow / oh > cur_w / cur_h:
        new_h = int(oh * (cur_w / ow))
        pad = (cur_h - new_h) // 2
        return 0, 0, pad, pad
    new_w = int(ow * (cur_h / oh))
    pad = (cur_w - new_w) // 2
    return pad, pad, 0, 0


def prepare_image_info(image_size, image_token_len: int, newline: bool = False):
    """train_fsdp.py:1057-1085, including the ``-right_offset-1:-1`` newline-column quirk (:1069)."""
    n = int(image_token_len ** 0.5)
    mask = torch.ones(n, n + 1 if newline else n, dtype=torch.bool)
    left, right, top, bottom = get_padding_offset((n, n), image_size)
    if left > 0:
        mask[:, :left] = 0
    if right > 0:
        if newline:
            mask[:, -right - 1:-1] = 0
        else:
            mask[:, -right:] = 0
    if top > 0:
        mask[:top, :] = 0
    if bottom > 0:
        mask[-bottom:, :] = 0
    mask = mask.flatten()
    return mask, mask.cumsum(0) - 1


def prepare_multimodal_data(input_ids, labels, attention_mask, image_sizes, image_token_len=576,
                            image_aux_token_len_list=(192 * 192,), max_length=2048):
    """train_fsdp.py:1089-1165: static layout (image token + 599 zeros), labels, attention mask, position ids and
    the per-tower window-major SVA masks [B, side*side, r*r] (all-False windows forced all-True, :1133-1137)."""
    out_ids, out_lab, out_att, out_pos = [], [], [], []
    aux_masks: List[List[torch.Tensor]] = [[] for _ in image_aux_token_len_list]
    base = int(image_token_len ** 0.5)
    aux_sides = [int(t ** 0.5) for t in image_aux_token_len_list]
    span = image_token_len + base
    for b, ids in enumerate(input_ids):
        where = torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist()
        assert len(where) == 1, len(where)
        bounds = [-1] + where + [ids.shape[0]]
        lab, att = labels[b], attention_mask[b]
        c_ids, c_lab, c_att, c_pos = [], [], [], []
        index = 0
        for i in range(len(bounds) - 1):
            lo, hi = bounds[i] + 1, bounds[i + 1]
            c_ids.append(ids[lo:hi + 1])          # keeps the image token itself (:1112)
            c_lab.append(lab[lo:hi])
            c_att.append(att[lo:hi])
            c_pos.append(torch.arange(index, index + hi - lo, dtype=torch.long))
            index += hi - lo
            if i < len(bounds) - 2:
                c_ids.append(torch.full((span - 1,), 0, dtype=ids.dtype))
                c_lab.append(torch.full((span,), IGNORE_INDEX, dtype=lab.dtype))
                im_att, im_pos = prepare_image_info(image_sizes[b], image_token_len, newline=True)
                for a, aside in enumerate(aux_sides):
                    assert aside >= base
                    r = aside // base
                    m, _ = prepare_image_info(image_sizes[b], aside * aside)
                    m = m.view(base, r, base, r).permute(0, 2, 1, 3).contiguous().flatten(0, 1).flatten(1, 2)
                    m[m.sum(dim=1) == 0] = True
                    aux_masks[a].append(m)
                im_pos = im_pos + index
                if att[hi]:
                    c_att.append(im_att)
                    c_pos.append(im_pos.to(torch.long))
                    index = im_pos.max() + 1
                else:
                    c_att.append(torch.full((span,), 0, dtype=att.dtype))
                    c_pos.append(torch.full((span,), 0, dtype=torch.long))
        out_ids.append(torch.cat(c_ids)[:max_length])
        out_lab.append(torch.cat(c_lab)[:max_length])
        out_att.append(torch.cat(c_att)[:max_length])
        out_pos.append(torch.cat(c_pos)[:max_length])
    return (torch.stack(out_ids), torch.stack(out_lab), torch.stack(out_att), torch.stack(out_pos),
            [torch.stack(m) for m in aux_masks])


# -------------------------------------------------------------------------------------------- dynamic (eval) branch
def unmask_attention_mask(mask: torch.Tensor, original_size):
    """cambrian_arch.py:203-225: zero the rows / columns of a [1,h,w] grid mask that are padding of the squared image."""
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


def unpad_image(tensor: torch.Tensor, original_size):
    """cambrian_arch.py:228-256: crop a [C?,H,W,...] grid (dims 1, 2) back to the image's aspect ratio."""
    original_width, original_height = original_size
    current_height, current_width = tensor.shape[1:3]
    if original_width / original_height > current_width / current_height:
        new_height = int(original_height * (current_width / original_width))
        padding = (current_height - new_height) // 2
        return tensor[:, padding:current_height - padding, :]
    new_width = int(original_width * (current_height / original_height))
    padding = (current_width - new_width) // 2
    return tensor[:, :, padding:current_width - padding]


def rearrange_inference(feats: Sequence[torch.Tensor], query_side: int, image_sizes, unpad: bool = False):
    """cambrian_arch.py:289-330."""
    out_f, out_m = [], []
    bs = feats[0].shape[0]
    for f in feats:
        side = int(f.shape[1] ** 0.5)
        assert (side // query_side) * query_side == side
        r = side // query_side
        fl, ml = [], []
        for b in range(bs):
            m = torch.ones((1, side, side), dtype=torch.bool)
            x = f[b].view(1, query_side, r, query_side, r, -1).permute(0, 1, 3, 2, 4, 5).contiguous()
            if unpad:
                x = unpad_image(x, image_sizes[b])
            x = x.flatten(0, 2).flatten(1, 2)
            m = unmask_attention_mask(m, image_sizes[b])
            m = m.view(1, query_side, r, query_side, r).permute(0, 1, 3, 2, 4).contiguous()
            if unpad:
                m = unpad_image(m, image_sizes[b])
            m = m.flatten(0, 2).flatten(1, 2)
            m[m.sum(-1) == 0] = True
            fl.append(x)
            ml.append(m)
        out_f.append(torch.cat(fl, 0))
        out_m.append(torch.cat(ml, 0))
    return out_f, out_m


def prepare_inputs_dynamic(p: Dict[str, torch.Tensor], cfg, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                           tower_feats: Sequence[torch.Tensor], image_sizes, embed_table: torch.Tensor,
                           padding_side: str = "right", max_length: Optional[int] = None):
    """The eval / generate branch (IS_XLA_AVAILABLE False) of prepare_inputs_labels_for_multimodal for the SVA
    projector, labels = None, position_ids = None: cambrian_arch.py:366-402 (towers -> aux projectors -> connector SVA on
    rearrange_inference lists), :422-451 (per-sample unpad + newline, in-LLM KV lists with unpad=True), :492-609 (drop
    padding, splice the visual rows at the image token, right/left pad to the longest row).
    Returns (inputs_embeds [B,L,H], attention_mask | None, kv_final, mask_final, final_size, ctx_final)."""
    bs = tower_feats[0].shape[0]
    side = int(cfg.image_token_len ** 0.5)
    feats = [mlp_projector(p, f"mm_projector_aux_{i}.", f, True) for i, f in enumerate(tower_feats)]
    ctx = feats[0].mean(1).view(bs, 1, 1, -1)
    finals = []
    for g, query_num in enumerate(cfg.query_num_list):
        q = p["vision_query"][g].view(1, 1, 1, -1).expand(bs, query_num, -1, -1).flatten(0, 1)
        ctx_g = ctx.expand(-1, query_num, 1, -1).flatten(0, 1)
        qs = int(query_num ** 0.5)
        kv, masks = rearrange_inference(feats, qs, image_sizes)
        out = O.vision_token_sampler(p, q, ctx_g, kv, masks, prefix=f"vision_sampler_{g}.").view(bs, query_num, -1)
        if qs != side:
            out = out.permute(0, 2, 1).contiguous().view(bs, -1, qs, qs)
            out = F.interpolate(out.float(), size=(side, side), mode="bilinear", align_corners=False)
            out = out.permute(0, 2, 3, 1).contiguous().flatten(1, 2)
        finals.append(out)
    img = mlp_projector(p, "mm_projector.", torch.cat(finals, -1), False).view(bs, side, side, -1)
    kv_final, mask_final = rearrange_inference(feats, side, image_sizes, unpad=True)
    vis, final_size, ctx_final = [], [], []
    for b in range(bs):
        cur = unpad_image(img[b].unsqueeze(0), image_sizes[b])
        h, w = cur.shape[1:3]
        final_size.append((h, w))
        cur = torch.cat((cur.view(1, h, w, -1), p["image_newline"].view(1, 1, 1, -1).expand(1, h, 1, -1)), dim=2)
        vis.append(cur.flatten(1, 2).squeeze(0))
        ctx_final.append(ctx[b].expand(h * w, 1, -1))
    ctx_final = torch.cat(ctx_final, 0)
    att = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    rows = []
    for b in range(bs):
        ids = input_ids[b][att[b]]
        idx = torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist()
        if not idx:
            rows.append(embed_table[ids])
            continue
        assert len(idx) == 1, "one image per sample (the reference consumes image_features[cur_image_idx] per image token)"
        rows.append(torch.cat([embed_table[ids[:idx[0]]], vis[b], embed_table[ids[idx[0] + 1:]]]))
    if max_length is not None:
        rows = [r[:max_length] for r in rows]
    max_len = max(r.shape[0] for r in rows)
    emb = torch.zeros(bs, max_len, rows[0].shape[1], dtype=rows[0].dtype)
    new_att = torch.zeros(bs, max_len, dtype=torch.bool)
    pos = torch.zeros(bs, max_len, dtype=torch.long)
    for b, r in enumerate(rows):
        n = r.shape[0]
        if padding_side == "left":
            emb[b, max_len - n:] = r
            new_att[b, max_len - n:] = True
            pos[b, max_len - n:] = torch.arange(n)
        else:
            emb[b, :n] = r
            new_att[b, :n] = True
            pos[b, :n] = torch.arange(n)
    return emb, (None if attention_mask is None else new_att.to(attention_mask.dtype)), kv_final, mask_final, final_size, ctx_final
