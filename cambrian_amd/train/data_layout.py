"""Static 2048-token batch layout of the training path — the input contract of the hot path
(``cambrian/train/train_fsdp.py``: get_padding_offset :1039-1055, prepare_image_info :1057-1085,
prepare_multimodal_data :1089-1165, DataCollatorForSupervisedDataset :1169-1236).  Host-side integer/bool
arithmetic; results are bit-identical to the reference's (tests/test_data_layout.py checks them against the
golden vectors produced by the reference's own functions).

Written as closed-form index arithmetic instead of the reference's slice assignments:
  letter-box offsets  : a w x h image padded to a square and mapped on an n x n token grid leaves
                        pad = (n - int(short * n / long)) // 2 empty tokens on both sides of the short axis;
  visual mask (n x n+1): valid(y) & (col == n  or  valid(x))   — the newline column of every valid row stays
                        visible (the reference's ``[-right-1:-1]`` slice, :1069);
  position ids         : cumsum(mask) - 1 + offset, text after the image continues at max+1;
  SVA masks            : the tower grid (n*r)^2 cut into r x r windows, window-major [n*n, r*r]; an all-False
                        window becomes all-True (:1133-1137) so no softmax row is empty.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch

from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX


def get_padding_offset(cur_size: Tuple[int, int], original_size: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """(left, right, top, bottom) in tokens."""
    cur_w, cur_h = cur_size
    ow, oh = original_size
    if ow / oh > cur_w / cur_h:      # wider than the grid: bars above and below
        pad = (cur_h - int(oh * (cur_w / ow))) // 2
        return 0, 0, pad, pad
    pad = (cur_w - int(ow * (cur_h / oh))) // 2
    return pad, pad, 0, 0


def _valid_axes(n: int, image_size) -> Tuple[torch.Tensor, torch.Tensor]:
    left, right, top, bottom = get_padding_offset((n, n), image_size)
    idx = torch.arange(n)
    # a non-positive pad leaves the axis untouched (the reference only edits when the offset is > 0)
    col_ok = (idx >= max(left, 0)) & (idx < n - max(right, 0))
    row_ok = (idx >= max(top, 0)) & (idx < n - max(bottom, 0))
    return row_ok, col_ok


def prepare_image_info(image_size, image_token_len: int, newline: bool = False):
    n = int(image_token_len ** 0.5)
    row_ok, col_ok = _valid_axes(n, image_size)
    if newline:
        col_ok = torch.cat([col_ok, torch.ones(1, dtype=torch.bool)])
    mask = (row_ok[:, None] & col_ok[None, :]).flatten()
    return mask, mask.cumsum(0) - 1


def sva_window_mask(image_size, base_side: int, aux_side: int) -> torch.Tensor:
    """bool [base_side^2, r^2] (r = aux_side // base_side), window-major."""
    r = aux_side // base_side
    row_ok, col_ok = _valid_axes(aux_side, image_size)
    m = (row_ok[:, None] & col_ok[None, :]).view(base_side, r, base_side, r).permute(0, 2, 1, 3).reshape(base_side ** 2, r * r)
    empty = ~m.any(dim=1)
    return m | empty[:, None]


def prepare_multimodal_data(input_ids, labels, attention_mask, image_sizes, image_token_len=576,
                            image_aux_token_len_list=(192 * 192,), max_length=2048):
    base = int(image_token_len ** 0.5)
    span = image_token_len + base
    aux_sides = [int(t ** 0.5) for t in image_aux_token_len_list]
    ids_o, lab_o, att_o, pos_o = [], [], [], []
    aux: List[List[torch.Tensor]] = [[] for _ in aux_sides]
    for b in range(len(input_ids)):
        ids, lab, att = input_ids[b], labels[b], attention_mask[b]
        hits = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten()
        assert hits.numel() == 1, hits.numel()
        p = int(hits[0])
        tail = ids.shape[0] - p - 1
        im_att, im_pos = prepare_image_info(image_sizes[b], image_token_len, newline=True)
        for a, aside in enumerate(aux_sides):
            assert aside >= base
            aux[a].append(sva_window_mask(image_sizes[b], base, aside))
        if bool(att[p]):
            vis_att, vis_pos = im_att.to(att.dtype), (im_pos + p).to(torch.long)
            nxt = int(vis_pos.max()) + 1
        else:
            vis_att = torch.zeros(span, dtype=att.dtype)
            vis_pos = torch.zeros(span, dtype=torch.long)
            nxt = p
        ids_o.append(torch.cat([ids[:p + 1], torch.zeros(span - 1, dtype=ids.dtype), ids[p + 1:]])[:max_length])
        lab_o.append(torch.cat([lab[:p], torch.full((span,), IGNORE_INDEX, dtype=lab.dtype), lab[p + 1:]])[:max_length])
        att_o.append(torch.cat([att[:p], vis_att, att[p + 1:]])[:max_length])
        pos_o.append(torch.cat([torch.arange(p), vis_pos, torch.arange(nxt, nxt + tail)])[:max_length])
    return (torch.stack(ids_o), torch.stack(lab_o), torch.stack(att_o), torch.stack(pos_o),
            [torch.stack(m) for m in aux])


@dataclass
class DataCollatorForSupervisedDataset:
    """train_fsdp.py:1169-1236."""
    tokenizer: object
    image_token_len: int
    image_aux_token_len_list: list
    image_position: int

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        max_length = self.tokenizer.model_max_length
        pad_id = self.tokenizer.pad_token_id
        left = self.tokenizer.padding_side == "left"

        def fit(t, value):
            if t.shape[0] >= max_length:
                return t[:max_length]
            padw = (max_length - t.shape[0], 0) if left else (0, max_length - t.shape[0])
            return torch.nn.functional.pad(t, padw, "constant", value)

        input_ids = torch.stack([fit(i["input_ids"], pad_id) for i in instances])
        labels = torch.stack([fit(i["labels"], IGNORE_INDEX) for i in instances])
        attention_mask = input_ids.ne(pad_id)
        ip = self.image_position
        for r in range(len(input_ids)):  # rows without an image get a dummy (masked) one at image_position (:1201-1217)
            if (input_ids[r] == IMAGE_TOKEN_INDEX).sum() == 0:
                for t, fill in ((input_ids, IMAGE_TOKEN_INDEX), (labels, IGNORE_INDEX), (attention_mask, False)):
                    row = t[r].clone()
                    row[ip + 1:] = t[r, ip:-1]
                    row[ip] = fill
                    t[r] = row
        image_sizes = [i["image_size"] for i in instances]
        ids, lab, att, pos, aux = prepare_multimodal_data(input_ids, labels, attention_mask, image_sizes,
                                                          self.image_token_len, self.image_aux_token_len_list, max_length)
        batch = dict(input_ids=ids, labels=lab, attention_mask=att, position_ids=pos, image_aux_attention_masks_list=aux)
        if "image_raw" in instances[0]:
            # step-boundary route: the dataset only decodes; the uint8 pixels travel as they are
            # and train/image_pipeline.py produces the per-tower tensors on the GPU (DevicePrefetcher)
            batch["raw_images"] = [i["image_raw"] for i in instances]
            batch["image_sizes"] = image_sizes
        elif "image_aux_list" in instances[0]:
            per_tower = [list(x) for x in zip(*[i["image_aux_list"] for i in instances])]
            if all(x is not None and x.shape == per_tower[0][0].shape for x in per_tower[0]):
                batch["images"] = [torch.stack(x) for x in per_tower]
            else:
                batch["images"] = per_tower
        return batch


def synthetic_batch(batch_size: int, seq_len: int = 2048, image_position: int = 91, image_token_len: int = 576,
                    aux_token_lens: Sequence[int] = (576, 576, 576, 9216), image_res: Sequence[int] = (384, 336, 378, 1024),
                    image_sizes=None, seed: int = 1234, vocab_lo: int = 1000, vocab_hi: int = 30000) -> Dict[str, object]:
    """SURVEY.md §8d synthetic inputs: post-normalisation pixels ~ N(0,1) per tower (seed 1234+i), ids uniform in
    [1000, 30000) with the image token at ``image_position``, labels masked over the prompt and the visual span."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(vocab_lo, vocab_hi, (batch_size, seq_len), generator=g)
    ids[:, image_position] = IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, :image_position + 1] = IGNORE_INDEX
    att = torch.ones(batch_size, seq_len, dtype=torch.bool)
    sizes = image_sizes or [(336, 336)] * batch_size
    ids2, lab2, att2, pos2, aux = prepare_multimodal_data(ids, labels, att, sizes, image_token_len, list(aux_token_lens), seq_len)
    images = [torch.randn(batch_size, 3, r, r, generator=torch.Generator().manual_seed(seed + i)) for i, r in enumerate(image_res)]
    return dict(input_ids=ids2, labels=lab2, attention_mask=att2, position_ids=pos2, image_aux_attention_masks_list=aux,
                images=images, image_sizes=sizes)
