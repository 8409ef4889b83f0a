"""Length- and modality-grouped sampling of the data-parallel partitioning ("global batch split by sample;
the reference's LengthGroupedSampler groups by modality/length per world_size x batch megabatch — keep, it is CPU-side";
``cambrian/train/cambrian_trainer.py:69-161,203-216``).

Host-side integer work: the index order is bit-identical to the reference's for the same ``torch`` generator
(tests/test_sampler.py replays fixtures produced by the reference's own functions).  What the order means for data
parallelism: the flat list is a sequence of megabatches of ``world_size * batch_size`` indices; inside a megabatch the
indices are dealt into ``world_size`` chunks of near-equal total length (longest first, always to the currently
shortest chunk), so rank r's per-step batch is chunk r of megabatch m — ``rank_batches`` below cuts exactly that.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence

import torch
from torch.utils.data import Sampler


def split_to_even_chunks(indices: Sequence[int], lengths: Sequence[int], num_chunks: int) -> List[List[int]]:
    """cambrian_trainer.py:69-88.  Ragged input falls back to a strided deal; otherwise greedy longest-processing-time:
    each index (callers pass them longest first) goes to the chunk with the smallest running length that is not yet
    full (first such chunk on ties)."""
    n = len(indices)
    if n % num_chunks != 0:
        return [list(indices[i::num_chunks]) for i in range(num_chunks)]
    cap = n // num_chunks
    chunks: List[List[int]] = [[] for _ in range(num_chunks)]
    load = [0.0] * num_chunks
    for idx in indices:
        target = min(range(num_chunks), key=lambda c: load[c])      # min() keeps the first of equal loads
        chunks[target].append(idx)
        load[target] += lengths[idx]
        if len(chunks[target]) == cap:
            load[target] = float("inf")
    return chunks


def get_length_grouped_indices(lengths: Sequence[int], batch_size: int, world_size: int, generator=None) -> List[int]:
    """cambrian_trainer.py:122-130: random permutation -> megabatches -> each sorted by length (descending, stable) and
    dealt into balanced per-rank chunks."""
    perm = torch.randperm(len(lengths), generator=generator).tolist()
    mega = world_size * batch_size
    out: List[int] = []
    for start in range(0, len(perm), mega):
        block = sorted(perm[start:start + mega], key=lambda i: lengths[i], reverse=True)
        for chunk in split_to_even_chunks(block, lengths, world_size):
            out.extend(chunk)
    return out


def get_modality_length_grouped_indices(lengths: Sequence[int], batch_size: int, world_size: int, generator=None) -> List[int]:
    """cambrian_trainer.py:91-119: positive lengths = multimodal samples, negative = language-only.  Each modality is
    length-grouped on its own (with the GLOBAL torch RNG, as the reference does: ``generator=None``), full megabatches of
    both are shuffled together with ``generator``, the two ragged tails form one last (sorted) megabatch."""
    if any(l == 0 for l in lengths):
        raise AssertionError("Should not have zero length.")
    if all(l > 0 for l in lengths) or all(l < 0 for l in lengths):
        return get_length_grouped_indices(lengths, batch_size, world_size, generator=generator)
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    mega = world_size * batch_size

    def grouped(pairs):
        order = get_length_grouped_indices([l for _, l in pairs], batch_size, world_size, generator=None)
        flat = [pairs[i][0] for i in order]
        return [flat[s:s + mega] for s in range(0, len(flat), mega)]

    mm_mega, lang_mega = grouped(mm), grouped(lang)
    tail = mm_mega[-1] + lang_mega[-1]
    full = mm_mega[:-1] + lang_mega[:-1]
    order = torch.randperm(len(full), generator=generator).tolist()
    out = [i for m in order for i in full[m]]
    if tail:
        out.extend(sorted(tail))
    return out


class LengthGroupedSampler(Sampler):
    """cambrian_trainer.py:133-161 (same constructor and iteration contract)."""

    def __init__(self, batch_size: int, world_size: int, lengths: Optional[List[int]] = None, generator=None,
                 group_by_modality: bool = False):
        if lengths is None:
            raise ValueError("Lengths must be provided.")
        self.batch_size, self.world_size, self.lengths = batch_size, world_size, lengths
        self.generator, self.group_by_modality = generator, group_by_modality

    def __len__(self) -> int:
        return len(self.lengths)

    def __iter__(self) -> Iterator[int]:
        fn = get_modality_length_grouped_indices if self.group_by_modality else get_length_grouped_indices
        return iter(fn(self.lengths, self.batch_size, self.world_size, generator=self.generator))


def rank_batches(indices: Sequence[int], rank: int, world_size: int, batch_size: int, drop_last: bool = True) -> List[List[int]]:
    """The per-step batches of one data-parallel rank: chunk ``rank`` of every megabatch of the sampler's order (the
    chunks are the length-balanced ones ``split_to_even_chunks`` produced).  A ragged last megabatch is dropped
    (``drop_last``) or — as ``torch.utils.data.DistributedSampler`` does — padded by wrapping around to the start of the
    order until it divides by ``world_size``, then cut into ``world_size`` equal contiguous chunks: every rank gets the
    same number (>= 1) of samples, so no rank ever sits out a step of the gradient collectives."""
    mega = world_size * batch_size
    out = []
    for start in range(0, len(indices), mega):
        block = list(indices[start:start + mega])
        if len(block) == mega:
            out.append(block[rank * batch_size:(rank + 1) * batch_size])
        elif not drop_last and block:
            per = -(-len(block) // world_size)
            pad = per * world_size - len(block)
            src = list(indices)
            block = block + [src[i % len(src)] for i in range(pad)]
            out.append(block[rank * per:(rank + 1) * per])
    return out
