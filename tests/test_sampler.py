"""CPU: length / modality grouped sampling (SURVEY.md §8e partitioning; cambrian_trainer.py:69-161) — index order
bit-identical to fixtures produced by the reference's own functions, and the per-rank batch cut."""
import os

import pytest
import torch

from cambrian_amd.train import sampler as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_cases.pt")


def test_index_order_matches_reference():
    fx = torch.load(GOLD, weights_only=False)
    for c in fx["cases"]:
        torch.manual_seed(c["global_seed"])
        gen = torch.Generator().manual_seed(c["gen_seed"])
        smp = S.LengthGroupedSampler(c["batch_size"], c["world_size"], lengths=c["lengths"], generator=gen,
                                     group_by_modality=c["mixed"])
        got = list(iter(smp))
        assert got == c["indices"], (len(c["lengths"]), c["batch_size"], c["world_size"], c["mixed"])
        assert sorted(got) == list(range(len(c["lengths"]))) and len(smp) == len(got)
    for c in fx["chunks"]:
        assert S.split_to_even_chunks(c["indices"], c["lengths"], c["k"]) == c["out"]


def test_rank_batches_partition_every_megabatch():
    lengths = torch.randint(1, 500, (64,), generator=torch.Generator().manual_seed(0)).tolist()
    order = S.get_length_grouped_indices(lengths, 4, 4, generator=torch.Generator().manual_seed(1))
    per_rank = [S.rank_batches(order, r, 4, 4) for r in range(4)]
    assert all(len(b) == 4 for b in per_rank) and all(len(x) == 4 for b in per_rank for x in b)
    for step in range(4):          # the four ranks' batches of a step are exactly one megabatch, disjoint
        union = [i for r in range(4) for i in per_rank[r][step]]
        assert sorted(union) == sorted(order[step * 16:(step + 1) * 16])
        loads = [sum(lengths[i] for i in per_rank[r][step]) for r in range(4)]
        assert max(loads) - min(loads) <= max(lengths)          # greedy LPT balance
    assert S.rank_batches(list(range(10)), 1, 2, 4) == [[4, 5, 6, 7]]
    # ragged tail kept: padded by wrap-around to a multiple of the world size, every rank gets an equal, non-empty share
    assert S.rank_batches(list(range(10)), 0, 2, 4, drop_last=False) == [[0, 1, 2, 3], [8]]
    assert S.rank_batches(list(range(10)), 1, 2, 4, drop_last=False) == [[4, 5, 6, 7], [9]]
    tails = [S.rank_batches(list(range(17)), r, 4, 4, drop_last=False)[-1] for r in range(4)]   # one sample left for four ranks
    assert tails == [[16], [0], [1], [2]]
    with pytest.raises(ValueError):
        S.LengthGroupedSampler(2, 2)
    with pytest.raises(AssertionError):
        S.get_modality_length_grouped_indices([3, 0, -2], 1, 1)
