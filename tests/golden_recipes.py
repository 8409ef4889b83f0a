"""Deterministic parameter recipes shared by tests/golden/make_golden.py (which runs the REFERENCE's modules in the build
container) and the replay tests (which run on machines without /root/reference).  A fixture that would need tens of MB of
reference weights stores none: both sides fill the same state-dict names from per-name seeded CPU generators."""
import zlib

import torch


def fill_state(state: dict, seed: int) -> dict:
    """New values for every tensor of a VisionTokenSampler state dict (reference key names, vision_sampler.py:170-175,
    254-267): linear weights N(0, 1/fan_in), LayerNorm weights 1 + 0.1 N(0,1), biases 0.1 N(0,1), pos_embed N(0,1)."""
    out = {}
    for name in sorted(state):
        t = state[name]
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        r = torch.randn(t.shape, generator=g, dtype=torch.float32)
        if "pos_embed" in name:
            v = r
        elif t.dim() == 2:
            v = r / (t.shape[1] ** 0.5)
        elif name.endswith(".weight"):   # LayerNorm scale
            v = 1.0 + 0.1 * r
        else:                            # LayerNorm shift
            v = 0.1 * r
        out[name] = v.to(t.dtype)
    return out


def grad_summary(g: torch.Tensor, name: str, seed: int) -> torch.Tensor:
    """[sum, l2 norm, three seeded random projections] of a gradient tensor — what the fixture keeps per parameter."""
    g = g.detach().double().flatten().cpu()
    gen = torch.Generator().manual_seed((seed * 7919 + zlib.crc32(name.encode())) % (2 ** 31))
    proj = [(g * torch.randn(g.numel(), generator=gen, dtype=torch.float64)).sum() for _ in range(3)]
    return torch.stack([g.sum(), g.norm(), *proj]).float()


def sva_k1024_inputs(q_dim: int, hidden: int, kv_sizes, qside: int, B: int, seed: int):
    """Seeded inputs of the sva_k1024 fixture (not stored in it): q, ctx, window-major kv tensors, bool masks (every row
    keeps at least one key, as the collator guarantees, train_fsdp.py:1133-1137), output cotangent."""
    g = torch.Generator().manual_seed(seed)
    bq = B * qside * qside
    q = torch.randn(bq, 1, q_dim, generator=g)
    ctx = torch.randn(bq, 1, hidden, generator=g)
    kvs = [torch.randn(bq, s * s, hidden, generator=g) for s in kv_sizes]
    masks = [torch.rand(bq, s * s, generator=g) > 0.3 for s in kv_sizes]
    for mk in masks:
        mk[mk.sum(1) == 0] = True
    w = torch.randn(bq, 1, q_dim, generator=g)
    return q, ctx, kvs, masks, w
