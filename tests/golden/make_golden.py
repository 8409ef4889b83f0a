"""Generates the golden fixtures in this directory by running the REFERENCE's own Python
(/root/reference, read-only) on seeded inputs.  Run in the build container only:

    python tests/golden/make_golden.py

The GPU box has no /root/reference; tests read the committed ``*.pt`` files.  Nothing here is imported
by the product.  Loader recipe: SURVEY.md §8c (vision_sampler.py imports as-is; cambrian_arch.py needs
stub packages because cambrian/__init__.py drags in timm/open_clip; the collator functions are exec'd
from their line range because train_fsdp.py imports torch_xla unconditionally).
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref_vision_sampler():
    spec = importlib.util.spec_from_file_location("ref_vision_sampler", f"{REF}/cambrian/model/vision_sampler.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_arch():
    """cambrian_arch.py with stub packages (no timm/open_clip/ezcolorlog needed)."""
    for name, path in [("cambrian", f"{REF}/cambrian"), ("cambrian.model", f"{REF}/cambrian/model"),
                       ("cambrian.model.multimodal_encoder", f"{REF}/cambrian/model/multimodal_encoder"),
                       ("cambrian.model.multimodal_projector", f"{REF}/cambrian/model/multimodal_projector"),
                       ("cambrian.model.language_model", f"{REF}/cambrian/model/language_model")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    if "ezcolorlog" not in sys.modules:
        ez = types.ModuleType("ezcolorlog")
        ez.root_logger = logging.getLogger("ref")
        sys.modules["ezcolorlog"] = ez
    if "cambrian.model.multimodal_encoder.builder" not in sys.modules:
        b = types.ModuleType("cambrian.model.multimodal_encoder.builder")
        b.build_vision_tower_aux_list = lambda cfg, **kw: cfg._fake_towers
        sys.modules["cambrian.model.multimodal_encoder.builder"] = b
    import cambrian.model.cambrian_arch as A  # noqa: E402
    A.IS_XLA_AVAILABLE = True  # static (training) path
    return A


def load_ref_collator():
    src = open(f"{REF}/cambrian/train/train_fsdp.py").read().split("\n")
    ns = {"torch": torch, "IMAGE_TOKEN_INDEX": -200, "IGNORE_INDEX": -100}
    exec("\n".join(src[1038:1165]), ns)  # get_padding_offset, prepare_image_info, prepare_multimodal_data
    return ns


# ------------------------------------------------------------------------------------------------
def golden_sva():
    vs = load_ref_vision_sampler()
    torch.manual_seed(1234)
    q_dim, ctx_dim, hidden = 96, 80, 64
    kv_dims, kv_sizes, layers, bq = [64, 64, 64], [1, 1, 2], 2, 8
    m = vs.VisionTokenSampler(q_dim, ctx_dim, kv_dims, kv_sizes, hidden, layers).float()
    with torch.no_grad():  # perturb LayerNorm affines so they are exercised
        for n, p in m.named_parameters():
            if n.endswith(".0.weight") or n.endswith("norm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith(".0.bias") or n.endswith("norm.bias"):
                p.add_(0.1 * torch.randn_like(p))
    q = torch.randn(bq, 1, q_dim, requires_grad=True)
    ctx = torch.randn(bq, 1, ctx_dim, requires_grad=True)
    kvs = [torch.randn(bq, s * s, d, requires_grad=True) for s, d in zip(kv_sizes, kv_dims)]
    masks = [torch.ones(bq, s * s, dtype=torch.bool) for s in kv_sizes]
    masks[2][1, 0] = False
    masks[2][3, 1:] = False
    masks[0][5, 0] = False
    out = m(q, ctx, *kvs, *masks)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    fx = {
        "cfg": dict(q_dim=q_dim, ctx_dim=ctx_dim, hidden=hidden, kv_dims=kv_dims, kv_sizes=kv_sizes, layers=layers),
        "state": {k: v.detach().clone() for k, v in m.state_dict().items()},
        "q": q.detach(), "ctx": ctx.detach(), "kvs": [k.detach() for k in kvs], "masks": masks,
        "out": out.detach(), "w": w,
        "dq": q.grad, "dctx": ctx.grad, "dkvs": [k.grad for k in kvs],
        "dparams": {n: p.grad.clone() for n, p in m.named_parameters()},
    }
    torch.save(fx, f"{OUT}/sva_small.pt")
    print("sva_small.pt", sum(v.numel() for v in fx["state"].values()), "params")


if __name__ == "__main__":
    which = sys.argv[1:] or ["sva"]
    for w in which:
        globals()[f"golden_{w}"]()
