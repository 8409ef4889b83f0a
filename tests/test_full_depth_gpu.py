"""`-m gpu`: FULL-DEPTH towers at the release dimensions against oracle/towers.py (VERDICT r2 "missing #5" / next #4a).

tests/test_release_dims_gpu.py compares one block of every tower; rounding accumulates with depth, so here the whole
trunks run: CLIP-L/14@336 (24 layers, hidden_states[-2] = 23 layers run), SigLIP-SO400M/14@384 (27 layers + final LN),
DINOv2-g/14@378 (40 layers, SwiGLU, LayerScale, final LN), ConvNeXt-XXL@1024 (depths 3/4/30/3, multi-stage resample to
96 x 96 x 5760).  B = 1, random-init canonical parameters (the oracle is pinned to the HF modules by
tests/test_oracle_golden.py), fp32 (exact-fp32 MFMA path) and bf16 (the benched dtype).  CPU oracle: 2-7 s per tower.

Tolerances (max-abs error / max-abs reference, conftest.rel_err; slope / L2 from conftest.fit_err): fp32 5e-4; bf16 6e-2
max-abs with |slope - 1| < 1e-2 and relative L2 < 2.5e-2 — what 23-40 layers of bf16 storage rounding give against an
fp32 oracle (observed values: DESIGN.md §3 table; CAMBRIAN_PARITY_LOG=<file> appends them as JSON lines)."""
import json
import os

import pytest
import torch

from conftest import fit_err, rel_err

pytestmark = pytest.mark.gpu

DTYPES = [("fp32", torch.float32), ("bf16", torch.bfloat16)]
MAX_TOL = {"fp32": 5e-4, "bf16": 6e-2}
SLOPE_TOL = {"fp32": 2e-4, "bf16": 1e-2}
L2_TOL = {"fp32": 2e-4, "bf16": 2.5e-2}


def _log(**row):
    path = os.environ.get("CAMBRIAN_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(row) + "\n")


def _full_vit(kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig
    if kind == "clip_l_336":      # clip_encoder.py:57-68: hidden_states[-2] of 24 layers
        return ViTConfig(image_size=336, patch_size=14, hidden_size=1024, num_layers=24, num_heads=16, mlp_dim=4096,
                         act="quick_gelu", ln_eps=1e-5, has_cls=True, pre_ln=True, final_ln=False, patch_bias=False,
                         run_layers=23)
    if kind == "so400m_384":      # siglip_encoder.py:97: the whole trunk incl. final norm
        return ViTConfig(image_size=384, patch_size=14, hidden_size=1152, num_layers=27, num_heads=16, mlp_dim=4304,
                         act="gelu", ln_eps=1e-6, has_cls=False, final_ln=True)
    if kind == "dinov2_g_378":    # dino_encoder.py:115-126: last_hidden_state of 40 layers
        return ViTConfig(image_size=378, patch_size=14, hidden_size=1536, num_layers=40, num_heads=24, mlp_dim=4096,
                         act="swiglu", ln_eps=1e-6, has_cls=True, final_ln=True, layerscale=True)
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["clip_l_336", "so400m_384", "dinov2_g_378"])
def test_full_depth_vit_towers_match_oracle(dev, kind):
    from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk, resample_tokens
    from oracle import towers as O
    cfg = _full_vit(kind)
    gen = torch.Generator().manual_seed(sum(map(ord, kind)) + 3)
    p = ViTTrunk.random_canonical(cfg, gen)
    img = torch.randn(1, 3, cfg.image_size, cfg.image_size, generator=gen)
    with torch.no_grad():
        ref = O.vit_forward(cfg, p, img)
        ref576 = ref if ref.shape[1] == 576 else O.interpolate_tokens(ref, 576)
    for name, dt in DTYPES:
        out = ViTTrunk(cfg, dt).load_canonical(p, dev)(img.to(dev))
        assert out.shape == ref.shape
        e, (sl, l2) = rel_err(out, ref), fit_err(out, ref)
        _log(test="full_depth_tower", tower=kind, dtype=name, layers=cfg.run_layers or cfg.num_layers, max_rel=e, slope_err=sl, l2=l2)
        assert e < MAX_TOL[name], (kind, name, e)
        assert sl < SLOPE_TOL[name] and l2 < L2_TOL[name], (kind, name, sl, l2)
        if out.shape[1] != 576:   # the wrappers' 27^2 -> 24^2 bilinear token resize
            assert rel_err(resample_tokens(out, 576, force_copy=True), ref576) < MAX_TOL[name]
        del out
        torch.cuda.empty_cache()


def test_full_depth_convnext_xxl_matches_oracle(dev):
    """depths (3, 4, 30, 3), dims (384, 768, 1536, 3072) at 1024 px: every stage map and the 9216 x 5760 multi-stage output
    (clip_convnext_encoder.py:99-144)."""
    from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig, ConvNeXtTrunk
    from oracle import towers as O
    cfg = ConvNeXtConfig(depths=(3, 4, 30, 3), dims=(384, 768, 1536, 3072), ln_eps=1e-5)
    gen = torch.Generator().manual_seed(4096)
    p = ConvNeXtTrunk.random_canonical(cfg, gen)
    img = torch.randn(1, 3, 1024, 1024, generator=gen)
    with torch.no_grad():
        refs = O.convnext_stages(cfg, p, img)
        ref = O.convnext_forward(cfg, p, img, 96, multi_stage=True)
    for name, dt in DTYPES:
        trunk = ConvNeXtTrunk(cfg, dt).load_canonical(p, dev)
        stages = trunk.forward_stages(img.to(dev))
        for i, (a, b, side) in enumerate(zip(stages, refs, (256, 128, 64, 32))):
            e = rel_err(a.permute(0, 3, 1, 2), b)
            _log(test="full_depth_tower", tower=f"convnext_xxl_stage{i}", dtype=name, blocks=sum(cfg.depths[:i + 1]), max_rel=e)
            assert a.shape[1] == side and e < MAX_TOL[name], (i, name, e)
        out = trunk(img.to(dev), 96, multi_stage=True)
        e, (sl, l2) = rel_err(out, ref), fit_err(out, ref)
        _log(test="full_depth_tower", tower="convnext_xxl_1024", dtype=name, blocks=40, max_rel=e, slope_err=sl, l2=l2)
        assert out.shape == (1, 9216, 5760) and e < MAX_TOL[name], (name, e)
        assert sl < SLOPE_TOL[name] and l2 < L2_TOL[name], (name, sl, l2)
        del trunk, stages, out
        torch.cuda.empty_cache()
