"""CPU: the round arithmetic that decides which two frozen ViT trunks encode_images runs in lock-step (cambrian_arch.py
``_pair_rounds_gain``, mirrored per call by cmb_gemm_pair in gemm.hip / gemm_p5.hip): at the release sizes and 24 images DINOv2 +
SigLIP gain (414 + 345 tiles = 1.62 + 1.35 rounds of 256 workgroups each, 3.0 + 2.9 rounds side by side), pairs with CLIP do not
beat it, and the generator form of ViTTrunk.forward hands out exactly two residual linears per block."""
import torch

from cambrian_amd.model.cambrian_arch import _pair_rounds_gain
from cambrian_amd.model.multimodal_encoder.vit import ViTConfig


class _T:
    def __init__(self, **kw):
        self.cfg = ViTConfig(**kw)


DINO = _T(image_size=378, patch_size=14, hidden_size=1536, num_layers=40, num_heads=24, mlp_dim=4096, act="swiglu")
SIG = _T(image_size=384, patch_size=14, hidden_size=1152, num_layers=27, num_heads=16, mlp_dim=4304, has_cls=False)
CLIP = _T(image_size=336, patch_size=14, hidden_size=1024, num_layers=24, num_heads=16, mlp_dim=4096)


def test_release_towers_pair_dinov2_with_siglip():
    g = {name: _pair_rounds_gain(a, b, 24) for name, a, b in (("ds", DINO, SIG), ("dc", DINO, CLIP), ("sc", SIG, CLIP))}
    assert g["ds"] > 50 and g["ds"] > g["dc"] and g["ds"] > g["sc"]
    assert _pair_rounds_gain(DINO, SIG, 24) == _pair_rounds_gain(SIG, DINO, 24)


def test_no_gain_when_rounds_are_whole():
    # 256-row multiples that fill whole rounds on their own: 65536 rows x 2048 columns = 2048 tiles = 8 rounds each
    a = _T(image_size=224, patch_size=14, hidden_size=2048, num_layers=2, num_heads=16, mlp_dim=2048, has_cls=False)
    assert _pair_rounds_gain(a, a, 256) == 0.0


def test_forward_steps_yields_two_linears_per_block():
    import inspect
    from cambrian_amd.model.multimodal_encoder.vit import ViTTrunk
    src = inspect.getsource(ViTTrunk.forward_steps)
    assert src.count("yield (") == 2 and '"proj"' in src and '"fc2"' in src
