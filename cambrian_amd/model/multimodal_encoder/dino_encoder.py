"""DINOv2 tower wrapper — drop-in for cambrian/model/multimodal_encoder/dino_encoder.py:11-176."""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .base_encoder import (BaseVisionTower, ProcessorWrapper, SimpleImageTransform, find_local_checkpoint,
                           load_checkpoint_state, logger)
from .vit import ViTConfig, ViTTrunk, resample_tokens

IMAGENET_MEAN, IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]

DINO_ARCH = {  # HF Dinov2Config of each checkpoint (native image_size 518, patch 14)
    "facebook/dinov2-small": dict(hidden_size=384, num_layers=12, num_heads=6, mlp_dim=1536, act="gelu"),
    "facebook/dinov2-base": dict(hidden_size=768, num_layers=12, num_heads=12, mlp_dim=3072, act="gelu"),
    "facebook/dinov2-large": dict(hidden_size=1024, num_layers=24, num_heads=16, mlp_dim=4096, act="gelu"),
    # giant: use_swiglu_ffn, hidden_features = (int(4*1536*2/3)+7)//8*8 = 4096 (HF Dinov2SwiGLUFFN)
    "facebook/dinov2-giant": dict(hidden_size=1536, num_layers=40, num_heads=24, mlp_dim=4096, act="swiglu"),
}
NATIVE_IMAGE_SIZE = 518


def extract_res_interp(model_name):
    """dino_encoder.py:11-37."""
    valid_model_prefixes = ["facebook/dinov2-small", "facebook/dinov2-base", "facebook/dinov2-large",
                            "facebook/dinov2-giant-imagenet1k-1-layer", "facebook/dinov2-giant"]
    for prefix in valid_model_prefixes:
        if model_name.startswith(prefix):
            base_model_name = prefix
            break
    else:
        raise ValueError(f"Unknown vision tower: {model_name}")
    res = None
    interp = None
    for part in model_name[len(base_model_name):].split("-"):
        if part.startswith("res"):
            res = int(part[3:])
        elif part.startswith("interp"):
            interp = int(part[6:])
    return base_model_name, res, interp


def interpolate_pos_encoding(pos: torch.Tensor, grid: int) -> torch.Tensor:
    """HF Dinov2Embeddings.interpolate_pos_encoding as pinned by the reference (transformers==4.37.0):
    bicubic, align_corners=False, scale_factor = (grid + 0.1) / sqrt(N) — the "+0.1" guard is part of the
    pinned arithmetic.  Input-independent, so it is evaluated once at weight-pack time."""
    cls_pos, patch_pos = pos[:1], pos[1:]
    n = patch_pos.shape[0]
    side = int(math.sqrt(n))
    if side == grid:
        return pos
    dim = pos.shape[-1]
    pp = patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2).float()
    sf = (grid + 0.1) / math.sqrt(n)
    pp = F.interpolate(pp, scale_factor=(sf, sf), mode="bicubic", align_corners=False)
    if pp.shape[-1] != grid or pp.shape[-2] != grid:
        raise ValueError("Width or height does not match with the interpolated position embeddings")
    pp = pp.permute(0, 2, 3, 1).reshape(grid * grid, dim)
    return torch.cat([cls_pos.float(), pp], 0)


class DinoVisionTower(BaseVisionTower):
    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__(vision_tower, args, delay_load)
        base_model_name, res, interp = extract_res_interp(self.vision_tower_name)
        self._vision_tower_name = vision_tower
        self.vision_tower_name = base_model_name
        self._image_size = res if res is not None else NATIVE_IMAGE_SIZE
        self._interp_size = interp
        self._patch_size = 14
        name = "facebook/dinov2-giant" if base_model_name.startswith("facebook/dinov2-giant") else base_model_name
        self._arch = DINO_ARCH[name]
        self._hidden_size = self._arch["hidden_size"]
        self.cfg_only = SimpleNamespace(hidden_size=self._hidden_size, image_size=NATIVE_IMAGE_SIZE, patch_size=14,
                                        num_hidden_layers=self._arch["num_layers"])
        if not self.delay_load:
            self.load_model()

    def load_model(self, device_map=None):
        if self.is_loaded:
            return
        a = self._arch
        native = ViTConfig(image_size=NATIVE_IMAGE_SIZE, patch_size=14, ln_eps=1e-6, has_cls=True, pre_ln=False,
                           final_ln=True, layerscale=True, patch_bias=True, **a)
        run = ViTConfig(image_size=self._image_size, patch_size=14, ln_eps=1e-6, has_cls=True, pre_ln=False,
                        final_ln=True, layerscale=True, patch_bias=True, **a)
        dtype = getattr(self, "_compute_dtype", torch.bfloat16)
        gen = torch.Generator(device=self._target_device()).manual_seed(self._seed_for(self.vision_tower_name))
        ckpt = find_local_checkpoint(self.vision_tower_name)
        if ckpt is not None:   # HF Dinov2Model checkpoint (dino_encoder.py:81)
            from .weight_maps import hf_dinov2_to_canonical
            canon = hf_dinov2_to_canonical(load_checkpoint_state(ckpt), native.num_layers, native.act == "swiglu")
            logger.info(f"{self.vision_tower_name}: weights from {ckpt}")
        else:
            self._random_init_or_raise("no network for from_pretrained")
            canon = ViTTrunk.random_canonical(native, gen)
        # 37x37 -> e.g. 27x27 at 378 px: once at load when frozen, inside every forward (differentiably) when training
        from .weight_maps import canonical_to_hf_dinov2, hf_dinov2_to_canonical as _from_hf
        sw = native.act == "swiglu"
        self.vision_tower = self._make_vit(run, canon, dtype, pos_fn=lambda pos: interpolate_pos_encoding(pos, run.grid),
                                           ref_keys=(lambda p_: canonical_to_hf_dinov2(p_, native.num_layers, sw),   # HF Dinov2Model
                                                     lambda sd_: _from_hf(sd_, native.num_layers, sw)))
        self.image_processor = ProcessorWrapper(SimpleImageTransform(self._image_size, IMAGENET_MEAN, IMAGENET_STD, flavour="hf"),
                                                height=self._image_size, width=self._image_size,
                                                image_mean=IMAGENET_MEAN)
        self.is_loaded = True

    @property
    def image_size(self):
        return self._image_size

    def feature_select(self, sequence_output):
        if self.select_feature == "patch":
            return sequence_output  # the trunk already drops CLS (dino_encoder.py:120-121)
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def interpolate(self, image_features):
        """dino_encoder.py:128-154."""
        target = self._interp_size if self._interp_size is not None else image_features.shape[1]
        return self._resample(image_features, target)

    def _forward(self, images, trunk_out=None):
        """``trunk_out``: the trunk's output for ``images`` when the caller has already run it (the paired tower launch of
        encode_images); the wrapper's own steps follow unchanged."""
        with self._grad_mode():  # dino_encoder.py:158
            seq = self.vision_tower(images.to(device=self.device)) if trunk_out is None else trunk_out
            feats = self.interpolate(self.feature_select(seq))
            return feats.to(images.dtype) if images.dtype in (torch.float32, torch.bfloat16) else feats

    @property
    def num_patches_per_side(self):
        return int(self.num_patches ** 0.5)

    @property
    def num_patches(self):
        if self._interp_size is None:
            return (self._image_size // self._patch_size) ** 2
        return self._interp_size
