"""CLIP-ConvNeXt tower wrapper — drop-in for cambrian/model/multimodal_encoder/clip_convnext_encoder.py:11-176."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .base_encoder import (BaseVisionTower, ProcessorWrapper, SimpleImageTransform, find_local_checkpoint,
                           load_checkpoint_state, logger)
from .convnext import ConvNeXtConfig, ConvNeXtTrunk

CONVNEXT_ARCH = {
    "hf-hub:laion/CLIP-convnext_large_d_320.laion2B-s29B-b131K-ft-soup": dict(depths=(3, 3, 27, 3),
                                                                            dims=(192, 384, 768, 1536), ln_eps=1e-6),
    "hf-hub:laion/CLIP-convnext_xxlarge-laion2B-s34B-b82K-augreg-soup": dict(depths=(3, 4, 30, 3),
                                                                            dims=(384, 768, 1536, 3072), ln_eps=1e-5),
}


def extract_res_interp(model_name):
    """clip_convnext_encoder.py:11-35."""
    valid_model_prefixes = {
        "clip-convnext-L": "hf-hub:laion/CLIP-convnext_large_d_320.laion2B-s29B-b131K-ft-soup",
        "clip-convnext-XXL": "hf-hub:laion/CLIP-convnext_xxlarge-laion2B-s34B-b82K-augreg-soup",
    }
    res = None
    interp = None
    for prefix in valid_model_prefixes:
        if model_name.startswith(prefix):
            base_model_name = valid_model_prefixes[prefix]
            break
    else:
        raise ValueError(f"Unknown vision tower: {model_name}")
    for part in model_name.split("-"):
        if part.startswith("res"):
            res = int(part[3:])
        elif part.startswith("interp"):
            interp = int(part[6:])
    return base_model_name, res, interp


class CLIPConvNextTower(BaseVisionTower):
    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__(vision_tower, args, delay_load)
        self.is_multi_stage = "multi-stage" in vision_tower
        base_model_name, res, interp = extract_res_interp(vision_tower)
        self.vision_tower_name = base_model_name
        self._image_size = res if res is not None else 1024
        self._interp_size = interp
        self._reduction = 32
        self._arch = CONVNEXT_ARCH[base_model_name]
        dims = self._arch["dims"]
        self._hidden_size = sum(dims) if self.is_multi_stage else dims[-1]
        self.cfg_only = SimpleNamespace(hidden_size=self._hidden_size, image_size=self._image_size)
        if not delay_load:
            self.load_model()

    def load_model(self, device_map=None):
        if self.is_loaded:
            return
        assert "clip-convnext" in self.vision_tower_name.lower() or "convnext" in self.vision_tower_name.lower()
        self.vision_model = "convnext"
        cfg = ConvNeXtConfig(**self._arch)
        dtype = getattr(self, "_compute_dtype", torch.bfloat16)
        gen = torch.Generator(device=self._target_device()).manual_seed(self._seed_for(self.vision_tower_name))
        ckpt = find_local_checkpoint(self.vision_tower_name)
        if ckpt is not None:   # open_clip checkpoint (visual.trunk.* = timm ConvNeXt, clip_convnext_encoder.py:84-90) or HF ConvNextModel
            from .weight_maps import hf_convnext_to_canonical, timm_convnext_to_canonical
            sd = load_checkpoint_state(ckpt)
            if any(k.startswith(("visual.trunk.", "trunk.", "stem.")) for k in sd):
                canon = timm_convnext_to_canonical({k: v for k, v in sd.items() if not k.startswith("text.")}, cfg.depths)
            else:
                canon = hf_convnext_to_canonical(sd, cfg.depths)
            logger.info(f"{self.vision_tower_name}: weights from {ckpt}")
        else:
            self._random_init_or_raise("no network for open_clip hub download")
            canon = ConvNeXtTrunk.random_canonical(cfg, gen)
        if self.unfreeze_mm_vision_tower:      # fp32 master parameters + autograd operators
            from .convnext_train import TrainableConvNeXt
            from .weight_maps import ReferenceKeys, canonical_to_timm_convnext, timm_convnext_to_canonical as _from_timm
            self.vision_tower = TrainableConvNeXt(cfg, canon, self._target_device(), dtype)
            ReferenceKeys(lambda p_: canonical_to_timm_convnext(p_, cfg.depths),           # keys of the timm trunk (:89)
                          lambda sd_: _from_timm(sd_, cfg.depths)).install(self.vision_tower)
        else:
            self.vision_tower = ConvNeXtTrunk(cfg, dtype).load_canonical(canon, self._target_device())
        self.image_processor = ProcessorWrapper(SimpleImageTransform(self._image_size), height=self._image_size,
                                                width=self._image_size)
        self.is_loaded = True

    def _forward(self, images):
        with self._grad_mode():  # clip_convnext_encoder.py:147: torch.set_grad_enabled(self.unfreeze_mm_vision_tower)
            side = None if self._interp_size is None else self.num_patches_per_side
            feats = self.vision_tower(images.to(device=self.device), side, multi_stage=self.is_multi_stage)
            return feats.to(images.dtype) if images.dtype in (torch.float32, torch.bfloat16) else feats

    @property
    def image_size(self):
        return self._image_size

    @property
    def patch_size(self):
        return self._reduction

    @property
    def num_patches_per_side(self):
        if self._interp_size is None:
            return self._image_size // self._reduction
        return int(self._interp_size ** 0.5)

    @property
    def num_patches(self):
        if self._interp_size is None:
            return (self._image_size // self._reduction) ** 2
        return self._interp_size
