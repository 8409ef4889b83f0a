"""Tower registry — drop-in for cambrian/model/multimodal_encoder/builder.py:23-148 restricted to the four
production towers of every Cambrian-1 release config (scripts/cambrian/*.sh:15); the 14 ablation towers of the
reference's encoder study are out of scope and raise the reference's own ValueError."""
from __future__ import annotations

import copy

from .base_encoder import logger
from .clip_convnext_encoder import CLIPConvNextTower
from .clip_encoder import ClipVisionTower
from .dino_encoder import DinoVisionTower
from .siglip_encoder import SiglipVisionTower

_DISPATCH = (  # same substring tests, same order as builder.py:32-56
    ("openai/clip", "OpenAI CLIP", ClipVisionTower),
    ("siglip", "SigLIP CLIP", SiglipVisionTower),
    ("clip-convnext", "ConvNeXt CLIP", CLIPConvNextTower),
    ("dinov2", "DINO", DinoVisionTower),
)


def _make(name: str, cfg, **kwargs):
    for key, label, cls in _DISPATCH:
        if key in name.lower():
            logger.info(f"Loading **{label}** Vision Tower: {name}")
            return cls(name, args=cfg, **kwargs)
    raise ValueError(f"Unknown vision tower: {name}")


def build_vision_tower(vision_tower_cfg, **kwargs):
    vision_tower = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if vision_tower is None or not isinstance(vision_tower, str):
        raise ValueError(f"Vision Tower is not specified in the config: {vision_tower_cfg}")
    return _make(vision_tower, vision_tower_cfg, **kwargs)


def build_vision_tower_aux_list(vision_tower_cfg, **kwargs):
    names = getattr(vision_tower_cfg, "mm_vision_tower_aux_list", getattr(vision_tower_cfg, "vision_tower_aux_list", None))
    lens = getattr(vision_tower_cfg, "mm_vision_tower_aux_token_len_list",
                   getattr(vision_tower_cfg, "vision_tower_aux_token_len_list", None))
    towers = []
    for name, token_len in zip(names, lens):
        config = copy.deepcopy(vision_tower_cfg)
        name = name + "-interp{}".format(token_len)  # builder.py:92
        towers.append(_make(name, config, **kwargs))
    return towers
