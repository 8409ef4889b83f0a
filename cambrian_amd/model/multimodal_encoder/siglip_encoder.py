"""SigLIP tower wrapper — drop-in for cambrian/model/multimodal_encoder/siglip_encoder.py:9-99."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .base_encoder import (BaseVisionTower, ProcessorWrapper, SimpleImageTransform, find_local_checkpoint,
                           load_checkpoint_state, logger)
from .clip_encoder import ClipVisionTower
from .vit import ViTConfig, ViTTrunk

SIGLIP_ARCH = {
    # timm vit_so400m_patch14_siglip_{224,384} (timm==0.9.16: mlp_ratio 3.7362 -> 4304, class_token=False,
    # default nn.GELU, LayerNorm eps 1e-6, final norm applied by forward_features)
    "hf-hub:timm/ViT-SO400M-14-SigLIP-384": dict(image_size=384, patch_size=14, hidden_size=1152, num_layers=27,
                                                 num_heads=16, mlp_dim=4304),
    "hf-hub:timm/ViT-SO400M-14-SigLIP": dict(image_size=224, patch_size=14, hidden_size=1152, num_layers=27,
                                             num_heads=16, mlp_dim=4304),
    # BASELINE.json config 2: "SigLIP-style ViT-L/14" at 336 px (no CLS, conv bias, eps 1e-6, gelu-tanh)
    "siglip/ViT-L-14-336": dict(image_size=336, patch_size=14, hidden_size=1024, num_layers=24, num_heads=16,
                                mlp_dim=4096, act="gelu_tanh"),
}


def extract_res_interp(model_name):
    """siglip_encoder.py:9-34."""
    valid_model_prefixes = {
        "siglip/CLIP-ViT-SO400M-14-384": "hf-hub:timm/ViT-SO400M-14-SigLIP-384",
        "timm/ViT-SO400M-14-SigLIP-384": "hf-hub:timm/ViT-SO400M-14-SigLIP-384",
        "siglip/CLIP-ViT-SO400M-14": "hf-hub:timm/ViT-SO400M-14-SigLIP",
        "timm/ViT-SO400M-14-SigLIP": "hf-hub:timm/ViT-SO400M-14-SigLIP",
        "siglip/ViT-L-14-336": "siglip/ViT-L-14-336",
    }
    res = 384 if "384" in model_name else 224
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


class SiglipVisionTower(ClipVisionTower):
    def __init__(self, vision_tower_name, args, delay_load=False):
        BaseVisionTower.__init__(self, vision_tower_name, args, delay_load)
        base_model_name, res, interp = extract_res_interp(vision_tower_name)
        self.vision_tower_name = base_model_name
        self._interp_size = interp
        a = dict(SIGLIP_ARCH[base_model_name])
        self._act = a.pop("act", "gelu")
        self._arch = a
        self._hidden_size, self._image_size, self._patch_size = a["hidden_size"], a["image_size"], a["patch_size"]
        self.cfg_only = SimpleNamespace(hidden_size=a["hidden_size"], image_size=a["image_size"],
                                        patch_size=a["patch_size"], num_hidden_layers=a["num_layers"])
        if not self.delay_load:
            self.load_model()

    def _vit_config(self) -> ViTConfig:
        # siglip_encoder.py:97 runs the whole trunk (forward_features) and ignores select_layer
        return ViTConfig(act=self._act, ln_eps=1e-6, has_cls=False, pre_ln=False, final_ln=True, patch_bias=True,
                         **self._arch)

    def load_model(self, device_map=None):
        if self.is_loaded:
            return
        self.vision_model = "siglip"
        cfg = self._vit_config()
        dtype = getattr(self, "_compute_dtype", torch.bfloat16)
        gen = torch.Generator(device=self._target_device()).manual_seed(self._seed_for(self.vision_tower_name))
        ckpt = find_local_checkpoint(self.vision_tower_name)
        if ckpt is not None:   # open_clip checkpoint (visual.trunk.* = timm ViT, siglip_encoder.py:53-56) or HF SiglipVisionModel
            from .weight_maps import hf_siglip_to_canonical, timm_vit_to_canonical
            sd = load_checkpoint_state(ckpt)
            if any(k.startswith(("visual.trunk.", "trunk.")) or k == "pos_embed" for k in sd):
                sd = {k: v for k, v in sd.items() if k.startswith(("visual.trunk.", "trunk.")) or not k.startswith(("text.", "visual."))}
                canon = timm_vit_to_canonical(sd, cfg.num_layers)
            else:
                sd = {k: v for k, v in sd.items() if not k.startswith(("text_model.", "logit_"))}
                canon = hf_siglip_to_canonical(sd, cfg.num_layers)
            logger.info(f"{self.vision_tower_name}: weights from {ckpt}")
        else:
            self._random_init_or_raise("no network for open_clip hub download")
            canon = ViTTrunk.random_canonical(cfg, gen)
        from .weight_maps import canonical_to_timm_vit, timm_vit_to_canonical as _from_timm
        self.vision_tower = self._make_vit(cfg, canon, dtype, ref_keys=(   # keys of the timm trunk (siglip_encoder.py:55)
            lambda p_: canonical_to_timm_vit(p_, cfg.num_layers), lambda sd_: _from_timm(sd_, cfg.num_layers)))
        self.image_processor = ProcessorWrapper(SimpleImageTransform(self._image_size, [0.5] * 3, [0.5] * 3),
                                                height=self._image_size, width=self._image_size, image_mean=[0.5] * 3)
        self.is_loaded = True

    def _forward(self, images, interpolate_token=576, trunk_out=None):
        """``trunk_out``: the trunk's output for ``images`` when the caller has already run it (the paired tower launch of
        encode_images)."""
        with self._grad_mode():  # siglip_encoder.py:96
            seq = self.vision_tower(images.to(device=self.device)) if trunk_out is None else trunk_out
            feats = self.interpolate(seq)
            return feats.to(images.dtype) if images.dtype in (torch.float32, torch.bfloat16) else feats
