"""OpenAI-CLIP tower wrapper — drop-in for cambrian/model/multimodal_encoder/clip_encoder.py:13-107."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .base_encoder import find_local_checkpoint, load_checkpoint_state, BaseVisionTower, ProcessorWrapper, SimpleImageTransform, logger
from .vit import ViTConfig, ViTTrunk, resample_tokens

# architecture table (no network: the HF config.json files cannot be fetched)
CLIP_ARCH = {
    "openai/clip-vit-large-patch14-336": dict(image_size=336, patch_size=14, hidden_size=1024, num_layers=24,
                                              num_heads=16, mlp_dim=4096),
    "openai/clip-vit-large-patch14": dict(image_size=224, patch_size=14, hidden_size=1024, num_layers=24,
                                          num_heads=16, mlp_dim=4096),
    "openai/clip-vit-base-patch16": dict(image_size=224, patch_size=16, hidden_size=768, num_layers=12,
                                         num_heads=12, mlp_dim=3072),
    "openai/clip-vit-base-patch32": dict(image_size=224, patch_size=32, hidden_size=768, num_layers=12,
                                         num_heads=12, mlp_dim=3072),
}


def extract_interp(model_name):
    """clip_encoder.py:13-25."""
    interp = None
    base_model_name = model_name
    if "interp" in model_name:
        base_model_name = model_name.split("-interp")[0]
    for part in model_name.split("-"):
        if part.startswith("interp"):
            interp = int(part[6:])
    return base_model_name, interp


class ClipVisionTower(BaseVisionTower):
    def __init__(self, vision_tower_name, args, delay_load=False):
        super().__init__(vision_tower_name, args, delay_load)
        base_model_name, interp = extract_interp(vision_tower_name)
        self.vision_tower_name = base_model_name
        self._interp_size = interp
        self._set_arch()
        if not self.delay_load:
            self.load_model()

    def _set_arch(self):
        if self.vision_tower_name not in CLIP_ARCH:
            raise ValueError(f"Unknown vision tower: {self.vision_tower_name}")
        a = CLIP_ARCH[self.vision_tower_name]
        self._arch = a
        self._hidden_size, self._image_size, self._patch_size = a["hidden_size"], a["image_size"], a["patch_size"]
        self.cfg_only = SimpleNamespace(hidden_size=a["hidden_size"], image_size=a["image_size"],
                                        patch_size=a["patch_size"], num_hidden_layers=a["num_layers"])

    def _vit_config(self) -> ViTConfig:
        a = self._arch
        # hidden_states[select_layer] (clip_encoder.py:66): index L+1+select_layer of [emb, l1..lL]
        run = a["num_layers"] + 1 + self.select_layer if self.select_layer < 0 else self.select_layer
        return ViTConfig(act="quick_gelu", ln_eps=1e-5, has_cls=True, pre_ln=True, final_ln=False, patch_bias=False,
                         run_layers=run, **a)

    def load_model(self, device_map=None):
        if self.is_loaded:
            logger.debug(f"{self.vision_tower_name} is already loaded, `load_model` called again, skipping.")
            return
        cfg = self._vit_config()
        dtype = getattr(self, "_compute_dtype", torch.bfloat16)
        gen = torch.Generator(device=self._target_device()).manual_seed(self._seed_for(self.vision_tower_name))
        ckpt = find_local_checkpoint(self.vision_tower_name)
        if ckpt is not None:   # HF CLIPVisionModel / CLIPModel checkpoint (clip_encoder.py:47)
            from .weight_maps import hf_clip_to_canonical
            sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v
                  for k, v in load_checkpoint_state(ckpt).items() if not k.startswith(("text_model.", "text_projection", "logit_scale"))}
            # an unfrozen tower keeps (and checkpoints) all layers, the frozen trunk only those before select_layer
            canon = hf_clip_to_canonical(sd, cfg.num_layers if self.unfreeze_mm_vision_tower else (cfg.run_layers or cfg.num_layers))
            logger.info(f"{self.vision_tower_name}: weights from {ckpt}")
        else:
            self._random_init_or_raise("no network for from_pretrained")
            canon = ViTTrunk.random_canonical(cfg, gen)
        from .weight_maps import canonical_to_hf_clip, hf_clip_to_canonical as _from_hf
        self.vision_tower = self._make_vit(cfg, canon, dtype, ref_keys=(   # keys of HF CLIPVisionModel (clip_encoder.py:47)
            lambda p_: canonical_to_hf_clip(p_, cfg.num_layers), lambda sd_: _from_hf(sd_, cfg.num_layers)))
        self.image_processor = ProcessorWrapper(SimpleImageTransform(self._image_size, flavour="hf"), height=self._image_size,
                                                width=self._image_size)
        self.is_loaded = True

    def _feature_select(self, image_features):
        if self.select_feature == "patch":
            return image_features  # the trunk already drops CLS
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def interpolate(self, image_features):
        """clip_encoder.py:70-96 (also produces the contiguous [B,T,C] copy of the CLS-stripped view)."""
        target = self._interp_size if self._interp_size is not None else image_features.shape[1]
        return self._resample(image_features, target)

    def _forward(self, images):
        with self._grad_mode():  # clip_encoder.py:103: torch.set_grad_enabled(self.unfreeze_mm_vision_tower)
            feats = self.vision_tower(images.to(device=self.device))
            feats = self.interpolate(self._feature_select(feats))
            return feats.to(images.dtype) if images.dtype in (torch.float32, torch.bfloat16) else feats
