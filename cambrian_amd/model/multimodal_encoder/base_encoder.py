"""Tower protocol of the reference (cambrian/model/multimodal_encoder/base_encoder.py:12-134): same attribute
and property names, so cambrian_arch / the train harness can treat these towers exactly like the originals.
The arithmetic lives in ``vit.py`` / ``convnext.py`` (HIP kernels); there is no HF/timm module inside."""
from __future__ import annotations

import logging
import os
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn

logger = logging.getLogger("cambrian_amd")

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


class ProcessorWrapper:
    """base_encoder.py:12-30 — wraps a callable transform behind the HF image-processor protocol."""

    def __init__(self, transform, height=378, width=378, image_mean=OPENAI_CLIP_MEAN):
        self._crop_size = {"height": height, "width": width}
        self._transforms = transform
        self.image_mean = image_mean

    @property
    def crop_size(self):
        return self._crop_size

    def preprocess(self, image, return_tensors="pt"):
        return {"pixel_values": [self._transforms(image)]}


class SimpleImageTransform:
    """Resize (bicubic) to size x size and normalise: the offline stand-in for CLIPImageProcessor / the
    open_clip transform (no network to fetch their configs).  ``flavour`` names whose float arithmetic the pointwise
    stage follows — 'hf' (rescale by 1/255 in float64 -> float32, then (x - mean) / std: CLIP, DINOv2) or
    'torchvision' (ToTensor float32 / 255, Normalize: SigLIP, ConvNeXt).  This is the per-sample CPU route of the
    reference (train_fsdp.py:1004-1008); the step-boundary route that does the same arithmetic for a whole batch on
    the GPU is cambrian_amd/train/image_pipeline.py, which reads ``mean`` / ``std`` / ``flavour`` from here."""

    def __init__(self, size: int, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD, flavour: str = "torchvision"):
        self.size, self.mean, self.std, self.flavour = size, mean, std, flavour

    def __call__(self, image):
        import numpy as np
        img = image.convert("RGB").resize((self.size, self.size), resample=3)
        u = np.asarray(img)
        x = (u * (1 / 255)).astype(np.float32) if self.flavour == "hf" else u.astype(np.float32) / np.float32(255)
        x = torch.from_numpy(x).permute(2, 0, 1)
        mean = torch.tensor(self.mean, dtype=torch.float32).view(3, 1, 1)
        std = torch.tensor(self.std, dtype=torch.float32).view(3, 1, 1)
        return (x - mean) / std


# ------------------------------------------------------------------------------------------------------------------
# Real-weight loading.  The reference downloads the towers (HF ``from_pretrained``:
# clip_encoder.py:47, dino_encoder.py:81; open_clip hub: siglip_encoder.py:53-56, clip_convnext_encoder.py:84-90).
# Here a tower looks for a LOCAL checkpoint — the name itself if it is a directory, else
# ``$CAMBRIAN_WEIGHTS_DIR/<name>``, ``…/<name with '/' -> '--'>`` or the HF cache layout ``models--<org>--<repo>`` — and
# maps its keys onto the canonical names of the native trunks (weight_maps.py); without one it falls back to seeded
# random-init weights of the same architecture and says so.
# ------------------------------------------------------------------------------------------------------------------
_CKPT_FILES = ("model.safetensors", "open_clip_model.safetensors", "pytorch_model.bin", "open_clip_pytorch_model.bin")


def find_local_checkpoint(name: str):
    import glob
    import os
    roots = []
    if os.path.isdir(name):
        roots.append(name)
    wd = os.environ.get("CAMBRIAN_WEIGHTS_DIR")
    if wd:
        flat = name.replace("hf-hub:", "")
        roots += [os.path.join(wd, flat), os.path.join(wd, flat.replace("/", "--"))]
        roots += glob.glob(os.path.join(wd, "models--" + flat.replace("/", "--"), "snapshots", "*"))
    for r in roots:
        for fn in _CKPT_FILES:
            path = os.path.join(r, fn)
            if os.path.isfile(path):
                return path
    return None


def load_checkpoint_state(path: str):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


class BaseVisionTower(nn.Module):
    """base_encoder.py:33-134."""

    def __init__(self, vision_tower_name, args, delay_load=False):
        super().__init__()
        self.is_loaded = False
        self.args = args
        self.vision_tower_name = vision_tower_name
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.unfreeze_mm_vision_tower = getattr(args, "unfreeze_mm_vision_tower", False)
        self.delay_load = delay_load
        self._interp_size = None

    # -- subclasses ------------------------------------------------------------------------------
    def load_model(self, device_map=None):
        raise NotImplementedError("Subclasses must implement load_model")

    def _forward(self, images):
        raise NotImplementedError("Subclasses must implement forward")

    def forward(self, images):
        if type(images) is list:  # base_encoder.py:55-64
            return [self._forward(image.unsqueeze(0)) for image in images]
        return self._forward(images)

    # -- helpers ---------------------------------------------------------------------------------
    def _make_vit(self, cfg, canon, dtype, pos_fn=None, ref_keys=None):
        """Frozen: packed buffers + raw kernels, no graph (vit.py).  ``unfreeze_mm_vision_tower``: fp32 master
        parameters + autograd operators (vit_train.py); ``ref_keys`` = (to_ref, from_ref) makes its
        state_dict use the reference module's key names."""
        if self.unfreeze_mm_vision_tower:
            from .vit_train import TrainableViT
            trunk = TrainableViT(cfg, canon, self._target_device(), dtype, pos_fn=pos_fn)
            if ref_keys is not None:
                from .weight_maps import ReferenceKeys
                ReferenceKeys(*ref_keys).install(trunk)
            return trunk
        from .vit import ViTTrunk
        if pos_fn is not None:
            canon = dict(canon)
            canon["pos"] = pos_fn(canon["pos"])
        return ViTTrunk(cfg, dtype).load_canonical(canon, self._target_device())

    def _random_init_or_raise(self, why: str) -> None:
        """No local checkpoint was found.  The reference fails here (``from_pretrained`` / the open_clip hub download
        raise); silently running on random weights would hand ``load_pretrained_model()`` a model whose vision features are
        noise, so random initialisation of the same architecture needs an explicit opt-in: ``CAMBRIAN_AMD_RANDOM_INIT=1``
        (set by bench.py and the tests, which have no network for the released weights)."""
        import os
        if os.environ.get("CAMBRIAN_AMD_RANDOM_INIT", "0") not in ("1", "true", "True"):
            raise FileNotFoundError(
                f"{self.vision_tower_name}: no local checkpoint ({why}); put the weights under $CAMBRIAN_WEIGHTS_DIR or the HF "
                f"cache, or set CAMBRIAN_AMD_RANDOM_INIT=1 to run on seeded random weights of the same architecture")
        logger.warning(f"{self.vision_tower_name}: random-init weights ({why}; CAMBRIAN_AMD_RANDOM_INIT=1)")

    def _resample(self, feats, target):
        """Token-grid resize of the wrappers; differentiable when the tower trains."""
        if self.unfreeze_mm_vision_tower:
            from .vit_train import resample_tokens_autograd
            return resample_tokens_autograd(feats, target)
        from .vit import resample_tokens
        return resample_tokens(feats, target, force_copy=True)

    def _grad_mode(self):
        return torch.set_grad_enabled(bool(self.unfreeze_mm_vision_tower) and torch.is_grad_enabled())

    @staticmethod
    def _seed_for(name: str) -> int:
        return int.from_bytes(name.encode()[:8].ljust(8, b"\0"), "little") % (2 ** 31)

    def _target_device(self):
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        vt = getattr(self, "vision_tower", None)
        return vt.compute_dtype if vt is not None else torch.float32

    @property
    def device(self):
        vt = getattr(self, "vision_tower", None)
        if vt is not None:
            for b in vt.buffers():
                return b.device
            for q in vt.parameters():     # trainable trunk: fp32 master parameters instead of packed buffers
                return q.device
        return torch.device("cpu")

    @property
    def config(self):
        return self.cfg_only

    @property
    def hidden_size(self):
        return self._hidden_size

    @property
    def image_size(self):
        return self._image_size

    @property
    def patch_size(self):
        return self._patch_size

    @property
    def num_patches_per_side(self):
        if self._interp_size is not None:
            return int(self._interp_size ** 0.5)
        return self.image_size // self.patch_size

    @property
    def num_patches(self):
        if self._interp_size is not None:
            return self._interp_size
        return self.num_patches_per_side ** 2

    def to(self, *args, **kwargs):
        """The reference moves frozen towers with ``.to(dtype=bf16, device=...)`` (train_fsdp.py:1659); the packed
        weights are re-created in the requested dtype because packing (padding, K-step) depends on it."""
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
        out = super().to(*args, **kwargs)
        if dtype is not None and self.is_loaded and dtype != self.vision_tower.compute_dtype:
            dev = self.device
            self.is_loaded = False
            self._compute_dtype = dtype
            self.load_model()
            self.vision_tower.to(dev)
        return out
