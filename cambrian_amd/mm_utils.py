"""``cambrian/mm_utils.py`` surface for the image side (reference :153-201): same names and argument meaning, with
the per-image PIL work done by the GPU kernel pair behind ``cmb_image_preprocess`` for the whole batch at once."""
from __future__ import annotations

from typing import List, Sequence

import torch

from .train.image_pipeline import GpuImagePreprocessor

_PREPROCESSORS = {}


def expand2square(pil_img, background_color):
    """mm_utils.py:153-165 (kept for callers that letter-box on the CPU; the GPU route never materialises it)."""
    from PIL import Image
    width, height = pil_img.size
    if width == height:
        return pil_img
    side = max(width, height)
    result = Image.new(pil_img.mode, (side, side), background_color)
    result.paste(pil_img, (0, (side - height) // 2) if width > height else ((side - width) // 2, 0))
    return result


def process_images(images: Sequence, image_processor: Sequence, model_cfg=None, device=None,
                   dtype: torch.dtype = torch.float16) -> List[torch.Tensor]:
    """mm_utils.py:183-201: a list of PIL images and the towers' processors -> one ``[B,3,R_t,R_t]`` tensor per
    tower on the GPU.  The reference returns ``.half().cuda()`` tensors, hence the float16 default; the values are
    the reference's (float32 pixels rounded once to ``dtype``)."""
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    key = (tuple(id(p) for p in image_processor), str(device), dtype)
    hit = _PREPROCESSORS.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[0], image_processor)):
        # the entry keeps the processors alive, so their ids cannot be recycled for other objects while it exists
        hit = _PREPROCESSORS[key] = (list(image_processor), GpuImagePreprocessor(image_processor, device, dtype))
    return hit[1](images)
