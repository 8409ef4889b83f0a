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


# ------------------------------------------------------------------------------------------------------------------
# prompt side of the eval harness (mm_utils.py:203-250): integer list work, same results as the reference
# ------------------------------------------------------------------------------------------------------------------
def _tokenize_around_images(prompt: str, tokenizer, image_token_index: int, keep_bos_once: bool):
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    ids: list = []
    skip = 0
    if keep_bos_once and chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        skip = 1                                 # every chunk starts with BOS: keep the first, drop the others
        ids.append(chunks[0][0])
    for n, chunk in enumerate(chunks):
        if n:
            ids.append(image_token_index)        # one -200 placeholder where each "<image>" stood
        ids.extend(chunk[skip:])
    return ids


def _maybe_tensor(ids, return_tensors):
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def tokenizer_image_token(prompt, tokenizer, image_token_index=-200, return_tensors=None):
    """mm_utils.py:203-222: tokenise the text around every ``<image>`` and put IMAGE_TOKEN_INDEX in between; the BOS the
    tokenizer prepends to each chunk is kept once."""
    return _maybe_tensor(_tokenize_around_images(prompt, tokenizer, image_token_index, True), return_tensors)


def tokenizer_image_token_llama3(prompt, tokenizer, image_token_index=-200, return_tensors=None):
    """mm_utils.py:225-240: the Llama-3 tokenizer adds no BOS per chunk, so the chunks are joined as they are."""
    return _maybe_tensor(_tokenize_around_images(prompt, tokenizer, image_token_index, False), return_tensors)


def get_model_name_from_path(model_path: str) -> str:
    """mm_utils.py:243-249."""
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]
