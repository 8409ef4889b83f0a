"""TEST INFRASTRUCTURE ONLY — CPU restatement of the image pre-processing the reference performs per sample and per
tower before the hot path:

    cambrian/train/train_fsdp.py:985-1008   (LazySupervisedDataset.__getitem__)  and
    cambrian/mm_utils.py:153-165,183-201    (expand2square, process_images):

        image_aux = expand2square(image, tuple(int(x*255) for x in processor.image_mean)).resize((R, R))
        image_aux = processor.preprocess(image_aux, return_tensors='pt')['pixel_values'][0]

``Image.resize`` is Pillow (third-party, unpinned in the reference's pyproject.toml; installed here: 12.2.0).  Its
default filter for RGB images is BICUBIC, computed by libImaging/Resample.c: double-precision coefficient rows
(``precompute_coeffs``), rounded to 22-bit fixed point (``normalize_coeffs_8bpc``), a horizontal pass into a uint8
image and a vertical pass over that, each rounding with ``clip8((1 << 21) + sum) >> 22``.  A same-size resize returns
a copy.  ``processor.preprocess`` on an already R x R image is pointwise: HF image processors (CLIP, DINOv2 —
clip_encoder.py:46, dino_encoder.py:96) rescale by 1/255 in float64, cast to float32, then (x - mean) / std in
float32; the open_clip / torchvision transform (SigLIP, ConvNeXt — siglip_encoder.py:53-60,
clip_convnext_encoder.py:84-88) is ToTensor (float32 / 255) and Normalize (sub, div in float32).

Pinned: tests/test_preprocess.py checks ``resize_square`` bit-for-bit against Pillow itself on seeded images (Pillow
is importable wherever the tests run) and against the committed fixtures tests/golden/preprocess_*.npz that
tests/golden/make_preprocess_golden.py produced by running the reference's own expression above.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2            # Resample.c
BICUBIC_SUPPORT = 2.0


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def resize_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [out,2] = (first, count), coefficients int32 [out, ksize], ksize) — precompute_coeffs +
    normalize_coeffs_8bpc for the full-image box (in0 = 0, in1 = in_size)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                      # sequential sum, as the C loop does
            ww += v
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return bounds, np.trunc(fixed).astype(np.int32), ksize


def _pass(src: np.ndarray, bounds: np.ndarray, coefs: np.ndarray) -> np.ndarray:
    """Resample axis 1 of uint8 [rows, n, c] to [rows, out, c]."""
    out = np.empty((src.shape[0], bounds.shape[0], src.shape[2]), np.uint8)
    s64 = src.astype(np.int64)
    for xx, (first, count) in enumerate(bounds):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(s64[:, first:first + count], coefs[xx, :count].astype(np.int64),
                                                          axes=([1], [0]))
        out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def expand2square(img: np.ndarray, background: Sequence[int]) -> np.ndarray:
    """mm_utils.py:153-165 on a uint8 [h, w, 3] array."""
    h, w = img.shape[:2]
    if w == h:
        return img
    side = max(w, h)
    out = np.empty((side, side, 3), np.uint8)
    out[:] = np.asarray(background, np.uint8)
    if w > h:
        top = (w - h) // 2
        out[top:top + h] = img
    else:
        left = (h - w) // 2
        out[:, left:left + w] = img
    return out


def resize_square(img: np.ndarray, out_side: int) -> np.ndarray:
    """``Image.resize((R, R))`` (bicubic) of a square uint8 [S, S, 3] array."""
    side = img.shape[0]
    assert img.shape[1] == side
    if side == out_side:
        return img.copy()
    bounds, coefs, _ = resize_coeffs(side, out_side)
    tmp = _pass(img, bounds, coefs)                                  # horizontal first (ImagingResample)
    return _pass(tmp.transpose(1, 0, 2), bounds, coefs).transpose(1, 0, 2)


def background_of(image_mean: Sequence[float]) -> Tuple[int, int, int]:
    return tuple(int(x * 255) for x in image_mean)                   # train_fsdp.py:1006


def pixel_lut(mean: Sequence[float], std: Sequence[float], flavour: str) -> np.ndarray:
    """float32 [3, 256]: the normalised value of every uint8 level, per channel.
    'hf'          : transformers image_transforms.rescale (uint8 * python float -> float64 -> float32) then
                    normalize ((x - float32 mean) / float32 std);
    'torchvision' : ToTensor (float32 / 255) then Normalize (sub, div, float32)."""
    u = np.arange(256, dtype=np.uint8)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    if flavour == "hf":
        x = (u * (1 / 255)).astype(np.float32)
    elif flavour == "torchvision":
        x = u.astype(np.float32) / np.float32(255)
    else:
        raise ValueError(flavour)
    return ((x[None, :] - m) / s).astype(np.float32)


def preprocess(img: np.ndarray, out_side: int, image_mean, mean, std, flavour: str) -> np.ndarray:
    """uint8 [h, w, 3] -> float32 [3, R, R]: the two reference lines quoted in the header."""
    sq = resize_square(expand2square(img, background_of(image_mean)), out_side)
    lut = pixel_lut(mean, std, flavour)
    return np.stack([lut[c][sq[:, :, c]] for c in range(3)])
