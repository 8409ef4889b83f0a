"""Generates tests/golden/preprocess_cases.npz by running the REFERENCE's own pre-processing expression
(cambrian/train/train_fsdp.py:1004-1008 == cambrian/mm_utils.py:190-195) on seeded images:

    image_aux = expand2square(image, tuple(int(x*255) for x in processor.image_mean)).resize((R, R))
    image_aux = processor.preprocess(image_aux, return_tensors='pt')['pixel_values'][0]

``expand2square`` is exec'd from its line range in /root/reference/cambrian/mm_utils.py (the module itself imports
the cambrian package, which drags in timm/open_clip); ``Image`` is the installed Pillow; ``processor`` is the
installed transformers CLIPImageProcessor / BitImageProcessor (PIL backend; the classes clip_encoder.py:46 and
dino_encoder.py:96 instantiate) with the towers' published mean/std, plus — for the open_clip towers, whose
torchvision transform is not installable here — ToTensor/Normalize restated with the torch ops torchvision's
functional.to_tensor / normalize consist of (uint8 -> float32 .div(255); sub_(mean).div_(std)).

Run in the build container only:  python tests/golden/make_preprocess_golden.py
"""
from __future__ import annotations

import os

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

OPENAI_MEAN, OPENAI_STD = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
IMAGENET_MEAN, IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def load_ref_expand2square():
    src = open(f"{REF}/cambrian/mm_utils.py").read().split("\n")
    ns = {"Image": Image}
    exec("\n".join(src[152:165]), ns)
    return ns["expand2square"]


class TorchvisionLike:
    """ToTensor + Normalize as torchvision computes them, behind the ProcessorWrapper protocol
    (base_encoder.py:12-30: image_mean defaults to the OpenAI mean whatever the transform normalises with)."""

    def __init__(self, side, mean, std):
        self.crop_size = {"height": side, "width": side}
        self.image_mean = OPENAI_MEAN
        self.mean, self.std = mean, std

    def preprocess(self, image, return_tensors="pt"):
        x = torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        x = x.sub_(torch.tensor(self.mean).view(3, 1, 1)).div_(torch.tensor(self.std).view(3, 1, 1))
        return {"pixel_values": [x]}


def towers(scale=1.0):
    from transformers import BitImageProcessor, CLIPImageProcessor

    def s(v):
        return max(8, int(v * scale))
    clip = CLIPImageProcessor(size={"shortest_edge": s(336)}, crop_size={"height": s(336), "width": s(336)},
                              image_mean=OPENAI_MEAN, image_std=OPENAI_STD)
    dino = BitImageProcessor(size={"shortest_edge": s(378)}, crop_size={"height": s(378), "width": s(378)},
                             image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD)
    return [("siglip", TorchvisionLike(s(384), [0.5] * 3, [0.5] * 3), "torchvision", [0.5] * 3, [0.5] * 3),
            ("clip", clip, "hf", OPENAI_MEAN, OPENAI_STD),
            ("dino", dino, "hf", IMAGENET_MEAN, IMAGENET_STD),
            ("convnext", TorchvisionLike(s(1024), OPENAI_MEAN, OPENAI_STD), "torchvision", OPENAI_MEAN, OPENAI_STD)]


def main():
    expand2square = load_ref_expand2square()
    rng = np.random.default_rng(20240627)
    out = {}
    # (w, h): wide, tall, square = a tower's side (copy), tiny, letter-boxed to a tower's side (copy), strong downscale
    cases = [(61, 40), (37, 90), (24, 24), (5, 3), (64, 48), (200, 150)]
    tw = towers(scale=0.0625)         # 24 / 21 / 23 / 64 pixel towers keep the fixture small
    meta = []
    for ci, (w, h) in enumerate(cases):
        # smooth + noise so that both interpolation and clipping paths are exercised
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 9 % 256)], -1)
        img = np.clip(base + rng.integers(-60, 60, (h, w, 3)), 0, 255).astype(np.uint8)
        out[f"img{ci}"] = img
        pil = Image.fromarray(img).convert("RGB")
        for name, proc, flavour, mean, std in tw:
            R = proc.crop_size["height"]
            aux = expand2square(pil, tuple(int(x * 255) for x in proc.image_mean)).resize((R, R))
            px = proc.preprocess(aux, return_tensors="pt")["pixel_values"][0]
            out[f"u8_{ci}_{name}"] = np.asarray(aux).copy()
            if ci < 2:                # the float stage is pointwise: two cases pin it
                out[f"px_{ci}_{name}"] = np.asarray(px, dtype=np.float32)
    for name, proc, flavour, mean, std in tw:
        meta.append((name, proc.crop_size["height"], flavour, list(proc.image_mean), list(mean), list(std)))
    out["n_cases"] = np.int64(len(cases))
    out["tower_names"] = np.array([m[0] for m in meta])
    out["tower_sides"] = np.array([m[1] for m in meta], np.int64)
    out["tower_flavours"] = np.array([m[2] for m in meta])
    out["tower_pad_mean"] = np.array([m[3] for m in meta], np.float64)
    out["tower_mean"] = np.array([m[4] for m in meta], np.float64)
    out["tower_std"] = np.array([m[5] for m in meta], np.float64)
    path = os.path.join(OUT, "preprocess_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
