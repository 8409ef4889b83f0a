"""Times the image pre-processing launch pair (csrc/preprocess.hip) on one MI355X and the reference's CPU expression
beside it.  One JSON line: kernel time (HIP events on the launch stream), end-to-end time of the host call (table
build + pinned pack + H2D + kernels), HBM roofline of the pair from the algorithmic bytes, CPU baseline images/s.

    python tools/bench_preprocess.py [--batch 16] [--iters 20] [--cpu-images 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cambrian_amd.train import image_pipeline as IP  # noqa: E402

OPENAI_MEAN, OPENAI_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
SPECS = [IP.TowerPixelSpec(384, OPENAI_MEAN, (0.5,) * 3, (0.5,) * 3, "torchvision"),
         IP.TowerPixelSpec(336, OPENAI_MEAN, OPENAI_MEAN, OPENAI_STD, "hf"),
         IP.TowerPixelSpec(378, IMAGENET_MEAN, IMAGENET_MEAN, IMAGENET_STD, "hf"),
         IP.TowerPixelSpec(1024, OPENAI_MEAN, OPENAI_MEAN, OPENAI_STD, "torchvision")]
# (w, h) mix of a web-image corpus: VGA/COCO-like, portrait, large photo, small icon, already square
SIZES = [(640, 480), (480, 640), (500, 375), (1024, 768), (1333, 800), (336, 336), (224, 224), (1920, 1080)]


def cpu_reference(images):
    """the reference expression, per image and tower, on one core (train_fsdp.py:1004-1008)"""
    from PIL import Image
    luts = [sp.lut() for sp in SPECS]
    t0 = time.perf_counter()
    for im in images:
        pil = Image.fromarray(im)
        w, h = pil.size
        for sp, lut in zip(SPECS, luts):
            sq = pil
            if w != h:
                s = max(w, h)
                sq = Image.new("RGB", (s, s), tuple(int(x * 255) for x in sp.pad_mean))
                sq.paste(pil, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
            u8 = np.asarray(sq.resize((sp.out_side, sp.out_side)))
            x = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)   # ToTensor
            m = torch.tensor(sp.mean).view(3, 1, 1)
            s_ = torch.tensor(sp.std).view(3, 1, 1)
            x.sub_(m).div_(s_)                                                                          # Normalize
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-images", type=int, default=8)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    images = [rng.integers(0, 256, (SIZES[i % len(SIZES)][1], SIZES[i % len(SIZES)][0], 3), dtype=np.uint8)
              for i in range(args.batch)]
    dev = torch.device("cuda:0")
    pre = IP.GpuImagePreprocessor(SPECS, dev, torch.bfloat16)
    for _ in range(3):
        pre(images)
    torch.cuda.synchronize()
    plan = pre.last_plan
    # kernels alone: replay the launch on resident buffers
    lib = pre._lib
    slot = pre._slots[(pre._turn - 1) % len(pre._slots)]
    sec, off = {}, 0
    for name, nb in (("jobs", plan.jobs.nbytes), ("bounds", plan.bounds.nbytes), ("coefs", plan.coefs.nbytes),
                     ("lut", plan.lut.nbytes), ("src", plan.src_bytes)):
        sec[name] = off
        off = IP._align(off + nb, 256)
    out = torch.empty(plan.out_elems, dtype=torch.bfloat16, device=dev)
    base = slot.blob.data_ptr()
    st = torch.cuda.current_stream(dev)

    def launch():
        rc = lib.cmb_image_preprocess(base + sec["jobs"], plan.jobs.ctypes.data, len(plan.jobs), base + sec["src"],
                                      base + sec["bounds"], base + sec["coefs"], base + sec["lut"], 0,
                                      slot.tmp.data_ptr(), out.data_ptr(), st.cuda_stream)
        assert rc == 0, rc
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.iters):
        launch()
    e1.record(st)
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / args.iters
    t0 = time.perf_counter()
    for _ in range(args.iters):
        pre(images)
    torch.cuda.synchronize()
    e2e_ms = (time.perf_counter() - t0) * 1e3 / args.iters
    src_bytes = sum(im.size for im in images)
    out_bytes = sum(3 * sp.out_side ** 2 * 2 for sp in SPECS) * len(images)
    tmp_bytes = 2 * plan.tmp_bytes
    cpu_s = cpu_reference(images[:args.cpu_images])
    res = {"metric": "preprocessed images/sec (4 towers)", "batch": len(images),
           "kernel_pair_ms": round(k_ms, 4), "host_call_ms": round(e2e_ms, 3),
           "value_kernels": round(len(images) / k_ms * 1e3, 1), "value_host_call": round(len(images) / e2e_ms * 1e3, 1),
           "roofline": {"bound": "hbm", "achieved": round((src_bytes + out_bytes) / k_ms / 1e6, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round((src_bytes + out_bytes) / k_ms / 1e6 / 8000.0, 4),
                        "algorithmic_bytes": src_bytes + out_bytes, "intermediate_bytes": tmp_bytes},
           "h2d_bytes": plan.src_bytes + plan.jobs.nbytes + plan.bounds.nbytes + plan.coefs.nbytes,
           "reference_h2d_bytes": sum(3 * sp.out_side ** 2 * 4 for sp in SPECS) * len(images),
           "cpu_baseline": {"value": round(args.cpu_images / cpu_s, 2), "unit": "images/sec", "cores": 1,
                            "kind": "reference", "sample": f"{args.cpu_images} images x 4 towers, Pillow resize + "
                            "ToTensor/Normalize per train_fsdp.py:1004-1008"}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
