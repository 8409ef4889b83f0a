"""The eval / generate branch at RELEASE size: Cambrian-8B geometry, random-init weights, a batch of
images of different aspect ratios, a short prompt, greedy decode.  Prints one JSON line (prefill ms, ms per decoded token).
NOT a headline number: it exists so that the dynamic branch has run at the release dimensions on the GPU.

    python tools/bench_generate.py [--batch 2] [--prompt 96] [--new 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CAMBRIAN_AMD_RANDOM_INIT", "1")
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--prompt", type=int, default=96)
    ap.add_argument("--new", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, cfg = bench.build_model(dev)
    model.eval()
    towers = model.get_model().get_vision_tower_aux_list()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1000, 30000, (a.batch, a.prompt), generator=g)
    ids[:, 8] = -200                                                   # IMAGE_TOKEN_INDEX
    sizes = [(336, 336), (336, 224), (200, 400), (1024, 768)][:a.batch] * ((a.batch + 3) // 4)
    sizes = sizes[:a.batch]
    images = [torch.randn(a.batch, 3, t.image_size, t.image_size, generator=g).to(dev, torch.bfloat16) for t in towers]

    def run(n_new):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = model.generate(ids.to(dev), images=images, image_sizes=sizes, max_new_tokens=n_new)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, toks

    run(2)                                                              # warm-up (kernel attributes, allocator)
    t1, _ = run(1)
    tn, toks = run(a.new)
    assert toks.shape == (a.batch, a.new) and bool(torch.isfinite(toks.float()).all())
    print(json.dumps({"what": "generate() at release size (eval branch: per-sample unpad, dynamic SVA hook, KV-cache decode)",
                      "batch": a.batch, "image_sizes": sizes, "prompt_tokens": a.prompt, "new_tokens": a.new,
                      "prefill_plus_first_token_ms": t1, "ms_per_decoded_token": (tn - t1) / max(1, a.new - 1),
                      "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "data": "synthetic, random-init weights",
                      "NOT_HEADLINE": "the north star is the training step"}))


if __name__ == "__main__":
    main()
