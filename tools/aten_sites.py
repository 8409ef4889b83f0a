"""Which repo lines issue the stray ATen kernels of one bench step (copies, fills, adds, casts): a TorchDispatchMode logs every
aten copy_ / clone / fill_ / zero_ / add / mul / _to_copy / cat on a CUDA tensor of >= --min-numel elements with its innermost
repo frame (works inside autograd's backward too: python custom Functions carry their own stack).  GPU box, repo root:
    python tools/aten_sites.py --batch 24 > gpurun_out/r06/aten_sites.txt
Not product code."""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

WATCH = ("copy_", "clone", "fill_", "zero_", "add", "add_", "mul", "mul_", "_to_copy", "cat", "zeros", "zeros_like", "sum", "div", "div_",
         "masked_fill", "index", "where", "contiguous", "empty_strided", "new_zeros")


class Log(TorchDispatchMode):
    def __init__(self, min_numel):
        super().__init__()
        self.rows = collections.Counter()
        self.bytes = collections.Counter()
        self.min_numel = min_numel

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is not None and t.is_cuda and t.numel() >= self.min_numel:
                site = "(no repo frame)"
                for fr in reversed(traceback.extract_stack()):
                    if fr.filename.startswith(ROOT) and "tools/aten_sites" not in fr.filename:
                        site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                        break
                key = (name, site, tuple(t.shape), str(t.dtype).replace("torch.", ""))
                self.rows[key] += 1
                self.bytes[key] += t.numel() * t.element_size()
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--min-numel", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from cambrian_amd.train.data_layout import synthetic_batch
    from cambrian_amd.train.dp import GradSync
    model, cfg = bench.build_model(dev, None, "8b")
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.0, fused=True)
    sync = GradSync(params)
    b = synthetic_batch(args.batch, seed=1234, image_position=cfg.image_position)
    kw = dict(input_ids=b["input_ids"].to(dev), labels=b["labels"].to(dev), position_ids=b["position_ids"].to(dev),
              attention_mask=None, images=[i.to(dev, torch.bfloat16) for i in b["images"]],
              image_aux_attention_masks_list=[m.to(dev) for m in b["image_aux_attention_masks_list"]],
              image_sizes=b["image_sizes"])

    def step():
        out = model(**kw)
        out.loss.backward()
        sync.finish()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with Log(args.min_numel) as log:
        out = model(**kw)
        out.loss.backward()
    torch.cuda.synchronize()
    print("count   MB_total  op  shape dtype  site")
    for key, n in sorted(log.rows.items(), key=lambda kv: -log.bytes[kv[0]]):
        print(f"{n:5d} {log.bytes[key] / 1e6:10.1f}  {key[0]:10s} {str(key[2]):28s} {key[3]:9s} {key[1]}")


if __name__ == "__main__":
    main()
