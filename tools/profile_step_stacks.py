"""Which Python call sites launch the non-GEMM kernels of one bench step (torch.profiler, with_stack): for every GPU
kernel whose name matches --match (default: copy / fill / add / clone / index / cat kernels — the ones the rocprofv3
table cannot attribute), the summed device time per (kernel name, innermost repo frame).  GPU box, repo root:
    python tools/profile_step_stacks.py [--batch 16] [--llm-layers N] [--match REGEX] > gpurun_out/stacks.txt
Not product code; the numbers behind DESIGN.md's "avoidable non-GEMM time" list."""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--llm-layers", type=int, default=None)
    ap.add_argument("--match", default=r"copyBuffer|Memcpy|Memset|FillFunctor|CUDAFunctor_add|direct_copy|bfloat16_copy|"
                                       r"CatArray|index|elementwise_kernel|fillBuffer")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from cambrian_amd.train.data_layout import synthetic_batch
    from cambrian_amd.train.dp import GradSync
    model, cfg = bench.build_model(dev, args.llm_layers, "8b")
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
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
        step()
        torch.cuda.synchronize()
    pat = re.compile(args.match)
    ev = prof.events()
    # device kernels carry no stack; their launching CPU op does: link through the correlation of kernel -> cpu parent
    per = collections.defaultdict(lambda: [0.0, 0])
    total = collections.defaultdict(lambda: [0.0, 0])
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CUDA:
            continue
        name = e.name
        total[name][0] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
        total[name][1] += 1
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA:
            continue
        ks = getattr(e, "kernels", None) or []
        if not ks:
            continue
        stack = [f for f in (e.stack or []) if ROOT in f]
        site = stack[0].replace(ROOT + "/", "") if stack else "(no repo frame) " + e.name
        for k in ks:
            if pat.search(k.name):
                per[(k.name[:70], site)][0] += k.duration
                per[(k.name[:70], site)][1] += 1
    print("== kernels matching %r by call site (device us per step, launches)" % args.match)
    for (k, site), (us, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{us:10.1f} us {n:5d}  {k:70s}  {site}")
    print("\n== all device kernels of the step (us, launches)")
    for k, (us, n) in sorted(total.items(), key=lambda kv: -kv[1][0])[:70]:
        print(f"{us:10.1f} us {n:5d}  {k[:110]}")


if __name__ == "__main__":
    main()
