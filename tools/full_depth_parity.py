#!/usr/bin/env python
"""Full-depth bf16 error of the HIP path against its own fp32 instantiation (GPU; VERDICT r4 next #5 i).

The end-to-end oracle comparison (tests/test_release_width_gpu.py) runs 4 decoder layers / depth-1 towers because the CPU oracle
bounds it; the fp32 HIP path is pinned to that oracle at 1.2e-5.  This script uses the fp32 HIP path as the yardstick at the FULL
release depth: 32 decoder layers, 10 in-LLM SVA layers + 3 connector layers, full-depth towers (CLIP-L 24, SigLIP-SO400M 27,
DINOv2-g 40, ConvNeXt-XXL 3/4/30/3), vocabulary 128256, S = 2048, a collator batch of two letter-boxed images — once in fp32, once
in bf16 (the benched dtype) on the SAME weights (the bf16 model loads the fp32 model's state: fp32 masters for what trains, the
decoder's weights rounded to bf16, as the bench line runs them).  Logits max-abs / max, relative L2, least-squares slope; loss;
per-tensor gradient errors of the trainable parameters.  One JSON line -> stdout and profiles-style log (--out)."""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("CAMBRIAN_AMD_RANDOM_INIT", "1")
import torch  # noqa: E402


def build(dev, dt, llm_layers=None):
    from cambrian_amd.model.language_model.cambrian_llama import (CambrianLlamaForCausalLM, apply_release_8b_vision_config,
                                                                llama3_8b_config)
    cfg = llama3_8b_config(**({} if llm_layers is None else {"num_hidden_layers": llm_layers}))
    apply_release_8b_vision_config(cfg)
    depth = cfg.num_hidden_layers
    cfg.num_of_vision_sampler_layers = len([k for k in range(10) if 3 * k < depth])
    cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers, cfg.image_position = 0, 3, 91
    torch.manual_seed(0)
    model = CambrianLlamaForCausalLM(cfg, device=dev, llm_dtype=dt).to(dev)
    for t in model.model.vision_tower_aux_list:
        t._compute_dtype = dt
        t.load_model()
    train_keys = ("mm_projector", "pos_emb", "vision_sampler", "vision_sampler_layers", "vision_query", "image_newline")
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in train_keys))
    return model, cfg


def run(model, cfg, dev, dt, batch):
    out = model(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev),
                position_ids=batch["position_ids"].to(dev), labels=batch["labels"].to(dev),
                images=[i.to(dev, dt) for i in batch["images"]],
                image_aux_attention_masks_list=[m.to(dev) for m in batch["image_aux_attention_masks_list"]],
                image_sizes=batch["image_sizes"])
    out.loss.backward()
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    return out.logits.detach().float().cpu(), float(out.loss.item()), grads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm-layers", type=int, default=None, help="debug: fewer decoder layers")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05", "full_depth_parity.jsonl"))
    a = ap.parse_args()
    from conftest import fit_err
    from cambrian_amd.train.data_layout import synthetic_batch
    dev = torch.device("cuda:0")
    batch = synthetic_batch(2, seed=77, image_position=91, image_sizes=[(336, 200), (224, 336)])
    m32, cfg = build(dev, torch.float32, a.llm_layers)
    with torch.no_grad():
        for n, p in m32.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
        m32.model.image_newline.data = (torch.randn(cfg.hidden_size) / cfg.hidden_size ** 0.5).to(dev)
    state = {k: v.detach().cpu() for k, v in m32.state_dict().items()}
    ref_logits, ref_loss, ref_grads = run(m32, cfg, dev, torch.float32, batch)
    peak32 = torch.cuda.max_memory_allocated() / 2 ** 30
    del m32
    gc.collect()
    torch.cuda.empty_cache()
    m16, cfg = build(dev, torch.bfloat16, a.llm_layers)
    missing = m16.load_state_dict(state, strict=False)
    logits, loss, grads = run(m16, cfg, dev, torch.bfloat16, batch)
    valid = batch["attention_mask"].bool()
    lr, lg = ref_logits[valid], logits[valid]
    slope_err, l2 = fit_err(lg, lr)
    max_rel = float((lg - lr).abs().max() / lr.abs().max())
    per = []
    for n, g in grads.items():
        r = ref_grads[n]
        if r.abs().max() == 0:
            continue
        per.append((float((g - r).abs().max() / r.abs().max()), n))
    per.sort()
    row = {"what": "full_depth_bf16_vs_fp32_hip", "decoder_layers": cfg.num_hidden_layers,
           "in_llm_sva_layers": cfg.num_of_vision_sampler_layers, "towers": "full depth (24 / 27 / 40 / 3-4-30-3)",
           "images": 2, "seq_len": 2048, "vocab": cfg.vocab_size,
           "logits_max_rel": max_rel, "logits_slope_err": slope_err, "logits_rel_l2": l2,
           "loss_fp32": ref_loss, "loss_bf16": loss,
           "grad_tensors": len(per), "grad_max_rel_median": per[len(per) // 2][0], "grad_max_rel_p90": per[int(0.9 * len(per))][0],
           "grad_max_rel_worst": per[-1][0], "grad_worst_tensor": per[-1][1],
           "state_missing": len(missing.missing_keys), "state_unexpected": len(missing.unexpected_keys),
           "peak_hbm_gb_fp32": peak32, "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "a") as f:
        f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
