"""How far is the bf16 HIP path from (A) the fp32 oracle and (B) the same oracle with every parameter and input rounded
to bf16 first (rounding-matched at the storage level: what remains is activation rounding + accumulation order)?
Prints the observed max-abs/max-abs errors of logits and of every trainable gradient (tests/test_model_gpu.py config)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402
from conftest import rel_err  # noqa: E402


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


dev = torch.device("cuda:0")
dt = torch.bfloat16
from cambrian_amd.train.data_layout import synthetic_batch  # noqa: E402
for qn in (None, [16, 4]):
    model, cfg, towers = T._build(dev, dt, MP(), query_nums=qn)
    batch = synthetic_batch(2, seq_len=T.S, image_position=T.P0, image_token_len=16, aux_token_lens=[16, 64],
                            image_res=[56, 64], image_sizes=[(336, 336), (336, 150)], vocab_lo=1, vocab_hi=300)
    out = model(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev),
                position_ids=batch["position_ids"].to(dev), labels=batch["labels"].to(dev),
                images=[i.to(dev, dt) for i in batch["images"]],
                image_aux_attention_masks_list=[m.to(dev) for m in batch["image_aux_attention_masks_list"]],
                image_sizes=batch["image_sizes"])
    logits = out.logits.float().cpu()
    out.loss.backward()
    grads = {n: q.grad.float().cpu() for n, q in model.named_parameters() if q.requires_grad}
    for label, rnd in (("fp32 oracle", False), ("oracle on bf16-rounded params+inputs", True)):
        if rnd:
            sd = model.state_dict()
            backup = {k: v.clone() for k, v in sd.items()}
            for k, v in sd.items():
                if v.is_floating_point():
                    v.copy_(v.to(torch.bfloat16).to(v.dtype))
            canon_bak = [dict(t.canon) for t in towers]
            for t in towers:
                t.canon = {k: v.to(torch.bfloat16).float() for k, v in t.canon.items()}
            b2 = dict(batch)
            b2["images"] = [i.to(torch.bfloat16).float() for i in batch["images"]]
        else:
            b2 = batch
        ref_loss, ref_logits, p = T._oracle_run(model, cfg, towers, b2)
        ref_loss.backward()
        worst = ("", 0.0)
        for n, g in grads.items():
            gr = p[n].grad
            if gr is None or gr.abs().max() == 0:
                continue
            e = rel_err(g, gr)
            if e > worst[1]:
                worst = (n, e)
        print(f"groups={qn} {label}: logits {rel_err(logits, ref_logits):.3e} loss {abs(out.loss.item()-ref_loss.item()):.3e} "
              f"worst grad {worst[1]:.3e} ({worst[0]})", flush=True)
        if rnd:
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(backup[k])
            for t, c in zip(towers, canon_bak):
                t.canon = c
