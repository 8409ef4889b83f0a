"""fp32 master weights for parameters that compute in bf16 (finetune stage, unsharded).

The reference trains with fp32 parameters and bf16 compute: ``train_fsdp.py:1324-1326`` up-casts every FSDP-managed
parameter to fp32 before sharding, ``fsdp_config.json:6`` sets ``compute_dtype`` bf16.  An optimizer that steps bf16
parameters with bf16 moments (what ``bench.py --stage finetune`` did in round 4) is a different algorithm: at the
reference's finetune learning rate (4e-5, scripts/cambrian/finetune_cambrian_8b.sh) an AdamW update is ~4e-5 in absolute
terms, below half a bf16 ulp of any weight larger than ~1e-2, so most updates round away
(tests/test_master_weights.py shows it).  ``MasterAdamW`` keeps, for every parameter narrower than ``master_dtype``, an fp32
master and fp32 moments (torch's fused AdamW on the masters), feeds it the up-cast gradient and writes the master's cast back
into the compute parameter after the step: 2 (compute) + 2 (gradient) + 4 (master) + 8 (moments) = 16 bytes per parameter, the
accounting SURVEY.md §8d uses for BASELINE configs[3] / [4].  Sharded twins: ``Zero2AdamW(master_dtype=...)`` and
``Zero3Unit(master_dtype=...)`` hold the same state for the owned 1 / world shard only.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch


class MasterAdamW(torch.optim.AdamW):
    """AdamW (decoupled weight decay, torch.optim.AdamW's update rule) over fp32 masters of low-precision parameters;
    parameters already in ``master_dtype`` are stepped in place.  It IS a ``torch.optim.AdamW`` over (masters + full-precision
    parameters) — ``param_groups`` / ``state`` are the real ones, so ``torch.optim.lr_scheduler.*`` and HF ``get_scheduler``
    (the reference's cosine schedule with warm-up) attach to it like to any optimizer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, master_dtype: torch.dtype = torch.float32):
        plist = [p for p in params if p.requires_grad]
        self.low: List[torch.nn.Parameter] = [p for p in plist if p.dtype != master_dtype and p.dtype.is_floating_point]
        self.full: List[torch.nn.Parameter] = [p for p in plist if not (p.dtype != master_dtype and p.dtype.is_floating_point)]
        self.masters = [torch.nn.Parameter(p.detach().to(master_dtype), requires_grad=True) for p in self.low]
        every = self.masters + self.full
        fused = bool(every) and all(t.is_cuda for t in every)
        super().__init__(every, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **({"fused": True} if fused else {}))

    @property
    def inner(self):   # (round-5 name of the wrapped optimizer)
        return self

    def step(self, closure=None):
        live = [(m, p) for m, p in zip(self.masters, self.low) if p.grad is not None]
        if live:
            # one multi-tensor up-cast of the gradients (transient: 4 B per low-precision parameter, returned to the caching
            # allocator after the step — the same peak as persistent buffers would pin for good)
            g32 = [torch.empty_like(m) for m, _ in live]
            torch._foreach_copy_(g32, [p.grad for _, p in live])
            for (m, _), g in zip(live, g32):
                m.grad = g
        out = super().step(closure)
        if live:
            torch._foreach_copy_([p.data for _, p in live], [m.data for m, _ in live])   # compute copy = the master's cast
            for m, _ in live:
                m.grad = None
        return out

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.low + self.full:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        for m in self.masters:
            m.grad = None

    def state_bytes(self) -> int:
        """master copies + two moments per stepped element"""
        return (sum(m.numel() * m.element_size() * 3 for m in self.masters)
                + sum(p.numel() * p.element_size() * 2 for p in self.full))

    def state_dict(self) -> dict:
        sd = super().state_dict()
        sd["masters"] = [m.detach().clone() for m in self.masters]
        return sd

    def load_state_dict(self, sd: dict) -> None:
        sd = dict(sd)
        masters = sd.pop("masters")
        if "inner" in sd:   # a round-5 checkpoint
            sd = sd["inner"]
        if len(masters) != len(self.masters) or any(t.shape != m.shape for t, m in zip(masters, self.masters)):
            raise ValueError(f"MasterAdamW: the checkpoint holds {len(masters)} masters, this optimizer {len(self.masters)} "
                             "(or their shapes differ): other parameters, or another order")
        super().load_state_dict(sd)
        with torch.no_grad():
            for m, t, p in zip(self.masters, masters, self.low):
                m.copy_(t)
                p.copy_(m)

    def resync_masters(self) -> None:
        """Masters := the parameters as they are now (weights loaded into the module after the optimizer was built)."""
        with torch.no_grad():
            for m, p in zip(self.masters, self.low):
                m.copy_(p)
