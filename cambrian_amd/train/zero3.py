"""ZeRO-3 parameter sharding over RCCL/xGMI (BASELINE config 5; the reference trains with XLA-FSDP
``full_shard`` wrapping every decoder layer: fsdp_config.json:1-11, train_fsdp.py:1267-1398; DeepSpeed twin:
scripts/zero3.json).

One ``Zero3Unit`` per wrapped module (a decoder layer).  Each rank owns 1/world of the unit's flat parameter vector;
the full vector lives in ONE persistent flat tensor whose *storage* is sized only while the unit computes:
    forward:   pre-hook  all_gather(shards) -> full   ...   post-hook  storage.resize_(0)
    backward:  pre-hook  all_gather again             ...   when every trainable parameter of the unit has its gradient
               (post-accumulate hooks; frozen units: the module's backward hook): gradients are flattened,
               reduce_scatter(SUM)/world leaves the owned shard's gradient on ``unit.shard.grad``, storage.resize_(0).
The module's parameters are views of that flat tensor throughout (autograd's saved weights included), which is what lets
the storage be dropped and refilled under them — the FSDP trick.  The optimizer steps on ``unit.shard`` only
(``zero3_parameters``), so parameter, gradient and Adam-state memory all scale 1/world; a frozen unit (the LLM in the
pre-training stage, train_fsdp.py:1677-1685) is sharded for memory and never produces a gradient.
Collectives are per unit (one decoder layer of Yi-34B = 0.55 B parameters = 1.1 GB bf16: few, large messages for the
point-to-point xGMI links).  CPU coverage: tests/test_zero3.py, gloo world 2.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class Zero3Unit:
    def __init__(self, module: nn.Module, process_group: Optional[dist.ProcessGroup] = None):
        self.module = module
        self.group = process_group
        on = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if on else 1
        self.rank = dist.get_rank(process_group) if on else 0
        self.params: List[nn.Parameter] = [p for p in module.parameters()]
        if not self.params:
            raise ValueError("Zero3Unit needs a module with parameters")
        # the decoder blocks cache fused / transposed copies of FROZEN weights on their sub-modules
        # (cambrian_llama.py::_fused_frozen_weight, _frozen_transposed): under ZeRO-3 those copies would stay resident
        # after release() and undo the sharding, so sharded modules opt out of them
        for sub in module.modules():
            sub.__dict__["_cmb_no_weight_cache"] = True
            for k in [k for k in sub.__dict__ if k.startswith("_w_")]:
                del sub.__dict__[k]
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("all parameters of a Zero3Unit must share device and dtype")
        n = sum(p.numel() for p in self.params)
        self.shard_len = (n + self.world - 1) // self.world
        self.padded = self.shard_len * self.world
        self.full = torch.zeros(self.padded, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            self.full[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.full[off:off + p.numel()].view(p.shape)   # views of the flat tensor from now on
            off += p.numel()
        lo = self.rank * self.shard_len
        self.trainable = any(p.requires_grad for p in self.params)
        self.shard = nn.Parameter(self.full[lo:lo + self.shard_len].clone(), requires_grad=self.trainable)
        self._nbytes = self.full.untyped_storage().nbytes()
        self._resident = True
        self._pending = 0
        self.release()
        module.register_forward_pre_hook(lambda m, a: self.gather())
        module.register_forward_hook(self._post_forward)
        if self.trainable:
            for p in self.params:
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._on_grad)

    # A plain tensor hook on the unit's output marks the start of its backward.  (Module backward hooks would wrap the
    # output in a custom-Function view, which the in-LLM SVA hook — an in-place scatter into the layer's output,
    # cambrian_llama.py:181-207 — is not allowed to modify.)  A frozen unit has no gradient hook to release it: it is
    # dropped when the next unit's backward starts, the last one by finalize().
    _bw_prev: Optional["Zero3Unit"] = None

    def _post_forward(self, module, args, out):
        self.release()
        t = out if torch.is_tensor(out) else next((o for o in out if torch.is_tensor(o)), None)
        if t is not None and t.requires_grad:
            t.register_hook(self._grad_of_output)

    def _grad_of_output(self, grad):
        prev = Zero3Unit._bw_prev
        if prev is not None and prev is not self and not prev.trainable and prev.resident:
            prev.release()
        Zero3Unit._bw_prev = self
        self._pre_backward()
        return grad

    # ---- residency -----------------------------------------------------------------------------------------------
    def gather(self) -> None:
        if self._resident:
            return
        self.full.untyped_storage().resize_(self._nbytes)
        if self.world > 1:
            dist.all_gather_into_tensor(self.full, self.shard.data, group=self.group)
        else:
            self.full[: self.shard_len].copy_(self.shard.data)
        self._resident = True

    def release(self) -> None:
        if self._resident:
            self.full.untyped_storage().resize_(0)
            self._resident = False

    @property
    def resident(self) -> bool:
        return self._resident

    # ---- backward ------------------------------------------------------------------------------------------------
    def _pre_backward(self) -> None:
        self.gather()
        self._pending = sum(1 for p in self.params if p.requires_grad)

    def _on_grad(self, p: nn.Parameter) -> None:
        self._pending -= 1
        if self._pending == 0:
            self._reduce_grads()

    def _reduce_grads(self) -> None:
        flat = torch.zeros(self.padded, device=self.full.device, dtype=self.full.dtype)
        off = 0
        for p in self.params:
            if p.grad is not None:
                flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = None
            off += p.numel()
        if self.world > 1:
            g = torch.empty(self.shard_len, device=flat.device, dtype=flat.dtype)
            dist.reduce_scatter_tensor(g, flat, op=dist.ReduceOp.SUM, group=self.group)
            g.div_(self.world)
        else:
            g = flat[: self.shard_len].clone()
        self.shard.grad = g if self.shard.grad is None else self.shard.grad + g
        self.release()

    def finalize(self) -> None:
        """After backward(), before optimizer.step(): a trainable unit some of whose parameters received no gradient in
        this step (unused branch) never counts down to zero in ``_on_grad`` — reduce what arrived (missing gradients
        enter the collective as zeros, every rank calls this, so the collectives stay matched) and drop the storage."""
        if self.trainable and self._pending > 0:
            self._pending = 0
            self._reduce_grads()
        elif self._resident:
            self.release()
        if Zero3Unit._bw_prev is self:
            Zero3Unit._bw_prev = None

    def full_state(self) -> List[torch.Tensor]:
        """Every parameter, materialised (for checkpointing / tests)."""
        self.gather()
        out = [p.detach().clone() for p in self.params]
        self.release()
        return out


def zero3_wrap(modules: Iterable[nn.Module], process_group: Optional[dist.ProcessGroup] = None) -> List[Zero3Unit]:
    return [Zero3Unit(m, process_group) for m in modules]


def zero3_finalize(units: Iterable[Zero3Unit]) -> None:
    """Call between ``loss.backward()`` and ``optimizer.step()`` (see Zero3Unit.finalize)."""
    for u in units:
        u.finalize()


def zero3_parameters(units: Iterable[Zero3Unit]) -> List[nn.Parameter]:
    """What the optimizer steps on: the owned shard of every trainable unit."""
    return [u.shard for u in units if u.trainable]
