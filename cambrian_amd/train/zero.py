"""ZeRO-2 for the trainable parameters over RCCL/xGMI (BASELINE config 4; DeepSpeed config in the
reference: scripts/zero2.json:1-22 — gradient partitioning + optimizer-state partitioning, parameters replicated).

Layout: parameters are re-homed into a few flat fp32 *buckets* (default 64 MiB; xGMI is point-to-point, so few large
collectives), each padded to a multiple of the world size and cut into ``world`` equal shards; rank r owns shard r of
every bucket.  Per step
  * backward: a bucket's gradients are ADDED into its flat gradient buffer by post-accumulate hooks (so micro-batches
    accumulate, as ``--gradient_accumulation_steps`` of the reference launch scripts needs); on the LAST micro-batch of
    a step — every backward outside ``no_sync()`` — the arrival of a bucket's last gradient launches an async
    ``reduce_scatter`` (RCCL's own stream) that delivers the SUM of the owned shard and overlaps the rest of backward;
  * ``step()``: waits, turns sums into means, runs AdamW (torch's fused multi-tensor kernel) on the owned shards only —
    exp_avg / exp_avg_sq exist for 1/world of the parameters — and ``all_gather``s the updated shards straight into the
    flat parameter buckets, of which the module's parameters are views (no copy back).
With world == 1 it degenerates to AdamW over flat buckets.  CPU coverage: tests/test_zero.py (gloo, world_size 2) checks
bit-for-bit agreement of the updated parameters with an unsharded AdamW on the averaged gradients.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _ZBucket:
    def __init__(self, params: List[torch.nn.Parameter], world: int, rank: int, master_dtype: Optional[torch.dtype] = None):
        self.params = params
        dev, dt = params[0].device, params[0].dtype
        n = sum(p.numel() for p in params)
        self.shard_len = (n + world - 1) // world
        self.padded = self.shard_len * world
        self.flat_param = torch.zeros(self.padded, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(self.padded, device=dev, dtype=dt)
        self.grad_views = []
        off = 0
        for p in params:
            v = self.flat_param[off:off + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v  # the module's parameter now lives in the bucket
            self.grad_views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        lo = rank * self.shard_len
        self.param_shard = self.flat_param[lo:lo + self.shard_len]
        self.grad_shard = torch.zeros(self.shard_len, device=dev, dtype=dt)
        # fp32 master of the OWNED shard when the bucket computes in a narrower type (the reference up-casts every FSDP
        # parameter to fp32 before sharding and computes in bf16: train_fsdp.py:1324-1326, fsdp_config.json:6): AdamW steps on
        # the master, the compute-dtype shard is its cast, refreshed after every step and all-gathered from there
        self.master_shard = None
        if master_dtype is not None and dt != master_dtype and dt.is_floating_point:
            self.master_shard = self.param_shard.detach().to(master_dtype).clone()
        self.pending = len(params)
        self.work = None
        self.dirty = False  # flat_grad holds something since it was last zeroed (set by every accumulation)


class Zero2AdamW:
    """Sharded-gradient, sharded-state AdamW (decoupled weight decay, same update rule as torch.optim.AdamW)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, bucket_mb: float = 64.0, process_group: Optional[dist.ProcessGroup] = None,
                 master_dtype: Optional[torch.dtype] = torch.float32):
        """``master_dtype``: parameters of a narrower floating type (a bf16 decoder in the finetune stage) get an fp32 master
        copy of the owned shard, fp32 AdamW moments and a cast back per step — 16 B per parameter / world, the accounting of
        BASELINE configs[3] / [4]; ``None`` steps in the parameter's own dtype (the round-4 behaviour: at lr 4e-5 most
        updates of a bf16 parameter vanish below its ulp, tests/test_master_weights.py)."""
        self.group = process_group
        on = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if on else 1
        self.rank = dist.get_rank(process_group) if on else 0
        plist = list(reversed([p for p in params if p.requires_grad]))  # ~ the order autograd finishes them
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets: List[_ZBucket] = []
        cur, cur_bytes = [], 0
        for p in plist:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(_ZBucket(cur, self.world, self.rank, master_dtype))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(_ZBucket(cur, self.world, self.rank, master_dtype))
        # optimizer state exists for the owned shards only
        self._shards = [torch.nn.Parameter(b.param_shard if b.master_shard is None else b.master_shard, requires_grad=True)
                        for b in self.buckets]
        fused = self._shards[0].is_cuda if self._shards else False
        self.inner = torch.optim.AdamW(self._shards, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                       **({"fused": True} if fused else {}))
        self._where = {}
        self._handles = []
        self._sync = True  # False inside no_sync(): accumulate only, no collective
        self._masters_checked = False   # first step(): masters still equal the parameters? (resync_masters / load_state_dict)
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---- backward side -------------------------------------------------------------------------------------------
    def _launch(self, b: _ZBucket) -> None:
        if self.world > 1:
            b.work = dist.reduce_scatter_tensor(b.grad_shard, b.flat_grad, op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True)
        else:
            b.grad_shard.copy_(b.flat_grad[: b.shard_len])

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b, i = self._where[p]
        if b.work is not None or b.pending <= 0:
            raise RuntimeError("Zero2AdamW: a gradient arrived after this step's reduce-scatter was launched; run every "
                               "micro-batch but the last one of a step under `with opt.no_sync():`")
        b.grad_views[i].add_(p.grad)  # accumulate (flat_grad is zeroed by step())
        b.dirty = True
        p.grad = None  # the full gradient is not kept: ZeRO-2 holds 1/world of it after the reduce-scatter
        if not self._sync:
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def no_sync(self):
        """Context manager for gradient accumulation (same contract as DistributedDataParallel.no_sync): backward passes
        inside it only accumulate into the flat gradient buckets; the first backward outside it reduces."""
        opt = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev, opt._sync = opt._sync, False

            def __exit__(self_inner, *exc):
                opt._sync = self_inner.prev

        return _NoSync()

    # ---- optimizer side ------------------------------------------------------------------------------------------
    def step(self) -> None:
        if not self._masters_checked:
            self._check_masters()
        gathers = []
        for b, s in zip(self.buckets, self._shards):
            if b.work is None and (b.pending != 0 or self.world == 1):
                # some parameter got no gradient on the last micro-batch (its slot enters the collective as whatever the
                # earlier micro-batches accumulated, 0 if none), or the whole step ran under no_sync()
                for i, p in enumerate(b.params):
                    if p.grad is not None:
                        b.grad_views[i].add_(p.grad)
                        b.dirty = True
                        p.grad = None
                self._launch(b)
            if b.work is not None:
                b.work.wait()
                b.work = None
            g = b.grad_shard if self.world == 1 else b.grad_shard.div_(self.world)
            s.grad = g if b.master_shard is None else g.to(b.master_shard.dtype)
        self.inner.step()
        for b, s in zip(self.buckets, self._shards):
            if b.master_shard is not None:
                b.param_shard.copy_(b.master_shard)   # the compute-dtype shard = the master's cast
                s.grad = None                           # (the up-cast gradient copy is per step)
            if self.world > 1:
                gathers.append(dist.all_gather_into_tensor(b.flat_param, b.param_shard, group=self.group, async_op=True))
            b.flat_grad.zero_()
            b.dirty = False
            b.pending = len(b.params)
        for w in gathers:
            w.wait()

    def zero_grad(self, set_to_none: bool = True) -> None:
        """Drops this step's gradients INCLUDING what the hooks already accumulated / launched: after a skipped step
        (non-finite loss followed by ``zero_grad()``, an exception between backward and step) the next backward starts
        from zero instead of adding to stale sums or tripping the "gradient after the reduce-scatter" check."""
        for b in self.buckets:
            if b.work is not None:       # a launched reduce-scatter must finish before its buffers are reused
                b.work.wait()
                b.work = None
            if b.dirty:                  # `pending` does not move under no_sync(): the flag is set by every accumulation
                b.flat_grad.zero_()
                b.dirty = False
            b.pending = len(b.params)
            for p in b.params:
                p.grad = None
        for s_ in self._shards:
            s_.grad = None

    # ---- masters and checkpoints ---------------------------------------------------------------------------
    # The fp32 master of the owned shard is a SNAPSHOT taken when the optimizer is built; step() overwrites the compute-dtype
    # shard with the master's cast.  Weights loaded into the module afterwards (a checkpoint, resize_token_embeddings, a
    # re-initialisation) must therefore be handed to the masters — ``resync_masters()`` — or they are reverted by the first
    # step.  Build the optimizer AFTER loading weights where possible; the first step() checks the two agree and raises.
    def resync_masters(self) -> None:
        """Masters := the parameters as they are now (after loading weights into a module whose optimizer already exists)."""
        for b in self.buckets:
            if b.master_shard is not None:
                b.master_shard.copy_(b.param_shard)
        self._masters_checked = True

    def _check_masters(self) -> None:
        for i, b in enumerate(self.buckets):
            if b.master_shard is not None and not torch.equal(b.master_shard.to(b.param_shard.dtype), b.param_shard):
                raise RuntimeError(f"Zero2AdamW: bucket {i}'s parameters changed after the optimizer was built (weights loaded "
                                   "into the module afterwards?); call resync_masters() or they would be reverted by this step")
        self._masters_checked = True

    def state_dict(self) -> dict:
        """This rank's state: AdamW moments of the owned shards, the fp32 masters, and the sharding it belongs to."""
        return {"inner": self.inner.state_dict(), "world": self.world, "rank": self.rank,
                "shard_len": [b.shard_len for b in self.buckets],
                "masters": [None if b.master_shard is None else b.master_shard.detach().clone() for b in self.buckets],
                "param_shards": [b.param_shard.detach().clone() for b in self.buckets]}

    def load_state_dict(self, sd: dict) -> None:
        """Restores what ``state_dict()`` of the SAME sharding (world, rank, bucket layout) saved; the module's parameters are
        rewritten from the masters (all ranks must call it)."""
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError(f"Zero2AdamW state of rank {sd['rank']} / world {sd['world']} loaded on rank {self.rank} / world {self.world}")
        if list(sd["shard_len"]) != [b.shard_len for b in self.buckets]:
            raise ValueError("Zero2AdamW: the checkpoint's bucket layout differs (other parameters, bucket_mb or world size)")
        for b, m, ps in zip(self.buckets, sd["masters"], sd["param_shards"]):
            if (m is None) != (b.master_shard is None):
                raise ValueError("Zero2AdamW: checkpoint and optimizer disagree on which buckets carry fp32 masters")
            if m is not None:
                if m.shape != b.master_shard.shape:
                    raise ValueError("Zero2AdamW: master shard shape mismatch")
                b.master_shard.copy_(m)
                b.param_shard.copy_(b.master_shard)
            else:
                b.param_shard.copy_(ps)
        self.inner.load_state_dict(sd["inner"])
        gathers = [dist.all_gather_into_tensor(b.flat_param, b.param_shard, group=self.group, async_op=True)
                   for b in self.buckets] if self.world > 1 else []
        for w in gathers:
            w.wait()
        self._masters_checked = True

    def state_bytes(self) -> int:
        """optimizer-state bytes held by THIS rank: two moments per owned element in the stepping dtype (+ the master copy)."""
        tot = 0
        for b in self.buckets:
            step_t = b.param_shard if b.master_shard is None else b.master_shard
            tot += 2 * b.shard_len * step_t.element_size() + (0 if b.master_shard is None else b.shard_len * b.master_shard.element_size())
        return tot

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
