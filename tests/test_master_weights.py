"""CPU (gloo, world_size 2 for the sharded cases): fp32 master weights for a bf16-compute finetune stage — the reference
up-casts every FSDP parameter to fp32 before sharding and computes in bf16 (train_fsdp.py:1324-1326, fsdp_config.json:6).
``MasterAdamW`` (unsharded), ``Zero2AdamW`` and ``Zero3Unit`` with ``master_dtype=float32`` follow a single-process fp32
AdamW fed the same gradients to 1e-6 over 3 steps at the reference's finetune learning rate (4e-5,
scripts/cambrian/finetune_cambrian_8b.sh) — and an optimizer that steps the bf16 parameters themselves (the round-4
``bench.py --stage finetune``) does NOT: most of its updates round away (VERDICT r4 missing #4)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

LR, WD, STEPS, SHAPES = 4e-5, 0.0, 3, [(64, 48), (48,), (33, 7), (129,)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init_params(dtype):
    g = torch.Generator().manual_seed(0)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.02).to(torch.bfloat16).to(dtype)) for s in SHAPES]


def _grads(world, step, rank):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return [(torch.randn(s, generator=g) * 3e-3).to(torch.bfloat16) for s in SHAPES]


def _mean_grad(world, step):
    """what a bf16 reduce(SUM) / world delivers: one bf16 rounding of the sum, an exact division by 2"""
    tot = None
    for k in range(world):
        g = _grads(world, step, k)
        tot = g if tot is None else [a + b for a, b in zip(tot, g)]
    return [t / world for t in tot]


def _reference(world):
    """single process, fp32 parameters and moments, the bf16 gradients up-cast"""
    params = _init_params(torch.float32)
    opt = torch.optim.AdamW(params, lr=LR, weight_decay=WD)
    for step in range(STEPS):
        for p, g in zip(params, _mean_grad(world, step)):
            p.grad = g.float()
        opt.step()
    return [p.detach().clone() for p in params]


class _Dot(torch.nn.Module):
    """loss = sum_i <p_i, c_i>: the gradient of p_i is exactly c_i (so every implementation sees identical gradients)"""

    def __init__(self, params):
        super().__init__()
        self.ps = torch.nn.ParameterList(params)

    def forward(self, cs):
        return sum((p * c).sum() for p, c in zip(self.ps, cs))


def test_master_adamw_follows_fp32_and_bf16_stepping_does_not():
    from cambrian_amd.train.master import MasterAdamW
    want = _reference(1)
    # (a) masters
    m = _Dot(_init_params(torch.bfloat16))
    opt = MasterAdamW(m.parameters(), lr=LR, weight_decay=WD)
    assert opt.state_bytes() == 12 * sum(p.numel() for p in m.parameters())
    for step in range(STEPS):
        m(_grads(1, step, 0)).backward()
        opt.step()
        opt.zero_grad()
    for mast, p, w in zip(opt.masters, m.ps, want):
        assert mast.dtype == torch.float32 and torch.allclose(mast, w, atol=1e-6, rtol=0)
        assert torch.equal(p.detach(), w.to(torch.bfloat16))          # the compute copy is the master's cast
    # (b) the bug being fixed: AdamW on the bf16 parameters with bf16 moments
    b = _Dot(_init_params(torch.bfloat16))
    start = [p.detach().clone() for p in b.ps]
    bo = torch.optim.AdamW(b.parameters(), lr=LR, weight_decay=WD)
    for step in range(STEPS):
        b(_grads(1, step, 0)).backward()
        bo.step()
        bo.zero_grad()
    moved = sum(int((p.detach() != s).sum()) for p, s in zip(b.ps, start))
    total = sum(p.numel() for p in b.ps)
    should = sum(int((w.to(torch.bfloat16) != s).sum()) for w, s in zip(want, start))
    err_master = max(float((mast.detach() - w).abs().max()) for mast, w in zip(opt.masters, want))
    err_bf16 = max(float((p.detach().float() - w).abs().max()) for p, w in zip(b.ps, want))
    step_size = max(float((w - s.float()).abs().max()) for w, s in zip(want, start))     # ~ 3 x lr
    assert step_size > 5e-5
    assert moved < 0.75 * total                     # a large part of the bf16 weights (every |w| > 2^-7) never moves at lr 4e-5 ...
    assert err_bf16 > 0.5 * step_size               # ... so the parameters miss the fp32 trajectory by about a whole update
    assert err_master < 1e-6 < err_bf16 / 10
    assert should >= 0                              # (how many SHOULD have changed after rounding is not asserted: it is data)


def _worker(rank, world, port, q, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import init_distributed
    init_distributed("gloo")
    want = _reference(world)
    ok = True
    if kind == "zero2":
        from cambrian_amd.train.zero import Zero2AdamW
        m = _Dot(_init_params(torch.bfloat16))
        opt = Zero2AdamW(list(m.parameters()), lr=LR, weight_decay=WD, bucket_mb=0.004)      # 4 KiB buckets -> several
        ok = ok and len(opt.buckets) > 1 and all(b.master_shard is not None and b.master_shard.dtype == torch.float32 for b in opt.buckets)
        ok = ok and opt.state_bytes() == sum(12 * b.shard_len for b in opt.buckets)
        for step in range(STEPS):
            m(_grads(world, step, rank)).backward()
            opt.step()
        for p, w in zip(m.ps, want):
            ok = ok and torch.equal(p.detach(), w.to(torch.bfloat16))
        # the owned master shards against the matching slice of the fp32 trajectory
        for b in opt.buckets:
            flat_w = torch.cat([want[[id(x) for x in m.ps].index(id(p))].reshape(-1) for p in b.params])
            flat_w = torch.cat([flat_w, torch.zeros(b.padded - flat_w.numel())])
            lo = rank * b.shard_len
            ok = ok and torch.allclose(b.master_shard, flat_w[lo:lo + b.shard_len], atol=1e-6, rtol=0)
        # without masters (round-4 behaviour) the same run misses the trajectory
        m2 = _Dot(_init_params(torch.bfloat16))
        o2 = Zero2AdamW(list(m2.parameters()), lr=LR, weight_decay=WD, bucket_mb=0.004, master_dtype=None)
        for step in range(STEPS):
            m2(_grads(world, step, rank)).backward()
            o2.step()
        ok = ok and any(not torch.equal(p.detach(), w.to(torch.bfloat16)) for p, w in zip(m2.ps, want))
        # (a) checkpoint / resume — state_dict() carries the fp32 masters and the moments: a run resumed after one
        # step lands where the uninterrupted run does
        m3 = _Dot(_init_params(torch.bfloat16))
        o3 = Zero2AdamW(list(m3.parameters()), lr=LR, weight_decay=WD, bucket_mb=0.004)
        m3(_grads(world, 0, rank)).backward()
        o3.step()
        sd = o3.state_dict()
        m4 = _Dot(_init_params(torch.bfloat16))
        o4 = Zero2AdamW(list(m4.parameters()), lr=LR, weight_decay=WD, bucket_mb=0.004)
        o4.load_state_dict(sd)
        for step in range(1, STEPS):
            m4(_grads(world, step, rank)).backward()
            o4.step()
        for p, w in zip(m4.ps, want):
            ok = ok and torch.equal(p.detach(), w.to(torch.bfloat16))
        bad = dict(sd, rank=1 - rank)
        try:
            o4.load_state_dict(bad)
            ok = False
        except ValueError:
            pass
        # (b) weights loaded AFTER the optimizer was built: the first step refuses to revert them; resync_masters() adopts them
        m5 = _Dot(_init_params(torch.bfloat16))
        o5 = Zero2AdamW(list(m5.parameters()), lr=LR, weight_decay=WD, bucket_mb=0.004)
        with torch.no_grad():
            for p in m5.ps:
                p.mul_(2.0)
        m5(_grads(world, 0, rank)).backward()
        try:
            o5.step()
            ok = False
        except RuntimeError as e:
            ok = ok and "resync_masters" in str(e)
        o5.resync_masters()
        o5.step()
        ok = ok and all(torch.allclose(b.master_shard, b.param_shard.float(), atol=2e-2, rtol=1e-2) for b in o5.buckets)
        ok = ok and all((p.detach().float().abs() > 0.5 * (2.0 * w0.float().abs()) - 1e-2).all() for p, w0 in zip(m5.ps, _init_params(torch.float32)))
    else:
        from cambrian_amd.train.zero3 import zero3_parameters, zero3_wrap
        m = _Dot(_init_params(torch.bfloat16))
        (u,) = zero3_wrap([m])
        ok = ok and u.master and u.shard.dtype == torch.float32 and u.full.dtype == torch.bfloat16
        opt = torch.optim.AdamW(zero3_parameters([u]), lr=LR, weight_decay=WD)
        for step in range(STEPS):
            m(_grads(world, step, rank)).backward()
            u.finalize()
            opt.step()
            opt.zero_grad()
        for a, w in zip(u.full_state(), want):
            ok = ok and a.dtype == torch.bfloat16 and torch.equal(a, w.to(torch.bfloat16))
        flat_w = torch.cat([w.reshape(-1) for w in want])
        flat_w = torch.cat([flat_w, torch.zeros(u.padded - flat_w.numel())])
        lo = rank * u.shard_len
        ok = ok and torch.allclose(u.shard.detach(), flat_w[lo:lo + u.shard_len], atol=1e-6, rtol=0)
        # the fp32 masters are what a checkpoint keeps — full_state(master=True) is the fp32 trajectory, a new unit
        # takes it (load_full_state) or this rank's shard (state_dict / load_state_dict) without losing precision
        for a, w in zip(u.full_state(master=True), want):
            ok = ok and a.dtype == torch.float32 and torch.allclose(a, w, atol=1e-6, rtol=0)
        m6 = _Dot(_init_params(torch.bfloat16))
        (u6,) = zero3_wrap([m6])
        u6.load_full_state(u.full_state(master=True))
        ok = ok and torch.equal(u6.shard.detach(), u.shard.detach())
        m7 = _Dot(_init_params(torch.bfloat16))
        (u7,) = zero3_wrap([m7])
        u7.load_state_dict(u.state_dict())
        ok = ok and torch.equal(u7.shard.detach(), u.shard.detach())
        for a, b_ in zip(u7.full_state(), u.full_state()):
            ok = ok and torch.equal(a, b_)
        try:
            u7.load_state_dict(dict(u.state_dict(), rank=1 - rank))
            ok = False
        except ValueError:
            pass
        # a frozen unit carries no master
        f = _Dot(_init_params(torch.bfloat16))
        for p in f.parameters():
            p.requires_grad_(False)
        (uf,) = zero3_wrap([f])
        ok = ok and not uf.master and uf.shard.dtype == torch.bfloat16
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["zero2", "zero3"])
def test_sharded_masters_world2_gloo(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
