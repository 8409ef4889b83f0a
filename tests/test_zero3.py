"""CPU, world_size 2, gloo: ZeRO-3 units (cambrian_amd/train/zero3.py) — parameters sharded, gathered only around each
unit's forward / backward, gradients reduce-scattered — reproduce unsharded AdamW training on the rank-averaged
gradients; full parameter storage is released between uses; a frozen unit is sharded and never gets a gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _blocks():
    torch.manual_seed(0)
    return [torch.nn.Sequential(torch.nn.Linear(16, 33), torch.nn.GELU()),
            torch.nn.Sequential(torch.nn.Linear(33, 33), torch.nn.LayerNorm(33)),      # frozen in the test
            torch.nn.Sequential(torch.nn.Linear(33, 7))]


def _data(world, step):
    return [torch.randn(8, 16, generator=torch.Generator().manual_seed(100 * step + k)) for k in range(world)]


def _run(blocks, x):
    for b in blocks:
        x = b(x)
    return x.pow(2).mean()


def _reference(world, steps):
    blocks = _blocks()
    for p in blocks[1].parameters():
        p.requires_grad_(False)
    params = [p for b in blocks for p in b.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.1)
    for step in range(steps):
        grads = None
        for k in range(world):
            for p in params:
                p.grad = None
            _run(blocks, _data(world, step)[k]).backward()
            g = [p.grad.clone() for p in params]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(params, grads):
            p.grad = g / world
        opt.step()
    return [[p.detach().clone() for p in b.parameters()] for b in blocks]


EXPECTED_LOG = [  # one step of the 3-unit chain (unit 1 frozen) with prefetch: each neighbour's all-gather is issued while
    # the current unit is resident, and consumed by a wait instead of a blocking gather
    ("gather", 0), ("prefetch", 1), ("release", 0), ("wait", 1), ("prefetch", 2), ("release", 1), ("wait", 2), ("release", 2),
    ("gather", 2), ("prefetch", 1), ("release", 2), ("wait", 1), ("prefetch", 0), ("release", 1), ("wait", 0), ("release", 0)]


def _worker(rank, world, port, q, prefetch=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import init_distributed
    from cambrian_amd.train.zero3 import zero3_parameters, zero3_wrap
    init_distributed("gloo")
    blocks = _blocks()
    for p in blocks[1].parameters():
        p.requires_grad_(False)
    units = zero3_wrap(blocks, prefetch=prefetch)
    log = []
    for u in units:
        u.log = log
    ok = all(not u.resident for u in units)                       # nothing is resident outside a forward / backward
    ok = ok and units[0].shard.numel() * world >= sum(p.numel() for p in blocks[0].parameters())
    opt = torch.optim.AdamW(zero3_parameters(units), lr=1e-2, weight_decay=0.1)
    steps = 3
    for step in range(steps):
        del log[:]
        _run(blocks, _data(world, step)[rank]).backward()
        ok = ok and all(not u.resident for u in units)
        ok = ok and units[1].shard.grad is None and not units[1].trainable
        if prefetch:
            ok = ok and log == EXPECTED_LOG
        else:
            ok = ok and not any(w in ("prefetch", "wait") for w, _ in log)
        opt.step()
        opt.zero_grad()
    want = _reference(world, steps)
    for u, w in zip(units, want):
        for a, b in zip(u.full_state(), w):
            ok = ok and torch.allclose(a, b, atol=1e-6, rtol=1e-5)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("prefetch", [True, False])
def test_zero3_world2_gloo(prefetch):
    """Same parameters after 3 steps as unsharded AdamW on the rank-averaged gradients, with the one-unit-ahead
    all-gather prefetch (asynchronous collectives; the order of the collectives is asserted) and without it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, prefetch)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_zero3_single_process():
    from cambrian_amd.train.zero3 import zero3_parameters, zero3_wrap
    blocks = _blocks()
    units = zero3_wrap(blocks)
    opt = torch.optim.AdamW(zero3_parameters(units), lr=1e-2, weight_decay=0.1)
    for step in range(2):
        _run(blocks, _data(1, step)[0]).backward()
        opt.step()
        opt.zero_grad()
    ref = _blocks()
    o2 = torch.optim.AdamW([p for b in ref for p in b.parameters()], lr=1e-2, weight_decay=0.1)
    for step in range(2):
        o2.zero_grad()
        _run(ref, _data(1, step)[0]).backward()
        o2.step()
    for u, b in zip(units, ref):
        for a, w in zip(u.full_state(), b.parameters()):
            assert torch.allclose(a, w, atol=1e-7, rtol=1e-6)


def test_zero3_finalize_flushes_units_with_unused_parameters():
    """A trainable unit one of whose parameters gets no gradient never counts down in the hooks: ``zero3_finalize``
    reduces what arrived (the missing gradient as zeros) and drops the full storage (zero3.py)."""
    from cambrian_amd.train.zero3 import zero3_finalize, zero3_parameters, zero3_wrap

    class TwoBranch(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.used = torch.nn.Linear(8, 8)
            self.unused = torch.nn.Linear(8, 8)

        def forward(self, x):
            return self.used(x)

    torch.manual_seed(0)
    m = TwoBranch()
    ref = [p.detach().clone() for p in m.parameters()]
    (u,) = zero3_wrap([m])
    x = torch.randn(4, 8)
    m(x).pow(2).mean().backward()
    assert u.shard.grad is None and u.resident            # stuck: the hooks wait for `unused`
    zero3_finalize([u])
    assert not u.resident and u.shard.grad is not None
    g = u.shard.grad
    n_used = ref[0].numel() + ref[1].numel()
    assert g[:n_used].abs().sum() > 0 and torch.count_nonzero(g[n_used:]) == 0
    assert zero3_parameters([u])[0] is u.shard


def test_zero3_units_opt_out_of_frozen_weight_caches():
    """cambrian_llama.py caches fused / transposed copies of frozen weights on the owning module; a ZeRO-3 unit must
    not (they would keep the whole layer resident on every rank after release())."""
    from cambrian_amd.model.language_model import cambrian_llama as CL
    from cambrian_amd.train.zero3 import zero3_wrap
    cfg = CL.CambrianConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                            num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-5, max_position_embeddings=64,
                            rope_theta=10000.0)
    layer = CL.LlamaDecoderLayer(cfg, None, torch.float32)
    for p in layer.parameters():
        p.requires_grad_(False)
    att = layer.self_attn
    assert CL._fused_frozen_weight(att, "_w_qkv", (att.q_proj, att.k_proj, att.v_proj)) is not None
    assert "_w_qkv" in att.__dict__
    zero3_wrap([layer])
    assert "_w_qkv" not in att.__dict__                                      # dropped at wrap time
    assert CL._fused_frozen_weight(att, "_w_qkv", (att.q_proj, att.k_proj, att.v_proj)) is None
    assert all(m.__dict__.get("_cmb_no_weight_cache") for m in layer.modules())


def test_zero3_backward_starts_from_any_output():
    """the gather hook sits on EVERY output tensor that requires grad and runs once per backward — a unit whose
    first output is not differentiable (or not used by the loss) still gathers, reduces and releases."""
    from cambrian_amd.train.zero3 import zero3_finalize, zero3_wrap

    class TwoOut(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(8, 8)

        def forward(self, x):
            return x.detach().sum(), self.a(x), self.a(x) * 2.0     # (no grad, grad, grad)

    torch.manual_seed(0)
    m = TwoOut()
    ref = torch.nn.Linear(8, 8)
    ref.load_state_dict(m.a.state_dict())
    (u,) = zero3_wrap([m])
    x = torch.randn(4, 8)
    for _ in range(2):                                              # two steps: the per-backward flag is re-armed
        u.shard.grad = None
        _, y1, y2 = m(x)
        (y1.pow(2).mean() + y2.mean()).backward()
        zero3_finalize([u])
        assert not u.resident and u.shard.grad is not None
    (ref(x).pow(2).mean() + (ref(x) * 2.0).mean()).backward()
    want = torch.cat([ref.weight.grad.reshape(-1), ref.bias.grad.reshape(-1)])
    assert torch.allclose(u.shard.grad[: want.numel()], want, atol=1e-6)


def test_zero3_under_activation_recompute():
    """Every unit inside a non-reentrant checkpoint region (the finetune stage's decoder layers): the recompute's
    forward runs inside the unit's backward and must neither release the parameters nor restart the gradient count."""
    from torch.utils.checkpoint import checkpoint
    from cambrian_amd.train.zero3 import zero3_finalize, zero3_parameters, zero3_wrap

    def run(blocks, x):
        for b in blocks:
            x = checkpoint(b, x, use_reentrant=False)
        return x.pow(2).mean()

    blocks = _blocks()
    for p in blocks[1].parameters():
        p.requires_grad_(False)
    units = zero3_wrap(blocks)
    log = []
    for u in units:
        u.log = log
    opt = torch.optim.AdamW(zero3_parameters(units), lr=1e-2, weight_decay=0.1)
    for step in range(2):
        x = _data(1, step)[0].requires_grad_(True)
        run(blocks, x).backward()
        zero3_finalize(units)
        assert all(not u.resident for u in units)
        opt.step()
        opt.zero_grad()
    # one gather per unit per pass, recompute included: forward 3 + backward 3 per step (prefetches count as gathers)
    assert sum(1 for w, _ in log if w in ("gather", "prefetch")) == 12
    want = _reference(1, 2)
    for u, w in zip(units, want):
        for a, b in zip(u.full_state(), w):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_zero3_refuses_recompute_without_the_probe(monkeypatch):
    """ZeRO-3 under activation re-computation needs torch's graph-task probe to tell a recompute forward from a new one; a
    torch build without it must be refused at wrap time, not run into released storage."""
    import torch
    from cambrian_amd.train.zero3 import Zero3Unit, zero3_wrap
    m = torch.nn.Linear(4, 4)
    monkeypatch.setattr(Zero3Unit, "recompute_probe_available", staticmethod(lambda: False))
    with pytest.raises(RuntimeError, match="re-computation"):
        zero3_wrap([m], gradient_checkpointing=True)
