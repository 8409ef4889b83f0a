"""CPU, world_size 2, gloo: ZeRO-2 AdamW (cambrian_amd/train/zero.py) — reduce-scattered gradient shards, sharded Adam
state, all-gathered parameters — leaves every rank with exactly the parameters an unsharded AdamW produces from the
rank-averaged gradients, over several steps, including a parameter that never receives a gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 33), torch.nn.GELU(), torch.nn.Linear(33, 32), torch.nn.LayerNorm(32),
                               torch.nn.Linear(32, 5))


def _data(world, step):
    return [torch.randn(8, 16, generator=torch.Generator().manual_seed(100 * step + k)) for k in range(world)]


def _reference(world, steps, lr, wd, accum=1):
    m = _model()
    unused = torch.nn.Parameter(torch.ones(7))
    params = list(m.parameters()) + [unused]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    for step in range(steps):
        grads = None
        for k in range(world):
            for p in params:
                p.grad = None
            for mb in range(accum):           # micro-batches accumulate (sum), as loss.backward() twice does
                m(_data(world, step * accum + mb)[k]).pow(2).mean().backward()
            g = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in params]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(params, grads):
            p.grad = g / world
        opt.step()
    return [p.detach().clone() for p in params]


def _worker(rank, world, port, q, accum=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import init_distributed
    from cambrian_amd.train.zero import Zero2AdamW
    init_distributed("gloo")
    m = _model()
    unused = torch.nn.Parameter(torch.ones(7))
    params = list(m.parameters()) + [unused]
    opt = Zero2AdamW(params, lr=1e-2, weight_decay=0.1, bucket_mb=0.002)   # tiny buckets -> several collectives
    assert len(opt.buckets) > 2
    full = sum(p.numel() for p in params) * 8
    assert opt.state_bytes() < 0.6 * full                                    # Adam moments are sharded
    steps = 3
    for step in range(steps):
        for mb in range(accum - 1):           # gradient accumulation: no collective until the last micro-batch
            with opt.no_sync():
                m(_data(world, step * accum + mb)[rank]).pow(2).mean().backward()
        m(_data(world, step * accum + accum - 1)[rank]).pow(2).mean().backward()
        opt.step()
        opt.zero_grad()
    if accum > 1:                             # a second un-synchronised backward in one step must fail loudly
        m(_data(world, 0)[rank]).pow(2).mean().backward()
        try:
            m(_data(world, 1)[rank]).pow(2).mean().backward()
            q.put((rank, False))
            return
        except RuntimeError:
            pass
    want = _reference(world, steps, 1e-2, 0.1, accum)
    ok = all(torch.allclose(p.detach(), w, atol=1e-6, rtol=1e-5) for p, w in zip(params, want))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("accum", [1, 2])
def test_zero2_world2_gloo(accum):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, accum)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_zero2_single_process_equals_adamw():
    from cambrian_amd.train.zero import Zero2AdamW
    m = _model()
    params = list(m.parameters())
    opt = Zero2AdamW(params, lr=1e-2, weight_decay=0.1)
    for step in range(2):
        m(_data(1, step)[0]).pow(2).mean().backward()
        opt.step()
        opt.zero_grad()
    m2 = _model()
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-2, weight_decay=0.1)
    for step in range(2):
        o2.zero_grad()
        m2(_data(1, step)[0]).pow(2).mean().backward()
        o2.step()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=1e-7, rtol=1e-6)


def test_zero2_skipped_step_leaves_no_stale_gradients():
    """backward -> zero_grad() WITHOUT step() (a skipped non-finite step) must reset the buckets: the next
    backward neither raises nor adds to the skipped gradients."""
    from cambrian_amd.train.zero import Zero2AdamW
    m = _model()
    opt = Zero2AdamW(list(m.parameters()), lr=1e-2, weight_decay=0.1)
    (m(_data(1, 7)[0]).pow(2).mean() * 1e3).backward()      # the step that gets skipped
    opt.zero_grad()
    with opt.no_sync():                                     # ... and the same under accumulation
        (m(_data(1, 8)[0]).pow(2).mean() * 1e3).backward()
    opt.zero_grad()
    m(_data(1, 0)[0]).pow(2).mean().backward()
    opt.step()
    opt.zero_grad()
    m2 = _model()
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-2, weight_decay=0.1)
    m2(_data(1, 0)[0]).pow(2).mean().backward()
    o2.step()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=1e-7, rtol=1e-6)


def _worker_skipped(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import init_distributed
    from cambrian_amd.train.zero import Zero2AdamW
    init_distributed("gloo")
    m = _model()
    params = list(m.parameters())
    opt = Zero2AdamW(params, lr=1e-2, weight_decay=0.1, bucket_mb=0.002)
    with opt.no_sync():                                     # a micro-batch whose step is then skipped: `pending` never moved
        (m(_data(world, 7)[rank]).pow(2).mean() * 1e3).backward()
    opt.zero_grad()
    (m(_data(world, 8)[rank]).pow(2).mean() * 1e3).backward()   # a launched reduce-scatter that is then dropped
    opt.zero_grad()
    m(_data(world, 0)[rank]).pow(2).mean().backward()
    opt.step()
    opt.zero_grad()
    m2 = _model()                                            # unsharded AdamW on the rank-averaged gradients of step 0 only
    p2 = list(m2.parameters())
    grads = None
    for k in range(world):
        for p in p2:
            p.grad = None
        m2(_data(world, 0)[k]).pow(2).mean().backward()
        g = [p.grad.clone() for p in p2]
        grads = g if grads is None else [a + b for a, b in zip(grads, g)]
    for p, g in zip(p2, grads):
        p.grad = g / world
    torch.optim.AdamW(p2, lr=1e-2, weight_decay=0.1).step()
    ok = all(torch.allclose(a.detach(), b.detach(), atol=1e-6, rtol=1e-5) for a, b in zip(params, p2))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_world2_skipped_no_sync_step_leaves_no_stale_gradients():
    """at world > 1 a zero_grad() after a no_sync() micro-batch (whose backward adds into flat_grad without
    moving `pending`) must still clear the bucket: the next step's reduce-scatter carries only its own gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_skipped, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _decoder_and_params(seed=0):
    """The product's Llama backbone + lm_head as a PARAMETER CONTAINER on the CPU (its forward needs the GPU: no fallback) —
    the parameter set of the finetune stage — and a functional forward on those very tensors from the oracle."""
    from cambrian_amd.model.language_model import cambrian_llama as CL
    cfg = CL.CambrianConfig(vocab_size=97, hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                            num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=64)
    torch.manual_seed(seed)
    m = torch.nn.Module()
    m.model = CL.LlamaBackbone(cfg, device="cpu", llm_dtype=torch.float32)
    m.lm_head = torch.nn.Linear(64, 97, bias=False)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.05)
    return m, cfg


def _decoder_loss(m, cfg, ids):
    from oracle import llama as OL
    p = dict(m.named_parameters())
    emb = p["model.embed_tokens.weight"][ids]
    pos = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1)
    hidden = OL.decoder_forward(p, cfg, emb, pos)
    return OL.lm_loss(hidden, p["lm_head.weight"], ids)[0]


def _ids(world, step):
    return [torch.randint(0, 97, (2, 12), generator=torch.Generator().manual_seed(1000 * step + k)) for k in range(world)]


def _worker_decoder(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import init_distributed
    from cambrian_amd.train.zero import Zero2AdamW
    init_distributed("gloo")
    m, cfg = _decoder_and_params()
    params = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    opt = Zero2AdamW(params, lr=1e-2, weight_decay=0.05, bucket_mb=0.02)      # 20 KiB buckets: the decoder spans several
    owned = {n for b in opt.buckets for p_, n in zip(params, names) if any(p_ is x for x in b.params)}
    assert len(opt.buckets) >= 4 and any("q_proj" in n for n in owned) and any("embed_tokens" in n for n in owned)
    full = sum(p.numel() for p in params) * 8
    assert opt.state_bytes() < 0.6 * full                                      # the decoder's Adam moments are sharded
    for step in range(2):
        _decoder_loss(m, cfg, _ids(world, step)[rank]).backward()
        opt.step()
        opt.zero_grad()
    m2, cfg2 = _decoder_and_params()
    p2 = list(m2.parameters())
    o2 = torch.optim.AdamW(p2, lr=1e-2, weight_decay=0.05)
    for step in range(2):
        grads = None
        for k in range(world):
            for p in p2:
                p.grad = None
            _decoder_loss(m2, cfg2, _ids(world, step)[k]).backward()
            g = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in p2]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(p2, grads):
            p.grad = g / world
        o2.step()
    ok = all(torch.allclose(a.detach(), b.detach(), atol=2e-6, rtol=1e-4) for a, b in zip(params, p2))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_world2_shards_the_decoder_parameter_set():
    """the FINETUNE stage's parameter set — the product's decoder modules (embeddings, q / k / v / o, gate /
    up / down, norms) + lm_head — under ZeRO-2 at world 2 on gloo: reduce-scattered gradient shards, Adam moments for the
    owned half only, all-gathered parameters == unsharded AdamW on the rank-averaged gradients (forward by oracle/llama.py on
    the same tensors: the product's decoder kernels need the GPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_decoder, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
