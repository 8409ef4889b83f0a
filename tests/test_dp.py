"""CPU, world_size 2, gloo: the bucketed backward-overlapped gradient all-reduce (cambrian_amd/train/dp.py)
yields exactly the mean of the per-rank gradients, including parameters that receive no gradient on a step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.LayerNorm(32),
                               torch.nn.Linear(32, 4))


def _worker(rank, world, port, q, comm_dtype=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import GradSync, init_distributed
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    m = _model()
    unused = torch.nn.Parameter(torch.ones(5))         # never used in the loss: must still sync (as zero)
    params = list(m.parameters()) + [unused]
    sync = GradSync(params, bucket_mb=0.002, comm_dtype=comm_dtype)   # tiny buckets -> several collectives
    assert len(sync.buckets) > 2
    data = [torch.randn(8, 16, generator=torch.Generator().manual_seed(10 + k)) for k in range(world)]
    for step in range(2):                               # two steps: buckets are reusable
        for p in params:
            p.grad = None
        m(data[rank] * (step + 1)).pow(2).mean().backward()
        sync.finish()
    got = [p.grad.clone() for p in params]
    # expected: mean over ranks of the local gradients, computed without any communication
    want = None
    for k in range(world):
        mk = _model()
        mk(data[k] * 2).pow(2).mean().backward()
        gk = [p.grad for p in mk.parameters()] + [torch.zeros(5)]
        want = gk if want is None else [a + b for a, b in zip(want, gk)]
    want = [g / world for g in want]
    if comm_dtype is None:
        ok = all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(got, want))
    else:   # bf16 on the wire: the mean to the wire's rounding (1e-2 against the fp32 buckets), fp32 slots
        ok = all(a.dtype == torch.float32 and float((a - b).abs().max()) <= 1e-2 * float(b.abs().max()) + 1e-12 for a, b in zip(got, want))
        ok = ok and any(not torch.equal(a, b) for a, b in zip(got, want))   # ... and it really went through the narrow dtype
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype", [None, torch.bfloat16])
def test_gradsync_world2_gloo(comm_dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, comm_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_gradsync_single_process_is_identity():
    from cambrian_amd.train.dp import GradSync
    m = _model()
    sync = GradSync(m.parameters())
    x = torch.randn(4, 16)
    m(x).sum().backward()
    sync.finish()
    m2 = _model()
    m2(x).sum().backward()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a.grad, b.grad)


def test_gradsync_reset_after_an_abandoned_step():
    """A backward that is not followed by finish() (bench.py's out-of-memory probe, an exception) leaves bucket countdowns
    part-way: reset() re-arms them, and the next step is the plain one."""
    from cambrian_amd.train.dp import GradSync
    m = _model()
    sync = GradSync(list(m.parameters()), bucket_mb=0.0001)     # one bucket per parameter
    x = torch.randn(4, 16)
    (m(x).sum() * 7.0).backward()                               # abandoned: no finish()
    for p in m.parameters():
        p.grad = None
    sync.reset()
    assert all(b.pending == len(b.params) and b.work is None for b in sync.buckets)
    m(x).sum().backward()
    sync.finish()
    m2 = _model()
    m2(x).sum().backward()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a.grad, b.grad)


def _worker_sampler(rank, world, port, q):
    """The whole N > 1 recipe on CPU: LengthGroupedSampler order -> rank_batches cut -> per-rank step -> GradSync.  The
    averaged gradient must equal the single-process gradient of the mean loss over the whole megabatch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cambrian_amd.train.dp import GradSync, init_distributed
    from cambrian_amd.train.sampler import LengthGroupedSampler, rank_batches
    init_distributed("gloo")
    n, bs = 32, 4
    data = torch.randn(n, 16, generator=torch.Generator().manual_seed(3))
    lengths = torch.randint(1, 100, (n,), generator=torch.Generator().manual_seed(4)).tolist()
    order = list(LengthGroupedSampler(bs, world, lengths=lengths, generator=torch.Generator().manual_seed(5)))
    mine = rank_batches(order, rank, world, bs)
    m = _model()
    sync = GradSync(list(m.parameters()), bucket_mb=0.01)
    ok = True
    for step, idx in enumerate(mine):
        for p in m.parameters():
            p.grad = None
        m(data[idx]).pow(2).mean().backward()
        sync.finish()
        ref = _model()
        mega = order[step * world * bs:(step + 1) * world * bs]
        ref(data[mega]).pow(2).mean().backward()      # equal per-rank batch sizes: mean of means == mean over the megabatch
        ok &= all(torch.allclose(a.grad, b.grad, atol=1e-6, rtol=1e-5) for a, b in zip(m.parameters(), ref.parameters()))
    q.put((rank, ok, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_sampler_cut_plus_gradsync_equals_megabatch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sampler, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, 4), (1, True, 4)]
