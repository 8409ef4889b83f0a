"""`-m gpu`: the gradient all-reduce on the REAL backend.  One-GPU boxes cannot show scaling, but they can show that
RCCL ("nccl" on ROCm) initialises over the 127.0.0.1 rendezvous and that GradSync's bucketed asynchronous all-reduce
runs on it (world size 1: SUM over one rank, then the mean) — the CPU/gloo world-2 tests (tests/test_dp.py) cover the
arithmetic across ranks, this covers the device path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gradsync_runs_on_rccl_world1(dev):
    from cambrian_amd.train.dp import GradSync
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        t = torch.arange(1 << 20, device=dev, dtype=torch.float32)
        dist.all_reduce(t)                                   # a bare RCCL collective
        torch.cuda.synchronize()
        assert torch.equal(t, torch.arange(1 << 20, device=dev, dtype=torch.float32))
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.GELU(), torch.nn.Linear(512, 64)).to(dev)
        ref = [None]
        x = torch.randn(32, 256, device=dev)
        m(x).pow(2).mean().backward()
        ref = [p.grad.clone() for p in m.parameters()]
        for p in m.parameters():
            p.grad = None
        sync = GradSync(m.parameters(), bucket_mb=0.25, always_reduce=True)    # several buckets -> several collectives
        assert len(sync.buckets) > 1 and sync.reduce
        m(x).pow(2).mean().backward()
        assert any(b.work is not None for b in sync.buckets)                   # launched from the autograd hooks
        sync.finish()
        for p, g in zip(m.parameters(), ref):
            assert torch.allclose(p.grad, g, atol=1e-7, rtol=1e-6)
        sync.remove()
    finally:
        dist.destroy_process_group()
