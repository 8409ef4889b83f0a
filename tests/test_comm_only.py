"""CPU, gloo, world_size 2: bench.py --comm-only — the gradient exchange of a parameter set alone, GradSync
all-reduce and the ZeRO-2 reduce-scatter + all-gather, K steps, one JSON line from rank 0 with the bus bandwidth per bucket;
the bucket size comes from the command line (--bucket-mb)."""
import argparse
import io
import json
import os
import socket
import sys
from contextlib import redirect_stdout

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, zero2, bucket_mb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    from cambrian_amd.train.dp import GradSync, init_distributed
    from cambrian_amd.train.zero import Zero2AdamW
    init_distributed("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 3000, 500, 7000, 64)]
    args = argparse.Namespace(steps=3, warmup=1, stage="pretrain", bucket_mb=bucket_mb, zero2=zero2)
    if zero2:
        opt, sync = Zero2AdamW(params, lr=1e-3, bucket_mb=bucket_mb), None
    else:
        opt, sync = None, GradSync(params, bucket_mb=bucket_mb)
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.comm_only(args, rank, world, torch.device("cpu"), params, opt, sync)
    q.put((rank, rc, buf.getvalue()))


@pytest.mark.parametrize("zero2", [False, True])
def test_comm_only_world2_gloo(zero2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    bucket_mb = 0.012       # 12 KiB buckets -> several collectives for 46 KB of gradients
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, zero2, bucket_mb)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [0, 0]
    assert res[1][2].strip() == ""                       # only rank 0 prints
    line = json.loads(res[0][2].strip())
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["higher_is_better"] is False and line["unit"] == "ms/step"
    assert line["config"]["bucket_mb"] == bucket_mb and "NOT_HEADLINE" in line["config"]
    kinds = {b["collective"] for b in line["buckets"]}
    assert kinds == ({"reduce_scatter", "all_gather"} if zero2 else {"all_reduce"})
    n_buckets = len({b["bucket"] for b in line["buckets"]})
    assert n_buckets >= 3                                 # the CLI bucket size took effect
    payload = sum(b["bytes"] for b in line["buckets"] if b["collective"] != "all_gather")
    assert payload >= line["config"]["gradient_bytes"]    # every gradient byte is in some bucket (ZeRO-2 pads to the world size)
    assert all(b["bus_gb_per_s"] > 0 and b["avg_ms"] > 0 for b in line["buckets"])
    assert line["aggregate_bus_gb_per_s"] > 0


def _fields_worker(rank, world, port, q, zero2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    from cambrian_amd.train.dp import GradSync, init_distributed
    from cambrian_amd.train.zero import Zero2AdamW
    init_distributed("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 3000, 500, 7000, 64)]
    args = argparse.Namespace(steps=4, no_comm_pass=False)
    if zero2:
        opt, sync = Zero2AdamW(params, lr=1e-3, bucket_mb=0.012), None
    else:
        opt, sync = None, GradSync(params, bucket_mb=0.012)
    # rank 1 pretends to be the slower rank: 0.5 s vs 0.4 s for the 4 steps
    out = bench.multi_gpu_fields(args, rank, world, torch.device("cpu"), 0.4 + 0.1 * rank, [], sync, opt, comm_steps=2)
    q.put((rank, out))


@pytest.mark.parametrize("zero2", [False, True])
def test_multi_gpu_fields_world2_gloo(zero2):
    """The scalar N > 1 fields bench.py adds to its line: rank spread, and the communication-only pass
    over the real bucket table after the timed region."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fields_worker, args=(r, 2, port, q, zero2)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):   # every rank computes the same scalars
        o = res[r]
        assert abs(o["rank_step_ms_min"] - 100.0) < 1e-6 and abs(o["rank_step_ms_max"] - 125.0) < 1e-6
        assert o["sync_wait_ms_per_step"] is None            # no HIP events on the CPU
        assert o["comm_only_steps"] == 2 and o["comm_only_ms"] > 0 and o["bus_gb_per_s"] > 0
        assert o["comm_only_bytes"] == 4 * (1000 + 3000 + 500 + 7000 + 64)
        assert o["comm_only_collective"] == ("reduce_scatter+all_gather" if zero2 else "all_reduce")
        assert all(not isinstance(v, (dict, list)) for v in o.values())   # scalars only: the driver keeps nothing else
    assert res[0]["comm_only_ms"] == res[1]["comm_only_ms"]


def test_compact_line_is_small_and_flat():
    """The default JSON line: < 8 KB with the region / all-own-GEMM / A-B figures as scalar keys of ``roofline`` (the driver
    keeps only scalars of parsed.roofline), built from a full line of an earlier round."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r04f_bench_driver_cmd_run1.json")) as f:
        full = json.load(f)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8192
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "region_frac", "region_ms_per_step", "region_executed_frac",
              "all_own_gemm_frac", "ab_p5_gelu_tflops"):
        assert k in rf and not isinstance(rf[k], (dict, list)), k
    assert abs(rf["region_frac"] - full["roofline"]["region"]["frac"]) < 1e-12
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == 8
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert line[k] == full[k]
