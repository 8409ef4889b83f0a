import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# no network for the released weights: towers run on seeded random weights of the same architecture (explicit opt-in,
# base_encoder.py::_random_init_or_raise)
os.environ.setdefault("CAMBRIAN_AMD_RANDOM_INIT", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def rel_err(a, b):
    """max |a-b| / max |b|  — the "rel" of the parity statements in DESIGN.md."""
    import torch
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    denom = b.abs().max().clamp_min(1e-12)
    return ((a - b).abs().max() / denom).item()


# stated tolerances: fp32 path = exact-fp32 MFMA / VALU kernels vs fp32 CPU oracle;
# bf16 path = bf16 storage, fp32 accumulation, vs the same fp32 oracle.
TOL = {"fp32": 2e-5, "bf16": 2e-2}
