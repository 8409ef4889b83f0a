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


def fit_err(a, b):
    """(|slope - 1|, relative L2 error) of ``a`` against the reference ``b``: slope = <a, b> / <b, b>.  A max-abs bound of
    a few percent would let a SYSTEMATIC scale error of that size through; the least-squares slope sees a 0.5 % scale
    error even under bf16 rounding noise (which is zero-mean), and the L2 error averages the noise down."""
    import torch
    a = a.detach().double().flatten().cpu()
    b = b.detach().double().flatten().cpu()
    bb = float((b * b).sum())
    if bb == 0.0:
        return 0.0, float(a.norm())
    slope = float((a * b).sum()) / bb
    return abs(slope - 1.0), float((a - b).norm() / b.norm())
