"""TEST INFRASTRUCTURE ONLY — CPU restatement of the fp8 projection GEMM (BASELINE configs[4]: "fp8 MFMA projection
GEMMs").  The reference has no fp8 path of its own (its release runs are bf16 on TPU; `SURVEY.md` §8d lists fp8
projections as the MI355X-side plan for the 34B config), so there is nothing in /root/reference to pin against:
parity for this mode is defined as (a) the quantiser being bit-exact against this restatement, which itself leans on
PyTorch's OCP ``float8_e4m3fn`` cast (round to nearest even), (b) the GEMM on the quantised operands equal to an
fp32 matmul of the de-quantised operands up to accumulation order, and (c) the end result within the fp8 tolerance
of the un-quantised fp32 reference arithmetic (tests/test_fp8_gpu.py states the numbers).

Scheme: every row of X [M,K] and W [N,K] is scaled so its largest magnitude maps to 448 (e4m3fn max):
    s = 448 / amax (1 if amax == 0);  q = e4m3fn(clamp(x * s, -448, 448));  inv = amax / 448 (1 if amax == 0)
    Y[m,n] = inv_x[m] * inv_w[n] * sum_k q_x[m,k] * q_w[n,k]   (+ the usual epilogue)
"""
from __future__ import annotations

import torch

E4M3_MAX = 448.0


def quantize_rows(x: torch.Tensor):
    """-> (uint8 e4m3fn bytes [rows, K], fp32 inv_scale [rows]); fp32 arithmetic, as the kernel."""
    xf = x.to(torch.float32)
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, torch.tensor(E4M3_MAX, dtype=torch.float32) / amax, torch.ones_like(amax))
    inv = torch.where(amax > 0, amax / torch.tensor(E4M3_MAX, dtype=torch.float32), torch.ones_like(amax))
    q = (xf * scale[:, None]).clamp_(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), inv


def dequantize(q_bytes: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    return q_bytes.view(torch.float8_e4m3fn).to(torch.float32) * inv[:, None]


def linear(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """fp32 result of the fp8-quantised linear."""
    xq, xi = quantize_rows(x)
    wq, wi = quantize_rows(w)
    y = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * xi[:, None] * wi[None, :]
    return y if bias is None else y + bias.float()
