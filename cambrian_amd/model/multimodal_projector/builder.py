"""Projectors — drop-in for cambrian/model/multimodal_projector/builder.py:54-78 (``linear``, ``mlpNx_gelu``,
``identity``; se_mlp / CAbstractor are unused ablation leftovers) plus the Sequential
container used for ``mm_projector`` / ``mm_projector_aux_i`` in cambrian_arch.py:49,56.  The module tree and
state-dict keys ("0.weight", "2.bias", "3.weight", ...) equal the reference's nn.Sequential; the forward runs on
the HIP GEMM with bias / exact-erf GELU fused in the epilogue and the LayerNorm kernel."""
from __future__ import annotations

import re

import torch
import torch.nn as nn

from ... import lib as L
from ... import ops


class HipSequential(nn.Sequential):
    """nn.Sequential of Linear / GELU / LayerNorm whose forward is executed by the C-ABI kernels.  ``fp8_heavy`` (set on
    the aux projectors, whose rows are all tower tokens) lets ``ops.fp8_projections`` move its GEMMs to the fp8 MFMA."""
    fp8_heavy = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        y = x.reshape(-1, shape[-1])
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU)
                if fuse and getattr(mods[i + 1], "approximate", "none") != "none":
                    raise L.CambrianAmdError("only the exact-erf nn.GELU() of the reference is fused")
                y = ops.linear(y, m.weight, m.bias, act=L.ACT_GELU_ERF if fuse else L.ACT_NONE, heavy=self.fp8_heavy)
                i += 2 if fuse else 1
            elif isinstance(m, nn.LayerNorm):
                y = ops.layernorm(y, m.weight, m.bias, m.eps)
                i += 1
            elif isinstance(m, nn.Identity):
                i += 1
            else:
                raise L.CambrianAmdError(f"HipSequential cannot run {type(m).__name__}")
        return y.view(*shape[:-1], y.shape[-1])


class HipLinear(nn.Linear):
    """nn.Linear (same keys: weight, bias) executed by the HIP GEMM."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        y = ops.linear(x.reshape(-1, shape[-1]), self.weight, self.bias)
        return y.view(*shape[:-1], y.shape[-1])


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def build_vision_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return HipLinear(config.mm_hidden_size, config.hidden_size)
    mlp_gelu_match = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if mlp_gelu_match:
        mlp_depth = int(mlp_gelu_match.group(1))
        modules = [nn.Linear(config.mm_hidden_size, config.hidden_size)]
        for _ in range(1, mlp_depth):
            modules.append(nn.GELU())
            modules.append(nn.Linear(config.hidden_size, config.hidden_size))
        return HipSequential(*modules)
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")
