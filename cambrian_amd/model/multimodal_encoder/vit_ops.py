"""Raw launchers of the vision-tower kernels (forward only: the towers are frozen)."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from ... import lib as L

# Round 6 (measured, OFF by default): the frozen towers' LayerNorm -> linear pairs (ViT ln1 -> qkv, ln2 -> fc1; ConvNeXt ln -> fc1)
# as cmb_row_stats + ONE GEMM on the un-normalised rows with the LayerNorm folded into its epilogue (cmb_gemm_desc.row_mean): the
# normalised copy is never written.  Same-box A/B at 24 images: region 301.1 / 301.6 ms without, 301.0 / 301.0 with — the statistics
# pass saves 40 % of a LayerNorm (ConvNeXt stage 3: 98.6 -> 57.2 us) and the two extra packed multiply-adds + the colsum vector cost
# the 4-wave GEMM's one-wave-per-SIMD epilogue 3-9 % (1634 -> 1688 us): profiles/r06_lab.md.  CAMBRIAN_AMD_LN_FUSE=1 packs the
# bf16 towers that way (tests/test_ln_fold_gpu.py keeps the path correct).  Read when a tower packs its weights.
LN_FUSE = os.environ.get("CAMBRIAN_AMD_LN_FUSE", "0") == "1"


def fold_ln_into_linear(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                        dt: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """LN(x) W^T + b = rstd (x W'^T - mean colsum) + b':  W' = W diag(gamma) in ``dt`` (what the GEMM reads), colsum = the row
    sums of W' AS STORED (so that the mean term cancels against what the matrix pipe accumulates), b' = b + W beta (fp32)."""
    w32, g32, be32 = w.float(), gamma.float(), beta.float()
    w2 = (w32 * g32[None, :]).to(dt)
    col_sum = w2.float().sum(1)
    b2 = w32 @ be32
    if b is not None:
        b2 = b2 + b.float()
    return w2, col_sum, b2


def k_vit_attn(qkv: torch.Tensor, B: int, N: int, heads: int, hd: int, scale: float, force_simple: bool = False):
    """qkv [B*N, 3*heads*hd] -> [B*N, heads*hd] (non-causal MHA)."""
    L.require_gpu(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * N, 3 * heads * hd)
    out = torch.empty((B * N, heads * hd), dtype=qkv.dtype, device=qkv.device)
    rc = L.load().cmb_vit_attn_fwd(L.dtype_code(qkv.dtype), qkv.data_ptr(), B, N, heads, hd, scale, out.data_ptr(),
                                   1 if force_simple else 0, L.stream_ptr(qkv.device))
    L.check(rc, "cmb_vit_attn_fwd")
    return out


def k_patchify(img: torch.Tensor, p: int, kpad: int, out_dtype: torch.dtype) -> torch.Tensor:
    """NCHW image -> [B*(H/p)*(W/p), kpad] patch rows, column order (c, dy, dx), zero padded."""
    L.require_gpu(img)
    img = img if img.is_contiguous() else img.contiguous()
    B, C, H, W = img.shape
    rows = B * (H // p) * (W // p)
    cols = torch.empty((rows, kpad), dtype=out_dtype, device=img.device)
    rc = L.load().cmb_patchify_nchw(L.dtype_code(img.dtype), img.data_ptr(), B, C, H, W, p, L.dtype_code(out_dtype),
                                    cols.data_ptr(), kpad, L.stream_ptr(img.device))
    L.check(rc, "cmb_patchify_nchw")
    return cols


def k_patchify2x2(x: torch.Tensor) -> torch.Tensor:
    """NHWC [B,H,W,C] -> [B*(H/2)*(W/2), 4C], column order (dy, dx, c)."""
    L.require_gpu(x)
    assert x.is_contiguous()
    B, H, W, C = x.shape
    cols = torch.empty((B * (H // 2) * (W // 2), 4 * C), dtype=x.dtype, device=x.device)
    rc = L.load().cmb_patchify2x2_nhwc(L.dtype_code(x.dtype), x.data_ptr(), B, H, W, C, cols.data_ptr(),
                                       L.stream_ptr(x.device))
    L.check(rc, "cmb_patchify2x2_nhwc")
    return cols


def k_dwconv7x7(x: torch.Tensor, w49: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x NHWC, w49 fp32 [49, C] (tap-major), bias fp32 [C]."""
    L.require_gpu(x, w49, bias)
    assert x.is_contiguous() and w49.dtype == torch.float32 and bias.dtype == torch.float32
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    rc = L.load().cmb_dwconv7x7_nhwc(L.dtype_code(x.dtype), x.data_ptr(), B, H, W, C, w49.data_ptr(), bias.data_ptr(),
                                     y.data_ptr(), L.stream_ptr(x.device))
    L.check(rc, "cmb_dwconv7x7_nhwc")
    return y


def k_resample(x: torch.Tensor, hi: int, wi: int, out: torch.Tensor, ho: int, wo: int, col_offset: int = 0) -> None:
    """x [B, hi*wi, C] -> out[:, :, col_offset:col_offset+C] with out [B, ho*wo, Ctot] (bilinear, fp32 lerp)."""
    L.require_gpu(x, out)
    B, T, C = x.shape
    assert T == hi * wi and out.shape[0] == B and out.shape[1] == ho * wo and x.stride(2) == 1 and out.stride(2) == 1
    optr = out.data_ptr() + col_offset * out.element_size()
    rc = L.load().cmb_resample_bilinear(L.dtype_code(x.dtype), x.data_ptr(), B, hi, wi, C, x.stride(1), x.stride(0), optr,
                                        ho, wo, out.stride(1), out.stride(0), L.stream_ptr(x.device))
    L.check(rc, "cmb_resample_bilinear")


def k_act_mul(a: torch.Tensor, b: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """y = act(a) * b over 2-D (possibly column-sliced) operands."""
    L.require_gpu(a, b)
    rows, D = a.shape
    y = torch.empty((rows, D), dtype=a.dtype, device=a.device)
    rc = L.load().cmb_act_mul(L.dtype_code(a.dtype), act, a.data_ptr(), a.stride(0), L.ptr(b),
                              0 if b is None else b.stride(0), rows, D, y.data_ptr(), D, L.stream_ptr(a.device))
    L.check(rc, "cmb_act_mul")
    return y


def k_bcast_rows(dst: torch.Tensor, ld: int, nrows: int, src: torch.Tensor) -> None:
    """dst[r*ld : r*ld + D] = src for r in range(nrows) (CLS rows of the token buffers)."""
    L.require_gpu(dst, src)
    rc = L.load().cmb_bcast_rows(L.dtype_code(dst.dtype), dst.data_ptr(), ld, nrows, src.numel(), src.data_ptr(),
                                 L.stream_ptr(dst.device))
    L.check(rc, "cmb_bcast_rows")
