"""Host-side operators over the C-ABI kernels.

Two layers:
  * ``k_*``  — thin, stateless launchers (no autograd): validate, fill the C structs, call the library.
  * ``*Fn`` / functional wrappers — ``torch.autograd.Function`` objects that pair each forward kernel
    with its backward kernels, so the reference's modules (cambrian/model/vision_sampler.py,
    cambrian_arch.py) can be re-stated on top of them without touching torch's own math kernels.

Compute dtype is the dtype of the activation tensor (bf16 for training, fp32 for the parity path);
parameters stay fp32 masters and are cast per call (the cast is itself ``cmb_cast``).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import lib as L
from .lib import RowMap, GemmDesc, SvaDesc


# ================================================================================================
# raw launchers
# ================================================================================================
def _kstep(dtype: torch.dtype) -> int:
    return 64 if dtype == torch.bfloat16 else 32


def pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def k_cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous cast fp32<->bf16 through cmb_cast (returns x itself if nothing to do)."""
    if x.dtype == dtype and x.is_contiguous():
        return x
    L.require_gpu(x)
    xc = x if x.is_contiguous() else x.contiguous()
    if xc.dtype == dtype:
        return xc
    out = torch.empty(xc.shape, dtype=dtype, device=xc.device)
    rc = L.load().cmb_cast(L.dtype_code(xc.dtype), xc.data_ptr(), L.dtype_code(dtype), out.data_ptr(), xc.numel(),
                           L.stream_ptr(xc.device))
    L.check(rc, "cmb_cast")
    return out


def k_gemm(a: torch.Tensor, w: torch.Tensor, *, M: Optional[int] = None, a_map: Optional[RowMap] = None,
           bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, colscale: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, r_map: Optional[RowMap] = None,
           out: Optional[torch.Tensor] = None, c_map: Optional[RowMap] = None, out_dtype: Optional[torch.dtype] = None,
           pre_out: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 0.0,
           split_k: int = 1, tile: int = 0, row_stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
           col_sum: Optional[torch.Tensor] = None, _collect: Optional[list] = None) -> torch.Tensor:
    """C[M,N] = epilogue(alpha * A[M,K] @ W[N,K]^T).  ``a`` is [M,K] (row stride a.stride(0)) unless an
    explicit ``a_map``/``M`` is given, in which case ``a`` is just the base tensor.  ``row_stats`` = (mean, rstd) [M] fp32
    of ``k_row_stats`` + ``col_sum`` [N]: the LayerNorm in front of this linear folded into its epilogue
    (cmb_gemm_desc.row_mean; ``w`` = W diag(gamma), ``bias`` = b + W beta, ``col_sum`` = rowsum of ``w`` as stored)."""
    L.require_gpu(a, w, bias, colscale, residual, out, pre_out)
    dt = a.dtype
    if w.dtype != dt:
        raise L.CambrianAmdError(f"gemm operand dtypes differ: {dt} vs {w.dtype}")
    N, K = w.shape
    if w.stride(1) != 1:
        raise L.CambrianAmdError("gemm weight must be K-contiguous")
    if a_map is None:
        if a.dim() != 2 or a.stride(1) != 1 or a.shape[1] != K:
            raise L.CambrianAmdError(f"gemm A must be [M,{K}] K-contiguous, got {tuple(a.shape)}")
        M = a.shape[0]
        a_map = L.identity_map(a.stride(0))
    assert M is not None
    odt = out_dtype or (out.dtype if out is not None else dt)
    n_out = N // 2 if act == L.ACT_SWIGLU_PAIRS else N   # gated epilogue: (gate, up) column pairs -> N / 2 outputs
    if out is None:
        out = torch.empty((M, n_out), dtype=odt, device=a.device)
        c_map = L.identity_map(n_out)
    elif c_map is None:
        if out.dim() != 2 or out.stride(1) != 1:
            raise L.CambrianAmdError("gemm out must be 2-D row-major unless c_map is given")
        c_map = L.identity_map(out.stride(0))
    d = GemmDesc()
    d.dtype = L.dtype_code(dt)
    d.out_dtype = L.dtype_code(out.dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.a_map = a.data_ptr(), a_map
    d.B, d.ldb = w.data_ptr(), w.stride(0)
    d.C, d.c_map = out.data_ptr(), c_map
    for t in (bias, colscale):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != N):
            raise L.CambrianAmdError("bias / colscale must be contiguous fp32 [N]")
    d.bias = L.ptr(bias)
    d.colscale = L.ptr(colscale)
    if residual is not None:
        if residual.dtype != dt:
            raise L.CambrianAmdError("residual dtype must equal the operand dtype")
        if r_map is None:
            if residual.dim() != 2 or residual.stride(1) != 1:
                raise L.CambrianAmdError("residual must be 2-D row-major unless r_map is given")
            r_map = L.identity_map(residual.stride(0))
        d.residual, d.r_map = residual.data_ptr(), r_map
    else:
        d.residual, d.r_map = None, L.identity_map(0)
    if pre_out is not None:
        if pre_out.dtype != dt or not pre_out.is_contiguous():
            raise L.CambrianAmdError("pre_out must be contiguous and of the operand dtype")
        d.pre_out, d.p_map = pre_out.data_ptr(), L.identity_map(N)
    else:
        d.pre_out, d.p_map = None, L.identity_map(0)
    d.act, d.alpha, d.beta = act, alpha, beta
    d.split_k = split_k
    d.tile_hint = tile
    if row_stats is not None:
        mean, rstd = row_stats
        L.require_gpu(mean, rstd, col_sum)
        if (col_sum is None or bias is None or any(t.dtype != torch.float32 or not t.is_contiguous() for t in (mean, rstd, col_sum))
                or mean.numel() != M or rstd.numel() != M or col_sum.numel() != N):
            raise L.CambrianAmdError("folded LayerNorm: mean / rstd fp32 [M], col_sum fp32 [N] and a bias are required")
        d.row_mean, d.row_rstd, d.col_sum = mean.data_ptr(), rstd.data_ptr(), col_sum.data_ptr()
    ws = None
    if split_k > 1:
        ws = torch.empty((split_k * M * N,), dtype=torch.float32, device=a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    else:
        d.workspace, d.workspace_bytes = None, 0
    if GEMM_CENSUS is not None and dt == torch.bfloat16 and split_k <= 1 and tile == 0:
        key = (M, N, K, act, pre_out is not None)
        GEMM_CENSUS[key] = GEMM_CENSUS.get(key, 0) + 1
    if _collect is not None:   # k_gemm_pair: the descriptor (and what it points into) instead of the launch
        _collect.append((d, out, (a, w, bias, colscale, residual, pre_out, ws, row_stats, col_sum), (M, N, K, act, dt, split_k, tile)))
        return out
    prof = GEMM_PROFILE
    tile_used = 0
    if prof is not None:  # HIP events on the launch stream around this launch (bench.py roofline leg)
        tile_used = L.load().cmb_gemm_tile(d.dtype, M, N, split_k, tile)
        if GEMM_PROFILE_TILE and tile_used != GEMM_PROFILE_TILE:
            prof = None  # only the dominant kernel is timed: every event pair costs the stream a barrier packet
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = L.load().cmb_gemm(C.byref(d), L.stream_ptr(a.device))
    L.check(rc, f"cmb_gemm(M={M},N={N},K={K},{dt})")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, dt, split_k, tile_used, (M, N, K, act, out.dtype == torch.float32),
                     L.load().cmb_gemm_last_kernel()))
    return out


def k_gemm_pair(kw0: dict, kw1: dict) -> Tuple[torch.Tensor, torch.Tensor]:
    """Two independent ``k_gemm`` calls (keyword dicts with ``a`` and ``w``) through ``cmb_gemm_pair``: ONE launch of the
    persistent 256 x 256 kernel with the workgroups split between the problems when the library's round arithmetic says it pays
    (DINOv2's and SigLIP's same-position linears), else the two launches; the results are bit-identical either way."""
    col: list = []
    k_gemm(_collect=col, **kw0)
    k_gemm(_collect=col, **kw1)
    (d0, o0, _k0, m0), (d1, o1, _k1, m1) = col
    prof = GEMM_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = L.load()
    rc = lib.cmb_gemm_pair(C.byref(d0), C.byref(d1), L.stream_ptr(o0.device))
    L.check(rc, f"cmb_gemm_pair({m0[:3]}, {m1[:3]})")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        paired = bool(lib.cmb_gemm_pair_last())
        # one row for the call: the FLOPs of both problems over the span of the launch (or of the two launches)
        prof.append((e0, e1, 2.0 * (m0[0] * m0[1] * m0[2] + m1[0] * m1[1] * m1[2]), m0[4], 1, 256,
                     (m0[0], m0[1], m0[2], m0[3], False), lib.cmb_gemm_last_kernel(),
                     ("pair:" if paired else "seq:") + f"{m1[0]}x{m1[1]}x{m1[2]}"))
    return o0, o1



def k_quantize_fp8_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [rows, K] (bf16 / fp32, K-contiguous) -> (uint8 e4m3fn bytes [rows, K], fp32 dequantisation factor [rows]):
    each row scaled so that its largest magnitude maps to 448 (cmb_quantize_fp8_rows)."""
    L.require_gpu(x)
    if x.dim() != 2 or x.stride(1) != 1:
        raise L.CambrianAmdError("fp8 quantisation needs a 2-D K-contiguous tensor")
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rc = L.load().cmb_quantize_fp8_rows(L.dtype_code(x.dtype), x.data_ptr(), x.stride(0), rows, K, q.data_ptr(), K,
                                        inv.data_ptr(), L.stream_ptr(x.device))
    L.check(rc, f"cmb_quantize_fp8_rows({rows}x{K})")
    return q, inv


def k_gemm_fp8(a_q: torch.Tensor, a_inv: torch.Tensor, w_q: torch.Tensor, w_inv: torch.Tensor, *,
               bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, colscale: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, r_map: Optional[RowMap] = None,
               pre_out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """C[M,N] = epilogue(a_inv[m] * w_inv[n] * (A_q[M,K] . W_q[N,K]^T)) on v_mfma_f32_32x32x64_f8f6f4; operands are the
    e4m3fn bytes / factors of k_quantize_fp8_rows; residual / pre_out are bf16; C is bf16 or fp32."""
    L.require_gpu(a_q, a_inv, w_q, w_inv, bias, colscale, residual, pre_out)
    M, K = a_q.shape
    N = w_q.shape[0]
    if a_q.dtype != torch.uint8 or w_q.dtype != torch.uint8 or w_q.shape[1] != K or not (a_q.is_contiguous() and w_q.is_contiguous()):
        raise L.CambrianAmdError("fp8 gemm operands must be contiguous uint8 [M,K] / [N,K]")
    out = torch.empty((M, N), dtype=out_dtype, device=a_q.device)
    d = GemmDesc()
    d.dtype, d.out_dtype = L.FP8_E4M3, L.dtype_code(out_dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.a_map = a_q.data_ptr(), L.identity_map(K)
    d.B, d.ldb = w_q.data_ptr(), K
    d.C, d.c_map = out.data_ptr(), L.identity_map(N)
    d.bias, d.colscale = L.ptr(bias), L.ptr(colscale)
    if residual is not None:
        if residual.dtype != torch.bfloat16:
            raise L.CambrianAmdError("fp8 gemm residual must be bf16")
        d.residual, d.r_map = residual.data_ptr(), (r_map or L.identity_map(residual.stride(0)))
    else:
        d.residual, d.r_map = None, L.identity_map(0)
    if pre_out is not None:
        if pre_out.dtype != torch.bfloat16 or not pre_out.is_contiguous():
            raise L.CambrianAmdError("fp8 gemm pre_out must be contiguous bf16")
        d.pre_out, d.p_map = pre_out.data_ptr(), L.identity_map(N)
    else:
        d.pre_out, d.p_map = None, L.identity_map(0)
    d.act, d.alpha, d.beta, d.split_k, d.tile_hint = act, 1.0, 0.0, 1, 0
    d.workspace, d.workspace_bytes = None, 0
    d.a_scale, d.b_scale = a_inv.data_ptr(), w_inv.data_ptr()
    rc = L.load().cmb_gemm(C.byref(d), L.stream_ptr(a_q.device))
    L.check(rc, f"cmb_gemm(fp8, M={M},N={N},K={K})")
    return out


# Forward GEMMs of ops.linear in fp8 (BASELINE configs[4]: "fp8 MFMA projection GEMMs").  The model glue switches it on
# around the SVA-side projections when config.fp8_projections is set; towers and the decoder never see it.
_FP8_LINEAR = False
_FP8_WEIGHT_CACHE: dict = {}


class fp8_projections:
    """Context manager: ``with ops.fp8_projections(True): ...`` — ops.linear(..., heavy=True) quantises x and W row-wise
    to e4m3fn and runs the forward GEMM on the fp8 MFMA; the backward stays bf16 (dX, dW from the saved bf16 operands)."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        global _FP8_LINEAR
        self.prev, _FP8_LINEAR = _FP8_LINEAR, self.on
        return self

    def __exit__(self, *exc):
        global _FP8_LINEAR
        _FP8_LINEAR = self.prev
        return False


def _fp8_weight(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Quantised copy of a projection weight.  nn.Parameters are cached by identity and re-made when the parameter
    changes (optimizer step => _version bump); anything else (views, weights folded on the fly) is quantised per call —
    a pointer-keyed cache would hand a recycled allocation somebody else's bytes."""
    w2 = weight.detach()
    if w2.stride(1) != 1 or (w2.stride(0) * w2.element_size()) % 16 != 0 or w2.data_ptr() % 16 != 0:
        w2 = w2.contiguous()
    if not isinstance(weight, torch.nn.Parameter):
        return k_quantize_fp8_rows(w2)
    key = id(weight)
    hit = _FP8_WEIGHT_CACHE.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != (weight._version, weight.data_ptr()):
        import weakref
        hit = (weakref.ref(weight, lambda _r, k=key: _FP8_WEIGHT_CACHE.pop(k, None)), (weight._version, weight.data_ptr())) \
            + k_quantize_fp8_rows(w2)
        _FP8_WEIGHT_CACHE[key] = hit
    return hit[2], hit[3]


def fp8_linear_eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (_FP8_LINEAR and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] % 128 == 0
            and weight.shape[0] % 8 == 0 and weight.dtype in (torch.bfloat16, torch.float32))


# bench.py sets this to a list to collect (start_event, end_event, flops, dtype, split_k, tile, shape, kernel id) per GEMM launch
GEMM_PROFILE = None
KERNEL_ID_TN = 7128   # the `kernel` field of cmb_gemm_tn launches in GEMM_PROFILE rows (cmb_gemm_last_kernel ids: 128, 256, 2590)
GEMM_PROFILE_TILE = 0  # 0 = time every GEMM launch; 128 / 256 = only launches of that tile configuration
# {(M, N, K, act, has_pre_out): launches} of the bf16 launches that left the kernel choice to the library, while set to a dict
GEMM_CENSUS = None


class gemm_census:
    """``with ops.gemm_census() as c: step()`` — counts the bf16 GEMM problems of one step (no events, no syncs); the
    input of ``calibrate_gemm_dispatch``."""

    def __enter__(self):
        global GEMM_CENSUS
        self.prev, GEMM_CENSUS = GEMM_CENSUS, {}
        self.shapes = GEMM_CENSUS
        return self

    def __exit__(self, *exc):
        global GEMM_CENSUS
        GEMM_CENSUS = self.prev
        return False

    def top(self, n: int = 8, min_tiles: int = 257):
        """The n problems with the most FLOPs per step among those where both 256 x 256 kernels apply: whole tile
        columns, K of at least two 64-deep tiles, no pre-activation copy, at least ``min_tiles`` tiles (257 = more than one
        round of the 256 CUs: where the library's own default is the 4-wave kernel; lower values also put single-round
        problems — the 8-wave kernel's by default — to the test)."""
        rows = []
        for (M, N, K, act, pre), cnt in self.shapes.items():
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            if pre or N % 128 or K < 128 or K % 64 or tiles < min_tiles:
                continue
            if L.load().cmb_gemm_tail_rows(M, N):
                continue   # launched as 256-tile head + 128-tile tail (gemm.hip "Tail split"): the policy does not apply
            rows.append((2.0 * M * N * K * cnt, (M, N, K, act)))
        rows.sort(reverse=True)
        return [r[1] for r in rows[:n]]


def calibrate_gemm_dispatch(shapes, iters: int = 3, device=None, candidates=(2590, 2560)):
    """Start-up calibration of the 256 x 256 kernel choice on THIS device (VERDICT r2 #1d): every (M, N, K, act) in
    ``shapes`` is run ``iters`` times (after one untimed launch) with each candidate kernel — interleaved, random bf16
    operands, HIP events on the current stream — and the fastest is registered in the library's per-shape policy
    (``cmb_gemm_policy_set``), which every later ``cmb_gemm`` of that problem without a ``tile_hint`` follows.  The
    candidates (2590 = 4-wave register-buffered kernel, 2560 = 8-wave kernel) give bit-identical results, so this
    changes speed only.  Returns one dict per shape (microseconds per candidate, the choice).  ~1-2 ms per shape."""
    lib = L.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    report = []
    for (M, N, K, act) in shapes:
        a = torch.randn((M, K), device=device, dtype=torch.bfloat16)
        w = torch.randn((N, K), device=device, dtype=torch.bfloat16)
        out = torch.empty((M, N), device=device, dtype=torch.bfloat16)
        evs = {c: [] for c in candidates}
        for it in range(iters + 1):
            for c in candidates:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                k_gemm(a, w, act=act, out=out, tile=c)
                e1.record()
                if it:
                    evs[c].append((e0, e1))
        torch.cuda.synchronize(device)
        us = {c: min(e0.elapsed_time(e1) for e0, e1 in evs[c]) * 1e3 for c in candidates}
        best = min(us, key=us.get)
        L.check(lib.cmb_gemm_policy_set(M, N, K, act, best), "cmb_gemm_policy_set")
        report.append({"M": M, "N": N, "K": K, "act": act, "us": {str(c): round(v, 1) for c, v in us.items()},
                       "choice": best, "TFLOPs": {str(c): round(2.0 * M * N * K / v / 1e6, 1) for c, v in us.items()}})
        del a, w, out
    return report


# ---- region timing (bench.py ``roofline.region``: the tower + SVA part of the step, forward and backward) -----------
# While REGION_PROFILE is a list, the model glue brackets each piece of the region with HIP events on the launch
# stream: ``region_begin`` / ``region_fwd_end`` in the forward, and identity autograd nodes (``region_mark``) whose
# backward records the events of the matching backward span — the node on the piece's OUTPUT fires when its backward
# starts ("b0"), the node on its INPUT when the gradient leaves the piece ("b1": autograd runs the nodes of the piece,
# created after the input mark, before it).  Spans without an input mark (the first piece: towers have no input
# gradient) are closed by ``region_close`` after ``backward()`` returned.
REGION_PROFILE = None


def _event():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


class _RegionMarkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, span, key):
        ctx.span, ctx.key = span, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.key not in ctx.span:
            ctx.span[ctx.key] = _event()
        return g, None, None


def region_begin(tag: str):
    """Forward start of a span (None when nobody profiles)."""
    if REGION_PROFILE is None:
        return None
    span = {"tag": tag, "f0": _event()}
    REGION_PROFILE.append(span)
    return span


def region_mark(x: torch.Tensor, span, key: str) -> torch.Tensor:
    """key "b1" on a piece's input, "b0" on its output (see above)."""
    if span is None or not x.requires_grad:
        return x
    return _RegionMarkFn.apply(x, span, key)


def region_fwd_end(span) -> None:
    if span is not None:
        span["f1"] = _event()


def region_close() -> None:
    """After backward(): spans whose backward started ("b0") but has no input mark end here."""
    if REGION_PROFILE is None:
        return
    e = None
    for span in REGION_PROFILE:
        if "b0" in span and "b1" not in span:
            e = e or _event()
            span["b1"] = e


def region_ms(spans) -> dict:
    """Forward / backward milliseconds of the recorded spans (call after a device synchronise)."""
    fwd = sum(s["f0"].elapsed_time(s["f1"]) for s in spans if "f1" in s)
    bwd = sum(s["b0"].elapsed_time(s["b1"]) for s in spans if "b0" in s and "b1" in s)
    return {"fwd_ms": fwd, "bwd_ms": bwd, "spans": len(spans)}


def k_transpose(x: torch.Tensor, r_pad: Optional[int] = None) -> torch.Tensor:
    """[R,C] -> [C,R_pad] (zero padded)."""
    L.require_gpu(x)
    R, Cc = x.shape
    if x.stride(1) != 1:
        x = x.contiguous()
    r_pad = r_pad or pad_to(R, 8)
    out = torch.empty((Cc, r_pad), dtype=x.dtype, device=x.device)
    rc = L.load().cmb_transpose(L.dtype_code(x.dtype), x.data_ptr(), R, Cc, x.stride(0), out.data_ptr(), r_pad,
                                L.stream_ptr(x.device))
    L.check(rc, "cmb_transpose")
    return out


def k_gemm_tn(at: torch.Tensor, bt: torch.Tensor, *, out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32,
              split_k: int = 1, alpha: float = 1.0, beta: float = 0.0, M: Optional[int] = None, N: Optional[int] = None,
              batch: int = 1, a_bs: int = 0, b_bs: int = 0, c_bs: int = 0, ldc: Optional[int] = None) -> torch.Tensor:
    """C[M,N] = alpha * At[K,M]^T @ Bt[K,N] (+ beta C) on cmb_gemm_tn: both operands row-major over the K contraction rows,
    bf16, any K.  ``at`` [K, >=M] / ``bt`` [K, >=N] (row strides taken from the tensors); with ``batch`` > 1 problem z
    reads columns from a_bs*z / b_bs*z on and writes C + c_bs*z (elements).  The weight gradient dW = g^T x without
    transposed copies of g and x."""
    L.require_gpu(at, bt, out)
    if at.dtype != torch.bfloat16 or bt.dtype != torch.bfloat16 or at.dim() != 2 or bt.dim() != 2:
        raise L.CambrianAmdError("gemm_tn operands must be 2-D bf16")
    if at.stride(1) != 1 or bt.stride(1) != 1 or at.shape[0] != bt.shape[0]:
        raise L.CambrianAmdError("gemm_tn operands must be row-major with equal row counts")
    K = at.shape[0]
    M = at.shape[1] if M is None else M
    N = bt.shape[1] if N is None else N
    if out is None:
        if batch > 1:
            raise L.CambrianAmdError("gemm_tn: a batched launch needs the caller's output tensor")
        out = torch.empty((M, N), dtype=out_dtype, device=at.device)
    d = GemmDesc()
    d.dtype, d.out_dtype = L.dtype_code(at.dtype), L.dtype_code(out.dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.a_map = at.data_ptr(), L.identity_map(at.stride(0))
    d.B, d.ldb = bt.data_ptr(), bt.stride(0)
    d.C, d.c_map = out.data_ptr(), L.identity_map(ldc if ldc is not None else out.stride(0))
    d.bias = d.colscale = d.residual = d.pre_out = None
    d.r_map = d.p_map = L.identity_map(0)
    d.act, d.alpha, d.beta, d.split_k, d.tile_hint = L.ACT_NONE, alpha, beta, split_k, 0
    ws = None
    if split_k > 1:
        ws = torch.empty((split_k * max(batch, 1) * M * N,), dtype=torch.float32, device=at.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    else:
        d.workspace, d.workspace_bytes = None, 0
    d.batch, d.a_batch_stride, d.b_batch_stride, d.c_batch_stride = batch, a_bs, b_bs, c_bs
    prof = GEMM_PROFILE if not GEMM_PROFILE_TILE else None   # (the timed region times the dominant kernel only)
    if prof is not None:
        e0 = _event()
    rc = L.load().cmb_gemm_tn(C.byref(d), L.stream_ptr(at.device))
    L.check(rc, f"cmb_gemm_tn(M={M}, N={N}, K={K}, batch={batch})")
    if prof is not None:   # the span covers the split-K reduce the library launches behind the kernel
        prof.append((e0, _event(), 2.0 * max(batch, 1) * M * N * K, at.dtype, split_k, 128,
                     (M, N, K, L.ACT_NONE, out.dtype == torch.float32), KERNEL_ID_TN, max(batch, 1)))
    return out


def k_colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
             group: int = 0) -> torch.Tensor:
    """out[c] (+)= sum_r x[r, c]; with ``row_scale`` [R, C / group] fp32: sum_r row_scale[r, c // group] * x[r, c]."""
    L.require_gpu(x)
    R, Cc = x.shape
    if out is None:
        out = torch.zeros((Cc,), dtype=torch.float32, device=x.device)
    if row_scale is not None:
        if row_scale.dtype != torch.float32 or row_scale.dim() != 2 or row_scale.stride(1) != 1 or row_scale.shape != (R, Cc // group):
            raise L.CambrianAmdError("colsum row_scale must be fp32 [R, C / group]")
        rc = L.load().cmb_colsum_scaled(L.dtype_code(x.dtype), x.data_ptr(), R, Cc, x.stride(0), row_scale.data_ptr(),
                                        row_scale.stride(0), group, out.data_ptr(), L.stream_ptr(x.device))
        L.check(rc, "cmb_colsum_scaled")
        return out
    rc = L.load().cmb_colsum(L.dtype_code(x.dtype), x.data_ptr(), R, Cc, x.stride(0), out.data_ptr(),
                             L.stream_ptr(x.device))
    L.check(rc, "cmb_colsum")
    return out


def k_token_mean(x3: torch.Tensor) -> torch.Tensor:
    B, T, D = x3.shape
    out = torch.empty((B, D), dtype=x3.dtype, device=x3.device)
    rc = L.load().cmb_token_mean_fwd(L.dtype_code(x3.dtype), x3.data_ptr(), B, T, D, out.data_ptr(), L.stream_ptr(x3.device))
    L.check(rc, "cmb_token_mean_fwd")
    return out


def k_segment_sum(x: torch.Tensor, rep: int) -> torch.Tensor:
    """[M, D] -> [M/rep, D], summing each run of `rep` consecutive rows (the mean kernel, scaled by rep)."""
    M, D = x.shape
    xc = x if x.is_contiguous() else x.contiguous()
    m = k_token_mean(xc.view(M // rep, rep, D))
    return (m.to(torch.float32) * float(rep)).to(x.dtype)


def k_layernorm_fwd(x, gamma, beta, eps, add=None, side=1, grid_r=1, want_stats=True):
    L.require_gpu(x, gamma, beta, add)
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=x.dtype, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_stats else None
    rc = L.load().cmb_layernorm_fwd(L.dtype_code(x.dtype), x.data_ptr(), rows, D, x.stride(0), L.ptr(add), side, grid_r,
                                    L.ptr(gamma), L.ptr(beta), eps, y.data_ptr(), D, L.ptr(mean), L.ptr(rstd),
                                    L.stream_ptr(x.device))
    L.check(rc, "cmb_layernorm_fwd")
    return y, mean, rstd


def k_layernorm_fwd_multi(x: torch.Tensor, adds, eps: float, side: int = 1, grid_r: int = 1):
    """cmb_layernorm_fwd_multi: [(y_l, mean_l, rstd_l)] = the non-affine LayerNorms of x + adds[l][window position] for every
    table of ``adds`` (fp32 [grid_r^2, D] or None) in one pass over x; each triple equals k_layernorm_fwd(x, None, None, eps,
    add=adds[l], ...) bit for bit."""
    L.require_gpu(x, *[a for a in adds if a is not None])
    rows, D = x.shape
    out = []
    for lo in range(0, len(adds), L.LN_MULTI_MAX):
        chunk = adds[lo:lo + L.LN_MULTI_MAX]
        d = L.LnFwdMultiDesc()
        d.dtype, d.layers = L.dtype_code(x.dtype), len(chunk)
        d.x, d.ldx, d.rows, d.D, d.side, d.grid_r, d.eps = x.data_ptr(), x.stride(0), rows, D, side, grid_r, eps
        for i, a in enumerate(chunk):
            if a is not None and (a.dtype != torch.float32 or not a.is_contiguous() or tuple(a.shape) != (grid_r * grid_r, D)):
                raise L.CambrianAmdError("layernorm_fwd_multi: every table must be a dense fp32 [grid_r^2, D] tensor")
            y = torch.empty((rows, D), dtype=x.dtype, device=x.device)
            mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
            rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
            d.add[i] = None if a is None else a.data_ptr()
            d.y[i], d.mean[i], d.rstd[i] = y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            out.append((y, mean, rstd))
        L.check(L.load().cmb_layernorm_fwd_multi(C.byref(d), L.stream_ptr(x.device)), "cmb_layernorm_fwd_multi")
    return out


def k_row_stats(x: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean, rstd) fp32 [rows] of a LayerNorm over the rows of x — no normalised output (cmb_row_stats)."""
    L.require_gpu(x)
    rows, D = x.shape
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rc = L.load().cmb_row_stats(L.dtype_code(x.dtype), x.data_ptr(), rows, D, x.stride(0), eps, mean.data_ptr(), rstd.data_ptr(),
                                L.stream_ptr(x.device))
    L.check(rc, "cmb_row_stats")
    return mean, rstd


def k_layernorm_bwd(dy, x, mean, rstd, gamma=None, add=None, side=1, grid_r=1, dx_acc: Optional[torch.Tensor] = None,
                    want_dgamma=True, want_dadd=True):
    """Returns (dx, dgamma, dbeta, dadd).  With ``dx_acc`` (fp32 [rows,D]) dx is accumulated there and the
    returned dx is None."""
    L.require_gpu(dy, x, mean, rstd, gamma, add, dx_acc)
    rows, D = x.shape
    dev = x.device
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    dgamma = torch.zeros((D,), dtype=torch.float32, device=dev) if (gamma is not None and want_dgamma) else None
    dbeta = torch.zeros((D,), dtype=torch.float32, device=dev) if (gamma is not None and want_dgamma) else None
    dadd = torch.zeros_like(add) if (add is not None and want_dadd) else None
    if dx_acc is not None:
        dx, acc = dx_acc, 1
    else:
        dx, acc = torch.empty((rows, D), dtype=x.dtype, device=dev), 0
    rc = L.load().cmb_layernorm_bwd(L.dtype_code(x.dtype), dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), rows, D,
                                    L.ptr(add), side, grid_r, L.ptr(gamma), mean.data_ptr(), rstd.data_ptr(),
                                    dx.data_ptr(), D, acc, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dadd),
                                    L.stream_ptr(dev))
    L.check(rc, "cmb_layernorm_bwd")
    return (None if dx_acc is not None else dx), dgamma, dbeta, dadd


def _fill_sva(desc: SvaDesc, q, kvs, masks, r_list, B, qside, heads, hd, window_major=False):
    desc.dtype = L.dtype_code(q.dtype)
    desc.window_major = 1 if window_major else 0
    desc.B, desc.qside, desc.heads, desc.hd, desc.ntowers = B, qside, heads, hd, len(kvs)
    desc.q, desc.ldq = q.data_ptr(), q.stride(0)
    for i, (kv, r) in enumerate(zip(kvs, r_list)):
        G = qside * r
        if kv.dtype != q.dtype or tuple(kv.shape) != (B * G * G, 2 * heads * hd) or kv.stride(1) != 1:
            raise L.CambrianAmdError(f"kv[{i}] must be [{B * G * G},{2 * heads * hd}] of {q.dtype}, got {tuple(kv.shape)}")
        desc.r[i] = r
        desc.kv[i], desc.ldkv[i] = kv.data_ptr(), kv.stride(0)
        m = masks[i] if masks is not None else None
        if m is not None:
            if m.dtype != torch.uint8 or not m.is_contiguous() or m.numel() != B * qside * qside * r * r:
                raise L.CambrianAmdError(f"mask[{i}] must be contiguous uint8 [{B * qside * qside},{r * r}]")
            desc.mask[i] = m.data_ptr()
        else:
            desc.mask[i] = None


def k_sva_attn_fwd(q, kvs, masks, r_list, B, qside, heads, hd, window_major=False):
    L.require_gpu(q, *kvs)
    d = SvaDesc()
    _fill_sva(d, q, kvs, masks, r_list, B, qside, heads, hd, window_major)
    Bq = B * qside * qside
    out = torch.empty((Bq, heads * hd), dtype=q.dtype, device=q.device)
    lse = torch.empty((Bq, heads), dtype=torch.float32, device=q.device)
    d.out, d.ldo, d.lse = out.data_ptr(), out.stride(0), lse.data_ptr()
    rc = L.load().cmb_sva_attn_fwd(C.byref(d), L.stream_ptr(q.device))
    L.check(rc, "cmb_sva_attn_fwd")
    return out, lse


def k_sva_attn_bwd(dout, q, kvs, masks, r_list, out, lse, B, qside, heads, hd, window_major=False):
    d = SvaDesc()
    _fill_sva(d, q, kvs, masks, r_list, B, qside, heads, hd, window_major)
    if dout.stride(1) != 1:
        dout = dout.contiguous()
    dq = torch.empty_like(q)
    dkvs = [torch.empty_like(kv) for kv in kvs]
    d.out, d.ldo, d.lse = out.data_ptr(), out.stride(0), lse.data_ptr()
    d.dout, d.lddo = dout.data_ptr(), dout.stride(0)
    d.dq, d.lddq = dq.data_ptr(), dq.stride(0)
    for i, t in enumerate(dkvs):
        d.dkv[i] = t.data_ptr()
    rc = L.load().cmb_sva_attn_bwd(C.byref(d), L.stream_ptr(q.device))
    L.check(rc, "cmb_sva_attn_bwd")
    return dq, dkvs


# ================================================================================================
# prepared weights: bf16 copy + transposed bf16 copy of the trainable 2-D weights, refreshed in ONE launch per step
# ================================================================================================
# Round 5 cast every fp32 master per use (cmb_cast in LinearFn.forward: 171 launches per step) and transposed it again per
# backward (cmb_transpose for dx = g W: 166 launches) — 3.1 ms of 5-12 us kernels per step.  The model now opens a window per
# training forward (weight_step_begin / weight_step_end): begin() refreshes the copies of every weight registered so far with
# one cmb_weight_prep launch (one pass over the masters), the linears inside the window take them from the cache (a weight
# seen for the first time is prepared on the spot and registered), the backward uses what the forward captured.  Outside a
# window (direct calls of a layer, the re-computation forward of activation checkpointing, eval) nothing is cached: weights
# may have changed in ways no version counter shows (ZeRO-2 writes the all-gathered shards straight into the flat buckets).
# CAMBRIAN_AMD_PREP_WEIGHTS=0 restores the per-use launches (A/B runs).
PREP_WEIGHTS = os.environ.get("CAMBRIAN_AMD_PREP_WEIGHTS", "1") != "0"
_PREP: dict = {}
_PREP_EPOCH = 0
_PREP_ACTIVE = False
_PREP_TABLE = None   # (signature, device job table, n_jobs, total_tiles)
_PREP_TMP: dict = {}   # id(tensor) -> (weakref, w_c, w_t): per-step weights made inside a window (the folded K|V weights)


class _PrepEntry:
    __slots__ = ("ref", "off", "shape", "stride", "w_c", "w_t", "rows_pad", "epoch", "src_ptr")


def _prep_src_ptr(base: torch.Tensor, e: "_PrepEntry") -> int:
    return base.data_ptr() + (e.off - base.storage_offset()) * base.element_size()


def _prep_job(e: "_PrepEntry", base: torch.Tensor, tile0: int) -> "L.PrepJob":
    j = L.PrepJob()
    j.src, j.dst, j.dst_t = _prep_src_ptr(base, e), e.w_c.data_ptr(), e.w_t.data_ptr()
    j.ld_src, j.src_dtype = e.stride[0], L.dtype_code(base.dtype)
    j.rows, j.cols, j.rows_pad, j.tile0, j.reserved = e.shape[0], e.shape[1], e.rows_pad, tile0, 0
    return j


def weight_step_begin() -> None:
    """Open the prepared-weight window of a training forward and refresh every registered weight's copies in one launch."""
    global _PREP_EPOCH, _PREP_ACTIVE, _PREP_TABLE
    if not PREP_WEIGHTS:
        return
    _PREP_EPOCH += 1
    _PREP_ACTIVE = True
    _PREP_TMP.clear()
    live = []
    for key in list(_PREP):
        e = _PREP[key]
        base = e.ref()
        if base is None or not base.is_cuda:
            del _PREP[key]
            continue
        live.append((key, e, base))
    if not live:
        return
    dev = live[0][2].device
    sig = (dev, tuple((key, _prep_src_ptr(base, e), e.w_c.data_ptr()) for key, e, base in live))
    if _PREP_TABLE is None or _PREP_TABLE[0] != sig:
        jobs = (L.PrepJob * len(live))()
        tile0 = 0
        lib = L.load()
        for i, (key, e, base) in enumerate(live):
            jobs[i] = _prep_job(e, base, tile0)
            tile0 += lib.cmb_weight_prep_tiles(e.rows_pad, e.shape[1])
        table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        _PREP_TABLE = (sig, table, len(live), tile0)
    _, table, n, tiles = _PREP_TABLE
    L.check(L.load().cmb_weight_prep(table.data_ptr(), n, tiles, L.stream_ptr(dev)), "cmb_weight_prep")
    for key, e, base in live:
        e.epoch, e.src_ptr = _PREP_EPOCH, _prep_src_ptr(base, e)


def weight_step_end() -> None:
    global _PREP_ACTIVE
    _PREP_ACTIVE = False
    _PREP_TMP.clear()


def prepare_step_weight(w: torch.Tensor) -> None:
    """A weight computed inside the window (fp32 [N, K], e.g. the folded K|V projection): its bf16 copy and transposed copy in
    one launch, found by ``prepared_weight`` for ``w`` itself and for row slices of it (the absorbed tower's K / V halves)."""
    if not (_PREP_ACTIVE and w.is_cuda and w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float32 and w.shape[1] % 8 == 0):
        return
    import weakref
    N, K = w.shape
    w_c = torch.empty((N, K), dtype=torch.bfloat16, device=w.device)
    w_t = torch.empty((K, pad_to(N, 64)), dtype=torch.bfloat16, device=w.device)
    j = L.PrepJob()
    j.src, j.dst, j.dst_t, j.ld_src, j.src_dtype = w.data_ptr(), w_c.data_ptr(), w_t.data_ptr(), K, L.F32
    j.rows, j.cols, j.rows_pad, j.tile0, j.reserved = N, K, w_t.shape[1], 0, 0
    L.check(L.load().cmb_weight_prep_one(C.byref(j), L.stream_ptr(w.device)), "cmb_weight_prep_one")
    _PREP_TMP[id(w)] = (weakref.ref(w), w_c, w_t)


def prepared_weight(weight: torch.Tensor, dt: torch.dtype):
    """(bf16 copy [N, K], transposed bf16 copy [K, pad64(N)]) of a trainable weight inside a prepared-weight window, else None.
    Only nn.Parameters (or views of one) qualify: their identity is what the cache is keyed on."""
    if not (_PREP_ACTIVE and dt == torch.bfloat16 and weight.is_cuda and weight.dim() == 2 and weight.stride(1) == 1):
        return None
    base = weight._base if weight._base is not None else weight
    tmp = _PREP_TMP.get(id(base))
    if tmp is not None and tmp[0]() is base:   # a weight made inside this window, or rows [r0, r1) of it
        ldw = base.shape[1]
        if weight is base:
            return tmp[1], tmp[2]
        off = weight.storage_offset() - base.storage_offset()
        if weight.shape[1] == ldw and weight.stride(0) == ldw and off % ldw == 0:
            r0 = off // ldw
            return tmp[1][r0:r0 + weight.shape[0]], tmp[2][:, r0:r0 + weight.shape[0]]
        return None
    if not isinstance(base, torch.nn.Parameter) or base.dtype not in (torch.float32, torch.bfloat16):
        return None
    N, K = weight.shape
    if K % 8 or weight.stride(0) % 8 or weight.data_ptr() % 16:
        return None
    key = (id(base), weight.storage_offset(), (N, K), tuple(weight.stride()))
    e = _PREP.get(key)
    if e is not None and e.ref() is not base:
        e = None   # the id was recycled
    if e is None:
        import weakref
        e = _PrepEntry()
        e.ref = weakref.ref(base)
        e.off, e.shape, e.stride, e.rows_pad, e.epoch, e.src_ptr = weight.storage_offset(), (N, K), tuple(weight.stride()), pad_to(N, 64), -1, 0
        e.w_c = torch.empty((N, K), dtype=torch.bfloat16, device=weight.device)
        e.w_t = torch.empty((K, e.rows_pad), dtype=torch.bfloat16, device=weight.device)
        _PREP[key] = e
    if e.epoch != _PREP_EPOCH or e.src_ptr != weight.data_ptr():   # first sight in this window: prepare it on the spot
        j = _prep_job(e, base, 0)
        L.check(L.load().cmb_weight_prep_one(C.byref(j), L.stream_ptr(weight.device)), "cmb_weight_prep_one")
        e.epoch, e.src_ptr = _PREP_EPOCH, weight.data_ptr()
    return e.w_c, e.w_t


# ================================================================================================
# autograd: linear
# ================================================================================================
# CAMBRIAN_AMD_TN_WGRAD=0: weight gradients through transposed copies + the NT kernels (A/B runs, tests of the old path)
TN_WGRAD = os.environ.get("CAMBRIAN_AMD_TN_WGRAD", "1") != "0"


def _tn_wgrad_wins(rows: int, n_out: int, k_in: int) -> bool:
    """Where dW = g^T x on cmb_gemm_tn beats transposed copies + the 256-tile NT kernels (tools/bench_tn.py,
    profiles/r03c_wgrad_tn_vs_nt.md): every weight of up to 256 output tiles (4096 x 1024), with one exception — the
    ConvNeXt-side projector's 1024 x 3072 at 147456 rows (192 tiles: two row slices of 74 k rows each run 1.5 rounds; 1556
    vs 1323 us).  The 4096 x 4096 mm_projector weight (1024 tiles, no split) stays on the 4-wave NT kernel (341 vs 303 us)."""
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    return tiles <= 256 and not (rows > 65536 and tiles > 128)


def _tn_splits(m_cols: int, n_cols: int, rows: int, batch: int = 1) -> int:
    """Split-K factor of a cmb_gemm_tn weight gradient: as many row slices as fill ONE round of the kernel's two workgroups
    per CU (512 on the 256-CU part).  Measured (13824 x 1024 x 1024: 2 / 4 / 8 / 12 / 16 slices = 112 / 67 / 52 / 65 / 67 us;
    2048 x 1024: best at 4; 1024 x 1152: best at 6): one full round beats 1.5 rounds with smaller slabs."""
    tiles = batch * ((m_cols + 127) // 128) * ((n_cols + 127) // 128)
    return max(1, min(512 // max(tiles, 1), rows // 128, 64))


def _wgrad_splits(n_out: int, k_in: int, m_pad: int, kstep: int) -> int:
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    want = max(1, 768 // tiles)
    return max(1, min(want, m_pad // kstep // 4, 64))


def _as_dtype_contig(t: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    if t.dtype == dt and t.is_contiguous():
        return t
    return k_cast(t.contiguous(), dt)


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) (* colscale) (+ residual).  x: [M,K] compute dtype; W: [N,K] fp32 master (or
    compute dtype); gradients for W / b come out in fp32 straight from the fp32 accumulators.
    ``res_rep`` > 0 means residual is [M/res_rep, N] and output row r adds residual[r // res_rep]
    (the per-image context term of proj_in, vision_sampler.py:279-292, never materialised per query)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: int, residual, res_rep: int, colscale, heavy: bool = False, link=None, role: int = 0):
        dt = x.dtype
        # ``link`` (a dict private to one block) joins the two uses of a block's input t — y1 = linear(t, ...) (role 1, the
        # "sink") ... out = linear(h, ..., residual=t) (role 2, the "source"): autograd would sum d(t) = g1 W1 + d(out) with an
        # ATen add over [rows, width]; the source's backward (it always runs first: its input descends from the sink's output)
        # parks d(out) in the link instead of returning it and the sink's d(x) GEMM adds it in its epilogue.
        ctx.link = None
        if link is not None and LINK_RESIDUAL_GRADS:
            if role == 1 and ctx.needs_input_grad[0]:
                link["armed"] = (x.data_ptr(), tuple(x.shape))
                ctx.link = link
            elif role == 2 and residual is not None and res_rep == 0 and ctx.needs_input_grad[4] and residual.dtype == dt \
                    and residual.is_contiguous() and link.get("armed") == (residual.data_ptr(), tuple(residual.shape)):
                ctx.link = link
        ctx.role = role
        prep = prepared_weight(weight, dt)
        w_c, ctx.w_t = prep if prep is not None else (k_cast(weight, dt), None)
        b_c = None if bias is None else k_cast(bias, torch.float32)
        need_pre = act != L.ACT_NONE and (x.requires_grad or weight.requires_grad
                                          or (bias is not None and bias.requires_grad))
        pre = torch.empty((x.shape[0], weight.shape[0]), dtype=dt, device=x.device) if need_pre else None
        r_map = None
        if residual is not None:
            residual = _as_dtype_contig(residual, dt)
            if res_rep > 0:
                if residual.dim() != 2 or residual.shape[0] * res_rep != x.shape[0]:
                    raise L.CambrianAmdError("broadcast residual must be [M/res_rep, N]")
                r_map = L.make_map(res_rep, 1, residual.stride(0), 0, 0)
        if heavy and fp8_linear_eligible(x, weight):
            xq, xinv = k_quantize_fp8_rows(x if x.is_contiguous() else x.contiguous())
            wq, winv = _fp8_weight(weight)
            y = k_gemm_fp8(xq, xinv, wq, winv, bias=b_c, act=act, residual=residual, r_map=r_map, colscale=colscale,
                           pre_out=pre)
        else:
            y = k_gemm(x, w_c, bias=b_c, act=act, residual=residual, r_map=r_map, colscale=colscale, pre_out=pre)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.res_rep = res_rep
        ctx.save_for_backward(x, w_c, pre, colscale)
        ctx.w_dtype = weight.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_c, pre, colscale = ctx.saved_tensors
        dt = x.dtype
        dy = _as_dtype_contig(dy, dt)
        need_x, need_w, need_b, _, need_res = ctx.needs_input_grad[:5]
        M, K = x.shape
        N = w_c.shape[0]
        dres = None
        if need_res:
            if ctx.link is not None and ctx.role == 2:
                ctx.link["g"] = dy                      # handed to the sink's d(x) GEMM (see forward)
            else:
                dres = k_segment_sum(dy, ctx.res_rep) if ctx.res_rep > 0 else dy
        g = dy
        if colscale is not None:
            g = dy * colscale.to(dt)  # LayerScale belongs to the frozen towers; kept for completeness
        if ctx.act != L.ACT_NONE:
            gp = torch.empty_like(pre)
            rc = L.load().cmb_act_bwd(L.dtype_code(dt), ctx.act, g.data_ptr(), pre.data_ptr(), pre.numel(), gp.data_ptr(),
                                      L.stream_ptr(g.device))
            L.check(rc, "cmb_act_bwd")
            g = gp
        dx = dw = db = None
        ks = _kstep(dt)
        if need_x:
            n_pad = pad_to(N, ks)
            w_t = ctx.w_t if (ctx.w_t is not None and ctx.w_t.shape[1] == n_pad) else k_transpose(w_c, n_pad)  # [K, N_pad]
            parked = ctx.link.pop("g", None) if (ctx.link is not None and ctx.role == 1) else None
            if n_pad != N:
                gpad = torch.zeros((M, n_pad), dtype=dt, device=g.device)
                gpad[:, :N] = g
                dx = k_gemm(gpad, w_t, residual=parked)
            else:
                dx = k_gemm(g, w_t, residual=parked)
        elif ctx.link is not None and ctx.role == 1 and "g" in ctx.link:
            dx = ctx.link.pop("g")                       # (never expected: the sink was armed because x requires grad)
        if need_w:
            if (dt == torch.bfloat16 and M > 0 and N % 8 == 0 and K % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0
                    and x.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and g.stride(1) == 1 and g.stride(0) % 8 == 0
                    and TN_WGRAD and _tn_wgrad_wins(M, N, K)):  # everything cmb_gemm_tn checks (gemm_tn_dispatch)
                # dW = g^T x with g and x as they lie (cmb_gemm_tn: transposing LDS reads, rows beyond M contribute zero)
                dw = k_gemm_tn(g, x, split_k=_tn_splits(N, K, M))
            else:  # fp32 parity path / odd widths: transposed copies for the NT kernel
                m_pad = pad_to(M, ks)
                g_t = k_transpose(g, m_pad)  # [N, M_pad]
                x_t = k_transpose(x, m_pad)  # [K, M_pad]
                dw = k_gemm(g_t, x_t, out_dtype=torch.float32, split_k=_wgrad_splits(N, K, m_pad, ks))
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        if need_b and ctx.has_bias:
            db = k_colsum(g)
            if ctx.b_dtype != torch.float32:
                db = db.to(ctx.b_dtype)
        return dx, dw, db, None, dres, None, None, None, None, None


# CAMBRIAN_AMD_LINK_RESIDUAL_GRADS=0: autograd sums the two gradients of a block's input itself (A/B runs)
LINK_RESIDUAL_GRADS = os.environ.get("CAMBRIAN_AMD_LINK_RESIDUAL_GRADS", "1") != "0"


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE,
           residual: Optional[torch.Tensor] = None, res_rep: int = 0,
           colscale: Optional[torch.Tensor] = None, heavy: bool = False, link: Optional[dict] = None, role: int = 0) -> torch.Tensor:
    """2-D linear on the HIP GEMM.  x [M,K] -> [M,N].  ``heavy`` marks the KV-side projections (aux projectors, SVA
    K/V projections: M = every tower token, >85 % of the SVA-side FLOPs) — the ones ``fp8_projections`` moves to the
    fp8 MFMA; the query-side GEMMs (M = 576 per image) stay bf16.  ``link`` / ``role``: see LinearFn.forward."""
    return LinearFn.apply(x, weight, bias, act, residual, res_rep, colscale, heavy, link, role)


# ================================================================================================
# autograd: LayerNorm (affine) and the SVA "normalise-only, shared input" variant
# ================================================================================================
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        g32 = k_cast(gamma, torch.float32)
        b32 = k_cast(beta, torch.float32)
        y, mean, rstd = k_layernorm_fwd(x, g32, b32, eps)
        ctx.save_for_backward(x, g32, mean, rstd)
        ctx.p_dtype = gamma.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, mean, rstd = ctx.saved_tensors
        dy = _as_dtype_contig(dy, x.dtype)
        dx, dgamma, dbeta, _ = k_layernorm_bwd(dy, x, mean, rstd, gamma=g32)
        if ctx.p_dtype != torch.float32:
            dgamma, dbeta = dgamma.to(ctx.p_dtype), dbeta.to(ctx.p_dtype)
        return dx, dgamma, dbeta, None


def layernorm(x, gamma, beta, eps: float = 1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


DEFER_LN_BWD = os.environ.get("CAMBRIAN_AMD_DEFER_LN_BWD", "1") != "0"   # (A/B runs: 0 = one LayerNorm backward per SVA layer)
# parked layers per flush (= the 5-layer launches of cmb_layernorm_bwd_multi, CMB_KNOB_LN_MULTI_CHUNK; 0 = only when the
# shared-gradient node runs)
DEFER_LN_FLUSH = int(os.environ.get("CAMBRIAN_AMD_DEFER_LN_FLUSH", "5"))


# the last multi-layer launch of a tower's deferred LayerNorm backward writes d(x) in x's dtype (no separate cast of the fp32
# accumulator); CAMBRIAN_AMD_LN_BWD_FINAL_CAST=0: accumulator + cmb_cast as before (A/B runs)
LN_BWD_FINAL_CAST = os.environ.get("CAMBRIAN_AMD_LN_BWD_FINAL_CAST", "1") != "0"


class GradAccumulator:
    """fp32 side buffer that the 13 SVA layers' LayerNorm backwards accumulate into (all layers read the
    same aux feature tensor; SURVEY.md §7 "hard parts").

    Round 4: the layers' LayerNorm backwards can be DEFERRED — each SvaNormFn.backward parks (d xh_l, mean_l, rstd_l, pos_l)
    here instead of running, and SharedGradFn.backward (which autograd runs after all of them) does them in ONE pass over x
    (cmb_layernorm_bwd_multi: 32 instead of 156 bytes per element for 13 layers).  The gradients of the position tables come
    out of that pass, so they are returned by SharedGradFn.backward: the tables are formally inputs of shared_grad()
    (``pos_params``) and SvaNormFn returns no gradient for them.  A layer whose table was not announced runs the immediate
    path."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.shape = None
        self.defer = False
        self.pos_index = {}     # id() of an announced position table (the Parameter object) -> its slot in SharedGradFn's inputs
        self.deferred = []      # (dn, x, mean, rstd, add fp32 | None, side, grid_r, slot | -1)
        self.pos_meta = []      # (shape, dtype, needs grad) per announced table
        self.dpos = []          # fp32 table gradients filled by flush()
        self.pos_params = ()    # the announced tables themselves (SvaNormFn.forward: all layers' normalisations in one pass)
        self.fwd_key = None     # (x identity, side, grid_r, eps) of the one multi-layer forward this holder has run
        self.fwd_cache = {}     # slot -> (n, mean, rstd, add) not yet handed to its layer; "nopos" -> the table-less result

    def get(self, rows: int, D: int, device) -> torch.Tensor:
        if self.buf is None:
            self.buf = torch.zeros((rows, D), dtype=torch.float32, device=device)
        return self.buf

    def flush(self, final_dtype: Optional[torch.dtype] = None) -> Optional[torch.Tensor]:
        """Run the parked LayerNorm backwards (one cmb_layernorm_bwd_multi call per window geometry) into the shared fp32
        buffer and release their gradient tensors.  Called by SvaNormFn.backward whenever DEFER_LN_FLUSH layers are parked
        (= the kernel's 5-layer launches: same launches as one call at the end, but at most 5 layers' d(x-hat) — 2.3 GB at
        24 images instead of 5.9 GB — are alive at a time) and by SharedGradFn.backward for the rest."""
        items, self.deferred = self.deferred, []
        if not items:
            return None
        x = items[0][1]
        groups = {}
        for dn, _x, mean, rstd, add, side, grid_r, slot in items:
            if slot >= 0 and self.dpos[slot] is None and self.pos_meta[slot][2]:
                self.dpos[slot] = torch.zeros(self.pos_meta[slot][0], dtype=torch.float32, device=x.device)
            groups.setdefault((side, grid_r) if add is not None else (1, 1), []).append((dn, mean, rstd, add, slot))
        have = self.buf is not None
        if self.buf is None:
            self.buf = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        # ``final_dtype`` (SharedGradFn.backward's last flush): the kernel's last launch writes the finished sum in that dtype —
        # the fp32 accumulator's cast (906 MB read + 453 MB written for the 9216-token tower at 24 images) and the launch's own
        # fp32 write are gone.  Returns that tensor (the accumulator is then stale), else None.
        out = None
        if final_dtype is not None and final_dtype == x.dtype and final_dtype != torch.float32 and len(groups) == 1:
            out = torch.empty((x.shape[0], x.shape[1]), dtype=final_dtype, device=x.device)
        for (side, grid_r), its in groups.items():
            k_layernorm_bwd_multi(x, its, side, grid_r, self.buf, have, self.dpos, dx_out=out)
            have = True
        return out


def k_layernorm_bwd_multi(x: torch.Tensor, items, side: int, grid_r: int, dx: torch.Tensor, accumulate: bool, dadd_out,
                          dx_out: Optional[torch.Tensor] = None):
    """cmb_layernorm_bwd_multi over ``items`` = [(dn, mean, rstd, add | None, slot)]: dx (fp32 [rows, D]) (+)= the summed
    LayerNorm backwards; ``dadd_out[slot]`` (fp32, zero-filled, same shape as the table) receives each layer's table gradient.
    ``dx_out`` (x's dtype, dense [rows, D]): the last launch writes the finished sum THERE instead of updating dx."""
    L.require_gpu(x, dx, dx_out)
    rows, D = x.shape
    if dx_out is not None and (dx_out.dtype != x.dtype or not dx_out.is_contiguous() or tuple(dx_out.shape) != (rows, D)):
        raise L.CambrianAmdError("layernorm_bwd_multi: dx_out must be a dense [rows, D] tensor of x's dtype")
    for lo in range(0, len(items), L.LN_MULTI_MAX):
        chunk = items[lo:lo + L.LN_MULTI_MAX]
        d = L.LnMultiDesc()
        d.dtype, d.layers = L.dtype_code(x.dtype), len(chunk)
        d.x, d.ldx, d.rows, d.D, d.side, d.grid_r = x.data_ptr(), x.stride(0), rows, D, side, grid_r
        for i, (dn, mean, rstd, add, slot) in enumerate(chunk):
            if dn.dtype != x.dtype or not dn.is_contiguous() or tuple(dn.shape) != (rows, D):
                raise L.CambrianAmdError("layernorm_bwd_multi: every gradient must be a dense [rows, D] tensor of x's dtype")
            d.dy[i], d.mean[i], d.rstd[i] = dn.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            d.add[i] = None if add is None else add.data_ptr()
            d.dadd[i] = None if (add is None or slot < 0 or dadd_out[slot] is None) else dadd_out[slot].data_ptr()
        d.dx, d.lddx, d.accumulate = dx.data_ptr(), dx.stride(0), 1 if (accumulate or lo > 0) else 0
        d.dx_out = dx_out.data_ptr() if (dx_out is not None and lo + L.LN_MULTI_MAX >= len(items)) else None
        L.check(L.load().cmb_layernorm_bwd_multi(C.byref(d), L.stream_ptr(x.device)), "cmb_layernorm_bwd_multi")


class SharedGradFn(torch.autograd.Function):
    """Identity whose backward hands out the accumulator filled by the consumers (which themselves return
    no gradient for this tensor).  Autograd runs this node only after every consumer node has run, because
    dependencies are counted on graph edges, not on defined gradients.  ``pos_params``: the position tables of the SVA
    layers that will normalise this tensor (GradAccumulator: deferred LayerNorm backwards)."""

    @staticmethod
    def forward(ctx, x, holder: GradAccumulator, *pos_params):
        ctx.holder = holder
        holder.shape = x.shape
        holder.defer = DEFER_LN_BWD
        holder.deferred = []
        holder.pos_meta = [(tuple(p.shape), p.dtype, bool(ctx.needs_input_grad[2 + i])) for i, p in enumerate(pos_params)]
        holder.dpos = [None] * len(pos_params)
        ctx.x_dtype = x.dtype
        ctx.n_pos = len(pos_params)
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        done = h.flush(final_dtype=ctx.x_dtype if (g is None and LN_BWD_FINAL_CAST) else None)
        buf, h.buf = h.buf, None
        dpos = [None if t is None else (t if h.pos_meta[i][1] == torch.float32 else t.to(h.pos_meta[i][1]))
                for i, t in enumerate(h.dpos)] + [None] * (ctx.n_pos - len(h.dpos))
        h.dpos = [None] * ctx.n_pos
        if done is not None:     # the last multi-layer launch already wrote d(x) in x's dtype
            return (done.view(h.shape), None, *dpos)
        if buf is None:
            return (g, None, *dpos)
        if g is not None:
            buf = buf + g.to(torch.float32).reshape(buf.shape)
        return (k_cast(buf, ctx.x_dtype).view(h.shape), None, *dpos)


def shared_grad(x: torch.Tensor, holder: GradAccumulator, pos_params=()) -> torch.Tensor:
    # tables are recognised by the IDENTITY of the tensor object handed to sva_norm(), not by their storage address: a
    # sharded / re-materialised parameter (ZeRO-3: storage.resize_(0), every released table at data_ptr 0) keeps its id but
    # not its address, and two tables must never alias
    pos_params = tuple(pos_params)
    index = {id(p): i for i, p in enumerate(pos_params)}
    if len(index) != len(pos_params):
        raise L.CambrianAmdError("shared_grad: the same position table was announced twice")
    holder.pos_index = index
    holder.pos_params = pos_params
    holder.fwd_key, holder.fwd_cache = None, {}
    return SharedGradFn.apply(x, holder, *pos_params)


# The 13 SVA layers normalise the SAME aux features; only the position table differs (vision_sampler.py:304-309), and a one-key
# tower has no table at all: its 13 normalisations are one and the same tensor.  Round 6: the first layer that asks runs
# cmb_layernorm_fwd_multi over every table announced to shared_grad() (x read once instead of 13 times) and parks the other
# layers' results in the holder; a table-less result is computed once per holder and shared.  Bit-identical to the per-layer
# launches.  A slot is served from the multi-layer pass at most once per holder: re-computation forwards (activation
# checkpointing) and geometries that differ from the first request's take the per-layer kernel.  The normalised tensors were
# alive until the backward anyway (the attention core saves them).  CAMBRIAN_AMD_LN_FWD_MULTI=0: per-layer launches (A/B runs).
LN_FWD_MULTI = os.environ.get("CAMBRIAN_AMD_LN_FWD_MULTI", "1") != "0"


def _materialised(p: torch.Tensor) -> bool:
    return p.is_cuda and p.numel() > 0 and p.untyped_storage().size() >= (p.storage_offset() + p.numel()) * p.element_size()


def _sva_norm_forward(x, add, holder: "GradAccumulator", side: int, grid_r: int, eps: float, pos_key):
    if not (LN_FWD_MULTI and x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[1] <= 1024 and holder.pos_index is not None):
        return k_layernorm_fwd(x, None, None, eps, add=add, side=side, grid_r=grid_r)
    xid = (x.data_ptr(), x._version, tuple(x.shape), x.dtype)
    if add is None:   # every layer's normalisation of a table-less tower is the same tensor: one launch per holder
        hit = holder.fwd_cache.get("nopos")
        if hit is None or hit[0] != (xid, eps):
            hit = ((xid, eps), k_layernorm_fwd(x, None, None, eps))
            holder.fwd_cache["nopos"] = hit
        n, mean, rstd = hit[1]
        return n.detach(), mean, rstd          # a fresh tensor object per autograd node, the same storage
    slot = holder.pos_index.get(pos_key, -1)
    key = (xid, side, grid_r, eps)
    if slot >= 0 and holder.fwd_key is None and len(holder.pos_params) > 1 and all(_materialised(p) for p in holder.pos_params):
        adds = [k_cast(p.detach(), torch.float32) for p in holder.pos_params]
        if all(tuple(a.shape) == (grid_r * grid_r, x.shape[1]) for a in adds):
            holder.fwd_key = key
            for i, res in enumerate(k_layernorm_fwd_multi(x, adds, eps, side, grid_r)):
                holder.fwd_cache[i] = res
    if slot >= 0 and holder.fwd_key == key:
        hit = holder.fwd_cache.pop(slot, None)
        if hit is not None:
            return hit
    return k_layernorm_fwd(x, None, None, eps, add=add, side=side, grid_r=grid_r)


class SvaNormFn(torch.autograd.Function):
    """n = (x + pos[window_pos]) normalised without affine (the K- and V-LayerNorm affines of
    vision_sampler.py:173-174 are folded into the projection weights).  The gradient w.r.t. x goes into
    the shared fp32 accumulator instead of being returned."""

    @staticmethod
    def forward(ctx, x, pos, holder: GradAccumulator, side: int, grid_r: int, eps: float, pos_key=None):
        add = None if pos is None else k_cast(pos, torch.float32)
        n, mean, rstd = _sva_norm_forward(x, add, holder, side, grid_r, eps, pos_key)
        ctx.holder, ctx.side, ctx.grid_r = holder, side, grid_r
        ctx.pos_key = pos_key          # id() of the table object the caller passed (sva_norm)
        ctx.pos_dtype = None if pos is None else pos.dtype
        ctx.has_pos = pos is not None
        if pos is None:
            ctx.save_for_backward(x, mean, rstd)
        else:
            ctx.save_for_backward(x, mean, rstd, add)
        return n

    @staticmethod
    def backward(ctx, dn):
        if ctx.has_pos:
            x, mean, rstd, add = ctx.saved_tensors
        else:
            (x, mean, rstd), add = ctx.saved_tensors, None
        dn = _as_dtype_contig(dn, x.dtype)
        need_x, need_pos = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        h = ctx.holder
        if need_x and h.defer and x.is_contiguous() and x.shape[1] <= 1024:
            slot = -1 if add is None else h.pos_index.get(ctx.pos_key, -2)
            if slot != -2 and not (h.deferred and h.deferred[0][1].data_ptr() != x.data_ptr()):
                # parked: SharedGradFn.backward runs every layer's LayerNorm backward in one pass and returns d(pos)
                h.deferred.append((dn, x, mean, rstd, add, ctx.side, ctx.grid_r, slot))
                if DEFER_LN_FLUSH > 0 and len(h.deferred) >= DEFER_LN_FLUSH:
                    h.flush()
                return None, None, None, None, None, None, None
        dpos = None
        if need_x:
            acc = h.get(x.shape[0], x.shape[1], x.device)
            _, _, _, dpos = k_layernorm_bwd(dn, x, mean, rstd, add=add, side=ctx.side, grid_r=ctx.grid_r, dx_acc=acc,
                                            want_dadd=need_pos)
        elif need_pos and add is not None:
            _, _, _, dpos = k_layernorm_bwd(dn, x, mean, rstd, add=add, side=ctx.side, grid_r=ctx.grid_r)
        if dpos is not None and ctx.pos_dtype != torch.float32:
            dpos = dpos.to(ctx.pos_dtype)
        return None, dpos, None, None, None, None, None


def sva_norm(x, pos, holder, side, grid_r, eps=1e-5):
    return SvaNormFn.apply(x, pos, holder, side, grid_r, eps, None if pos is None else id(pos))


# ================================================================================================
# autograd: K|V weight folding of one tower (vision_sampler.py:173-174,188-189)
# ================================================================================================
class FoldKVFn(torch.autograd.Function):
    """(Wk, gk, bk, Wv, gv, bv) -> ([Wk*gk ; Wv*gv] [2H, K], [Wk@bk ; Wv@bv] [2H]) in fp32, two HIP launches per step."""

    @staticmethod
    def forward(ctx, wk, gk, bk, wv, gv, bv):
        ts = [t.detach().contiguous() for t in (wk, gk, bk, wv, gv, bv)]
        for t in ts:
            L.require_gpu(t)
            if t.dtype != torch.float32:
                raise L.CambrianAmdError("fold_kv: fp32 tensors expected (ops.fold_kv up-casts other parameter dtypes)")
        H, K = ts[0].shape
        w = torch.empty((2 * H, K), dtype=torch.float32, device=ts[0].device)
        b = torch.empty((2 * H,), dtype=torch.float32, device=ts[0].device)
        rc = L.load().cmb_sva_fold_kv_fwd(*[t.data_ptr() for t in ts], H, K, w.data_ptr(), b.data_ptr(), L.stream_ptr(w.device))
        L.check(rc, "cmb_sva_fold_kv_fwd")
        ctx.save_for_backward(*ts)
        return w, b

    @staticmethod
    def backward(ctx, dw, db):
        ts = ctx.saved_tensors
        H, K = ts[0].shape
        dev = ts[0].device
        dw = torch.zeros((2 * H, K), dtype=torch.float32, device=dev) if dw is None else dw.contiguous().float()
        db = torch.zeros((2 * H,), dtype=torch.float32, device=dev) if db is None else db.contiguous().float()
        outs = [torch.empty_like(t) for t in ts]
        ws = torch.empty((L.load().cmb_sva_fold_kv_bwd_workspace(H, K) // 4,), dtype=torch.float32, device=dev)
        rc = L.load().cmb_sva_fold_kv_bwd(dw.data_ptr(), db.data_ptr(), *[t.data_ptr() for t in ts], H, K,
                                          *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel() * 4, L.stream_ptr(dev))
        L.check(rc, "cmb_sva_fold_kv_bwd")
        return tuple(outs)


def fold_kv(wk, gk, bk, wv, gv, bv):
    """Folded K|V weight / bias in fp32.  Parameters that are not fp32 masters (a module moved with ``.to(bfloat16)`` /
    ``.half()``) are up-cast by autograd-tracked ``.float()`` first, as the torch expression this replaces did."""
    ts = [t if t.dtype == torch.float32 else t.float() for t in (wk, gk, bk, wv, gv, bv)]
    w, b = FoldKVFn.apply(*ts)
    prepare_step_weight(w)   # (inside a training window: the bf16 copies the consuming linears would otherwise make per use)
    return w, b


# ================================================================================================
# autograd: SVA attention core
# ================================================================================================
class SvaAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, B: int, qside: int, heads: int, hd: int, r_list, masks, window_major: bool, *kvs):
        out, lse = k_sva_attn_fwd(q, list(kvs), masks, r_list, B, qside, heads, hd, window_major)
        ctx.cfg = (B, qside, heads, hd, tuple(r_list), window_major)
        ctx.masks = masks
        ctx.save_for_backward(q, out, lse, *kvs)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, out, lse, *kvs = ctx.saved_tensors
        B, qside, heads, hd, r_list, window_major = ctx.cfg
        dout = _as_dtype_contig(dout, q.dtype)
        dq, dkvs = k_sva_attn_bwd(dout, q, kvs, ctx.masks, list(r_list), out, lse, B, qside, heads, hd, window_major)
        return (dq, None, None, None, None, None, None, None, *dkvs)


def sva_attention(q, kvs: Sequence[torch.Tensor], masks, r_list, B, qside, heads=16, hd=64, window_major=False):
    return SvaAttnFn.apply(q, B, qside, heads, hd, list(r_list), masks, window_major, *kvs)


# ================================================================================================
# absorbed K / V projections of the windowed tower (csrc/sva_absorbed.hip)
# ================================================================================================
def k_gemm_batched(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, batch: int, M: int, N: int, K: int, lda: int,
                   ldb: int, ldc: int, a_bs: int, b_bs: int, c_bs: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``batch`` independent C_z[M,N] = A_z[M,K] @ B_z[N,K]^T in one launch (cmb_gemm, 128 x 128 tile): problem z reads
    A + z*a_bs (row stride lda), B + z*b_bs (row stride ldb) and writes C + z*c_bs (row stride ldc), all in elements of
    the tensors' dtype (``out`` may be fp32 for bf16 operands).  The per-head GEMMs of the absorbed SVA projections."""
    L.require_gpu(a, w, out)
    if a.dtype != w.dtype:
        raise L.CambrianAmdError("batched gemm operand dtypes differ")
    d = GemmDesc()
    d.dtype, d.out_dtype = L.dtype_code(a.dtype), L.dtype_code(out.dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.a_map = a.data_ptr(), L.identity_map(lda)
    d.B, d.ldb = w.data_ptr(), ldb
    d.C, d.c_map = out.data_ptr(), L.identity_map(ldc)
    d.bias = d.colscale = d.residual = d.pre_out = None
    d.r_map = d.p_map = L.identity_map(0)
    if residual is not None:   # laid out as ``out`` (row stride ldc, batch stride c_bs), the operands' dtype
        L.require_gpu(residual)
        if residual.dtype != a.dtype or out.dtype != a.dtype or residual.stride(-1) != 1 or residual.stride(0) != ldc \
                or residual.data_ptr() % 16:
            raise L.CambrianAmdError("batched gemm residual must have the operands' dtype and the output's layout")
        d.residual, d.r_map = residual.data_ptr(), L.identity_map(ldc)
    d.act, d.alpha, d.beta, d.split_k, d.tile_hint = L.ACT_NONE, 1.0, 0.0, 1, 0
    d.workspace, d.workspace_bytes = None, 0
    d.batch, d.a_batch_stride, d.b_batch_stride, d.c_batch_stride = batch, a_bs, b_bs, c_bs
    prof = GEMM_PROFILE if not GEMM_PROFILE_TILE else None
    if prof is not None:
        e0 = _event()
    rc = L.load().cmb_gemm(C.byref(d), L.stream_ptr(a.device))
    L.check(rc, f"cmb_gemm(batch={batch}, M={M}, N={N}, K={K})")
    if prof is not None:
        prof.append((e0, _event(), 2.0 * batch * M * N * K, a.dtype, 1, 128, (M, N, K, L.ACT_NONE, out.dtype == torch.float32),
                     L.load().cmb_gemm_last_kernel(), batch))
    return out


def _per_head_wgrad(x: torch.Tensor, dy: torch.Tensor, heads: int, hd: int, Cin: int) -> torch.Tensor:
    """dW[h*hd + j, :] = sum_q x[q, h*hd + j] dy[q, h, :]  ([heads*hd, Cin], fp32) — the weight gradient of the per-head
    projections.  bf16: one batched split-K cmb_gemm_tn on the operands as they lie; other dtypes (the fp32 parity path):
    transposed copies and one NT GEMM per head (cmb_gemm_tn is a bf16 kernel)."""
    Bq = x.shape[0]
    dw = torch.empty((heads * hd, Cin), dtype=torch.float32, device=x.device)
    if x.dtype == torch.bfloat16:
        k_gemm_tn(x, dy.view(Bq, heads * Cin), out=dw, M=hd, N=Cin, batch=heads, a_bs=hd, b_bs=Cin, c_bs=hd * Cin, ldc=Cin,
                  split_k=_tn_splits(hd, Cin, Bq, heads))
        return dw
    m_pad = pad_to(Bq, _kstep(x.dtype))
    x_t = k_transpose(x, m_pad)                                      # [heads*hd, m_pad]
    dy_t = k_transpose(dy.view(Bq, heads * Cin), m_pad)              # [heads*Cin, m_pad]
    for h in range(heads):
        k_gemm(x_t[h * hd:(h + 1) * hd], dy_t[h * Cin:(h + 1) * Cin], out=dw[h * hd:(h + 1) * hd])
    return dw


class HeadExpandFn(torch.autograd.Function):
    """U[q, h, :] = x[q, h*hd:(h+1)*hd] @ W[h*hd:(h+1)*hd, :]   (x [Bq, H*hd], W [H*hd, Cin] -> U [Bq, H, Cin]): the K
    projection of the windowed tower applied to the QUERY (U = W_k,h^T q_h) and, with x = d(o), the V projection's backward
    (d Xb = W_v,h^T d o_h).  H GEMMs with K = hd = 64 in one batched launch; backward: dx_h = dU_h W_h^T (N = 64),
    dW_h = x_h^T dU_h (M = 64, contraction over the queries: cmb_gemm_tn, batched)."""

    @staticmethod
    def forward(ctx, x, w, heads: int):
        Bq, C = x.shape
        hd, Cin = C // heads, w.shape[1]
        prep = prepared_weight(w, x.dtype)
        if prep is not None and prep[1].shape[1] == C:
            w_c, w_t = prep
        else:
            w_c = k_cast(w, x.dtype)
            w_t = k_transpose(w_c, C)                                # [Cin, C]: rows c, the head's 64 inputs contiguous
        U = torch.empty((Bq, heads, Cin), dtype=x.dtype, device=x.device)
        xc = x if x.is_contiguous() else x.contiguous()
        k_gemm_batched(xc, w_t, U, batch=heads, M=Bq, N=Cin, K=hd, lda=C, ldb=w_t.stride(0), ldc=heads * Cin, a_bs=hd, b_bs=hd,
                       c_bs=Cin)
        ctx.save_for_backward(xc, w_c)
        ctx.heads, ctx.w_dtype = heads, w.dtype
        return U

    @staticmethod
    def backward(ctx, dU):
        x, w_c = ctx.saved_tensors
        heads = ctx.heads
        Bq, C = x.shape
        hd, Cin = C // heads, w_c.shape[1]
        dU = _as_dtype_contig(dU, x.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((Bq, C), dtype=x.dtype, device=x.device)
            k_gemm_batched(dU, w_c, dx, batch=heads, M=Bq, N=hd, K=Cin, lda=heads * Cin, ldb=Cin, ldc=C, a_bs=Cin,
                           b_bs=hd * Cin, c_bs=hd)
        if ctx.needs_input_grad[1]:
            dw = _per_head_wgrad(x, dU, heads, hd, Cin)
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        return dx, dw, None


class HeadContractFn(torch.autograd.Function):
    """y[q, h*hd + j] = Xb[q, h, :] . W[h*hd + j, :]   (Xb [Bq, H, Cin], W [H*hd, Cin] -> y [Bq, H*hd]): the V projection of
    the windowed tower applied to the attention-weighted token mix (o_h = W_v,h Xb_h).  H GEMMs with N = hd = 64 in one
    batched launch; backward: dXb_h = dy_h W_h (K = 64), dW_h = dy_h^T Xb_h."""

    @staticmethod
    def forward(ctx, xb, w, heads: int, addend=None):
        """``addend`` [Bq, C] (optional): y = addend + the per-head products, added in the GEMM's epilogue (the one-key towers'
        part of the attention output: no separate add over [Bq, C])."""
        Bq, H, Cin = xb.shape
        C = w.shape[0]
        hd = C // heads
        prep = prepared_weight(w, xb.dtype)
        w_c, ctx.w_t = prep if (prep is not None and prep[1].shape[1] == C) else (k_cast(w, xb.dtype), None)
        xbc = xb if xb.is_contiguous() else xb.contiguous()
        y = torch.empty((Bq, C), dtype=xb.dtype, device=xb.device)
        if addend is not None:
            addend = _as_dtype_contig(addend, xb.dtype)
        k_gemm_batched(xbc, w_c, y, batch=heads, M=Bq, N=hd, K=Cin, lda=heads * Cin, ldb=Cin, ldc=C, a_bs=Cin,
                       b_bs=hd * Cin, c_bs=hd, residual=addend)
        ctx.save_for_backward(xbc, w_c)
        ctx.heads, ctx.w_dtype = heads, w.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, w_c = ctx.saved_tensors
        heads = ctx.heads
        Bq, H, Cin = xb.shape
        C = w_c.shape[0]
        hd = C // heads
        dy = _as_dtype_contig(dy, xb.dtype)
        dxb = dw = None
        if ctx.needs_input_grad[0]:
            w_t = ctx.w_t if ctx.w_t is not None else k_transpose(w_c, C)   # [Cin, C]
            dxb = torch.empty((Bq, heads, Cin), dtype=xb.dtype, device=xb.device)
            k_gemm_batched(dy, w_t, dxb, batch=heads, M=Bq, N=Cin, K=hd, lda=C, ldb=w_t.stride(0), ldc=heads * Cin, a_bs=hd,
                           b_bs=hd, c_bs=Cin)
        if ctx.needs_input_grad[1]:
            dw = _per_head_wgrad(dy, xb, heads, hd, Cin)
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        if len(ctx.needs_input_grad) < 4:     # called without an addend
            return dxb, dw, None
        return dxb, dw, None, (dy if ctx.needs_input_grad[3] else None)


def _fill_sva_abs(d, q, kvs, masks, xhat, mask_a, ra, U, bk, bv, B, qside, window_major):
    d.B, d.qside, d.heads, d.hd = B, qside, 16, 64
    d.dtype = L.dtype_code(q.dtype)
    d.ntowers, d.window_major, d.ra = len(kvs), 1 if window_major else 0, ra
    d.q, d.ldq = q.data_ptr(), q.stride(0)
    for i, kv in enumerate(kvs):
        d.r[i] = 1
        d.kv[i], d.ldkv[i] = kv.data_ptr(), kv.stride(0)
        d.mask[i] = None if masks[i] is None else masks[i].data_ptr()
    d.xhat, d.ldx = xhat.data_ptr(), xhat.stride(0)
    d.mask_a = None if mask_a is None else mask_a.data_ptr()
    d.U, d.bk, d.bv = U.data_ptr(), bk.data_ptr(), bv.data_ptr()


class SvaAbsorbedFn(torch.autograd.Function):
    """cmb_sva_abs_fwd / _bwd: per query the joint softmax over the one-key towers' projected K|V rows and the windowed
    tower's tokens (scored against U, plus b_k . q), returning (the direct towers' part of the output + m3 b_v [Bq, 1024],
    Xb [Bq, 16, 1024])."""

    @staticmethod
    def forward(ctx, q, U, bk, bv, xhat, B: int, qside: int, ra: int, masks, mask_a, window_major: bool, *kvs):
        L.require_gpu(q, U, bk, bv, xhat, *kvs)
        Bq = B * qside * qside
        if q.dtype not in (torch.bfloat16, torch.float32) or tuple(q.shape) != (Bq, 1024) or tuple(U.shape) != (Bq, 16, 1024):
            raise L.CambrianAmdError("absorbed SVA attention: bf16 / fp32, 16 heads x 64, 1024-wide features")
        if any(t.dtype != q.dtype for t in (U, xhat, *kvs)):
            raise L.CambrianAmdError("absorbed SVA attention: q, U, xhat and the K|V rows must share one dtype")
        if xhat.shape[0] != B * (qside * ra) ** 2 or xhat.shape[1] != 1024 or xhat.stride(1) != 1:
            raise L.CambrianAmdError("absorbed SVA attention: xhat must be [B*(qside*ra)^2, 1024]")
        # dense rows: the backward kernel writes dK|dV with the K|V row stride (cmb_sva_abs_desc has no separate lddkv) into
        # torch.empty_like(kv), which is contiguous whatever kv's strides were
        kvs = [kv.contiguous() for kv in kvs]
        Uc = U.contiguous()
        bkc, bvc = bk.detach().to(torch.float32).contiguous(), bv.detach().to(torch.float32).contiguous()
        out = torch.empty((Bq, 1024), dtype=q.dtype, device=q.device)
        xbar = torch.empty((Bq, 16, 1024), dtype=q.dtype, device=q.device)
        m3 = torch.empty((Bq, 16), dtype=torch.float32, device=q.device)
        P = torch.empty((Bq, 16, 20), dtype=torch.float32, device=q.device)   # [0, 4) direct keys, [4, 20) window tokens
        d = L.SvaAbsDesc()
        _fill_sva_abs(d, q, kvs, masks, xhat, mask_a, ra, Uc, bkc, bvc, B, qside, window_major)
        d.out, d.ldo, d.xbar, d.m3, d.P = out.data_ptr(), out.stride(0), xbar.data_ptr(), m3.data_ptr(), P.data_ptr()
        L.check(L.load().cmb_sva_abs_fwd(C.byref(d), L.stream_ptr(q.device)), "cmb_sva_abs_fwd")
        ctx.cfg = (B, qside, ra, window_major)
        ctx.masks, ctx.mask_a = masks, mask_a
        ctx.b_dtypes = (bk.dtype, bv.dtype)
        ctx.save_for_backward(q, Uc, bkc, bvc, xhat, P, m3, *kvs)
        ctx.mark_non_differentiable(m3)
        return out, xbar, m3

    @staticmethod
    def backward(ctx, dout, dxbar, _dm3):
        q, U, bk, bv, xhat, P, m3, *kvs = ctx.saved_tensors
        B, qside, ra, window_major = ctx.cfg
        Bq = q.shape[0]
        dev = q.device
        dout = torch.zeros_like(q) if dout is None else _as_dtype_contig(dout, q.dtype)
        dxbar = torch.zeros_like(U) if dxbar is None else _as_dtype_contig(dxbar, q.dtype)
        dq = torch.empty_like(q)
        dkvs = [torch.empty_like(kv) for kv in kvs]
        dU = torch.empty_like(U)
        dcb = torch.empty((Bq, 16), dtype=torch.float32, device=dev)
        dxhat = torch.empty((xhat.shape[0], 1024), dtype=q.dtype, device=dev)
        d = L.SvaAbsDesc()
        _fill_sva_abs(d, q, kvs, ctx.masks, xhat, ctx.mask_a, ra, U, bk, bv, B, qside, window_major)
        # (the forward outputs are not read by the backward kernel, but the descriptor's forward fields must be set)
        d.out, d.ldo, d.xbar, d.m3, d.P = dq.data_ptr(), dq.stride(0), dU.data_ptr(), dcb.data_ptr(), P.data_ptr()
        d.dout, d.lddo, d.dxbar = dout.data_ptr(), dout.stride(0), dxbar.data_ptr()
        d.dq, d.lddq = dq.data_ptr(), dq.stride(0)
        for i, t in enumerate(dkvs):
            d.dkv[i] = t.data_ptr()
        d.dU, d.dcb, d.dxhat, d.lddx = dU.data_ptr(), dcb.data_ptr(), dxhat.data_ptr(), dxhat.stride(0)
        L.check(L.load().cmb_sva_abs_bwd(C.byref(d), L.stream_ptr(dev)), "cmb_sva_abs_bwd")
        dbk = dbv = None
        if ctx.needs_input_grad[2]:   # d b_k[c] = sum_q d(cb)[q, h(c)] q[q, c]
            dbk = k_colsum(q, row_scale=dcb, group=64).to(ctx.b_dtypes[0])
        if ctx.needs_input_grad[3]:   # d b_v[c] = sum_q m3[q, h(c)] d out[q, c]
            dbv = k_colsum(dout, row_scale=m3, group=64).to(ctx.b_dtypes[1])
        return (dq, dU, dbk, dbv, dxhat, None, None, None, None, None, None, *dkvs)


# CAMBRIAN_AMD_FUSE_CONTRACT_ADD=1: the one-key towers' output is added to the windowed tower's W_v Xb inside the per-head GEMM's
# epilogue (a batched residual) instead of by an ATen add over [Bq, 1024].  Measured neutral (profiles/r06_lab.md section 25): off.
FUSE_CONTRACT_ADD = os.environ.get("CAMBRIAN_AMD_FUSE_CONTRACT_ADD", "0") == "1"


def sva_absorbed_attention(qh, kvs_direct, masks_direct, xhat, mask_a, ra: int, wk, bk, wv, bv, B: int, qside: int,
                           window_major: bool = False) -> torch.Tensor:
    """The SVA attention core with the windowed tower's K / V projections absorbed (csrc/sva_absorbed.hip): ``qh`` [Bq, 1024]
    projected queries; ``kvs_direct`` K|V rows [Bq, 2048] of the one-key towers with their uint8 masks [Bq] (or None);
    ``xhat`` the windowed tower's normalised tokens; (wk, bk, wv, bv) its folded projection [1024, 1024] / [1024] (fp32,
    autograd-tracked).  Returns the attention output [Bq, 1024] (before o_proj), identical to ``sva_attention`` on the K|V
    rows ``xhat @ [wk; wv]^T + [bk; bv]``."""
    U = HeadExpandFn.apply(qh, wk, 16)                                           # [Bq, 16, 1024]
    out_d, xbar, _ = SvaAbsorbedFn.apply(qh, U, bk, bv, xhat, B, qside, ra, list(masks_direct), mask_a, window_major, *kvs_direct)
    if FUSE_CONTRACT_ADD:
        return HeadContractFn.apply(xbar, wv, 16, out_d)                          # out_d + W_v,h Xb, added in the GEMM's epilogue
    return out_d + HeadContractFn.apply(xbar, wv, 16)                             # + W_v,h Xb


# ================================================================================================
# autograd: token mean, embedding splice
# ================================================================================================
class TokenMeanFn(torch.autograd.Function):
    """[B,T,D] -> [B,D] mean over tokens (cambrian_arch.py:377).  If ``holder`` is given the gradient is
    accumulated into the shared fp32 buffer of the aux feature instead of being returned."""

    @staticmethod
    def forward(ctx, x, holder: Optional[GradAccumulator]):
        B, T, D = x.shape
        xc = x if x.is_contiguous() else x.contiguous()
        out = k_token_mean(xc)
        ctx.shape = (B, T, D)
        ctx.holder = holder
        ctx.x_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, D = ctx.shape
        g = _as_dtype_contig(g, ctx.x_dtype)
        if ctx.holder is not None:
            acc = ctx.holder.get(B * T, D, g.device)
        else:
            acc = torch.zeros((B * T, D), dtype=torch.float32, device=g.device)
        rc = L.load().cmb_token_mean_bwd(L.dtype_code(g.dtype), g.data_ptr(), B, T, D, acc.data_ptr(), L.stream_ptr(g.device))
        L.check(rc, "cmb_token_mean_bwd")
        if ctx.holder is not None:
            return None, None
        return k_cast(acc, ctx.x_dtype).view(B, T, D), None


def token_mean(x, holder=None):
    return TokenMeanFn.apply(x, holder)


class EmbedSpliceFn(torch.autograd.Function):
    """inputs_embeds of the static path: embedding gather + newline column + visual splice in one kernel
    (cambrian_arch.py:413-420,457-490).  Gradients flow to feat and newline, and — finetune stage — to the embedding table
    (frozen in the pre-training stage: train_fsdp.py:1677-1685)."""

    @staticmethod
    def forward(ctx, ids, table, feat, newline, side: int, image_token: int):
        B, S = ids.shape
        V, H = table.shape
        dt = feat.dtype
        tab = k_cast(table, dt)
        nl = k_cast(newline, dt)
        featc = feat if feat.is_contiguous() else feat.contiguous()
        out = torch.empty((B, S, H), dtype=dt, device=feat.device)
        pos = torch.empty((B,), dtype=torch.int32, device=feat.device)
        idc = ids if ids.is_contiguous() else ids.contiguous()
        rc = L.load().cmb_embed_splice_fwd(L.dtype_code(dt), idc.data_ptr(), B, S, H, image_token, tab.data_ptr(), V,
                                           featc.data_ptr(), side, nl.data_ptr(), out.data_ptr(), pos.data_ptr(),
                                           L.stream_ptr(feat.device))
        L.check(rc, "cmb_embed_splice_fwd")
        ctx.cfg = (B, S, H, side, dt, newline.dtype, V, table.dtype, image_token)
        ctx.save_for_backward(pos, idc)
        ctx.mark_non_differentiable(pos)
        return out, pos

    @staticmethod
    def backward(ctx, dout, _dpos):
        pos, ids = ctx.saved_tensors
        B, S, H, side, dt, nl_dtype, V, tab_dtype, image_token = ctx.cfg
        dout = _as_dtype_contig(dout, dt)
        dtable = None
        if ctx.needs_input_grad[1]:
            # finetune stage (the embedding table trains): nn.Embedding's backward for the TEXT rows — every position outside
            # the visual span [p_b, p_b + side (side + 1)) gathers table[ids'] (stock index_add_, fp32 accumulation)
            t = torch.arange(S, device=dout.device)[None]
            p_b = pos.to(torch.int64)[:, None]
            text = (p_b < 0) | (t < p_b) | (t >= p_b + side * (side + 1))
            # the same id map as embed_splice_fwd_kernel: the image token -> row 0, everything else clamped into the table
            # (a pad / ignore id such as -100 reads row 0 in the forward; it must add to row 0 here, not assert)
            idx = torch.where(ids == image_token, torch.zeros_like(ids), ids).clamp_(0, V - 1)[text]
            dtable = torch.zeros((V, H), dtype=torch.float32, device=dout.device)
            dtable.index_add_(0, idx, dout[text].float())
            dtable = dtable.to(tab_dtype)
        dfeat = torch.empty((B, side * side, H), dtype=dt, device=dout.device)
        dnl = torch.zeros((H,), dtype=torch.float32, device=dout.device)
        rc = L.load().cmb_embed_splice_bwd(L.dtype_code(dt), dout.data_ptr(), pos.data_ptr(), B, S, H, side,
                                           dfeat.data_ptr(), dnl.data_ptr(), L.stream_ptr(dout.device))
        L.check(rc, "cmb_embed_splice_bwd")
        return None, dtable, dfeat, dnl.to(nl_dtype), None, None


def embed_splice(ids, table, feat, newline, side: int, image_token: int = -200):
    return EmbedSpliceFn.apply(ids, table, feat, newline, side, image_token)


# ================================================================================================
# autograd: RMSNorm / RoPE (LLM side)
# ================================================================================================
class RmsNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps: float):
        shape = x.shape
        D = shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w32 = k_cast(weight, torch.float32)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rc = L.load().cmb_rmsnorm_fwd(L.dtype_code(x.dtype), x2.data_ptr(), rows, D, w32.data_ptr(), eps, y.data_ptr(),
                                      rstd.data_ptr(), L.stream_ptr(x.device))
        L.check(rc, "cmb_rmsnorm_fwd")
        ctx.save_for_backward(x2, w32, rstd)
        ctx.shape, ctx.w_dtype = shape, weight.dtype
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, rstd = ctx.saved_tensors
        rows, D = x2.shape
        dy2 = _as_dtype_contig(dy.reshape(rows, D), x2.dtype)
        dx = torch.empty_like(x2)
        dw = torch.zeros((D,), dtype=torch.float32, device=x2.device) if ctx.needs_input_grad[1] else None
        if dw is None:  # frozen weight: single-pass kernel (row kept in registers)
            rc = L.load().cmb_rmsnorm_bwd_add(L.dtype_code(x2.dtype), dy2.data_ptr(), x2.data_ptr(), None, rows, D,
                                              w32.data_ptr(), rstd.data_ptr(), dx.data_ptr(), L.stream_ptr(x2.device))
            L.check(rc, "cmb_rmsnorm_bwd_add")
        else:
            rc = L.load().cmb_rmsnorm_bwd(L.dtype_code(x2.dtype), dy2.data_ptr(), x2.data_ptr(), rows, D, w32.data_ptr(),
                                          rstd.data_ptr(), dx.data_ptr(), L.ptr(dw), L.stream_ptr(x2.device))
            L.check(rc, "cmb_rmsnorm_bwd")
        if dw is not None and ctx.w_dtype != torch.float32:
            dw = dw.to(ctx.w_dtype)
        return dx.view(ctx.shape), dw, None


def rmsnorm(x, weight, eps: float = 1e-6):
    return RmsNormFn.apply(x, weight, eps)


class RmsNormForkFn(torch.autograd.Function):
    """(x, rmsnorm(x) * w): the decoder layer's "normalise for the attention branch AND keep x for the skip connection" as
    ONE autograd node.  With two consumers of x (RmsNormFn and the later residual add) autograd sums their two dense
    [tokens, hidden] gradients with an extra elementwise pass per layer (33 x 131 us per step at 16 x 2048 tokens); here
    the skip path's gradient enters the RMSNorm backward kernel as its additive term (cmb_rmsnorm_bwd_add): one pass."""

    @staticmethod
    def forward(ctx, x, weight, eps: float):
        shape = x.shape
        D = shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w32 = k_cast(weight, torch.float32)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rc = L.load().cmb_rmsnorm_fwd(L.dtype_code(x.dtype), x2.data_ptr(), rows, D, w32.data_ptr(), eps, y.data_ptr(),
                                      rstd.data_ptr(), L.stream_ptr(x.device))
        L.check(rc, "cmb_rmsnorm_fwd")
        ctx.save_for_backward(x2, w32, rstd)
        ctx.shape, ctx.w_dtype = shape, weight.dtype
        return x, y.view(shape)          # x returned as is: autograd hands out an alias of it

    @staticmethod
    def backward(ctx, g_x, g_y):
        x2, w32, rstd = ctx.saved_tensors
        rows, D = x2.shape
        if g_y is None:
            return g_x, None, None
        gy = _as_dtype_contig(g_y.reshape(rows, D), x2.dtype)
        gs = None if g_x is None else _as_dtype_contig(g_x.reshape(rows, D), x2.dtype)
        dx = torch.empty_like(x2)
        dw = None
        if ctx.needs_input_grad[1]:  # trainable norm weight (finetune stage): the two-pass kernel that also reduces dw
            dw = torch.zeros((D,), dtype=torch.float32, device=x2.device)
            rc = L.load().cmb_rmsnorm_bwd(L.dtype_code(x2.dtype), gy.data_ptr(), x2.data_ptr(), rows, D, w32.data_ptr(),
                                          rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), L.stream_ptr(x2.device))
            L.check(rc, "cmb_rmsnorm_bwd")
            if gs is not None:
                dx = dx + gs
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        else:
            rc = L.load().cmb_rmsnorm_bwd_add(L.dtype_code(x2.dtype), gy.data_ptr(), x2.data_ptr(), L.ptr(gs), rows, D,
                                              w32.data_ptr(), rstd.data_ptr(), dx.data_ptr(), L.stream_ptr(x2.device))
            L.check(rc, "cmb_rmsnorm_bwd_add")
        out = dx.view(ctx.shape)
        # a buffer this call allocated and hands to exactly one receiver: whoever gets THIS object as its incoming gradient may
        # write into it (ScatterQueryRowsFn.backward: no 400 MB clone per in-LLM SVA layer).  Autograd's own sums of several
        # gradients are new tensors without the mark.
        out._cmb_exclusive = True
        return out, dw, None


def rmsnorm_fork(x, weight, eps: float = 1e-6):
    """(x, rmsnorm(x) * weight) — see RmsNormForkFn.  Use the RETURNED x for the skip connection."""
    return RmsNormForkFn.apply(x, weight, eps)


def rope_table(position_ids: torch.Tensor, head_dim: int, base: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [ntok, head_dim/2] fp32 for position_ids [B,S] (built once per forward)."""
    L.require_gpu(position_ids)
    pid = position_ids.reshape(-1).to(torch.int64).contiguous()
    ntok = pid.numel()
    cos = torch.empty((ntok, head_dim // 2), dtype=torch.float32, device=pid.device)
    sin = torch.empty_like(cos)
    rc = L.load().cmb_rope_table(pid.data_ptr(), ntok, head_dim, base, cos.data_ptr(), sin.data_ptr(),
                                 L.stream_ptr(pid.device))
    L.check(rc, "cmb_rope_table")
    return cos, sin


class RopeFn(torch.autograd.Function):
    """x [ntok, H, Dh] (contiguous) rotated out of place; the backward is the inverse rotation."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        y = x.contiguous().clone()
        ntok, H, Dh = y.shape
        rc = L.load().cmb_rope_apply(L.dtype_code(y.dtype), y.data_ptr(), cos.data_ptr(), sin.data_ptr(), ntok, H, Dh,
                                     H * Dh, 0, L.stream_ptr(y.device))
        L.check(rc, "cmb_rope_apply")
        ctx.save_for_backward(cos, sin)
        return y

    @staticmethod
    def backward(ctx, dy):
        cos, sin = ctx.saved_tensors
        g = dy.contiguous().clone()
        ntok, H, Dh = g.shape
        rc = L.load().cmb_rope_apply(L.dtype_code(g.dtype), g.data_ptr(), cos.data_ptr(), sin.data_ptr(), ntok, H, Dh,
                                     H * Dh, 1, L.stream_ptr(g.device))
        L.check(rc, "cmb_rope_apply(inverse)")
        return g, None, None


def rope(x, cos, sin):
    return RopeFn.apply(x, cos, sin)


# ================================================================================================
# autograd: row gather / scatter of the in-LLM SVA hook (cambrian_llama.py:181-207)
# ================================================================================================
def k_copy_rows(src: Optional[torch.Tensor], src_map: Optional[RowMap], dst: torch.Tensor, dst_map: RowMap, rows: int,
                D: int) -> None:
    L.require_gpu(src, dst)
    rc = L.load().cmb_copy_rows(L.dtype_code(dst.dtype), L.ptr(src), None if src_map is None else C.byref(src_map),
                                dst.data_ptr(), C.byref(dst_map), rows, D, L.stream_ptr(dst.device))
    L.check(rc, "cmb_copy_rows")


def hook_row_map(S: int, H: int, side: int) -> RowMap:
    """query row (b, i, j) -> element offset of hidden[b, i*(side+1)+j, :] relative to hidden[0, image_position]."""
    return L.make_map(side * side, side, S * H, (side + 1) * H, H)


class GatherQueryRowsFn(torch.autograd.Function):
    """hidden [B,S,H] -> the side*side latent-query rows [B*side*side, H] starting at ``pos`` (newline column
    skipped): ``hidden[:, pos:pos+side*(side+1)].view(B,side,side+1,H)[:, :, :side]`` as one strided copy.

    ``link`` (a dict shared with the ScatterQueryRowsFn of the same hook, or None): the hook reads these rows of
    ``hidden`` and then overwrites exactly them, so d(hidden) = [text / newline rows: the scatter's incoming gradient;
    query rows: this gather's].  Returned as two dense tensors autograd would zero-fill one, clone the other and add
    them (4 extra passes over [B,S,H] per hook).  With a link the scatter's backward — which always runs first: this
    node's incoming gradient is computed from the scatter's d(rows) — HANDS its dense result over in ``link["dh"]`` and
    returns no gradient for ``hidden`` itself; this backward writes its rows into that buffer and returns it as THE
    gradient of ``hidden``.  Autograd therefore sees one ordinary contribution (no assumption about which buffer it
    keeps, views / CopySlices and further consumers of ``hidden`` included: ADVICE r3)."""

    @staticmethod
    def forward(ctx, hidden, pos: int, side: int, link=None):
        B, S, H = hidden.shape
        assert hidden.is_contiguous()
        out = torch.empty((B * side * side, H), dtype=hidden.dtype, device=hidden.device)
        k_copy_rows(hidden.view(-1)[pos * H:], hook_row_map(S, H, side), out, L.identity_map(H), B * side * side, H)
        ctx.cfg = (B, S, H, pos, side)
        ctx.link = link
        if link is not None:
            link["armed"] = bool(ctx.needs_input_grad[0])  # this node's backward will run iff the scatter's does
        return out

    @staticmethod
    def backward(ctx, g):
        B, S, H, pos, side = ctx.cfg
        g = g.contiguous()
        dh = ctx.link.pop("dh", None) if ctx.link is not None else None
        if dh is None:
            dh = torch.zeros((B, S, H), dtype=g.dtype, device=g.device)
        elif dh.dtype != g.dtype:
            g = g.to(dh.dtype)
        k_copy_rows(g, L.identity_map(H), dh.view(-1)[pos * H:], hook_row_map(S, H, side), B * side * side, H)
        return dh, None, None, None


class ScatterQueryRowsFn(torch.autograd.Function):
    """In-place write-back of the updated latent queries into ``hidden`` (the reference's
    ``hidden_states[:, a:b] = latent_query_with_newline``, cambrian_llama.py:207); the newline column and all
    text rows are untouched.  ``link``: see GatherQueryRowsFn."""

    @staticmethod
    def forward(ctx, hidden, rows, pos: int, side: int, link=None):
        B, S, H = hidden.shape
        assert hidden.is_contiguous() and rows.is_contiguous()
        k_copy_rows(rows, L.identity_map(H), hidden.view(-1)[pos * H:], hook_row_map(S, H, side), B * side * side, H)
        ctx.mark_dirty(hidden)
        ctx.cfg = (B, S, H, pos, side)
        ctx.link = link
        return hidden

    @staticmethod
    def backward(ctx, g):
        B, S, H, pos, side = ctx.cfg
        g = g.contiguous()
        drows = torch.empty((B * side * side, H), dtype=g.dtype, device=g.device)
        k_copy_rows(g.view(-1)[pos * H:], hook_row_map(S, H, side), drows, L.identity_map(H), B * side * side, H)
        # d(hidden) = g with the overwritten rows zeroed: in place when g is a buffer its producer marked as handed to this node
        # alone (RmsNormForkFn.backward: the decoder layer that follows), else on a copy (round 6: the copy was 400 MB x 10 layers)
        dh = g if getattr(g, "_cmb_exclusive", False) else g.clone()
        k_copy_rows(None, None, dh.view(-1)[pos * H:], hook_row_map(S, H, side), B * side * side, H)  # zero the rows
        if ctx.link is not None and ctx.link.get("armed") and ctx.needs_input_grad[0]:
            ctx.link["dh"] = dh  # the paired gather's backward completes it and returns it for `hidden`
            return None, drows, None, None, None
        return dh, drows, None, None, None


def gather_query_rows(hidden, pos: int, side: int, link=None):
    return GatherQueryRowsFn.apply(hidden, pos, side, link)


def scatter_query_rows(hidden, rows, pos: int, side: int, link=None):
    return ScatterQueryRowsFn.apply(hidden, rows, pos, side, link)


# ================================================================================================
# autograd: fused shifted cross-entropy and SwiGLU (LLM side; cambrian_llama.py:402-422, Llama MLP)
# ================================================================================================
class CrossEntropyFn(torch.autograd.Function):
    """mean_{labels != ignore} ( logsumexp(logits[t]) - logits[t, labels[t]] ) over [T, V] logits kept in their
    compute dtype (the reference's ``logits.float()`` + ``CrossEntropyLoss``, cambrian_llama.py:409-422: fp32 math on
    the same bf16-rounded values).  The backward overwrites ``logits`` with dlogits when ``inplace`` (the tensor that
    lm_head produced is not needed by anybody else's backward)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index: int, inplace: bool):
        L.require_gpu(logits, labels)
        T, V = logits.shape
        if logits.stride(1) != 1:
            raise L.CambrianAmdError("logits must be row-major")
        labels = labels.to(torch.int64).contiguous()
        lse = torch.empty((T,), dtype=torch.float32, device=logits.device)
        loss_rows = torch.empty((T,), dtype=torch.float32, device=logits.device)
        rc = L.load().cmb_cross_entropy_fwd(L.dtype_code(logits.dtype), logits.data_ptr(), T, V, logits.stride(0),
                                            labels.data_ptr(), ignore_index, lse.data_ptr(), loss_rows.data_ptr(),
                                            L.stream_ptr(logits.device))
        L.check(rc, "cmb_cross_entropy_fwd")
        n_valid = (labels != ignore_index).sum().to(torch.float32)
        ctx.save_for_backward(logits, labels, lse, n_valid)
        ctx.ignore_index, ctx.inplace = ignore_index, inplace
        return loss_rows.sum() / n_valid

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, n_valid = ctx.saved_tensors
        T, V = logits.shape
        scale = (g.to(torch.float32) / n_valid).reshape(1).contiguous()
        out = logits if ctx.inplace else torch.empty_like(logits)
        rc = L.load().cmb_cross_entropy_bwd(L.dtype_code(logits.dtype), logits.data_ptr(), T, V, logits.stride(0),
                                            labels.data_ptr(), ctx.ignore_index, lse.data_ptr(), scale.data_ptr(),
                                            out.data_ptr(), out.stride(0), L.stream_ptr(logits.device))
        L.check(rc, "cmb_cross_entropy_bwd")
        return out, None, None, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100, inplace: bool = False):
    return CrossEntropyFn.apply(logits, labels, ignore_index, inplace)


class SwiGLUFn(torch.autograd.Function):
    """h = silu(g) * u in one pass; backward (dg, du) in one pass (Llama MLP: down_proj(silu(gate) * up))."""

    @staticmethod
    def forward(ctx, g, u):
        L.require_gpu(g, u)
        shape = g.shape
        D = shape[-1]
        g2, u2 = g.reshape(-1, D), u.reshape(-1, D)
        if g2.stride(1) != 1 or u2.stride(1) != 1:
            g2, u2 = g2.contiguous(), u2.contiguous()
        rows = g2.shape[0]
        h = torch.empty((rows, D), dtype=g.dtype, device=g.device)
        rc = L.load().cmb_act_mul(L.dtype_code(g.dtype), L.ACT_SILU, g2.data_ptr(), g2.stride(0), u2.data_ptr(), u2.stride(0),
                                  rows, D, h.data_ptr(), D, L.stream_ptr(g.device))
        L.check(rc, "cmb_act_mul(silu)")
        ctx.save_for_backward(g2, u2)
        ctx.shape = shape
        return h.view(shape)

    @staticmethod
    def backward(ctx, dh):
        g2, u2 = ctx.saved_tensors
        rows, D = g2.shape
        dh2 = dh.reshape(rows, D)
        if dh2.stride(1) != 1:
            dh2 = dh2.contiguous()
        dg = torch.empty((rows, D), dtype=g2.dtype, device=g2.device)
        du = torch.empty((rows, D), dtype=g2.dtype, device=g2.device)
        rc = L.load().cmb_swiglu_bwd(L.dtype_code(g2.dtype), dh2.data_ptr(), dh2.stride(0), g2.data_ptr(), g2.stride(0),
                                     u2.data_ptr(), u2.stride(0), rows, D, dg.data_ptr(), D, du.data_ptr(), D,
                                     L.stream_ptr(g2.device))
        L.check(rc, "cmb_swiglu_bwd")
        return dg.view(ctx.shape), du.view(ctx.shape)


def swiglu(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    return SwiGLUFn.apply(g, u)


class QkvRopeFn(torch.autograd.Function):
    """packed QKV projection [B,S,(nh+2nkv)*Dh] -> q [B,nh,S,Dh], k, v [B,nkv,S,Dh] (transposed VIEWS of token-major
    buffers) with RoPE on q and k, one kernel;
    the backward merges (dq, dk, dv) into one d(packed) with the inverse rotation (no slice-backward zero fills)."""

    @staticmethod
    def forward(ctx, packed, cos, sin, nh: int, nkv: int, hd: int):
        L.require_gpu(packed, cos, sin)
        B, S, W = packed.shape
        if W != (nh + 2 * nkv) * hd or not packed.is_contiguous():
            raise L.CambrianAmdError("packed QKV must be contiguous [B,S,(nh+2*nkv)*hd]")
        dt, dev = packed.dtype, packed.device
        q = torch.empty((B, S, nh, hd), dtype=dt, device=dev)
        k = torch.empty((B, S, nkv, hd), dtype=dt, device=dev)
        v = torch.empty((B, S, nkv, hd), dtype=dt, device=dev)
        rc = L.load().cmb_qkv_rope(L.dtype_code(dt), 0, packed.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, S, nh, nkv, hd,
                                   q.data_ptr(), k.data_ptr(), v.data_ptr(), L.stream_ptr(dev))
        L.check(rc, "cmb_qkv_rope(split)")
        ctx.save_for_backward(cos, sin)
        ctx.cfg = (B, S, nh, nkv, hd)
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)  # [B,H,S,hd] views of token-major storage

    @staticmethod
    def backward(ctx, dq, dk, dv):
        cos, sin = ctx.saved_tensors
        B, S, nh, nkv, hd = ctx.cfg
        # token-major storage; free when the attention backward returns grads laid out like its inputs
        dq, dk, dv = (t.transpose(1, 2).contiguous() for t in (dq, dk, dv))
        dp = torch.empty((B, S, (nh + 2 * nkv) * hd), dtype=dq.dtype, device=dq.device)
        rc = L.load().cmb_qkv_rope(L.dtype_code(dq.dtype), 1, dp.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, S, nh, nkv, hd,
                                   dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), L.stream_ptr(dq.device))
        L.check(rc, "cmb_qkv_rope(merge)")
        return dp, None, None, None, None, None


def qkv_rope(packed, cos, sin, nh: int, nkv: int, hd: int):
    return QkvRopeFn.apply(packed, cos, sin, nh, nkv, hd)


class SwiGLUPackedFn(torch.autograd.Function):
    """h = silu(gu[:, :I]) * gu[:, I:] for the packed output of a fused gate|up projection [T, 2I]; the backward
    writes both halves of d(gu) straight into one [T, 2I] buffer."""

    @staticmethod
    def forward(ctx, gu):
        L.require_gpu(gu)
        shape = gu.shape
        I2 = shape[-1]
        I = I2 // 2
        g2 = gu.reshape(-1, I2)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        rows = g2.shape[0]
        h = torch.empty((rows, I), dtype=gu.dtype, device=gu.device)
        es = g2.element_size()
        rc = L.load().cmb_act_mul(L.dtype_code(gu.dtype), L.ACT_SILU, g2.data_ptr(), I2, g2.data_ptr() + I * es, I2,
                                  rows, I, h.data_ptr(), I, L.stream_ptr(gu.device))
        L.check(rc, "cmb_act_mul(silu, packed)")
        ctx.save_for_backward(g2)
        ctx.shape = shape
        return h.view(*shape[:-1], I)

    @staticmethod
    def backward(ctx, dh):
        (g2,) = ctx.saved_tensors
        rows, I2 = g2.shape
        I = I2 // 2
        dh2 = dh.reshape(rows, I)
        if dh2.stride(1) != 1:
            dh2 = dh2.contiguous()
        dgu = torch.empty((rows, I2), dtype=g2.dtype, device=g2.device)
        es = g2.element_size()
        rc = L.load().cmb_swiglu_bwd(L.dtype_code(g2.dtype), dh2.data_ptr(), dh2.stride(0), g2.data_ptr(), I2,
                                     g2.data_ptr() + I * es, I2, rows, I, dgu.data_ptr(), I2, dgu.data_ptr() + I * es, I2,
                                     L.stream_ptr(g2.device))
        L.check(rc, "cmb_swiglu_bwd(packed)")
        return dgu.view(ctx.shape)


def swiglu_packed(gu: torch.Tensor) -> torch.Tensor:
    return SwiGLUPackedFn.apply(gu)


# ================================================================================================
# autograd: causal self-attention of the decoder — stock forward, HIP backward
# ================================================================================================
def _token_major(t: torch.Tensor) -> torch.Tensor:
    """[B,H,S,D] tensor whose storage is [B,S,H,D]-contiguous (a transposed view); copies only if it is not."""
    tm = t.transpose(1, 2)
    return t if tm.is_contiguous() else tm.contiguous().transpose(1, 2)


class CausalAttnFn(torch.autograd.Function):
    """GQA attention, head_dim 128, on flash_bwd.hip: forward kernel (online softmax, writes the log-sum-exp),
    backward = dQ kernel + dK/dV kernel (no atomics).  K/V are read un-expanded (grouped heads): no
    repeat_interleave copies of K and V per layer as F.scaled_dot_product_attention(enable_gqa=True) makes.
    ``kv_len`` None: causal (the decoder).  ``kv_len`` = n: bidirectional over keys [0, n), rows [n, S) are padding
    (trainable vision towers, see ``vit_attention``).  ``key_valid`` (causal only): uint8 / bool [B, S] key-padding mask of
    the collator; query q sees key k iff k <= q and (key_valid[b, k] or k == q)."""

    @staticmethod
    def forward(ctx, q, k, v, scale=None, kv_len=None, key_valid=None):
        L.require_gpu(q, k, v, key_valid)
        B, H, S, D = q.shape
        HKV = k.shape[1]
        scale = 1.0 / math.sqrt(D) if scale is None else float(scale)
        causal, n = (1, S) if kv_len is None else (0, int(kv_len))
        q, k, v = _token_major(q), _token_major(k), _token_major(v)
        out = torch.empty((B, S, H, D), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        if key_valid is not None:
            if not causal or key_valid.shape != (B, S):
                raise L.CambrianAmdError("key_valid is a [B, S] mask of the causal form")
            key_valid = key_valid.view(torch.uint8) if key_valid.dtype == torch.bool else key_valid.to(torch.uint8)
            key_valid = key_valid.contiguous()
        rc = L.load().cmb_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), B, S, H, HKV, D, S * H * D, H * D, D,
                                         S * HKV * D, HKV * D, D, scale, causal, n, L.ptr(key_valid), out.data_ptr(),
                                         lse.data_ptr(), L.stream_ptr(q.device))
        L.check(rc, "cmb_flash_attn_fwd")
        out = out.transpose(1, 2)  # [B,H,S,D] view of token-major storage: the o_proj input needs no copy
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.key_valid = key_valid
        ctx.cfg = (scale, causal, n)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        scale, causal, n = ctx.cfg
        B, H, S, D = q.shape
        HKV = k.shape[1]
        q, out, dout = _token_major(q), _token_major(out), _token_major(dout)
        k, v = _token_major(k), _token_major(v)
        dq = torch.empty((B, S, H, D), dtype=q.dtype, device=q.device)
        dk = torch.empty((B, S, HKV, D), dtype=q.dtype, device=q.device)
        dv = torch.empty((B, S, HKV, D), dtype=q.dtype, device=q.device)
        dvec = torch.empty((2, B, H, S), dtype=torch.float32, device=q.device)   # [0] D, [1] lse * log2 e (flash2.hip)
        lse = lse.contiguous()
        rc = L.load().cmb_flash_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                         lse.data_ptr(), B, S, H, HKV, D, S * H * D, H * D, D, S * HKV * D, HKV * D, D,
                                         scale, causal, n, L.ptr(ctx.key_valid), dvec.data_ptr(), dq.data_ptr(),
                                         dk.data_ptr(), dv.data_ptr(), L.stream_ptr(q.device))
        L.check(rc, "cmb_flash_attn_bwd")
        return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2), None, None, None


def vit_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """Bidirectional self-attention WITH a backward, for towers that train: q, k, v [B,H,N,hd] bf16
    with any N and hd <= 128.  Runs on the decoder's flash kernels: tokens zero-padded to a multiple of 128 and masked
    through ``kv_len``, head_dim zero-padded to 128 (zero columns change neither q.k nor the first hd output columns);
    the padding and the final slice are ordinary autograd ops, so padded rows receive zero gradients."""
    B, H, N, hd = q.shape
    if hd > 128 or q.dtype != torch.bfloat16:
        raise L.CambrianAmdError("vit_attention: bf16 and head_dim <= 128 only")
    S = (N + 127) // 128 * 128
    pad = (0, 128 - hd, 0, S - N)
    qp, kp, vp = (F.pad(t, pad) for t in (q, k, v))
    return CausalAttnFn.apply(qp, kp, vp, scale, N)[:, :, :N, :hd]


def causal_attention_supported(q: torch.Tensor, k: torch.Tensor) -> bool:
    return (q.is_cuda and q.dtype == torch.bfloat16 and q.shape[-1] == 128 and q.shape[2] % 128 == 0
            and q.shape[2] == k.shape[2] and q.shape[1] % k.shape[1] == 0)


def causal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B,H,S,128], k / v [B,HKV,S,128] (any strides with a contiguous last dim) -> [B,H,S,128].  ``key_valid``: optional
    bool / uint8 [B,S] key-padding mask (the collator's ``attention_mask``); the diagonal stays open."""
    return CausalAttnFn.apply(q, k, v, None, None, key_valid)


class AddRmsNormFn(torch.autograd.Function):
    """(s, y) = (x + delta, rmsnorm(x + delta) * w) in one pass — the decoder layer's "h = h + attn; mlp_in = norm(h)".
    Backward: d(x) = d(delta) = g_s + rmsnorm_backward(g_y), also one pass when the weight is frozen."""

    @staticmethod
    def forward(ctx, x, delta, weight, eps: float):
        L.require_gpu(x, delta, weight)
        shape = x.shape
        D = shape[-1]
        x2, d2 = x.reshape(-1, D), delta.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        d2 = d2 if d2.is_contiguous() else d2.contiguous()
        if d2.dtype != x2.dtype:
            d2 = d2.to(x2.dtype)
        w32 = k_cast(weight, torch.float32)
        rows = x2.shape[0]
        s = torch.empty_like(x2)
        y = torch.empty_like(x2)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rc = L.load().cmb_add_rmsnorm_fwd(L.dtype_code(x2.dtype), x2.data_ptr(), d2.data_ptr(), rows, D, w32.data_ptr(), eps,
                                          s.data_ptr(), y.data_ptr(), rstd.data_ptr(), L.stream_ptr(x.device))
        L.check(rc, "cmb_add_rmsnorm_fwd")
        ctx.save_for_backward(s, w32, rstd)
        ctx.shape, ctx.w_dtype = shape, weight.dtype
        return s.view(shape), y.view(shape)

    @staticmethod
    def backward(ctx, g_s, g_y):
        s, w32, rstd = ctx.saved_tensors
        rows, D = s.shape
        need_w = ctx.needs_input_grad[2]
        if g_y is None:
            return g_s, g_s, None, None
        gy = _as_dtype_contig(g_y.reshape(rows, D), s.dtype)
        gs = None if g_s is None else _as_dtype_contig(g_s.reshape(rows, D), s.dtype)
        dw = None
        dx = torch.empty_like(s)
        if need_w:  # trainable norm weight (finetune stage): the two-pass kernel that also reduces dw
            dw = torch.zeros((D,), dtype=torch.float32, device=s.device)
            rc = L.load().cmb_rmsnorm_bwd(L.dtype_code(s.dtype), gy.data_ptr(), s.data_ptr(), rows, D, w32.data_ptr(),
                                          rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), L.stream_ptr(s.device))
            L.check(rc, "cmb_rmsnorm_bwd")
            if gs is not None:
                dx = dx + gs
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        else:
            rc = L.load().cmb_rmsnorm_bwd_add(L.dtype_code(s.dtype), gy.data_ptr(), s.data_ptr(), L.ptr(gs), rows, D,
                                              w32.data_ptr(), rstd.data_ptr(), dx.data_ptr(), L.stream_ptr(s.device))
            L.check(rc, "cmb_rmsnorm_bwd_add")
        dx = dx.view(ctx.shape)
        return dx, dx, dw, None


def add_rmsnorm(x, delta, weight, eps: float = 1e-6):
    return AddRmsNormFn.apply(x, delta, weight, eps)
