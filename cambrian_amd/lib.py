"""ctypes binding of ``libcambrian_amd.so`` (the C-ABI declared in ``include/cambrian_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C cambrian_amd/csrc``.  There is
deliberately NO fallback: if the shared object is missing, or a kernel returns a non-zero status, this
module raises — a silent eager/PyTorch path would void every parity claim.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAMBRIAN_AMD_LIB: another build of the same C-ABI (same-box A/B runs of two kernel versions); default = the in-tree build
LIB_PATH = os.environ.get("CAMBRIAN_AMD_LIB") or os.path.join(_HERE, "csrc", "libcambrian_amd.so")

BF16, F32 = 0, 1
F16 = 2   # output type of cmb_image_preprocess only
FP8_E4M3 = 3   # operand type of cmb_gemm (OCP e4m3fn bytes from cmb_quantize_fp8_rows)
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU, ACT_SILU, ACT_SWIGLU_PAIRS = 0, 1, 2, 3, 4, 5
ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "gelu": ACT_GELU_ERF, "gelu_erf": ACT_GELU_ERF,
             "gelu_tanh": ACT_GELU_TANH, "gelu_pytorch_tanh": ACT_GELU_TANH,
             "quick_gelu": ACT_QUICK_GELU, "silu": ACT_SILU}
SVA_MAX_TOWERS = 8
KNOB_LN_FWD, KNOB_DWCONV, KNOB_VIT_ATTN, KNOB_SVA_ABS, KNOB_LN_MULTI_CHUNK, KNOB_FLASH, KNOB_COLSUM_WGS, KNOB_LN_BWD_ROWS = 0, 1, 2, 3, 4, 5, 6, 7   # enum cmb_knob_id
ABI_VERSION = 10   # CMB_ABI_VERSION of the include/cambrian_amd.h this binding was written against

STATUS = {0: "CMB_OK", -1: "CMB_ERR_BAD_ARG", -2: "CMB_ERR_ALIGNMENT", -3: "CMB_ERR_SHAPE",
          -4: "CMB_ERR_WORKSPACE", -5: "CMB_ERR_LAUNCH"}


class CambrianAmdError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("n1", C.c_int64), ("n2", C.c_int64), ("s0", C.c_int64), ("s1", C.c_int64), ("s2", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("A", C.c_void_p), ("a_map", RowMap),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("c_map", RowMap),
        ("bias", C.c_void_p), ("colscale", C.c_void_p),
        ("residual", C.c_void_p), ("r_map", RowMap),
        ("pre_out", C.c_void_p), ("p_map", RowMap),
        ("act", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float),
        ("split_k", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("tile_hint", C.c_int32),
        ("a_scale", C.c_void_p), ("b_scale", C.c_void_p),
        ("batch", C.c_int32), ("a_batch_stride", C.c_int64), ("b_batch_stride", C.c_int64), ("c_batch_stride", C.c_int64),
        ("row_mean", C.c_void_p), ("row_rstd", C.c_void_p), ("col_sum", C.c_void_p),
    ]


class SvaDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("B", C.c_int32), ("qside", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32), ("ntowers", C.c_int32),
        ("window_major", C.c_int32),
        ("r", C.c_int32 * SVA_MAX_TOWERS),
        ("q", C.c_void_p), ("ldq", C.c_int64),
        ("kv", C.c_void_p * SVA_MAX_TOWERS), ("ldkv", C.c_int64 * SVA_MAX_TOWERS),
        ("mask", C.c_void_p * SVA_MAX_TOWERS),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("lse", C.c_void_p),
        ("dout", C.c_void_p), ("lddo", C.c_int64),
        ("dq", C.c_void_p), ("lddq", C.c_int64),
        ("dkv", C.c_void_p * SVA_MAX_TOWERS),
    ]


class SvaAbsDesc(C.Structure):
    """cmb_sva_abs_desc (include/cambrian_amd.h): SVA attention with one windowed tower's K / V projections absorbed."""
    _fields_ = [
        ("B", C.c_int32), ("qside", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32),
        ("ntowers", C.c_int32), ("window_major", C.c_int32),
        ("r", C.c_int32 * SVA_MAX_TOWERS),
        ("q", C.c_void_p), ("ldq", C.c_int64),
        ("kv", C.c_void_p * SVA_MAX_TOWERS), ("ldkv", C.c_int64 * SVA_MAX_TOWERS),
        ("mask", C.c_void_p * SVA_MAX_TOWERS),
        ("ra", C.c_int32), ("dtype", C.c_int32),
        ("xhat", C.c_void_p), ("ldx", C.c_int64),
        ("mask_a", C.c_void_p),
        ("U", C.c_void_p), ("bk", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("xbar", C.c_void_p), ("m3", C.c_void_p), ("P", C.c_void_p),
        ("dout", C.c_void_p), ("lddo", C.c_int64),
        ("dxbar", C.c_void_p), ("bv", C.c_void_p),
        ("dq", C.c_void_p), ("lddq", C.c_int64),
        ("dkv", C.c_void_p * SVA_MAX_TOWERS),
        ("dU", C.c_void_p), ("dcb", C.c_void_p),
        ("dxhat", C.c_void_p), ("lddx", C.c_int64),
    ]


LN_MULTI_MAX = 16


class LnMultiDesc(C.Structure):
    """cmb_ln_multi_desc (include/cambrian_amd.h): backward of several LayerNorms of one input in one pass."""
    _fields_ = [
        ("dtype", C.c_int32), ("layers", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("rows", C.c_int64), ("D", C.c_int64),
        ("side", C.c_int32), ("grid_r", C.c_int32),
        ("dy", C.c_void_p * LN_MULTI_MAX), ("add", C.c_void_p * LN_MULTI_MAX), ("mean", C.c_void_p * LN_MULTI_MAX),
        ("rstd", C.c_void_p * LN_MULTI_MAX), ("dadd", C.c_void_p * LN_MULTI_MAX),
        ("dx", C.c_void_p), ("lddx", C.c_int64),
        ("accumulate", C.c_int32), ("reserved", C.c_int32),
        ("dx_out", C.c_void_p),
    ]


class LnFwdMultiDesc(C.Structure):
    """cmb_ln_fwd_multi_desc (include/cambrian_amd.h): forward of several non-affine LayerNorms of one input in one pass."""
    _fields_ = [
        ("dtype", C.c_int32), ("layers", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("rows", C.c_int64), ("D", C.c_int64),
        ("side", C.c_int32), ("grid_r", C.c_int32),
        ("eps", C.c_float), ("reserved", C.c_int32),
        ("add", C.c_void_p * LN_MULTI_MAX), ("y", C.c_void_p * LN_MULTI_MAX), ("mean", C.c_void_p * LN_MULTI_MAX),
        ("rstd", C.c_void_p * LN_MULTI_MAX),
    ]


class ImageJob(C.Structure):
    """cmb_image_job (include/cambrian_amd.h): one (sample, tower) unit of the image pre-processing launch."""
    _fields_ = [
        ("src_off", C.c_int64), ("tmp_off", C.c_int64), ("dst_off", C.c_int64),
        ("w", C.c_int32), ("h", C.c_int32), ("side", C.c_int32), ("off_x", C.c_int32), ("off_y", C.c_int32),
        ("out_side", C.c_int32), ("ksize", C.c_int32), ("coef_off", C.c_int32), ("bounds_off", C.c_int32),
        ("lut_off", C.c_int32), ("background", C.c_uint32), ("reserved", C.c_int32),
    ]


class PrepJob(C.Structure):
    """cmb_prep_job (include/cambrian_amd.h): one weight of a cmb_weight_prep launch."""
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("ld_src", C.c_int64),
        ("src_dtype", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("rows_pad", C.c_int32),
        ("tile0", C.c_int32), ("reserved", C.c_int32),
    ]


# symbol -> (restype, argtypes); every symbol of include/cambrian_amd.h must be listed here
# (tests/test_abi.py cross-checks this table against the header).
_i32, _i64, _f, _p = C.c_int32, C.c_int64, C.c_float, C.c_void_p
SIGNATURES = {
    "cmb_version": (C.c_char_p, []),
    "cmb_abi_version": (C.c_int, []),
    "cmb_knob_set": (C.c_int, [_i32, _i32]),
    "cmb_knob_get": (C.c_int, [_i32]),
    "cmb_gemm": (C.c_int, [C.POINTER(GemmDesc), _p]),
    "cmb_gemm_tn": (C.c_int, [C.POINTER(GemmDesc), _p]),
    "cmb_gemm_pair": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), _p]),
    "cmb_gemm_pair_last": (C.c_int, []),
    "cmb_gemm_tile": (C.c_int, [C.c_int, _i64, _i64, _i32, _i32]),
    "cmb_gemm_last_kernel": (C.c_int, []),
    "cmb_gemm_policy_set": (C.c_int, [_i64, _i64, _i64, _i32, _i32]),
    "cmb_gemm_policy_clear": (C.c_int, []),
    "cmb_gemm_tail_rows": (_i64, [_i64, _i64]),
    "cmb_quantize_fp8_rows": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _i64, _p, _p]),
    "cmb_transpose": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "cmb_colsum": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _p]),
    "cmb_colsum_scaled": (C.c_int, [C.c_int, _p, C.c_int64, C.c_int64, C.c_int64, _p, C.c_int64, C.c_int32, _p, _p]),
    "cmb_cast": (C.c_int, [C.c_int, _p, C.c_int, _p, _i64, _p]),
    "cmb_weight_prep_tiles": (_i64, [_i64, _i64]),
    "cmb_weight_prep_one": (C.c_int, [C.POINTER(PrepJob), _p]),
    "cmb_weight_prep": (C.c_int, [_p, _i32, _i64, _p]),
    "cmb_row_stats": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _f, _p, _p, _p]),
    "cmb_layernorm_bwd_multi": (C.c_int, [C.POINTER(LnMultiDesc), _p]),
    "cmb_layernorm_fwd_multi": (C.c_int, [C.POINTER(LnFwdMultiDesc), _p]),
    "cmb_layernorm_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _i32, _i32, _p, _p, _f, _p, _i64, _p, _p, _p]),
    "cmb_layernorm_bwd": (C.c_int, [C.c_int, _p, _i64, _p, _i64, _i64, _i64, _p, _i32, _i32, _p, _p, _p, _p, _i64,
                                    _i32, _p, _p, _p, _p]),
    "cmb_rmsnorm_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _p, _f, _p, _p, _p]),
    "cmb_rmsnorm_bwd": (C.c_int, [C.c_int, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "cmb_add_rmsnorm_fwd": (C.c_int, [C.c_int, _p, _p, _i64, _i64, _p, _f, _p, _p, _p, _p]),
    "cmb_rmsnorm_bwd_add": (C.c_int, [C.c_int, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "cmb_rope_table": (C.c_int, [_p, _i64, _i64, _f, _p, _p, _p]),
    "cmb_rope_apply": (C.c_int, [C.c_int, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p]),
    "cmb_sva_attn_fwd": (C.c_int, [C.POINTER(SvaDesc), _p]),
    "cmb_sva_attn_bwd": (C.c_int, [C.POINTER(SvaDesc), _p]),
    "cmb_sva_abs_fwd": (C.c_int, [C.POINTER(SvaAbsDesc), _p]),
    "cmb_sva_abs_bwd": (C.c_int, [C.POINTER(SvaAbsDesc), _p]),
    "cmb_embed_splice_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i32, _p, _p, _p, _p]),
    "cmb_embed_splice_bwd": (C.c_int, [C.c_int, _p, _p, _i64, _i64, _i64, _i32, _p, _p, _p]),
    "cmb_sva_fold_kv_fwd": (C.c_int, [_p] * 6 + [_i64, _i64, _p, _p, _p]),
    "cmb_sva_fold_kv_bwd": (C.c_int, [_p] * 8 + [_i64, _i64] + [_p] * 7 + [_i64, _p]),
    "cmb_sva_fold_kv_bwd_workspace": (_i64, [_i64, _i64]),
    "cmb_token_mean_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _p]),
    "cmb_token_mean_bwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _p]),
    "cmb_vit_attn_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i32, _i32, _f, _p, _i32, _p]),
    "cmb_patchify_nchw": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _i64, _i32, C.c_int, _p, _i64, _p]),
    "cmb_patchify2x2_nhwc": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "cmb_dwconv7x7_nhwc": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p]),
    "cmb_dwconv7x7_wgrad": (C.c_int, [C.c_int, _p, _p, _i64, _i64, _i64, _i64, _p, _i32, _p]),
    "cmb_resample_bilinear": (C.c_int, [C.c_int, _p, _i64, _i32, _i32, _i64, _i64, _i64, _p, _i32, _i32, _i64, _i64, _p]),
    "cmb_act_mul": (C.c_int, [C.c_int, _i32, _p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "cmb_act_bwd": (C.c_int, [C.c_int, _i32, _p, _p, _i64, _p, _p]),
    "cmb_bcast_rows": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _p]),
    "cmb_cross_entropy_fwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _i64, _p, _p, _p]),
    "cmb_cross_entropy_bwd": (C.c_int, [C.c_int, _p, _i64, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p]),
    "cmb_qkv_rope": (C.c_int, [C.c_int, _i32, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p]),
    "cmb_flash_attn_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i32, _i64,
                                     _p, _p, _p, _p]),
    "cmb_flash_attn_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _f,
                                     _i32, _i64, _p, _p, _p, _p, _p, _p]),
    "cmb_swiglu_bwd": (C.c_int, [C.c_int, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "cmb_resize_coeffs": (C.c_int, [_i32, _i32, _p, _p]),
    "cmb_image_preprocess": (C.c_int, [_p, _p, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "cmb_copy_rows": (C.c_int, [C.c_int, _p, C.POINTER(RowMap), _p, C.POINTER(RowMap), _i64, _i64, _p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared object (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CambrianAmdError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cambrian_amd/csrc`.  There is no PyTorch fallback for the hot path.")
        lib = C.CDLL(LIB_PATH)
        # lab only (tools/r06_lab.py, same-box A/B of an OLDER build of the library given by CAMBRIAN_AMD_LIB): entry points the
        # old build lacks stay unbound and its ABI revision is accepted — the caller uses only what both builds share
        lenient = bool(os.environ.get("CAMBRIAN_AMD_LIB")) and os.environ.get("CAMBRIAN_AMD_LIB_LENIENT") == "1"
        for name, (res, args) in SIGNATURES.items():
            if lenient and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = lib.cmb_abi_version()
        if got != ABI_VERSION and not lenient:
            raise CambrianAmdError(
                f"{LIB_PATH} reports ABI revision {got}, this binding is written against {ABI_VERSION}: every symbol of a "
                "stale build still resolves but argument lists have shifted — rebuild it (`make -C cambrian_amd/csrc`)")
        # A/B runs: CAMBRIAN_AMD_KNOBS="<knob>=<value>,..." (enum cmb_knob_id) overrides the library's kernel-selection
        # defaults for this process
        for item in filter(None, os.environ.get("CAMBRIAN_AMD_KNOBS", "").split(",")):
            k, v = item.split("=")
            if lib.cmb_knob_set(int(k), int(v)) != 0:
                raise CambrianAmdError(f"CAMBRIAN_AMD_KNOBS: unknown knob {k}")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise CambrianAmdError(f"{what} failed with {STATUS.get(rc, rc)}")


def dtype_code(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return BF16
    if t == torch.float32:
        return F32
    raise CambrianAmdError(f"unsupported dtype {t} (bf16 and fp32 only)")


def stream_ptr(device: Optional[torch.device] = None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CambrianAmdError("cambrian_amd kernels need tensors on a ROCm device; got a CPU tensor "
                                   "(there is no CPU fallback on the product path)")


def identity_map(ld: int) -> RowMap:
    return RowMap(0, 1, 0, 0, ld)


def make_map(n1: int, n2: int, s0: int, s1: int, s2: int) -> RowMap:
    return RowMap(n1, n2, s0, s1, s2)


def knob_set(knob: int, value: int) -> None:
    """cmb_knob_set: which of several equivalent kernels an entry point launches (include/cambrian_amd.h)."""
    check(load().cmb_knob_set(knob, value), "cmb_knob_set")


def knob_get(knob: int) -> int:
    return load().cmb_knob_get(knob)
