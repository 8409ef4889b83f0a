"""Image pre-processing on the step boundary, on the GPU.

The reference prepares every sample on the CPU, once per tower (``train_fsdp.py:985-1008``, ``mm_utils.py:183-201``):

    image_aux = expand2square(image, tuple(int(x*255) for x in processor.image_mean)).resize((R, R))
    image_aux = processor.preprocess(image_aux, return_tensors='pt')['pixel_values'][0]

and ships four float tensors per image to the device.  Here only the decoded uint8 pixels cross PCIe (one pinned
blob per batch, one async copy) and ``cmb_image_preprocess`` (csrc/preprocess.hip) does the letter-box, Pillow's
fixed-point bicubic resample and the processor's pointwise rescale/normalise for the whole batch x all towers in two
launches, on a side stream, one batch ahead of the step.  Integer work: outputs are bit-identical to the reference
expression above (tests/test_preprocess.py, tests/test_preprocess_gpu.py).

Host logic in this file: the per-tower pixel spec (background colour, 3x256 level table in the processor's own
float arithmetic), the job table (`lib.ImageJob` = ``cmb_image_job``), the blob layout, slot reuse and the
prefetcher.  Coefficient rows come from the library's host function ``cmb_resize_coeffs`` and are cached per
(source side, R).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import lib as L

_JOB_DTYPE = np.dtype([("src_off", "<i8"), ("tmp_off", "<i8"), ("dst_off", "<i8"), ("w", "<i4"), ("h", "<i4"),
                       ("side", "<i4"), ("off_x", "<i4"), ("off_y", "<i4"), ("out_side", "<i4"), ("ksize", "<i4"),
                       ("coef_off", "<i4"), ("bounds_off", "<i4"), ("lut_off", "<i4"), ("background", "<u4"),
                       ("reserved", "<i4")])
assert _JOB_DTYPE.itemsize == C.sizeof(L.ImageJob)

_OUT_CODES = {torch.bfloat16: L.BF16, torch.float32: L.F32, torch.float16: L.F16}


# ---------------------------------------------------------------------------------------------------------------
# per-tower pixel spec
# ---------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class TowerPixelSpec:
    out_side: int
    pad_mean: Tuple[float, ...]          # processor.image_mean: only the letter-box colour comes from it
    mean: Tuple[float, ...]
    std: Tuple[float, ...]
    flavour: str = "hf"                  # 'hf' (rescale+normalize of HF image processors) | 'torchvision'

    @property
    def background(self) -> int:
        r, g, b = (int(x * 255) for x in self.pad_mean)          # train_fsdp.py:1006
        return r | (g << 8) | (b << 16)

    def lut(self) -> np.ndarray:
        """float32 [3, 256]: what the processor turns each uint8 level into, computed in the processor's own
        arithmetic so the table is exact, not merely close."""
        u = np.arange(256, dtype=np.uint8)
        m = np.asarray(self.mean, np.float32)[:, None]
        s = np.asarray(self.std, np.float32)[:, None]
        if self.flavour == "hf":               # uint8 * python float -> float64 -> float32, then (x - m) / s
            x = (u * (1 / 255)).astype(np.float32)
        elif self.flavour == "torchvision":    # ToTensor: float32 / 255 ; Normalize: sub, div
            x = u.astype(np.float32) / np.float32(255)
        else:
            raise ValueError(f"unknown processor flavour {self.flavour!r}")
        return ((x[None, :] - m) / s).astype(np.float32)


def spec_from_processor(proc) -> TowerPixelSpec:
    """Read a tower's ``image_processor`` (HF image processor or ``ProcessorWrapper`` around an open_clip /
    torchvision-style transform, base_encoder.py:12-30) into a pixel spec."""
    if not hasattr(proc, "image_mean"):
        raise ValueError("processor has no image_mean (mm_utils.py:192 skips the letter-box for those; unsupported)")
    side = int(proc.crop_size["height"])
    if int(proc.crop_size.get("width", side)) != side:
        raise ValueError("towers take square inputs")
    pad_mean = tuple(float(x) for x in proc.image_mean)
    if hasattr(proc, "image_std"):                                    # HF CLIPImageProcessor / BitImageProcessor
        factor = getattr(proc, "rescale_factor", 1 / 255)
        if abs(factor - 1 / 255) > 1e-12 or not getattr(proc, "do_normalize", True):
            raise ValueError("only rescale 1/255 + normalize processors are supported")
        return TowerPixelSpec(side, pad_mean, pad_mean, tuple(float(x) for x in proc.image_std), "hf")
    t = getattr(proc, "_transforms", None)
    stages = list(getattr(t, "transforms", [t]))
    for st in stages:                                                  # Normalize-like stage carries mean / std
        if hasattr(st, "mean") and hasattr(st, "std"):
            mean = tuple(float(x) for x in st.mean)
            std = tuple(float(x) for x in st.std)
            return TowerPixelSpec(side, pad_mean, mean, std, getattr(st, "flavour", "torchvision"))
    raise ValueError(f"cannot find mean/std in processor {type(proc).__name__}")


# ---------------------------------------------------------------------------------------------------------------
# job table
# ---------------------------------------------------------------------------------------------------------------
CoefFn = Callable[[int, int], Tuple[np.ndarray, np.ndarray, int]]


def library_coefficients(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [R,2], coefs int32 [ksize,R] tap-major, ksize) from the library's host function."""
    lib = L.load()
    ksize = lib.cmb_resize_coeffs(in_size, out_size, None, None)
    if ksize <= 0:
        L.check(ksize, "cmb_resize_coeffs")
    bounds = np.empty((out_size, 2), np.int32)
    coefs = np.empty((ksize, out_size), np.int32)
    rc = lib.cmb_resize_coeffs(in_size, out_size, bounds.ctypes.data, coefs.ctypes.data)
    if rc != ksize:
        L.check(rc if rc < 0 else -1, "cmb_resize_coeffs")
    return bounds, coefs, ksize


def _align(n: int, a: int = 16) -> int:
    return (n + a - 1) // a * a


@dataclass
class ImageBatchPlan:
    """Host-side description of one launch: everything the device needs, in numpy."""
    jobs: np.ndarray                      # structured, [n_images * n_towers], tower-major within an image
    bounds: np.ndarray                    # int32, concatenated (first, count) rows
    coefs: np.ndarray                     # int32, concatenated tap-major coefficient blocks
    lut: np.ndarray                       # float32 [n_towers * 768]
    src_bytes: int
    tmp_bytes: int
    out_elems: int
    tower_out: List[Tuple[int, int]]      # (element offset, side) of each tower's [B,3,R,R] block
    n_images: int
    src_offsets: List[int] = field(default_factory=list)


def as_uint8_hwc(image) -> np.ndarray:
    """PIL image (converted to RGB, train_fsdp.py:983) or array -> contiguous uint8 [h, w, 3]."""
    if hasattr(image, "convert"):
        image = np.asarray(image.convert("RGB"))
    a = np.ascontiguousarray(image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected uint8 [h, w, 3] pixels, got {a.dtype} {a.shape}")
    return a


def build_plan(shapes: Sequence[Tuple[int, int]], specs: Sequence[TowerPixelSpec], coef_fn: CoefFn,
               cache: Optional[Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, int]]] = None) -> ImageBatchPlan:
    """``shapes`` = (h, w) of every image of the batch.  Output block of tower t is [B,3,R_t,R_t]."""
    cache = {} if cache is None else cache
    B = len(shapes)
    jobs = np.zeros(B * len(specs), _JOB_DTYPE)
    tower_out, off = [], 0
    for sp in specs:
        tower_out.append((off, sp.out_side))
        off += _align(B * 3 * sp.out_side * sp.out_side, 8)          # keep every block 16-byte aligned
    out_elems = off
    src_offsets, soff = [], 0
    for h, w in shapes:
        src_offsets.append(soff)
        soff += _align(h * w * 3, 16)
    used: Dict[Tuple[int, int], Tuple[int, int, int]] = {}            # (S, R) -> (bounds_off, coef_off, ksize)
    b_parts, c_parts, b_off, c_off, tmp_off = [], [], 0, 0, 0
    for b, (h, w) in enumerate(shapes):
        if h <= 0 or w <= 0:
            raise ValueError(f"empty image {w}x{h}")
        side = max(h, w)
        for t, sp in enumerate(specs):
            j = jobs[b * len(specs) + t]
            R = sp.out_side
            j["src_off"], j["w"], j["h"], j["side"] = src_offsets[b], w, h, side
            j["off_x"], j["off_y"] = ((side - w) // 2, 0) if h > w else (0, (side - h) // 2)   # mm_utils.py:157-165
            j["out_side"], j["background"], j["lut_off"] = R, sp.background, t * 768
            j["dst_off"] = tower_out[t][0] + b * 3 * R * R
            if side == R:
                continue                                               # Image.resize returns a copy: ksize stays 0
            key = (side, R)
            if key not in used:
                if key not in cache:
                    cache[key] = coef_fn(side, R)
                bounds, coefs, ksize = cache[key]
                used[key] = (b_off, c_off, ksize)
                b_parts.append(bounds.reshape(-1))
                c_parts.append(coefs.reshape(-1))
                b_off += bounds.size
                c_off += coefs.size
            j["bounds_off"], j["coef_off"], j["ksize"] = used[key]
            j["tmp_off"] = tmp_off
            tmp_off += _align(3 * side * ((R + 3) & ~3), 16)
    bounds = np.concatenate(b_parts) if b_parts else np.zeros(0, np.int32)
    coefs = np.concatenate(c_parts) if c_parts else np.zeros(0, np.int32)
    lut = np.concatenate([sp.lut().reshape(-1) for sp in specs]).astype(np.float32)
    return ImageBatchPlan(jobs, bounds.astype(np.int32, copy=False), coefs.astype(np.int32, copy=False), lut,
                          soff, tmp_off, out_elems, tower_out, B, src_offsets)


# ---------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------
class _Slot:
    def __init__(self):
        self.pinned: Optional[torch.Tensor] = None
        self.blob: Optional[torch.Tensor] = None
        self.tmp: Optional[torch.Tensor] = None
        self.event: Optional[torch.cuda.Event] = None


class GpuImagePreprocessor:
    """uint8 images -> the per-tower pixel tensors the model's ``images`` argument takes.

    ``pre(images)`` returns ``[Tensor [B,3,R_t,R_t] for t in towers]`` on ``device`` in ``out_dtype``.  Work is
    enqueued on the current stream (or ``stream``); nothing synchronises the host except re-use of a staging slot
    whose previous batch is still in flight (``slots`` deep)."""

    def __init__(self, processors_or_specs: Sequence, device, out_dtype: torch.dtype = torch.bfloat16,
                 slots: int = 2):
        self.specs = [p if isinstance(p, TowerPixelSpec) else spec_from_processor(p) for p in processors_or_specs]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.CambrianAmdError("GpuImagePreprocessor needs a ROCm device (no CPU fallback on the product path)")
        if out_dtype not in _OUT_CODES:
            raise L.CambrianAmdError(f"unsupported pixel dtype {out_dtype}")
        self.out_dtype = out_dtype
        self._lib = L.load()
        self._cache: Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, int]] = {}
        self._slots = [_Slot() for _ in range(max(1, slots))]
        self._turn = 0
        self.last_plan: Optional[ImageBatchPlan] = None

    @staticmethod
    def _grow(t: Optional[torch.Tensor], nbytes: int, **kw) -> torch.Tensor:
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 1) * 5 // 4 + 64, dtype=torch.uint8, **kw)
        return t

    def __call__(self, images: Sequence, stream: Optional[torch.cuda.Stream] = None) -> List[torch.Tensor]:
        arrays = [as_uint8_hwc(im) for im in images]
        plan = build_plan([a.shape[:2] for a in arrays], self.specs, library_coefficients, self._cache)
        self.last_plan = plan
        slot = self._slots[self._turn % len(self._slots)]
        self._turn += 1
        if slot.event is not None:
            slot.event.synchronize()                       # the staging buffers of this slot are free again
        sec = {}
        off = 0
        for name, nbytes in (("jobs", plan.jobs.nbytes), ("bounds", plan.bounds.nbytes), ("coefs", plan.coefs.nbytes),
                             ("lut", plan.lut.nbytes), ("src", plan.src_bytes)):
            sec[name] = off
            off = _align(off + nbytes, 256)
        total = off
        slot.pinned = self._grow(slot.pinned, total, pin_memory=True)
        host = slot.pinned.numpy()
        host[sec["jobs"]:sec["jobs"] + plan.jobs.nbytes] = plan.jobs.view(np.uint8)
        host[sec["bounds"]:sec["bounds"] + plan.bounds.nbytes] = plan.bounds.view(np.uint8)
        host[sec["coefs"]:sec["coefs"] + plan.coefs.nbytes] = plan.coefs.view(np.uint8)
        host[sec["lut"]:sec["lut"] + plan.lut.nbytes] = plan.lut.view(np.uint8)
        for a, so in zip(arrays, plan.src_offsets):
            host[sec["src"] + so:sec["src"] + so + a.size] = a.reshape(-1)
        stream = stream or torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            slot.blob = self._grow(slot.blob, total, device=self.device)
            slot.tmp = self._grow(slot.tmp, plan.tmp_bytes, device=self.device)
            slot.blob[:total].copy_(slot.pinned[:total], non_blocking=True)
            out = torch.empty(plan.out_elems, dtype=self.out_dtype, device=self.device)
            base = slot.blob.data_ptr()
            rc = self._lib.cmb_image_preprocess(base + sec["jobs"], plan.jobs.ctypes.data, len(plan.jobs),
                                                base + sec["src"], base + sec["bounds"], base + sec["coefs"],
                                                base + sec["lut"], _OUT_CODES[self.out_dtype], slot.tmp.data_ptr(),
                                                out.data_ptr(), stream.cuda_stream)
            L.check(rc, "cmb_image_preprocess")
            slot.event = torch.cuda.Event()
            slot.event.record(stream)
        B = plan.n_images
        return [out[o:o + B * 3 * r * r].view(B, 3, r, r) for o, r in plan.tower_out]


class DevicePrefetcher:
    """Runs the H2D copy and the pre-processing of batch k+1 on a side stream while the step of batch k computes.

    ``batches`` yields collator dicts (train/data_layout.py) whose ``raw_images`` entry is the list of decoded
    uint8 images; the yielded dict has ``images`` (per-tower device tensors) instead and every tensor moved to the
    device.  The consumer's stream waits on the batch's event — no host synchronisation."""

    def __init__(self, batches: Iterable[dict], preprocessor: GpuImagePreprocessor):
        self._it: Iterator[dict] = iter(batches)
        self._pre = preprocessor
        self._stream = torch.cuda.Stream(device=preprocessor.device)
        self._next: Optional[Tuple[dict, torch.cuda.Event]] = None
        self._stage()

    def _stage(self) -> None:
        try:
            host = next(self._it)
        except StopIteration:
            self._next = None
            return
        dev = self._pre.device
        out = {}
        with torch.cuda.stream(self._stream):
            for k, v in host.items():
                if k == "raw_images":
                    out["images"] = self._pre(v, stream=self._stream)
                elif torch.is_tensor(v):
                    out[k] = v.pin_memory().to(dev, non_blocking=True) if not v.is_cuda else v
                elif isinstance(v, list) and v and all(torch.is_tensor(x) for x in v):
                    out[k] = [x.pin_memory().to(dev, non_blocking=True) for x in v]
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._next = (out, ev)

    def __iter__(self):
        return self

    def __next__(self) -> dict:
        if self._next is None:
            raise StopIteration
        batch, ev = self._next
        cur = torch.cuda.current_stream(self._pre.device)
        cur.wait_event(ev)
        for v in batch.values():
            for t in (v if isinstance(v, list) else [v]):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        self._stage()
        return batch
