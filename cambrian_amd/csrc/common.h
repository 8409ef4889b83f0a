// common.h — shared device helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cambrian_amd.h"
#include <atomic>

// "this kernel's dynamic-LDS attribute has been set" per DEVICE (hipFuncSetAttribute applies to the current device; a process-wide
// bool left a second device of the process, or a first call racing on another thread, without it: ADVICE r5).  Setting an
// attribute twice is harmless, so the flag only needs to be monotonic.
struct CmbAttrOnce {
  std::atomic<uint32_t> mask{0};
  uint32_t need() {   // 0 = already done on the current device, else the device's bit
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    const uint32_t bit = 1u << (d & 31);
    return (mask.load(std::memory_order_acquire) & bit) ? 0u : bit;
  }
  void done(uint32_t bit) { mask.fetch_or(bit, std::memory_order_release); }
};

typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CMB_WAVE 64

// 8 fp32 -> 8 bf16 with four v_cvt_pk_bf16_f32 (a scalar (bf16_t)x per element costs a convert + a permute each)
__device__ __forceinline__ bf16x8_t cvt8_bf16(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                              float a7) {
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  typedef float f32x2_v __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
  const f32x2_v v0 = {a0, a1}, v1 = {a2, a3}, v2 = {a4, a5}, v3 = {a6, a7};
  const u32x4_v w = {__builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2_v)),
                     __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2_v)),
                     __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2_v)),
                     __builtin_bit_cast(uint32_t, __builtin_convertvector(v3, bf16x2_v))};
  return __builtin_bit_cast(bf16x8_t, w);
}

#define CMB_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return CMB_ERR_LAUNCH;            \
  } while (0)

static inline bool cmb_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// run-time kernel-selection knobs (cmb_knob_set, include/cambrian_amd.h): storage in elementwise.hip
extern int g_cmb_knobs[CMB_KNOB_COUNT];
static inline int cmb_knob(int k) { return g_cmb_knobs[k]; }

// ---- rowmap -------------------------------------------------------------------------------
struct RowMap {
  uint32_t n1, n2;
  int64_t s0, s1, s2;
};
static inline RowMap make_rowmap(const cmb_rowmap& m) {
  RowMap r;
  r.n1 = (uint32_t)m.n1; r.n2 = (uint32_t)(m.n2 ? m.n2 : 1);
  r.s0 = m.s0; r.s1 = m.s1; r.s2 = m.s2;
  return r;
}
__host__ __device__ static inline int64_t row_off(const RowMap& m, uint32_t r) {
  if (m.n1 == 0) return (int64_t)r * m.s2;
  uint32_t a = r / m.n1, rem = r - a * m.n1;
  uint32_t b = rem / m.n2, c = rem - b * m.n2;
  return (int64_t)a * m.s0 + (int64_t)b * m.s1 + (int64_t)c * m.s2;
}

// ---- 8-element vector load/store with fp32 math ------------------------------------------------
// A "vec8" is 8 consecutive elements: 16 bytes of bf16 or 32 bytes of fp32.
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8_t x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8_t*>(p) = x;
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
    f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4_t a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4_t*>(p) = a;
    *reinterpret_cast<f32x4_t*>(p + 4) = b;
  }
};

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) { Vec8<float>::load(p, v); }

// ---- wave reductions (64 lanes) ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- activations (fp32) -------------------------------------------------------------------------
// Branch-free, division-free forms: the GEMM epilogue applies them to every output element with the matrix
// pipe idle, so their VALU cost is paid in full (ConvNeXt fc1 + GELU: 805 M elements per launch).
//   cmb_exp     v_exp_f32 on x*log2(e)                       (~2 ulp)
//   cmb_rcp     v_rcp_f32                                    (1 ulp)
//   cmb_erf     two minimax polynomials (|x| <= 0.927734375: odd polynomial in x; else 1 - exp(poly(|x|))),
//               both evaluated, selected per lane; < 1 ulp against erf() on [-6, 6] (tests/test_act_math.py
//               re-evaluates the same coefficients in numpy against math.erf).
__device__ __forceinline__ float cmb_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float cmb_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float cmb_sigmoid(float x) {
#pragma clang fp contract(off)
  return cmb_rcp(1.0f + cmb_exp(-x));
}
__device__ __forceinline__ float cmb_tanh(float u) {
#pragma clang fp contract(off)
  return 1.0f - 2.0f * cmb_rcp(1.0f + cmb_exp(2.0f * u));
}
__device__ __forceinline__ float cmb_erf(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - cmb_exp(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}

// The bf16 GEMM epilogues' erf GELU (the activation runs with the matrix pipe idle and the result is rounded to bf16; the
// fp32 kernels and the stand-alone activation kernels keep cmb_erf):
//     GELU(x) = relu(x) - t * Phi(-t),  t = min(|x|, 7),  Phi(-t) = 2^P(t)
// with P a degree-5 weighted minimax fit of log2 Phi(-t) on [0, 7] (monotone decreasing to P(7) = -39.9: beyond the clamp the
// correction term is 7e-12).  |error| <= 6.4e-6 absolute (2e-6 relative at x = 3), <= 4e-5 relative wherever |GELU| > 1e-3 —
// 50 x below half a bf16 ulp (tests/test_act_math.py re-evaluates these coefficients in numpy).  ONE transcendental
// (v_exp_f32) and 5 fma + min + max + fma per element; the first round-4 form had degree 7 (1e-5 relative: more than a bf16
// result can show), rounds 2-3 used Abramowitz-Stegun 7.1.26 (v_rcp_f32 AND v_exp_f32, 5 fma + 5 other operations): the
// epilogue of the 4-wave GEMM is issue-bound at one wave per SIMD, every instruction less is ~5.5 cycles per element pair
// (profiles/r04_lab.md).  Every operation is an explicit fma / min / max (no contraction left to the compiler), so all GEMM
// kernels agree bit for bit.
__device__ __forceinline__ float cmb_gelu_erf_bf16(float x) {
  const float t = fminf(fabsf(x), 7.0f);
  float p = -0.0003763301356229931f;
  p = fmaf(p, t, 0.0063428147695958614f);
  p = fmaf(p, t, -0.0498826801776886f);
  p = fmaf(p, t, -0.4620113670825958f);
  p = fmaf(p, t, -1.1501027345657349f);
  p = fmaf(p, t, -1.0000579357147217f);
  return fmaf(-t, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.0f));
}
// the same, two elements at a time on 2-vectors (identical results): hipcc keeps the scalar form as v_fmaak_f32 (literal
// constants) per element; on vector operands the Horner steps become v_pk_fma_f32 — half the instructions, which is what
// counts: one wave per SIMD issues an instruction every ~5.5 cycles whatever it is (profiles/r04_lab.md)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cmb_gelu_erf_bf16_pair(float& a, float& b) {
  const f32x2_t t = {fminf(fabsf(a), 7.0f), fminf(fabsf(b), 7.0f)};
  f32x2_t p = {-0.0003763301356229931f, -0.0003763301356229931f};
  p = __builtin_elementwise_fma(p, t, (f32x2_t){0.0063428147695958614f, 0.0063428147695958614f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){-0.0498826801776886f, -0.0498826801776886f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){-0.4620113670825958f, -0.4620113670825958f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){-1.1501027345657349f, -1.1501027345657349f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){-1.0000579357147217f, -1.0000579357147217f});
  const f32x2_t h = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  const f32x2_t r = {fmaxf(a, 0.0f), fmaxf(b, 0.0f)};
  const f32x2_t o = __builtin_elementwise_fma(-t, h, r);
  a = o[0];
  b = o[1];
}

// eight elements at a time, the four pairs' Horner chains INTERLEAVED step by step: a dependent v_pk_fma_f32 right behind
// its producer costs a wait state (hipcc puts an s_nop between them), and with one wave per SIMD nothing else fills it —
// the pair-by-pair order above ran one s_nop per multiply-add (same results, operation for operation)
__device__ __forceinline__ void cmb_gelu_erf_bf16_x8(float (&v)[8]) {
  f32x2_t t[4], p[4], r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[k] = (f32x2_t){fminf(fabsf(v[2 * k]), 7.0f), fminf(fabsf(v[2 * k + 1]), 7.0f)};
    r[k] = (f32x2_t){fmaxf(v[2 * k], 0.0f), fmaxf(v[2 * k + 1], 0.0f)};
    p[k] = (f32x2_t){-0.0003763301356229931f, -0.0003763301356229931f};
  }
#define CMB_GELU_STEP(c)                                                              \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) p[k] = __builtin_elementwise_fma(p[k], t[k], (f32x2_t){c, c});
  CMB_GELU_STEP(0.0063428147695958614f)
  CMB_GELU_STEP(-0.0498826801776886f)
  CMB_GELU_STEP(-0.4620113670825958f)
  CMB_GELU_STEP(-1.1501027345657349f)
  CMB_GELU_STEP(-1.0000579357147217f)
#undef CMB_GELU_STEP
  f32x2_t h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = (f32x2_t){__builtin_amdgcn_exp2f(p[k][0]), __builtin_amdgcn_exp2f(p[k][1])};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2_t o = __builtin_elementwise_fma(-t[k], h[k], r[k]);
    v[2 * k] = o[0];
    v[2 * k + 1] = o[1];
  }
}

// (no floating-point contraction in the activation helpers and the GEMM epilogues: which a * b + c pairs the compiler fuses
// depends on the surrounding code, and the GEMM kernels must agree bit for bit — a problem's rows may be split between two of
// them, and the batch an image arrives in decides which kernel its rows take)
__device__ __forceinline__ float act_apply(int act, float x) {
#pragma clang fp contract(off)
  switch (act) {
    case CMB_ACT_GELU_ERF: return 0.5f * x * (1.0f + cmb_erf(x * 0.70710678118654752440f));
    case CMB_ACT_GELU_TANH: {
      const float k0 = 0.7978845608028654f, k1 = 0.044715f;
      return 0.5f * x * (1.0f + cmb_tanh(k0 * (x + k1 * x * x * x)));
    }
    case CMB_ACT_QUICK_GELU: return x * cmb_sigmoid(1.702f * x);
    case CMB_ACT_SILU: return x * cmb_sigmoid(x);
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(int act, float x) {
#pragma clang fp contract(off)
  switch (act) {
    case CMB_ACT_GELU_ERF: {
      const float cdf = 0.5f * (1.0f + cmb_erf(x * 0.70710678118654752440f));
      const float pdf = 0.3989422804014327f * cmb_exp(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case CMB_ACT_GELU_TANH: {
      const float k0 = 0.7978845608028654f, k1 = 0.044715f;
      const float u = k0 * (x + k1 * x * x * x);
      const float t = cmb_tanh(u);
      return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * x * x);
    }
    case CMB_ACT_QUICK_GELU: {
      const float s = cmb_sigmoid(1.702f * x);
      return s + 1.702f * x * s * (1.0f - s);
    }
    case CMB_ACT_SILU: {
      const float s = cmb_sigmoid(x);
      return s + x * s * (1.0f - s);
    }
    default: return 1.0f;
  }
}
