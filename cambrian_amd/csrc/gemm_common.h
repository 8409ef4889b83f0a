// gemm_common.h — parameter block, MFMA wrappers and the fused row-contiguous epilogue shared by the two
// tile configurations of the NT GEMM (gemm.hip: 128x128 / 4 waves; gemm256.hip: 256x256 / 8 waves, 8-phase).
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_layout.h"

namespace cmb_gemm_detail {

struct GemmParams {
  int M, N, K;
  const char* A; RowMap a_map;
  const char* B; int64_t ldb;
  char* C; RowMap c_map;
  const float* bias;
  const float* colscale;
  const char* R; RowMap r_map;
  char* P; RowMap p_map;
  int act;
  float alpha, beta;
  int out_f32;
  int tiles_m, tiles_n;
  int k_per_split;
  float* slabs;
  const float* a_scale;  // fp8 operands: per-row dequantisation factors (nullptr otherwise)
  const float* b_scale;
  int batch;                 // > 1: blockIdx.z walks independent problems of the same shape (128 x 128 kernel only)
  int64_t a_bs, b_bs, c_bs;  // element strides of A / B / C between consecutive problems of a batch
  int slab_rows;             // rows of one split-K slab (= M; batch * M for cmb_gemm_tn's batched split-K)
  const float* row_mean;     // LayerNorm folded into this linear (cmb_gemm_desc.row_mean): v = rstd[m] (acc - mean[m] colsum[n]) + bias[n]
  const float* row_rstd;
  const float* col_sum;
};

// storage tag of an OCP e4m3fn operand byte (gfx950 native fp8)
struct fp8e4m3_t { uint8_t v; };
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

// Mfma<T>: operand fragment of one 128-byte LDS tile row, the MFMA that consumes it, and the element type the
// epilogue reads / writes (residual, pre_out, C).  ``load`` reads sub-step ks (of KSTEPS) of the swizzled row at
// ``row`` (gemm_layout.h): 16-byte chunk 2*ks + (lane >> 5) for the 16-bit / 32-bit types, the chunk pair
// 4*ks + 2*(lane >> 5) + {0, 1} for fp8 (32 consecutive k per lane).  A and B use the same k <-> slot map, which is all
// a dot product needs.
template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  typedef bf16x8_t frag_t;
  typedef bf16_t out_t;
  static constexpr int KSTEPS = 4;
  static __device__ __forceinline__ frag_t load(const char* row, int swz, int ks, int lane) {
    return *reinterpret_cast<const frag_t*>(row + ((gl_frag_chunk(ks, lane) ^ swz) << 4));
  }
  static __device__ __forceinline__ void run(const frag_t& a, const frag_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  typedef f32x4_t frag_t;
  typedef float out_t;
  static constexpr int KSTEPS = 4;
  static __device__ __forceinline__ frag_t load(const char* row, int swz, int ks, int lane) {
    return *reinterpret_cast<const frag_t*>(row + ((gl_frag_chunk(ks, lane) ^ swz) << 4));
  }
  static __device__ __forceinline__ void run(const frag_t& a, const frag_t& b, f32x16_t& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  }
};
template <> struct Mfma<fp8e4m3_t> {
  typedef i32x8_t frag_t;
  typedef bf16_t out_t;
  static constexpr int KSTEPS = 2;
  static __device__ __forceinline__ frag_t load(const char* row, int swz, int ks, int lane) {
    const int c0 = 4 * ks + 2 * (lane >> 5);
    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(row + ((c0 ^ swz) << 4));
    const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(row + (((c0 | 1) ^ swz) << 4));
    return frag_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  static __device__ __forceinline__ void run(const frag_t& a, const frag_t& b, f32x16_t& c) {
    // cbsz = blgp = 0: both operands e4m3; zero scale operands select the unscaled v_mfma_f32_32x32x64_f8f6f4
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}


// Fused epilogue on 8 consecutive columns of output row gm (fp32 values v[8] straight from the accumulators):
// split-K slab store, or alpha/bias/pre_out/activation/LayerScale/residual/beta and the final store.
// ACT is a compile-time activation code: the callers dispatch on p.act ONCE, outside their row loops (a run-time
// switch in here is re-evaluated for each of the 8 elements: ~10 scalar compare/branch instructions per element).
template <typename T, int ACT>
__device__ __forceinline__ void gemm_epilogue8(const GemmParams& p, int kz, int gm, int gn, float (&v)[8]) {
#pragma clang fp contract(off)   // every fused multiply-add below is written as one: the same roundings as gemm_p5_epilogue.inc
  if (p.slabs) {  // split-K partial: raw fp32 slab, reduced by splitk_reduce_kernel
    Vec8<float>::store(p.slabs + ((int64_t)kz * p.slab_rows + gm) * p.N + gn, v);
    return;
  }
  if (p.a_scale || p.b_scale) {  // fp8 operands: undo the row-wise quantisation scales
    const float sa = p.a_scale ? p.a_scale[gm] : 1.0f;
    float sb[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    if (p.b_scale) load8f(p.b_scale + gn, sb);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sa * sb[e];
  }
  if (p.row_mean) {   // folded LayerNorm: the same two fused multiply-adds per element as gemm_p5_epilogue.inc
    float bb[8], cs[8];
    load8f(p.bias + gn, bb);
    load8f(p.col_sum + gn, cs);
    const float nm = -p.row_mean[gm], rs = p.row_rstd[gm];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(rs, fmaf(nm, cs[e], v[e]), bb[e]);
  } else if (p.bias) {
    float bb[8];
    load8f(p.bias + gn, bb);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], p.alpha, bb[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
  }
  if constexpr (ACT == CMB_ACT_SWIGLU_PAIRS) {
    // (gate, up) column pairs -> 4 outputs at columns gn / 2 .. gn / 2 + 3 of an N / 2 wide C; nothing else applies (dispatch)
    T o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (T)(v[2 * k] * cmb_sigmoid(v[2 * k]) * v[2 * k + 1]);
    T* cp = reinterpret_cast<T*>(p.C) + row_off(p.c_map, (uint32_t)gm) + (gn >> 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) cp[k] = o[k];
    return;
  }
  if (p.P) Vec8<T>::store(reinterpret_cast<T*>(p.P) + row_off(p.p_map, (uint32_t)gm) + gn, v);
  if constexpr (ACT == CMB_ACT_GELU_ERF && std::is_same<T, bf16_t>::value) {
    cmb_gelu_erf_bf16_x8(v);  // bf16 operands / results: common.h (the four pairs' chains interleaved)
  } else if constexpr (ACT != CMB_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_apply(ACT, v[e]);
  }
  if (p.colscale) {
    float ss[8];
    load8f(p.colscale + gn, ss);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= ss[e];
  }
  if (p.R) {
    float rr[8];
    Vec8<T>::load(reinterpret_cast<const T*>(p.R) + row_off(p.r_map, (uint32_t)gm) + gn, rr);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rr[e];
  }
  const int64_t coff = row_off(p.c_map, (uint32_t)gm) + gn;
  if (p.out_f32) {
    float* cp = reinterpret_cast<float*>(p.C) + coff;
    if (p.beta != 0.0f) {
      float old[8];
      load8f(cp, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += p.beta * old[e];
    }
    Vec8<float>::store(cp, v);
  } else {
    Vec8<T>::store(reinterpret_cast<T*>(p.C) + coff, v);
  }
}

// Calls f(std::integral_constant<int, ACT>) with the compile-time twin of the run-time activation code.
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
  switch (act) {
    case CMB_ACT_GELU_ERF: f(std::integral_constant<int, CMB_ACT_GELU_ERF>{}); break;
    case CMB_ACT_GELU_TANH: f(std::integral_constant<int, CMB_ACT_GELU_TANH>{}); break;
    case CMB_ACT_QUICK_GELU: f(std::integral_constant<int, CMB_ACT_QUICK_GELU>{}); break;
    case CMB_ACT_SILU: f(std::integral_constant<int, CMB_ACT_SILU>{}); break;
    case CMB_ACT_SWIGLU_PAIRS: f(std::integral_constant<int, CMB_ACT_SWIGLU_PAIRS>{}); break;
    default: f(std::integral_constant<int, CMB_ACT_NONE>{}); break;
  }
}

// gemm256.hip: 256x256x64 bf16 tile, 8 waves, 8-phase schedule.  Returns CMB_OK / CMB_ERR_LAUNCH.
// sched: 0 = 8-phase ping-pong (two barriers per phase), 1 = in-wave pipeline (one barrier per K-tile).
int launch_gemm256_bf16(GemmParams& p, int splits, int sched, hipStream_t s);

// gemm_p5.hip: persistent 256x256 bf16 tile, 4 waves x (128 x 128), 64-deep tiles, the tile's fragments in registers,
// two LDS buffers (gemm_nt_p5_kernel)
// q != nullptr: a pair launch — q (same activation, whole K, no slabs) runs beside p on its own share of the workgroups
int launch_gemm_p5_bf16(GemmParams& p, int splits, hipStream_t s, GemmParams* q = nullptr);
double gemm_p5_pair_gain(const GemmParams& a, const GemmParams& b, int n_cu);
bool gemm_p5_pair_act_ok(int act);   // activation templates that have a pair instantiation

// gemm_k64.hip: batched problems with a 64-deep contraction and a wide N (the per-head expand products): HBM-write-bound kernel
bool gemm_k64_eligible(const GemmParams& p);
int launch_gemm_k64_batched(const GemmParams& p, hipStream_t s);

// gemm_smallm.hip: M <= 32 rows (the SVA layers' per-image context vectors): 32 output columns per workgroup, K split over its waves
bool gemm_small_m_eligible(const GemmParams& p, int splits);
int launch_gemm_small_m(const GemmParams& p, hipStream_t s);

// gemm_tn.hip: C[M,N] = At[K,M]^T Bt[K,N] (both operands row-major over the contraction rows; p.a_map.s2 = lda, p.K = rows),
// 128 x 128 tile, transposing LDS reads
int launch_gemm_tn_bf16(GemmParams& p, int splits, hipStream_t s);

}  // namespace cmb_gemm_detail
