// llm_ops.hip — HBM-bound kernels on the LLM side of the step (cambrian_llama.py:402-422 loss; Llama MLP gate):
//   * fused shifted cross-entropy over bf16/fp32 logits: forward = one streaming pass (online log-sum-exp in fp32),
//     backward = one pass that overwrites the logits with dlogits.  The reference materialises logits.float()
//     ([B*2048, 128256] fp32 = 8.4 GB at B = 8), a shifted .contiguous() copy of it, and softmax / nll passes over
//     those; here the bf16 logits are read twice and written once.
//   * SwiGLU gate: h = silu(g) * u and its backward (dg, du) in one pass each.
// Roofline class: HBM (one wave-instruction = 1 KiB contiguous; fp32 math).
#include "common.h"

namespace {

// One workgroup (256 threads) per row of `logits` [rows, V].  lse[row] = log(sum_v exp(x_v)); loss[row] =
// lse - x[label] (0 for label == ignore_index).  V % 8 == 0.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) ce_fwd_kernel(const T* __restrict__ logits, int64_t ld, int V,
                                                     const int64_t* __restrict__ labels, int64_t ignore_index,
                                                     float* __restrict__ lse_out, float* __restrict__ loss_out) {
  __shared__ float red_m[4], red_s[4];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  const int nv = VEC ? (V >> 3) : 0;  // !VEC: rows are not 16-byte aligned (odd test vocabularies): scalar loop only
  float m = -INFINITY, s = 0.f;
  for (int i = threadIdx.x; i < nv; i += 256) {
    float v[8];
    Vec8<T>::load(x + i * 8, v);
    float vm = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) vm = fmaxf(vm, v[e]);
    const float mn = fmaxf(m, vm);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += cmb_exp(v[e] - mn);
    s = s * cmb_exp(m - mn) + acc;
    m = mn;
  }
  for (int i = nv * 8 + threadIdx.x; i < V; i += 256) {
    const float v = (float)x[i];
    const float mn = fmaxf(m, v);
    s = s * cmb_exp(m - mn) + cmb_exp(v - mn);
    m = mn;
  }
  // wave reduce (m, s), then the 4 waves through LDS
  const float wm = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * cmb_exp(m - wm));  // idle lanes (V < 2048) carry m = -inf
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red_m[wave] = wm; red_s[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) bs += (red_m[w] == -INFINITY) ? 0.f : red_s[w] * cmb_exp(red_m[w] - bm);
    const float lse = bm + logf(bs);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    loss_out[row] = (lab == ignore_index) ? 0.f : (lse - (float)x[lab]);
  }
}

// dlogits[row, v] = (exp(x_v - lse[row]) - [v == label]) * scale[0]   (0 for ignored rows); may alias logits.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) ce_bwd_kernel(const T* logits, int64_t ld, int V, const int64_t* __restrict__ labels,
                                                     int64_t ignore_index, const float* __restrict__ lse,
                                                     const float* __restrict__ scale, T* dlogits, int64_t ldd) {
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  T* d = dlogits + row * ldd;
  const int nv = VEC ? (V >> 3) : 0;
  const int64_t lab = labels[row];
  if (lab == ignore_index) {
    const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nv; i += 256) Vec8<T>::store(d + i * 8, z);
    for (int i = nv * 8 + threadIdx.x; i < V; i += 256) d[i] = (T)0.f;
    return;
  }
  const float l = lse[row], sc = scale[0];
  const int lab_vec = (int)(lab >> 3), lab_e = (int)(lab & 7);
  for (int i = threadIdx.x; i < nv; i += 256) {
    float v[8];
    Vec8<T>::load(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = cmb_exp(v[e] - l);
    if (i == lab_vec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] -= (e == lab_e) ? 1.f : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
    Vec8<T>::store(d + i * 8, v);
  }
  for (int i = nv * 8 + threadIdx.x; i < V; i += 256)
    d[i] = (T)((cmb_exp((float)x[i] - l) - (i == lab ? 1.f : 0.f)) * sc);
}

template <typename T>
__global__ void __launch_bounds__(256) swiglu_bwd_kernel(const T* __restrict__ dh, int64_t lddh, const T* __restrict__ g,
                                                         int64_t ldg, const T* __restrict__ u, int64_t ldu, int64_t rows,
                                                         int D, T* __restrict__ dg, int64_t lddg, T* __restrict__ du,
                                                         int64_t lddu) {
  const int nv = D >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv) * 8;
    float hv[8], gv[8], uv[8], og[8], ou[8];
    Vec8<T>::load(dh + r * lddh + c, hv);
    Vec8<T>::load(g + r * ldg + c, gv);
    Vec8<T>::load(u + r * ldu + c, uv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = cmb_sigmoid(gv[e]);
      const float silu = gv[e] * sg;
      ou[e] = hv[e] * silu;
      og[e] = hv[e] * uv[e] * (sg + silu * (1.0f - sg));  // silu'(g) = s + g*s*(1-s)
    }
    Vec8<T>::store(dg + r * lddg + c, og);
    Vec8<T>::store(du + r * lddu + c, ou);
  }
}

// Fused QKV hand-off of a decoder layer: packed projection output [B*S, (nh + 2*nkv) * Dh] (q heads | k heads |
// v heads per token) <-> the three token-major tensors q [B,S,nh,Dh], k / v [B,S,nkv,Dh] (the attention takes their
// transposed views, so its output comes back token-major and the o_proj input needs no copy),
// with RoPE (rotate-half, fp32 tables) applied to q and k on the way.  MERGE = backward: (dq, dk, dv) -> d(packed)
// with the transposed rotation.  One workgroup per token; a head's Dh elements move as 16-byte vectors.
template <typename T, bool MERGE>
__global__ void __launch_bounds__(256) qkv_rope_kernel(T* __restrict__ packed, const float* __restrict__ cos_t,
                                                       const float* __restrict__ sin_t, int64_t ntok, int S, int nh, int nkv,
                                                       int Dh, T* __restrict__ q, T* __restrict__ k, T* __restrict__ v) {
  const int half = Dh >> 1, gph = half >> 3;
  const int H = nh + 2 * nkv;
  const int items = H * gph;
  for (int64_t tok = blockIdx.x; tok < ntok; tok += gridDim.x) {
    T* pr = packed + tok * (int64_t)H * Dh;
    const float* ct = cos_t + tok * half;
    const float* st = sin_t + tok * half;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
      const int h = it / gph, gi = it - h * gph;
      T* hp;  // this token's row of head h in its head-major tensor
      bool rot = true;
      if (h < nh) hp = q + (tok * nh + h) * Dh;
      else if (h < nh + nkv) hp = k + (tok * nkv + (h - nh)) * Dh;
      else { hp = v + (tok * nkv + (h - nh - nkv)) * Dh; rot = false; }
      T* src = MERGE ? hp : pr + h * Dh;
      T* dst = MERGE ? pr + h * Dh : hp;
      float a[8], bb[8];
      Vec8<T>::load(src + gi * 8, a);
      Vec8<T>::load(src + half + gi * 8, bb);
      if (rot) {
        float c[8], sn[8], oa[8], ob[8];
        load8f(ct + gi * 8, c);
        load8f(st + gi * 8, sn);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s_ = MERGE ? -sn[e] : sn[e];
          oa[e] = a[e] * c[e] - bb[e] * s_;
          ob[e] = bb[e] * c[e] + a[e] * s_;
        }
        Vec8<T>::store(dst + gi * 8, oa);
        Vec8<T>::store(dst + half + gi * 8, ob);
      } else {
        Vec8<T>::store(dst + gi * 8, a);
        Vec8<T>::store(dst + half + gi * 8, bb);
      }
    }
  }
}

}  // namespace

extern "C" int cmb_cross_entropy_fwd(int dtype, const void* logits, int64_t rows, int64_t V, int64_t ld,
                                     const int64_t* labels, int64_t ignore_index, float* lse, float* loss, void* stream) {
  if (!logits || !labels || !lse || !loss || rows < 0 || V <= 0) return CMB_ERR_BAD_ARG;
  if (dtype != CMB_BF16 && dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  const bool vec = cmb_aligned16(logits) && (ld * (dtype == CMB_BF16 ? 2 : 4)) % 16 == 0;
  hipStream_t s = (hipStream_t)stream;
#define CE_FWD(T, VEC)                                                                                              \
  hipLaunchKernelGGL((ce_fwd_kernel<T, VEC>), dim3((unsigned)rows), dim3(256), 0, s, (const T*)logits, ld, (int)V, \
                     labels, ignore_index, lse, loss)
  if (dtype == CMB_BF16) { if (vec) CE_FWD(bf16_t, true); else CE_FWD(bf16_t, false); }
  else { if (vec) CE_FWD(float, true); else CE_FWD(float, false); }
#undef CE_FWD
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_cross_entropy_bwd(int dtype, const void* logits, int64_t rows, int64_t V, int64_t ld,
                                     const int64_t* labels, int64_t ignore_index, const float* lse, const float* scale,
                                     void* dlogits, int64_t ldd, void* stream) {
  if (!logits || !labels || !lse || !scale || !dlogits || rows < 0 || V <= 0) return CMB_ERR_BAD_ARG;
  if (dtype != CMB_BF16 && dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  const int64_t es = dtype == CMB_BF16 ? 2 : 4;
  const bool vec = cmb_aligned16(logits) && cmb_aligned16(dlogits) && (ld * es) % 16 == 0 && (ldd * es) % 16 == 0;
  hipStream_t s = (hipStream_t)stream;
#define CE_BWD(T, VEC)                                                                                              \
  hipLaunchKernelGGL((ce_bwd_kernel<T, VEC>), dim3((unsigned)rows), dim3(256), 0, s, (const T*)logits, ld, (int)V, \
                     labels, ignore_index, lse, scale, (T*)dlogits, ldd)
  if (dtype == CMB_BF16) { if (vec) CE_BWD(bf16_t, true); else CE_BWD(bf16_t, false); }
  else { if (vec) CE_BWD(float, true); else CE_BWD(float, false); }
#undef CE_BWD
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_swiglu_bwd(int dtype, const void* dh, int64_t lddh, const void* g, int64_t ldg, const void* u, int64_t ldu,
                              int64_t rows, int64_t D, void* dg, int64_t lddg, void* du, int64_t lddu, void* stream) {
  if (!dh || !g || !u || !dg || !du || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (rows * (D >> 3) + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  if (dtype == CMB_BF16)
    hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dh, lddh,
                       (const bf16_t*)g, ldg, (const bf16_t*)u, ldu, rows, (int)D, (bf16_t*)dg, lddg, (bf16_t*)du, lddu);
  else if (dtype == CMB_F32)
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dh, lddh,
                       (const float*)g, ldg, (const float*)u, ldu, rows, (int)D, (float*)dg, lddg, (float*)du, lddu);
  else
    return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_qkv_rope(int dtype, int32_t merge, void* packed, const float* cos_t, const float* sin_t, int64_t B,
                            int64_t S, int32_t nh, int32_t nkv, int32_t Dh, void* q, void* k, void* v, void* stream) {
  if (!packed || !cos_t || !sin_t || !q || !k || !v || B < 0 || S <= 0 || nh <= 0 || nkv <= 0 || Dh <= 0 || (Dh & 15))
    return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t ntok = B * S;
  const int64_t blocks = ntok > 65535 ? 65535 : ntok;
#define QKV(T, M)                                                                                                  \
  hipLaunchKernelGGL((qkv_rope_kernel<T, M>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)packed, cos_t, sin_t, ntok, \
                     (int)S, nh, nkv, Dh, (T*)q, (T*)k, (T*)v)
  if (dtype == CMB_BF16) { if (merge) QKV(bf16_t, true); else QKV(bf16_t, false); }
  else if (dtype == CMB_F32) { if (merge) QKV(float, true); else QKV(float, false); }
  else return CMB_ERR_BAD_ARG;
#undef QKV
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
