// sva_attn.hip — Spatial-Vision-Aggregator cross-attention core (forward + backward) for gfx950.
//
// Reference semantics: MultiKVCrossAttention.forward, vision_sampler.py:193-230 — one query row per
// problem, keys/values = concatenation over towers of the s_i x s_i window of tower i that lies
// under the query's cell, bool mask (True = attend), softmax in fp32, 16 heads x 64.
// The window partition of cambrian_arch.py:271-287 (view/permute/contiguous) is NOT materialised:
// K/V projections are produced in tower-token-major order and this kernel walks the window by index
// arithmetic (bit-exact gather).
//
// Roofline class: HBM.  q_len = 1 and 19 keys per query => 0.045 GFLOP/img/layer but
// ~2 x 10 944 x 2048 x 2 B = 90 MB/img/layer of K/V traffic: one wave owns one query, each lane owns
// 8*NV contiguous channels (16-byte vector loads, a wave-instruction covers 1 KiB contiguous), the
// per-head dot products are reduced with wave shuffles over the hd/8 lanes that share a head, and an
// online softmax keeps every key's K and V touched exactly once.
#include "common.h"

namespace {

struct SvaParams {
  int B, qside, heads, hd, ntowers, window_major;
  int r[CMB_SVA_MAX_TOWERS];
  const char* q; int64_t ldq;
  const char* kv[CMB_SVA_MAX_TOWERS]; int64_t ldkv[CMB_SVA_MAX_TOWERS];
  const uint8_t* mask[CMB_SVA_MAX_TOWERS];
  char* out; int64_t ldo;
  float* lse;
  const char* dout; int64_t lddo;
  char* dq; int64_t lddq;
  char* dkv[CMB_SVA_MAX_TOWERS];
  float scale;
};

template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
#pragma unroll
  for (int o = LPH >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NV = vec8 chunks per lane (heads*hd = NV*512); LPH = lanes per head = hd/8.
template <typename T, int NV, int LPH>
__global__ void __launch_bounds__(256) sva_fwd_kernel(const SvaParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int C = p.heads * p.hd;
  for (int64_t qi = wave_global; qi < nq; qi += nwaves) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    float qv[NV][8], acc[NV][8], m[NV], l[NV];
    const T* qr = reinterpret_cast<const T*>(p.q) + qi * p.ldq;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      Vec8<T>::load(qr + (lane + c * 64) * 8, qv[c]);
      m[c] = -INFINITY;
      l[c] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { qv[c][e] *= p.scale; acc[c][e] = 0.f; }
    }
    for (int i = 0; i < p.ntowers; ++i) {
      const int r = p.r[i], G = p.qside * r;
      const T* kvb = reinterpret_cast<const T*>(p.kv[i]) + (int64_t)b * G * G * p.ldkv[i];
      const uint8_t* mk = p.mask[i] ? p.mask[i] + qi * r * r : nullptr;
      for (int j = 0; j < r * r; ++j) {
        if (mk && mk[j] == 0) continue;  // wave-uniform: masked key contributes p = 0
        const int ry = j / r, rx = j - ry * r;
        // tower-token-major (window walked by index arithmetic) or the reference's pre-rearranged
        // window-major layout [Bq, r*r, C] (cambrian_arch.py:280-281)
        const int64_t tok = p.window_major ? ((int64_t)t * r * r + j) : ((int64_t)(qy * r + ry) * G + (qx * r + rx));
        const T* kr = kvb + tok * p.ldkv[i];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          float kk[8], vv[8];
          Vec8<T>::load(kr + (lane + c * 64) * 8, kk);
          Vec8<T>::load(kr + C + (lane + c * 64) * 8, vv);
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s += qv[c][e] * kk[e];
          s = head_sum<LPH>(s);
          const float mn = fmaxf(m[c], s);
          const float alpha = expf(m[c] - mn);
          const float pj = expf(s - mn);
          l[c] = l[c] * alpha + pj;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[c][e] = acc[c][e] * alpha + pj * vv[e];
          m[c] = mn;
        }
      }
    }
    T* orow = reinterpret_cast<T*>(p.out) + qi * p.ldo;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const float inv = 1.0f / l[c];
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = acc[c][e] * inv;
      Vec8<T>::store(orow + (lane + c * 64) * 8, o);
      if (p.lse && (lane % LPH) == 0) {
        const int head = ((lane + c * 64) * 8) / p.hd;
        p.lse[qi * p.heads + head] = m[c] + logf(l[c]);
      }
    }
  }
}

template <typename T, int NV, int LPH>
__global__ void __launch_bounds__(256) sva_bwd_kernel(const SvaParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int C = p.heads * p.hd;
  for (int64_t qi = wave_global; qi < nq; qi += nwaves) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    float qv[NV][8], dov[NV][8], dqv[NV][8], dsum[NV], lse[NV];
    const T* qr = reinterpret_cast<const T*>(p.q) + qi * p.ldq;
    const T* dor = reinterpret_cast<const T*>(p.dout) + qi * p.lddo;
    const T* orow = reinterpret_cast<const T*>(p.out) + qi * p.ldo;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float ov[8];
      Vec8<T>::load(qr + (lane + c * 64) * 8, qv[c]);
      Vec8<T>::load(dor + (lane + c * 64) * 8, dov[c]);
      Vec8<T>::load(orow + (lane + c * 64) * 8, ov);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += dov[c][e] * ov[e]; dqv[c][e] = 0.f; }
      dsum[c] = head_sum<LPH>(s);  // D = rowsum(dO * O) per head
      const int head = ((lane + c * 64) * 8) / p.hd;
      lse[c] = p.lse[qi * p.heads + head];
    }
    for (int i = 0; i < p.ntowers; ++i) {
      const int r = p.r[i], G = p.qside * r;
      const int64_t base = (int64_t)b * G * G;
      const T* kvb = reinterpret_cast<const T*>(p.kv[i]) + base * p.ldkv[i];
      T* dkvb = reinterpret_cast<T*>(p.dkv[i]) + base * p.ldkv[i];
      const uint8_t* mk = p.mask[i] ? p.mask[i] + qi * r * r : nullptr;
      for (int j = 0; j < r * r; ++j) {
        const int ry = j / r, rx = j - ry * r;
        const int64_t tok = p.window_major ? ((int64_t)t * r * r + j) : ((int64_t)(qy * r + ry) * G + (qx * r + rx));
        const T* kr = kvb + tok * p.ldkv[i];
        T* dkr = dkvb + tok * p.ldkv[i];
        const bool masked = mk && mk[j] == 0;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          float dk[8], dv[8];
          if (masked) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
          } else {
            float kk[8], vv[8];
            Vec8<T>::load(kr + (lane + c * 64) * 8, kk);
            Vec8<T>::load(kr + C + (lane + c * 64) * 8, vv);
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s += qv[c][e] * kk[e]; dp += dov[c][e] * vv[e]; }
            s = head_sum<LPH>(s) * p.scale;
            dp = head_sum<LPH>(dp);
            const float pj = expf(s - lse[c]);
            const float ds = pj * (dp - dsum[c]) * p.scale;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              dqv[c][e] += ds * kk[e];
              dk[e] = ds * qv[c][e];
              dv[e] = pj * dov[c][e];
            }
          }
          Vec8<T>::store(dkr + (lane + c * 64) * 8, dk);
          Vec8<T>::store(dkr + C + (lane + c * 64) * 8, dv);
        }
      }
    }
    T* dqr = reinterpret_cast<T*>(p.dq) + qi * p.lddq;
#pragma unroll
    for (int c = 0; c < NV; ++c) Vec8<T>::store(dqr + (lane + c * 64) * 8, dqv[c]);
  }
}

int fill_params(const cmb_sva_desc* d, SvaParams& p, bool bwd) {
  if (!d || !d->q || !d->out || d->B < 0 || d->qside <= 0 || d->heads <= 0 || d->hd <= 0) return CMB_ERR_BAD_ARG;
  if (d->ntowers <= 0 || d->ntowers > CMB_SVA_MAX_TOWERS) return CMB_ERR_BAD_ARG;
  p.B = d->B; p.qside = d->qside; p.heads = d->heads; p.hd = d->hd; p.ntowers = d->ntowers;
  p.window_major = d->window_major;
  p.q = (const char*)d->q; p.ldq = d->ldq;
  p.out = (char*)d->out; p.ldo = d->ldo;
  p.lse = d->lse;
  p.dout = (const char*)d->dout; p.lddo = d->lddo;
  p.dq = (char*)d->dq; p.lddq = d->lddq;
  for (int i = 0; i < d->ntowers; ++i) {
    if (!d->kv[i] || d->r[i] <= 0) return CMB_ERR_BAD_ARG;
    p.r[i] = d->r[i];
    p.kv[i] = (const char*)d->kv[i]; p.ldkv[i] = d->ldkv[i];
    p.mask[i] = d->mask[i];
    p.dkv[i] = (char*)d->dkv[i];
    if (bwd && !d->dkv[i]) return CMB_ERR_BAD_ARG;
  }
  if (bwd && (!d->dout || !d->dq || !d->lse)) return CMB_ERR_BAD_ARG;
  p.scale = 1.0f / sqrtf((float)d->hd);
  return CMB_OK;
}

template <typename T, bool BWD>
int launch(const SvaParams& p, hipStream_t s) {
  const int C = p.heads * p.hd;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  if (nq == 0) return CMB_OK;
  int64_t blocks = (nq + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  const int lph = p.hd / 8;
#define SVA_LAUNCH(NV, LPH)                                                                             \
  do {                                                                                                  \
    if (BWD) hipLaunchKernelGGL((sva_bwd_kernel<T, NV, LPH>), dim3((unsigned)blocks), dim3(256), 0, s, p); \
    else hipLaunchKernelGGL((sva_fwd_kernel<T, NV, LPH>), dim3((unsigned)blocks), dim3(256), 0, s, p);  \
  } while (0)
  if (C == 1024 && lph == 8) SVA_LAUNCH(2, 8);
  else if (C == 512 && lph == 8) SVA_LAUNCH(1, 8);
  else if (C == 512 && lph == 4) SVA_LAUNCH(1, 4);
  else if (C == 1024 && lph == 16) SVA_LAUNCH(2, 16);
  else return CMB_ERR_SHAPE;
#undef SVA_LAUNCH
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace

extern "C" int cmb_sva_attn_fwd(const cmb_sva_desc* d, void* stream) {
  SvaParams p;
  int rc = fill_params(d, p, false);
  if (rc != CMB_OK) return rc;
  if (d->dtype == CMB_BF16) return launch<bf16_t, false>(p, (hipStream_t)stream);
  if (d->dtype == CMB_F32) return launch<float, false>(p, (hipStream_t)stream);
  return CMB_ERR_BAD_ARG;
}

extern "C" int cmb_sva_attn_bwd(const cmb_sva_desc* d, void* stream) {
  SvaParams p;
  int rc = fill_params(d, p, true);
  if (rc != CMB_OK) return rc;
  if (d->dtype == CMB_BF16) return launch<bf16_t, true>(p, (hipStream_t)stream);
  if (d->dtype == CMB_F32) return launch<float, true>(p, (hipStream_t)stream);
  return CMB_ERR_BAD_ARG;
}
