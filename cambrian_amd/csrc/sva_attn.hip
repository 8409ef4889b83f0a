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

// ---------------------------------------------------------------------------------------------------------------------
// K|V weight folding (vision_sampler.py:173-174,188-189): the K- and V-LayerNorm affines (gamma, beta) of a tower are folded
// into ONE projection so that one x-hat and one GEMM serve both:
//     w[n, :] = W[n, :] * gamma            b[n] = W[n, :] . beta          (rows 0..H-1 from (Wk, gk, bk), H..2H-1 from V)
// and back:  dW = dw * gamma + db (x) beta,   dgamma = colsum(dw o W),   dbeta = W^T db.
// In PyTorch ops this was ~22 small kernels per (layer, tower) and step — 1150 launches, 5-6 ms of GPU time per step for
// the 13 x 4 pairs; here it is two.  Everything fp32 (master parameters); both kernels are deterministic (no atomics).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// one wave per output row
__global__ void __launch_bounds__(256) fold_kv_fwd_kernel(const float* __restrict__ wk, const float* __restrict__ gk,
                                                          const float* __restrict__ bk, const float* __restrict__ wv,
                                                          const float* __restrict__ gv, const float* __restrict__ bv, int H,
                                                          int K, float* __restrict__ w_out, float* __restrict__ b_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * H) return;
  const bool isv = row >= H;
  const float* w = (isv ? wv : wk) + (int64_t)(isv ? row - H : row) * K;
  const float* g = isv ? gv : gk;
  const float* b = isv ? bv : bk;
  float dot = 0.f;
  for (int c = lane * 4; c < K; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(w + c);
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    float4 o = {x.x * gg.x, x.y * gg.y, x.z * gg.z, x.w * gg.w};
    *reinterpret_cast<float4*>(w_out + (int64_t)row * K + c) = o;
    dot += x.x * bb.x + x.y * bb.y + x.z * bb.z + x.w * bb.w;
  }
  dot = wave_sum(dot);
  if (lane == 0) b_out[row] = dot;
}

// Backward, two deterministic passes (round 3; the first version ran 32 workgroups of 4-byte accesses: 119 us for 24 MB):
//   pass 1  grid (K / 256, H / 32, 2 halves): a workgroup owns 32 rows x 256 columns of one half, a wave 8 of those rows,
//           a lane 4 consecutive columns (16-byte accesses); it writes dW for its block and the block's column sums of
//           dw o W (-> dgamma) and db[r] W (-> dbeta) into part[half][slab][2][K] — the four waves' sums meet in LDS in
//           a fixed order;
//   pass 2  one thread per (half, which, column): the H / 32 slab partials summed in slab order.
// No atomics: run-to-run bit-identical.  `part` is caller-owned workspace (cmb_sva_fold_kv_bwd_workspace bytes).
constexpr int kFoldRows = 32;

__global__ void __launch_bounds__(256) fold_kv_bwd_kernel(const float* __restrict__ dw_out, const float* __restrict__ db_out,
                                                          const float* __restrict__ wk, const float* __restrict__ gk,
                                                          const float* __restrict__ bk, const float* __restrict__ wv,
                                                          const float* __restrict__ gv, const float* __restrict__ bv, int H,
                                                          int K, float* __restrict__ dwk, float* __restrict__ dwv,
                                                          float* __restrict__ part, int nslab) {
  __shared__ float4 red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool isv = blockIdx.z != 0;
  const int c = blockIdx.x * 256 + lane * 4;
  const int slab = blockIdx.y;
  const float* w = isv ? wv : wk;
  const float* dwo = dw_out + (isv ? (int64_t)H * K : 0);
  const float* dbo = db_out + (isv ? H : 0);
  float* dw = isv ? dwv : dwk;
  float4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  if (c < K) {
    const float4 g = *reinterpret_cast<const float4*>((isv ? gv : gk) + c);
    const float4 b = *reinterpret_cast<const float4*>((isv ? bv : bk) + c);
    const int r0 = slab * kFoldRows + wave * (kFoldRows / 4);
#pragma unroll
    for (int i = 0; i < kFoldRows / 4; ++i) {
      const int r = r0 + i;
      if (r < H) {
        const float4 x = *reinterpret_cast<const float4*>(w + (int64_t)r * K + c);
        const float4 d = *reinterpret_cast<const float4*>(dwo + (int64_t)r * K + c);
        const float e = dbo[r];
        const float4 o = {d.x * g.x + e * b.x, d.y * g.y + e * b.y, d.z * g.z + e * b.z, d.w * g.w + e * b.w};
        *reinterpret_cast<float4*>(dw + (int64_t)r * K + c) = o;
        sg.x += d.x * x.x; sg.y += d.y * x.y; sg.z += d.z * x.z; sg.w += d.w * x.w;
        sb.x += e * x.x; sb.y += e * x.y; sb.z += e * x.z; sb.w += e * x.w;
      }
    }
  }
  red[0][wave][lane] = sg;
  red[1][wave][lane] = sb;
  __syncthreads();
  if (wave < 2 && c < K) {  // wave 0: dgamma partial, wave 1: dbeta partial
    const float4 a0 = red[wave][0][lane], a1 = red[wave][1][lane], a2 = red[wave][2][lane], a3 = red[wave][3][lane];
    const float4 t = {((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z,
                      ((a0.w + a1.w) + a2.w) + a3.w};
    *reinterpret_cast<float4*>(part + (((int64_t)(isv ? 1 : 0) * nslab + slab) * 2 + wave) * K + c) = t;
  }
}

__global__ void __launch_bounds__(256) fold_kv_bwd_reduce_kernel(const float* __restrict__ part, int nslab, int K,
                                                                 float* __restrict__ dgk, float* __restrict__ dbk,
                                                                 float* __restrict__ dgv, float* __restrict__ dbv) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (half, which, column)
  if (idx >= 4 * K) return;
  const int c = idx % K, which = (idx / K) & 1, half = idx / (2 * K);
  float s = 0.f;
  for (int sl = 0; sl < nslab; ++sl) s += part[(((int64_t)half * nslab + sl) * 2 + which) * K + c];
  (half ? (which ? dbv : dgv) : (which ? dbk : dgk))[c] = s;
}

}  // namespace

extern "C" int cmb_sva_fold_kv_fwd(const float* wk, const float* gk, const float* bk, const float* wv, const float* gv,
                                   const float* bv, int64_t H, int64_t K, float* w_out, float* b_out, void* stream) {
  if (!wk || !gk || !bk || !wv || !gv || !bv || !w_out || !b_out || H <= 0 || K <= 0 || (K & 3)) return CMB_ERR_BAD_ARG;
  hipLaunchKernelGGL(fold_kv_fwd_kernel, dim3((unsigned)((2 * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, wk, gk, bk, wv, gv,
                     bv, (int)H, (int)K, w_out, b_out);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int64_t cmb_sva_fold_kv_bwd_workspace(int64_t H, int64_t K) {
  if (H <= 0 || K <= 0) return 0;
  return 2 * ((H + kFoldRows - 1) / kFoldRows) * 2 * K * (int64_t)sizeof(float);
}

extern "C" int cmb_sva_fold_kv_bwd(const float* dw_out, const float* db_out, const float* wk, const float* gk, const float* bk,
                                   const float* wv, const float* gv, const float* bv, int64_t H, int64_t K, float* dwk,
                                   float* dgk, float* dbk, float* dwv, float* dgv, float* dbv, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (!dw_out || !db_out || !wk || !gk || !bk || !wv || !gv || !bv || !dwk || !dgk || !dbk || !dwv || !dgv || !dbv || H <= 0 ||
      K <= 0 || (K & 3))
    return CMB_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < cmb_sva_fold_kv_bwd_workspace(H, K)) return CMB_ERR_WORKSPACE;
  const int nslab = (int)((H + kFoldRows - 1) / kFoldRows);
  float* part = (float*)workspace;
  hipLaunchKernelGGL(fold_kv_bwd_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)nslab, 2), dim3(256), 0, (hipStream_t)stream,
                     dw_out, db_out, wk, gk, bk, wv, gv, bv, (int)H, (int)K, dwk, dwv, part, nslab);
  CMB_CHECK_LAUNCH();
  hipLaunchKernelGGL(fold_kv_bwd_reduce_kernel, dim3((unsigned)((4 * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     nslab, (int)K, dgk, dbk, dgv, dbv);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
