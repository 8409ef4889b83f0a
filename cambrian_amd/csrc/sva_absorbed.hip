// sva_absorbed.hip — SVA cross-attention core with the WINDOWED tower's K / V projections absorbed into the query side
// (forward + backward, bf16, gfx950).  Round 3; DESIGN.md §4.5.
//
// Reference semantics: MultiKVCrossAttention.forward, vision_sampler.py:187-230 — for every query one softmax over the
// keys of all towers (s_i x s_i window of tower i under the query's cell), 16 heads x 64.  For a tower with an s x s
// window every token is seen by exactly ONE query, yet the reference (and sva_attn.hip's direct form) projects K and V for
// every token: 2 x 9216 x 1024 x 1024 MACs per image and layer for ConvNeXt's 96 x 96 grid.  With xh_t the normalised
// token (LayerNorm affines folded into W_k, W_v, b_k, b_v: vision_sampler.py:173-174,188-189)
//     score[q,h,t] = scale * ( xh_t . U[q,h,:] + cb[q,h] ),   U[q,h,:] = W_k,h^T q_h,   cb[q,h] = b_k,h . q_h
//     o[q,h,:]     = W_v,h Xb[q,h,:] + m3[q,h] b_v,h,         Xb[q,h,:] = sum_t p[q,h,t] xh_t,   m3 = sum_t p[q,h,t]
// the projections act on 576 x 16 (query, head) rows per image instead of 2 x 9216 token rows: U and W_v Xb are sixteen-way
// batched GEMMs with K or N = 64 (cmb_gemm, `batch`), 4.6x fewer FLOPs for identical results, and neither K|V nor dK|dV of
// that tower exist.  This file is the part in between: per query, scores of the absorbed tower's tokens against U, the joint
// softmax with the directly projected towers' keys (their K|V rows as in sva_attn.hip), Xb / m3 / the direct towers' part of
// the output; and the backward of exactly that.
//
// Roofline class: HBM (three [Bq, 16, 1024] bf16 tensors — U, Xb, and the tower's xh — read or written once or twice).
// One wave owns one query.  A query's window is 16 tokens, the layer has 16 heads: every product of the query is a
// 16 x 16 output tile over 1024 channels, i.e. MFMA-shaped with nothing to pad:
//     S[t][h]      = sum_c X[t][c] U[h][c]          32 x v_mfma_f32_16x16x32_bf16   (A: X rows from LDS, B: U rows from HBM)
//     Xb^T[c][h]   = sum_t X^T[c][t] P[t][h]        64 x v_mfma_f32_16x16x16_bf16   (A: ds_read_b64_tr_b16 of X, B: P in registers)
//     dP[t][h]     = sum_c X[t][c] dXb[h][c]        as S
//     dU^T[c][h]   = sum_t X^T[c][t] dS[t][h]       as Xb
//     dX^T[c][t]   = sum_k W^T[c][k] Coef[k][t]     64 x 16x16x32, k = (dXb heads, U heads), Coef = (P^T; dS^T)
// The window lands in LDS by LDS-DMA (2 KiB per token, no registers); the MFMA output layout (lane = column, 4 rows per
// lane quad) of S IS the B-operand layout of the 16x16x16 product, so P and dS never leave the registers between the
// softmax and the token mix.  ds_read_b64_tr_b16 hands each lane 4 tokens (or heads) of ONE channel out of a row-major
// image; its four 8-byte pieces per row are aimed at channels 8p + 4T + e so that the two tiles T = 0, 1 of a 32-channel
// group leave a lane with 8 consecutive channels: 16-byte stores.  Measured at 16 images (9216 queries): forward 209 us,
// backward 373 us stand-alone = 5.1 / 4.8 TB/s over the 1069 / 1806 MB the kernels touch (PMC: every byte once;
// profiles/r03_hbm_kernels_table.md).  (The v_dot2c / v_fma forms of these products were VALU-bound at one wave per SIMD:
// 372 us forward and 646 us backward.)
#include "common.h"
#include "sva_abs_layout.h"

namespace {

constexpr int kHeads = 16, kHd = 64, kC = 1024, kMaxKeys = 16;
constexpr int kWinBytes = kMaxKeys * kC * 2;                       // one wave's token window in LDS: 32 KiB
constexpr int kCoefBytes = kMaxKeys * 32 * 2;                      // backward: (P | dS)[t][32] bf16 per wave
constexpr int kSmemFwd = kWinBytes;                  // one wave per workgroup, four workgroups per CU (LDS, and > 256 registers)
constexpr int kSmemBwd = kWinBytes + kCoefBytes;
constexpr int kMaxD = 4;      // directly projected towers (one key each) beside the absorbed one
constexpr int kPStride = 20;  // P row of a (query, head): [0, 4) the direct towers' keys, [4, 20) the absorbed tokens

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;

struct AbsParams {
  int B, qside, ntowers, window_major;
  const bf16_t* q; int64_t ldq;
  const bf16_t* kv[kMaxD]; int64_t ldkv[kMaxD];
  const uint8_t* mask[kMaxD];
  int ra;
  const bf16_t* xhat; int64_t ldx;
  const uint8_t* mask_a;
  const bf16_t* U;
  const float* bk;   // b_k [1024] (fp32): cb[q,h] = b_k,h . q_h is formed here
  const float* bv;   // b_v [1024] (fp32): the output carries m3[q,h] b_v,h
  bf16_t* out; int64_t ldo;
  bf16_t* xbar;
  float* m3;
  float* P;
  // backward
  const bf16_t* dout; int64_t lddo;
  const bf16_t* dxbar;
  bf16_t* dq; int64_t lddq;
  bf16_t* dkv[kMaxD];
  bf16_t* dU;
  float* dcb;
  bf16_t* dxhat; int64_t lddx;
  float scale;
};

// reductions over the four lanes that share a head (lane & 15): lane ^ 16, lane ^ 32
__device__ __forceinline__ float qsum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float qmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ s16x4_t pack4_bf16(float a, float b, float c, float d) {
  typedef uint32_t u32x2_v __attribute__((ext_vector_type(2)));
  const u32x2_v w = {pack_bf16(a, b), pack_bf16(c, d)};
  return __builtin_bit_cast(s16x4_t, w);
}

__device__ __forceinline__ int64_t token_row(const AbsParams& p, int t, int qy, int qx, int r, int j) {
  const int ry = j / r, rx = j - ry * r, G = p.qside * r;
  return p.window_major ? ((int64_t)t * r * r + j) : ((int64_t)(qy * r + ry) * G + (qx * r + rx));
}

// LDS images and their slot rotation: sva_abs_layout.h (simulated on the host by tests/csrc/sva_abs_layout_sim.cpp).
//
// the query's window: token j's 1024 channels -> xs[j * 2048 ...] by two 1 KiB LDS-DMA instructions; asynchronous —
// covered by the s_waitcnt vmcnt(0) in front of the first read
__device__ __forceinline__ void stage_window(const AbsParams& p, const bf16_t* xb, char* xs, int na, int t, int qy, int qx,
                                             int lane) {
  for (int j = 0; j < na; ++j) {
    const bf16_t* xr = xb + token_row(p, t, qy, qx, p.ra, j) * p.ldx + abs_win_src_slot(lane, j) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xr,
                                     (__attribute__((address_space(3))) void*)(xs + j * 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xr + 512),
                                     (__attribute__((address_space(3))) void*)(xs + j * 2048 + 1024), 16, 0, 0);
  }
}
// window rows [na, 16) of a smaller window: zeros (their probabilities are zero, but 0 x stale LDS bits may be NaN)
__device__ __forceinline__ void zero_tail_rows(char* xs, int na, int lane) {
  const u32x4_t z = {0u, 0u, 0u, 0u};
  for (int off = na * 2048 + lane * 16; off < kWinBytes; off += 1024) *reinterpret_cast<u32x4_t*>(xs + off) = z;
}

__device__ __forceinline__ void cvt8(const bf16x8_t& x, float* f) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)x[e];
}
struct Raw16 { bf16x8_t lo, hi; };   // 16 consecutive channels as loaded
__device__ __forceinline__ Raw16 load16(const bf16_t* p) {
  Raw16 r;
  r.lo = *reinterpret_cast<const bf16x8_t*>(p);
  r.hi = *reinterpret_cast<const bf16x8_t*>(p + 8);
  return r;
}
__device__ __forceinline__ void cvt16(const Raw16& r, float (&f)[16]) {
  cvt8(r.lo, f);
  cvt8(r.hi, f + 8);
}
__device__ __forceinline__ void store16f(bf16_t* p, const float (&f)[16]) {
  *reinterpret_cast<bf16x8_t*>(p) = cvt8_bf16(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
  *reinterpret_cast<bf16x8_t*>(p + 8) = cvt8_bf16(f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15]);
}

// acc[t][h] += sum_c X[t][c] Y[h][c] over the 1024 channels: lane (i = lane & 15, qd = lane >> 4) feeds row i of X (LDS)
// and row i of Y (registers yv[s]: channels [32 s + 8 qd, + 8)) to 32 MFMAs on four independent accumulators
__device__ __forceinline__ f32x4_t rows_dot(const char* xs, const bf16x8_t (&yv)[32], int i, int qd) {
  f32x4_t acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const bf16x8_t xa = *reinterpret_cast<const bf16x8_t*>(xs + abs_rows_off(i, qd, s));
    acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, yv[s], acc[s & 3], 0, 0, 0);
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// out[h][c] = sum_t coef[t][h] X[t][c] for all 1024 channels (Xb from P, dU from dS): per 32-channel group two
// transposing LDS reads + two 16x16x16 MFMAs; lane (i, qd) ends with head i, channels [32 cg + 8 qd, + 8): one 16-byte store
__device__ __forceinline__ void token_mix(const char* xs, s16x4_t coef, bf16_t* orow, int i, int qd) {
  // transposing read: this lane supplies row 4 qd + i / 4 (token), piece i % 4 (channels 8 p + 4 T + e of the group)
  bf16_t* op = orow + i * kC + qd * 8;
#pragma unroll 4
  for (int cg = 0; cg < 32; ++cg) {
    const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(xs + abs_mix_off(i, qd, cg, 0)));
    const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(xs + abs_mix_off(i, qd, cg, 1)));
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, coef, z, 0, 0, 0);
    const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, coef, z, 0, 0, 0);
    *reinterpret_cast<bf16x8_t*>(op + cg * 32) = cvt8_bf16(d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) sva_abs_fwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, i = lane & 15, qd = lane >> 4;
  char* xs = smem;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int na = p.ra * p.ra;
  if (na < kMaxKeys) zero_tail_rows(xs, na, lane);
  for (int64_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    const int G = p.qside * p.ra;
    // ---- everything the query reads from HBM is requested here, in one round trip
    stage_window(p, p.xhat + (int64_t)b * G * G * p.ldx, xs, na, t, qy, qx, lane);
    // U[h = i][32 s + 8 qd ...]: the B operands of the score product
    bf16x8_t u[32];
    {
      const bf16_t* ur = p.U + qi * (int64_t)(kHeads * kC) + i * kC + qd * 8;
#pragma unroll
      for (int s = 0; s < 32; ++s) u[s] = *reinterpret_cast<const bf16x8_t*>(ur + s * 32);
    }
    // the directly projected towers (one key each, r == 1): lane (i, qd) owns channels [64 i + 16 qd, + 16) of head i
    const int ch = i * kHd + qd * 16;
    const Raw16 qraw = load16(p.q + qi * p.ldq + ch);
    Raw16 kraw[kMaxD], vraw[kMaxD];
    bool live[kMaxD];
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) {
      live[d] = d < p.ntowers && !(p.mask[d] && p.mask[d][qi] == 0);
      if (live[d]) {
        const bf16_t* kr = p.kv[d] + qi * p.ldkv[d] + ch;   // r == 1: the query's own token, same index in both layouts
        kraw[d] = load16(kr);
        vraw[d] = load16(kr + kC);
      }
    }
    float bkf[16], bvf[16];   // this lane's 16 channels of b_k, b_v (L2-resident)
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p.bk + ch + e), c = *reinterpret_cast<const f32x4_t*>(p.bv + ch + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) { bkf[e + j] = a[j]; bvf[e + j] = c[j]; }
    }
    uint32_t valid = 0;   // bit r: token 4 qd + r exists and may be attended
    {
      const uint8_t* mka = p.mask_a ? p.mask_a + qi * na : nullptr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tk = 4 * qd + r;
        if (tk < na && !(mka && mka[tk] == 0)) valid |= 1u << r;
      }
    }
    float sd[kMaxD], cbh;
    {
      float qv[16];
      cvt16(qraw, qv);
      float cacc = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) cacc += qv[e] * bkf[e];
      cbh = qsum(cacc);   // b_k,h . q_h
#pragma unroll
      for (int d = 0; d < kMaxD; ++d) {
        sd[d] = -INFINITY;
        if (live[d]) {
          float kf[16];
          cvt16(kraw[d], kf);
          float acc = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc += qv[e] * kf[e];
          sd[d] = qsum(acc) * p.scale;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window has landed
    // ---- scores, joint softmax per head: lane (i, qd) holds tokens 4 qd + r of head i
    const f32x4_t sacc = rows_dot(xs, u, i, qd);
    float sc[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] = (valid >> r & 1u) ? (sacc[r] + cbh) * p.scale : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = qmax(mx);
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) mx = fmaxf(mx, sd[d]);
    float l = 0.f, pd[kMaxD];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = cmb_exp(sc[r] - mx); l += sc[r]; }   // exp(-inf) = 0
    l = qsum(l);
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) { pd[d] = cmb_exp(sd[d] - mx); l += pd[d]; }
    const float inv = 1.0f / l;
    float m3v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] *= inv; m3v += sc[r]; }
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) pd[d] *= inv;
    m3v = qsum(m3v);
    {
      float* prow = p.P + (qi * kHeads + i) * kPStride;
      *reinterpret_cast<f32x4_t*>(prow + 4 + 4 * qd) = f32x4_t{sc[0], sc[1], sc[2], sc[3]};
      if (qd == 0) {
        *reinterpret_cast<f32x4_t*>(prow) = f32x4_t{pd[0], pd[1], pd[2], pd[3]};
        p.m3[qi * kHeads + i] = m3v;
      }
    }
    // ---- the direct towers' part of the output, sum_k p_k V_k, plus the absorbed tower's bias term m3 b_v
    {
      float o[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = m3v * bvf[e];
#pragma unroll
      for (int d = 0; d < kMaxD; ++d) {
        if (live[d]) {
          float vf[16];
          cvt16(vraw[d], vf);
#pragma unroll
          for (int e = 0; e < 16; ++e) o[e] += pd[d] * vf[e];
        }
      }
      store16f(p.out + qi * p.ldo + ch, o);
    }
    // ---- Xb[h][c] = sum_t p[t][h] xh_t[c]: the probabilities are already the B operand
    token_mix(xs, pack4_bf16(sc[0], sc[1], sc[2], sc[3]), p.xbar + qi * (int64_t)(kHeads * kC), i, qd);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this query is done before the next window's DMA
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) sva_abs_bwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, i = lane & 15, qd = lane >> 4;
  char* xs = smem;
  bf16_t* coef = reinterpret_cast<bf16_t*>(smem + kWinBytes);   // [t][32]: P[t][h], dS[t][h]
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int na = p.ra * p.ra;
  for (int64_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    const int G = p.qside * p.ra;
    const int64_t xbase = (int64_t)b * G * G;
    if (na < kMaxKeys) zero_tail_rows(xs, na, lane);   // (the window region is also the dX pass's operand stage)
    // ---- everything the query reads from HBM is requested here, in one round trip
    stage_window(p, p.xhat + xbase * p.ldx, xs, na, t, qy, qx, lane);
    // dXb[h = i][32 s + 8 qd ...] and U[h = i][...]: B operands of the dP product, then the rows of W = (dXb; U)
    bf16x8_t dxv[32], uv[32];
    {
      const bf16_t* dp = p.dxbar + qi * (int64_t)(kHeads * kC) + i * kC + qd * 8;
      const bf16_t* up = p.U + qi * (int64_t)(kHeads * kC) + i * kC + qd * 8;
#pragma unroll
      for (int s = 0; s < 32; ++s) dxv[s] = *reinterpret_cast<const bf16x8_t*>(dp + s * 32);
#pragma unroll
      for (int s = 0; s < 32; ++s) uv[s] = *reinterpret_cast<const bf16x8_t*>(up + s * 32);
    }
    const int ch = i * kHd + qd * 16;
    const Raw16 qraw = load16(p.q + qi * p.ldq + ch), doraw = load16(p.dout + qi * p.lddo + ch);
    Raw16 kraw[kMaxD], vraw[kMaxD];
    bool live[kMaxD];
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) {
      live[d] = d < p.ntowers && !(p.mask[d] && p.mask[d][qi] == 0);
      if (live[d]) {
        const bf16_t* kr = p.kv[d] + qi * p.ldkv[d] + ch;
        kraw[d] = load16(kr);
        vraw[d] = load16(kr + kC);
      }
    }
    const float* prow = p.P + (qi * kHeads + i) * kPStride;
    const f32x4_t pa = *reinterpret_cast<const f32x4_t*>(prow + 4 + 4 * qd);
    const f32x4_t pdv = *reinterpret_cast<const f32x4_t*>(prow);
    float qv[16], dov[16], bkf[16];
    cvt16(qraw, qv);
    cvt16(doraw, dov);
    float dm3h;   // d m3[h] = b_v,h . do_h
    {
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p.bk + ch + e), c = *reinterpret_cast<const f32x4_t*>(p.bv + ch + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bkf[e + j] = a[j]; acc += dov[e + j] * c[j]; }
      }
      dm3h = qsum(acc);
    }
    // ---- dP of every key; D = sum_k P_k dP_k per head.  Direct towers: dP = do . V
    float pd[kMaxD], dsd[kMaxD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) {
      pd[d] = dsd[d] = 0.f;
      if (live[d]) {
        pd[d] = pdv[d];
        float vf[16];
        cvt16(vraw[d], vf);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += dov[e] * vf[e];
        dsd[d] = qsum(acc);
        D += pd[d] * dsd[d];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window has landed
    const f32x4_t dpa = rows_dot(xs, dxv, i, qd);        // dP[t][h] - dm3[h]  (a masked or absent token has P = 0)
    {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) part += pa[r] * (dpa[r] + dm3h);
      D += qsum(part);
    }
    // ---- dS (scale included); d(cb)
    float ds[4], dcb = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ds[r] = pa[r] * (dpa[r] + dm3h - D) * p.scale; dcb += ds[r]; }
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) dsd[d] = pd[d] * (dsd[d] - D) * p.scale;
    dcb = qsum(dcb);
    if (qd == 0) p.dcb[qi * kHeads + i] = dcb;
    // (P | dS)[t][k] for the dX product: this lane's tokens 4 qd + r, columns i and 16 + i
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      coef[(4 * qd + r) * 32 + i] = (bf16_t)pa[r];
      coef[(4 * qd + r) * 32 + 16 + i] = (bf16_t)ds[r];
    }
    // ---- dq = d(cb) b_k  +  over the direct towers dS K;  dK = dS q, dV = P do
    {
      float dq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = dcb * bkf[e];
#pragma unroll
      for (int d = 0; d < kMaxD; ++d) {
        if (d < p.ntowers) {
          bf16_t* dkr = p.dkv[d] + qi * p.ldkv[d] + ch;
          float dk[16], dv[16];
          if (!live[d]) {
#pragma unroll
            for (int e = 0; e < 16; ++e) dk[e] = dv[e] = 0.f;
          } else {
            float kf[16];
            cvt16(kraw[d], kf);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              dq[e] += dsd[d] * kf[e];
              dk[e] = dsd[d] * qv[e];
              dv[e] = pd[d] * dov[e];
            }
          }
          store16f(dkr, dk);
          store16f(dkr + kC, dv);
        }
      }
      store16f(p.dq + qi * p.lddq + ch, dq);
    }
    // ---- dU[h][c] = sum_t dS[t][h] xh_t[c]
    token_mix(xs, pack4_bf16(ds[0], ds[1], ds[2], ds[3]), p.dU + qi * (int64_t)(kHeads * kC), i, qd);
    // ---- d(xh_t)[c] = sum_h P[t][h] dXb[h][c] + dS[t][h] U[h][c]: one 512-channel half of W = (dXb; U) [32][512] at a time
    //      in the window's LDS region, written from the registers that fed the dP product (dXb) and were loaded beside
    //      them (U): no second trip to memory.  W^T by transposing reads, Coef^T from the scratch.
    {
      const bf16x8_t cf = *reinterpret_cast<const bf16x8_t*>(coef + i * 32 + qd * 8);   // Coef[k = 8 qd + j][t = i]
      // transposing read: rows 8 qd + i / 4 and + 4 of W (k = 8 qd + j), piece i % 4; write: rows i (dXb) and 16 + i (U)
      bf16_t* dxrow = p.dxhat + (xbase + token_row(p, t, qy, qx, p.ra, i < na ? i : 0)) * p.lddx + qd * 8;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          *reinterpret_cast<bf16x8_t*>(xs + abs_w_write_off(i, qd, s)) = dxv[half * 16 + s];
          *reinterpret_cast<bf16x8_t*>(xs + abs_w_write_off(16 + i, qd, s)) = uv[half * 16 + s];
        }
#pragma unroll 2
        for (int cg = 0; cg < 16; ++cg) {
          f32x4_t dd[2];
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(xs + abs_w_read_off(i, qd, 0, cg, T)));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(xs + abs_w_read_off(i, qd, 1, cg, T)));
            const s16x8_t a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            dd[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), cf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0,
                                                            0, 0);
          }
          if (i < na)
            *reinterpret_cast<bf16x8_t*>(dxrow + half * 512 + cg * 32) =
                cvt8_bf16(dd[0][0], dd[0][1], dd[0][2], dd[0][3], dd[1][0], dd[1][1], dd[1][2], dd[1][3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS access of this query is done before the next window's DMA
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The exact instantiation (round 4, VERDICT r3 next #3b): the SAME absorbed algorithm — U and Xb as [Bq, 16, 1024]
// intermediates, one joint softmax over the direct towers' keys and the window's tokens, P / m3 / d(cb) side outputs, dU and
// d(xhat) from the coefficient products — in plain fp32 arithmetic on operands of type T (float: the fp32 parity path, so
// that an fp32 model runs the algorithm the bf16 bench line runs, tests to 1e-4 / 5e-4 like every other kernel; bf16_t:
// the MFMA kernels above against it on identical operands, CMB_KNOB_SVA_ABS = 1).  One workgroup of 256 threads per query,
// thread (t = tid / 16, h = tid % 16) owns token t of head h; the window and U sit in LDS as fp32 rows of 1025 floats
// (odd stride: the 16 rows a wave reads at one channel fall on different banks); d(Xb) is read from memory where needed.
// No MFMA, no rounding of probabilities: test speed only (release geometry: ~1 ms forward).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSimpleLd = kC + 1;
constexpr int kSmemSimple = (2 * kMaxKeys * kSimpleLd + 6 * kMaxKeys * kHeads + 2 * kMaxD * kHeads + 2 * kHeads) * 4;

template <typename T>
__device__ __forceinline__ void simple_stage(const AbsParams& p, const T* xb, const T* urow, float* sx, float* su, int na, int t0,
                                             int qy, int qx, int tid) {
  for (int idx = tid; idx < kMaxKeys * kC; idx += 256) {
    const int j = idx >> 10, c = idx & (kC - 1);
    sx[j * kSimpleLd + c] = j < na ? (float)xb[token_row(p, t0, qy, qx, p.ra, j) * p.ldx + c] : 0.f;
    su[j * kSimpleLd + c] = (float)urow[idx];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) sva_abs_simple_fwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);            // X[t][c]
  float* su = sx + kMaxKeys * kSimpleLd;                  // U[h][c]
  float* sS = su + kMaxKeys * kSimpleLd;                  // scores / probabilities [t][h]
  float* sD = sS + 6 * kMaxKeys * kHeads;                 // direct towers [d][h]
  float* sM = sD + 2 * kMaxD * kHeads;                    // m3[h]
  const int tid = threadIdx.x, t = tid >> 4, h = tid & 15;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int na = p.ra * p.ra;
  for (int64_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t0 = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t0 / p.qside, qx = t0 - qy * p.qside, G = p.qside * p.ra;
    const T* xb = reinterpret_cast<const T*>(p.xhat) + (int64_t)b * G * G * p.ldx;
    const T* qrow = reinterpret_cast<const T*>(p.q) + qi * p.ldq;
    __syncthreads();
    simple_stage<T>(p, xb, reinterpret_cast<const T*>(p.U) + qi * (int64_t)(kHeads * kC), sx, su, na, t0, qy, qx, tid);
    __syncthreads();
    float cbh = 0.f;
    for (int e = 0; e < kHd; ++e) cbh += (float)qrow[h * kHd + e] * p.bk[h * kHd + e];
    const bool ok = t < na && !(p.mask_a && p.mask_a[qi * na + t] == 0);
    float sc = -INFINITY;
    if (ok) {
      float acc = 0.f;
      for (int c = 0; c < kC; ++c) acc += sx[t * kSimpleLd + c] * su[h * kSimpleLd + c];
      sc = (acc + cbh) * p.scale;
    }
    sS[t * kHeads + h] = sc;
    if (t < kMaxD) {
      float sd = -INFINITY;
      if (t < p.ntowers && !(p.mask[t] && p.mask[t][qi] == 0)) {
        const T* kr = reinterpret_cast<const T*>(p.kv[t]) + qi * p.ldkv[t] + h * kHd;
        float acc = 0.f;
        for (int e = 0; e < kHd; ++e) acc += (float)qrow[h * kHd + e] * (float)kr[e];
        sd = acc * p.scale;
      }
      sD[t * kHeads + h] = sd;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = 0; k < kMaxKeys; ++k) mx = fmaxf(mx, sS[k * kHeads + h]);
    for (int d = 0; d < kMaxD; ++d) mx = fmaxf(mx, sD[d * kHeads + h]);
    float l = 0.f;
    for (int k = 0; k < kMaxKeys; ++k) l += expf(sS[k * kHeads + h] - mx);
    for (int d = 0; d < kMaxD; ++d) l += expf(sD[d * kHeads + h] - mx);
    const float pa = expf(sc - mx) / l;
    const float pdv = t < kMaxD ? expf(sD[t * kHeads + h] - mx) / l : 0.f;
    __syncthreads();
    sS[t * kHeads + h] = pa;
    if (t < kMaxD) sD[t * kHeads + h] = pdv;
    float* prow = p.P + (qi * kHeads + h) * kPStride;
    prow[4 + t] = pa;
    if (t < kMaxD) prow[t] = pdv;
    __syncthreads();
    if (t == 0) {
      float m3v = 0.f;
      for (int k = 0; k < kMaxKeys; ++k) m3v += sS[k * kHeads + h];
      sM[h] = m3v;
      p.m3[qi * kHeads + h] = m3v;
    }
    __syncthreads();
    T* orow = reinterpret_cast<T*>(p.out) + qi * p.ldo;
    for (int c = tid; c < kC; c += 256) {
      const int hh = c >> 6;
      float o = sM[hh] * p.bv[c];
      for (int d = 0; d < p.ntowers; ++d)
        if (sD[d * kHeads + hh] != 0.f) o += sD[d * kHeads + hh] * (float)(reinterpret_cast<const T*>(p.kv[d]) + qi * p.ldkv[d])[kC + c];
      orow[c] = (T)o;
    }
    T* xbar = reinterpret_cast<T*>(p.xbar) + qi * (int64_t)(kHeads * kC);
    for (int idx = tid; idx < kHeads * kC; idx += 256) {
      const int hh = idx >> 10, c = idx & (kC - 1);
      float acc = 0.f;
      for (int k = 0; k < kMaxKeys; ++k) acc += sS[k * kHeads + hh] * sx[k * kSimpleLd + c];
      xbar[idx] = (T)acc;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) sva_abs_simple_bwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);
  float* su = sx + kMaxKeys * kSimpleLd;
  float* sP = su + kMaxKeys * kSimpleLd;                  // P[t][h]
  float* sDS = sP + kMaxKeys * kHeads;                    // dS[t][h]
  float* sT = sDS + kMaxKeys * kHeads;                    // P dP partial sums [t][h] (+ direct towers behind: [16 + d][h])
  float* sPd = sP + 6 * kMaxKeys * kHeads;                // direct towers: P[d][h], then dS[d][h]
  float* sDSd = sPd + kMaxD * kHeads;
  float* sM = sDSd + kMaxD * kHeads;                      // d(cb)[h]
  const int tid = threadIdx.x, t = tid >> 4, h = tid & 15;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int na = p.ra * p.ra;
  for (int64_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t0 = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t0 / p.qside, qx = t0 - qy * p.qside, G = p.qside * p.ra;
    const int64_t xbase = (int64_t)b * G * G;
    const T* xb = reinterpret_cast<const T*>(p.xhat) + xbase * p.ldx;
    const T* qrow = reinterpret_cast<const T*>(p.q) + qi * p.ldq;
    const T* dorow = reinterpret_cast<const T*>(p.dout) + qi * p.lddo;
    const T* dxb = reinterpret_cast<const T*>(p.dxbar) + qi * (int64_t)(kHeads * kC);
    __syncthreads();
    simple_stage<T>(p, xb, reinterpret_cast<const T*>(p.U) + qi * (int64_t)(kHeads * kC), sx, su, na, t0, qy, qx, tid);
    __syncthreads();
    float dm3h = 0.f;
    for (int e = 0; e < kHd; ++e) dm3h += p.bv[h * kHd + e] * (float)dorow[h * kHd + e];
    const float* prow = p.P + (qi * kHeads + h) * kPStride;
    const float pa = prow[4 + t];
    float dpa = 0.f;
    if (pa != 0.f) {
      float acc = 0.f;
      for (int c = 0; c < kC; ++c) acc += sx[t * kSimpleLd + c] * (float)dxb[h * kC + c];
      dpa = acc + dm3h;
    }
    float pdv = 0.f, dpd = 0.f;
    const bool live = t < p.ntowers && !(p.mask[t < kMaxD ? t : 0] && p.mask[t < kMaxD ? t : 0][qi] == 0);
    if (t < kMaxD && live) {
      pdv = prow[t];
      const T* vr = reinterpret_cast<const T*>(p.kv[t]) + qi * p.ldkv[t] + kC + h * kHd;
      for (int e = 0; e < kHd; ++e) dpd += (float)dorow[h * kHd + e] * (float)vr[e];
    }
    sT[t * kHeads + h] = pa * dpa + pdv * dpd;
    __syncthreads();
    float D = 0.f;
    for (int k = 0; k < kMaxKeys; ++k) D += sT[k * kHeads + h];
    const float ds = pa * (dpa - D) * p.scale;
    sP[t * kHeads + h] = pa;
    sDS[t * kHeads + h] = ds;
    if (t < kMaxD) {
      sPd[t * kHeads + h] = pdv;
      sDSd[t * kHeads + h] = pdv * (dpd - D) * p.scale;
    }
    __syncthreads();
    if (t == 0) {
      float dcb = 0.f;
      for (int k = 0; k < kMaxKeys; ++k) dcb += sDS[k * kHeads + h];
      sM[h] = dcb;
      p.dcb[qi * kHeads + h] = dcb;
    }
    __syncthreads();
    T* dqrow = reinterpret_cast<T*>(p.dq) + qi * p.lddq;
    for (int c = tid; c < kC; c += 256) {
      const int hh = c >> 6;
      float dq = sM[hh] * p.bk[c];
      for (int d = 0; d < p.ntowers; ++d) {
        const bool lv = !(p.mask[d] && p.mask[d][qi] == 0);
        const T* kr = reinterpret_cast<const T*>(p.kv[d]) + qi * p.ldkv[d];
        T* dkr = reinterpret_cast<T*>(p.dkv[d]) + qi * p.ldkv[d];
        float dk = 0.f, dv = 0.f;
        if (lv) {
          dq += sDSd[d * kHeads + hh] * (float)kr[c];
          dk = sDSd[d * kHeads + hh] * (float)qrow[c];
          dv = sPd[d * kHeads + hh] * (float)dorow[c];
        }
        dkr[c] = (T)dk;
        dkr[kC + c] = (T)dv;
      }
      dqrow[c] = (T)dq;
    }
    T* dU = reinterpret_cast<T*>(p.dU) + qi * (int64_t)(kHeads * kC);
    for (int idx = tid; idx < kHeads * kC; idx += 256) {
      const int hh = idx >> 10, c = idx & (kC - 1);
      float acc = 0.f;
      for (int k = 0; k < kMaxKeys; ++k) acc += sDS[k * kHeads + hh] * sx[k * kSimpleLd + c];
      dU[idx] = (T)acc;
    }
    T* dxh = reinterpret_cast<T*>(p.dxhat);
    for (int idx = tid; idx < na * kC; idx += 256) {
      const int j = idx >> 10, c = idx & (kC - 1);
      float acc = 0.f;
      for (int hh = 0; hh < kHeads; ++hh)
        acc += sP[j * kHeads + hh] * (float)dxb[hh * kC + c] + sDS[j * kHeads + hh] * su[hh * kSimpleLd + c];
      dxh[(xbase + token_row(p, t0, qy, qx, p.ra, j)) * p.lddx + c] = (T)acc;
    }
  }
}

template <typename T>
int launch_simple(const AbsParams& p, bool bwd, hipStream_t s) {
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int64_t blocks = nq < 65535 ? nq : 65535;
  auto kf = sva_abs_simple_fwd_kernel<T>;
  auto kb = sva_abs_simple_bwd_kernel<T>;
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemSimple) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemSimple) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  if (bwd) hipLaunchKernelGGL(kb, dim3((unsigned)blocks), dim3(256), kSmemSimple, s, p);
  else hipLaunchKernelGGL(kf, dim3((unsigned)blocks), dim3(256), kSmemSimple, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

int fill(const cmb_sva_abs_desc* d, AbsParams& p, bool bwd) {
  if (!d || !d->q || !d->xhat || !d->U || !d->bk || !d->bv || !d->out || !d->xbar || !d->m3 || !d->P) return CMB_ERR_BAD_ARG;
  if (d->dtype != CMB_BF16 && d->dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  if (d->B < 0 || d->qside <= 0 || d->heads != kHeads || d->hd != kHd) return CMB_ERR_SHAPE;
  if (d->ntowers < 0 || d->ntowers > kMaxD || d->ra <= 0 || d->ra * d->ra > kMaxKeys) return CMB_ERR_SHAPE;
  // 16-byte accesses everywhere: bases 16-byte aligned, row strides multiples of 8 elements
  if (!cmb_aligned16(d->q) || !cmb_aligned16(d->xhat) || !cmb_aligned16(d->U) || !cmb_aligned16(d->out) || !cmb_aligned16(d->xbar) ||
      !cmb_aligned16(d->P) || !cmb_aligned16(d->bk) || !cmb_aligned16(d->bv) || (d->ldq & 7) || (d->ldx & 7) || (d->ldo & 7))
    return CMB_ERR_ALIGNMENT;
  p.B = d->B; p.qside = d->qside; p.ntowers = d->ntowers; p.window_major = d->window_major;
  p.q = (const bf16_t*)d->q; p.ldq = d->ldq;
  for (int i = 0; i < d->ntowers; ++i) {
    if (!d->kv[i]) return CMB_ERR_BAD_ARG;
    if (d->r[i] != 1) return CMB_ERR_SHAPE;   // directly projected towers: one key per query
    if (!cmb_aligned16(d->kv[i]) || (d->ldkv[i] & 7) || (bwd && !cmb_aligned16(d->dkv[i]))) return CMB_ERR_ALIGNMENT;
    p.kv[i] = (const bf16_t*)d->kv[i]; p.ldkv[i] = d->ldkv[i];
    p.mask[i] = d->mask[i];
    p.dkv[i] = (bf16_t*)d->dkv[i];
    if (bwd && !d->dkv[i]) return CMB_ERR_BAD_ARG;
  }
  p.ra = d->ra;
  p.xhat = (const bf16_t*)d->xhat; p.ldx = d->ldx;
  p.mask_a = d->mask_a;
  p.U = (const bf16_t*)d->U; p.bk = d->bk; p.bv = d->bv;
  p.out = (bf16_t*)d->out; p.ldo = d->ldo;
  p.xbar = (bf16_t*)d->xbar; p.m3 = d->m3; p.P = d->P;
  p.dout = (const bf16_t*)d->dout; p.lddo = d->lddo;
  p.dxbar = (const bf16_t*)d->dxbar;
  p.dq = (bf16_t*)d->dq; p.lddq = d->lddq;
  p.dU = (bf16_t*)d->dU; p.dcb = d->dcb;
  p.dxhat = (bf16_t*)d->dxhat; p.lddx = d->lddx;
  if (bwd && d->dout && d->dq && d->dU && d->dxhat && d->dxbar &&
      (!cmb_aligned16(d->dout) || !cmb_aligned16(d->dq) || !cmb_aligned16(d->dU) || !cmb_aligned16(d->dxhat) ||
       !cmb_aligned16(d->dxbar) || (d->lddo & 7) || (d->lddq & 7) || (d->lddx & 7)))
    return CMB_ERR_ALIGNMENT;
  if (bwd && (!d->dout || !d->dxbar || !d->dq || !d->dU || !d->dcb || !d->dxhat)) return CMB_ERR_BAD_ARG;
  p.scale = 1.0f / sqrtf((float)d->hd);
  return CMB_OK;
}

}  // namespace

extern "C" int cmb_sva_abs_fwd(const cmb_sva_abs_desc* d, void* stream) {
  AbsParams p;
  const int rc = fill(d, p, false);
  if (rc != CMB_OK) return rc;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  if (nq == 0) return CMB_OK;
  if (d->dtype == CMB_F32) return launch_simple<float>(p, false, (hipStream_t)stream);
  if (cmb_knob(CMB_KNOB_SVA_ABS) == 1) return launch_simple<bf16_t>(p, false, (hipStream_t)stream);
  const int64_t blocks = nq < (1 << 20) ? nq : (1 << 20);
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sva_abs_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSmemFwd) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  hipLaunchKernelGGL(sva_abs_fwd_kernel, dim3((unsigned)blocks), dim3(64), kSmemFwd, (hipStream_t)stream, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_sva_abs_bwd(const cmb_sva_abs_desc* d, void* stream) {
  AbsParams p;
  const int rc = fill(d, p, true);
  if (rc != CMB_OK) return rc;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  if (nq == 0) return CMB_OK;
  if (d->dtype == CMB_F32) return launch_simple<float>(p, true, (hipStream_t)stream);
  if (cmb_knob(CMB_KNOB_SVA_ABS) == 1) return launch_simple<bf16_t>(p, true, (hipStream_t)stream);
  const int64_t blocks = nq < (1 << 20) ? nq : (1 << 20);
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sva_abs_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSmemBwd) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  hipLaunchKernelGGL(sva_abs_bwd_kernel, dim3((unsigned)blocks), dim3(64), kSmemBwd, (hipStream_t)stream, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
