// sva_absorbed.hip — SVA cross-attention core with the WINDOWED tower's K / V projections absorbed into the query side
// (forward + backward, bf16, gfx950).  Round 3; DESIGN.md §4 "Absorbed K/V".
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
// One wave owns one query; a lane owns channels [8 l, 8 l + 8) and [512 + 8 l, 512 + 8 l + 8) of every 1024-wide row
// (16-byte accesses, a wave-instruction covers 1 KiB contiguous).  The 16 x 16 x 1024 products of a query run on
// v_dot2c_f32_bf16 over packed operands; a 16-value butterfly (lane bits 5, 4, 3 <-> head bits 2, 1, 0) leaves lane group
// g = lane >> 3 with the sums of heads g and g + 8 — the same head ownership sva_attn.hip's per-chunk dot products have.
#include "common.h"

namespace {

constexpr int kHeads = 16, kHd = 64, kC = 1024, kMaxKeys = 16;
constexpr int kWinBytes = kMaxKeys * kC * 2;                       // one wave's token window in LDS: 32 KiB
constexpr int kScratchF32 = 4 * kMaxKeys * kHeads * 4;            // [wave][token][head] fp32: 4 KiB
constexpr int kScratchPair = 4 * (kMaxKeys / 2) * kHeads * 4;     // [wave][token pair][head] bf16x2: 2 KiB
constexpr int kSmemFwd = 4 * kWinBytes + kScratchF32 + kScratchPair;                    // + scores, + P pairs
constexpr int kSmemBwd = 4 * kWinBytes + 2 * kScratchF32 + kScratchF32 + kScratchPair;  // + P, dP, + (P, dS), + dS pairs
constexpr int kMaxD = 4;  // directly projected towers (one key each) beside the absorbed one

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

struct AbsParams {
  int B, qside, ntowers, window_major;
  const bf16_t* q; int64_t ldq;
  const bf16_t* kv[kMaxD]; int64_t ldkv[kMaxD];
  const uint8_t* mask[kMaxD];
  int ra;
  const bf16_t* xhat; int64_t ldx;
  const uint8_t* mask_a;
  const bf16_t* U;
  const float* cb;
  bf16_t* out; int64_t ldo;
  bf16_t* xbar;
  float* m3;
  float* P;
  int nd, nkeys;  // direct keys per query, all keys per query
  // backward
  const bf16_t* dout; int64_t lddo;
  const bf16_t* dxbar;
  const float* dm3;
  bf16_t* dq; int64_t lddq;
  bf16_t* dkv[kMaxD];
  bf16_t* dU;
  float* dcb;
  bf16_t* dxhat; int64_t lddx;
  float scale;
};

// x + (x of the lane `ctrl` names), one v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float add_dpp(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// sum over the 8 lanes of a group (lane ^ 1, lane ^ 2, 7 - lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror)
__device__ __forceinline__ float head_sum8(float v) {
  v = add_dpp<0xB1>(v);
  v = add_dpp<0x4E>(v);
  v = add_dpp<0x141>(v);
  return v;
}

// 8 bf16 . 8 bf16 accumulated into acc (4 x v_dot2c_f32_bf16)
__device__ __forceinline__ float dot8(const bf16x8_t& a, const bf16x8_t& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2_t x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(x, y, acc, false);
  }
  return acc;
}

// packed-pair arithmetic of the token-mix products.  A sum over tokens (or heads) of coefficient x row-element runs two terms
// per v_dot2c_f32_bf16: the row elements of two tokens are interleaved into bf16 pairs (v_perm_b32, once per pair of rows)
// and the two wave-uniform coefficients travel as one bf16 pair read from LDS (broadcast) — no bf16 -> fp32 converts and
// half the multiply instructions of the fp32 FMA form; the coefficients (probabilities, dS) are rounded to bf16, as the P
// operand of every MFMA attention kernel is.
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint32_t pair_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }  // {a.lo, b.lo}
__device__ __forceinline__ uint32_t pair_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }  // {a.hi, b.hi}
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}
// rows a, b (8 bf16 each) -> pr[e] = {a[e], b[e]}
__device__ __forceinline__ void interleave8(const u32x4_t& a, const u32x4_t& b, uint32_t (&pr)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pr[2 * i] = pair_lo(a[i], b[i]);
    pr[2 * i + 1] = pair_hi(a[i], b[i]);
  }
}
// acc[h][e] += pr[e] . c[h] for the 16 heads' coefficient pairs at cp (LDS, wave-uniform address)
__device__ __forceinline__ void mix16(const uint32_t (&pr)[8], const uint32_t* cp, float (&acc)[kHeads][8]) {
#pragma unroll
  for (int hq = 0; hq < 4; ++hq) {
    const u32x4_t c4 = *reinterpret_cast<const u32x4_t*>(cp + hq * 4);
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[hq * 4 + hh][e] = dot2(pr[e], c4[hh], acc[hq * 4 + hh][e]);
  }
}

// the halving exchanges of heads_reduce are gfx950's lane-swap instructions: v_permlane32_swap exchanges lanes 32..63 of
// its first operand with lanes 0..31 of its second (v_permlane16_swap: the odd 16-lane rows of the first with the even
// rows of the second), so first' + second' is, in the lanes whose bit 5 (bit 4) is clear, the first operand summed over
// the lane pair, and in the other lanes the second operand summed over the pair: one swap + one add per pair of heads,
// no select, no ds_bpermute.  (Written as "bit ? v[h1] : v[h0]" the compiler turned the selects of array elements into
// per-lane indexed reads of the 16-element array: 15 v_cmp + 15 v_cndmask each, ~650 instructions per token.)
__device__ __forceinline__ float swap32_add(float first, float second) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(first), __float_as_uint(second), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float first, float second) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(first), __float_as_uint(second), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// v[h], h = 0..15: every lane's partial sum for head h  ->  (a, b) = the wave-wide totals of heads g and g + 8, g = lane >> 3
// (three halving exchanges pair lane bit 5 / 4 / 3 with head bit 2 / 1 / 0, then the sum over the 8 lanes of the group).
__device__ __forceinline__ void heads_reduce(const float (&v)[16], int lane, float& a, float& b) {
  float w8[8], w4[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // i = (h3, h1, h0); partner heads differ in bit 2
    const int h0 = ((i & 4) << 1) | (i & 3), h1 = h0 | 4;
    w8[i] = swap32_add(v[h0], v[h1]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // i = (h3, h0); partner entries of w8 differ in bit 1 of their index
    const int i0 = ((i & 2) << 1) | (i & 1), i1 = i0 | 2;
    w4[i] = swap16_add(w8[i0], w8[i1]);
  }
  // head bit 0 <-> lane bit 3: both candidates summed over the lane pair (row_ror:8 = lane ^ 8 inside a 16-lane row), then
  // one select of the finished values
  const bool b3 = (lane & 8) != 0;
  const float e0 = add_dpp<0x128>(w4[0]), o0 = add_dpp<0x128>(w4[1]);
  const float e1 = add_dpp<0x128>(w4[2]), o1 = add_dpp<0x128>(w4[3]);
  a = head_sum8(b3 ? o0 : e0);
  b = head_sum8(b3 ? o1 : e1);
}

__device__ __forceinline__ int64_t token_row(const AbsParams& p, int t, int qy, int qx, int r, int j) {
  const int ry = j / r, rx = j - ry * r, G = p.qside * r;
  return p.window_major ? ((int64_t)t * r * r + j) : ((int64_t)(qy * r + ry) * G + (qx * r + rx));
}

// the query's window: token j's 1024 channels -> xs[j * 2048 ...] by two 1 KiB LDS-DMA instructions (lane l's 16 bytes of
// each land at + 16 l: exactly the lane's own two chunks); asynchronous — covered by the s_waitcnt vmcnt(0) in front of
// the first read
__device__ __forceinline__ void stage_window(const AbsParams& p, const bf16_t* xb, char* xs, int na, int t, int qy, int qx,
                                             int lane) {
  for (int j = 0; j < na; ++j) {
    const bf16_t* xr = xb + token_row(p, t, qy, qx, p.ra, j) * p.ldx + lane * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xr,
                                     (__attribute__((address_space(3))) void*)(xs + j * 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xr + 512),
                                     (__attribute__((address_space(3))) void*)(xs + j * 2048 + 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ void cvt8(const bf16x8_t& x, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)x[e];
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sva_abs_fwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per wave: the query's token window (16 x 2 KiB, filled by LDS-DMA: no registers, all rows in flight at once) and the
  // absorbed-token scores / probabilities of all heads (broadcast reads)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3;
  char* xs = smem + wave * kWinBytes;
  float (*pw)[kMaxKeys][kHeads] = reinterpret_cast<float (*)[kMaxKeys][kHeads]>(smem + 4 * kWinBytes);
  uint32_t (*pp)[kMaxKeys / 2][kHeads] =
      reinterpret_cast<uint32_t (*)[kMaxKeys / 2][kHeads]>(smem + 4 * kWinBytes + kScratchF32);
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int na = p.ra * p.ra;
  for (int64_t qi = wave_global; qi < nq; qi += nwaves) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    const int c0 = lane * 8, c1 = 512 + lane * 8;
    const int G = p.qside * p.ra;
    const bf16_t* xb = p.xhat + (int64_t)b * G * G * p.ldx;
    const uint8_t* mka = p.mask_a ? p.mask_a + qi * na : nullptr;
    stage_window(p, xb, xs, na, t, qy, qx, lane);

    // ---- scores of the directly projected towers' keys (one key per tower: r_i == 1): sd[i][cc] = head g (cc 0) / g + 8 (cc 1)
    float sd[kMaxD][2];
    {
      const bf16_t* qr = p.q + qi * p.ldq;
      float q0[8], q1[8];
      cvt8(*reinterpret_cast<const bf16x8_t*>(qr + c0), q0);
      cvt8(*reinterpret_cast<const bf16x8_t*>(qr + c1), q1);
#pragma unroll
      for (int i = 0; i < kMaxD; ++i) {
        sd[i][0] = -INFINITY;
        sd[i][1] = -INFINITY;
        if (i < p.ntowers && !(p.mask[i] && p.mask[i][qi] == 0)) {
          const bf16_t* kr = p.kv[i] + qi * p.ldkv[i];   // r == 1: the query's own token, same index in both layouts
          float k0[8], k1[8];
          cvt8(*reinterpret_cast<const bf16x8_t*>(kr + c0), k0);
          cvt8(*reinterpret_cast<const bf16x8_t*>(kr + c1), k1);
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { s0 += q0[e] * k0[e]; s1 += q1[e] * k1[e]; }
          sd[i][0] = head_sum8(s0) * p.scale;
          sd[i][1] = head_sum8(s1) * p.scale;
        }
      }
    }
    // ---- scores of the absorbed tower's tokens against U -> pw[j][head] (this wave's LDS scratch); running max per head
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < kMaxD; ++i) { mx[0] = fmaxf(mx[0], sd[i][0]); mx[1] = fmaxf(mx[1], sd[i][1]); }
    {
      bf16x8_t u[kHeads][2];
      const bf16_t* ur = p.U + qi * (int64_t)(kHeads * kC);
#pragma unroll
      for (int h = 0; h < kHeads; ++h) {
        u[h][0] = *reinterpret_cast<const bf16x8_t*>(ur + h * kC + c0);
        u[h][1] = *reinterpret_cast<const bf16x8_t*>(ur + h * kC + c1);
      }
      const float cb0 = p.cb[qi * kHeads + g], cb1 = p.cb[qi * kHeads + g + 8];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window has landed (and U, the direct towers' rows)
#pragma unroll 2
      for (int j = 0; j < na; ++j) {
        float s0 = -INFINITY, s1 = -INFINITY;
        if (!(mka && mka[j] == 0)) {
          const bf16x8_t x0 = *reinterpret_cast<const bf16x8_t*>(xs + j * 2048 + lane * 16);
          const bf16x8_t x1 = *reinterpret_cast<const bf16x8_t*>(xs + j * 2048 + 1024 + lane * 16);
          float part[kHeads];
#pragma unroll
          for (int h = 0; h < kHeads; ++h) part[h] = dot8(x1, u[h][1], dot8(x0, u[h][0], 0.f));
          float a, bsum;
          heads_reduce(part, lane, a, bsum);
          s0 = (a + cb0) * p.scale;
          s1 = (bsum + cb1) * p.scale;
        }
        mx[0] = fmaxf(mx[0], s0);
        mx[1] = fmaxf(mx[1], s1);
        if ((lane & 7) == 0) { pw[wave][j][g] = s0; pw[wave][j][g + 8] = s1; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- joint softmax per head (every lane for its two heads; the absorbed scores come back from the scratch)
    float l[2] = {0.f, 0.f}, m3v[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kMaxD; ++i)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) { sd[i][cc] = cmb_exp(sd[i][cc] - mx[cc]); l[cc] += sd[i][cc]; }   // exp(-inf) = 0
    for (int j = 0; j < na; ++j) {
      l[0] += cmb_exp(pw[wave][j][g] - mx[0]);
      l[1] += cmb_exp(pw[wave][j][g + 8] - mx[1]);
    }
    const float inv[2] = {1.0f / l[0], 1.0f / l[1]};
    float* prow0 = p.P + (qi * kHeads + g) * (int64_t)p.nkeys;
    float* prow1 = p.P + (qi * kHeads + g + 8) * (int64_t)p.nkeys;
#pragma unroll
    for (int i = 0; i < kMaxD; ++i) {
      sd[i][0] *= inv[0];
      sd[i][1] *= inv[1];
      if (i < p.ntowers && (lane & 7) == 0) { prow0[i] = sd[i][0]; prow1[i] = sd[i][1]; }
    }
    // the absorbed tokens' probabilities: fp32 to P (the backward's input), bf16 pairs of consecutive tokens to pp (the
    // coefficient operand of the token mix below; an odd window's last pair carries a zero)
    for (int j = 0; j < na; j += 2) {
      const bool two = j + 1 < na;
      const float p0a = cmb_exp(pw[wave][j][g] - mx[0]) * inv[0], p1a = cmb_exp(pw[wave][j][g + 8] - mx[1]) * inv[1];
      const float p0b = two ? cmb_exp(pw[wave][j + 1][g] - mx[0]) * inv[0] : 0.f;
      const float p1b = two ? cmb_exp(pw[wave][j + 1][g + 8] - mx[1]) * inv[1] : 0.f;
      m3v[0] += p0a + p0b;
      m3v[1] += p1a + p1b;
      if ((lane & 7) == 0) {
        prow0[p.nd + j] = p0a;
        prow1[p.nd + j] = p1a;
        if (two) { prow0[p.nd + j + 1] = p0b; prow1[p.nd + j + 1] = p1b; }
        pp[wave][j >> 1][g] = pack_bf16(p0a, p0b);
        pp[wave][j >> 1][g + 8] = pack_bf16(p1a, p1b);
      }
    }
    if ((lane & 7) == 0) {
      p.m3[qi * kHeads + g] = m3v[0];
      p.m3[qi * kHeads + g + 8] = m3v[1];
    }
    // ---- the direct towers' part of the output: sum_k p_k V_k
    {
      float o0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < kMaxD; ++i) {
        if (i < p.ntowers && !(p.mask[i] && p.mask[i][qi] == 0)) {
          const bf16_t* vr = p.kv[i] + qi * p.ldkv[i] + kC;
          float v0[8], v1[8];
          cvt8(*reinterpret_cast<const bf16x8_t*>(vr + c0), v0);
          cvt8(*reinterpret_cast<const bf16x8_t*>(vr + c1), v1);
#pragma unroll
          for (int e = 0; e < 8; ++e) { o0[e] += sd[i][0] * v0[e]; o1[e] += sd[i][1] * v1[e]; }
        }
      }
      bf16_t* orow = p.out + qi * p.ldo;
      *reinterpret_cast<bf16x8_t*>(orow + c0) = cvt8_bf16(o0[0], o0[1], o0[2], o0[3], o0[4], o0[5], o0[6], o0[7]);
      *reinterpret_cast<bf16x8_t*>(orow + c1) = cvt8_bf16(o1[0], o1[1], o1[2], o1[3], o1[4], o1[5], o1[6], o1[7]);
    }
    // ---- Xb[h][c] = sum_t p[t][h] xh_t[c]: one 512-channel half per pass (16 heads x 8 channels = 128 accumulators),
    //      two tokens per v_dot2c (a masked token has p = 0 in every head)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      bf16_t* xo = p.xbar + qi * (int64_t)(kHeads * kC);
#pragma unroll 1
      for (int cc = 0; cc < 2; ++cc) {
        float acc[kHeads][8];
#pragma unroll
        for (int h = 0; h < kHeads; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
        const char* xc = xs + cc * 1024 + lane * 16;
#pragma unroll 1
        for (int j = 0; j < na; j += 2) {
          const int j1 = j + 1 < na ? j + 1 : j;   // (coefficient 0)
          const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(xc + j * 2048);
          const u32x4_t xb2 = *reinterpret_cast<const u32x4_t*>(xc + j1 * 2048);
          uint32_t pr[8];
          interleave8(xa, xb2, pr);
          mix16(pr, &pp[wave][j >> 1][0], acc);
        }
        const int cs = cc ? c1 : c0;
#pragma unroll
        for (int h = 0; h < kHeads; ++h)
          *reinterpret_cast<bf16x8_t*>(xo + h * kC + cs) = cvt8_bf16(acc[h][0], acc[h][1], acc[h][2], acc[h][3], acc[h][4],
                                                                      acc[h][5], acc[h][6], acc[h][7]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this query is done before the next window's DMA
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sva_abs_bwd_kernel(const AbsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per wave: the token window (LDS-DMA), P and dS (scale included) of the absorbed tokens for all heads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3;
  char* xs = smem + wave * kWinBytes;
  float (*pw)[kMaxKeys][kHeads] = reinterpret_cast<float (*)[kMaxKeys][kHeads]>(smem + 4 * kWinBytes);
  float (*dw)[kMaxKeys][kHeads] = pw + 4;
  // (P, dS) of every token and head as one bf16 pair; dS of consecutive tokens as one bf16 pair
  uint32_t (*cf)[kMaxKeys][kHeads] = reinterpret_cast<uint32_t (*)[kMaxKeys][kHeads]>(smem + 4 * kWinBytes + 2 * kScratchF32);
  uint32_t (*dsp)[kMaxKeys / 2][kHeads] =
      reinterpret_cast<uint32_t (*)[kMaxKeys / 2][kHeads]>(smem + 4 * kWinBytes + 3 * kScratchF32);
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int na = p.ra * p.ra;
  for (int64_t qi = wave_global; qi < nq; qi += nwaves) {
    const int b = (int)(qi / (p.qside * p.qside));
    const int t = (int)(qi - (int64_t)b * p.qside * p.qside);
    const int qy = t / p.qside, qx = t - qy * p.qside;
    const int c0 = lane * 8, c1 = 512 + lane * 8;
    const float* prow0 = p.P + (qi * kHeads + g) * (int64_t)p.nkeys;
    const float* prow1 = p.P + (qi * kHeads + g + 8) * (int64_t)p.nkeys;
    const uint8_t* mka = p.mask_a ? p.mask_a + qi * na : nullptr;
    const int G = p.qside * p.ra;
    const int64_t xbase = (int64_t)b * G * G;
    stage_window(p, p.xhat + xbase * p.ldx, xs, na, t, qy, qx, lane);
    float q0[8], q1[8], do0[8], do1[8];
    cvt8(*reinterpret_cast<const bf16x8_t*>(p.q + qi * p.ldq + c0), q0);
    cvt8(*reinterpret_cast<const bf16x8_t*>(p.q + qi * p.ldq + c1), q1);
    cvt8(*reinterpret_cast<const bf16x8_t*>(p.dout + qi * p.lddo + c0), do0);
    cvt8(*reinterpret_cast<const bf16x8_t*>(p.dout + qi * p.lddo + c1), do1);

    // ---- dP of every key; D = sum_k P_k dP_k per head.  Direct towers: one key each (r == 1), dP = do . V
    float pd[kMaxD][2], dsd[kMaxD][2];
    float D[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kMaxD; ++i) {
      pd[i][0] = pd[i][1] = dsd[i][0] = dsd[i][1] = 0.f;
      if (i < p.ntowers && !(p.mask[i] && p.mask[i][qi] == 0)) {
        pd[i][0] = prow0[i];
        pd[i][1] = prow1[i];
        const bf16_t* vr = p.kv[i] + qi * p.ldkv[i] + kC;
        float v0[8], v1[8];
        cvt8(*reinterpret_cast<const bf16x8_t*>(vr + c0), v0);
        cvt8(*reinterpret_cast<const bf16x8_t*>(vr + c1), v1);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0 += do0[e] * v0[e]; s1 += do1[e] * v1[e]; }
        dsd[i][0] = head_sum8(s0);
        dsd[i][1] = head_sum8(s1);
        D[0] += pd[i][0] * dsd[i][0];
        D[1] += pd[i][1] * dsd[i][1];
      }
    }
    {  // absorbed tokens: dP[t][h] = dXb[h] . xh_t + dm3[h]  ->  pw = P, dw = dP (this wave's LDS scratch)
      bf16x8_t dx[kHeads][2];
      const bf16_t* dr = p.dxbar + qi * (int64_t)(kHeads * kC);
#pragma unroll
      for (int h = 0; h < kHeads; ++h) {
        dx[h][0] = *reinterpret_cast<const bf16x8_t*>(dr + h * kC + c0);
        dx[h][1] = *reinterpret_cast<const bf16x8_t*>(dr + h * kC + c1);
      }
      const float dm0 = p.dm3[qi * kHeads + g], dm1 = p.dm3[qi * kHeads + g + 8];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window has landed
#pragma unroll 2
      for (int j = 0; j < na; ++j) {
        float p0 = 0.f, p1 = 0.f, d0 = 0.f, d1 = 0.f;
        if (!(mka && mka[j] == 0)) {
          p0 = prow0[p.nd + j];
          p1 = prow1[p.nd + j];
          const bf16x8_t x0 = *reinterpret_cast<const bf16x8_t*>(xs + j * 2048 + lane * 16);
          const bf16x8_t x1 = *reinterpret_cast<const bf16x8_t*>(xs + j * 2048 + 1024 + lane * 16);
          float part[kHeads];
#pragma unroll
          for (int h = 0; h < kHeads; ++h) part[h] = dot8(x1, dx[h][1], dot8(x0, dx[h][0], 0.f));
          float a, bsum;
          heads_reduce(part, lane, a, bsum);
          d0 = a + dm0;
          d1 = bsum + dm1;
          D[0] += p0 * d0;
          D[1] += p1 * d1;
        }
        if ((lane & 7) == 0) {
          pw[wave][j][g] = p0;
          pw[wave][j][g + 8] = p1;
          dw[wave][j][g] = d0;
          dw[wave][j][g + 8] = d1;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- dS (scale included); d(cb); the coefficient pairs of the two token mixes below
    float dcb[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kMaxD; ++i) {
      dsd[i][0] = pd[i][0] * (dsd[i][0] - D[0]) * p.scale;
      dsd[i][1] = pd[i][1] * (dsd[i][1] - D[1]) * p.scale;
    }
    for (int j = 0; j < na; j += 2) {
      const bool two = j + 1 < na;
      const float p0a = pw[wave][j][g], p1a = pw[wave][j][g + 8];
      const float s0a = p0a * (dw[wave][j][g] - D[0]) * p.scale, s1a = p1a * (dw[wave][j][g + 8] - D[1]) * p.scale;
      float p0b = 0.f, p1b = 0.f, s0b = 0.f, s1b = 0.f;
      if (two) {
        p0b = pw[wave][j + 1][g];
        p1b = pw[wave][j + 1][g + 8];
        s0b = p0b * (dw[wave][j + 1][g] - D[0]) * p.scale;
        s1b = p1b * (dw[wave][j + 1][g + 8] - D[1]) * p.scale;
      }
      dcb[0] += s0a + s0b;
      dcb[1] += s1a + s1b;
      if ((lane & 7) == 0) {
        cf[wave][j][g] = pack_bf16(p0a, s0a);
        cf[wave][j][g + 8] = pack_bf16(p1a, s1a);
        if (two) {
          cf[wave][j + 1][g] = pack_bf16(p0b, s0b);
          cf[wave][j + 1][g + 8] = pack_bf16(p1b, s1b);
        }
        dsp[wave][j >> 1][g] = pack_bf16(s0a, s0b);
        dsp[wave][j >> 1][g + 8] = pack_bf16(s1a, s1b);
      }
    }
    if ((lane & 7) == 0) {
      p.dcb[qi * kHeads + g] = dcb[0];
      p.dcb[qi * kHeads + g + 8] = dcb[1];
    }
    // ---- direct towers: dq += dS K, dK = dS q, dV = P do
    {
      float dq0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dq1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < kMaxD; ++i) {
        if (i < p.ntowers) {
          bf16_t* dkr = p.dkv[i] + qi * p.ldkv[i];
          float dk0[8], dk1[8], dv0[8], dv1[8];
          if (p.mask[i] && p.mask[i][qi] == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk0[e] = dk1[e] = dv0[e] = dv1[e] = 0.f; }
          } else {
            const bf16_t* kr = p.kv[i] + qi * p.ldkv[i];
            float k0[8], k1[8];
            cvt8(*reinterpret_cast<const bf16x8_t*>(kr + c0), k0);
            cvt8(*reinterpret_cast<const bf16x8_t*>(kr + c1), k1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              dq0[e] += dsd[i][0] * k0[e];
              dq1[e] += dsd[i][1] * k1[e];
              dk0[e] = dsd[i][0] * q0[e];
              dk1[e] = dsd[i][1] * q1[e];
              dv0[e] = pd[i][0] * do0[e];
              dv1[e] = pd[i][1] * do1[e];
            }
          }
          *reinterpret_cast<bf16x8_t*>(dkr + c0) = cvt8_bf16(dk0[0], dk0[1], dk0[2], dk0[3], dk0[4], dk0[5], dk0[6], dk0[7]);
          *reinterpret_cast<bf16x8_t*>(dkr + c1) = cvt8_bf16(dk1[0], dk1[1], dk1[2], dk1[3], dk1[4], dk1[5], dk1[6], dk1[7]);
          *reinterpret_cast<bf16x8_t*>(dkr + kC + c0) = cvt8_bf16(dv0[0], dv0[1], dv0[2], dv0[3], dv0[4], dv0[5], dv0[6], dv0[7]);
          *reinterpret_cast<bf16x8_t*>(dkr + kC + c1) = cvt8_bf16(dv1[0], dv1[1], dv1[2], dv1[3], dv1[4], dv1[5], dv1[6], dv1[7]);
        }
      }
      bf16_t* dqr = p.dq + qi * p.lddq;
      *reinterpret_cast<bf16x8_t*>(dqr + c0) = cvt8_bf16(dq0[0], dq0[1], dq0[2], dq0[3], dq0[4], dq0[5], dq0[6], dq0[7]);
      *reinterpret_cast<bf16x8_t*>(dqr + c1) = cvt8_bf16(dq1[0], dq1[1], dq1[2], dq1[3], dq1[4], dq1[5], dq1[6], dq1[7]);
    }
    // ---- absorbed tower, one 512-channel half per pass, two terms per v_dot2c:
    //      d(xh_t)[c] = sum_h P[t][h] dXb[h][c] + dS[t][h] U[h][c]     ((dXb, U) element pairs x (P, dS) pairs)
    //      dU[h][c]   = sum_t dS[t][h] xh_t[c]                          (token-pair element pairs x dS pairs)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      bf16_t* dxb = p.dxhat + xbase * p.lddx;
      const bf16_t* ur = p.U + qi * (int64_t)(kHeads * kC);
      const bf16_t* dr = p.dxbar + qi * (int64_t)(kHeads * kC);
      bf16_t* duo = p.dU + qi * (int64_t)(kHeads * kC);
#pragma unroll 1
      for (int cc = 0; cc < 2; ++cc) {
        const int cs = cc ? c1 : c0;
        {
          uint32_t pr[kHeads][8];
#pragma unroll
          for (int h = 0; h < kHeads; ++h)
            interleave8(*reinterpret_cast<const u32x4_t*>(dr + h * kC + cs), *reinterpret_cast<const u32x4_t*>(ur + h * kC + cs),
                        pr[h]);
#pragma unroll 1
          for (int j = 0; j < na; ++j) {   // (a masked token has P = dS = 0 in every head: its gradient row is zero)
            float dx[2][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dx[0][e] = dx[1][e] = 0.f;
#pragma unroll
            for (int hq = 0; hq < 4; ++hq) {
              const u32x4_t c4 = *reinterpret_cast<const u32x4_t*>(&cf[wave][j][hq * 4]);
#pragma unroll
              for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int e = 0; e < 8; ++e) dx[hh & 1][e] = dot2(pr[hq * 4 + hh][e], c4[hh], dx[hh & 1][e]);
            }
            const int64_t row = token_row(p, t, qy, qx, p.ra, j);
            *reinterpret_cast<bf16x8_t*>(dxb + row * p.lddx + cs) =
                cvt8_bf16(dx[0][0] + dx[1][0], dx[0][1] + dx[1][1], dx[0][2] + dx[1][2], dx[0][3] + dx[1][3],
                          dx[0][4] + dx[1][4], dx[0][5] + dx[1][5], dx[0][6] + dx[1][6], dx[0][7] + dx[1][7]);
          }
        }
        {
          float du[kHeads][8];
#pragma unroll
          for (int h = 0; h < kHeads; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) du[h][e] = 0.f;
          const char* xc = xs + cc * 1024 + lane * 16;
#pragma unroll 1
          for (int j = 0; j < na; j += 2) {
            const int j1 = j + 1 < na ? j + 1 : j;   // (coefficient 0)
            const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(xc + j * 2048);
            const u32x4_t xb2 = *reinterpret_cast<const u32x4_t*>(xc + j1 * 2048);
            uint32_t xp[8];
            interleave8(xa, xb2, xp);
            mix16(xp, &dsp[wave][j >> 1][0], du);
          }
#pragma unroll
          for (int h = 0; h < kHeads; ++h)
            *reinterpret_cast<bf16x8_t*>(duo + h * kC + cs) =
                cvt8_bf16(du[h][0], du[h][1], du[h][2], du[h][3], du[h][4], du[h][5], du[h][6], du[h][7]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this query is done before the next window's DMA
    __builtin_amdgcn_wave_barrier();
  }
}

int fill(const cmb_sva_abs_desc* d, AbsParams& p, bool bwd) {
  if (!d || !d->q || !d->xhat || !d->U || !d->cb || !d->out || !d->xbar || !d->m3 || !d->P) return CMB_ERR_BAD_ARG;
  if (d->B < 0 || d->qside <= 0 || d->heads != kHeads || d->hd != kHd) return CMB_ERR_SHAPE;
  if (d->ntowers < 0 || d->ntowers > kMaxD || d->ra <= 0 || d->ra * d->ra > kMaxKeys) return CMB_ERR_SHAPE;
  p.B = d->B; p.qside = d->qside; p.ntowers = d->ntowers; p.window_major = d->window_major;
  p.q = (const bf16_t*)d->q; p.ldq = d->ldq;
  p.nd = 0;
  for (int i = 0; i < d->ntowers; ++i) {
    if (!d->kv[i]) return CMB_ERR_BAD_ARG;
    if (d->r[i] != 1) return CMB_ERR_SHAPE;   // directly projected towers: one key per query
    p.kv[i] = (const bf16_t*)d->kv[i]; p.ldkv[i] = d->ldkv[i];
    p.mask[i] = d->mask[i];
    p.dkv[i] = (bf16_t*)d->dkv[i];
    if (bwd && !d->dkv[i]) return CMB_ERR_BAD_ARG;
    p.nd += d->r[i] * d->r[i];
  }
  p.ra = d->ra;
  p.xhat = (const bf16_t*)d->xhat; p.ldx = d->ldx;
  p.mask_a = d->mask_a;
  p.U = (const bf16_t*)d->U; p.cb = d->cb;
  p.out = (bf16_t*)d->out; p.ldo = d->ldo;
  p.xbar = (bf16_t*)d->xbar; p.m3 = d->m3; p.P = d->P;
  p.nkeys = p.nd + d->ra * d->ra;
  p.dout = (const bf16_t*)d->dout; p.lddo = d->lddo;
  p.dxbar = (const bf16_t*)d->dxbar; p.dm3 = d->dm3;
  p.dq = (bf16_t*)d->dq; p.lddq = d->lddq;
  p.dU = (bf16_t*)d->dU; p.dcb = d->dcb;
  p.dxhat = (bf16_t*)d->dxhat; p.lddx = d->lddx;
  if (bwd && (!d->dout || !d->dxbar || !d->dm3 || !d->dq || !d->dU || !d->dcb || !d->dxhat)) return CMB_ERR_BAD_ARG;
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
  int64_t blocks = (nq + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sva_abs_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSmemFwd) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(sva_abs_fwd_kernel, dim3((unsigned)blocks), dim3(256), kSmemFwd, (hipStream_t)stream, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_sva_abs_bwd(const cmb_sva_abs_desc* d, void* stream) {
  AbsParams p;
  const int rc = fill(d, p, true);
  if (rc != CMB_OK) return rc;
  const int64_t nq = (int64_t)p.B * p.qside * p.qside;
  if (nq == 0) return CMB_OK;
  int64_t blocks = (nq + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sva_abs_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSmemBwd) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(sva_abs_bwd_kernel, dim3((unsigned)blocks), dim3(256), kSmemBwd, (hipStream_t)stream, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
