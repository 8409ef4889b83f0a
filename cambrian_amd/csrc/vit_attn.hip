// vit_attn.hip — non-causal multi-head self-attention of the ViT towers (CLIP-L, SigLIP-SO400M,
// DINOv2-g; N = 577 / 729 / 730 tokens, head_dim 64 or 72->96) for gfx950, forward only (the towers
// are frozen: clip_encoder.py:103, siglip_encoder.py:96, dino_encoder.py:158 run under
// torch.set_grad_enabled(False)).
//
// bf16 path: flash-style, one workgroup = 4 waves = 128 queries of one (image, head); K/V tiles of 64
// keys go through LDS (register-staged, issued before the compute of the previous tile).  Both GEMMs
// run on v_mfma_f32_32x32x16_bf16 with the operands arranged so that a lane owns ONE query:
//   S^T = K·Q^T   -> lane (query j, half g) holds 16 of the 32 keys of a sub-tile: the row max / sum
//                    need a single cross-lane exchange (lane ^ 32);
//   O^T = V^T·P^T -> the contraction slots of the MFMA are assigned to keys in exactly the order the
//                    S^T accumulator already has them (keys {0-3,8-11} for g=0, {4-7,12-15} for g=1 of
//                    every 16-key block), so P goes from accumulator to operand with a bf16 convert and
//                    no shuffle; V^T fragments are two 8-byte LDS reads in the same key order.
// fp32 path (parity tests): one wave per query, VALU dot products, same online softmax.
#include "common.h"

namespace {

// WPE: waves per SIMD the register allocator must make room for (1 = whatever the kernel needs: 3 for head_dim 64, 2-3 for
// 96); DB: two LDS tile buffers, ONE barrier per key tile (the next tile is stored into the buffer the previous iteration
// read, which every wave left at that iteration's barrier) instead of two.
template <int HD, int WPE, bool DB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) vit_attn_bf16_kernel(const bf16_t* __restrict__ qkv, int N, int heads,
                                                            float scale_log2e, bf16_t* __restrict__ out) {
  constexpr int KS = HD / 16;   // MFMA k-steps of QK^T
  constexpr int DT = HD / 32;   // 32-wide output tiles along d
  constexpr int LDK = HD + 8;   // K tile row stride (elements): conflict-free ds_read_b128
  constexpr int LDV = 68;       // V^T tile row stride (elements): conflict-free ds_read_b64
  constexpr int KV8 = HD / 8;   // vec8 per key row
  constexpr int K_IT = (64 * KV8) / 256;
  constexpr int V_TASKS = 32 * KV8;
  constexpr int V_IT = (V_TASKS + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t sK_[(DB ? 2 : 1) * 64 * LDK];
  __shared__ __attribute__((aligned(16))) bf16_t sV_[(DB ? 2 : 1) * HD * LDV];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, j = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int C = heads * HD;
  const int64_t ld = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * ld;
  const bf16_t* qp = base + h * HD;
  const bf16_t* kp = base + C + h * HD;
  const bf16_t* vp = base + 2 * C + h * HD;

  bf16x8_t qf[KS];
  {
    const int qrow = (q0 + j < N) ? (q0 + j) : (N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + qrow * ld + ks * 16 + g * 8);
  }

  bf16x8_t kreg[K_IT], v0reg[V_IT], v1reg[V_IT];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int id = tid + 256 * i, key = id / KV8, cv = id - key * KV8;
      int row = t * 64 + key;
      row = row < N ? row : N - 1;
      kreg[i] = *reinterpret_cast<const bf16x8_t*>(kp + row * ld + cv * 8);
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int id = tid + 256 * i;
      if (id < V_TASKS) {
        const int kpair = id & 31, dg = id >> 5;
        int r0 = t * 64 + 2 * kpair, r1 = r0 + 1;
        r0 = r0 < N ? r0 : N - 1;
        r1 = r1 < N ? r1 : N - 1;
        v0reg[i] = *reinterpret_cast<const bf16x8_t*>(vp + r0 * ld + dg * 8);
        v1reg[i] = *reinterpret_cast<const bf16x8_t*>(vp + r1 * ld + dg * 8);
      }
    }
  };
  auto store_tile = [&](int buf) {
    bf16_t* sK = sK_ + buf * 64 * LDK;
    bf16_t* sV = sV_ + buf * HD * LDV;
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int id = tid + 256 * i, key = id / KV8, cv = id - key * KV8;
      *reinterpret_cast<bf16x8_t*>(sK + key * LDK + cv * 8) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int id = tid + 256 * i;
      if (id < V_TASKS) {
        const int kpair = id & 31, dg = id >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          bf16x2_t pr;
          pr[0] = v0reg[i][e];
          pr[1] = v1reg[i][e];
          *reinterpret_cast<bf16x2_t*>(sV + (dg * 8 + e) * LDV + 2 * kpair) = pr;
        }
      }
    }
  };

  f32x16_t acc_o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
  float m = -INFINITY;
  f32x2_t l2 = {0.f, 0.f};   // the row sum as two partial sums: v_pk_add_f32 (round 4: the kernel is VALU-bound in the softmax)

  const int nt = (N + 63) / 64;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_tile(t + 1);  // in flight while this tile is consumed
    const bf16_t* sK = sK_ + (DB ? (t & 1) : 0) * 64 * LDK;
    const bf16_t* sV = sV_ + (DB ? (t & 1) : 0) * HD * LDV;

    f32x16_t s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + (kt * 32 + j) * LDK + ks * 16 + g * 8);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    // VALU budget (one pass per 32 MFMAs of the wave; see flash_bwd.hip): only the last tile can hold keys >= N, scores
    // stay un-scaled until one fma feeds v_exp_f32, the accumulators are rescaled only when some lane's maximum moved,
    // P is converted two elements per v_cvt_pk_bf16_f32.
    if (t * 64 + 64 > N) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (key >= N) s[kt][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;   // scale > 0
    const float m_new = fmaxf(m, mx);
    if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l2 *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
    }
    m = m_new;
    {
      const f32x2_t sc2 = {scale_log2e, scale_log2e}, mn2 = {-m_new, -m_new};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t a = __builtin_elementwise_fma((f32x2_t){s[kt][r], s[kt][r + 1]}, sc2, mn2);
          const f32x2_t pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          l2 += pv;
          s[kt][r] = pv[0];
          s[kt][r + 1] = pv[1];
        }
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int kt = kb >> 1, hh = kb & 1;
      const bf16x8_t pf = cvt8_bf16(s[kt][8 * hh + 0], s[kt][8 * hh + 1], s[kt][8 * hh + 2], s[kt][8 * hh + 3],
                                    s[kt][8 * hh + 4], s[kt][8 * hh + 5], s[kt][8 * hh + 6], s[kt][8 * hh + 7]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const bf16_t* vrow = sV + (d * 32 + j) * LDV + kt * 32 + 16 * hh + 4 * g;
        const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(vrow);
        const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(vrow + 8);
        bf16x8_t vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
        acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc_o[d], 0, 0, 0);
      }
    }
    if constexpr (DB) {
      if (t + 1 < nt) store_tile((t + 1) & 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (t + 1 < nt) store_tile(0);
      __syncthreads();
    }
  }

  const float l = l2[0] + l2[1];
  const float l_tot = l + __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + j;
  if (q < N) {
    bf16_t* orow = out + ((int64_t)b * N + q) * C + h * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc_o[d][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4_t*>(orow + d * 32 + 8 * qd + 4 * g) = o;
      }
  }
}

// Reference-grade generic kernel (fp32 parity path, also bf16 for A/B checks): one wave per query.
template <typename T>
__global__ void __launch_bounds__(256) vit_attn_simple_kernel(const T* __restrict__ qkv, int64_t B, int N, int heads,
                                                              int hd, float scale, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t total = B * heads * (int64_t)N;
  const int C = heads * hd;
  const int64_t ld = 3 * (int64_t)C;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += (int64_t)gridDim.x * 4) {
    const int q = (int)(w % N);
    const int64_t bh = w / N;
    const int h = (int)(bh % heads);
    const int64_t b = bh / heads;
    const T* base = qkv + b * N * ld + h * hd;
    const int d0 = lane, d1 = lane + 64;
    const float q0 = (d0 < hd) ? (float)base[q * ld + d0] * scale : 0.f;
    const float q1 = (d1 < hd) ? (float)base[q * ld + d1] * scale : 0.f;
    float m = -INFINITY, l = 0.f, a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < N; ++k) {
      const T* kr = base + C + k * ld;
      const T* vr = base + 2 * C + k * ld;
      float s = 0.f;
      if (d0 < hd) s += q0 * (float)kr[d0];
      if (d1 < hd) s += q1 * (float)kr[d1];
      s = wave_sum(s);
      const float mn = fmaxf(m, s);
      const float alpha = expf(m - mn), pj = expf(s - mn);
      l = l * alpha + pj;
      a0 = a0 * alpha + ((d0 < hd) ? pj * (float)vr[d0] : 0.f);
      a1 = a1 * alpha + ((d1 < hd) ? pj * (float)vr[d1] : 0.f);
      m = mn;
    }
    T* orow = out + (b * N + q) * C + h * hd;
    if (d0 < hd) orow[d0] = (T)(a0 / l);
    if (d1 < hd) orow[d1] = (T)(a1 / l);
  }
}

}  // namespace

extern "C" int cmb_vit_attn_fwd(int dtype, const void* qkv, int64_t B, int64_t N, int32_t heads, int32_t hd,
                                float scale, void* out, int32_t force_simple, void* stream) {
  if (!qkv || !out || B < 0 || N <= 0 || heads <= 0 || hd <= 0 || hd > 128) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16 && !force_simple && (hd == 64 || hd == 96)) {
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)heads, (unsigned)B);
    const float sl2 = scale * 1.4426950408889634f;
    // CMB_KNOB_VIT_ATTN: 0 = round 3's structure (two barriers per tile); 1 = one barrier per tile (double-buffered LDS).
    // (Capping the registers for 4 / 3 waves per SIMD spills 13 / 19 registers and measured 25-40 % slower: not kept.)
    const int variant = cmb_knob(CMB_KNOB_VIT_ATTN);
#define VIT_LAUNCH(HD_, WPE_, DB_)                                                                                  \
  hipLaunchKernelGGL((vit_attn_bf16_kernel<HD_, WPE_, DB_>), grid, dim3(256), 0, s, (const bf16_t*)qkv, (int)N, heads, sl2, \
                     (bf16_t*)out)
    if (hd == 64) {
      if (variant == 1) VIT_LAUNCH(64, 1, true);
      else VIT_LAUNCH(64, 1, false);
    } else {
      if (variant == 1) VIT_LAUNCH(96, 1, true);
      else VIT_LAUNCH(96, 1, false);
    }
#undef VIT_LAUNCH
  } else {
    int64_t blocks = (B * heads * N + 3) / 4;
    if (blocks > 65535) blocks = 65535;
    if (dtype == CMB_BF16)
      hipLaunchKernelGGL(vit_attn_simple_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)qkv, B,
                         (int)N, heads, hd, scale, (bf16_t*)out);
    else if (dtype == CMB_F32)
      hipLaunchKernelGGL(vit_attn_simple_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)qkv, B,
                         (int)N, heads, hd, scale, (float*)out);
    else
      return CMB_ERR_BAD_ARG;
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
