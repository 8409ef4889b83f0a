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
#include <type_traits>
#include "common.h"
#include "flash_layout.h"
#include "vit_layout.h"

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

// =====================================================================================================================
// Round 6: the same attention on the operand path of flash2.hip.  Lane <-> element arrangement, products and
// softmax of vit_attn_bf16_kernel above (a lane owns a query; S^T = K Q^T; O^T += V^T P^T; P from accumulator to operand with a
// convert); what changed is how K / V reach the matrix pipe:
//   * K / V tiles of 64 keys go global -> LDS by LDS-DMA, un-padded and XOR-swizzled on the SOURCE side (head_dim 64: 128-byte
//     rows, vit_layout.h; head_dim 96: the 256-byte rows of flash_layout.h with 12 of 16 slots used): no tile registers, no
//     ds_write, no 16-bit shuffles packing a transposed V image (the register-staged pair above: 4 global loads, 2 + 8 LDS stores
//     and ~20 shuffles / address instructions per thread and tile);
//   * V is consumed from the row-major image by transposing reads (ds_read_b64_tr_b16), a ring of four fragments ahead;
//   * two K and two V buffers, ONE barrier per tile: top of tile t = my pieces of tile t landed (vmcnt(0)) + barrier, then the
//     DMA of tile t + 1 goes out and flies for the whole tile;
//   * the exponentials of key block kb + 1 are issued between the MFMAs of key block kb (packed exponent argument, two partial
//     row sums), only key block 0's run with the matrix pipe idle;
//   * a 1-D grid whose linear block id is dealt so that an XCD owns whole (image, head) groups: the 5-6 query blocks of a head
//     read its K / V out of ONE L2 (the 3-D grid above spread them over 6 XCDs).
// N is ragged (577 / 729 / 730): rows past N - 1 are fetched from row N - 1 (finite values), their scores masked to -inf in the
// last tile, query rows past N - 1 computed and not stored.
template <int HD> struct VitTile;
template <> struct VitTile<64> {
  static constexpr int ROWB = VL_ROW_BYTES, TILE = VL_TILE_BYTES, PIECES = 8;
  static __device__ __forceinline__ int dma_row(int piece, int lane) { return vl_dma_row(piece, lane); }
  static __device__ __forceinline__ int dma_slot(int piece, int lane) { return vl_dma_src_slot(piece, lane); }
  static __device__ __forceinline__ int row_frag(int row0, int ks, int lane) { return vl_row_frag_off(row0, ks, lane); }
  static __device__ __forceinline__ int tr_frag(int r16, int c32, int read, int lane) { return vl_tr_frag_off(r16, c32, read, lane); }
};
template <> struct VitTile<96> {
  static constexpr int ROWB = FL_ROW_BYTES, TILE = FL_TILE_BYTES, PIECES = 16;
  static __device__ __forceinline__ int dma_row(int piece, int lane) { return fl_dma_row(piece, lane); }
  static __device__ __forceinline__ int dma_slot(int piece, int lane) {   // logical slots 12..15 do not exist: any valid address
    const int sl = fl_dma_src_slot(piece, lane);
    return sl < 12 ? sl : sl - 12;
  }
  static __device__ __forceinline__ int row_frag(int row0, int ks, int lane) { return fl_row_frag_off(row0, ks, lane); }
  static __device__ __forceinline__ int tr_frag(int r16, int c32, int read, int lane) { return fl_tr_frag_off(r16, c32, read, lane); }
};

#define VA_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef short va_s16x4_t __attribute__((ext_vector_type(4)));
typedef short va_s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) va_s16x4_t* va_lds_tr_ptr;
typedef __bf16 va_bf16x2_v __attribute__((ext_vector_type(2)));
typedef uint32_t va_u32x4_v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t va_cvt2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, va_bf16x2_v));
}
__device__ __forceinline__ const char* va_uniform_ptr(const char* q) {
  const uint64_t v = (uint64_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}
// one LDS-DMA piece: 64 lanes x 16 bytes from base + voff (per lane) to LDS byte address lds (wave-uniform), lane-linear
__device__ __forceinline__ void va_dma_piece(const char* base, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

template <int HD, int NW, int WPE>
__global__ void __launch_bounds__(64 * NW, WPE) vit_attn_dma_kernel(const bf16_t* __restrict__ qkv, int N, int heads, int nqb,
                                                                float c2, bf16_t* __restrict__ out) {
  typedef VitTile<HD> T;
  constexpr int KS = HD / 16, DT = HD / 32, PW = T::PIECES / NW, NP = 4 * DT;   // NW waves = 32 NW queries per workgroup
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, j = lane & 31;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem));
  // XCD r (= blockIdx.x % 8) owns a contiguous range of linear ids; ids run query-block fastest inside an (image, head) group
  const int id = gl_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int qb = id % nqb, bh = id / nqb;
  const int h = bh % heads, b = bh / heads;
  const int C = heads * HD;
  const int64_t ld = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
  const int q0 = qb * (32 * NW) + wave * 32, qi = q0 + j;
  const bool active = q0 < N;   // (wave-uniform) a wave past the last query only fetches its share of the tiles

  bf16x8_t qf[KS];
  {
    const bf16_t* qrow = base + (int64_t)(qi < N ? qi : N - 1) * ld;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + g * 8);
  }
  const int nt = (N + 63) / 64;
  const int last_rows = N - 64 * (nt - 1);   // valid rows of the last tile (1 .. 64)
  // DMA: this wave's pieces PW w .. PW w + PW - 1 of a tile (byte offsets from the tile's first row); a second set of offsets
  // for the ragged last tile.  (A fifth wave issuing every piece — an LDS-DMA instruction costs its wave 60-180 cycles — was
  // built and measured: 84 vs 60 us on CLIP-L, one wave cannot issue 16-32 pieces per tile fast enough; profiles/r06_lab.md.)
  uint32_t voff[PW], voff_l[PW];
  const uint32_t ld2 = (uint32_t)(ld * 2);
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int piece = PW * wave + i, row = T::dma_row(piece, lane), sl = T::dma_slot(piece, lane);
    voff[i] = (uint32_t)row * ld2 + (uint32_t)sl * 16u;
    voff_l[i] = (uint32_t)(row < last_rows ? row : last_rows - 1) * ld2 + (uint32_t)sl * 16u;
  }
  uint32_t kro[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kro[ks] = (uint32_t)T::row_frag(0, ks, lane);
  uint32_t tro[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 2; ++r) tro[d][r] = (uint32_t)T::tr_frag(0, 32 * d, r, lane);
  // (the compiler's waits for the Q loads belong here, not inside the tile loop where they would await the next tile's fills)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

  const char* kbase = va_uniform_ptr(reinterpret_cast<const char*>(base + C));
  const char* vbase = va_uniform_ptr(reinterpret_cast<const char*>(base + 2 * C));
  const int64_t tile_bytes = 64 * ld * 2;
  auto issue = [&](int t) {   // K tile t -> K buffer t & 1, V tile t -> V buffer t & 1
    const bool last = t == nt - 1;
    const uint32_t dk = lds0 + (uint32_t)((t & 1) * T::TILE) + (uint32_t)wave * (uint32_t)(PW * 1024);
    const uint32_t dv = dk + 2u * T::TILE;
    const char* ks = kbase + (int64_t)t * tile_bytes;
    const char* vs = vbase + (int64_t)t * tile_bytes;
#pragma unroll
    for (int i = 0; i < PW; ++i) va_dma_piece(ks, last ? voff_l[i] : voff[i], dk + (uint32_t)i * 1024u);
#pragma unroll
    for (int i = 0; i < PW; ++i) va_dma_piece(vs, last ? voff_l[i] : voff[i], dv + (uint32_t)i * 1024u);
  };
  auto kfrag = [&](const char* kbuf, int i) __attribute__((always_inline)) -> bf16x8_t {   // product i = 2 ks + kt
    return *reinterpret_cast<const bf16x8_t*>(kbuf + kro[i >> 1] + (i & 1) * 32 * T::ROWB);
  };
  auto vfrag_half = [&](const char* vbuf, int i, int r) __attribute__((always_inline)) -> va_s16x4_t {   // product i = DT kb + d
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((va_lds_tr_ptr)(vbuf + tro[i % DT][r] + (i / DT) * 16 * T::ROWB));
  };

  f32x16_t acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY;
  f32x2_t l2 = {0.f, 0.f};

  // One key tile.  LAST (compile time): the ragged last tile masks its keys >= N — as a run-time test inside one loop body the
  // compiler turned the mask into 32 v_cndmask executed on EVERY tile (19 % of the kernel's vector instructions).
  auto tile = [&](int t, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of K(t), V(t) have landed
    __syncthreads();                                   // everybody's have; everybody is done with K(t - 1), V(t - 1)
    if (!LAST) issue(t + 1);                           // the next tile flies during this tile's products
    if (!active) return;
    const char* kc = smem + (t & 1) * T::TILE;
    const char* vc = kc + 2 * T::TILE;
    f32x16_t s0 = {0}, s1 = {0};   // (overwritten by the first two products)
    {   // S^T = K Q^T: fragments a ring of four ahead
      bf16x8_t kr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kr[i] = kfrag(kc, i);
      VA_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        if (i < 2) (i ? s1 : s0) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[0], (f32x16_t){0}, 0, 0, 0);
        else (i & 1 ? s1 : s0) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[i >> 1], (i & 1 ? s1 : s0), 0, 0, 0);
        if (i + 4 < 2 * KS) kr[i & 3] = kfrag(kc, i + 4);
        VA_FENCE();
      }
    }
    if (LAST && last_rows < 64) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (key >= last_rows) (kt ? s1 : s0)[r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s0[r]), s0[r + 1]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s1[r]), s1[r + 1]);
    {   // the other lane half's maximum: v_permlane32_swap (one vector instruction; __shfl_xor is an LDS round trip)
      float a = mx, b = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: [x_lo, x_lo], b: [x_hi, x_hi]
      mx = fmaxf(a, b) * c2;   // c2 > 0: max of the scaled scores
    }
    // LAZY reference maximum: m follows the row maximum only when that has moved by more than kLazy (in log2 units), so the
    // rescale pass (32 or 48 accumulator multiplies) runs on the first tile and then almost never; until then probabilities are
    // taken against the stale m and may reach 2^kLazy (bf16 / fp32 have the range; relative precision is unchanged; the final
    // division by the row sum, accumulated against the same m, makes the result exact).
    constexpr float kLazy = 6.0f;
    if (__builtin_amdgcn_ballot_w64(mx > m + kLazy) != 0) {
      const float m_new = mx > m + kLazy ? mx : m;
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l2 *= (f32x2_t){alpha, alpha};
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
      m = m_new;
    }
    {   // O^T += V^T P^T with the exponentials inside it
      const f32x2_t nm2 = {-m, -m}, c22 = {c2, c2};
      auto pel = [&](int e) -> float { return e < 16 ? s0[e] : s1[e - 16]; };
      auto exp_pair = [&](int e) __attribute__((always_inline)) {
        f32x16_t& s = (e < 16) ? s0 : s1;
        const int r = e & 15;
        const f32x2_t a = __builtin_elementwise_fma((f32x2_t){s[r], s[r + 1]}, c22, nm2);
        const f32x2_t pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        l2 += pv;
        s[r] = pv[0];
        s[r + 1] = pv[1];
      };
      va_u32x4_v pf[2];
      va_s16x4_t vlo[4], vhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vlo[i] = vfrag_half(vc, i, 0);
        vhi[i] = vfrag_half(vc, i, 1);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        exp_pair(2 * c);
        pf[0][c] = va_cvt2(pel(2 * c), pel(2 * c + 1));
      }
      VA_FENCE();
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int kb = i / DT, d = i % DT;
        const va_s16x8_t vv = {vlo[i & 3][0], vlo[i & 3][1], vlo[i & 3][2], vlo[i & 3][3],
                               vhi[i & 3][0], vhi[i & 3][1], vhi[i & 3][2], vhi[i & 3][3]};
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), __builtin_bit_cast(bf16x8_t, pf[kb & 1]),
                                                         acc[d], 0, 0, 0);
        if (i + 4 < NP) {
          vlo[i & 3] = vfrag_half(vc, i + 4, 0);
          vhi[i & 3] = vfrag_half(vc, i + 4, 1);
        }
        if (kb < 3) {   // the NEXT key block's operand: its 4 element pairs spread over this key block's DT products
          const int c_lo = (4 * d) / DT, c_hi = (4 * (d + 1)) / DT;
          for (int c = c_lo; c < c_hi; ++c) {
            const int e = 8 * (kb + 1) + 2 * c;
            exp_pair(e);
            pf[(kb + 1) & 1][c] = va_cvt2(pel(e), pel(e + 1));
          }
        }
        VA_FENCE();
      }
    }
  };
  issue(0);
  for (int t = 0; t < nt - 1; ++t) tile(t, std::false_type{});
  tile(nt - 1, std::true_type{});
  const float l = l2[0] + l2[1];
  const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
  if (qi < N) {
    bf16_t* orow = out + ((int64_t)b * N + qi) * C + h * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[d][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4_t*>(orow + d * 32 + 8 * qd + 4 * g) = o;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
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
    // CMB_KNOB_VIT_ATTN: 0 = round 3's structure (two barriers per tile); 1 = one barrier per tile (double-buffered LDS);
    // 2 (default) / 3 = vit_attn_dma_kernel with 128 / 256 queries per workgroup.
    // (Capping the registers for 4 / 3 waves per SIMD spills 13 / 19 registers and measured 25-40 % slower: not kept.)
    const int variant = cmb_knob(CMB_KNOB_VIT_ATTN);
    if (variant >= 2) {   // round 6: LDS-DMA tiles + transposing reads (vit_attn_dma_kernel), 1-D XCD-grouped grid;
                          // 2 = 4 waves (128 queries) per workgroup, 3 = 8 waves (256 queries: half the K / V bytes out of L2)
      const int nw = variant == 2 ? 4 : 8;
      const int nqb = (int)((N + 32 * nw - 1) / (32 * nw));
      const int64_t nblk = (int64_t)B * heads * nqb;
      if (nblk > 0x7fffffff || (3 * (int64_t)heads * hd * 2) * 64 > 0x7fffffff) return CMB_ERR_BAD_ARG;
#define VIT_DMA_LAUNCH(HD_, NW_, WPE_, SMEM_)                                                                                  \
  do {                                                                                                                         \
    auto kern = vit_attn_dma_kernel<HD_, NW_, WPE_>;                                                                           \
    static CmbAttrOnce attr_once; /* up to 64 KiB of dynamic LDS: above the 48 KiB a kernel gets without asking */           \
    if (const uint32_t attr_bit = attr_once.need()) {                                                                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_) !=       \
          hipSuccess)                                                                                                          \
        return CMB_ERR_LAUNCH;                                                                                                 \
      attr_once.done(attr_bit);                                                                                                        \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NW_), SMEM_, s, (const bf16_t*)qkv, (int)N, heads, nqb, sl2,     \
                       (bf16_t*)out);                                                                                          \
  } while (0)
      if (hd == 64 && nw == 4) VIT_DMA_LAUNCH(64, 4, 3, 4 * VL_TILE_BYTES);
      else if (hd == 64) VIT_DMA_LAUNCH(64, 8, 4, 4 * VL_TILE_BYTES);
      else if (nw == 4) VIT_DMA_LAUNCH(96, 4, 2, 4 * FL_TILE_BYTES);
      else VIT_DMA_LAUNCH(96, 8, 2, 4 * FL_TILE_BYTES);
#undef VIT_DMA_LAUNCH
      CMB_CHECK_LAUNCH();
      return CMB_OK;
    }
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
