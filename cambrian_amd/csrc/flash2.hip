// flash2.hip — round-5 forward of the decoder's causal grouped-query attention (S = 2048, head_dim 128) for gfx950.
// Same problem, same lane <-> element arrangement and the same products as flash_fwd_kernel (flash_bwd.hip: a LANE owns a
// query, S^T = K Q^T, online softmax in base 2, O^T += V^T P^T, accumulator -> operand hand-off without shuffles); what
// changed is how the operands reach the matrix pipe and in which order a wave issues its work:
//   * K / V tiles (64 keys x 128) go global -> LDS by LDS-DMA (16 one-KiB pieces per tile, four per wave), un-padded and
//     XOR-swizzled on the source side (flash_layout.h): no tile registers, no 16-bit shuffles, no ds_write — the round-4
//     kernel spent ~80 of its ~230 vector instructions per tile and wave packing the transposed V image and 32 registers on
//     the tile in flight.  V is consumed straight from the row-major image with transposing reads (ds_read_b64_tr_b16);
//   * two K and two V buffers, ONE barrier per tile (the round-4 kernel: two): top of iteration t = "my pieces of K(t), V(t)
//     have landed" (s_waitcnt vmcnt(0)) + barrier, then the DMA of tile t + 1 goes out and flies for the whole tile;
//   * the LDS fragment reads of both products run a ring of four fragments ahead of the MFMAs in a fixed order (FL_FENCE);
//   * the exponentials sit INSIDE the P V phase: key block kb's eight probabilities are computed and converted while the four
//     MFMAs of key block kb - 1 run (five vector instructions per MFMA), packed fp32 for the exponent argument and the row sum
//     (two partial sums per lane); only the row maximum and the first key block's exponentials run with the matrix pipe idle.
// Two workgroups per CU as before (the other workgroup's waves fill a wave's softmax phase).  Variants measured on the way
// (profiles/r05_lab.md): two score sets per wave with the next tile's S product issued beside this tile's exponentials — 256
// registers do not hold them (289 spills), one workgroup per CU with 386-512 registers ran 578 TFLOP/s against the round-4
// kernel's 715: hipcc moved the score sets between the register halves ~100 times per tile; 64-query waves: 512 registers + spills.
// Results agree with flash_fwd_kernel to fp32 rounding (the row sum is accumulated in two partial sums here), not bit for
// bit; tests/test_flash_bwd_gpu.py holds both against fp32 autograd of the plain formula.
#include <type_traits>
#include "flash_common.h"
#include "flash_layout.h"

namespace cmb_flash {
namespace {

#define FL_FENCE() __builtin_amdgcn_sched_barrier(0)

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));

constexpr int kTile = FL_TILE_BYTES;            // 16 KiB
constexpr int kSmemFwd = 4 * kTile;             // K ring (2) + V ring (2)

__device__ __forceinline__ uint32_t cvt2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_v));
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
  const uint64_t v = (uint64_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}
// one LDS-DMA piece: 64 lanes x 16 bytes from base + voff (per lane) to LDS byte address lds (wave-uniform), lane-linear
__device__ __forceinline__ void dma_piece(const char* base, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(base), "s"(lds)
               : "memory", "m0");
}

template <bool CAUSAL, bool MASKED>
__global__ void __launch_bounds__(256, 2) flash_fwd2_kernel(const FlashParams p, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse_out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, j = lane & 31;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem));
  const int nqb = p.S / 128;
  const FlashBlock fb = flash_block_qh((int)blockIdx.x, (int)gridDim.x, flash_items(nqb, CAUSAL), p.H, p.HKV);
  const int b = fb.b, h = fb.h, hk = fb.hk;
  const int nrep = flash_pair_count(nqb, fb.blk, CAUSAL);
  const float c2 = p.scale * LOG2E;

  // ---- per-lane constants of the operand tile (flash_layout.h)
  // DMA: this wave's pieces 4 w .. 4 w + 3 of a tile (rows 16 w .. 16 w + 15)
  uint32_t voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = 4 * wave + i;
    voff[i] = (uint32_t)(fl_dma_row(piece, lane) * (int)(p.kv_ss * 2) + fl_dma_src_slot(piece, lane) * 16);
  }
  // row-major K fragments: product i = 2 ks + kt reads row 32 kt + j, logical slot 2 ks + g (rows 32 apart share the swizzle)
  uint32_t kro[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kro[ks] = (uint32_t)fl_row_frag_off(0, ks, lane);
  // transposed V fragments: product i = 4 kb + d reads rows 16 kb + ..., columns 32 d + j (16 rows further = + 4096 bytes)
  uint32_t tro[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 2; ++r) tro[d][r] = (uint32_t)fl_tr_frag_off(0, 32 * d, r, lane);

  for (int rep = 0; rep < nrep; ++rep) {
    const int qb = flash_pair_q(nqb, fb.blk, rep, CAUSAL);
    const int q0 = qb * 128 + wave * 32;
    const int qi = q0 + j;
    const bf16_t* qrow = p.q + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + g * 8);
    // The fragments are awaited HERE.  Left to the compiler, their first uses inside the tile loop carry `s_waitcnt vmcnt(7 .. 0)`
    // (the eight loads above, counted without the LDS-DMA instructions it cannot see) on EVERY tile — and a vmcnt(0) in the
    // middle of a tile waits for the next tile's fills, issued a moment earlier: the prefetch ran serialised with the S products.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    const char* kbase = uniform_ptr(reinterpret_cast<const char*>(p.k + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh));
    const char* vbase = uniform_ptr(reinterpret_cast<const char*>(p.v + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh));
    const int64_t tile_bytes = (int64_t)64 * p.kv_ss * 2;   // global bytes from one tile's first row to the next's
    f32x16_t acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    float m = -INFINITY;
    f32x2_t l2 = {0.f, 0.f};
    const int nt = CAUSAL ? (qb * 128 + 128) / 64 : (p.kv_len + 63) / 64;
    const int t_last = CAUSAL ? (q0 + 31) / 64 : nt - 1;   // the last tile that holds an open key for some query of this wave

    auto issue_k = [&](int t) {   // K tile t -> K buffer t & 1
      const char* src = kbase + (int64_t)t * tile_bytes;
      const uint32_t dst = lds0 + (uint32_t)((t & 1) * kTile) + (uint32_t)wave * 4096u;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_piece(src, voff[i], dst + (uint32_t)i * 1024u);
    };
    auto issue_v = [&](int t) {   // V tile t -> V buffer t & 1
      const char* src = vbase + (int64_t)t * tile_bytes;
      const uint32_t dst = lds0 + (uint32_t)(2 * kTile + (t & 1) * kTile) + (uint32_t)wave * 4096u;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_piece(src, voff[i], dst + (uint32_t)i * 1024u);
    };
    auto kfrag = [&](const char* kbuf, int i) __attribute__((always_inline)) -> bf16x8_t {   // product i = 2 ks + kt
      return *reinterpret_cast<const bf16x8_t*>(kbuf + kro[i >> 1] + (i & 1) * 32 * FL_ROW_BYTES);
    };
    auto vfrag_half = [&](const char* vbuf, int i, int r) __attribute__((always_inline)) -> s16x4_t {   // product i = 4 kb + d
      return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vbuf + tro[i & 3][r] + (i >> 2) * 16 * FL_ROW_BYTES));
    };
    // does tile t need a mask for some query of this wave (diagonal / ragged end / padded keys)?
    auto edge_tile = [&](int t, uint64_t vw) -> bool {
      return (CAUSAL ? (t * 64 + 63 > q0) : (t * 64 + 64 > p.kv_len)) || (MASKED && vw != ~0ull);
    };
    auto apply_mask = [&](int t, uint64_t vw, f32x16_t& s0, f32x16_t& s1) __attribute__((always_inline)) {
      if (CAUSAL ? (t * 64 + 63 > q0) : (t * 64 + 64 > p.kv_len)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (!(CAUSAL ? key <= qi : key < p.kv_len)) (kt ? s1 : s0)[r] = -INFINITY;
          }
      }
      if (MASKED && vw != ~0ull) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const uint32_t w = flash_open_bits(vw, kt, g, qi - (t * 64 + kt * 32));
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!(w & (1u << ((r & 3) + 8 * (r >> 2))))) (kt ? s1 : s0)[r] = -INFINITY;
        }
      }
    };
    auto row_max = [&](const f32x16_t& s0, const f32x16_t& s1) __attribute__((always_inline)) -> float {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s0[r]), s0[r + 1]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s1[r]), s1[r + 1]);
      return mx;
    };
    auto finish_max = [&](float mx) __attribute__((always_inline)) -> float {
      return fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;   // c2 > 0: max of the scaled scores
    };
    // S^T of the tile at kbuf, nothing else (prologue / tile after a skipped one)
    auto s_plain = [&](const char* kbuf, f32x16_t& s0, f32x16_t& s1) __attribute__((always_inline)) {
      bf16x8_t kr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kr[i] = kfrag(kbuf, i);
      FL_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        if (i < 2) (i ? s1 : s0) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[0], (f32x16_t){0}, 0, 0, 0);
        else (i & 1 ? s1 : s0) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[i >> 1], (i & 1 ? s1 : s0), 0, 0, 0);
        if (i + 4 < 2 * KS) kr[i & 3] = kfrag(kbuf, i + 4);
        FL_FENCE();
      }
    };
    // probabilities of elements (e, e + 1) (in place) and their share of the row sum: packed exponent argument, two partial sums
    auto exp_pair = [&](f32x16_t& s0, f32x16_t& s1, int e, f32x2_t nm2, f32x2_t c22) __attribute__((always_inline)) {
      f32x16_t& s = (e < 16) ? s0 : s1;
      const int r = e & 15;
      const f32x2_t t = __builtin_elementwise_fma((f32x2_t){s[r], s[r + 1]}, c22, nm2);
      const f32x2_t pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      l2 += pv;
      s[r] = pv[0];
      s[r + 1] = pv[1];
    };
    // O += V P with the exponentials inside it: the operand of key block kb (elements 8 kb .. 8 kb + 7) is exponentiated and
    // converted while the four MFMAs of key block kb - 1 run (one element pair per MFMA: pk_fma, 2 exp, pk_add, cvt — the five
    // instructions an MFMA hides); only key block 0's eight exponentials run in front of the first MFMA.  V fragments by
    // transposing reads, a ring of four ahead of the MFMAs.
    auto exp_pv = [&](const char* vbuf, f32x16_t& c0, f32x16_t& c1, float m_new) __attribute__((always_inline)) {
      const f32x2_t nm2 = {-m_new, -m_new}, c22 = {c2, c2};
      auto pel = [&](int e) -> float { return e < 16 ? c0[e] : c1[e - 16]; };
      u32x4_v pf[2];
      s16x4_t vlo[4], vhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vlo[i] = vfrag_half(vbuf, i, 0);
        vhi[i] = vfrag_half(vbuf, i, 1);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        exp_pair(c0, c1, 2 * c, nm2, c22);
        pf[0][c] = cvt2(pel(2 * c), pel(2 * c + 1));
      }
      FL_FENCE();
#pragma unroll
      for (int i = 0; i < 4 * DT; ++i) {
        const s16x8_t vv = {vlo[i & 3][0], vlo[i & 3][1], vlo[i & 3][2], vlo[i & 3][3],
                            vhi[i & 3][0], vhi[i & 3][1], vhi[i & 3][2], vhi[i & 3][3]};
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv),
                                                              __builtin_bit_cast(bf16x8_t, pf[(i >> 2) & 1]), acc[i & 3], 0, 0, 0);
        if (i + 4 < 4 * DT) {
          vlo[i & 3] = vfrag_half(vbuf, i + 4, 0);
          vhi[i & 3] = vfrag_half(vbuf, i + 4, 1);
        }
        if (i < 12) {   // component c = i & 3 of the NEXT key block's operand: elements 8 (kb + 1) + 2 c, + 1
          const int e = 8 * ((i >> 2) + 1) + 2 * (i & 3);
          exp_pair(c0, c1, e, nm2, c22);
          pf[((i >> 2) + 1) & 1][i & 3] = cvt2(pel(e), pel(e + 1));
        }
        FL_FENCE();
      }
    };

    // ---- prologue: K(0), V(0) in flight
    uint8_t vb = MASKED ? kv_byte(p, b, 0, lane) : (uint8_t)1;
    __syncthreads();                    // the previous rep is done with every buffer
    issue_k(0);
    issue_v(0);
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of K(t), V(t) have landed
      __syncthreads();                                   // everybody's have; everybody is done with K(t - 1), V(t - 1)
      const uint64_t vw = MASKED ? __builtin_amdgcn_ballot_w64(vb != 0) : ~0ull;
      if (t + 1 < nt) {                                  // the next tile flies during this tile's products
        issue_k(t + 1);
        issue_v(t + 1);
        if (MASKED) vb = kv_byte(p, b, t + 1, lane);
      }
      if (t > t_last) continue;                                   // whole tile above this wave's diagonal
      if (MASKED && vw == 0 && t * 64 + 63 < q0) continue;        // a tile of padding below the diagonal: nothing to add
      const char* kc = smem + (t & 1) * kTile;
      const char* vc = smem + 2 * kTile + (t & 1) * kTile;
      f32x16_t s0, s1;
      s_plain(kc, s0, s1);
      if (edge_tile(t, vw)) apply_mask(t, vw, s0, s1);
      const float m_new = fmaxf(m, finish_max(row_max(s0, s1)));
      if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        l2 *= (f32x2_t){alpha, alpha};
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
      }
      m = m_new;
      exp_pv(vc, s0, s1, m_new);
    }
    const float l = l2[0] + l2[1];
    const float l_tot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l_tot;
    bf16_t* orow = out + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[d][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4_t*>(orow + d * 32 + 8 * qd + 4 * g) = o;
      }
    if (g == 0) lse_out[((int64_t)b * p.H + h) * p.S + qi] = (m + __builtin_amdgcn_logf(l_tot)) * (1.0f / LOG2E);
  }  // rep
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ on the same operand tiles: S^T = K Q^T and dP^T = V dO^T from row-major fragments of the K / V tiles, dS^T = P^T o (dP^T - D),
// dQ^T += K^T dS^T with K^T fragments read transposing out of the SAME K tile (flash_dq_kernel keeps a row-major K, a row-major
// V and a transposed K image, written through registers: 48 ds_write + ~90 vector instructions per tile and thread).  Lane <->
// element arrangement, products and accumulation order are flash_dq_kernel's: bit-identical results.  D = rowsum(dO o O) is
// computed in the prologue and handed to the dK/dV kernel through dvec, as there.
// ---------------------------------------------------------------------------------------------------------------------
template <bool CAUSAL, bool MASKED>
__global__ void __launch_bounds__(256, 2) flash_dq2_kernel(const FlashParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, j = lane & 31;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem));
  const int nqb = p.S / 128;
  const FlashBlock fb = flash_block_qh((int)blockIdx.x, (int)gridDim.x, flash_items(nqb, CAUSAL), p.H, p.HKV);
  const int b = fb.b, h = fb.h, hk = fb.hk;
  const int nrep = flash_pair_count(nqb, fb.blk, CAUSAL);
  const float c2 = p.scale * LOG2E;
  uint32_t voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = 4 * wave + i;
    voff[i] = (uint32_t)(fl_dma_row(piece, lane) * (int)(p.kv_ss * 2) + fl_dma_src_slot(piece, lane) * 16);
  }
  uint32_t kro[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kro[ks] = (uint32_t)fl_row_frag_off(0, ks, lane);
  uint32_t tro[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 2; ++r) tro[d][r] = (uint32_t)fl_tr_frag_off(0, 32 * d, r, lane);

  for (int rep = 0; rep < nrep; ++rep) {
    const int qb = flash_pair_q(nqb, fb.blk, rep, CAUSAL);
    const int q0 = qb * 128 + wave * 32;
    const int qi = q0 + j;
    const bf16_t* qrow = p.q + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
    const bf16_t* dorow = p.dout + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
    bf16x8_t qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + g * 8);
      dof[ks] = *reinterpret_cast<const bf16x8_t*>(dorow + ks * 16 + g * 8);
    }
    const int64_t st = ((int64_t)b * p.H + h) * p.S + qi;
    float dq_d = 0.f;
    {
      const bf16_t* orow = p.o + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(orow + ks * 16 + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dq_d += (float)of[e] * (float)dof[ks][e];
      }
      dq_d += __shfl_xor(dq_d, 32, 64);
      if (g == 0) p.dvec[st] = dq_d;
    }
    float lse2 = p.lse[st] * LOG2E;
    // every prologue load is awaited here, not by stale vmcnt waits inside the tile loop (see flash_fwd2_kernel)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
    asm volatile("" : "+v"(lse2), "+v"(dq_d));
    const char* kbase = uniform_ptr(reinterpret_cast<const char*>(p.k + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh));
    const char* vbase = uniform_ptr(reinterpret_cast<const char*>(p.v + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh));
    const int64_t tile_bytes = (int64_t)64 * p.kv_ss * 2;
    f32x16_t acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    const int nt = CAUSAL ? (qb * 128 + 128) / 64 : (p.kv_len + 63) / 64;
    const int t_last = CAUSAL ? (q0 + 31) / 64 : nt - 1;

    auto issue_kv = [&](int t) {   // K tile t -> K buffer t & 1, V tile t -> V buffer t & 1
      const char* ks_ = kbase + (int64_t)t * tile_bytes;
      const char* vs_ = vbase + (int64_t)t * tile_bytes;
      const uint32_t kd = lds0 + (uint32_t)((t & 1) * kTile) + (uint32_t)wave * 4096u;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_piece(ks_, voff[i], kd + (uint32_t)i * 1024u);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_piece(vs_, voff[i], kd + (uint32_t)(2 * kTile) + (uint32_t)i * 1024u);
    };
    uint8_t vb = MASKED ? kv_byte(p, b, 0, lane) : (uint8_t)1;
    __syncthreads();
    issue_kv(0);
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const uint64_t vw = MASKED ? __builtin_amdgcn_ballot_w64(vb != 0) : ~0ull;
      if (t + 1 < nt) {
        issue_kv(t + 1);
        if (MASKED) vb = kv_byte(p, b, t + 1, lane);
      }
      if (t > t_last) continue;                                   // whole tile above this wave's diagonal
      if (MASKED && vw == 0 && t * 64 + 63 < q0) continue;        // padding only, below the diagonal
      const char* kc = smem + (t & 1) * kTile;
      const char* vc = kc + 2 * kTile;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        // product i = 2 ks + w: w = 0 S^T (K rows x Q fragment), w = 1 dP^T (V rows x dO fragment); a ring of four row fragments
        auto rfrag = [&](int i) __attribute__((always_inline)) -> bf16x8_t {
          return *reinterpret_cast<const bf16x8_t*>(((i & 1) ? vc : kc) + kro[i >> 1] + kt * 32 * FL_ROW_BYTES);
        };
        f32x16_t sc, dp;
        bf16x8_t rf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rf[i] = rfrag(i);
        FL_FENCE();
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
          if (i == 0) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[0], qf[0], (f32x16_t){0}, 0, 0, 0);
          else if (i == 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[1], dof[0], (f32x16_t){0}, 0, 0, 0);
          else if (i & 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], dof[i >> 1], dp, 0, 0, 0);
          else sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], qf[i >> 1], sc, 0, 0, 0);
          if (i + 4 < 2 * KS) rf[i & 3] = rfrag(i + 4);
          FL_FENCE();
        }
        // mask only where the half tile can hold a masked key for some query of the wave
        if (CAUSAL ? (t * 64 + kt * 32 + 31 > q0) : (t * 64 + kt * 32 + 32 > p.kv_len)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (!(CAUSAL ? key <= qi : key < p.kv_len)) sc[r] = -INFINITY;   // exp2(-inf) = 0
          }
        }
        if (MASKED && vw != ~0ull) {
          const uint32_t w = flash_open_bits(vw, kt, g, qi - (t * 64 + kt * 32));
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!(w & (1u << ((r & 3) + 8 * (r >> 2))))) sc[r] = -INFINITY;
        }
        // dS^T = P^T (dP^T - D); dQ^T += K^T dS^T: product i = 4 kb + d, the operand of key block kb = elements 8 kb .. + 7.
        // The second key block's eight elements are exponentiated while the first block's four MFMAs run.
        auto ds_pair = [&](int e) __attribute__((always_inline)) -> uint32_t {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c2, -lse2));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e + 1], c2, -lse2));
          return cvt2(p0 * (dp[e] - dq_d), p1 * (dp[e + 1] - dq_d));
        };
        u32x4_v pf[2];
        s16x4_t tlo[4], thi[4];
        auto tfrag = [&](int i, int r) __attribute__((always_inline)) -> s16x4_t {
          return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_tr_ptr)(kc + tro[i & 3][r] + (kt * 32 + 16 * (i >> 2)) * FL_ROW_BYTES));
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tlo[i] = tfrag(i, 0);
          thi[i] = tfrag(i, 1);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) pf[0][c] = ds_pair(2 * c);
        FL_FENCE();
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) {
          const s16x8_t tv = {tlo[i & 3][0], tlo[i & 3][1], tlo[i & 3][2], tlo[i & 3][3],
                              thi[i & 3][0], thi[i & 3][1], thi[i & 3][2], thi[i & 3][3]};
          acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, tv),
                                                                __builtin_bit_cast(bf16x8_t, pf[i >> 2]), acc[i & 3], 0, 0, 0);
          if (i + 4 < 2 * DT) {
            tlo[i & 3] = tfrag(i + 4, 0);
            thi[i & 3] = tfrag(i + 4, 1);
          }
          if (i < 4) pf[1][i] = ds_pair(8 + 2 * i);
          FL_FENCE();
        }
      }
    }
    bf16_t* outp = p.dq + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[d][4 * qd + e] * p.scale);
        *reinterpret_cast<bf16x4_t*>(outp + d * 32 + 8 * qd + 4 * g) = o;
      }
  }  // rep
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// (A dK / dV kernel on the same LDS-DMA tiles existed up to round 5 — flash_dkdv2_kernel, CMB_KNOB_FLASH bit 8, bit-identical and
// ~200 us per layer SLOWER than flash_dkdv_kernel<..., PIPE, TR>: at 460 registers hipcc moved the accumulators between the
// register halves every tile, profiles/r05_lab.md.  It was never the default and is removed in round 6: git history, commit a4e275e.)

}  // namespace

int launch_flash_fwd2(const FlashParams& p, bf16_t* out, float* lse, bool causal, hipStream_t stream) {
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    bool ok = true;
#define FWD2_ATTR(C_, M_)                                                                              \
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd2_kernel<C_, M_>),             \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kSmemFwd) == hipSuccess
    FWD2_ATTR(true, true); FWD2_ATTR(true, false); FWD2_ATTR(false, false);
#undef FWD2_ATTR
    if (!ok) return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  const int64_t nqb = p.S / 128;
  const dim3 grid((unsigned)((int64_t)flash_items((int)nqb, causal) * p.H * p.B));   // 1-D: flash_map.h
  if (causal && p.key_valid) hipLaunchKernelGGL((flash_fwd2_kernel<true, true>), grid, dim3(256), kSmemFwd, stream, p, out, lse);
  else if (causal) hipLaunchKernelGGL((flash_fwd2_kernel<true, false>), grid, dim3(256), kSmemFwd, stream, p, out, lse);
  else hipLaunchKernelGGL((flash_fwd2_kernel<false, false>), grid, dim3(256), kSmemFwd, stream, p, out, lse);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}


int launch_flash_dq2(const FlashParams& p, bool causal, hipStream_t stream) {
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    bool ok = true;
#define DQ2_ATTR(C_, M_)                                                                              \
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(flash_dq2_kernel<C_, M_>),              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kSmemFwd) == hipSuccess
    DQ2_ATTR(true, true); DQ2_ATTR(true, false); DQ2_ATTR(false, false);
#undef DQ2_ATTR
    if (!ok) return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  const int64_t nqb = p.S / 128;
  const dim3 grid((unsigned)((int64_t)flash_items((int)nqb, causal) * p.H * p.B));
  if (causal && p.key_valid) hipLaunchKernelGGL((flash_dq2_kernel<true, true>), grid, dim3(256), kSmemFwd, stream, p);
  else if (causal) hipLaunchKernelGGL((flash_dq2_kernel<true, false>), grid, dim3(256), kSmemFwd, stream, p);
  else hipLaunchKernelGGL((flash_dq2_kernel<false, false>), grid, dim3(256), kSmemFwd, stream, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}


}  // namespace cmb_flash
