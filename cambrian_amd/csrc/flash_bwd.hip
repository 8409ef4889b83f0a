// flash_bwd.hip — causal self-attention with grouped KV heads (the LLM decoder's attention: S = 2048, head_dim 128,
// 32 query / 8 KV heads), forward and backward, for gfx950.  bf16 in / out, fp32 accumulation.
//
// Forward: flash_fwd_kernel (below).  Backward: two kernels, no atomics, both on v_mfma_f32_32x32x16_bf16 with the operand arrangement of vit_attn.hip
// (accumulator -> operand hand-off without shuffles: the MFMA's contraction slots are assigned to rows in exactly
// the order the previous product's accumulator holds them):
//   * flash_dq_kernel   — a workgroup owns 128 queries of one (batch, head), a LANE owns one query and walks the key
//     tiles up to the diagonal:  S^T = K·Q^T, dP^T = V·dO^T (K, V rows from LDS, Q, dO fragments in registers),
//     dS^T = P^T∘(dP^T − D), dQ^T += K^T·dS^T (K^T image in LDS);
//   * flash_dkdv_kernel — a workgroup owns 128 keys of one (batch, KV head), a LANE owns one key and walks the query
//     tiles from the diagonal on, for the 4 query heads of the group:  S = Q·K^T, dP = dO·V^T (Q, dO rows from LDS, K,
//     V fragments in registers), dV^T += dO^T·P, dK^T += Q^T·dS (transposed Q / dO images in LDS).
// P is recomputed from the forward's log-sum-exp (natural log of the sum of exp(scale·q·k)); D = rowsum(dO∘O) is computed
// in the dQ kernel's prologue (each lane owns a query row) and handed to the dK/dV kernel through dvec.  7 tile products instead of the 5 of a single-pass backward, nothing is atomically
// accumulated, results are bit-reproducible.
// Layout: all of q, k, v, o, do, dq, dk, dv are addressed as [B, S, H, 128] through (batch, token, head) element strides
// (token-major storage, what ops.qkv_rope produces and the attention returns); lse / D are fp32 [B, H, S].
#include "flash_common.h"
#include "flash_layout.h"

using namespace cmb_flash;

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define cvt8 cvt8_bf16

// A 64-row x 128 tile moves global -> registers -> LDS in two halves so that the global loads of the NEXT tile can be
// in flight while the current one is being multiplied (tile_load after the barrier that publishes the current tile,
// tile_put after the barrier that retires it).  Thread <-> (row pair rp, d group dg) x 2 passes: 4 consecutive lanes
// cover 64 contiguous bytes of a row (global coalescing), 16 lane groups take consecutive row pairs (the transposed
// 32-bit stores then hit consecutive banks, 2-way at most: free for ds_write_b32).
struct TileRegs {
  bf16x8_t a[2], b[2];
};

__device__ __forceinline__ void tile_load(TileRegs& r, const bf16_t* src, int64_t rs, int row0, int nrows_valid, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;       // 0..511
    const int dg = ((id >> 6) & 3) * 4 + (id & 3);     // d group 0..15
    const int rp = (id >> 8) * 16 + ((id >> 2) & 15);  // row pair 0..31
    int r0 = row0 + 2 * rp, r1 = r0 + 1;
    r0 = r0 < nrows_valid ? r0 : nrows_valid - 1;
    r1 = r1 < nrows_valid ? r1 : nrows_valid - 1;
    r.a[i] = *reinterpret_cast<const bf16x8_t*>(src + (int64_t)r0 * rs + dg * 8);
    r.b[i] = *reinterpret_cast<const bf16x8_t*>(src + (int64_t)r1 * rs + dg * 8);
  }
}

// row-major LDS image (stride LDR) and / or the transposed image [128][LDT] (pairs of adjacent rows packed as one
// 32-bit store)
template <bool ROWMAJOR, bool TR>
__device__ __forceinline__ void tile_put(const TileRegs& r, bf16_t* sR, bf16_t* sT, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int dg = ((id >> 6) & 3) * 4 + (id & 3);
    const int rp = (id >> 8) * 16 + ((id >> 2) & 15);
    if (ROWMAJOR) {
      *reinterpret_cast<bf16x8_t*>(sR + (2 * rp) * LDR + dg * 8) = r.a[i];
      *reinterpret_cast<bf16x8_t*>(sR + (2 * rp + 1) * LDR + dg * 8) = r.b[i];
    }
    if (TR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bf16x2_t pr;
        pr[0] = r.a[i][e];
        pr[1] = r.b[i][e];
        *reinterpret_cast<bf16x2_t*>(sT + (dg * 8 + e) * LDT + 2 * rp) = pr;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward: grid S/128 * H * B (1-D, flash_block_qh), 256 threads, a lane owns one query (the dQ kernel's arrangement): S^T = K·Q^T from the
// row-major K tile, online softmax in base 2 (row max / sum need one lane^32 exchange), O^T += V^T·P^T with P going
// accumulator -> operand without shuffles and V^T fragments from the transposed V image.  K/V heads are shared by the
// query heads of a group without being expanded.  Writes O (bf16) and lse = log sum_j exp(scale q.k_j) (fp32).
// ------------------------------------------------------------------------------------------------------------------
// MASKED: a key-padding mask is given (causal only).  A separate instantiation: the unmasked kernels carry none of the mask's
// registers or code (with a run-time pointer test instead they ran 3-8 % slower although every tile took the all-valid path).
// PIPE (round 5, knob CMB_KNOB_FLASH = 1): the LDS fragment reads of the two products run a ring of four fragments AHEAD of
// the MFMAs that consume them, in a fixed order (FLASH_FENCE = scheduling barrier).  Left to itself hipcc reuses eight fragment
// registers: ds_read x 2, s_waitcnt, MFMA, s_waitcnt, MFMA, then the next two reads — every pair of MFMAs waits out a full
// LDS round trip, and all eight products of a 32-key half chain on one accumulator (the assembly of the round-4 kernel,
// profiles/r05_lab.md).  Here read i + 4 is issued right behind MFMA i and consecutive MFMAs alternate between the two
// halves' accumulators.  Same products, same accumulation order per accumulator: bit-identical results.
// (A cross-tile variant — S of tile t + 1 in the same basic block as the exponentials of tile t, two K buffers — was built
// first and measured SLOWER, 460 vs 397 us at 8 x 2048 tokens: 256 registers with spills, and the compiler's interleave still
// waited for every read.)
#define FLASH_FENCE() __builtin_amdgcn_sched_barrier(0)
template <bool CAUSAL, bool MASKED, bool PIPE>
__global__ void __launch_bounds__(256, 2) flash_fwd_kernel(const FlashParams p, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse_out) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[64 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t sVT[HD * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, j = lane & 31;
  // Causal: query block qb walks (qb + 1) * 2 key tiles, so a workgroup takes the PAIR (i, nqb-1-i) one after the
  // other — uniform work per workgroup, no tail.
  const int nqb = p.S / 128;
  const FlashBlock fb = flash_block_qh((int)blockIdx.x, (int)gridDim.x, flash_items(nqb, CAUSAL), p.H, p.HKV);
  const int b = fb.b, h = fb.h, hk = fb.hk;
  const int nrep = flash_pair_count(nqb, fb.blk, CAUSAL);
  for (int rep = 0; rep < nrep; ++rep) {
  const int qb = flash_pair_q(nqb, fb.blk, rep, CAUSAL);
  const int q0 = qb * 128 + wave * 32;
  const int qi = q0 + j;
  const bf16_t* qrow = p.q + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + g * 8);
  const float c2 = p.scale * LOG2E;
  const bf16_t* kbase = p.k + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh;
  const bf16_t* vbase = p.v + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh;
  f32x16_t acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nt = CAUSAL ? (qb * 128 + 128) / 64 : (p.kv_len + 63) / 64;
  TileRegs rk, rv;
  tile_load(rk, kbase, p.kv_ss, 0, p.S, tid);
  tile_load(rv, vbase, p.kv_ss, 0, p.S, tid);
  uint8_t vb = MASKED ? kv_byte(p, b, 0, lane) : (uint8_t)1;  // key-padding byte of this lane's key of the NEXT tile
  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // previous tile consumed
    tile_put<true, false>(rk, sK, nullptr, tid);
    tile_put<false, true>(rv, nullptr, sVT, tid);
    __syncthreads();
    const uint64_t vw = MASKED ? __builtin_amdgcn_ballot_w64(vb != 0) : ~0ull;  // validity of this tile's 64 keys
    if (t + 1 < nt) {  // the next tile's loads fly during this tile's MFMAs
      tile_load(rk, kbase, p.kv_ss, (t + 1) * 64, p.S, tid);
      tile_load(rv, vbase, p.kv_ss, (t + 1) * 64, p.S, tid);
      if (MASKED) vb = kv_byte(p, b, t + 1, lane);
    }
    if (CAUSAL && t * 64 > q0 + 31) continue;
    if (MASKED && vw == 0 && t * 64 + 63 < q0) continue;  // a tile of padding below the wave's diagonal: nothing to add
    f32x16_t s[2];
    if (PIPE) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      // product i = 2 ks + kt: K rows of half kt, k-step ks
      const bf16_t* kb0 = sK + j * LDR + g * 8;
      bf16x8_t kr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kr[i] = *reinterpret_cast<const bf16x8_t*>(kb0 + (i & 1) * 32 * LDR + (i >> 1) * 16);
      FLASH_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[i >> 1], s[i & 1], 0, 0, 0);
        if (i + 4 < 2 * KS)
          kr[i & 3] = *reinterpret_cast<const bf16x8_t*>(kb0 + ((i + 4) & 1) * 32 * LDR + ((i + 4) >> 1) * 16);
        FLASH_FENCE();
      }
    } else {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + (kt * 32 + j) * LDR + ks * 16 + g * 8);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    }
    // VALU budget: this block runs once per 32 MFMAs of the wave, so every instruction per score element counts.
    // The mask is applied only on tiles that can hold a masked key for some query of the wave (the diagonal tiles /
    // the tile with the padding), scores stay un-scaled until the exponent (one fma per element feeds exp2), the
    // accumulators are rescaled only when some lane's running maximum moved, P is converted two elements at a time.
    if (CAUSAL ? (t * 64 + 63 > q0) : (t * 64 + 64 > p.kv_len)) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (!(CAUSAL ? key <= qi : key < p.kv_len)) s[kt][r] = -INFINITY;
        }
    }
    if (MASKED && vw != ~0ull) {  // a tile with padded keys (wave-uniform: most tiles skip this)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const uint32_t w = flash_open_bits(vw, kt, g, qi - (t * 64 + kt * 32));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(w & (1u << ((r & 3) + 8 * (r >> 2))))) s[kt][r] = -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;   // c2 > 0: max of the scaled scores
    const float m_new = fmaxf(m, mx);
    if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
    }
    m = m_new;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c2, -m_new));
        l += pv;
        s[kt][r] = pv;
      }
    if (PIPE) {
      bf16x8_t pf[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int kt = kb >> 1, hh = kb & 1;
        pf[kb] = cvt8(s[kt][8 * hh + 0], s[kt][8 * hh + 1], s[kt][8 * hh + 2], s[kt][8 * hh + 3], s[kt][8 * hh + 4],
                      s[kt][8 * hh + 5], s[kt][8 * hh + 6], s[kt][8 * hh + 7]);
      }
      // product i = 4 kb + d: V^T rows of d tile d, keys 16 kb .. 16 kb + 15 of the tile (two 8-byte reads per fragment)
      const bf16_t* vb0 = sVT + j * LDT + 4 * g;
      bf16x4_t vlo[4], vhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vlo[i] = *reinterpret_cast<const bf16x4_t*>(vb0 + (i & 3) * 32 * LDT + (i >> 2) * 16);
        vhi[i] = *reinterpret_cast<const bf16x4_t*>(vb0 + (i & 3) * 32 * LDT + (i >> 2) * 16 + 8);
      }
      FLASH_FENCE();
#pragma unroll
      for (int i = 0; i < 4 * DT; ++i) {
        bf16x8_t vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = vlo[i & 3][e]; vf[4 + e] = vhi[i & 3][e]; }
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[i >> 2], acc[i & 3], 0, 0, 0);  // O^T[d][query]
        if (i + 4 < 4 * DT) {
          vlo[i & 3] = *reinterpret_cast<const bf16x4_t*>(vb0 + ((i + 4) & 3) * 32 * LDT + ((i + 4) >> 2) * 16);
          vhi[i & 3] = *reinterpret_cast<const bf16x4_t*>(vb0 + ((i + 4) & 3) * 32 * LDT + ((i + 4) >> 2) * 16 + 8);
        }
        FLASH_FENCE();
      }
    } else {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int kt = kb >> 1, hh = kb & 1;
      const bf16x8_t pf = cvt8(s[kt][8 * hh + 0], s[kt][8 * hh + 1], s[kt][8 * hh + 2], s[kt][8 * hh + 3],
                               s[kt][8 * hh + 4], s[kt][8 * hh + 5], s[kt][8 * hh + 6], s[kt][8 * hh + 7]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const bf16_t* vrow = sVT + (d * 32 + j) * LDT + kt * 32 + 16 * hh + 4 * g;
        const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(vrow);
        const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(vrow + 8);
        bf16x8_t vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc[d], 0, 0, 0);  // O^T[d][query]
      }
    }
    }
  }
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
}

// ------------------------------------------------------------------------------------------------------------------
// dQ: grid S/128 * H * B (1-D, flash_block_qh), 256 threads.  lane = (query j = lane & 31 of the wave's 32, half g = lane >> 5).
// ------------------------------------------------------------------------------------------------------------------
// MASKED: a key-padding mask is given (causal only).  A separate instantiation: the unmasked kernels carry none of the mask's
// registers or code (with a run-time pointer test instead they ran 3-8 % slower although every tile took the all-valid path).
template <bool CAUSAL, bool MASKED, bool PIPE>
__global__ void __launch_bounds__(256, (CAUSAL && !PIPE) ? 2 : 1) flash_dq_kernel(const FlashParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[64 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t sV[64 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t sKT[HD * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, j = lane & 31;
  const int nqb = p.S / 128;   // causal: pairs (i, nqb-1-i), as in the forward
  const FlashBlock fb = flash_block_qh((int)blockIdx.x, (int)gridDim.x, flash_items(nqb, CAUSAL), p.H, p.HKV);
  const int b = fb.b, h = fb.h, hk = fb.hk;
  const int nrep = flash_pair_count(nqb, fb.blk, CAUSAL);
  for (int rep = 0; rep < nrep; ++rep) {
  const int qb = flash_pair_q(nqb, fb.blk, rep, CAUSAL);
  const int q0 = qb * 128 + wave * 32;
  const int qi = q0 + j;                                  // this lane's query (S % 128 == 0: always valid)
  const bf16_t* qrow = p.q + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
  const bf16_t* dorow = p.dout + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
  bf16x8_t qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + g * 8);
    dof[ks] = *reinterpret_cast<const bf16x8_t*>(dorow + ks * 16 + g * 8);
  }
  const int64_t st = ((int64_t)b * p.H + h) * p.S + qi;
  // D = rowsum(dO * O) of this lane's query: the lane holds half of the dO row (the other half sits in lane ^ 32), so the
  // O row is read once here and D also goes to dvec for the dK/dV kernel, which runs after this one on the stream —
  // no separate pre-pass over O and dO.
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
  const float lse2 = p.lse[st] * LOG2E;
  const float c2 = p.scale * LOG2E;
  const bf16_t* kbase = p.k + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh;
  const bf16_t* vbase = p.v + (int64_t)b * p.kv_sb + (int64_t)hk * p.kv_sh;

  f32x16_t acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int nt = CAUSAL ? (qb * 128 + 128) / 64   // key tiles up to and including the diagonal ones
                        : (p.kv_len + 63) / 64;   // every tile that holds a real key
  TileRegs rk, rv;
  tile_load(rk, kbase, p.kv_ss, 0, p.S, tid);
  tile_load(rv, vbase, p.kv_ss, 0, p.S, tid);
  uint8_t vb = MASKED ? kv_byte(p, b, 0, lane) : (uint8_t)1;
  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // previous tile fully consumed
    tile_put<true, true>(rk, sK, sKT, tid);
    tile_put<true, false>(rv, sV, nullptr, tid);
    __syncthreads();
    const uint64_t vw = MASKED ? __builtin_amdgcn_ballot_w64(vb != 0) : ~0ull;
    if (t + 1 < nt) {  // the next tile's loads fly during this tile's MFMAs
      tile_load(rk, kbase, p.kv_ss, (t + 1) * 64, p.S, tid);
      tile_load(rv, vbase, p.kv_ss, (t + 1) * 64, p.S, tid);
      if (MASKED) vb = kv_byte(p, b, t + 1, lane);
    }
    if (CAUSAL && t * 64 > q0 + 31) continue;  // whole tile above this wave's diagonal (block-uniform barriers stay matched)
    if (MASKED && vw == 0 && t * 64 + 63 < q0) continue;  // padding only, below the diagonal
    auto sdp = [&](int kt, f32x16_t& s, f32x16_t& dp) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + (kt * 32 + j) * LDR + ks * 16 + g * 8);
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + (kt * 32 + j) * LDR + ks * 16 + g * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);      // S^T[key][query]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);   // dP^T[key][query]
      }
    };
    auto mask_s = [&](int kt, f32x16_t& s) __attribute__((always_inline)) {
      // mask only where the tile can hold a masked key for some query of the wave; one fma feeds exp2
      if (CAUSAL ? (t * 64 + kt * 32 + 31 > q0) : (t * 64 + kt * 32 + 32 > p.kv_len)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (!(CAUSAL ? key <= qi : key < p.kv_len)) s[r] = -INFINITY;   // exp2(-inf) = 0
        }
      }
      if (MASKED && vw != ~0ull) {  // padded keys in this tile
        const uint32_t w = flash_open_bits(vw, kt, g, qi - (t * 64 + kt * 32));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(w & (1u << ((r & 3) + 8 * (r >> 2))))) s[r] = -INFINITY;
      }
    };
    auto ds_dq = [&](int kt, f32x16_t& s, const f32x16_t& dp) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lse2));
        s[r] = pr * (dp[r] - dq_d);  // dS^T (the 1/sqrt(d) factor is applied once at the end)
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t pf = cvt8(s[8 * kb + 0], s[8 * kb + 1], s[8 * kb + 2], s[8 * kb + 3], s[8 * kb + 4], s[8 * kb + 5],
                                 s[8 * kb + 6], s[8 * kb + 7]);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bf16_t* trow = sKT + (d * 32 + j) * LDT + kt * 32 + 16 * kb + 4 * g;
          const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(trow);
          const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(trow + 8);
          bf16x8_t tf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { tf[e] = lo[e]; tf[4 + e] = hi[e]; }
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf, acc[d], 0, 0, 0);  // dQ^T[d][query]
        }
      }
    };
    const bool edge = (CAUSAL ? (t * 64 + 63 > q0) : (t * 64 + 64 > p.kv_len)) || (MASKED && vw != ~0ull);
    if (PIPE && !edge) {
      // Round 5 (knob CMB_KNOB_FLASH = 1), interior tiles: the dK/dV kernel's arrangement (see there) — four phases in a fixed
      // order, every MFMA followed by the LDS read of the fragment four products ahead and by a slice of the other half's
      // exponentials:  A  S0, dP0   B  S1, dP1 + dS0   C  dQ0 + dS1 (two elements per product)   D  dQ1.
      // This variant is compiled for one workgroup per CU (the second set of score registers does not fit 256).
      f32x16_t s0, dp0, s1, dp1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; dp0[r] = 0.f; s1[r] = 0.f; dp1[r] = 0.f; }
      // product i = 2 ks + w of a score phase: w = 0 S^T (K rows x Q fragment), w = 1 dP^T (V rows x dO fragment)
      auto row_frag = [&](int kt, int i) __attribute__((always_inline)) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(((i & 1) ? sV : sK) + (kt * 32 + j) * LDR + (i >> 1) * 16 + g * 8);
      };
      // product i = 4 kb + d of a dQ phase: K^T rows of d tile d, keys 16 kb .. of the half
      auto tr_ptr = [&](int kt, int i) __attribute__((always_inline)) -> const bf16_t* {
        return sKT + ((i & 3) * 32 + j) * LDT + kt * 32 + 16 * (i >> 2) + 4 * g;
      };
      auto elem = [&](int r, f32x16_t& s_, const f32x16_t& dp_) __attribute__((always_inline)) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s_[r], c2, -lse2));
        s_[r] = pr * (dp_[r] - dq_d);
      };
      bf16x8_t rf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rf[i] = row_frag(0, i);
      FLASH_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {   // A
        if (i & 1) dp0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], dof[i >> 1], dp0, 0, 0, 0);
        else s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], qf[i >> 1], s0, 0, 0, 0);
        rf[i & 3] = row_frag(i + 4 < 2 * KS ? 0 : 1, (i + 4) & (2 * KS - 1));
        FLASH_FENCE();
      }
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {   // B
        if (i & 1) dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], dof[i >> 1], dp1, 0, 0, 0);
        else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], qf[i >> 1], s1, 0, 0, 0);
        if (i + 4 < 2 * KS) rf[i & 3] = row_frag(1, i + 4);
        elem(i, s0, dp0);
        FLASH_FENCE();
      }
      bf16x8_t pf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        pf[kb] = cvt8(s0[8 * kb + 0], s0[8 * kb + 1], s0[8 * kb + 2], s0[8 * kb + 3], s0[8 * kb + 4], s0[8 * kb + 5],
                      s0[8 * kb + 6], s0[8 * kb + 7]);
      bf16x4_t tlo[4], thi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16_t* q_ = tr_ptr(0, i);
        tlo[i] = *reinterpret_cast<const bf16x4_t*>(q_);
        thi[i] = *reinterpret_cast<const bf16x4_t*>(q_ + 8);
      }
      FLASH_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * DT; ++i) {   // C
        bf16x8_t tf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { tf[e] = tlo[i & 3][e]; tf[4 + e] = thi[i & 3][e]; }
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[i >> 2], acc[i & 3], 0, 0, 0);
        {
          const bf16_t* q_ = tr_ptr(i + 4 < 2 * DT ? 0 : 1, (i + 4) & (2 * DT - 1));
          tlo[i & 3] = *reinterpret_cast<const bf16x4_t*>(q_);
          thi[i & 3] = *reinterpret_cast<const bf16x4_t*>(q_ + 8);
        }
        elem(2 * i, s1, dp1);
        elem(2 * i + 1, s1, dp1);
        FLASH_FENCE();
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        pf[kb] = cvt8(s1[8 * kb + 0], s1[8 * kb + 1], s1[8 * kb + 2], s1[8 * kb + 3], s1[8 * kb + 4], s1[8 * kb + 5],
                      s1[8 * kb + 6], s1[8 * kb + 7]);
      FLASH_FENCE();
#pragma unroll
      for (int i = 0; i < 2 * DT; ++i) {   // D
        bf16x8_t tf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { tf[e] = tlo[i & 3][e]; tf[4 + e] = thi[i & 3][e]; }
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[i >> 2], acc[i & 3], 0, 0, 0);
        if (i + 4 < 2 * DT) {
          const bf16_t* q_ = tr_ptr(1, i + 4);
          tlo[i & 3] = *reinterpret_cast<const bf16x4_t*>(q_);
          thi[i & 3] = *reinterpret_cast<const bf16x4_t*>(q_ + 8);
        }
        FLASH_FENCE();
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16_t s, dp;
        sdp(kt, s, dp);
        mask_s(kt, s);
        ds_dq(kt, s, dp);
      }
    }
  }
  bf16_t* out = p.dq + (int64_t)b * p.q_sb + (int64_t)qi * p.q_ss + (int64_t)h * p.q_sh;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      bf16x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[d][4 * qd + e] * p.scale);
      *reinterpret_cast<bf16x4_t*>(out + d * 32 + 8 * qd + 4 * g) = o;
    }
  }  // rep
}

// ------------------------------------------------------------------------------------------------------------------
// dK, dV: grid (key-block pairs | key blocks) * HKV * B (1-D, flash_block_kv), 256 threads.  lane = (key j of the wave's 32, half g).  One workgroup per CU (the
// accumulators + K/V fragments need ~350 registers), so the HBM/L2 latency of the next query tile is hidden inside the
// workgroup by a register prefetch (see the loop).  (A double-buffered-LDS variant with one barrier per tile measured
// slower: 506 registers, values shuffled through the accumulator file.)
// ------------------------------------------------------------------------------------------------------------------
// MASKED: a key-padding mask is given (causal only).  A separate instantiation: the unmasked kernels carry none of the mask's
// registers or code (with a run-time pointer test instead they ran 3-8 % slower although every tile took the all-valid path).
// TR (round 5, knob bit 16): the transposed fragments (dO^T for dV, Q^T for dK) are read out of the ROW-major Q / dO images with
// ds_read_b64_tr_b16 instead of out of transposed copies: the 64 ds_write_b32 and the ~100 16-bit shuffles per tile and thread
// that built those copies are gone.  The row-major images are then un-padded and XOR-swizzled (flash_layout.h: 256-byte rows, slot ^
// fl_swz(row)) — on the padded 272-byte rows a transposing read is 2-way bank-conflicted (the four rows of a [4][16] block start 16
// bytes apart; measured: SQ_LDS_BANK_CONFLICT 54 % of the LDS-active cycles) — so that row-major AND transposing reads are conflict-free.
template <bool CAUSAL, bool MASKED, bool PIPE, bool TR>
__global__ void __launch_bounds__(256) flash_dkdv_kernel(const FlashParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem_raw);   // [64][LDR]
  bf16_t* sDO = sQ + 64 * LDR;                         // [64][LDR]
  bf16_t* sQT = sDO + 64 * LDR;                        // [128][LDT]
  bf16_t* sDOT = sQT + HD * LDT;                       // [128][LDT]
  float* sLse = reinterpret_cast<float*>(sDOT + HD * LDT);  // [64] (already * log2 e)
  float* sD = sLse + 64;                                    // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, j = lane & 31;
  // TR: byte offsets of this lane's fragments inside a swizzled 64 x 128 image (flash_layout.h)
  int kro_[KS], tro_[DT][2];
  if (TR) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kro_[ks] = fl_row_frag_off(0, ks, lane);
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 2; ++r) tro_[d][r] = fl_tr_frag_off(0, 32 * d, r, lane);
  }
  // Causal: key block kb meets (nkb - kb) * 2 query tiles per head, so a workgroup takes the PAIR (i, nkb-1-i) one
  // after the other — every workgroup does the same amount of work and the grid has no tail.
  const int nkb = p.S / 128;
  const FlashBlock fb = flash_block_kv((int)blockIdx.x, (int)gridDim.x, flash_items(nkb, CAUSAL), p.HKV);
  const int b = fb.b, hk = fb.hk;
  const int nrep = flash_pair_count(nkb, fb.blk, CAUSAL);
  for (int rep = 0; rep < nrep; ++rep) {
  const int kb = flash_pair_k(nkb, fb.blk, rep, CAUSAL);
  const int k0 = kb * 128 + wave * 32;
  const int ki = k0 + j;
  const int group = p.H / p.HKV;
  const bf16_t* krow = p.k + (int64_t)b * p.kv_sb + (int64_t)ki * p.kv_ss + (int64_t)hk * p.kv_sh;
  const bf16_t* vrow = p.v + (int64_t)b * p.kv_sb + (int64_t)ki * p.kv_ss + (int64_t)hk * p.kv_sh;
  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8_t*>(krow + ks * 16 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8_t*>(vrow + ks * 16 + g * 8);
  }
  const float c2 = p.scale * LOG2E;
  // key-padding mask: this lane's key is padded -> only its own query (the open diagonal) contributes
  const bool kvalid = !MASKED || p.key_valid[(int64_t)b * p.S + ki] != 0;
  const bool wave_padded = MASKED && __builtin_amdgcn_ballot_w64(!kvalid) != 0;
  f32x16_t adk[DT], adv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { adk[d][r] = 0.f; adv[d][r] = 0.f; }

  const int qt0 = CAUSAL ? (kb * 128) / 64 : 0, nqt = p.S / 64;
  // Register prefetch: the global loads of the NEXT (head, query tile) are issued right after the barrier that
  // publishes the current tile and stay in flight during its 64 MFMAs; they are written to LDS after the next barrier.
  // Thread <-> (row pair rp, d group dg) x 2 passes, as in stage_tile64.
  const int sid1 = tid + 256;
  const int dg0 = ((tid >> 6) & 3) * 4 + (tid & 3), rp0 = (tid >> 8) * 16 + ((tid >> 2) & 15);
  const int dg1 = ((sid1 >> 6) & 3) * 4 + (sid1 & 3), rp1 = (sid1 >> 8) * 16 + ((sid1 >> 2) & 15);
  const int64_t o00 = (int64_t)(2 * rp0) * p.q_ss + dg0 * 8, o01 = o00 + p.q_ss;
  const int64_t o10 = (int64_t)(2 * rp1) * p.q_ss + dg1 * 8, o11 = o10 + p.q_ss;
  bf16x8_t rq00, rq01, rq10, rq11, rg00, rg01, rg10, rg11;
  float rl = 0.f;
#define DKDV_LOAD(h_, qt_)                                                                                    \
  do {                                                                                                        \
    const bf16_t* qb_ = p.q + (int64_t)b * p.q_sb + (int64_t)(h_) * p.q_sh + (int64_t)((qt_) * 64) * p.q_ss;   \
    const bf16_t* gb_ = p.dout + (int64_t)b * p.q_sb + (int64_t)(h_) * p.q_sh + (int64_t)((qt_) * 64) * p.q_ss; \
    rq00 = *reinterpret_cast<const bf16x8_t*>(qb_ + o00);                                                     \
    rq01 = *reinterpret_cast<const bf16x8_t*>(qb_ + o01);                                                     \
    rq10 = *reinterpret_cast<const bf16x8_t*>(qb_ + o10);                                                     \
    rq11 = *reinterpret_cast<const bf16x8_t*>(qb_ + o11);                                                     \
    rg00 = *reinterpret_cast<const bf16x8_t*>(gb_ + o00);                                                     \
    rg01 = *reinterpret_cast<const bf16x8_t*>(gb_ + o01);                                                     \
    rg10 = *reinterpret_cast<const bf16x8_t*>(gb_ + o10);                                                     \
    rg11 = *reinterpret_cast<const bf16x8_t*>(gb_ + o11);                                                     \
    const int64_t st_ = ((int64_t)b * p.H + (h_)) * p.S + (qt_) * 64;                                         \
    /* raw values only: any arithmetic on a loaded value here would force s_waitcnt vmcnt(0) before the MFMAs */ \
    if (tid < 64) rl = p.lse[st_ + tid];                                                                      \
    else if (tid < 128) rl = p.dvec[st_ + tid - 64];                                                          \
  } while (0)
#define DKDV_PUT(sR_, sT_, a_, b_, rp_, dg_)                                          \
  do {                                                                                \
    if (TR) {                                                                         \
      *reinterpret_cast<bf16x8_t*>((sR_) + (2 * (rp_)) * 128 + (((dg_) ^ fl_swz(2 * (rp_))) << 3)) = (a_);         \
      *reinterpret_cast<bf16x8_t*>((sR_) + (2 * (rp_) + 1) * 128 + (((dg_) ^ fl_swz(2 * (rp_) + 1)) << 3)) = (b_); \
    } else {                                                                          \
      *reinterpret_cast<bf16x8_t*>((sR_) + (2 * (rp_)) * LDR + (dg_) * 8) = (a_);     \
      *reinterpret_cast<bf16x8_t*>((sR_) + (2 * (rp_) + 1) * LDR + (dg_) * 8) = (b_); \
    }                                                                                 \
    if (!TR) {                                                                        \
      _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                              \
        bf16x2_t pr_;                                                                 \
        pr_[0] = (a_)[e_];                                                            \
        pr_[1] = (b_)[e_];                                                            \
        *reinterpret_cast<bf16x2_t*>((sT_) + ((dg_) * 8 + e_) * LDT + 2 * (rp_)) = pr_; \
      }                                                                               \
    }                                                                                 \
  } while (0)
  DKDV_LOAD(hk * group, qt0);
  for (int hq = 0; hq < group; ++hq) {
    const int h = hk * group + hq;
    for (int qt = qt0; qt < nqt; ++qt) {
      __syncthreads();  // the previous tile is fully consumed
      DKDV_PUT(sQ, sQT, rq00, rq01, rp0, dg0);
      DKDV_PUT(sQ, sQT, rq10, rq11, rp1, dg1);
      DKDV_PUT(sDO, sDOT, rg00, rg01, rp0, dg0);
      DKDV_PUT(sDO, sDOT, rg10, rg11, rp1, dg1);
      if (tid < 128) sLse[tid] = tid < 64 ? rl * LOG2E : rl;  // [0,64) lse * log2 e, [64,128) D  (sD == sLse + 64)
      __syncthreads();
      {
        const bool last_q = (qt + 1 == nqt);
        const int nh = last_q ? h + 1 : h, nq = last_q ? qt0 : qt + 1;
        if (!(last_q && hq + 1 == group)) DKDV_LOAD(nh, nq);
      }
      if (CAUSAL && qt * 64 + 63 < k0) continue;  // every query of the tile precedes this wave's keys
      // 8-byte half r (0 / 1: rows + 8) of transposed fragment i = 8 qb16 + 2 d + w of half qs: w = 0 dO^T (dV), w = 1 Q^T (dK)
      auto tr_half = [&](int qs, int i, int r) __attribute__((always_inline)) -> bf16x4_t {
        if (TR) {
          typedef short s16x4_v __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4_v* lds_s16x4_p;
          const char* a_ = reinterpret_cast<const char*>((i & 1) ? sQ : sDO) + tro_[(i >> 1) & 3][r] + (qs * 32 + 16 * (i >> 3)) * FL_ROW_BYTES;
          return __builtin_bit_cast(bf16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)a_));
        }
        const bf16_t* q_ = ((i & 1) ? sQT : sDOT) + (((i >> 1) & 3) * 32 + j) * LDT + qs * 32 + 16 * (i >> 3) + 4 * g;
        return *reinterpret_cast<const bf16x4_t*>(q_ + 8 * r);
      };
      // ---- one 64-query tile = two 32-query halves qs.  Pieces (all force-inlined; the same arithmetic in both orders):
      auto sdp = [&](int qs, f32x16_t& s, f32x16_t& dp) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t qf = TR ? *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const char*>(sQ) + kro_[ks] + qs * 32 * FL_ROW_BYTES)
                                 : *reinterpret_cast<const bf16x8_t*>(sQ + (qs * 32 + j) * LDR + ks * 16 + g * 8);
          const bf16x8_t gf = TR ? *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const char*>(sDO) + kro_[ks] + qs * 32 * FL_ROW_BYTES)
                                 : *reinterpret_cast<const bf16x8_t*>(sDO + (qs * 32 + j) * LDR + ks * 16 + g * 8);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], s, 0, 0, 0);    // S[query][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, vf[ks], dp, 0, 0, 0);  // dP[query][key]
        }
      };
      // pr = P, s = dP - D (in place of S)
      auto probs = [&](int qs, f32x16_t& s, const f32x16_t& dp, f32x16_t& pr) __attribute__((always_inline)) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int qrow = qs * 32 + 8 * r4 + 4 * g;  // rows qrow .. qrow+3 <-> registers 4*r4 .. 4*r4+3
          const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sLse + qrow);
          const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(sD + qrow);
          // four independent elements per step (fma x 4, exp x 4, sub x 4), not element by element: a dependent
          // instruction right behind its producer waits out the pipeline latency, and one wave per SIMD has nothing to fill it
          float t4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) t4[e] = __builtin_fmaf(s[4 * r4 + e], c2, -l4[e]);
#pragma unroll
          for (int e = 0; e < 4; ++e) pr[4 * r4 + e] = __builtin_amdgcn_exp2f(t4[e]);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[4 * r4 + e] = dp[4 * r4 + e] - d4[e];
        }
      };
      // The mask as ONE wave-uniform block over the 16 probabilities: inside the element loop hipcc if-converted the
      // test into a compare + select per element on EVERY tile (64 of the tile's ~680 vector / scalar instructions; one
      // wave per SIMD pays each in full), while only the tiles on the diagonal (or holding padding) need it.
      auto mask16 = [&](int qs, f32x16_t& pr) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qidx = qt * 64 + qs * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
          if (!(CAUSAL ? (ki <= qidx && (kvalid || ki == qidx)) : ki < p.kv_len)) pr[r] = 0.f;
        }
      };
      auto dvdk = [&](int qs, const f32x16_t& pr, f32x16_t& s) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = pr[r] * s[r];  // dS = P (dP - D)
#pragma unroll
        for (int qb16 = 0; qb16 < 2; ++qb16) {
          const int o8 = 8 * qb16;
          const bf16x8_t pf = cvt8(pr[o8], pr[o8 + 1], pr[o8 + 2], pr[o8 + 3], pr[o8 + 4], pr[o8 + 5], pr[o8 + 6], pr[o8 + 7]);
          const bf16x8_t dsf = cvt8(s[o8], s[o8 + 1], s[o8 + 2], s[o8 + 3], s[o8 + 4], s[o8 + 5], s[o8 + 6], s[o8 + 7]);
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const bf16x4_t g_lo = tr_half(qs, 8 * qb16 + 2 * d, 0), g_hi = tr_half(qs, 8 * qb16 + 2 * d, 1);
            const bf16x4_t q_lo = tr_half(qs, 8 * qb16 + 2 * d + 1, 0), q_hi = tr_half(qs, 8 * qb16 + 2 * d + 1, 1);
            bf16x8_t gt, qtf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { gt[e] = g_lo[e]; gt[4 + e] = g_hi[e]; qtf[e] = q_lo[e]; qtf[4 + e] = q_hi[e]; }
            adv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, adv[d], 0, 0, 0);    // dV^T[d][key]
            adk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, adk[d], 0, 0, 0);  // dK^T[d][key]
          }
        }
      };
      // mask only where some key of the wave can be masked for some query of a 32-row half tile
      const bool edge0 = (CAUSAL ? (k0 + 31 > qt * 64) : (k0 + 32 > p.kv_len)) || wave_padded;
      const bool edge1 = (CAUSAL ? (k0 + 31 > qt * 64 + 32) : (k0 + 32 > p.kv_len)) || wave_padded;
      if (PIPE && !edge0 && !edge1) {
        // Round 5 (knob CMB_KNOB_FLASH = 1), interior tiles.  One wave per SIMD has no other wave to fill its stalls, so the
        // tile is issued as four phases of 16 MFMAs in a FIXED order (FLASH_FENCE): every MFMA is followed by the LDS read of
        // the fragment four products ahead (ring of four) and by one element of the OTHER half's softmax arithmetic —
        //   A  S0, dP0                        B  S1, dP1   + P0, dS0 (element i per product)
        //   C  dV0, dK0  + P1, dS1            D  dV1, dK1
        // — where hipcc's own order was read, read, wait, MFMA, wait, MFMA on eight fragment registers with the exponentials
        // of a half strictly between its two groups of products (matrix pipe 30 % busy, the wave waiting 29-44 % of its
        // cycles: profiles/r04_pmc_flash.jsonl).  Same operations on the same operands, same accumulation order: bit-identical.
        f32x16_t s0, dp0, s1, dp1, pr0, pr1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; dp0[r] = 0.f; s1[r] = 0.f; dp1[r] = 0.f; }
        f32x4_t lq[4], dq4[4];
        // product i = 2 ks + w of a score phase: w = 0 S (Q rows x K fragment), w = 1 dP (dO rows x V fragment)
        auto row_frag = [&](int qs, int i) __attribute__((always_inline)) -> bf16x8_t {
          if (TR) return *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const char*>((i & 1) ? sDO : sQ) + kro_[i >> 1] + qs * 32 * FL_ROW_BYTES);
          return *reinterpret_cast<const bf16x8_t*>(((i & 1) ? sDO : sQ) + (qs * 32 + j) * LDR + (i >> 1) * 16 + g * 8);
        };
        auto load_ld = [&](int qs) __attribute__((always_inline)) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            lq[r4] = *reinterpret_cast<const f32x4_t*>(sLse + qs * 32 + 8 * r4 + 4 * g);
            dq4[r4] = *reinterpret_cast<const f32x4_t*>(sD + qs * 32 + 8 * r4 + 4 * g);
          }
        };
        // element r of a half: pr = P, s = dS = P (dP - D)
        auto elem = [&](int r, f32x16_t& s, const f32x16_t& dp, f32x16_t& pr) __attribute__((always_inline)) {
          const float t = __builtin_fmaf(s[r], c2, -lq[r >> 2][r & 3]);
          pr[r] = __builtin_amdgcn_exp2f(t);
          const float u = dp[r] - dq4[r >> 2][r & 3];
          s[r] = pr[r] * u;
        };
        bf16x8_t rf[4];
        // ---- A: S0, dP0
#pragma unroll
        for (int i = 0; i < 4; ++i) rf[i] = row_frag(0, i);
        FLASH_FENCE();
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
          if (i & 1) dp0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], vf[i >> 1], dp0, 0, 0, 0);
          else s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], kf[i >> 1], s0, 0, 0, 0);
          rf[i & 3] = row_frag(i + 4 < 2 * KS ? 0 : 1, (i + 4) & (2 * KS - 1));   // runs on into phase B's first four
          FLASH_FENCE();
        }
        load_ld(0);
        FLASH_FENCE();
        // ---- B: S1, dP1  +  the first half's probabilities
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
          if (i & 1) dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], vf[i >> 1], dp1, 0, 0, 0);
          else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[i & 3], kf[i >> 1], s1, 0, 0, 0);
          if (i + 4 < 2 * KS) rf[i & 3] = row_frag(1, i + 4);
          elem(i, s0, dp0, pr0);
          FLASH_FENCE();
        }
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const int o8 = 8 * h8;
          pf[h8] = cvt8(pr0[o8], pr0[o8 + 1], pr0[o8 + 2], pr0[o8 + 3], pr0[o8 + 4], pr0[o8 + 5], pr0[o8 + 6], pr0[o8 + 7]);
          dsf[h8] = cvt8(s0[o8], s0[o8 + 1], s0[o8 + 2], s0[o8 + 3], s0[o8 + 4], s0[o8 + 5], s0[o8 + 6], s0[o8 + 7]);
        }
        load_ld(1);
        bf16x4_t tlo[4], thi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tlo[i] = tr_half(0, i, 0);
          thi[i] = tr_half(0, i, 1);
        }
        FLASH_FENCE();
        // ---- C: dV0, dK0  +  the second half's probabilities
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          bf16x8_t tf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { tf[e] = tlo[i & 3][e]; tf[4 + e] = thi[i & 3][e]; }
          const int d = (i >> 1) & 3;
          if (i & 1) adk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, dsf[i >> 3], adk[d], 0, 0, 0);
          else adv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[i >> 3], adv[d], 0, 0, 0);
          {
            tlo[i & 3] = tr_half(i + 4 < 16 ? 0 : 1, (i + 4) & 15, 0);   // runs on into phase D's first four
            thi[i & 3] = tr_half(i + 4 < 16 ? 0 : 1, (i + 4) & 15, 1);
          }
          elem(i, s1, dp1, pr1);
          FLASH_FENCE();
        }
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const int o8 = 8 * h8;
          pf[h8] = cvt8(pr1[o8], pr1[o8 + 1], pr1[o8 + 2], pr1[o8 + 3], pr1[o8 + 4], pr1[o8 + 5], pr1[o8 + 6], pr1[o8 + 7]);
          dsf[h8] = cvt8(s1[o8], s1[o8 + 1], s1[o8 + 2], s1[o8 + 3], s1[o8 + 4], s1[o8 + 5], s1[o8 + 6], s1[o8 + 7]);
        }
        FLASH_FENCE();
        // ---- D: dV1, dK1
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          bf16x8_t tf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { tf[e] = tlo[i & 3][e]; tf[4 + e] = thi[i & 3][e]; }
          const int d = (i >> 1) & 3;
          if (i & 1) adk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, dsf[i >> 3], adk[d], 0, 0, 0);
          else adv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[i >> 3], adv[d], 0, 0, 0);
          if (i + 4 < 16) {
            tlo[i & 3] = tr_half(1, i + 4, 0);
            thi[i & 3] = tr_half(1, i + 4, 1);
          }
          FLASH_FENCE();
        }
      } else {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          f32x16_t s, dp, pr;
          sdp(qs, s, dp);
          probs(qs, s, dp, pr);
          if (qs == 0 ? edge0 : edge1) mask16(qs, pr);
          dvdk(qs, pr, s);
        }
      }
    }
  }
#undef DKDV_LOAD
#undef DKDV_PUT
  bf16_t* okr = p.dk + (int64_t)b * p.kv_sb + (int64_t)ki * p.kv_ss + (int64_t)hk * p.kv_sh;
  bf16_t* ovr = p.dv + (int64_t)b * p.kv_sb + (int64_t)ki * p.kv_ss + (int64_t)hk * p.kv_sh;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      bf16x4_t a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = (bf16_t)(adk[d][4 * qd + e] * p.scale);
        c[e] = (bf16_t)adv[d][4 * qd + e];
      }
      *reinterpret_cast<bf16x4_t*>(okr + d * 32 + 8 * qd + 4 * g) = a;
      *reinterpret_cast<bf16x4_t*>(ovr + d * 32 + 8 * qd + 4 * g) = c;
    }
  }  // rep
}

}  // namespace

extern "C" int cmb_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                  const float* lse, int64_t B, int64_t S, int32_t H, int32_t HKV, int32_t hd,
                                  int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t kv_sb, int64_t kv_ss, int64_t kv_sh,
                                  float scale, int32_t causal, int64_t kv_len, const uint8_t* key_valid, float* dvec,
                                  void* dq, void* dk, void* dv, void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dvec || !dq || !dk || !dv) return CMB_ERR_BAD_ARG;
  if (hd != HD || S <= 0 || (S % 128) != 0 || H <= 0 || HKV <= 0 || (H % HKV) != 0 || B < 0) return CMB_ERR_SHAPE;
  if (!causal && (kv_len <= 0 || kv_len > S)) return CMB_ERR_SHAPE;
  if (B == 0) return CMB_OK;
  if ((q_ss % 8) || (q_sh % 8) || (q_sb % 8) || (kv_ss % 8) || (kv_sh % 8) || (kv_sb % 8)) return CMB_ERR_ALIGNMENT;
  FlashParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o;
  p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.lse = lse; p.dvec = dvec;
  p.q_sb = q_sb; p.q_ss = q_ss; p.q_sh = q_sh; p.kv_sb = kv_sb; p.kv_ss = kv_ss; p.kv_sh = kv_sh;
  p.B = (int)B; p.S = (int)S; p.H = H; p.HKV = HKV; p.scale = scale; p.kv_len = causal ? (int)S : (int)kv_len;
  p.key_valid = causal ? key_valid : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nkb = S / 128;
  const int items = flash_items((int)nkb, causal != 0);
  const dim3 gq((unsigned)((int64_t)items * H * B)), gk((unsigned)((int64_t)items * HKV * B));   // 1-D: flash_map.h
  constexpr int smem = (2 * 64 * LDR + 2 * HD * LDT) * 2 + 128 * 4;
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    bool ok = true;
#define DKDV_ATTR(C_, M_, P_)                                                                                   \
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(flash_dkdv_kernel<C_, M_, P_, false>),           \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess &&              \
       hipFuncSetAttribute(reinterpret_cast<const void*>(flash_dkdv_kernel<C_, M_, P_, true>),                  \
                           hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess
    DKDV_ATTR(true, true, false); DKDV_ATTR(true, false, false); DKDV_ATTR(false, false, false);
    DKDV_ATTR(true, true, true); DKDV_ATTR(true, false, true); DKDV_ATTR(false, false, true);
#undef DKDV_ATTR
    if (!ok) return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  // knob bits: 2 = dQ on LDS-DMA tiles (flash2.hip), 4 = the round-4 dK/dV kernel with the round-5 four-phase tile body,
  // 16 = its transposed fragments by transposing reads; 0 = round 4  (bit 8, a dK/dV kernel on LDS-DMA tiles, is gone)
  const int knob = cmb_knob(CMB_KNOB_FLASH);
  const bool pipe_q = (knob & 2) != 0, pipe_k = (knob & 4) != 0, tr_k = (knob & 16) != 0;
#define FLASH_BWD_LAUNCH(C_, M_)                                                                      \
  do {                                                                                                \
    if (pipe_q) {                                                                                     \
      const int rc_ = launch_flash_dq2(p, C_, s);                                                     \
      if (rc_ != CMB_OK) return rc_;                                                                  \
    } else hipLaunchKernelGGL((flash_dq_kernel<C_, M_, false>), gq, dim3(256), 0, s, p);              \
    if (pipe_k && tr_k) hipLaunchKernelGGL((flash_dkdv_kernel<C_, M_, true, true>), gk, dim3(256), smem, s, p); \
    else if (pipe_k) hipLaunchKernelGGL((flash_dkdv_kernel<C_, M_, true, false>), gk, dim3(256), smem, s, p); \
    else if (tr_k) hipLaunchKernelGGL((flash_dkdv_kernel<C_, M_, false, true>), gk, dim3(256), smem, s, p); \
    else hipLaunchKernelGGL((flash_dkdv_kernel<C_, M_, false, false>), gk, dim3(256), smem, s, p);    \
  } while (0)
  if (causal && p.key_valid) FLASH_BWD_LAUNCH(true, true);
  else if (causal) FLASH_BWD_LAUNCH(true, false);
  else FLASH_BWD_LAUNCH(false, false);
#undef FLASH_BWD_LAUNCH
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t B, int64_t S, int32_t H, int32_t HKV,
                                  int32_t hd, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t kv_sb, int64_t kv_ss,
                                  int64_t kv_sh, float scale, int32_t causal, int64_t kv_len, const uint8_t* key_valid,
                                  void* out, float* lse, void* stream) {
  if (!q || !k || !v || !out || !lse) return CMB_ERR_BAD_ARG;
  if (hd != HD || S <= 0 || (S % 128) != 0 || H <= 0 || HKV <= 0 || (H % HKV) != 0 || B < 0) return CMB_ERR_SHAPE;
  if (!causal && (kv_len <= 0 || kv_len > S)) return CMB_ERR_SHAPE;
  if (B == 0) return CMB_OK;
  if ((q_ss % 8) || (q_sh % 8) || (q_sb % 8) || (kv_ss % 8) || (kv_sh % 8) || (kv_sb % 8)) return CMB_ERR_ALIGNMENT;
  FlashParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = nullptr; p.dout = nullptr;
  p.dq = p.dk = p.dv = nullptr; p.lse = nullptr; p.dvec = nullptr;
  p.q_sb = q_sb; p.q_ss = q_ss; p.q_sh = q_sh; p.kv_sb = kv_sb; p.kv_ss = kv_ss; p.kv_sh = kv_sh;
  p.B = (int)B; p.S = (int)S; p.H = H; p.HKV = HKV; p.scale = scale; p.kv_len = causal ? (int)S : (int)kv_len;
  p.key_valid = causal ? key_valid : nullptr;
  const int64_t nqb = S / 128;
  // knob bit 0: the round-5 forward on LDS-DMA tiles (flash2.hip); else the round-4 kernel
  if ((cmb_knob(CMB_KNOB_FLASH) & 1) != 0) return launch_flash_fwd2(p, (bf16_t*)out, lse, causal != 0, (hipStream_t)stream);
  const dim3 grid((unsigned)((int64_t)flash_items((int)nqb, causal != 0) * H * B));   // 1-D: flash_map.h
  if (causal && p.key_valid)
    hipLaunchKernelGGL((flash_fwd_kernel<true, true, false>), grid, dim3(256), 0, (hipStream_t)stream, p, (bf16_t*)out, lse);
  else if (causal)
    hipLaunchKernelGGL((flash_fwd_kernel<true, false, false>), grid, dim3(256), 0, (hipStream_t)stream, p, (bf16_t*)out, lse);
  else
    hipLaunchKernelGGL((flash_fwd_kernel<false, false, false>), grid, dim3(256), 0, (hipStream_t)stream, p, (bf16_t*)out, lse);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
