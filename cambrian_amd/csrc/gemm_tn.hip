// gemm_tn.hip — C[M,N] = alpha * At[K,M]^T · Bt[K,N] (+ beta C) for gfx950, bf16 operands, fp32 accumulation.
//
// The weight gradient of every trainable linear is this product with K = the rows of the batch: dW[N_out, K_in] =
// g[rows, N_out]^T · x[rows, K_in] (the reference leaves it to autograd's addmm on a transposed view, e.g. the SVA
// projections vision_sampler.py:159-189 and the projectors cambrian_arch.py:49-56).  The NT kernels of gemm.hip need both
// operands K-contiguous, so round 2 materialised g^T and x^T (two activation-sized transposes per trainable linear and
// step, 7.7 ms per step at 16 images).  Here both operands are read as they lie:
//   * a K-step is 64 ROWS of At and of Bt; a tile's slice of a row is 256 contiguous bytes (128 columns), fetched by
//     LDS-DMA as they are: one 1 KiB instruction covers 4 rows (16 lanes x 16 B per row);
//   * the MFMA wants, per lane, 8 consecutive k of ONE column: ds_read_b64_tr_b16 hands a lane 4 rows (k) of one column
//     out of a row-major [4][16] block — two reads per operand fragment.  Which k a register holds is free as long as At
//     and Bt agree, so the two halves of the wave take different 4-row blocks and a fragment's reads stay whole blocks;
//   * 16-byte slot x of row k is stored at slot x ^ 4 (k & 3) of its 256-byte row (applied to the lane's DMA source), so
//     the four rows of a [4][16] block, which lie 256 bytes apart, fall into different banks: the 512 bytes of a
//     transposing read take their minimum two passes;
//   * rows beyond K (the batch is not a multiple of 64) are fetched from a zero row instead of being padded by the caller.
// Epilogue (fp32 slabs for split-K, alpha / beta, fp32 or bf16 result), tile rasterisation and the accumulator layout are
// gemm.hip's (gemm_common.h).  128 x 128 tile, 4 waves (2 x 2), each 64 x 64 as 2 x 2 v_mfma_f32_32x32x16_bf16 tiles.
#include "gemm_common.h"
#include "gemm_tn_layout.h"

using namespace cmb_gemm_detail;

namespace {

__device__ __attribute__((aligned(256))) char g_zero_row[256];   // zero-initialised: the source of rows >= K

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;

// Round 6: a ring of FOUR 32-row stages instead of two 64-row ones (the same 64 KiB).  With two stages every K-step ended in
// s_waitcnt vmcnt(0) + barrier, so a stage's DMA had ONE step of MFMAs (16 per wave, ~0.25 us) to land in: the split-K weight
// gradients of the SVA layers ran 1.4-1.5 us per 64 rows — the round trip to HBM, not the arithmetic (2048 x 1024 x 13 824:
// 79 us for 58 GFLOP).  Now stage t + 3 is requested while stage t is consumed, the wait is counted (the two younger stages
// stay in flight) and a step has one raw barrier.
constexpr int BM = 128, BN = 128, BKR = 32;                // BKR: rows (k) per stage
constexpr int NST = 4;                                     // stages in the ring
constexpr int OP_BYTES = BKR * 256, STAGE = 2 * OP_BYTES;  // 8 KiB per operand, 16 KiB per stage
constexpr int CS = BN + 4;
constexpr int SMEM = (NST * STAGE) > (BM * CS * 4) ? (NST * STAGE) : (BM * CS * 4);

__global__ void __launch_bounds__(256) gemm_tn_kernel(const GemmParams p_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmParams p = p_in;
  if (p.batch > 1) {  // batched launch: operand / result bases advance by the batch strides (elements)
    const int64_t bz = blockIdx.z;
    p.A += bz * p.a_bs * 2;
    p.B += bz * p.b_bs * 2;
    p.C += bz * p.c_bs * (int64_t)(p.out_f32 ? 4 : 2);
    if (p.slabs) p.slabs += bz * (int64_t)p.M * p.N;   // slab rows [bz * M, bz * M + M) of slab_rows = batch * M
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = p.tiles_m * p.tiles_n;
  const int id = gl_xcd_remap((int)blockIdx.x, nblk);
  int tile_m, tile_n;
  gl_group_tile(id, p.tiles_m, p.tiles_n, 8, &tile_m, &tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz = blockIdx.y;
  const int kbeg = kz * p.k_per_split;
  const int kend = (kbeg + p.k_per_split < p.K) ? (kbeg + p.k_per_split) : p.K;
  const int nk = (kend > kbeg) ? (kend - kbeg + BKR - 1) / BKR : 0;
  const int64_t lda = p.a_map.s2, ldb = p.ldb;

  // ---- LDS-DMA sources.  An operand stage is 8 pieces of 1 KiB (4 rows each); wave w issues pieces w and w + 4 of At and of
  // Bt.  Lane l of a piece: row l >> 4, LDS slot l & 15 <- source slot (l & 15) ^ 4 (l >> 4).
  // Columns beyond M / N (whole 8-column slots: M, N are multiples of 8) re-read the tile's first slot; never stored.
  const int sx = lane & 15;
  const int sc = tn_dma_src_slot(lane);
  const int acol = (m0 + sc * 8 < p.M) ? (m0 + sc * 8) : m0;
  const int bcol = (n0 + sc * 8 < p.N) ? (n0 + sc * 8) : n0;
  const char* zsrc = g_zero_row + sx * 16;
  const char* a_src[2];
  const char* b_src[2];
  int krow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    krow[i] = kbeg + tn_dma_row(wave + 4 * i, lane);
    a_src[i] = p.A + ((int64_t)krow[i] * lda + acol) * 2;
    b_src[i] = p.B + ((int64_t)krow[i] * ldb + bcol) * 2;
  }
  // The DMA is inline asm (M0 + the load in one statement): through the builtin hipcc knows the instruction writes LDS and
  // parks s_waitcnt vmcnt(0) in front of the first fragment read of every step — the ring would drain each step.
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem));
  auto dma = [&](const char* g, uint32_t lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_byte) : "memory", "m0");
  };
  auto stage = [&](int s) {   // 4 LDS-DMA instructions per wave
    const uint32_t sa = lds0 + (uint32_t)(s * STAGE);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool live = krow[i] < kend;
      dma(live ? a_src[i] : zsrc, sa + (uint32_t)tn_dma_lds_off(wave + 4 * i, 0));
      dma(live ? b_src[i] : zsrc, sa + (uint32_t)(OP_BYTES + tn_dma_lds_off(wave + 4 * i, 0)));
      a_src[i] += (int64_t)BKR * lda * 2;
      b_src[i] += (int64_t)BKR * ldb * 2;
      krow[i] += BKR;
    }
  };

  // ---- fragment reads.  Lane (q = lane >> 4, i = lane & 15) supplies row i >> 2 and the 8-byte piece i & 3 of a
  // [4 k][16 columns] block and receives column i of it (4 consecutive k).  The 32 columns of an MFMA fragment are the
  // blocks of 16-column subtiles 2 f + (q & 1); lanes 0-31 read 4-row piece 4 s + 2 r, lanes 32-63 piece 4 s + 2 r + 1
  // (k-step s, read r): byte offset inside the operand stage, without the (4 s + 2 r) KiB.
  int a_off[2], b_off[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    a_off[f] = tn_frag_off((wm * 64 + f * 32) / 16, lane);
    b_off[f] = OP_BYTES + tn_frag_off((wn * 64 + f * 32) / 16, lane);
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nk > 0) {
    // prologue: stages 0, 1, 2 requested; step t: wait for stage t (the younger ones stay in flight), barrier (every wave's
    // pieces of stage t have landed AND every wave is done reading stage t - 1), request stage t + 3 into the buffer stage
    // t - 1 used, consume stage t.  Raw s_barrier: __syncthreads() would drain the LDS-DMA queue (vmcnt(0)) every step.
    stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = nk - 1 - kt;   // stages requested behind stage kt (capped at 2)
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (kt + 3 < nk) stage((kt + 3) & (NST - 1));
      const char* base = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t a[2], b[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(base + a_off[f] + tn_frag_piece(s, 0) * 1024));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(base + a_off[f] + tn_frag_piece(s, 1) * 1024));
          const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(base + b_off[f] + tn_frag_piece(s, 0) * 1024));
          const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(base + b_off[f] + tn_frag_piece(s, 1) * 1024));
          const s16x8_t av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          const s16x8_t bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          a[f] = __builtin_bit_cast(bf16x8_t, av);
          b[f] = __builtin_bit_cast(bf16x8_t, bv);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);  // swapped: rows = n, cols = m
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");   // every wave is done with the operand ring: the epilogue reuses it
  }

  // ---- epilogue: accumulators -> LDS (fp32, row stride BN + 4) -> row-contiguous global stores (as gemm.hip)
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int m = wm * 64 + i * 32 + gl_acc_m(lane);
        const int n = wn * 64 + j * 32 + gl_acc_n(4 * r4, lane);
        f32x4_t v;
        v[0] = acc[i][j][4 * r4 + 0];
        v[1] = acc[i][j][4 * r4 + 1];
        v[2] = acc[i][j][4 * r4 + 2];
        v[3] = acc[i][j][4 * r4 + 3];
        *reinterpret_cast<f32x4_t*>(cs + m * CS + n) = v;
      }
  __syncthreads();
  constexpr int GPR = BN / 8;
  for (int grp = tid; grp < BM * GPR; grp += 256) {
    const int row = grp / GPR, c8 = grp - row * GPR;
    const int gm = m0 + row, gn = n0 + c8 * 8;
    if (gm >= p.M || gn >= p.N) continue;
    float v[8];
    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8);
    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    gemm_epilogue8<bf16_t, CMB_ACT_NONE>(p, kz, gm, gn, v);
  }
}

}  // namespace

namespace cmb_gemm_detail {

int launch_gemm_tn_bf16(GemmParams& p, int splits, hipStream_t s) {
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) !=
        hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits, (unsigned)(p.batch > 1 ? p.batch : 1));
  hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), SMEM, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace cmb_gemm_detail
