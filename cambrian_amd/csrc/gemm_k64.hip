// gemm_k64.hip — batched C_z[M, N] = A_z[M, 64] · B_z[N, 64]^T in bf16 for gfx950: the per-head "expand" products of the
// absorbed SVA projections (U[q, h, :] = W_k,h^T q_h in the forward, d Xb[q, h, :] = W_v,h^T d o_h in the
// backward; vision_sampler.py:187-189 restated per head) — sixteen GEMMs with a contraction of ONE head (64) that write a
// [13 824 x 16 x 1024] bf16 tensor: 453 MB per launch at 24 images against 29 GFLOP.  The launch is an HBM WRITE: on the
// 128 x 128 tile kernel (gemm.hip, one 64-deep K step per tile, 8-byte pieces of 32 rows per store instruction) it ran
// 169 us = 2.7 TB/s (profiles/r06_gemm_shapes_b24.json); here
//   * a workgroup owns 128 rows of one problem z and walks ALL N columns: its A fragments (32 rows x 64 per wave) are loaded
//     once and stay in 16 registers, B_z (N x 64: 128 KiB, shared by every row tile of the head) comes out of L2 as fragments;
//   * 128 output columns at a time are staged through 8 KiB of LDS per wave (XOR-swizzled 16-byte slots: conflict-free writes)
//     and leave as whole 256-byte row segments — 4 rows x 256 B per store instruction, non-temporal.
// MFMA operand order (B, A) as everywhere in this library: a lane owns an output row.
#include <cstdlib>
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

__global__ void __launch_bounds__(256) gemm_k64_batched_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char stage[4 * 8192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, g = lane >> 5;
  const int z = blockIdx.y;
  const int64_t m0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (int64_t)z * p.a_bs;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)z * p.b_bs;
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (int64_t)z * p.c_bs;
  const int64_t lda = p.a_map.s2, ldc = p.c_map.s2;
  const int64_t arow = (m0 + m < p.M) ? (m0 + m) : (p.M - 1);
  bf16x8_t af[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const bf16x8_t*>(A + arow * lda + ks * 16 + g * 8);
  char* st = stage + wave * 8192;
  const int ngrp = p.N >> 7;
  for (int grp = 0; grp < ngrp; ++grp) {
    f32x16_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16_t* brow = B + (int64_t)(grp * 128 + c * 32 + m) * p.ldb + g * 8;
      bf16x8_t bf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], af[0], (f32x16_t){0}, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks], af[ks], acc[c], 0, 0, 0);
    }
    // lane (row m, half g) holds columns c * 32 + 8 k + 4 g + {0..3}: 8-byte piece 2 (4 c + k) + g of the row's 256 bytes
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[c][4 * k + e] * p.alpha);
        const int slot = 4 * c + k;
        *reinterpret_cast<bf16x4_t*>(st + m * 256 + ((slot ^ (m & 15)) << 4) + g * 8) = o;
      }
    // rows 4 i + (lane >> 4), 16-byte slot lane & 15: one 256-byte row segment per 16 lanes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + (lane >> 4), slot = lane & 15;
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(st + r * 256 + ((slot ^ (r & 15)) << 4));
      if (m0 + r < p.M) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8_t*>(C + (m0 + r) * ldc + grp * 128 + slot * 8));
    }
  }
}

}  // namespace

// Eligibility (gemm.hip): bf16 in and out, K == 64, N % 128 == 0, batch > 1, plain row maps, plain epilogue.
bool gemm_k64_eligible(const GemmParams& p) {
  static int on = -1;   // CMB_GEMM_K64=0: the 128 x 128 tile kernel as before (A/B runs)
  if (on < 0) {
    const char* e = getenv("CMB_GEMM_K64");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on && p.batch > 1 && p.K == 64 && (p.N % 128) == 0 && !p.out_f32 && !p.bias && !p.colscale && !p.R && !p.P && !p.slabs &&
         p.act == CMB_ACT_NONE && p.a_map.n1 == 0 && p.c_map.n1 == 0 && (p.c_map.s2 % 8) == 0 && (p.c_bs % 8) == 0 && !p.row_mean;
}

int launch_gemm_k64_batched(const GemmParams& p, hipStream_t s) {
  const dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)p.batch);
  hipLaunchKernelGGL(gemm_k64_batched_kernel, grid, dim3(256), 0, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace cmb_gemm_detail
