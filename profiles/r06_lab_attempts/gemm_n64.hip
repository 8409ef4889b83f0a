// gemm_n64.hip — batched C_z[M, 64] = alpha * A_z[M, K] · B_z[64, K]^T in bf16 for gfx950, K <= 1024: the per-head "contract"
// products of the absorbed SVA projections (W_v,h X̄[q, h, :] in the forward, d q_h = W_k,h dU[q, h, :] in the
// backward; vision_sampler.py:187-189 restated per head) — sixteen GEMMs that READ a [13 824 x 16 x 1024] bf16 tensor (453 MB
// per launch at 24 images, 29 GFLOP) and write 28 MB: an HBM read.  On the 128 x 128 tile kernel (half of every tile's columns
// empty, operands through LDS-DMA stages) it ran 126-131 us = 3.5 TB/s; here
//   * a workgroup belongs to ONE head z and keeps all of B_z (64 x K, <= 128 KiB) in LDS for its whole life (XOR-swizzled 16-byte
//     chunks: conflict-free ds_read_b128 fragments), walking row blocks wg, wg + G_z, ... of 128 rows;
//   * a wave owns 32 rows of the block and streams their A rows (K contiguous elements each) straight into MFMA fragments through
//     a ring of four 64-deep k-steps (64 registers, 16 KiB per wave in flight): no LDS round trip for the streamed operand;
//   * the 32 x 64 result leaves in 8-byte pieces (the output is 6 % of the launch's bytes).
// MFMA operand order (B, A) as everywhere in this library: a lane owns an output row.
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

__global__ void __launch_bounds__(256) gemm_n64_batched_kernel(const GemmParams p, const int wgs_per_head) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // B_z: 64 rows x (K * 2) bytes, chunk c of row n at n * K*2 + ((c ^ (n & 15)) << 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, g = lane >> 5;
  const int z = blockIdx.x / wgs_per_head, wg = blockIdx.x - z * wgs_per_head;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (int64_t)z * p.a_bs;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)z * p.b_bs;
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (int64_t)z * p.c_bs;
  const int64_t lda = p.a_map.s2, ldc = p.c_map.s2;
  const int row_bytes = p.K * 2, chunks = p.K >> 3;   // 16-byte chunks per row
  // ---- stage B_z -----------------------------------------------------------------------------------------------------------
  for (int idx = tid; idx < 64 * chunks; idx += 256) {
    const int n = idx / chunks, c = idx - n * chunks;
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(B + (int64_t)n * p.ldb + c * 8);
    *reinterpret_cast<bf16x8_t*>(smem + n * row_bytes + ((c ^ (n & 15)) << 4)) = v;
  }
  __syncthreads();
  const char* b0 = smem + m * row_bytes;            // column block 0: B rows 0..31
  const char* b1 = smem + (m + 32) * row_bytes;     // column block 1
  const int sw = m & 15;                            // (m + 32) & 15 == m & 15
  const int ksteps = p.K >> 6;                      // 64-deep steps per row block (<= 16)
  const int nblk = (p.M + 127) >> 7;
  // flat schedule: t = blk_i * ksteps + ks over this workgroup's row blocks blk = wg + blk_i * wgs_per_head
  const int my_blocks = wg < nblk ? (nblk - 1 - wg) / wgs_per_head + 1 : 0;
  const int total = my_blocks * ksteps;
  if (total == 0) return;
  auto a_ptr = [&](int t) -> const bf16_t* {
    const int bi = t / ksteps, ks = t - bi * ksteps;
    const int64_t r0 = ((int64_t)(wg + bi * wgs_per_head) << 7) + wave * 32 + m;
    const int64_t r = r0 < p.M ? r0 : p.M - 1;
    return A + r * lda + ks * 64 + g * 8;
  };
  bf16x8_t ring[4][4];   // [slot][16-deep sub-step]
  auto issue = [&](auto slot_c, int t) {
    constexpr int slot = decltype(slot_c)::value;
    if (t < total) {
      const bf16_t* q = a_ptr(t);
#pragma unroll
      for (int s = 0; s < 4; ++s) ring[slot][s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(q + s * 16));
    }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  typedef std::integral_constant<int, 2> S2;
  typedef std::integral_constant<int, 3> S3;
  issue(S0{}, 0); issue(S1{}, 1); issue(S2{}, 2); issue(S3{}, 3);
  f32x16_t acc0 = {0}, acc1 = {0};
  auto step = [&](auto slot_c, int t) {
    constexpr int slot = decltype(slot_c)::value;
    if (t >= total) return;
    const int bi = t / ksteps, ks = t - bi * ksteps;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = (ks * 8 + s * 2 + g) ^ sw;
      const bf16x8_t f0 = *reinterpret_cast<const bf16x8_t*>(b0 + (c << 4));
      const bf16x8_t f1 = *reinterpret_cast<const bf16x8_t*>(b1 + (c << 4));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, ring[slot][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, ring[slot][s], acc1, 0, 0, 0);
    }
    issue(slot_c, t + 4);
    if (ks == ksteps - 1) {   // the block's last k-step: write the wave's 32 x 64 results
      const int64_t r = ((int64_t)(wg + bi * wgs_per_head) << 7) + wave * 32 + m;
      if (r < p.M) {
        bf16_t* crow = C + r * ldc + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4_t o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = (bf16_t)(acc0[4 * q + e] * p.alpha);
            o1[e] = (bf16_t)(acc1[4 * q + e] * p.alpha);
          }
          *reinterpret_cast<bf16x4_t*>(crow + 8 * q) = o0;
          *reinterpret_cast<bf16x4_t*>(crow + 32 + 8 * q) = o1;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc0[e] = 0.f, acc1[e] = 0.f;
    }
  };
  for (int t = 0; t < total; t += 4) {
    step(S0{}, t); step(S1{}, t + 1); step(S2{}, t + 2); step(S3{}, t + 3);
  }
}

}  // namespace

// Eligibility (gemm.hip): bf16 in and out, N == 64, K % 64 == 0, K <= 1024 (B_z in LDS), batch > 1, plain row maps and epilogue.
bool gemm_n64_eligible(const GemmParams& p) {
  static int on = -1;   // CMB_GEMM_N64=0: the 128 x 128 tile kernel as before (A/B runs)
  if (on < 0) {
    const char* e = getenv("CMB_GEMM_N64");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on && p.batch > 1 && p.N == 64 && (p.K % 64) == 0 && p.K >= 256 && p.K <= 1024 && !p.out_f32 && !p.bias && !p.colscale && !p.R &&
         !p.P && !p.slabs && p.act == CMB_ACT_NONE && p.a_map.n1 == 0 && p.c_map.n1 == 0 && (p.c_map.s2 % 4) == 0 && (p.c_bs % 4) == 0 &&
         !p.row_mean && !p.a_scale && !p.b_scale;
}

int launch_gemm_n64_batched(const GemmParams& p, hipStream_t s) {
  static int n_cu = 0;
  static bool attr_done = false;
  if (!attr_done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu <= 0)
      return CMB_ERR_LAUNCH;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_n64_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            64 * 1024 * 2) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_done = true;
  }
  int per = n_cu / p.batch;
  if (per < 1) per = 1;
  const int nblk = (p.M + 127) / 128;
  if (per > nblk) per = nblk;
  hipLaunchKernelGGL(gemm_n64_batched_kernel, dim3((unsigned)(p.batch * per)), dim3(256), (size_t)64 * p.K * 2, s, p, per);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace cmb_gemm_detail
