// gemm_smallm.hip — C[M, N] = alpha * A[M, K] · B[N, K]^T for M <= 32 rows in bf16 (gfx950): the per-image context vectors of
// the SVA layers (vision_sampler.py:279-292 proj_context: one row per image, 24 x 1024 x 1024 at 24 images, 52 launches per
// step forward + backward).  On the 128 x 128 tile kernel such a problem is 8 workgroups walking the whole K one after the
// other — 22 us per launch for 50 MFLOP.  Here a workgroup owns 32 output columns and its four waves split K: 32 workgroups x 4
// waves, 16 k-steps each for K = 1024, the partial tiles summed through LDS.  Plain epilogue only (alpha, bf16 or fp32 C).
// MFMA operand order (B, A) as everywhere in this library: a lane owns an output row.
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

__global__ void __launch_bounds__(256) gemm_small_m_kernel(const GemmParams p) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, g = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int mrow = m < p.M ? m : p.M - 1;
  const bf16_t* arow = reinterpret_cast<const bf16_t*>(p.A) + row_off(p.a_map, (uint32_t)mrow) + g * 8;
  const int nrow = n0 + m < p.N ? n0 + m : p.N - 1;
  const bf16_t* brow = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)nrow * p.ldb + g * 8;
  const int ksteps = p.K >> 4, per = (ksteps + 3) >> 2;
  const int k_lo = wave * per, k_hi = (k_lo + per < ksteps) ? k_lo + per : ksteps;
  f32x16_t acc = {0};
  int ks = k_lo;
  for (; ks + 4 <= k_hi; ks += 4) {   // four k-steps' operands in flight
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const bf16x8_t*>(arow + (ks + u) * 16);
      b[u] = *reinterpret_cast<const bf16x8_t*>(brow + (ks + u) * 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u], a[u], acc, 0, 0, 0);
  }
  for (; ks < k_hi; ++ks) {
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + ks * 16), b = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave != 0 || m >= p.M) return;
  // lane (row m, half g): register r is column n0 + (r & 3) + 8 (r >> 2) + 4 g
  const int64_t coff = row_off(p.c_map, (uint32_t)m);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      v[e] = (red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane]) * p.alpha;
    }
    const int col = n0 + 8 * q + 4 * g;
    if (col + 4 <= p.N) {
      if (p.out_f32) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.C) + coff + col) = (f32x4_t){v[0], v[1], v[2], v[3]};
      } else {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<bf16_t*>(p.C) + coff + col) = o;
      }
    }
  }
}

}  // namespace

// Eligibility (gemm.hip): bf16 operands, M <= 32, one problem, whole K, N % 32 == 0, plain epilogue, no beta.
bool gemm_small_m_eligible(const GemmParams& p, int splits) {
  return p.M <= 32 && p.batch == 1 && splits == 1 && !p.slabs && (p.N % 32) == 0 && (p.K % 16) == 0 && !p.bias && !p.colscale && !p.R &&
         !p.P && p.act == CMB_ACT_NONE && !p.a_scale && !p.b_scale && !p.row_mean && p.beta == 0.0f && p.c_map.n1 == 0 &&
         (p.c_map.s2 % 4) == 0;
}

int launch_gemm_small_m(const GemmParams& p, hipStream_t s) {
  hipLaunchKernelGGL(gemm_small_m_kernel, dim3((unsigned)(p.N / 32)), dim3(256), 0, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace cmb_gemm_detail
