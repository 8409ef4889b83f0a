// fp8.hip — row-wise e4m3fn quantisation feeding the fp8 projection GEMMs (cmb_gemm with CMB_FP8_E4M3;
// BASELINE configs[4]: "fp8 MFMA projection GEMMs").  HBM-bound: reads each row once (2 or 4 bytes / element), keeps
// it in registers across the amax reduction, writes 1 byte / element + one fp32 per row.
//
// One 256-thread block per row; a thread owns up to PP 16-element pieces (K <= 256 * 16 * PP).  gfx950 converts
// with v_cvt_pk_fp8_f32 in OCP mode (e4m3fn, round to nearest even); values are clamped to +-448 first so the cast
// can never overflow to NaN.
#include "common.h"

namespace {

constexpr int Q_THREADS = 256;
constexpr int Q_PIECES = 4;      // K <= 256 * 16 * 4 = 16384

template <typename T>
__device__ __forceinline__ void load16(const T* p, float (&v)[16]) {
  float a[8], b[8];
  Vec8<T>::load(p, a);
  Vec8<T>::load(p + 8, b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = a[i]; v[8 + i] = b[i]; }
}

template <typename T>
__global__ __launch_bounds__(Q_THREADS) void quantize_fp8_rows_kernel(const T* __restrict__ x, int64_t ldx, int K,
                                                                      uint8_t* __restrict__ q, int64_t ldq,
                                                                      float* __restrict__ inv_scale) {
  __shared__ float red[Q_THREADS / 64];
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* xr = x + (int64_t)row * ldx;
  float v[Q_PIECES][16];
  float amax = 0.0f;
#pragma unroll
  for (int pc = 0; pc < Q_PIECES; ++pc) {
    const int k = (pc * Q_THREADS + tid) * 16;
    if (k < K) {
      load16<T>(xr + k, v[pc]);
#pragma unroll
      for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[pc][i]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = amax > 0.0f ? 448.0f / amax : 1.0f;
  if (tid == 0) inv_scale[row] = amax > 0.0f ? amax / 448.0f : 1.0f;
  uint8_t* qr = q + (int64_t)row * ldq;
#pragma unroll
  for (int pc = 0; pc < Q_PIECES; ++pc) {
    const int k = (pc * Q_THREADS + tid) * 16;
    if (k < K) {
      int w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = fminf(fmaxf(v[pc][4 * j + e] * scale, -448.0f), 448.0f);
        int packed = 0;
        packed = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], packed, false);
        packed = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], packed, true);
        w[j] = packed;
      }
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<i32x4*>(qr + k) = i32x4{w[0], w[1], w[2], w[3]};
    }
  }
}

}  // namespace

extern "C" int cmb_quantize_fp8_rows(int dtype, const void* x, int64_t ldx, int64_t rows, int64_t K, void* q,
                                     int64_t ldq, float* inv_scale, void* stream) {
  if (rows == 0) return CMB_OK;
  if (!x || !q || !inv_scale || rows < 0 || K <= 0) return CMB_ERR_BAD_ARG;
  if (dtype != CMB_BF16 && dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  if (K % 16 != 0 || K > Q_THREADS * 16 * Q_PIECES || ldq % 16 != 0) return CMB_ERR_SHAPE;
  const int64_t es = dtype == CMB_BF16 ? 2 : 4;
  if (!cmb_aligned16(x) || !cmb_aligned16(q) || (ldx * es) % 16 != 0) return CMB_ERR_ALIGNMENT;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16)
    hipLaunchKernelGGL(quantize_fp8_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(Q_THREADS), 0, s, (const bf16_t*)x,
                       ldx, (int)K, (uint8_t*)q, ldq, inv_scale);
  else
    hipLaunchKernelGGL(quantize_fp8_rows_kernel<float>, dim3((unsigned)rows), dim3(Q_THREADS), 0, s, (const float*)x, ldx,
                       (int)K, (uint8_t*)q, ldq, inv_scale);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
