// preprocess.hip — batch image pre-processing for the four towers on the GPU:
// letter-box + Pillow-exact two-pass bicubic resample + the processor's pointwise table, uint8 [h,w,3] in,
// planar [3,R,R] bf16 / fp16 / fp32 out.  Replaces the per-sample CPU work of train_fsdp.py:985-1008 and
// mm_utils.py:183-201; only the raw uint8 pixels cross PCIe (≈16x fewer bytes than four fp32 tensors).
//
// HBM-bound byte/integer work (no MFMA): algorithmic bytes per job = 3*w*h (source) + 3*R*R*sizeof(out);
// the uint8 intermediate adds 2 * 3*S*R.  Per-thread arithmetic lives in preprocess_core.h (shared with the CPU
// simulation in tests/csrc/preprocess_sim.cpp).
//
// Jobs differ in size by two orders of magnitude (a 224x224 icon into a 336 tower vs a 1920x1080 photo into the 1024
// tower), so a (max blocks, jobs) grid would be mostly empty blocks: each pass is ONE 1-D grid whose blocks are dealt
// to jobs through a prefix table passed by value in the kernel arguments (scalar binary search, no device memory).
// pass H  thread = one (y, xo); lanes of a wave walk neighbouring xo, so the
//         source bytes of a tap are a short contiguous run (upscale) or a strided run that stays in L1 across
//         taps (downscale); coefficients are tap-major [ksize, R] -> coalesced.  Plane writes are byte-coalesced.
// pass V  thread = 4 neighbouring columns of one output row: one 32-bit
//         load per plane and tap (coalesced across the wave), 12 accumulators, table lookup, 4-wide store.
#include "common.h"
#include "preprocess_core.h"

namespace {

constexpr int PP_THREADS = CMB_PP_THREADS;
constexpr int PP_MAX_JOBS = CMB_PP_MAX_JOBS;
typedef cmb_block_starts BlockStarts;
#define job_of_block cmb_job_of_block

__global__ __launch_bounds__(PP_THREADS) void image_resample_h_kernel(const cmb_image_job* __restrict__ jobs,
                                                                      const uint8_t* __restrict__ src,
                                                                      const int32_t* __restrict__ bounds,
                                                                      const int32_t* __restrict__ coefs,
                                                                      uint8_t* __restrict__ tmp,
                                                                      const BlockStarts starts) {
  const int j = job_of_block(starts, (int)blockIdx.x);
  const cmb_image_job J = jobs[j];
  const int64_t g = (int64_t)((int)blockIdx.x - starts.start[j]) * PP_THREADS + threadIdx.x;
  if (g >= (int64_t)J.side * J.out_side) return;
  const int y = (int)(g / J.out_side), xo = (int)(g - (int64_t)y * J.out_side);
  cmb_resample_h(src, bounds, coefs, tmp, J, y, xo);
}

template <typename T> struct OutCvt;
template <> struct OutCvt<float> { static __device__ __forceinline__ float cvt(float v) { return v; } };
template <> struct OutCvt<bf16_t> { static __device__ __forceinline__ bf16_t cvt(float v) { return (bf16_t)v; } };
template <> struct OutCvt<_Float16> { static __device__ __forceinline__ _Float16 cvt(float v) { return (_Float16)v; } };

template <typename T>
__global__ __launch_bounds__(PP_THREADS) void image_resample_v_kernel(const cmb_image_job* __restrict__ jobs,
                                                                      const uint8_t* __restrict__ src,
                                                                      const int32_t* __restrict__ bounds,
                                                                      const int32_t* __restrict__ coefs,
                                                                      const float* __restrict__ lut,
                                                                      const uint8_t* __restrict__ tmp,
                                                                      T* __restrict__ dst,
                                                                      const BlockStarts starts) {
  const int j = job_of_block(starts, (int)blockIdx.x);
  const cmb_image_job J = jobs[j];
  const int R = J.out_side, p4 = cmb_tmp_pitch(R) >> 2;
  const int64_t g = (int64_t)((int)blockIdx.x - starts.start[j]) * PP_THREADS + threadIdx.x;
  if (g >= (int64_t)R * p4) return;
  const int yo = (int)(g / p4), x4 = (int)(g - (int64_t)yo * p4);
  int levels[3][4];
  if (J.ksize == 0) cmb_copy_levels(src, J, yo, x4, levels);
  else cmb_resample_v(tmp, bounds, coefs, J, yo, x4, levels);
  const float* t = lut + J.lut_off;
  const int n = min(4, R - 4 * x4);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T* o = dst + J.dst_off + ((int64_t)c * R + yo) * R + 4 * x4;
    T v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = OutCvt<T>::cvt(t[c * 256 + levels[c][i]]);
    if (n == 4 && ((uintptr_t)o & (4 * sizeof(T) - 1)) == 0) {
      typedef T vec4 __attribute__((ext_vector_type(4)));
      vec4 pack = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<vec4*>(o) = pack;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < n) o[i] = v[i];
    }
  }
}

}  // namespace

extern "C" int cmb_resize_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs) {
  return cmb_resize_coeffs_host(in_size, out_size, bounds, coefs);
}

extern "C" int cmb_image_preprocess(const cmb_image_job* jobs_dev, const cmb_image_job* jobs_host, int32_t n_jobs,
                                    const uint8_t* src, const int32_t* bounds, const int32_t* coefs,
                                    const float* lut, int32_t out_dtype, uint8_t* tmp, void* dst, void* stream) {
  if (n_jobs == 0) return CMB_OK;
  if (!jobs_dev || !jobs_host || n_jobs < 0 || !src || !lut || !dst) return CMB_ERR_BAD_ARG;
  if (out_dtype != CMB_BF16 && out_dtype != CMB_F32 && out_dtype != CMB_F16) return CMB_ERR_BAD_ARG;
  for (int i = 0; i < n_jobs; ++i) {
    const cmb_image_job& J = jobs_host[i];
    if (J.w <= 0 || J.h <= 0 || J.out_side <= 0 || J.side != (J.w > J.h ? J.w : J.h)) return CMB_ERR_SHAPE;
    if (J.off_x < 0 || J.off_y < 0 || J.off_x + J.w > J.side || J.off_y + J.h > J.side) return CMB_ERR_SHAPE;
    if (J.ksize == 0) {
      if (J.side != J.out_side) return CMB_ERR_SHAPE;
    } else {
      if (!bounds || !coefs || !tmp) return CMB_ERR_BAD_ARG;
      if (J.tmp_off & 3) return CMB_ERR_ALIGNMENT;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  for (int j0 = 0; j0 < n_jobs; j0 += PP_MAX_JOBS) {
    const int nj = n_jobs - j0 < PP_MAX_JOBS ? n_jobs - j0 : PP_MAX_JOBS;
    BlockStarts hs, vs;
    int64_t hb, vb;
    if (!cmb_block_tables(jobs_host + j0, nj, &hs, &vs, &hb, &vb)) return CMB_ERR_SHAPE;
    const cmb_image_job* jd = jobs_dev + j0;
    if (hb > 0) {
      hipLaunchKernelGGL(image_resample_h_kernel, dim3((unsigned)hb), dim3(PP_THREADS), 0, s, jd, src, bounds, coefs, tmp,
                         hs);
      CMB_CHECK_LAUNCH();
    }
    const dim3 grid((unsigned)vb);
    if (out_dtype == CMB_F32)
      hipLaunchKernelGGL(image_resample_v_kernel<float>, grid, dim3(PP_THREADS), 0, s, jd, src, bounds, coefs, lut, tmp,
                         (float*)dst, vs);
    else if (out_dtype == CMB_BF16)
      hipLaunchKernelGGL(image_resample_v_kernel<bf16_t>, grid, dim3(PP_THREADS), 0, s, jd, src, bounds, coefs, lut, tmp,
                         (bf16_t*)dst, vs);
    else
      hipLaunchKernelGGL(image_resample_v_kernel<_Float16>, grid, dim3(PP_THREADS), 0, s, jd, src, bounds, coefs, lut,
                         tmp, (_Float16*)dst, vs);
    CMB_CHECK_LAUNCH();
  }
  return CMB_OK;
}
