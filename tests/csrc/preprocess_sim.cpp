// CPU simulation of the image pre-processing kernels: walks the 1-D grids of preprocess.hip block by block (same
// block -> job prefix tables), thread by thread, and runs
// the per-thread code of cambrian_amd/csrc/preprocess_core.h (the very header the kernels include) on host memory.
// Built as a shared object with g++ by tests/test_preprocess.py and driven through ctypes with the job table the
// product's host code (cambrian_amd/train/image_pipeline.py) builds — so the table layout, the letter-box offsets,
// the tap-major coefficient rows and the fixed-point arithmetic are all checked against the oracle without a GPU.
#include <cstdint>
#include "../../cambrian_amd/csrc/preprocess_core.h"

extern "C" int sim_resize_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs) {
  return cmb_resize_coeffs_host(in_size, out_size, bounds, coefs);
}

extern "C" int sim_image_preprocess(const cmb_image_job* jobs, int32_t n_jobs, const uint8_t* src,
                                    const int32_t* bounds, const int32_t* coefs, const float* lut, uint8_t* tmp,
                                    float* dst) {
  const int T = CMB_PP_THREADS;
  for (int j0 = 0; j0 < n_jobs; j0 += CMB_PP_MAX_JOBS) {     // one launch pair per CMB_PP_MAX_JOBS jobs, as the library does
    const int nj = n_jobs - j0 < CMB_PP_MAX_JOBS ? n_jobs - j0 : CMB_PP_MAX_JOBS;
    cmb_block_starts hs, vs;
    int64_t hb, vb;
    if (!cmb_block_tables(jobs + j0, nj, &hs, &vs, &hb, &vb)) return -3;
    const cmb_image_job* jd = jobs + j0;
    for (int64_t blk = 0; blk < hb; ++blk) {                  // pass H: blockIdx.x
      const int j = cmb_job_of_block(hs, (int)blk);
      const cmb_image_job J = jd[j];
      if (J.ksize == 0) return -100;                          // an identity job must own no H block
      for (int t = 0; t < T; ++t) {                           // threadIdx.x
        const int64_t g = (blk - hs.start[j]) * T + t;
        if (g >= (int64_t)J.side * J.out_side) continue;
        const int y = (int)(g / J.out_side), xo = (int)(g - (int64_t)y * J.out_side);
        cmb_resample_h(src, bounds, coefs, tmp, J, y, xo);
      }
    }
    for (int64_t blk = 0; blk < vb; ++blk) {                  // pass V
      const int j = cmb_job_of_block(vs, (int)blk);
      const cmb_image_job J = jd[j];
      const int R = J.out_side, p4 = cmb_tmp_pitch(R) >> 2;
      for (int t = 0; t < T; ++t) {
        const int64_t g = (blk - vs.start[j]) * T + t;
        if (g >= (int64_t)R * p4) continue;
        const int yo = (int)(g / p4), x4 = (int)(g - (int64_t)yo * p4);
        int levels[3][4];
        if (J.ksize == 0) cmb_copy_levels(src, J, yo, x4, levels);
        else cmb_resample_v(tmp, bounds, coefs, J, yo, x4, levels);
        const int nvalid = R - 4 * x4 < 4 ? R - 4 * x4 : 4;
        for (int c = 0; c < 3; ++c)
          for (int i = 0; i < nvalid; ++i)
            dst[J.dst_off + ((int64_t)c * R + yo) * R + 4 * x4 + i] = lut[J.lut_off + c * 256 + levels[c][i]];
      }
    }
  }
  return 0;
}
