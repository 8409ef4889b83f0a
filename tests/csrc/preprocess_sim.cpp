// CPU simulation of the image pre-processing kernels: walks the grids of preprocess.hip thread by thread and runs
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
  const int T = 256;
  for (int b = 0; b < n_jobs; ++b) {          // blockIdx.y
    const cmb_image_job J = jobs[b];
    if (J.ksize != 0) {
      const int64_t n = (int64_t)J.side * J.out_side, blocks = (n + T - 1) / T;
      for (int64_t g = 0; g < blocks * T; ++g) {
        if (g >= n) continue;
        const int y = (int)(g / J.out_side), xo = (int)(g - (int64_t)y * J.out_side);
        cmb_resample_h(src, bounds, coefs, tmp, J, y, xo);
      }
    }
    const int R = J.out_side, p4 = cmb_tmp_pitch(R) >> 2;
    const int64_t n = (int64_t)R * p4, blocks = (n + T - 1) / T;
    for (int64_t g = 0; g < blocks * T; ++g) {
      if (g >= n) continue;
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
  return 0;
}
