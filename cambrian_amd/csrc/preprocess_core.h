// preprocess_core.h — per-thread arithmetic of the image pre-processing kernels (preprocess.hip), written as
// host+device functions so tests/csrc/preprocess_sim.cpp can run the very same code on the CPU, thread by thread,
// against the oracle (tests/test_preprocess.py) before anything is launched on a GPU.
//
// Pillow's resample (libImaging/Resample.c) of a letter-boxed square S x S -> R x R:
//   pass H : tmp[c][y][xo] = clip8((2^21 + sum_j px(y, first(xo)+j, c) * k[j][xo]) >> 22)          y < S, xo < R
//   pass V : out[c][yo][x] = lut[c][ clip8((2^21 + sum_j tmp[c][first(yo)+j][x] * k[j][yo]) >> 22) ]
// px() is the virtual letter-boxed image: the source where it was pasted, the tower's mean colour elsewhere
// (mm_utils.py:153-165) — the padded square is never materialised.  The same (first, count, k) rows serve both
// passes because the box is square.  int32 accumulation as in Pillow (|sum| < 255 * 1.4 * 2^22 < 2^31).
#pragma once
#include <stdint.h>
#include "../../include/cambrian_amd.h"

#if defined(__HIPCC__)
#define CMB_PHD __host__ __device__ __forceinline__
#else
#define CMB_PHD static inline
#endif

#define CMB_RESAMPLE_BITS 22   // PRECISION_BITS = 32 - 8 - 2

CMB_PHD int cmb_tmp_pitch(int out_side) { return (out_side + 3) & ~3; }

CMB_PHD int cmb_clip8(int32_t acc) {
  const int v = acc >> CMB_RESAMPLE_BITS;   // arithmetic shift, as clip8_lookups[in >> PRECISION_BITS]
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// the three bytes of virtual pixel (y, x) of the letter-boxed square, packed r | g << 8 | b << 16
CMB_PHD uint32_t cmb_virtual_pixel(const uint8_t* src, const cmb_image_job& J, int y, int x) {
  const int sy = y - J.off_y, sx = x - J.off_x;
  if ((unsigned)sy < (unsigned)J.h && (unsigned)sx < (unsigned)J.w) {
    const uint8_t* p = src + J.src_off + ((int64_t)sy * J.w + sx) * 3;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  }
  return J.background;
}

// pass H, one (y, xo): writes the three planes of tmp
CMB_PHD void cmb_resample_h(const uint8_t* src, const int32_t* bounds, const int32_t* coefs, uint8_t* tmp,
                            const cmb_image_job& J, int y, int xo) {
  const int R = J.out_side, pitch = cmb_tmp_pitch(R);
  const int first = bounds[J.bounds_off + 2 * xo], count = bounds[J.bounds_off + 2 * xo + 1];
  const int32_t* k = coefs + J.coef_off + xo;
  int32_t a0 = 1 << (CMB_RESAMPLE_BITS - 1), a1 = a0, a2 = a0;
  const int sy = y - J.off_y;
  if ((unsigned)sy < (unsigned)J.h) {
    const uint8_t* row = src + J.src_off + (int64_t)sy * J.w * 3;
    const int bg0 = J.background & 255, bg1 = (J.background >> 8) & 255, bg2 = (J.background >> 16) & 255;
    // one unaligned 32-bit load per pixel (r, g, b + the next pixel's r); only the very last pixel of the image has
    // no byte behind it and is read bytewise
    const bool last_row = sy == J.h - 1;
    for (int j = 0; j < count; ++j) {
      const int32_t kj = k[(int64_t)j * R];
      const int sx = first + j - J.off_x;
      int p0 = bg0, p1 = bg1, p2 = bg2;
      if ((unsigned)sx < (unsigned)J.w) {
        const uint8_t* px = row + sx * 3;
        if (last_row && sx == J.w - 1) {
          p0 = px[0]; p1 = px[1]; p2 = px[2];
        } else {
          uint32_t v;
          __builtin_memcpy(&v, px, 4);
          p0 = v & 255; p1 = (v >> 8) & 255; p2 = (v >> 16) & 255;
        }
      }
      a0 += p0 * kj; a1 += p1 * kj; a2 += p2 * kj;
    }
  } else {  // a bar row: a constant colour through the same fixed-point taps
    int32_t ks = 0;
    for (int j = 0; j < count; ++j) ks += k[(int64_t)j * R];
    a0 += (int32_t)(J.background & 255) * ks;
    a1 += (int32_t)((J.background >> 8) & 255) * ks;
    a2 += (int32_t)((J.background >> 16) & 255) * ks;
  }
  uint8_t* t = tmp + J.tmp_off + (int64_t)y * pitch + xo;
  const int64_t plane = (int64_t)J.side * pitch;
  t[0] = (uint8_t)cmb_clip8(a0);
  t[plane] = (uint8_t)cmb_clip8(a1);
  t[2 * plane] = (uint8_t)cmb_clip8(a2);
}

// pass V, one (yo, x4): four neighbouring columns of the three planes -> twelve 8-bit levels
// levels[c][i] for column 4*x4 + i (columns >= R are padding and must be ignored by the caller)
CMB_PHD void cmb_resample_v(const uint8_t* tmp, const int32_t* bounds, const int32_t* coefs,
                            const cmb_image_job& J, int yo, int x4, int (&levels)[3][4]) {
  const int R = J.out_side, pitch = cmb_tmp_pitch(R);
  const int first = bounds[J.bounds_off + 2 * yo], count = bounds[J.bounds_off + 2 * yo + 1];
  const int32_t* k = coefs + J.coef_off + yo;
  const int64_t plane = (int64_t)J.side * pitch;
  int32_t acc[3][4];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 4; ++i) acc[c][i] = 1 << (CMB_RESAMPLE_BITS - 1);
  const uint8_t* t = tmp + J.tmp_off + (int64_t)first * pitch + 4 * x4;
  for (int j = 0; j < count; ++j) {
    const int32_t kj = k[(int64_t)j * R];
    for (int c = 0; c < 3; ++c) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(t + c * plane + (int64_t)j * pitch);
      for (int i = 0; i < 4; ++i) acc[c][i] += (int32_t)((v >> (8 * i)) & 255u) * kj;
    }
  }
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 4; ++i) levels[c][i] = cmb_clip8(acc[c][i]);
}

// same-size job (Image.resize returns a copy): the levels are the virtual pixels themselves
CMB_PHD void cmb_copy_levels(const uint8_t* src, const cmb_image_job& J, int yo, int x4, int (&levels)[3][4]) {
  for (int i = 0; i < 4; ++i) {
    const int x = 4 * x4 + i;
    const uint32_t p = x < J.out_side ? cmb_virtual_pixel(src, J, yo, x) : 0u;
    levels[0][i] = p & 255; levels[1][i] = (p >> 8) & 255; levels[2][i] = (p >> 16) & 255;
  }
}

// ---- block -> job mapping ---------------------------------------------------------------------------------------
// Each pass is one 1-D grid; blocks are dealt to jobs through a prefix table passed by value in the kernel arguments.
#define CMB_PP_THREADS 256
#define CMB_PP_MAX_JOBS 256   // jobs per launch (larger tables are cut into several launches)

struct cmb_block_starts {
  int32_t n;
  int32_t start[CMB_PP_MAX_JOBS + 1];  // start[j] = first block of job j; start[n] = grid size
};

// largest j with start[j] <= b (b < start[n]; empty jobs have start[j] == start[j+1]); uniform over the block
CMB_PHD int cmb_job_of_block(const cmb_block_starts& t, int b) {
  int lo = 0, hi = t.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

CMB_PHD int64_t cmb_h_blocks(const cmb_image_job& J) {
  return J.ksize == 0 ? 0 : ((int64_t)J.side * J.out_side + CMB_PP_THREADS - 1) / CMB_PP_THREADS;
}
CMB_PHD int64_t cmb_v_blocks(const cmb_image_job& J) {
  return ((int64_t)J.out_side * (cmb_tmp_pitch(J.out_side) >> 2) + CMB_PP_THREADS - 1) / CMB_PP_THREADS;
}

// host: prefix tables of both passes for nj <= CMB_PP_MAX_JOBS jobs; false if a grid would not fit in 31 bits
static inline bool cmb_block_tables(const cmb_image_job* jobs, int nj, cmb_block_starts* hs, cmb_block_starts* vs,
                                    int64_t* h_total, int64_t* v_total) {
  int64_t hb = 0, vb = 0;
  hs->n = vs->n = nj;
  for (int i = 0; i < nj; ++i) {
    hs->start[i] = (int32_t)hb;
    vs->start[i] = (int32_t)vb;
    hb += cmb_h_blocks(jobs[i]);
    vb += cmb_v_blocks(jobs[i]);
    if (hb > 0x7fffffff || vb > 0x7fffffff) return false;
  }
  for (int i = nj; i <= CMB_PP_MAX_JOBS; ++i) { hs->start[i] = (int32_t)hb; vs->start[i] = (int32_t)vb; }
  *h_total = hb; *v_total = vb;
  return true;
}

// ---- host: coefficient rows (precompute_coeffs + normalize_coeffs_8bpc) ------------------------------------
#include <math.h>
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
static inline double cmb_bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static inline int cmb_resize_ksize(int32_t in_size, int32_t out_size) {
  double filterscale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}
static inline int cmb_resize_coeffs_host(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs) {
  if (in_size <= 0 || out_size <= 0) return CMB_ERR_BAD_ARG;
  const int ksize = cmb_resize_ksize(in_size, out_size);
  if (!bounds || !coefs) return ksize;
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
  double* w = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      w[x] = cmb_bicubic_filter((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
      const double f = v * (double)(1 << CMB_RESAMPLE_BITS);
      coefs[(int64_t)x * out_size + xx] = v < 0 ? (int32_t)(-0.5 + f) : (int32_t)(0.5 + f);
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] w;
  return ksize;
}
