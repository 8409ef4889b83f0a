// elementwise.hip — the HBM-bound data-movement kernels of the path: transpose, column sums, casts,
// activation helpers, token mean, embedding splice (bit-exact gather/scatter), patch gathers and the
// bilinear token-grid resampler.  All use 16/32-byte vector accesses along the contiguous dimension.
#include "common.h"

namespace {

inline unsigned grid_for(int64_t items, int threads = 256, int64_t cap = 65535) {
  int64_t b = (items + threads - 1) / threads;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// transpose: out[C, R_pad] = in[R, C]^T, zero fill for r >= R.  64x64 tiles through LDS.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, int64_t R, int64_t C,
                                                        int64_t ld_in, T* __restrict__ out, int64_t R_pad) {
  constexpr int TS = 64, LD = TS + 2;
  __shared__ T tile[TS * LD];
  const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = t / 8 + 32 * i, cv = t % 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t gr = r0 + row, gc = c0 + cv * 8;
    if (gr < R && gc < C) Vec8<T>::load(in + gr * ld_in + gc, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[row * LD + cv * 8 + e] = (T)v[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = t / 8 + 32 * i, rv = t % 8;
    const int64_t gc = c0 + c, gr = r0 + rv * 8;
    if (gc < C && gr < R_pad) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)tile[(rv * 8 + e) * LD + c];
      Vec8<T>::store(out + gc * R_pad + gr, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight_prep: for a TABLE of 2-D weights (fp32 masters or bf16), one launch writes each one's bf16 copy [rows, cols] and its
// transposed bf16 copy [cols, rows_pad] (zero columns rows .. rows_pad - 1) — the per-use cmb_cast + cmb_transpose pairs of the
// trainable linears (round 5: 171 + 166 launches of 5-12 us per step) as ONE pass over the masters (round 6).  A workgroup
// owns one 64 x 64 tile; the job of a tile is found by bisection over the jobs' first-tile prefix.
// ---------------------------------------------------------------------------------------------
template <bool ONE>
__global__ void __launch_bounds__(256) weight_prep_kernel(const cmb_prep_job one, const cmb_prep_job* __restrict__ jobs, int n_jobs) {
  constexpr int TS = 64, LD = TS + 2;
  __shared__ bf16_t tile[TS * LD];
  const int b = (int)blockIdx.x;
  const cmb_prep_job* jp = &one;
  if (!ONE) {
    int lo = 0, hi = n_jobs - 1;   // last job with tile0 <= b
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].tile0 <= b) lo = mid;
      else hi = mid - 1;
    }
    jp = jobs + lo;
  }
  const cmb_prep_job j = *jp;
  const int tiles_c = (j.cols + TS - 1) / TS;
  const int tl = b - j.tile0;
  const int64_t r0 = (int64_t)(tl / tiles_c) * TS, c0 = (int64_t)(tl % tiles_c) * TS;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = t / 8 + 32 * i, cv = t % 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t gr = r0 + row, gc = c0 + cv * 8;
    if (gr < j.rows && gc < j.cols) {
      if (j.src_dtype == CMB_F32) Vec8<float>::load(reinterpret_cast<const float*>(j.src) + gr * j.ld_src + gc, v);
      else Vec8<bf16_t>::load(reinterpret_cast<const bf16_t*>(j.src) + gr * j.ld_src + gc, v);
      if (j.dst) Vec8<bf16_t>::store(reinterpret_cast<bf16_t*>(j.dst) + gr * j.cols + gc, v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[row * LD + cv * 8 + e] = (bf16_t)v[e];
  }
  if (!j.dst_t) return;   // (uniform per workgroup)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = t / 8 + 32 * i, rv = t % 8;
    const int64_t gc = c0 + c, gr = r0 + rv * 8;
    if (gc < j.cols && gr < j.rows_pad) {
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tile[(rv * 8 + e) * LD + c];
      *reinterpret_cast<bf16x8_t*>(reinterpret_cast<bf16_t*>(j.dst_t) + gc * j.rows_pad + gr) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// colsum: out[c] += sum_r in[r, c]  (fp32 atomics, one per column per block); with a row scale (cmb_colsum_scaled):
// out[c] += sum_r scale[r, c / group] * in[r, c], group % 8 == 0 (the 8 columns of a lane share one factor)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ in, int64_t R, int64_t C, int64_t ld_in,
                                                     float* __restrict__ out, int rows_per_block,
                                                     const float* __restrict__ scale = nullptr, int64_t ld_scale = 0, int group = 8) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t cv = (int64_t)blockIdx.x * 64 + lane;  // vec8 column group
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per_block;
  const int64_t rend = (rbeg + rows_per_block < R) ? rbeg + rows_per_block : R;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cv * 8 < C) {
    for (int64_t r = rbeg + wave; r < rend; r += 4) {
      float v[8];
      Vec8<T>::load(in + r * ld_in + cv * 8, v);
      if (scale) {
        const float f = scale[r * ld_scale + (cv * 8) / group];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(f, v[e], acc[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int64_t col = (int64_t)blockIdx.x * 512 + i;
    if (col < C) atomicAdd(out + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    Vec8<TI>::load(in + i * 8, v);
    Vec8<TO>::store(out + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[(nv << 3) + threadIdx.x] = (TO)(float)in[(nv << 3) + threadIdx.x];
}

// y = act(a) * b (b may be null: y = act(a))
template <typename T>
__global__ void __launch_bounds__(256) act_mul_kernel(int act, const T* __restrict__ a, int64_t lda,
                                                      const T* __restrict__ b, int64_t ldb, int64_t rows, int D,
                                                      T* __restrict__ y, int64_t ldy) {
  const int nv = D >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv) * 8;
    float av[8], o[8];
    Vec8<T>::load(a + r * lda + c, av);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = act_apply(act, av[e]);
    if (b) {
      float bv[8];
      Vec8<T>::load(b + r * ldb + c, bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= bv[e];
    }
    Vec8<T>::store(y + r * ldy + c, o);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(int act, const T* __restrict__ dy, const T* __restrict__ pre,
                                                      int64_t n, T* __restrict__ dx) {
  const int64_t nv = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float g[8], p[8], o[8];
    Vec8<T>::load(dy + i * 8, g);
    Vec8<T>::load(pre + i * 8, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = g[e] * act_grad(act, p[e]);
    Vec8<T>::store(dx + i * 8, o);
  }
}

// out[b, :] = mean_t x[b, t, :].  A workgroup owns 64 columns of one image (8 lanes x 16 bytes = one 128-byte line per token
// row) and splits the tokens over its 32 lane groups; round 4: the first version gave a workgroup 512 columns — 48 workgroups
// for the release shape [24, 576, 1024], 44 us = 0.10 of the HBM peak (profiles/r03_hbm_kernels_table.md) — now 384.
template <typename T>
__global__ void __launch_bounds__(256) token_mean_kernel(const T* __restrict__ x, int Tn, int D, T* __restrict__ out) {
  __shared__ float red[32][65];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 7, tg = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cl * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < D) {
    for (int t = tg; t < Tn; t += 32) {
      float v[8];
      Vec8<T>::load(x + ((int64_t)b * Tn + t) * D + c0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tg][cl * 8 + e] = acc[e];
  __syncthreads();
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && col < D) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
    out[(int64_t)b * D + col] = (T)(s / (float)Tn);
  }
}

// acc[b, t, :] (fp32) += g[b, :] / Tn
template <typename T>
__global__ void __launch_bounds__(256) token_mean_bwd_kernel(const T* __restrict__ g, int64_t B, int Tn, int D,
                                                             float* __restrict__ acc) {
  const int nv = D >> 3;
  const int64_t total = B * Tn * nv;
  const float inv = 1.0f / (float)Tn;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nv;
    const int c = (int)(i - row * nv) * 8;
    const int64_t b = row / Tn;
    float gv[8], a[8];
    Vec8<T>::load(g + b * D + c, gv);
    load8f(acc + row * D + c, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += gv[e] * inv;
    Vec8<float>::store(acc + row * D + c, a);
  }
}

// ---------------------------------------------------------------------------------------------
// embedding splice (cambrian_arch.py:413-420,457-490)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) find_image_pos_kernel(const int64_t* __restrict__ ids, int S,
                                                             int64_t image_token, int32_t* __restrict__ pos) {
  __shared__ int best;
  if (threadIdx.x == 0) best = 0x7fffffff;
  __syncthreads();
  const int64_t* row = ids + (int64_t)blockIdx.x * S;
  int mine = 0x7fffffff;
  for (int t = threadIdx.x; t < S; t += blockDim.x)
    if (row[t] == image_token && t < mine) mine = t;
  if (mine != 0x7fffffff) atomicMin(&best, mine);
  __syncthreads();
  if (threadIdx.x == 0) pos[blockIdx.x] = (best == 0x7fffffff) ? -1 : best;
}

template <typename T>
__global__ void __launch_bounds__(256) embed_splice_fwd_kernel(const int64_t* __restrict__ ids, int64_t B, int S,
                                                               int H, int64_t image_token, const T* __restrict__ table,
                                                               int64_t vocab, const T* __restrict__ feat, int side,
                                                               const T* __restrict__ newline,
                                                               const int32_t* __restrict__ pos, T* __restrict__ out) {
  const int nv = H >> 3;
  const int span = side * (side + 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t rowi = (int64_t)blockIdx.x * 4 + wave; rowi < B * S; rowi += (int64_t)gridDim.x * 4) {
    const int64_t b = rowi / S;
    const int t = (int)(rowi - b * S);
    const int p = pos[b];
    const T* src;
    if (p >= 0 && t >= p && t < p + span) {
      const int v = t - p, i = v / (side + 1), j = v - i * (side + 1);
      src = (j < side) ? feat + ((int64_t)b * side * side + i * side + j) * H : newline;
    } else {
      int64_t id = ids[rowi];
      if (id == image_token) id = 0;
      if (id < 0) id = 0;
      if (id >= vocab) id = vocab - 1;
      src = table + id * H;
    }
    T* dst = out + rowi * H;
    for (int c = lane; c < nv; c += 64) {
      // raw 16/32-byte copy: bit-exact
      if (sizeof(T) == 2) {
        *reinterpret_cast<f32x4_t*>(dst + c * 8) = *reinterpret_cast<const f32x4_t*>(src + c * 8);
      } else {
        *reinterpret_cast<f32x4_t*>(dst + c * 8) = *reinterpret_cast<const f32x4_t*>(src + c * 8);
        *reinterpret_cast<f32x4_t*>(dst + c * 8 + 4) = *reinterpret_cast<const f32x4_t*>(src + c * 8 + 4);
      }
    }
  }
}

// dfeat gather + dnewline reduction.  grid.x over (b, i) grid rows.
template <typename T>
__global__ void __launch_bounds__(256) embed_splice_bwd_kernel(const T* __restrict__ dout, const int32_t* __restrict__ pos,
                                                               int64_t B, int S, int H, int side, T* __restrict__ dfeat,
                                                               float* __restrict__ dnewline) {
  const int nv = H >> 3;
  const int64_t b = blockIdx.x / side;
  const int i = blockIdx.x - (int)(b * side);
  const int p = pos[b];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j <= side; j += 4) {
    const int t = p + i * (side + 1) + j;
    const bool valid = (p >= 0) && (t < S);
    for (int c = lane; c < nv; c += 64) {
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) Vec8<T>::load(dout + ((int64_t)b * S + t) * H + c * 8, v);
      if (j < side) {
        Vec8<T>::store(dfeat + ((int64_t)b * side * side + i * side + j) * H + c * 8, v);
      } else if (valid && dnewline) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dnewline + c * 8 + e, v[e]);
      }
    }
  }
}

// dst[dmap(r), 0:D] = src[smap(r), 0:D]  (src == nullptr: zero fill) — row gather / scatter for the in-LLM
// SVA hook (cambrian_llama.py:181-207) and its backward
template <typename T>
__global__ void __launch_bounds__(256) copy_rows_kernel(const T* __restrict__ src, RowMap smap, T* __restrict__ dst,
                                                        RowMap dmap, int64_t rows, int D) {
  const int nv = D >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (src) Vec8<T>::load(src + row_off(smap, (uint32_t)r) + c, v);
    Vec8<T>::store(dst + row_off(dmap, (uint32_t)r) + c, v);
  }
}

// dst[r, 0:D] = src[0:D] for nrows rows of stride ld (CLS token rows)
template <typename T>
__global__ void __launch_bounds__(256) bcast_rows_kernel(T* __restrict__ dst, int64_t ld, int64_t nrows, int D,
                                                         const T* __restrict__ src) {
  const int nv = D >> 3;
  const int64_t total = nrows * nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv) * 8;
    float v[8];
    Vec8<T>::load(src + c, v);
    Vec8<T>::store(dst + r * ld + c, v);
  }
}

// ---------------------------------------------------------------------------------------------
// patch gathers
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) patchify_nchw_kernel(const TI* __restrict__ img, int64_t B, int C, int H, int W,
                                                            int p, TO* __restrict__ cols, int Kpad) {
  const int ph = H / p, pw = W / p;
  const int64_t total = B * ph * pw * (int64_t)Kpad;
  const int K = C * p * p;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / Kpad;
    const int k = (int)(i - row * Kpad);
    float v = 0.f;
    if (k < K) {
      const int c = k / (p * p), rem = k - c * p * p, dy = rem / p, dx = rem - dy * p;
      const int64_t b = row / (ph * pw);
      const int pr = (int)(row - b * ph * pw), py = pr / pw, px = pr - py * pw;
      v = (float)img[((b * C + c) * H + (py * p + dy)) * (int64_t)W + px * p + dx];
    }
    cols[i] = (TO)v;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) patchify2x2_kernel(const T* __restrict__ x, int64_t B, int H, int W, int C,
                                                          T* __restrict__ cols) {
  const int nv = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = B * Ho * Wo * 4 * (int64_t)nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % nv);
    int64_t rest = i / nv;
    const int seg = (int)(rest & 3);
    rest >>= 2;  // output row (b, oy, ox)
    const int ox = (int)(rest % Wo);
    const int64_t r2 = rest / Wo;
    const int oy = (int)(r2 % Ho);
    const int64_t b = r2 / Ho;
    const int dy = seg >> 1, dx = seg & 1;
    float v[8];
    Vec8<T>::load(x + (((b * H + 2 * oy + dy) * W) + 2 * ox + dx) * (int64_t)C + cv * 8, v);
    Vec8<T>::store(cols + rest * 4 * C + (int64_t)seg * C + cv * 8, v);
  }
}

// ---------------------------------------------------------------------------------------------
// bilinear resample, align_corners = False, fp32 lerp (torch upsample_bilinear2d semantics)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_index(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

template <typename T>
__global__ void __launch_bounds__(256) resample_kernel(const T* __restrict__ in, int64_t B, int Hi, int Wi, int C,
                                                       int64_t ld_in, int64_t bs_in, T* __restrict__ out, int Ho,
                                                       int Wo, int64_t ld_out, int64_t bs_out) {
  const int nv = C >> 3;
  const int64_t total = B * Ho * Wo * (int64_t)nv;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % nv);
    int64_t rest = i / nv;
    const int ox = (int)(rest % Wo);
    rest /= Wo;
    const int oy = (int)(rest % Ho);
    const int64_t b = rest / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    lerp_index(oy, sy, Hi, y0, y1, ly);
    lerp_index(ox, sx, Wi, x0, x1, lx);
    const T* base = in + b * bs_in + cv * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    Vec8<T>::load(base + ((int64_t)y0 * Wi + x0) * ld_in, v00);
    Vec8<T>::load(base + ((int64_t)y0 * Wi + x1) * ld_in, v01);
    Vec8<T>::load(base + ((int64_t)y1 * Wi + x0) * ld_in, v10);
    Vec8<T>::load(base + ((int64_t)y1 * Wi + x1) * ld_in, v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    Vec8<T>::store(out + b * bs_out + ((int64_t)oy * Wo + ox) * ld_out + cv * 8, o);
  }
}

}  // namespace

#define DT_SWITCH(dtype, ...)                                   \
  if ((dtype) == CMB_BF16) { typedef bf16_t T; __VA_ARGS__; }   \
  else if ((dtype) == CMB_F32) { typedef float T; __VA_ARGS__; } \
  else return CMB_ERR_BAD_ARG;

extern "C" int cmb_transpose(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in, void* out,
                             int64_t R_pad, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || (C & 7) || (R_pad & 7) || R_pad < R || (ld_in & 7)) return CMB_ERR_BAD_ARG;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R_pad + 63) / 64));
  DT_SWITCH(dtype, hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in, R, C,
                                      ld_in, (T*)out, R_pad));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

static int prep_job_ok(const cmb_prep_job& j) {
  return j.src && (j.dst || j.dst_t) && (j.src_dtype == CMB_F32 || j.src_dtype == CMB_BF16) && j.rows > 0 && j.cols > 0 &&
         !(j.cols & 7) && !(j.rows_pad & 7) && j.rows_pad >= j.rows && !(j.ld_src & 7) && j.ld_src >= j.cols;
}

extern "C" int64_t cmb_weight_prep_tiles(int64_t rows_pad, int64_t cols) { return ((rows_pad + 63) / 64) * ((cols + 63) / 64); }

extern "C" int cmb_weight_prep_one(const cmb_prep_job* job, void* stream) {
  if (!job || !prep_job_ok(*job)) return CMB_ERR_BAD_ARG;
  cmb_prep_job j = *job;
  j.tile0 = 0;
  const int64_t tiles = cmb_weight_prep_tiles(j.rows_pad, j.cols);
  if (tiles > 0x7fffffff) return CMB_ERR_BAD_ARG;
  hipLaunchKernelGGL(weight_prep_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, j, nullptr, 1);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_weight_prep(const cmb_prep_job* jobs_device, int32_t n_jobs, int64_t total_tiles, void* stream) {
  if (!jobs_device || n_jobs <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffff) return CMB_ERR_BAD_ARG;
  cmb_prep_job none = {};
  hipLaunchKernelGGL(weight_prep_kernel<false>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, none,
                     jobs_device, n_jobs);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

// rows per workgroup of the column sums: 256 for the tall inputs (221 184 rows), fewer when that would leave most of the chip
// idle — the SVA query-side inputs are 13 824 x 1024: 108 workgroups of 256 rows ran 27-60 us for 28 MB.  The launch aims at
// CMB_KNOB_COLSUM_WGS workgroups (one atomicAdd per column and workgroup either way; 0 = always 256 rows, rounds 1-5)
static int colsum_rows_per_block(int64_t R, int64_t C) {
  const int target = cmb_knob(CMB_KNOB_COLSUM_WGS);
  if (target <= 0) return 256;
  const int64_t col_groups = (C + 511) / 512;
  int64_t rpb = (R * col_groups / target + 3) & ~(int64_t)3;
  if (rpb < 16) rpb = 16;
  if (rpb > 256) rpb = 256;
  return (int)rpb;
}

extern "C" int cmb_colsum(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in, float* out, void* stream) {
  if (!in || !out || R < 0 || C <= 0 || (C & 7) || (ld_in & 7)) return CMB_ERR_BAD_ARG;
  if (R == 0) return CMB_OK;
  const int rows_per_block = colsum_rows_per_block(R, C);
  dim3 grid((unsigned)((C + 511) / 512), (unsigned)((R + rows_per_block - 1) / rows_per_block));
  DT_SWITCH(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in, R, C,
                                      ld_in, out, rows_per_block));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_colsum_scaled(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in, const float* row_scale,
                                 int64_t ld_scale, int32_t group, float* out, void* stream) {
  if (!in || !out || !row_scale || R < 0 || C <= 0 || (C & 7) || group <= 0 || (group & 7) || C % group) return CMB_ERR_BAD_ARG;
  if (R == 0) return CMB_OK;
  const int rows_per_block = colsum_rows_per_block(R, C);
  dim3 grid((unsigned)((C + 511) / 512), (unsigned)((R + rows_per_block - 1) / rows_per_block));
  DT_SWITCH(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in, R, C, ld_in, out,
                                      rows_per_block, row_scale, ld_scale, (int)group));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_cast(int from_dtype, const void* in, int to_dtype, void* out, int64_t n, void* stream) {
  if (!in || !out || n < 0) return CMB_ERR_BAD_ARG;
  if (n == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned g = grid_for((n + 7) / 8, 256, 8192);
  if (from_dtype == CMB_F32 && to_dtype == CMB_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n);
  else if (from_dtype == CMB_BF16 && to_dtype == CMB_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n);
  else if (from_dtype == CMB_F32 && to_dtype == CMB_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)in, (float*)out, n);
  else if (from_dtype == CMB_BF16 && to_dtype == CMB_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, n);
  else
    return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_act_mul(int dtype, int32_t act, const void* a, int64_t lda, const void* b, int64_t ldb,
                           int64_t rows, int64_t D, void* y, int64_t ldy, void* stream) {
  if (!a || !y || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  DT_SWITCH(dtype, hipLaunchKernelGGL(act_mul_kernel<T>, dim3(grid_for(rows * (D / 8), 256, 16384)), dim3(256), 0,
                                      (hipStream_t)stream, act, (const T*)a, lda, (const T*)b, ldb, rows, (int)D,
                                      (T*)y, ldy));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_act_bwd(int dtype, int32_t act, const void* dy, const void* pre, int64_t n, void* dx, void* stream) {
  if (!dy || !pre || !dx || n < 0 || (n & 7)) return CMB_ERR_BAD_ARG;
  if (n == 0) return CMB_OK;
  DT_SWITCH(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(grid_for(n / 8, 256, 16384)), dim3(256), 0,
                                      (hipStream_t)stream, act, (const T*)dy, (const T*)pre, n, (T*)dx));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_token_mean_fwd(int dtype, const void* x, int64_t B, int64_t Tn, int64_t D, void* out, void* stream) {
  if (!x || !out || B < 0 || Tn <= 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  dim3 grid((unsigned)((D + 63) / 64), (unsigned)B);
  DT_SWITCH(dtype, hipLaunchKernelGGL(token_mean_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                      (int)Tn, (int)D, (T*)out));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_token_mean_bwd(int dtype, const void* g, int64_t B, int64_t Tn, int64_t D, float* acc, void* stream) {
  if (!g || !acc || B < 0 || Tn <= 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  DT_SWITCH(dtype, hipLaunchKernelGGL(token_mean_bwd_kernel<T>, dim3(grid_for(B * Tn * (D / 8), 256, 16384)),
                                      dim3(256), 0, (hipStream_t)stream, (const T*)g, B, (int)Tn, (int)D, acc));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_embed_splice_fwd(int dtype, const int64_t* ids, int64_t B, int64_t S, int64_t H,
                                    int64_t image_token, const void* table, int64_t vocab, const void* feat,
                                    int32_t side, const void* newline, void* out, int32_t* pos, void* stream) {
  if (!ids || !table || !feat || !newline || !out || !pos || B < 0 || S <= 0 || H <= 0 || (H & 7) || side <= 0)
    return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(find_image_pos_kernel, dim3((unsigned)B), dim3(256), 0, s, ids, (int)S, image_token, pos);
  DT_SWITCH(dtype, hipLaunchKernelGGL(embed_splice_fwd_kernel<T>, dim3(grid_for(B * S, 4, 16384)), dim3(256), 0, s, ids,
                                      B, (int)S, (int)H, image_token, (const T*)table, vocab, (const T*)feat, side,
                                      (const T*)newline, pos, (T*)out));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_embed_splice_bwd(int dtype, const void* dout, const int32_t* pos, int64_t B, int64_t S, int64_t H,
                                    int32_t side, void* dfeat, float* dnewline, void* stream) {
  if (!dout || !pos || !dfeat || B < 0 || S <= 0 || H <= 0 || (H & 7) || side <= 0) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  DT_SWITCH(dtype, hipLaunchKernelGGL(embed_splice_bwd_kernel<T>, dim3((unsigned)(B * side)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)dout, pos, B, (int)S, (int)H, side, (T*)dfeat,
                                      dnewline));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_copy_rows(int dtype, const void* src, const cmb_rowmap* src_map, void* dst,
                             const cmb_rowmap* dst_map, int64_t rows, int64_t D, void* stream) {
  if (!dst || !dst_map || (src && !src_map) || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  cmb_rowmap ident = {0, 1, 0, 0, D};
  const RowMap sm = make_rowmap(src ? *src_map : ident), dm = make_rowmap(*dst_map);
  DT_SWITCH(dtype, hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(grid_for(rows * (D / 8), 256, 16384)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)src, sm, (T*)dst, dm, rows, (int)D));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_bcast_rows(int dtype, void* dst, int64_t ld, int64_t nrows, int64_t D, const void* src,
                              void* stream) {
  if (!dst || !src || nrows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (nrows == 0) return CMB_OK;
  DT_SWITCH(dtype, hipLaunchKernelGGL(bcast_rows_kernel<T>, dim3(grid_for(nrows * (D / 8), 256, 4096)), dim3(256), 0,
                                      (hipStream_t)stream, (T*)dst, ld, nrows, (int)D, (const T*)src));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_patchify_nchw(int in_dtype, const void* img, int64_t B, int64_t C, int64_t H, int64_t W, int32_t p,
                                 int out_dtype, void* cols, int64_t Kpad, void* stream) {
  // H, W need not be multiples of p: like a stride-p convolution the trailing H % p rows / columns are
  // dropped (SigLIP-SO400M runs 384 px with 14 px patches -> 27 x 27)
  if (!img || !cols || B < 0 || C <= 0 || p <= 0 || H < p || W < p || Kpad < C * p * p) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = B * (H / p) * (W / p) * Kpad;
  const unsigned g = grid_for(total, 256, 32768);
#define PATCH_LAUNCH(TI, TO) \
  hipLaunchKernelGGL((patchify_nchw_kernel<TI, TO>), dim3(g), dim3(256), 0, s, (const TI*)img, B, (int)C, (int)H, (int)W, p, (TO*)cols, (int)Kpad)
  if (in_dtype == CMB_F32 && out_dtype == CMB_BF16) PATCH_LAUNCH(float, bf16_t);
  else if (in_dtype == CMB_F32 && out_dtype == CMB_F32) PATCH_LAUNCH(float, float);
  else if (in_dtype == CMB_BF16 && out_dtype == CMB_BF16) PATCH_LAUNCH(bf16_t, bf16_t);
  else if (in_dtype == CMB_BF16 && out_dtype == CMB_F32) PATCH_LAUNCH(bf16_t, float);
  else return CMB_ERR_BAD_ARG;
#undef PATCH_LAUNCH
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_patchify2x2_nhwc(int dtype, const void* x, int64_t B, int64_t H, int64_t W, int64_t C, void* cols,
                                    void* stream) {
  if (!x || !cols || B < 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  const int64_t total = B * (H / 2) * (W / 2) * 4 * (C / 8);
  DT_SWITCH(dtype, hipLaunchKernelGGL(patchify2x2_kernel<T>, dim3(grid_for(total, 256, 32768)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)x, B, (int)H, (int)W, (int)C, (T*)cols));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_resample_bilinear(int dtype, const void* in, int64_t B, int32_t Hi, int32_t Wi, int64_t C,
                                     int64_t ld_in, int64_t batch_stride_in, void* out, int32_t Ho, int32_t Wo,
                                     int64_t ld_out, int64_t batch_stride_out, void* stream) {
  if (!in || !out || B < 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 7) || (ld_in & 7) ||
      (ld_out & 7))
    return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  const int64_t total = B * Ho * Wo * (C / 8);
  DT_SWITCH(dtype, hipLaunchKernelGGL(resample_kernel<T>, dim3(grid_for(total, 256, 32768)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)in, B, Hi, Wi, (int)C, ld_in, batch_stride_in,
                                      (T*)out, Ho, Wo, ld_out, batch_stride_out));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" const char* cmb_version(void) { return "cambrian_amd 0.1.0 gfx950"; }
extern "C" int cmb_abi_version(void) { return CMB_ABI_VERSION; }

int g_cmb_knobs[CMB_KNOB_COUNT] = {CMB_KNOB_DEFAULTS};
extern "C" int cmb_knob_set(int32_t knob, int32_t value) {
  if (knob < 0 || knob >= CMB_KNOB_COUNT) return CMB_ERR_BAD_ARG;
  // every knob has a closed value set: a typo in an A/B run must fail here, not launch a bogus grid or silently select
  // another variant
  bool ok = false;
  switch (knob) {
    case CMB_KNOB_LN_FWD: ok = value >= 0 && value <= 65536; break;           // 0 / 1 / workgroup cap
    case CMB_KNOB_DWCONV: ok = value >= 0 && value <= 4096; break;            // 0 / 1 / rows per chunk
    case CMB_KNOB_VIT_ATTN: ok = value >= 0 && value <= 3; break;              // 2 = LDS-DMA tiles + transposing reads (round 6)
    case CMB_KNOB_SVA_ABS: ok = value == 0 || value == 1; break;
    case CMB_KNOB_LN_MULTI_CHUNK: ok = value >= 4 && value <= 7; break;
    case CMB_KNOB_COLSUM_WGS: ok = value >= 0 && value <= 65536; break;        // 0 = 256 rows per workgroup / workgroup target
    case CMB_KNOB_LN_BWD_ROWS: ok = value >= 4 && value <= 256; break;
    case CMB_KNOB_FLASH: ok = value >= 0 && value <= 31 && !(value & 8); break;   // bit mask: 1 forward, 2 dQ, 4 dK/dV body, 16 dK/dV transposing reads (8: removed)
    default: ok = value >= 0; break;
  }
  if (!ok) return CMB_ERR_BAD_ARG;
  g_cmb_knobs[knob] = value;
  return CMB_OK;
}
extern "C" int cmb_knob_get(int32_t knob) { return (knob < 0 || knob >= CMB_KNOB_COUNT) ? -1 : g_cmb_knobs[knob]; }
