// norm.hip — LayerNorm / RMSNorm forward + backward and RoPE for gfx950.  All HBM-bound:
// one wave (64 lanes) owns one row, 16-byte (bf16) / 32-byte (fp32) vector accesses, the row is
// held in registers between the statistics pass and the normalisation pass (one HBM read, one
// write per element), fp32 math throughout.
#include "common.h"

namespace {

// token row -> window-local position on a side x side grid partitioned into grid_r x grid_r windows
__device__ __forceinline__ int window_pos(uint32_t row, int side, int grid_r) {
  const uint32_t t = row % (uint32_t)(side * side);
  const uint32_t y = t / (uint32_t)side, x = t - y * (uint32_t)side;
  return (int)((y % (uint32_t)grid_r) * (uint32_t)grid_r + (x % (uint32_t)grid_r));
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward.  NCH = vec8 chunks per lane (D <= NCH*512).
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(
    const T* __restrict__ x, int64_t rows, int D, int64_t ldx, const float* __restrict__ add, int side,
    int grid_r, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    T* __restrict__ y, int64_t ldy, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  // the affine parameters live in registers across the wave's rows (re-loading 2 x 4 D bytes of fp32 per 2 D-byte row
  // put 4x the row's own bytes through the vector cache: 3.6-4.0 TB/s against 4.8 for the non-affine SVA variant)
  // Only for rows of up to 1024 elements (NCH <= 2: 32 more registers); wider rows would give up waves per SIMD for it.
  constexpr bool kHoist = NCH <= 2;
  float gg[kHoist ? NCH : 1][8], bb[kHoist ? NCH : 1][8];
  if (kHoist && gamma) {
#pragma unroll
    for (int c = 0; c < (kHoist ? NCH : 1); ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        load8f(gamma + vi * 8, gg[c]);
        load8f(beta + vi * 8, bb[c]);
      }
    }
  }
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    float v[NCH][8];
    const T* xr = x + row * ldx;
    const float* ar = add ? add + (int64_t)window_pos((uint32_t)row, side, grid_r) * D : nullptr;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        Vec8<T>::load(xr + vi * 8, v[c]);
        if (ar) {
          float a[8];
          load8f(ar + vi * 8, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] += a[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[c][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    T* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd;
        if (gamma) {
          if constexpr (kHoist) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * gg[c][e] + bb[c][e];
          } else {
            float g8[8], b8[8];
            load8f(gamma + vi * 8, g8);
            load8f(beta + vi * 8, b8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * g8[e] + b8[e];
          }
        }
        Vec8<T>::store(yr + vi * 8, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward of SEVERAL non-affine LayerNorms of the same input (round 6): the 13 SVA layers normalise the windowed tower's
// 221 184 x 1024 tokens once each, only the position table differs (vision_sampler.py:304-309) — 13 launches of the kernel
// above read x 13 times (52 bytes per element; 13 x 204 us per 24-image step).  Here a wave holds its row of x in registers
// and walks the layers: 2 + 2 L bytes per element (28 at L = 13), the next row's loads in flight while the current row's L
// outputs are reduced and written.  Per layer the arithmetic and its order are layernorm_fwd_kernel's: bit-identical.
// ------------------------------------------------------------------------------------------------
struct LnFwdMultiParams {
  int layers;
  const float* add[CMB_LN_MULTI_MAX];
  void* y[CMB_LN_MULTI_MAX];
  float* mean[CMB_LN_MULTI_MAX];
  float* rstd[CMB_LN_MULTI_MAX];
};

template <typename T, int NCH>
__global__ void __launch_bounds__(256) layernorm_fwd_multi_kernel(const T* __restrict__ x, int64_t rows, int D, int64_t ldx,
                                                                  int side, int grid_r, float eps, const LnFwdMultiParams mp) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  float nxt[NCH][8];
  auto fetch = [&](int64_t row) {
    if (row < rows) {
      const T* xr = x + row * ldx;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) Vec8<T>::load(xr + vi * 8, nxt[c]);
      }
    }
  };
  fetch(wave_global);
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    float xv[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[c][e] = nxt[c][e];
    fetch(row + nwaves);
    const int wp = window_pos((uint32_t)row, side, grid_r);
    for (int l = 0; l < mp.layers; ++l) {
      const float* ar = mp.add[l] ? mp.add[l] + (int64_t)wp * D : nullptr;
      float v[NCH][8];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] = xv[c][e];
          if (ar) {
            float a[8];
            load8f(ar + vi * 8, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] += a[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) s += v[c][e];
        }
      }
      const float mean = wave_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[c][e] - mean;
            q += d * d;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
      T* yr = (T*)mp.y[l] + row * (int64_t)D;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd;
          Vec8<T>::store(yr + vi * 8, o);
        }
      }
      if (lane == 0) {
        mp.mean[l][row] = mean;
        mp.rstd[l][row] = rstd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward, affine, no position table (the towers' LayerNorms: ViT blocks, ConvNeXt blocks): round 4.
// The kernel above re-loads gamma and beta (2 x 4 D bytes of fp32) for every 2 D-byte row once D > 1024 — four times
// the row's own bytes through the vector cache — and walks its rows one load -> reduce -> store round trip at a time
// (3.6-4.0 TB/s on ConvNeXt stage 3 against 5+ for the path's other streaming kernels, profiles/r03_hbm_kernels_table.md).
// Here the parameters are staged in LDS once per workgroup, and a wave keeps the NEXT row's loads in flight (raw 16-byte
// vectors in registers) while it reduces and writes the current one.  Arithmetic and summation order are those of
// layernorm_fwd_kernel: the two kernels are bit-identical (tests/test_kernels_gpu.py).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  bf16x8_t v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8_t*>(p); }
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};
template <> struct Raw8<float> {
  f32x4_t a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4_t*>(p);
    b = *reinterpret_cast<const f32x4_t*>(p + 4);
  }
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  }
};

template <typename T, int NCH>
__global__ void __launch_bounds__(256) layernorm_fwd_lds_kernel(
    const T* __restrict__ x, int64_t rows, int D, int64_t ldx, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, T* __restrict__ y, int64_t ldy, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  extern __shared__ __attribute__((aligned(16))) float ln_sm[];  // gamma[D] then beta[D]
  const int lane = threadIdx.x & 63;
  const int nvec = D >> 3;
  for (int i = threadIdx.x; i < nvec; i += 256) {
    float g8[8], b8[8];
    load8f(gamma + i * 8, g8);
    load8f(beta + i * 8, b8);
    Vec8<float>::store(ln_sm + i * 8, g8);
    Vec8<float>::store(ln_sm + D + i * 8, b8);
  }
  __syncthreads();
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  Raw8<T> cur[NCH], nxt[NCH];
  int64_t row = wave_global;
  if (row < rows) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) cur[c].load(x + row * ldx + vi * 8);
    }
  }
  for (; row < rows; row += nwaves) {
    const int64_t nrow = row + nwaves;
    if (nrow < rows) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) nxt[c].load(x + nrow * ldx + vi * 8);
      }
    }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        cur[c].get(v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[c][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    T* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float o[8], g8[8], b8[8];
        load8f(ln_sm + vi * 8, g8);
        load8f(ln_sm + D + vi * 8, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * g8[e] + b8[e];
        Vec8<T>::store(yr + vi * 8, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
  }
}

// Row statistics only (round 6): mean / rstd of every row, the arithmetic of layernorm_fwd_lds_kernel operation for operation
// (two passes over the row in registers, next row prefetched), no normalised output — the LayerNorm itself is applied by the
// epilogue of the linear that follows (cmb_gemm_desc.row_mean): half of a LayerNorm's HBM traffic.
template <typename T, int NCH>
__global__ void __launch_bounds__(256) row_stats_kernel(const T* __restrict__ x, int64_t rows, int D, int64_t ldx, float eps,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int nvec = D >> 3;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  Raw8<T> cur[NCH], nxt[NCH];
  int64_t row = wave_global;
  if (row < rows) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) cur[c].load(x + row * ldx + vi * 8);
    }
  }
  for (; row < rows; row += nwaves) {
    const int64_t nrow = row + nwaves;
    if (nrow < rows) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) nxt[c].load(x + nrow * ldx + vi * 8);
      }
    }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        cur[c].get(v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[c][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  Rows are enumerated window-major so that all rows a block touches share one
// position-table row: blockIdx.y = pos (0..grid_r^2-1), and "window" w enumerates
// (b, qy, qx); token row = b*side^2 + (qy*grid_r+py)*side + qx*grid_r+px.  With grid_r == 1 this
// is the identity enumeration.  Per-lane partial sums of dgamma / dbeta / dadd live in registers,
// are combined across the block's 4 waves through LDS and leave as one atomicAdd per column per block.
// ------------------------------------------------------------------------------------------------
template <typename T, typename TDx, int NCH, bool ACCUM>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, int64_t rows, int D,
    const float* __restrict__ add, int side, int grid_r, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, TDx* __restrict__ dx,
    int64_t lddx, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dadd) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* sred = reinterpret_cast<float*>(dyn_smem);  // [4 waves][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D >> 3;
  const int pos = blockIdx.y;
  const int py = pos / grid_r, px = pos - py * grid_r;
  const int qside = side / grid_r;
  const int64_t nwin = rows / ((int64_t)grid_r * grid_r);  // windows (= rows when grid_r == 1)
  const float* ar = add ? add + (int64_t)pos * D : nullptr;

  float pg[NCH][8], pb[NCH][8], pa[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[c][e] = 0.f; pb[c][e] = 0.f; pa[c][e] = 0.f; }

  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t w = wave_global; w < nwin; w += nwaves) {
    int64_t row;
    if (grid_r == 1) {
      row = w;
    } else {
      const int64_t b = w / ((int64_t)qside * qside);
      const int t = (int)(w - b * (int64_t)qside * qside);
      const int qy = t / qside, qx = t - qy * qside;
      row = b * (int64_t)side * side + (int64_t)(qy * grid_r + py) * side + (qx * grid_r + px);
    }
    const float mean = mean_in[row], rstd = rstd_in[row];
    const T* xr = x + row * ldx;
    const T* dyr = dy + row * lddy;
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float xv[8], dv[8];
        Vec8<T>::load(xr + vi * 8, xv);
        Vec8<T>::load(dyr + vi * 8, dv);
        if (ar) {
          float a[8];
          load8f(ar + vi * 8, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] += a[e];
        }
        float gg[8];
        if (gamma) load8f(gamma + vi * 8, gg);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[c][e] = (xv[e] - mean) * rstd;
          if (gamma) {
            pg[c][e] += dv[e] * xh[c][e];
            pb[c][e] += dv[e];
            g[c][e] = dv[e] * gg[e];
          } else {
            g[c][e] = dv[e];
          }
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)D;
    const float c2 = wave_sum(s2) / (float)D;
    TDx* dxr = dx + row * lddx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (g[c][e] - c1 - xh[c][e] * c2) * rstd;
          pa[c][e] += o[e];
        }
        if (ACCUM) {
          float old[8];
          Vec8<TDx>::load(dxr + vi * 8, old);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += old[e];
        }
        Vec8<TDx>::store(dxr + vi * 8, o);
      }
    }
  }

  // block-level combine of the partial sums, one quantity at a time through sred[4][D]
  auto combine = [&](float (&part)[NCH][8], float* out) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sred[wave * D + vi * 8 + e] = part[c][e];
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256) {
      const float t = sred[col] + sred[D + col] + sred[2 * D + col] + sred[3 * D + col];
      if (t != 0.f) atomicAdd(out + col, t);
    }
  };
  if (gamma && dgamma) combine(pg, dgamma);
  if (gamma && dbeta) combine(pb, dbeta);
  if (ar && dadd) combine(pa, dadd + (int64_t)pos * D);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward of SEVERAL non-affine LayerNorms of the SAME input (round 4): the 13 SVA layers normalise one aux
// feature tensor each with its own position table, xh_l = normalise(x + pos_l[window position]) (vision_sampler.py:304-309,
// cambrian_arch.py:271-287), and d(x) = sum_l J_l^T d(xh_l).  Layer by layer (layernorm_bwd_kernel with the fp32
// accumulator) that is 13 x (dy 2 + x 2 + accumulator 4 + 4) = 156 bytes per element of the 9216-token tower — 7.7 ms per
// 24-image step at 4.6 TB/s.  Here the layers' d(xh_l) are kept until the last of them has run (5.9 GB at 24 images), and
// ONE pass reads x once, the L gradients once each and writes d(x) once: 2 + 2 L + 4 bytes per element (32 at L = 13).
// Rows are enumerated window-major as above (blockIdx.y = window position), so a lane's d(pos_l) partial sums are registers
// — LC sets per launch (a first version kept L x D sums in LDS and added to them with ds_add_f32: 208 LDS atomics per row
// made the pass 2x SLOWER than layer by layer); a launch therefore covers 4 layers (13 layers = 4 + 4 + 4 + 1: 62 bytes
// per element; 7 per launch needs 320 registers = one wave per SIMD and measured SLOWER: 6.6 ms against 4.7 ms, and 6.9 ms
// layer by layer, 24 images, profiles/r04_lab.md), every layer's gradient row requested before the first is reduced.
// ------------------------------------------------------------------------------------------------
struct LnMultiParams {
  int layers;
  void* dx_out;   // non-null: this launch writes the finished sum here in x's dtype (dense rows) instead of updating dx
  const void* dy[CMB_LN_MULTI_MAX];
  const float* add[CMB_LN_MULTI_MAX];
  const float* mean[CMB_LN_MULTI_MAX];
  const float* rstd[CMB_LN_MULTI_MAX];
  float* dadd[CMB_LN_MULTI_MAX];
};

template <typename T, int NCH, bool ACCUM, int LC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LC <= 5 ? 2 : 1)))
layernorm_bwd_multi_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int D,
                                                                  int side, int grid_r, const LnMultiParams mp,
                                                                  float* __restrict__ dx, int64_t lddx) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* sred = reinterpret_cast<float*>(dyn_smem);  // [4 waves][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D >> 3;
  const int pos = blockIdx.y;
  const int py = pos / grid_r, px = pos - py * grid_r;
  const int qside = side / grid_r;
  const int64_t nwin = rows / ((int64_t)grid_r * grid_r);
  // d(pos_l) partial sums of this lane's columns, one set per layer of the launch (LC <= 7 layers: 7 x NCH x 8 registers)
  float pa[LC][NCH][8];
#pragma unroll
  for (int l = 0; l < LC; ++l)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) pa[l][c][e] = 0.f;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t w = wave_global; w < nwin; w += nwaves) {
    int64_t row;
    if (grid_r == 1) {
      row = w;
    } else {
      const int64_t b = w / ((int64_t)qside * qside);
      const int t = (int)(w - b * (int64_t)qside * qside);
      const int qy = t / qside, qx = t - qy * qside;
      row = b * (int64_t)side * side + (int64_t)(qy * grid_r + py) * side + (qx * grid_r + px);
    }
    float xv[NCH][8], acc[NCH][8];
    // every layer's gradient row is requested up front: LC + 1 round trips in flight per wave instead of one after another
    Raw8<T> raw[LC][NCH];
#pragma unroll
    for (int l = 0; l < LC; ++l) {
      if (l < mp.layers) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int vi = lane + c * 64;
          if (vi < nvec) raw[l][c].load(reinterpret_cast<const T*>(mp.dy[l]) + row * (int64_t)D + vi * 8);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        Vec8<T>::load(x + row * ldx + vi * 8, xv[c]);
        if (ACCUM) load8f(dx + row * lddx + vi * 8, acc[c]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int l = 0; l < LC; ++l) {
      if (l < mp.layers) {   // wave-uniform
        const float mean = mp.mean[l][row], rstd = mp.rstd[l][row];
        const float* ar = mp.add[l] ? mp.add[l] + (int64_t)pos * D : nullptr;
        float xh[NCH][8], g[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int vi = lane + c * 64;
          if (vi < nvec) {
            raw[l][c].get(g[c]);
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ar) load8f(ar + vi * 8, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              xh[c][e] = (xv[c][e] + a[e] - mean) * rstd;
              s1 += g[c][e];
              s2 += g[c][e] * xh[c][e];
            }
          }
        }
        const float c1 = wave_sum(s1) / (float)D;
        const float c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int vi = lane + c * 64;
          if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float o = (g[c][e] - c1 - xh[c][e] * c2) * rstd;
              acc[c][e] += o;
              pa[l][c][e] += o;
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        if (mp.dx_out) Vec8<T>::store(reinterpret_cast<T*>(mp.dx_out) + row * (int64_t)D + vi * 8, acc[c]);   // (block-uniform)
        else Vec8<float>::store(dx + row * lddx + vi * 8, acc[c]);
      }
    }
  }
  // block-level combine of each layer's partial sums through sred[4][D], then one atomicAdd per column per workgroup
#pragma unroll
  for (int l = 0; l < LC; ++l) {
    if (l < mp.layers && mp.add[l] && mp.dadd[l]) {   // block-uniform
      __syncthreads();
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int vi = lane + c * 64;
        if (vi < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sred[wave * D + vi * 8 + e] = pa[l][c][e];
        }
      }
      __syncthreads();
      float* out = mp.dadd[l] + (int64_t)pos * D;
      for (int col = threadIdx.x; col < D; col += 256) {
        const float t = sred[col] + sred[D + col] + sred[2 * D + col] + sred[3 * D + col];
        if (t != 0.f) atomicAdd(out + col, t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ void __launch_bounds__(256) rmsnorm_fwd_kernel(const T* __restrict__ x, int64_t rows, int D,
                                                          const float* __restrict__ w, float eps,
                                                          T* __restrict__ y, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    float v[NCH][8];
    const T* xr = x + row * (int64_t)D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        Vec8<T>::load(xr + vi * 8, v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e] * v[c][e];
      }
    }
    const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
    T* yr = y + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float ww[8], o[8];
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ww[e] * (v[c][e] * rstd);
        Vec8<T>::store(yr + vi * 8, o);
      }
    }
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  }
}

// Two passes over the row (the second one hits L2): keeps the register footprint independent of D so
// that D = 4096 / 5120 / 7168 (Llama-3-8B / Vicuna-13B / Yi-34B) all run without spills.
template <typename T, int NCH, bool HAS_DW>
__global__ void __launch_bounds__(256) rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          int64_t rows, int D, const float* __restrict__ w,
                                                          const float* __restrict__ rstd_in,
                                                          T* __restrict__ dx, float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* sred = reinterpret_cast<float*>(dyn_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  float pw[HAS_DW ? NCH : 1][8];
#pragma unroll
  for (int c = 0; c < (HAS_DW ? NCH : 1); ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) pw[c][e] = 0.f;
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    const float rstd = rstd_in[row];
    const T* xr = x + row * (int64_t)D;
    const T* dyr = dy + row * (int64_t)D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float xv[8], dv[8], ww[8];
        Vec8<T>::load(xr + vi * 8, xv);
        Vec8<T>::load(dyr + vi * 8, dv);
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = xv[e] * rstd;
          if (HAS_DW) pw[HAS_DW ? c : 0][e] += dv[e] * xh;
          s += dv[e] * ww[e] * xh;
        }
      }
    }
    const float c2 = wave_sum(s) / (float)D;
    T* dxr = dx + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float xv[8], dv[8], ww[8], o[8];
        Vec8<T>::load(xr + vi * 8, xv);
        Vec8<T>::load(dyr + vi * 8, dv);
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (dv[e] * ww[e] - xv[e] * rstd * c2) * rstd;
        Vec8<T>::store(dxr + vi * 8, o);
      }
    }
  }
  if (HAS_DW) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sred[wave * D + vi * 8 + e] = pw[HAS_DW ? c : 0][e];
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256) {
      const float t = sred[col] + sred[D + col] + sred[2 * D + col] + sred[3 * D + col];
      atomicAdd(dw + col, t);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RoPE, rotate-half form.  The cos/sin table depends only on position_ids, so it is built once per
// forward (rope_table_kernel) and shared by every decoder layer's q and k, forward and backward;
// rope_apply_kernel is then a pure 16-byte-vectorised streaming pass (in place).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rope_table_kernel(const int64_t* __restrict__ pos_ids, int64_t ntok,
                                                         int half, int Dh, float base,
                                                         float* __restrict__ cos_t, float* __restrict__ sin_t) {
  const int64_t n = ntok * half;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = idx / half;
    const int i = (int)(idx - tok * half);
    // inv_freq = 1 / base^(2i/Dh) in fp32, angle = pos * inv_freq (phi3/modeling_phi3.py:127-141)
    const float inv_freq = 1.0f / powf(base, (float)(2 * i) / (float)Dh);
    float sn, cs;
    sincosf((float)pos_ids[tok] * inv_freq, &sn, &cs);
    cos_t[idx] = cs;
    sin_t[idx] = sn;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) rope_apply_kernel(T* __restrict__ x, const float* __restrict__ cos_t,
                                                         const float* __restrict__ sin_t, int64_t ntok, int H,
                                                         int Dh, int64_t row_stride, int inverse) {
  const int half = Dh >> 1, gph = half >> 3;  // vec8 groups per half head
  const int items = H * gph;
  for (int64_t tok = blockIdx.x; tok < ntok; tok += gridDim.x) {
    T* xr = x + tok * row_stride;
    const float* ct = cos_t + tok * half;
    const float* st = sin_t + tok * half;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
      const int h = it / gph, gi = it - h * gph;
      float a[8], b[8], c[8], s[8];
      Vec8<T>::load(xr + h * Dh + gi * 8, a);
      Vec8<T>::load(xr + h * Dh + half + gi * 8, b);
      load8f(ct + gi * 8, c);
      load8f(st + gi * 8, s);
      float oa[8], ob[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sn = inverse ? -s[e] : s[e];
        oa[e] = a[e] * c[e] - b[e] * sn;
        ob[e] = b[e] * c[e] + a[e] * sn;
      }
      Vec8<T>::store(xr + h * Dh + gi * 8, oa);
      Vec8<T>::store(xr + h * Dh + half + gi * 8, ob);
    }
  }
}

// y = rmsnorm(x + res) * w, sum = x + res (the residual stream) in ONE pass: the decoder's "h = h + attn(...);
// mlp_in = post_attention_layernorm(h)" (HF LlamaDecoderLayer reached from cambrian_llama.py:157-166).
template <typename T, int NCH>
__global__ void __launch_bounds__(256) add_rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              int64_t rows, int D, const float* __restrict__ w, float eps,
                                                              T* __restrict__ sum, T* __restrict__ y,
                                                              float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    float v[NCH][8];
    const T* xr = x + row * (int64_t)D;
    const T* rr = res + row * (int64_t)D;
    T* sr = sum + row * (int64_t)D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float a[8];
        Vec8<T>::load(xr + vi * 8, v[c]);
        Vec8<T>::load(rr + vi * 8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = (float)(T)(v[c][e] + a[e]);  // the stream is stored in T: normalise what is stored
        Vec8<T>::store(sr + vi * 8, v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e] * v[c][e];
      }
    }
    const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
    T* yr = y + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float ww[8], o[8];
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ww[e] * (v[c][e] * rstd);
        Vec8<T>::store(yr + vi * 8, o);
      }
    }
    if (lane == 0) rstd_out[row] = rstd;
  }
}

// dx = rmsnorm_backward(dy; x, w, rstd) + dadd, single pass: x and dy of the row stay in registers in their storage
// type between the reduction and the update (frozen weight: no dw).
template <typename T, int NCH>
__global__ void __launch_bounds__(256) rmsnorm_bwd_add_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const T* __restrict__ dadd, int64_t rows, int D,
                                                              const float* __restrict__ w, const float* __restrict__ rstd_in,
                                                              T* __restrict__ dx) {
  typedef T vec_t __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D >> 3;
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    const float rstd = rstd_in[row];
    const T* xr = x + row * (int64_t)D;
    const T* dyr = dy + row * (int64_t)D;
    vec_t xs[NCH], ds[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        xs[c] = *reinterpret_cast<const vec_t*>(xr + vi * 8);
        ds[c] = *reinterpret_cast<const vec_t*>(dyr + vi * 8);
        float ww[8];
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)ds[c][e] * ww[e] * ((float)xs[c][e] * rstd);
      }
    }
    const float c2 = wave_sum(s) / (float)D;
    T* dxr = dx + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int vi = lane + c * 64;
      if (vi < nvec) {
        float ww[8], o[8];
        load8f(w + vi * 8, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ((float)ds[c][e] * ww[e] - (float)xs[c][e] * rstd * c2) * rstd;
        if (dadd) {
          float a[8];
          Vec8<T>::load(dadd + row * (int64_t)D + vi * 8, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += a[e];
        }
        Vec8<T>::store(dxr + vi * 8, o);
      }
    }
  }
}

int nch_for(int64_t D) {
  const int64_t need = (D / 8 + 63) / 64;
  if (need <= 2) return 2;
  if (need <= 4) return 4;
  if (need <= 8) return 8;
  if (need <= 16) return 16;
  return -1;
}

inline int row_grid(int64_t rows) {
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

#define DISPATCH_NCH(nch, ...)                \
  switch (nch) {                              \
    case 2: { constexpr int NCH = 2; __VA_ARGS__; break; }   \
    case 4: { constexpr int NCH = 4; __VA_ARGS__; break; }   \
    case 8: { constexpr int NCH = 8; __VA_ARGS__; break; }   \
    case 16: { constexpr int NCH = 16; __VA_ARGS__; break; } \
    default: return CMB_ERR_SHAPE;            \
  }

template <typename T>
int ln_fwd(const void* x, int64_t rows, int64_t D, int64_t ldx, const float* add, int side, int grid_r,
           const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd,
           hipStream_t s) {
  const int nch = nch_for(D);
  const int variant = cmb_knob(CMB_KNOB_LN_FWD);
  if (gamma && !add && variant != 0) {
    // the towers' affine LayerNorms: parameters in LDS, next row in flight (layernorm_fwd_lds_kernel)
    // ~4 rows per wave: enough for the prefetch to run ahead, few enough that small launches still fill the chip
    // (profiles/r04_lab.md: 2048 / 4096 / 8192 / 1024-workgroup caps per shape); knob values > 1 = an explicit cap
    int64_t blocks = (rows + 15) / 16;
    const int64_t cap = variant > 1 ? variant : 8192;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const size_t smem = (size_t)2 * D * sizeof(float);
    DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_fwd_lds_kernel<T, NCH>), dim3((unsigned)blocks), dim3(256), smem, s,
                                         (const T*)x, rows, (int)D, ldx, gamma, beta, eps, (T*)y, ldy, mean, rstd));
    CMB_CHECK_LAUNCH();
    return CMB_OK;
  }
  // (a grid capped at the resident-wave capacity, 2048 workgroups, so that a wave walks more rows per parameter load was
  // measured SLOWER: 152 vs 126 us on 147456 x 1024 — fewer waves per CU hide less of the one-row-at-a-time latency)
  DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_fwd_kernel<T, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                       (const T*)x, rows, (int)D, ldx, add, side, grid_r, gamma, beta, eps,
                                       (T*)y, ldy, mean, rstd));
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

template <typename T, typename TDx, bool ACCUM>
int ln_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, int64_t rows, int64_t D, const float* add,
           int side, int grid_r, const float* gamma, const float* mean, const float* rstd, void* dx,
           int64_t lddx, float* dgamma, float* dbeta, float* dadd, hipStream_t s) {
  const int nch = nch_for(D);
  if (nch > 8) return CMB_ERR_SHAPE;  // register budget: the backward supports D <= 4096 (ConvNeXt-XXL stage 4: 3072)
  const int npos = grid_r * grid_r;
  const int64_t nwin = rows / npos;
  int64_t blocks = (nwin + 63) / 64;  // ~16 rows per wave: amortises the end-of-block atomics
  // ... unless that leaves the chip mostly idle (the SVA query-side LayerNorms: 13 824 rows = 216 workgroups, 61 us = 0.16 of
  // the HBM peak, profiles/r03_hbm_kernels_table.md): then ~4 rows per wave
  // (CMB_KNOB_LN_BWD_ROWS rows per workgroup: every workgroup ends with one atomicAdd per column and parameter)
  if (blocks < 1024) blocks = (nwin + cmb_knob(CMB_KNOB_LN_BWD_ROWS) - 1) / cmb_knob(CMB_KNOB_LN_BWD_ROWS);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const size_t smem = (size_t)4 * D * sizeof(float);
  if (nch == 2)
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, TDx, 2, ACCUM>), dim3((unsigned)blocks, npos), dim3(256), smem, s,
                       (const T*)dy, lddy, (const T*)x, ldx, rows, (int)D, add, side, grid_r, gamma, mean, rstd,
                       (TDx*)dx, lddx, dgamma, dbeta, dadd);
  else if (nch == 4)
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, TDx, 4, ACCUM>), dim3((unsigned)blocks, npos), dim3(256), smem, s,
                       (const T*)dy, lddy, (const T*)x, ldx, rows, (int)D, add, side, grid_r, gamma, mean, rstd,
                       (TDx*)dx, lddx, dgamma, dbeta, dadd);
  else  // 2048 < D <= 4096: the per-lane partial sums spill into the AGPR half of the register file (one wave per SIMD)
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, TDx, 8, ACCUM>), dim3((unsigned)blocks, npos), dim3(256), smem, s,
                       (const T*)dy, lddy, (const T*)x, ldx, rows, (int)D, add, side, grid_r, gamma, mean, rstd,
                       (TDx*)dx, lddx, dgamma, dbeta, dadd);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace

extern "C" int cmb_layernorm_fwd(int dtype, const void* x, int64_t rows, int64_t D, int64_t ldx,
                                 const float* add, int32_t side, int32_t grid_r, const float* gamma,
                                 const float* beta, float eps, void* y, int64_t ldy, float* mean,
                                 float* rstd, void* stream) {
  if (!x || !y || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if ((gamma == nullptr) != (beta == nullptr)) return CMB_ERR_BAD_ARG;
  if (add && (side <= 0 || grid_r <= 0 || side % grid_r)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  if (!add) { side = 1; grid_r = 1; }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16)
    return ln_fwd<bf16_t>(x, rows, D, ldx, add, side, grid_r, gamma, beta, eps, y, ldy, mean, rstd, s);
  if (dtype == CMB_F32)
    return ln_fwd<float>(x, rows, D, ldx, add, side, grid_r, gamma, beta, eps, y, ldy, mean, rstd, s);
  return CMB_ERR_BAD_ARG;
}

extern "C" int cmb_layernorm_fwd_multi(const cmb_ln_fwd_multi_desc* d, void* stream) {
  if (!d || !d->x || d->rows < 0 || d->D <= 0 || (d->D & 7) || d->D > 1024 || (d->ldx & 7)) return CMB_ERR_BAD_ARG;
  if (d->layers <= 0 || d->layers > CMB_LN_MULTI_MAX) return CMB_ERR_BAD_ARG;
  if (d->dtype != CMB_BF16 && d->dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  int side = d->side, grid_r = d->grid_r;
  bool any_add = false;
  LnFwdMultiParams mp;
  mp.layers = d->layers;
  for (int l = 0; l < CMB_LN_MULTI_MAX; ++l) {
    const bool on = l < d->layers;
    if (on && (!d->y[l] || !d->mean[l] || !d->rstd[l])) return CMB_ERR_BAD_ARG;
    any_add = any_add || (on && d->add[l]);
    mp.add[l] = on ? d->add[l] : nullptr;
    mp.y[l] = on ? d->y[l] : nullptr;
    mp.mean[l] = on ? d->mean[l] : nullptr;
    mp.rstd[l] = on ? d->rstd[l] : nullptr;
  }
  if (any_add && (side <= 0 || grid_r <= 0 || side % grid_r)) return CMB_ERR_BAD_ARG;
  if (!any_add) { side = 1; grid_r = 1; }
  if (d->rows == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (d->rows + 15) / 16;   // ~4 rows per wave, each written `layers` times
  if (blocks > 8192) blocks = 8192;
  const int nch = nch_for(d->D);
  if (nch != 2) return CMB_ERR_SHAPE;
#define LN_FWD_MULTI(T_, NCH_)                                                                                                  \
  hipLaunchKernelGGL((layernorm_fwd_multi_kernel<T_, NCH_>), dim3((unsigned)blocks), dim3(256), 0, s, (const T_*)d->x, d->rows, \
                     (int)d->D, d->ldx, side, grid_r, d->eps, mp)
  if (d->dtype == CMB_BF16) LN_FWD_MULTI(bf16_t, 2);
  else LN_FWD_MULTI(float, 2);
#undef LN_FWD_MULTI
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_row_stats(int dtype, const void* x, int64_t rows, int64_t D, int64_t ldx, float eps, float* mean, float* rstd,
                             void* stream) {
  if (!x || !mean || !rstd || rows < 0 || D <= 0 || (D & 7) || D > 4096 || (ldx & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nch = nch_for(D);
  int64_t blocks = (rows + 15) / 16;   // ~4 rows per wave (as layernorm_fwd_lds_kernel's launch)
  if (blocks > 8192) blocks = 8192;
  if (dtype == CMB_BF16) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((row_stats_kernel<bf16_t, NCH>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x,
                                         rows, (int)D, ldx, eps, mean, rstd));
  } else if (dtype == CMB_F32) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((row_stats_kernel<float, NCH>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x,
                                         rows, (int)D, ldx, eps, mean, rstd));
  } else {
    return CMB_ERR_BAD_ARG;
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

// dx dtype: same as `dtype` when dx_accumulate == 0; fp32 when dx_accumulate != 0 (the cross-layer
// accumulator of the shared aux features is kept in fp32).
extern "C" int cmb_layernorm_bwd(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx,
                                 int64_t rows, int64_t D, const float* add, int32_t side, int32_t grid_r,
                                 const float* gamma, const float* mean, const float* rstd, void* dx,
                                 int64_t lddx, int32_t dx_accumulate, float* dgamma, float* dbeta,
                                 float* dadd, void* stream) {
  if (!dy || !x || !dx || !mean || !rstd || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (add && (side <= 0 || grid_r <= 0 || side % grid_r)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  if (!add) { side = 1; grid_r = 1; }
  if (add && rows % ((int64_t)side * side)) return CMB_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16) {
    if (dx_accumulate)
      return ln_bwd<bf16_t, float, true>(dy, lddy, x, ldx, rows, D, add, side, grid_r, gamma, mean, rstd, dx,
                                         lddx, dgamma, dbeta, dadd, s);
    return ln_bwd<bf16_t, bf16_t, false>(dy, lddy, x, ldx, rows, D, add, side, grid_r, gamma, mean, rstd, dx,
                                         lddx, dgamma, dbeta, dadd, s);
  }
  if (dtype == CMB_F32) {
    if (dx_accumulate)
      return ln_bwd<float, float, true>(dy, lddy, x, ldx, rows, D, add, side, grid_r, gamma, mean, rstd, dx,
                                        lddx, dgamma, dbeta, dadd, s);
    return ln_bwd<float, float, false>(dy, lddy, x, ldx, rows, D, add, side, grid_r, gamma, mean, rstd, dx,
                                       lddx, dgamma, dbeta, dadd, s);
  }
  return CMB_ERR_BAD_ARG;
}

extern "C" int cmb_layernorm_bwd_multi(const cmb_ln_multi_desc* d, void* stream) {
  if (!d || !d->x || !d->dx || d->rows < 0 || d->D <= 0 || (d->D & 7) || d->layers <= 0 || d->layers > CMB_LN_MULTI_MAX)
    return CMB_ERR_BAD_ARG;
  if (d->dtype != CMB_BF16 && d->dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  int side = d->side, grid_r = d->grid_r;
  bool any_add = false;
  for (int l = 0; l < d->layers; ++l) {
    if (!d->dy[l] || !d->mean[l] || !d->rstd[l]) return CMB_ERR_BAD_ARG;
    any_add = any_add || d->add[l];
  }
  if (any_add && (side <= 0 || grid_r <= 0 || side % grid_r)) return CMB_ERR_BAD_ARG;
  if (!any_add) { side = 1; grid_r = 1; }
  if (d->rows == 0) return CMB_OK;
  if (any_add && d->rows % ((int64_t)side * side)) return CMB_ERR_SHAPE;
  const int nch = nch_for(d->D);
  if (nch != 2) return CMB_ERR_SHAPE;   // D <= 1024 (the SVA feature width): 7 layers x 16 partial sums per lane
  const size_t smem = (size_t)4 * d->D * sizeof(float);
  const int npos = grid_r * grid_r;
  const int64_t nwin = d->rows / npos;
  int64_t blocks = (nwin + 31) / 32;   // ~8 rows per wave: amortises the end-of-workgroup combine
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipStream_t s = (hipStream_t)stream;
  int kChunk = cmb_knob(CMB_KNOB_LN_MULTI_CHUNK);   // layers per launch (4 ... 7; <= 6: two waves per SIMD)
  if (kChunk < 4 || kChunk > 7) kChunk = 4;
  for (int l0 = 0; l0 < d->layers; l0 += kChunk) {
    LnMultiParams mp;
    mp.layers = d->layers - l0 < kChunk ? d->layers - l0 : kChunk;
    mp.dx_out = (l0 + kChunk >= d->layers) ? d->dx_out : nullptr;   // the call's last launch
    for (int l = 0; l < CMB_LN_MULTI_MAX; ++l) {
      const bool on = l < mp.layers;
      mp.dy[l] = on ? d->dy[l0 + l] : nullptr;
      mp.add[l] = on ? d->add[l0 + l] : nullptr;
      mp.mean[l] = on ? d->mean[l0 + l] : nullptr;
      mp.rstd[l] = on ? d->rstd[l0 + l] : nullptr;
      mp.dadd[l] = on ? d->dadd[l0 + l] : nullptr;
    }
    const bool acc = d->accumulate || l0 > 0;
#define LN_MULTI_LAUNCH(T_, ACC_, LC_)                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd_multi_kernel<T_, 2, ACC_, LC_>), dim3((unsigned)blocks, npos), dim3(256), smem, s,  \
                     (const T_*)d->x, d->ldx, d->rows, (int)d->D, side, grid_r, mp, d->dx, d->lddx)
#define LN_MULTI_T(T_)                                                                      \
  do {                                                                                      \
    if (mp.layers > 6) { if (acc) LN_MULTI_LAUNCH(T_, true, 7); else LN_MULTI_LAUNCH(T_, false, 7); } \
    else if (mp.layers > 5) { if (acc) LN_MULTI_LAUNCH(T_, true, 6); else LN_MULTI_LAUNCH(T_, false, 6); } \
    else if (mp.layers > 4) { if (acc) LN_MULTI_LAUNCH(T_, true, 5); else LN_MULTI_LAUNCH(T_, false, 5); } \
    else if (mp.layers > 2) { if (acc) LN_MULTI_LAUNCH(T_, true, 4); else LN_MULTI_LAUNCH(T_, false, 4); } \
    else { if (acc) LN_MULTI_LAUNCH(T_, true, 2); else LN_MULTI_LAUNCH(T_, false, 2); }      \
  } while (0)
    if (d->dtype == CMB_BF16) LN_MULTI_T(bf16_t);
    else LN_MULTI_T(float);
#undef LN_MULTI_T
#undef LN_MULTI_LAUNCH
    CMB_CHECK_LAUNCH();
  }
  return CMB_OK;
}

extern "C" int cmb_rmsnorm_fwd(int dtype, const void* x, int64_t rows, int64_t D, const float* w, float eps,
                               void* y, float* rstd, void* stream) {
  if (!x || !y || !w || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nch = nch_for(D);
  if (dtype == CMB_BF16) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_fwd_kernel<bf16_t, NCH>), dim3(row_grid(rows)), dim3(256), 0,
                                         s, (const bf16_t*)x, rows, (int)D, w, eps, (bf16_t*)y, rstd));
  } else if (dtype == CMB_F32) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_fwd_kernel<float, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                         (const float*)x, rows, (int)D, w, eps, (float*)y, rstd));
  } else {
    return CMB_ERR_BAD_ARG;
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

template <typename T>
static int rms_bwd(const void* dy, const void* x, int64_t rows, int64_t D, const float* w, const float* rstd,
                   void* dx, float* dw, hipStream_t s) {
  const int nch = nch_for(D);
  int64_t blocks = dw ? (rows + 63) / 64 : (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const size_t smem = dw ? (size_t)4 * D * sizeof(float) : 0;
  if (dw) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, NCH, true>), dim3((unsigned)blocks), dim3(256),
                                         smem, s, (const T*)dy, (const T*)x, rows, (int)D, w, rstd, (T*)dx, dw));
  } else {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, NCH, false>), dim3((unsigned)blocks), dim3(256),
                                         smem, s, (const T*)dy, (const T*)x, rows, (int)D, w, rstd, (T*)dx, dw));
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_rmsnorm_bwd(int dtype, const void* dy, const void* x, int64_t rows, int64_t D,
                               const float* w, const float* rstd, void* dx, float* dw, void* stream) {
  if (!dy || !x || !w || !rstd || !dx || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16) return rms_bwd<bf16_t>(dy, x, rows, D, w, rstd, dx, dw, s);
  if (dtype == CMB_F32) return rms_bwd<float>(dy, x, rows, D, w, rstd, dx, dw, s);
  return CMB_ERR_BAD_ARG;
}

extern "C" int cmb_rope_table(const int64_t* position_ids, int64_t ntok, int64_t Dh, float base,
                              float* cos_t, float* sin_t, void* stream) {
  if (!position_ids || !cos_t || !sin_t || ntok < 0 || Dh <= 0 || (Dh & 1)) return CMB_ERR_BAD_ARG;
  if (ntok == 0) return CMB_OK;
  const int half = (int)(Dh / 2);
  int64_t blocks = (ntok * half + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, position_ids,
                     ntok, half, (int)Dh, base, cos_t, sin_t);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_rope_apply(int dtype, void* x, const float* cos_t, const float* sin_t, int64_t ntok,
                              int64_t H, int64_t Dh, int64_t row_stride, int32_t inverse, void* stream) {
  if (!x || !cos_t || !sin_t || ntok < 0 || H <= 0 || Dh <= 0 || (Dh & 15)) return CMB_ERR_BAD_ARG;
  if (ntok == 0) return CMB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t items = H * (Dh / 16);
  int threads = items >= 256 ? 256 : (int)((items + 63) / 64 * 64);
  int64_t blocks = ntok > 65535 ? 65535 : ntok;
  if (dtype == CMB_BF16)
    hipLaunchKernelGGL(rope_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(threads), 0, s, (bf16_t*)x, cos_t,
                       sin_t, ntok, (int)H, (int)Dh, row_stride, inverse);
  else if (dtype == CMB_F32)
    hipLaunchKernelGGL(rope_apply_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, s, (float*)x, cos_t,
                       sin_t, ntok, (int)H, (int)Dh, row_stride, inverse);
  else
    return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_add_rmsnorm_fwd(int dtype, const void* x, const void* res, int64_t rows, int64_t D, const float* w,
                                   float eps, void* sum, void* y, float* rstd, void* stream) {
  if (!x || !res || !w || !sum || !y || !rstd || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  const int nch = nch_for(D);
  if (nch < 0) return CMB_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<bf16_t, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                         (const bf16_t*)x, (const bf16_t*)res, rows, (int)D, w, eps, (bf16_t*)sum,
                                         (bf16_t*)y, rstd));
  } else if (dtype == CMB_F32) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<float, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                         (const float*)x, (const float*)res, rows, (int)D, w, eps, (float*)sum,
                                         (float*)y, rstd));
  } else {
    return CMB_ERR_BAD_ARG;
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_rmsnorm_bwd_add(int dtype, const void* dy, const void* x, const void* dadd, int64_t rows, int64_t D,
                                   const float* w, const float* rstd, void* dx, void* stream) {
  if (!dy || !x || !w || !rstd || !dx || rows < 0 || D <= 0 || (D & 7)) return CMB_ERR_BAD_ARG;
  if (rows == 0) return CMB_OK;
  const int nch = nch_for(D);
  if (nch < 0) return CMB_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_bwd_add_kernel<bf16_t, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                         (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)dadd, rows, (int)D, w, rstd,
                                         (bf16_t*)dx));
  } else if (dtype == CMB_F32) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((rmsnorm_bwd_add_kernel<float, NCH>), dim3(row_grid(rows)), dim3(256), 0, s,
                                         (const float*)dy, (const float*)x, (const float*)dadd, rows, (int)D, w, rstd,
                                         (float*)dx));
  } else {
    return CMB_ERR_BAD_ARG;
  }
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
