// dwconv.hip — depthwise 7x7 convolution, padding 3, channels-last, for the ConvNeXt-XXL tower
// (timm ConvNeXtBlock.conv_dw reached from clip_convnext_encoder.py:133-136).  Forward only (frozen).
//
// Roofline class: HBM/L2 + fp32 VALU (98 flop per output element, no reuse across channels, so no
// MFMA).  Layout NHWC: a lane owns 8 consecutive channels (one 16-byte bf16 vector), consecutive lanes
// own consecutive channel groups, so every load is a coalesced run along C.  Each thread produces a
// strip of XT outputs along x and reuses the XT+6 input vectors of a row for all 7 horizontal taps:
// (XT+6)*7 / XT = 17.5 vector loads per output instead of 49.
#include <type_traits>
#include "common.h"

namespace {

template <typename T, int XT>
__global__ void __launch_bounds__(256) dwconv7x7_kernel(const T* __restrict__ x, int64_t B, int H, int W, int C,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y) {
  const int CG = C >> 3;
  const int WX = (W + XT - 1) / XT;
  const int64_t total = B * H * (int64_t)WX * CG;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    int64_t rest = i / CG;
    const int xs = (int)(rest % WX);
    rest /= WX;
    const int yy = (int)(rest % H);
    const int64_t b = rest / H;
    const int x0 = xs * XT;
    float acc[XT][8];
    {
      float bb[8];
      load8f(bias + cg * 8, bb);
#pragma unroll
      for (int o = 0; o < XT; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = bb[e];
    }
    for (int dy = 0; dy < 7; ++dy) {
      const int iy = yy + dy - 3;
      if (iy < 0 || iy >= H) continue;
      const T* row = x + ((b * H + iy) * (int64_t)W) * C + cg * 8;
      float in[XT + 6][8];
#pragma unroll
      for (int k = 0; k < XT + 6; ++k) {
        const int ix = x0 + k - 3;
        if (ix >= 0 && ix < W) {
          Vec8<T>::load(row + (int64_t)ix * C, in[k]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) in[k][e] = 0.f;
        }
      }
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) {
        float ww[8];
        load8f(w + (int64_t)(dy * 7 + dx) * C + cg * 8, ww);
#pragma unroll
        for (int o = 0; o < XT; ++o)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[o][e] += ww[e] * in[o + dx][e];
      }
    }
    T* orow = y + ((b * H + yy) * (int64_t)W) * C + cg * 8;
#pragma unroll
    for (int o = 0; o < XT; ++o)
      if (x0 + o < W) Vec8<T>::store(orow + (int64_t)(x0 + o) * C, acc[o]);
  }
}

// ---- LDS-tiled variant (C % 64 == 0): the production kernel ---------------------------------------------------
// One workgroup = 16 (x) x 8 (y) outputs x 64 channels.  The (16+6) x (8+6) input patch and the 49 x 64 fp32 tap
// weights are staged in LDS once; a thread owns 8 channels and a strip of 4 outputs along x and walks the 7 input
// rows of its strip: 10 input vectors + 7 weight vectors per row from LDS feed 4 x 7 x 8 FMAs.  The direct-from-L1
// kernel above re-fetches every input 17.5 times and every weight vector once per 4 outputs through the texture
// path and measured 0.67 TB/s on ConvNeXt-XXL stage 3; LDS serves the same reuse at 256 B/clk/CU.
// Patch layout [y][x][64 ch] with an x-stride of 64*sizeof(T) + 32 B: the two strips (x, x+4) that share a
// ds_read_b128 lane group land on different halves of the 256-byte bank row (conflict-free).
template <typename T>
__global__ void __launch_bounds__(256) dwconv7x7_lds_kernel(const T* __restrict__ x, int H, int W, int C,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ y, int tiles_x) {
  constexpr int TX = 16, TY = 8, CC = 64, PX = TX + 6, PY = TY + 6;
  constexpr int XS = CC * (int)sizeof(T) + 32;  // bytes per patch position
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  float* wt = reinterpret_cast<float*>(smem + PY * PX * XS);  // [49][64]
  const int tid = threadIdx.x;
  const int tx0 = (blockIdx.x % tiles_x) * TX, ty0 = (blockIdx.x / tiles_x) * TY;
  const int c0 = blockIdx.y * CC;
  const int64_t b = blockIdx.z;
  const T* xb = x + b * (int64_t)H * W * C;
  // stage the weights and the input patch (zero padded)
  for (int i = tid; i < 49 * 8; i += 256) {
    const int tap = i >> 3, cv = i & 7;
    float v[8];
    load8f(w + (int64_t)tap * C + c0 + cv * 8, v);
    Vec8<float>::store(wt + tap * CC + cv * 8, v);
  }
  {
    // all of this thread's patch loads are issued before the first LDS write (one HBM/L2 round trip, not ten)
    typedef T vec_t __attribute__((ext_vector_type(8)));
    constexpr int NIT = (PY * PX * 8 + 255) / 256;
    vec_t r[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      const int cv = i & 7, pos = i >> 3;
      const int py = pos / PX, px = pos - py * PX;
      const int iy = ty0 + py - 3, ix = tx0 + px - 3;
      const bool ok = (i < PY * PX * 8) && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int64_t off = ok ? (((int64_t)iy * W + ix) * C + c0 + cv * 8) : (int64_t)(c0 + cv * 8);
      vec_t v = *reinterpret_cast<const vec_t*>(xb + off);  // (clamped address: always a valid element of x)
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
      }
      r[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      if (i < PY * PX * 8) {
        const int cv = i & 7, pos = i >> 3;
        *reinterpret_cast<vec_t*>(reinterpret_cast<T*>(patch + pos * XS) + cv * 8) = r[k];
      }
    }
  }
  __syncthreads();
  const int cl = tid & 7, sp = tid >> 3;
  const int sx = (sp & 3) * 4, sy = sp >> 2;
  float acc[4][8];
  {
    float bb[8];
    load8f(bias + c0 + cl * 8, bb);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[o][e] = bb[e];
  }
#pragma unroll 1
  for (int dy = 0; dy < 7; ++dy) {
    float in[10][8];
    const char* prow = patch + ((sy + dy) * PX + sx) * XS;
#pragma unroll
    for (int k = 0; k < 10; ++k) Vec8<T>::load(reinterpret_cast<const T*>(prow + k * XS) + cl * 8, in[k]);
#pragma unroll
    for (int dx = 0; dx < 7; ++dx) {
      float ww[8];
      load8f(wt + (dy * 7 + dx) * CC + cl * 8, ww);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] += ww[e] * in[o + dx][e];
    }
  }
  const int oy = ty0 + sy;
  if (oy < H) {
    T* orow = y + ((b * H + oy) * (int64_t)W) * C + c0 + cl * 8;
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (tx0 + sx + o < W) Vec8<T>::store(orow + (int64_t)(tx0 + sx + o) * C, acc[o]);
  }
}

// ---- column-walking variant (C % 64 == 0): round 4 ---------------------------------------------------------------
// The LDS kernel above is VALU-bound and almost half of its vector instructions are not multiply-adds: per tap row a
// thread converts 10 bf16x8 input vectors to fp32 (80 shift / mask operations) for 112 v_pk_fma_f32, and reads 7 fp32 weight
// vectors from LDS (1.8-2.0 TB/s = 0.24 of the HBM peak, profiles/r03_hbm_kernels_table.md).  Here a lane owns ONE
// channel pair — the natural operand of v_pk_fma_f32 — and a strip of XT = 4 outputs along x, and WALKS DOWN the rows of
// its chunk:
//   * the 49 x 2 tap weights of its channel pair live in 98 registers for the whole walk (no LDS, no weight traffic);
//   * an input row is loaded (10 dwords per lane, a wave's 32 channel pairs = one whole 128-byte line per position) and
//     converted ONCE, then feeds the 7 output rows it belongs to: 20 conversions per 196 v_pk_fma_f32 instead of 80 per 112;
//   * the 7 output rows in flight are 7 accumulator slots (7 x 4 x 2 registers); output row o sits in slot o % 7, so with
//     the row loop unrolled by 7 every slot index is a compile-time constant; a row leaves when its 7th input row is in.
// No LDS, no barriers.  A wave = 32 channel pairs x 2 adjacent strips; a workgroup = 4 waves = 32 consecutive outputs of
// one row chunk (neighbouring strips re-read 6 of their 10 positions: L1 / L2 hits).  The accumulation order per output
// element (tap rows ascending, taps ascending, fp32 fma) is that of the kernels above: results are bit-identical.
template <typename T> struct Pair;   // two consecutive channels of one position -> two floats
template <> struct Pair<bf16_t> {
  typedef uint32_t raw_t;
  static __device__ __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
  static __device__ __forceinline__ void unpack(raw_t r, bool ok, float (&v)[2]) {
    r = ok ? r : 0u;
    v[0] = __uint_as_float(r << 16);
    v[1] = __uint_as_float(r & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[2]) {
    typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
    typedef float f32x2_v __attribute__((ext_vector_type(2)));
    const f32x2_v f = {v[0], v[1]};
    *reinterpret_cast<uint32_t*>(p) = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_v));
  }
};
template <> struct Pair<float> {
  typedef float2 raw_t;
  static __device__ __forceinline__ raw_t load_raw(const float* p) { return *reinterpret_cast<const float2*>(p); }
  static __device__ __forceinline__ void unpack(raw_t r, bool ok, float (&v)[2]) {
    v[0] = ok ? r.x : 0.f; v[1] = ok ? r.y : 0.f;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  }
};

template <typename T>
__global__ void __launch_bounds__(256) dwconv7x7_col_kernel(const T* __restrict__ x, int H, int W, int C,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ y, int chunk_rows, int n_chunks, int xblocks) {
  constexpr int XT = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blockIdx.x = ((b * n_chunks + chunk) * (C / 64) + cgroup) * xblocks + xblock;  32 outputs along x per workgroup
  int bid = blockIdx.x;
  const int xb = bid % xblocks; bid /= xblocks;
  const int cg = bid % (C >> 6); bid /= (C >> 6);
  const int chunk = bid % n_chunks;
  const int64_t b = bid / n_chunks;
  const int c = cg * 64 + 2 * (lane & 31);
  const int x0 = xb * 32 + wave * 8 + (lane >> 5) * XT;
  const int y0 = chunk * chunk_rows;
  const int rows_out = (H - y0 < chunk_rows) ? (H - y0) : chunk_rows;
  const T* xb_ = x + b * (int64_t)H * W * C + c;
  T* yb_ = y + b * (int64_t)H * W * C + c;

  float wt[49][2];
#pragma unroll
  for (int t = 0; t < 49; ++t) {
    const float2 r = *reinterpret_cast<const float2*>(w + (int64_t)t * C + c);
    wt[t][0] = r.x; wt[t][1] = r.y;
  }
  const float2 bs = *reinterpret_cast<const float2*>(bias + c);
  float acc[7][XT][2];
#pragma unroll
  for (int s_ = 0; s_ < 7; ++s_)
#pragma unroll
    for (int o = 0; o < XT; ++o) { acc[s_][o][0] = bs.x; acc[s_][o][1] = bs.y; }

  // per-lane column offsets of the strip's 10 input positions (clamped; `ok` masks the zero padding)
  int64_t coff[XT + 6];
  bool cok[XT + 6];
#pragma unroll
  for (int k = 0; k < XT + 6; ++k) {
    const int ix = x0 + k - 3;
    cok[k] = ix >= 0 && ix < W;
    coff[k] = (int64_t)(cok[k] ? ix : 0) * C;
  }
  const int n_in = rows_out + 6;   // input rows y0 - 3 .. y0 + rows_out + 2, j = 0 .. n_in - 1
  // The NEXT input row's 10 dwords are requested before the current row's 196 multiply-adds (a wave alone on its strip has
  // nothing else to cover the round trip to L2 / HBM with: 2 waves per SIMD at 217 registers).
  typename Pair<T>::raw_t cur[XT + 6], nxt[XT + 6];
  auto fetch = [&](int j, typename Pair<T>::raw_t (&dst)[XT + 6]) {
    const int iy = y0 - 3 + j;
    if (j < n_in && iy >= 0 && iy < H) {        // wave-uniform
      const T* row = xb_ + (int64_t)iy * W * C;
#pragma unroll
      for (int k = 0; k < XT + 6; ++k) dst[k] = Pair<T>::load_raw(row + coff[k]);
    }
  };
  fetch(0, cur);
  auto phase = [&](int j, auto ph_c) {
    constexpr int ph = decltype(ph_c)::value;   // j % 7
    const int iy = y0 - 3 + j;
    fetch(j + 1, nxt);
    if (iy >= 0 && iy < H) {                    // wave-uniform: rows of the zero padding contribute nothing
      float in[XT + 6][2];
#pragma unroll
      for (int k = 0; k < XT + 6; ++k) Pair<T>::unpack(cur[k], cok[k], in[k]);
#pragma unroll
      for (int dy = 0; dy < 7; ++dy) {
        const int o = j - dy;                   // the output row (chunk-local) this input row is tap row dy of
        if (o >= 0 && o < rows_out) {           // wave-uniform
          const int slot_c = (ph - dy + 7) % 7; // == o % 7, a constant per (ph, dy) once the loop is unrolled
#pragma unroll
          for (int dx = 0; dx < 7; ++dx)
#pragma unroll
            for (int q = 0; q < XT; ++q) {
              acc[slot_c][q][0] += wt[dy * 7 + dx][0] * in[q + dx][0];
              acc[slot_c][q][1] += wt[dy * 7 + dx][1] * in[q + dx][1];
            }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < XT + 6; ++k) cur[k] = nxt[k];
    const int done = j - 6;                     // tap row 6 of output row j - 6 was this input row: that row is complete
    constexpr int dslot = (ph + 1) % 7;         // (ph - 6 + 7) % 7
    if (done >= 0 && done < rows_out) {
      T* orow = yb_ + ((int64_t)(y0 + done) * W) * C;
#pragma unroll
      for (int q = 0; q < XT; ++q)
        if (x0 + q < W) Pair<T>::store(orow + (int64_t)(x0 + q) * C, acc[dslot][q]);
    }
#pragma unroll
    for (int q = 0; q < XT; ++q) { acc[dslot][q][0] = bs.x; acc[dslot][q][1] = bs.y; }
  };
  for (int j0 = 0; j0 < n_in; j0 += 7) {
    phase(j0 + 0, std::integral_constant<int, 0>{});
    if (j0 + 1 < n_in) phase(j0 + 1, std::integral_constant<int, 1>{});
    if (j0 + 2 < n_in) phase(j0 + 2, std::integral_constant<int, 2>{});
    if (j0 + 3 < n_in) phase(j0 + 3, std::integral_constant<int, 3>{});
    if (j0 + 4 < n_in) phase(j0 + 4, std::integral_constant<int, 4>{});
    if (j0 + 5 < n_in) phase(j0 + 5, std::integral_constant<int, 5>{});
    if (j0 + 6 < n_in) phase(j0 + 6, std::integral_constant<int, 6>{});
  }
}

template <typename T>
int launch_dwconv_col(const void* x, int64_t B, int H, int W, int C, const float* w, const float* bias, void* y,
                      int chunk_rows, hipStream_t s) {
  const int n_chunks = (H + chunk_rows - 1) / chunk_rows;
  const int xblocks = (W + 31) / 32;
  const int64_t grid = B * n_chunks * (C / 64) * (int64_t)xblocks;
  if (grid > 0x7fffffff) return CMB_ERR_SHAPE;
  hipLaunchKernelGGL(dwconv7x7_col_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, (const T*)x, H, W, C, w, bias, (T*)y,
                     chunk_rows, n_chunks, xblocks);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

template <typename T>
int launch_dwconv_lds(const void* x, int64_t B, int H, int W, int C, const float* w, const float* bias, void* y,
                      hipStream_t s) {
  constexpr int smem = (8 + 6) * (16 + 6) * (64 * (int)sizeof(T) + 32) + 49 * 64 * 4;
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv7x7_lds_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 7) / 8;
  dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(C / 64), (unsigned)B);
  hipLaunchKernelGGL(dwconv7x7_lds_kernel<T>, grid, dim3(256), smem, s, (const T*)x, H, W, C, w, bias, (T*)y, tiles_x);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the depthwise 7x7 (ConvNeXt towers that train):
//   dW[t = dy*7+dx][c] = sum_{b,y,x} dY[b,y,x,c] * X[b, y+dy-3, x+dx-3, c]      (zero outside the map)
// A workgroup owns a 64-channel slice and walks 8x8 output tiles (slot, slot + nslots, ...): the 14x14 input halo and
// the 8x8 dY tile are staged in LDS (channel-contiguous rows -> coalesced 128-byte global reads), thread (c, rg) keeps
// 49 fp32 accumulators over rows 2*rg, 2*rg+1 of every tile it sees.  The four row groups are summed through LDS at the
// end and the slot's [49, 64] partial goes to `partial[slot][49][C]`; the caller column-sums the slots (cmb_colsum).
// No atomics: bit-reproducible.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) dwconv7x7_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int B,
                                                              int H, int W, int C, float* __restrict__ partial) {
  __shared__ float sx[14 * 14 * 64];   // halo tile, [row][col][channel]
  __shared__ float sg[8 * 8 * 64];     // dY tile
  const int tid = threadIdx.x, c = tid & 63, rg = tid >> 6;
  const int c0 = blockIdx.y * 64;
  const int tx = (W + 7) / 8, ty = (H + 7) / 8;
  const int ntiles = B * ty * tx;
  float acc[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (ty * tx), r = tile - b * ty * tx;
    const int y0 = (r / tx) * 8, x0 = (r % tx) * 8;
    __syncthreads();  // previous tile consumed
    for (int i = tid; i < 14 * 14 * 64; i += 256) {
      const int ch = i & 63, pos = i >> 6, py = pos / 14, px = pos - py * 14;
      const int gy = y0 + py - 3, gx = x0 + px - 3;
      float v = 0.f;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = (float)x[(((int64_t)b * H + gy) * W + gx) * C + c0 + ch];
      sx[i] = v;
    }
    for (int i = tid; i < 8 * 8 * 64; i += 256) {
      const int ch = i & 63, pos = i >> 6, py = pos >> 3, px = pos & 7;
      const int gy = y0 + py, gx = x0 + px;
      float v = 0.f;
      if (gy < H && gx < W) v = (float)dy[(((int64_t)b * H + gy) * W + gx) * C + c0 + ch];
      sg[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
      const int py = 2 * rg + ry;
      for (int px = 0; px < 8; ++px) {
        const float g = sg[(py * 8 + px) * 64 + c];
#pragma unroll
        for (int dy_ = 0; dy_ < 7; ++dy_)
#pragma unroll
          for (int dx_ = 0; dx_ < 7; ++dx_) acc[dy_ * 7 + dx_] += g * sx[((py + dy_) * 14 + px + dx_) * 64 + c];
      }
    }
  }
  // sum the four row groups: [rg][49][64] through LDS (reuses the halo buffer: 4 * 49 * 64 floats <= 14 * 14 * 64)
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 49; ++t) sx[(rg * 49 + t) * 64 + c] = acc[t];
  __syncthreads();
  for (int i = tid; i < 49 * 64; i += 256) {
    const float v = sx[i] + sx[49 * 64 + i] + sx[2 * 49 * 64 + i] + sx[3 * 49 * 64 + i];
    const int t = i >> 6, ch = i & 63;
    partial[((int64_t)blockIdx.x * 49 + t) * C + c0 + ch] = v;
  }
}

}  // namespace

extern "C" int cmb_dwconv7x7_nhwc(int dtype, const void* x, int64_t B, int64_t H, int64_t W, int64_t C,
                                  const float* w, const float* bias, void* y, void* stream) {
  if (!x || !w || !bias || !y || B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return CMB_ERR_BAD_ARG;
  if (B == 0) return CMB_OK;
  constexpr int XT = 4;
  const int64_t total = B * H * ((W + XT - 1) / XT) * (C / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipStream_t s = (hipStream_t)stream;
  const int variant = cmb_knob(CMB_KNOB_DWCONV);
  if (variant != 0 && (C & 63) == 0 && cmb_aligned16(x) && cmb_aligned16(y) && cmb_aligned16(w) && cmb_aligned16(bias)) {
    // column-walking kernel; rows per chunk: fewer rows = more waves, more halo rows re-read (profiles/r04_lab.md)
    const int chunk = variant == 1 ? (H >= 128 ? 64 : 32) : variant;
    if (dtype == CMB_BF16) return launch_dwconv_col<bf16_t>(x, B, (int)H, (int)W, (int)C, w, bias, y, chunk, s);
    if (dtype == CMB_F32) return launch_dwconv_col<float>(x, B, (int)H, (int)W, (int)C, w, bias, y, chunk, s);
    return CMB_ERR_BAD_ARG;
  }
  if ((C & 63) == 0 && B <= 65535 && cmb_aligned16(x) && cmb_aligned16(y)) {
    if (dtype == CMB_BF16) return launch_dwconv_lds<bf16_t>(x, B, (int)H, (int)W, (int)C, w, bias, y, s);
    if (dtype == CMB_F32) return launch_dwconv_lds<float>(x, B, (int)H, (int)W, (int)C, w, bias, y, s);
    return CMB_ERR_BAD_ARG;
  }
  if (dtype == CMB_BF16)
    hipLaunchKernelGGL((dwconv7x7_kernel<bf16_t, XT>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, B,
                       (int)H, (int)W, (int)C, w, bias, (bf16_t*)y);
  else if (dtype == CMB_F32)
    hipLaunchKernelGGL((dwconv7x7_kernel<float, XT>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, B,
                       (int)H, (int)W, (int)C, w, bias, (float*)y);
  else
    return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

extern "C" int cmb_dwconv7x7_wgrad(int dtype, const void* x, const void* dy, int64_t B, int64_t H, int64_t W, int64_t C,
                                   float* partial, int32_t slots, void* stream) {
  if (!x || !dy || !partial || B < 0 || H <= 0 || W <= 0 || C <= 0 || slots <= 0) return CMB_ERR_BAD_ARG;
  if (C % 64 != 0) return CMB_ERR_SHAPE;
  if (B == 0) return CMB_OK;
  const dim3 grid((unsigned)slots, (unsigned)(C / 64));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CMB_BF16)
    hipLaunchKernelGGL(dwconv7x7_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (int)B,
                       (int)H, (int)W, (int)C, partial);
  else if (dtype == CMB_F32)
    hipLaunchKernelGGL(dwconv7x7_wgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, (int)B,
                       (int)H, (int)W, (int)C, partial);
  else
    return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
