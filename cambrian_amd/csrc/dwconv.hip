// dwconv.hip — depthwise 7x7 convolution, padding 3, channels-last, for the ConvNeXt-XXL tower
// (timm ConvNeXtBlock.conv_dw reached from clip_convnext_encoder.py:133-136).  Forward only (frozen).
//
// Roofline class: HBM/L2 + fp32 VALU (98 flop per output element, no reuse across channels, so no
// MFMA).  Layout NHWC: a lane owns 8 consecutive channels (one 16-byte bf16 vector), consecutive lanes
// own consecutive channel groups, so every load is a coalesced run along C.  Each thread produces a
// strip of XT outputs along x and reuses the XT+6 input vectors of a row for all 7 horizontal taps:
// (XT+6)*7 / XT = 17.5 vector loads per output instead of 49.
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
