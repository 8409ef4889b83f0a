// Issue-rate probe for the f32 vector instructions the HBM / VALU-bound kernels lean on (gfx950): how many shader cycles a
// wave64 pays per v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, with 1 and 2 waves per SIMD.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip && ./valu_probe
// Prints one JSON line per (instruction, waves per SIMD).  Used for the roofline of dwconv7x7_col_kernel and of the GEMM
// epilogues: the "157 TFLOP/s packed-fp32 peak" is only a peak if v_pk_fma_f32 issues in 4 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) probe(float* out, long long* cycles, int iters) {
  f32x2 a[8];
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (f32x2){1.0f + threadIdx.x * 1e-6f + i, 2.0f + i}; s[i] = 0.5f + i + threadIdx.x * 1e-6f; }
  const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
  const float ms = 1.0000001f, cs = 1e-9f;
  __builtin_amdgcn_s_barrier();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(ms), "v"(cs));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
        if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(s[i]) : "v"(ms));
        if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(ms));
      }
    }
  }
  const long long t1 = clock64();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i][0] + a[i][1] + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 131072;   // ~10 ms per launch: clocks settle
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);   // warm-up at full length
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
  const double n_inst = (double)iters * 32;               // per wave
  const int waves_per_simd = threads / 256;
  // clock64() on gfx9 = s_memtime: a constant 100 MHz counter on some parts, the shader clock on others; report both views
  const double lanes_ops = n_inst * 64.0 * (threads / 64) * blocks;   // lane-instructions executed chip-wide
  printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"counter_ticks_per_inst_per_wave\": %.3f, \"ns_per_inst_per_simd\": %.4f, "
         "\"kernel_ms\": %.4f, \"T_lane_inst_per_s\": %.2f}\n", name, waves_per_simd, avg / n_inst,
         ms * 1e6 / (n_inst * waves_per_simd), ms, lanes_ops / (ms * 1e-3) * 1e-12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("v_fma_f32", threads);
    run<5>("v_fmaak_f32", threads);
    run<6>("v_mul_f32", threads);
    run<1>("v_pk_fma_f32", threads);
    run<2>("v_pk_mul_f32", threads);
    run<4>("v_pk_add_f32", threads);
    run<3>("v_exp_f32", threads);
  }
  return 0;
}
