"""Lab build (NOT product): timing variants of csrc/sva_absorbed.hip with one phase removed each, to locate where a query's
time goes.  Writes tools/abs_lab/build/v<N>.so (git-ignored); tools/abs_lab/run.py times them on the GPU."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SRC = open(os.path.join(ROOT, "cambrian_amd/csrc/sva_absorbed.hip")).read()
OUT = os.path.join(HERE, "build")
os.makedirs(OUT, exist_ok=True)

def sub(s, old, new, count=None):
    assert old in s, old[:60]
    return s.replace(old, new) if count is None else s.replace(old, new, count)

STORE_MIX = "    *reinterpret_cast<bf16x8_t*>(op + cg * 32) = cvt8_bf16(d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]);"
KEEP_MIX = "    if (d0[0] == 12345.678f) *reinterpret_cast<bf16x8_t*>(op + cg * 32) = cvt8_bf16(d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]);"
variants = {
    0: lambda s: s,
    1: lambda s: sub(s, STORE_MIX, KEEP_MIX),                                   # token mixes computed, not stored
    2: lambda s: sub(s, "__device__ __forceinline__ void token_mix(const char* xs, s16x4_t coef, bf16_t* orow, int i, int qd) {",
                     "__device__ __forceinline__ void token_mix(const char* xs, s16x4_t coef, bf16_t* orow, int i, int qd) {\n  if (coef[0] != 12345) return;"),
    3: lambda s: sub(s, "    acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, yv[s], acc[s & 3], 0, 0, 0);",
                     "    if (s == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, yv[s], acc[0], 0, 0, 0);"),  # one MFMA + read of the score product
    4: lambda s: sub(sub(s, "      for (int s = 0; s < 32; ++s) u[s] = *reinterpret_cast<const bf16x8_t*>(ur + s * 32);",
                         "      for (int s = 0; s < 32; ++s) u[s] = *reinterpret_cast<const bf16x8_t*>(ur + (s & 1) * 32);"),
                     "      for (int s = 0; s < 32; ++s) uv[s] = *reinterpret_cast<const bf16x8_t*>(up + s * 32);",
                     "      for (int s = 0; s < 32; ++s) uv[s] = *reinterpret_cast<const bf16x8_t*>(up + (s & 1) * 32);"),  # U: 2 loads instead of 32
    5: lambda s: sub(s, "  for (int j = 0; j < na; ++j) {\n    const bf16_t* xr = xb + token_row", "  for (int j = 0; j < (na > 99 ? na : 0); ++j) {\n    const bf16_t* xr = xb + token_row"),  # no window DMA
    6: lambda s: sub(s, "      for (int half = 0; half < 2; ++half) {", "      for (int half = 0; half < (na > 99 ? 2 : 0); ++half) {"),  # bwd: no dX pass
    7: lambda s: sub(s, "          if (i < na)\n            *reinterpret_cast<bf16x8_t*>(dxrow", "          if (i < na && dd[0][0] == 12345.678f)\n            *reinterpret_cast<bf16x8_t*>(dxrow"),  # dX computed, not stored
    8: lambda s: sub(s, "__global__ void __launch_bounds__(64) sva_abs_fwd_kernel", "__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) sva_abs_fwd_kernel"),  # fwd <= 256 registers: 5 waves per CU
}
only = [int(a) for a in sys.argv[1:]] or sorted(variants)
for n in only:
    src = variants[n](SRC)
    f = os.path.join(OUT, f"v{n}.hip")
    open(f, "w").write(src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "cambrian_amd/csrc"),
           "-o", os.path.join(OUT, f"v{n}.so"), f]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(n, "ok" if r.returncode == 0 else r.stderr[-2000:])
    os.remove(f)
