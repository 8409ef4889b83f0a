// gemm256.hip — the large-tile configuration of C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T), bf16, for gfx950.
//
// Geometry: block tile 256 x 256, K-step 64 (128 B per operand row), 8 waves as
// 2 (wr, along M) x 4 (wc, along N), one workgroup per CU, 128 KiB of LDS = 2 K-tile buffers x {A,B} x 2 halves of
// 128 rows x 128 B.  The wave (wr, wc) owns the four 64 x 32 output "quadrants"
//     rows  ra*128 + wr*64 + [0,64)   x   cols  cb*128 + wc*32 + [0,32)      (ra, cb in {0,1})
// so that EVERY wave reads EVERY half-tile, each half-tile in exactly one phase of the K-step:
//     phase q0: (ra0, cb0)  reads A-half0 (8 x ds_read_b128) + B-half0 (4)
//     phase q1: (ra0, cb1)  reads B-half1 (4)
//     phase q2: (ra1, cb1)  reads A-half1 (8)
//     phase q3: (ra1, cb0)  reads nothing (B-half0 fragments are still live)
// Each phase is { ds_read the fragments | issue ONE half-tile of LDS-DMA prefetch (2 x global_load_lds_dwordx4 per
// lane) -> s_barrier -> lgkmcnt(0) -> 8 x v_mfma_f32_32x32x16_bf16 under s_setprio 1 -> s_barrier }.  The two wave
// groups wr = 0 / wr = 1 (one wave of each per SIMD) run staggered by one barrier, so on every SIMD one wave is in
// its MFMA segment while its partner is in its load segment (cdna_hip_programming.md §5 "8-phase template").
//
// Prefetch schedule (tile t lives in buffer t & 1; "p" = global phase index 4t + q):
//     (t, q0): A-half1 of tile t+1        (t, q1): B-half0 of tile t+2 (issued AFTER the phase's first barrier)
//     (t, q2): A-half0 of tile t+2        (t, q3): B-half1 of tile t+2, then s_waitcnt vmcnt(6)
// RAW: the vmcnt(6) at (t, q3) leaves the 3 newest half-tiles (6 DMA instructions per lane) in flight, i.e. it
//      retires everything issued up to (t, q0) = all of tile t+1; the wait precedes the phase's first barrier and
//      tile t+1 is first read one phase later, which is safe for both wave groups (a reader departs a barrier that
//      the staging wave reached after its wait).
// WAR: a half-tile is re-staged >= 2 phases after the phase that read it (the reads are retired by lgkmcnt(0)
//      before the reading phase's second barrier): A-half0 read q0 -> restaged q2; B-half1 read q1 -> restaged q3;
//      A-half1 read q2 -> restaged q0 of the next tile.  B-half0 (read q0) is restaged in q1, one phase later, and
//      is therefore issued after q1's FIRST barrier, which every wave reaches only after retiring its q0 reads.
// Nothing else orders LDS-DMA against ds_read (MI355X_MICROARCH.md "Two waves per SIMD" item 7).
//
// LDS layout, swizzle and the LDS-DMA source permutation are those of gemm_layout.h (rows of 128 B, 16-B chunk c of
// row r at ((c ^ ((r >> 1) & 7)) << 4): conflict-free ds_read_b128 under the gfx950 lane grouping).
// Epilogue: each wave stages one quadrant at a time (64 x 32 fp32, row stride 36) in its PRIVATE 16 KiB slice of
// the LDS (no workgroup barrier), and leaves through gemm_epilogue8 as row-contiguous 16/32-byte stores.
#include <type_traits>
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

constexpr int kHalf = 16384;          // bytes of one half-tile (128 rows x 128 B)
constexpr int kBuf = 4 * kHalf;       // one K-tile buffer: A0 A1 B0 B1
constexpr int kSmem = 2 * kBuf;       // 128 KiB

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA issued from inline asm (saddr form: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset).
// Why not the builtin here: hipcc models global_load_lds as a FLAT operation that may also complete on the LGKM
// counter, and while one is pending it turns every counted lgkmcnt(N) it inserts in front of an MFMA into
// lgkmcnt(0) — which would serialise the in-wave ds_read prefetch of schedule 1.  M0 (the LDS destination base) is
// written, used and restored inside ONE statement (cdna_hip_programming.md §5.7); completion is tracked by hand
// (counted s_waitcnt vmcnt in the schedules below).
__device__ __forceinline__ void glds16_asm(const char* sbase, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}

#define CMB_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CMB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int SCHED>
__global__ void __launch_bounds__(512) gemm_nt_256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16x8_t frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nblk = p.tiles_m * p.tiles_n;
  const int id = gl_xcd_remap((int)blockIdx.x, nblk);
  int tile_m, tile_n;
  gl_group_tile(id, p.tiles_m, p.tiles_n, 4, &tile_m, &tile_n);  // 32 resident workgroups per XCD = 4 x 8 tiles
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int kz = blockIdx.y;
  const int kbeg = kz * p.k_per_split;
  const int kend = (kbeg + p.k_per_split < p.K) ? (kbeg + p.k_per_split) : p.K;
  const int nk = (kend > kbeg) ? (kend - kbeg) / 64 : 0;

  // ---- per-lane LDS-DMA sources: half-tile h, DMA group g = wave + 8*i covers rows g*8 .. g*8+7 ------------
  // Address = wave-uniform base (SGPR pair, advanced 128 B per K-tile) + 32-bit per-lane byte offset, so the DMA
  // uses the saddr form and costs 8 VGPRs, not 16 (the host guarantees the offsets fit: gemm.hip::fits_u32).
  const char* a_base = p.A + (row_off(p.a_map, (uint32_t)(m0 < p.M ? m0 : p.M - 1)) + kbeg) * 2;
  const char* b_base = p.B + ((int64_t)(n0 < p.N ? n0 : p.N - 1) * p.ldb + kbeg) * 2;
  uint32_t a_off[2][2], b_off[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int grp = wave + 8 * i;
      const int row = gl_dma_row(grp, lane), c = gl_dma_chunk(grp, lane);
      int gm = m0 + h * 128 + row;
      gm = gm < p.M ? gm : p.M - 1;
      a_off[h][i] = (uint32_t)((row_off(p.a_map, (uint32_t)gm) - row_off(p.a_map, (uint32_t)m0)) * 2 + c * 16);
      int gn = n0 + h * 128 + row;
      gn = gn < p.N ? gn : p.N - 1;
      b_off[h][i] = (uint32_t)((int64_t)(gn - n0) * p.ldb * 2 + c * 16);
    }
  char* const dma_dst = smem + wave * 1024;  // + buf*kBuf + half offset + i*8192
  const uint32_t dma_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)dma_dst));  // same, as an LDS byte address
  int a_k[2] = {0, 0}, b_k[2] = {0, 0};  // K-tiles already staged per half-tile (wave-uniform)

  // readfirstlane keeps the base provably wave-uniform AND opaque to loop strength reduction (which would
  // otherwise rebuild eight per-lane 64-bit induction pointers and spill them)
  // the zero-extension of the 32-bit offset must be selected in the SAME basic block as the DMA for the
  // "sgpr base + vgpr32 offset" addressing mode to match (instruction selection is per block): an empty asm
  // re-defines the value at the point of use so the extension cannot be hoisted out of the loop.
  auto pin32 = [](uint32_t v) -> uint32_t {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto uniform_ptr = [](const char* q) -> const char* {
    const uint64_t v = (uint64_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
  };
  auto stage_a = [&](int buf, int h) {
    const char* base = uniform_ptr(a_base + (int64_t)a_k[h] * 128);
    if constexpr (SCHED == 1) {
      glds16_asm(base, a_off[h][0], dma_lds + buf * kBuf + h * kHalf);
      glds16_asm(base, a_off[h][1], dma_lds + buf * kBuf + h * kHalf + 8192);
    } else {
      glds16(base + pin32(a_off[h][0]), dma_dst + buf * kBuf + h * kHalf);
      glds16(base + pin32(a_off[h][1]), dma_dst + buf * kBuf + h * kHalf + 8192);
    }
    ++a_k[h];
  };
  auto stage_b = [&](int buf, int h) {
    const char* base = uniform_ptr(b_base + (int64_t)b_k[h] * 128);
    if constexpr (SCHED == 1) {
      glds16_asm(base, b_off[h][0], dma_lds + buf * kBuf + 2 * kHalf + h * kHalf);
      glds16_asm(base, b_off[h][1], dma_lds + buf * kBuf + 2 * kHalf + h * kHalf + 8192);
    } else {
      glds16(base + pin32(b_off[h][0]), dma_dst + buf * kBuf + 2 * kHalf + h * kHalf);
      glds16(base + pin32(b_off[h][1]), dma_dst + buf * kBuf + 2 * kHalf + h * kHalf + 8192);
    }
    ++b_k[h];
  };

  // ---- fragment read addresses: row (lane & 31) of a 32-row sub-tile, chunk 2*ks + (lane >> 5), swizzled ----
  const int swz = (lane >> 1) & 7;  // == gl_swz(row) for every row this lane reads (row = 32*k + (lane & 31))
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ((gl_frag_chunk(ks, lane) ^ swz) << 4);
  const char* const a_rd = smem + (wr * 64 + (lane & 31)) * 128;              // + buf*kBuf + ra*kHalf + i*4096
  const char* const b_rd = smem + 2 * kHalf + (wc * 32 + (lane & 31)) * 128;  // + buf*kBuf + cb*kHalf

  f32x16_t acc[2][2][2];  // [ra][cb][i]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.0f;

  frag_t fa[2][4];      // A fragments of the current row-half: [i][ks]
  frag_t fb[2][4];      // B fragments: [cb][ks]

  auto read_a = [&](int buf, int ra) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fa[i][ks] = *reinterpret_cast<const frag_t*>(a_rd + buf * kBuf + ra * kHalf + i * 4096 + koff[ks]);
  };
  auto read_b = [&](int buf, int cb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      fb[cb][ks] = *reinterpret_cast<const frag_t*>(b_rd + buf * kBuf + cb * kHalf + koff[ks]);
  };
  auto mfma_quadrant = [&](int ra, int cb) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)  // swapped operands: accumulator rows walk n, columns walk m
        acc[ra][cb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cb][ks], fa[i][ks], acc[ra][cb][i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // one K-tile = 4 phases.  BUF is a compile-time constant so every LDS offset folds into an immediate.
  auto ktile = [&](auto buf_c, int t) {
    constexpr int BUF = decltype(buf_c)::value;
    const bool more1 = (t + 1 < nk), more2 = (t + 2 < nk);
    // ---- q0: (ra0, cb0)
    read_b(BUF, 0);
    read_a(BUF, 0);
    if (more1) stage_a(BUF ^ 1, 1);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CMB_SCHED_FENCE();
    mfma_quadrant(0, 0);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    // ---- q1: (ra0, cb1)
    read_b(BUF, 1);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more2) stage_b(BUF, 0);  // B-half0 was read one phase ago: only legal after this phase's first barrier
    CMB_SCHED_FENCE();
    mfma_quadrant(0, 1);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    // ---- q2: (ra1, cb1)
    read_a(BUF, 1);
    if (more2) stage_a(BUF, 0);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CMB_SCHED_FENCE();
    mfma_quadrant(1, 1);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    // ---- q3: (ra1, cb0)
    if (more2) {
      stage_b(BUF, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // retires all of tile t+1
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CMB_SCHED_FENCE();
    CMB_BARRIER();
    CMB_SCHED_FENCE();
    mfma_quadrant(1, 0);
    CMB_SCHED_FENCE();
    CMB_BARRIER();
  };

  if constexpr (SCHED == 0) {
    if (nk > 0) {
      // prologue: all of tile 0, and B0 / A0 / B1 of tile 1 (its A1 goes out in phase (0, q0))
      stage_a(0, 0);
      stage_a(0, 1);
      stage_b(0, 0);
      stage_b(0, 1);
      if (nk > 1) {
        stage_b(1, 0);
        stage_a(1, 0);
        stage_b(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      CMB_SCHED_FENCE();
      CMB_BARRIER();
      if (wr == 1) CMB_BARRIER();  // stagger the two wave groups by one barrier
      int t = 0;
  #pragma unroll 1
      for (; t + 1 < nk; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
      }
      if (t < nk) ktile(std::integral_constant<int, 0>{}, t);
      if (wr == 0) CMB_BARRIER();  // re-align: every wave has now executed the same number of barriers
    }
  } else {
    // ---- SCHED 1: in-wave software pipeline, ONE workgroup barrier per K-tile ---------------------------------
    // Each wave prefetches the fragments of the next quadrant with ds_reads issued ahead of the current
    // quadrant's 8 MFMAs (the LDS reads complete in the MFMAs' shadow), so no partner wave is needed to hide
    // them and the two waves of a SIMD simply interleave their MFMA streams.  LDS hand-off per K-tile t
    // (buffer t & 1), placed between quadrants q1 and q2:
    //     vmcnt(0)   tile t+1's DMA (issued one K-tile ago, after the previous hand-off) has landed
    //     lgkmcnt(0) this wave's last reads of tile t (issued in q0/q1) have returned
    //     s_barrier  => every wave may now read tile t+1 and nobody reads tile t any more
    //     issue the DMA of tile t+2 into buffer t & 1
    // Register sets: A0(t) in fa, A1(t) in fa1, B in fbx[parity]: B0(t) sits in set t & 1 for the whole tile,
    // B1(t) in the other set during q1..q2, which then receives B0(t+1) during q3.
    frag_t fa1[2][4];
    auto read_a1_half = [&](int buf, int i) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fa1[i][ks] = *reinterpret_cast<const frag_t*>(a_rd + buf * kBuf + kHalf + i * 4096 + koff[ks]);
    };
    auto read_b_set = [&](int set, int buf, int cb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fb[set][ks] = *reinterpret_cast<const frag_t*>(b_rd + buf * kBuf + cb * kHalf + koff[ks]);
    };
    auto mfma8 = [&](f32x16_t (&c)[2], frag_t (&a)[2][4], frag_t (&b)[4]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks], a[i][ks], c[i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // The loop body is ROTATED to start right after a hand-off ( = back half of tile t, front half of tile t+1),
    // so that no LDS read is outstanding across the back-edge: the compiler's wait-count scoreboard is merged
    // conservatively at loop headers and would otherwise turn the counted lgkmcnt(N) in front of each MFMA
    // cluster into lgkmcnt(0), serialising the prefetch it is meant to cover.
    auto front = [&](auto buf_c, int t) {  // q0, q1 and the hand-off of tile t
      constexpr int BUF = decltype(buf_c)::value;
      // q0: (ra0, cb0); prefetch B1(t) and the first 32 rows of A1(t)
      read_b_set(BUF ^ 1, BUF, 1);
      read_a1_half(BUF, 0);
      mfma8(acc[0][0], fa, fb[BUF]);
      // q1: (ra0, cb1); prefetch the other 32 rows of A1(t)
      read_a1_half(BUF, 1);
      mfma8(acc[0][1], fa, fb[BUF ^ 1]);
      // hand-off
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), visible to the compiler's own scoreboard
      CMB_SCHED_FENCE();
      CMB_BARRIER();
      CMB_SCHED_FENCE();
      if (t + 2 < nk) {
        stage_a(BUF, 0);
        stage_a(BUF, 1);
        stage_b(BUF, 0);
        stage_b(BUF, 1);
      }
    };
    auto back = [&](auto buf_c) {  // q2, q3 of tile t; prefetch A0 / B0 of tile t+1 (stale, unused LDS past the end)
      constexpr int BUF = decltype(buf_c)::value;
      read_a(BUF ^ 1, 0);
      mfma8(acc[1][1], fa1, fb[BUF ^ 1]);
      read_b_set(BUF ^ 1, BUF ^ 1, 0);  // into the set B1(t) just vacated
      mfma8(acc[1][0], fa1, fb[BUF]);
    };
    if (nk > 0) {
      typedef std::integral_constant<int, 0> B0;
      typedef std::integral_constant<int, 1> B1;
      stage_a(0, 0);
      stage_a(0, 1);
      stage_b(0, 0);
      stage_b(0, 1);
      if (nk > 1) {
        stage_a(1, 0);
        stage_a(1, 1);
        stage_b(1, 0);
        stage_b(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 landed, tile 1 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      CMB_SCHED_FENCE();
      CMB_BARRIER();
      CMB_SCHED_FENCE();
      read_a(0, 0);
      read_b_set(0, 0, 0);
      front(B0{}, 0);
      int t = 0;
#pragma unroll 1
      for (; t + 2 < nk; t += 2) {
        back(B0{});
        front(B1{}, t + 1);
        back(B1{});
        front(B0{}, t + 2);
      }
      back(B0{});
      if (t + 1 < nk) {
        front(B1{}, t + 1);
        back(B1{});
      }
      CMB_BARRIER();  // all LDS reads of the main loop are done before the slices are reused by the epilogue
    }
  }

  // ---- epilogue: quadrant -> private LDS slice (fp32, row stride 36) -> row-contiguous global stores -----------
  constexpr int CS = 36;
  float* cs = reinterpret_cast<float*>(smem + wave * 16384);
  // The per-element epilogue code exists ONCE (runtime loops): fully unrolled it is ~100 KB of straight-line
  // code that every wave streams through the instruction cache exactly once per tile.
  auto put = [&](auto ra_c, auto cb_c) {
    constexpr int ra = decltype(ra_c)::value, cb = decltype(cb_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4_t v;
        v[0] = acc[ra][cb][i][4 * q + 0];
        v[1] = acc[ra][cb][i][4 * q + 1];
        v[2] = acc[ra][cb][i][4 * q + 2];
        v[3] = acc[ra][cb][i][4 * q + 3];
        *reinterpret_cast<f32x4_t*>(cs + (i * 32 + gl_acc_m(lane)) * CS + gl_acc_n(4 * q, lane)) = v;
      }
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  const int act_code = p.slabs ? CMB_ACT_NONE : p.act;
#pragma unroll 1
  for (int qd = 0; qd < 4; ++qd) {
    const int ra = qd >> 1, cb = qd & 1;
    if (qd == 0) put(I0{}, I0{});
    else if (qd == 1) put(I0{}, I1{});
    else if (qd == 2) put(I1{}, I0{});
    else put(I1{}, I1{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the activation switch is taken once per quadrant, not once per element (the accumulators are not captured here)
    dispatch_act(act_code, [&p, cs, lane, m0, n0, ra, cb, wr, wc, kz](auto act_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const int item = it * 64 + lane;
        const int row = item >> 2, c8 = item & 3;
        const int gm = m0 + ra * 128 + wr * 64 + row;
        const int gn = n0 + cb * 128 + wc * 32 + c8 * 8;
        float v[8];
        {
          const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8);
          const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        }
        if (gm < p.M && gn < p.N) gemm_epilogue8<bf16_t, ACT>(p, kz, gm, gn, v);
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is rewritten by the next quadrant
  }
}

}  // namespace

template <int SCHED>
static int launch256(GemmParams& p, int splits, hipStream_t s) {
  static CmbAttrOnce attr_once;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<SCHED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits);
  hipLaunchKernelGGL(gemm_nt_256_kernel<SCHED>, grid, dim3(512), kSmem, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

int launch_gemm256_bf16(GemmParams& p, int splits, int sched, hipStream_t s) {
  return sched == 0 ? launch256<0>(p, splits, s) : launch256<1>(p, splits, s);
}

}  // namespace cmb_gemm_detail
