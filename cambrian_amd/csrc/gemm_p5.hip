// gemm_p5.hip — persistent large-tile configuration of C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T), bf16, gfx950:
// gemm_nt_p5_kernel, the default 256 x 256 kernel (gemm.hip::p5_default says where).
//
// Why a second 256 x 256 kernel beside the 8-wave / 8-phase kernel of gemm256.hip: that kernel keeps the matrix pipe 60 %
// busy — a wave owns 128 x 64 of the tile, so every K-step moves 24 KB of fragments per wave through the LDS, each phase
// is fenced by two workgroup barriers, and prologue + epilogue of every tile run with the pipe idle (one workgroup per CU).
// Here:
//   * 4 waves (one per SIMD, 512 registers each), a wave owns 128 x 128 = 4 x 4 tiles of 32 x 32: 256 accumulator
//     registers, 16 KB of fragments per wave per 64-deep K-step (2/3 of the LDS traffic per MFMA), the other waves'
//     instructions never contend for the SIMD;
//   * the workgroup is PERSISTENT (grid = number of CUs): the LDS-DMA cursor runs ahead across tile boundaries, so the
//     next item's first tiles are in flight / landed while the current item's epilogue runs, and the epilogue leaves
//     straight from the accumulators (gemm_p5_epilogue.inc).
// One wave per SIMD means every cycle an LDS / vector-memory instruction spends waiting to be ACCEPTED is a cycle the
// matrix pipe idles; the issue-slot schedule below is the result of the round-2 ablations (profiles/r02_gemm_lab.md).
// The designs measured on the way — a ring of five 32-deep stages (gemm_nt_p4_kernel), a 256 x 128 kernel with two
// workgroups per CU, and the timing-only ablation variants of this kernel — are not product code: they live in the
// history of this file up to commit ab4c6d4 (cambrian_amd/csrc/gemm_p4.hip) together with tools/gemm_lab.py.
//
// LDS tile layout (gemm_layout.h): rows of 128 B = 8 chunks of 16 B; chunk c of row r lives at r*128 + ((c ^ gl_swz(r)) << 4);
// ds_read_b128 fragment reads are conflict-free and the swizzle is applied to the DMA's per-lane SOURCE address.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

#define P4_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)

// 16 consecutive floats at a wave-uniform address (scalar loads: they count on lgkmcnt, so unlike a vector load they
// do not have to wait behind the tile's stores and LDS-DMA on vmcnt) -> this lane's 8: lanes 0-31 take floats 0..7,
// lanes 32-63 floats 8..15 (the column halves of one 16-column group after the permlane32 swap).
typedef float f32x16s_t __attribute__((ext_vector_type(16)));
// The staged 64 rows x 128 bytes of a wave leave as 8 stores of 8 rows (a lane: 16 bytes of row 8 k + (lane >> 3)).
// When every row of the wave's block is in range (wave-uniform `all`: every row tile but the last) the loop has no
// per-store row test: four LDS reads in flight, then their four stores, twice — the guarded form (each read behind its own
// compare / exec branch / s_waitcnt lgkmcnt(0), the row limit and eight precomputed row offsets reloaded from spilled
// scalars) cost ~110 issue slots and eight exposed LDS round trips per flush, four flushes per item (round 4, found by
// reading the assembly: profiles/r04_lab.md).  C is written once and read by a later kernel: streaming stores (+1.5 % on the
// path's large shapes, +7 % on the GELU one, round 2).
__device__ __forceinline__ void p4_flush8(const char* stage_r0, const char* stage_r1, bf16_t* cptr, int64_t step, bool all,
                                          int rows_left) {
  if (all) {
    bf16_t* q = cptr;
#pragma unroll
    for (int kb = 0; kb < 8; kb += 4) {
      bf16x8_t o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = *reinterpret_cast<const bf16x8_t*>((((kb + k) & 1) ? stage_r1 : stage_r0) + (kb + k) * 1024);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_nontemporal_store(o[k], reinterpret_cast<bf16x8_t*>(q));
        q += step;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bf16x8_t o8 = *reinterpret_cast<const bf16x8_t*>(((k & 1) ? stage_r1 : stage_r0) + k * 1024);
      if (8 * k < rows_left) __builtin_nontemporal_store(o8, reinterpret_cast<bf16x8_t*>(cptr + (int64_t)k * step));
    }
  }
}

__device__ __forceinline__ void p4_colvec(const float* base, bool upper, float (&v)[8]) {
  f32x16s_t o;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(o) : "s"(base) : "memory");
  uint32_t m = upper ? 0xffffffffu : 0u;   // lane-half select as a bit merge: v_mov + v_bfi_b32 per value (see p4_colvec4)
  asm volatile("" : "+v"(m));
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = __uint_as_float((__float_as_uint(o[e]) & ~m) | (__float_as_uint(o[8 + e]) & m));
}

// Four consecutive 16-float groups (one column half of a wave: 64 columns) with ONE wait: four back-to-back scalar loads
// expose one trip to the scalar cache / L2 instead of four (round 4: a bias cost the kernel ~7 us per item — 24 KB of bias
// across 256 CUs does not live in the 16 KB scalar caches, and every p4_colvec waited for its own load: eight exposed round
// trips per item; DINOv2 proj 78.9 -> 92.9 us with a bias, profiles/r04_lab.md).
__device__ __forceinline__ void p4_colvec4(const float* base, bool upper, float (&v)[4][8]) {
#ifndef CMB_P5_COLVEC_ONE_WAIT
  // two batches of two groups: 32 scalar registers in flight instead of 64 — with 64 the allocator spilled most of the
  // kernel's loop-invariant scalars to vector lanes around every half tile and reloaded them (v_readlane_b32) wherever the
  // epilogue needed one: ~250 reloads per item, every one an issue slot of the single wave on its SIMD
  auto pair = [&](int off, float (&d0)[8], float (&d1)[8]) {
    f32x16s_t o0, o1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(o0), "=&s"(o1)
                 : "s"(base + off)
                 : "memory");
    // lane-half select as a bit merge (v_bfi_b32: one scalar operand + a v_mov for the other = 2 vector instructions per
    // value, exact); `upper ? hi : lo` was v_mov + v_mov + v_cndmask + an s_mov per value
    uint32_t m = upper ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));   // opaque: hipcc otherwise sees through the mask and rebuilds the three-instruction select
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      d0[e] = __uint_as_float((__float_as_uint(o0[e]) & ~m) | (__float_as_uint(o0[8 + e]) & m));
      d1[e] = __uint_as_float((__float_as_uint(o1[e]) & ~m) | (__float_as_uint(o1[8 + e]) & m));
    }
  };
  pair(0, v[0], v[1]);
  pair(32, v[2], v[3]);
#else
  f32x16s_t o0, o1, o2, o3;
  asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\t"
               "s_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(o0), "=&s"(o1), "=&s"(o2), "=&s"(o3)
               : "s"(base)
               : "memory");
  auto pick = [&](const f32x16s_t& o, float (&d)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float lo = o[e], hi = o[8 + e];
      asm volatile("" : "+s"(lo), "+s"(hi));
      d[e] = upper ? hi : lo;
    }
  };
  pick(o0, v[0]);
  pick(o1, v[1]);
  pick(o2, v[2]);
  pick(o3, v[3]);
#endif
}

__device__ __forceinline__ const char* p4_uniform_ptr(const char* q) {
  const uint64_t v = (uint64_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}

// work item -> output tile / K slice.  Items are numbered tile-fastest (the 2-D grid order of the other kernels).
struct P4Item {
  int m0, n0, kz, kbeg, nk;  // nk = number of 32-deep stages
};
__device__ __forceinline__ P4Item p4_item(const GemmParams& p, int item, int ntiles) {
  P4Item it;
  const int kz = item / ntiles, lin = item - kz * ntiles;
  const int id = gl_xcd_remap(lin, ntiles);
  int tile_m, tile_n;
  gl_group_tile(id, p.tiles_m, p.tiles_n, 4, &tile_m, &tile_n);
  it.m0 = tile_m * 256;
  it.n0 = tile_n * 256;
  it.kz = kz;
  it.kbeg = kz * p.k_per_split;
  const int kend = (it.kbeg + p.k_per_split < p.K) ? (it.kbeg + p.k_per_split) : p.K;
  it.nk = (kend - it.kbeg) / 32;
  return it;
}

// order in which a phase reads the NEXT phase's fragments (0..3 = A row block i, 4..7 = B column block j): the MFMA
// order is (i, j) = (t >> 2, t & 3), so B0, A0 are needed first and A3 only by the 13th MFMA
__host__ __device__ constexpr int p4_read_order(int n) {
  return n == 0 ? 4 : n == 1 ? 0 : n == 2 ? 5 : n == 3 ? 6 : n == 4 ? 7 : n - 4;
}

// =====================================================================================================================
// gemm_nt_p5_kernel — the 4-wave persistent kernel with 64-deep K tiles, the WHOLE tile's fragments held in registers
// and a two-buffer LDS (2 x 64 KiB).  Differences from gemm_nt_p4_kernel and why (profiles/r02_gemm_lab.md):
//   * a stage row is 128 B = one full cache line per tile row (the 32-deep stages of the ring fetch every line of A
//     and B as two half-lines, one stage apart: twice the requests to the vector cache / L2 for the same bytes);
//   * the fragments of all four 16-deep sub-steps live in 128 VGPRs, so a buffer is free for the DMA of tile t + 2 as
//     soon as every wave has READ tile t (barrier B1, a quarter into the tile) — LDS + registers together hold three
//     tiles, and the DMA of a tile has a whole tile (2048 MFMA cycles) before it is needed;
//   * three barriers and ONE counted vmcnt wait per 64 MFMAs (the ring: one of each per 32).
// This is the loop structure of the vendor's 256 x 256 x 64 direct-to-LDS kernel, which runs the same problem sizes
// 1.2-1.4x faster than the ring; it is written here from scratch around this file's cursor / epilogue machinery.
//
// Tile t of an item, buffer c = t & 1 (holds tile t), buffer o = the other (tile t + 1 landing); MFMA slot L = 0..63,
// sub-step ks = L >> 4 uses fragment set ks:
//     L  0.. 7   one ds_read_b128 per slot: the A fragments of sets 2, 3 <- buffer c
//     L  8..15   the B fragments of sets 2, 3
//     after  9   s_waitcnt lgkmcnt(2); s_barrier (B1a)  every wave holds all A fragments of tile t: the A half is free
//     L 10..31   every third slot: LDS-DMA piece of tile t + 2 -> the A half of buffer c  (8 A pieces per wave)
//     after 17   s_waitcnt lgkmcnt(0); s_barrier (B1b)  ... and all B fragments: the B half is free
//     L 34..55   every third slot: the 8 B pieces  (a piece every 2 slots measured 3-4 % slower: the four waves' pieces
//                queue on the CU's one address unit; handing the halves back separately — the vendor loop does the
//                same — starts the DMA 8 slots earlier: +1-5 %)
//     after 50   s_waitcnt vmcnt(14); s_barrier (B2)    tile t + 1 has landed (the 14 pieces of this tile issued so far
//                may stay in flight; the 2 that follow are covered by the next tile's wait)
//     L 51..63   sets 0, 1 <- buffer o (16 reads in 13 slots); after 63: cursor + 128 B, buffers swap
// Across items: the last two tiles of an item stage the first two tiles of the workgroup's next item (the cursor is
// re-described two tiles before the end; K >= 128 per item is a launch condition), so the epilogue runs with the
// next item's tile 0 already in registers and tile 1 in flight.  The first B2 after an epilogue that issued exactly
// 32 stores waits with vmcnt(14 + 32): the stores are younger than the pieces it needs (a two-instruction uniform branch
// around the s_waitcnt — two copies of the tile body selected at run time cost the register allocator ~500 spills).  The
// second tile's wait cannot be relaxed (tile 2's pieces are younger than the stores).  Splitting the first tile around the
// previous item's epilogue, so that tile 2's pieces go out BEFORE the stores, was built and measured: 0 to -6 % (the
// front's 16 reads + 16 pieces run with the matrix pipe idle), so it is not here (profiles/r02_gemm_lab.md section 5).
//
// A piece is "s_add_u32 m0; s_nop 0; global_load_lds_dwordx4": one wait state for M0 and none for the SGPR base, which
// is only safe while hipcc never reloads that base from a spill slot (v_readlane) within 5 wait states of the piece;
// check_spills.py walks the assembly of every build and fails it otherwise.
//
// LDS buffer: A 256 rows x 128 B then B 256 rows x 128 B, chunk c of row r at r * 128 + ((c ^ gl_swz(r)) << 4)
// (gemm_layout.h); a DMA piece is 8 rows (8 lanes x 16 B per row), wave w stages rows [64 w, 64 w + 64) of both.
constexpr int kP5Op = 256 * 128;   // bytes of one operand's tile
constexpr int kP5Buf = 2 * kP5Op;  // A then B

struct P5Src {
  uint32_t a_off[8], b_off[8];  // per-lane byte offsets of the wave's 8 + 8 pieces
  const char* a_base;           // wave-uniform: the cursor's tile
  const char* b_base;
};

// Pair launch (round 6): TWO independent problems with the same activation template in one grid — workgroups [0, g0) run
// problem 0, the rest problem 1, each side persistent over its own items exactly as a single launch of that many workgroups.
// DINOv2's and SigLIP's linears at 24 images are 414 / 345 tiles: 1.62 / 1.35 rounds of 256 workgroups, i.e. two rounds each
// for 2.97 rounds of work; side by side on 138 + 118 workgroups they are 3.0 + 2.9 rounds (gemm.hip: cmb_gemm_pair picks g0).
// A workgroup picks its problem ONCE: everything below reads `p` through one reference as before.
}  // namespace
struct P5Args {
  GemmParams prob[2];
  int n_items[2];
  int g0;   // workgroups of problem 0 (single launches: the whole grid)
};
// gemm_p5_pair.hip (this file compiled with CMB_P5_PAIR_TU: the PAIR = true instantiations are their own translation unit, so the
// two halves of the ~3 minutes of hipcc run side by side under make -j)
int p5_pair_set_attr(int smem);
int p5_pair_launch(int act, unsigned grid, int smem, hipStream_t s, const P5Args& args);
namespace {

// PAIR = false is the single launch: problem 0 at fixed kernel-argument offsets, exactly the code of rounds 2-5.  (One
// instantiation for both, `p` chosen at run time, made every single launch 1-5 % slower — the parameter block is then read with
// scalar loads item by item instead of living in registers: profiles/r06_lab.md.)
template <int ACT, bool PAIR>
__global__ void __launch_bounds__(256) gemm_nt_p5_kernel(const P5Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int side = PAIR && (int)blockIdx.x >= args.g0 ? 1 : 0;
  const GemmParams& p = PAIR ? args.prob[side] : args.prob[0];
  const int n_items = PAIR ? args.n_items[side] : args.n_items[0];
  typedef bf16x8_t frag_t;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int stride = !PAIR ? (int)gridDim.x : side ? (int)gridDim.x - args.g0 : args.g0;
  const int first_item = side ? (int)blockIdx.x - args.g0 : (int)blockIdx.x;
  const int last_item = first_item + ((n_items - 1 - first_item) / stride) * stride;

  // ---- DMA side -------------------------------------------------------------------------------------------------
  const int d_row = wave * 64 + (lane >> 3);  // + 8 i
  const uint32_t dma_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem)) + (uint32_t)wave * 8192u;
  auto src_of = [&](int item) -> P5Src {
    P5Src r;
    const P4Item it = p4_item(p, item, ntiles);
    const int64_t a_row0 = row_off(p.a_map, (uint32_t)it.m0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = d_row + 8 * i;
      const int chunk = (lane & 7) ^ gl_swz(row);
      int gm = it.m0 + row;
      gm = gm < p.M ? gm : p.M - 1;
      r.a_off[i] = (uint32_t)((row_off(p.a_map, (uint32_t)gm) - a_row0) * 2 + chunk * 16);
      int gn = it.n0 + row;
      gn = gn < p.N ? gn : p.N - 1;
      r.b_off[i] = (uint32_t)((int64_t)(gn - it.n0) * p.ldb * 2 + chunk * 16);
    }
    r.a_base = p4_uniform_ptr(p.A + (a_row0 + it.kbeg) * 2);
    r.b_base = p4_uniform_ptr(p.B + ((int64_t)it.n0 * p.ldb + it.kbeg) * 2);
    return r;
  };
  P5Src cur = src_of(first_item);
  // piece pc (0..7 = A rows, 8..15 = B rows of this wave) of the cursor's tile -> buffer at LDS byte `buf` (+ this
  // wave's row block).  One statement: M0, the wait states an SGPR base re-read from a spill needs (gemm_nt_p4_kernel),
  // the DMA.
  auto piece = [&](uint32_t buf, auto pc_c) {
    constexpr int pc = decltype(pc_c)::value;
    P5Src& c = cur;
    if constexpr (pc < 8)
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(c.a_off[pc]), "s"(c.a_base), "s"(buf), "n"(pc * 1024)
                   : "memory", "m0", "scc");
    else
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(c.b_off[pc - 8]), "s"(c.b_base), "s"(buf), "n"(kP5Op + (pc - 8) * 1024)
                   : "memory", "m0", "scc");
  };
  auto advance = [&]() {
    cur.a_base += 128;
    cur.b_base += 128;
  };

  // ---- fragment read addresses (bytes inside a buffer): row (lane & 31) of block i, logical chunk 2 ks + (lane >> 5)
  uint32_t a_rd[4], b_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int koff = ((2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4;  // gl_swz(row) = (lane >> 1) & 7 for every block
    a_rd[ks] = (uint32_t)((wm * 128 + (lane & 31)) * 128 + koff);
    b_rd[ks] = (uint32_t)(kP5Op + (wn * 128 + (lane & 31)) * 128 + koff);
  }
  f32x16_t acc[4][4];
  frag_t fa[4][4], fb[4][4];  // [ks][block]
  auto read_frag = [&](int ks, uint32_t buf_off, int which) {  // which: 0..3 = A row block, 4..7 = B column block
    const char* base = smem + buf_off;
    if (which < 4) fa[ks][which] = *reinterpret_cast<const frag_t*>(base + a_rd[ks] + which * 4096);
    else fb[ks][which - 4] = *reinterpret_cast<const frag_t*>(base + b_rd[ks] + (which - 4) * 4096);
  };
  uint32_t off_c = 0, off_o = kP5Buf;  // LDS byte offsets of buffer c / o

  // ---- prologue: tiles 0 and 1 of the first item; tile 0's first half into sets 0, 1 -----------------------------------
  auto issue_tile = [&](uint32_t buf) {
    piece(buf, std::integral_constant<int, 0>{}); piece(buf, std::integral_constant<int, 1>{});
    piece(buf, std::integral_constant<int, 2>{}); piece(buf, std::integral_constant<int, 3>{});
    piece(buf, std::integral_constant<int, 4>{}); piece(buf, std::integral_constant<int, 5>{});
    piece(buf, std::integral_constant<int, 6>{}); piece(buf, std::integral_constant<int, 7>{});
    piece(buf, std::integral_constant<int, 8>{}); piece(buf, std::integral_constant<int, 9>{});
    piece(buf, std::integral_constant<int, 10>{}); piece(buf, std::integral_constant<int, 11>{});
    piece(buf, std::integral_constant<int, 12>{}); piece(buf, std::integral_constant<int, 13>{});
    piece(buf, std::integral_constant<int, 14>{}); piece(buf, std::integral_constant<int, 15>{});
  };
  issue_tile(dma_lds);
  advance();
  issue_tile(dma_lds + kP5Buf);
  advance();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  P4_BARRIER();
#pragma unroll
  for (int w = 0; w < 8; ++w) read_frag(0, 0u, p4_read_order(w));
#pragma unroll
  for (int w = 0; w < 8; ++w) read_frag(1, 0u, p4_read_order(w));

  // ---- one tile ------------------------------------------------------------------------------------------------------
  // kB2: the MFMA slot the "tile t + 1 has landed" barrier follows; the A half of a buffer is handed back (B1a, after
  // slot 9) before the B half has been read (B1b, after slot 17)
  constexpr int kB2 = 50, kTail = 63 - kB2, kDouble = 16 - kTail;
  // rl: the tile's counted wait may leave 32 more operations (an epilogue's stores) in flight; a two-instruction uniform
  // branch around the s_waitcnt, NOT a second copy of the tile body (two copies selected at run time cost the register
  // allocator hundreds of spills)
  auto run_tile = [&](bool rl) __attribute__((always_inline)) {
    auto extras = [&](auto l_c) {
      constexpr int L = decltype(l_c)::value;
      // A half first: which = 0..3 (A row blocks) of sub-steps 2, 3 at L 0..7, the B column blocks at L 8..15
      if constexpr (L < 8) read_frag(2 + (L >> 2), off_c, L & 3);
      else if constexpr (L < 16) read_frag(2 + ((L - 8) >> 2), off_c, 4 + (L & 3));
      if constexpr (L == 9) {  // the two youngest reads are B's: every A fragment of tile t is in registers
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        P4_BARRIER();
      }
      if constexpr (L == 17) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P4_BARRIER();
      }
      if constexpr (L >= 10 && L <= 31 && ((L - 10) % 3) == 0)   // A-row pieces 0..7
        piece(dma_lds + off_c, std::integral_constant<int, (L - 10) / 3>{});
      if constexpr (L >= 34 && L <= 55 && ((L - 34) % 3) == 0)   // B-row pieces 8..15
        piece(dma_lds + off_c, std::integral_constant<int, 8 + (L - 34) / 3>{});
      if constexpr (L == kB2) {  // 8 A + 6 B pieces of this tile may stay in flight
        if (rl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(14 + 32) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(14) : "memory");
        P4_BARRIER();
      }
      if constexpr (L == 63) advance();
      if constexpr (L > kB2) {  // sets 0, 1 <- buffer o: 16 reads in kTail slots, the first kDouble slots carry two
        constexpr int sl = L - kB2 - 1, r0 = sl < kDouble ? 2 * sl : sl + kDouble;
        read_frag(r0 >> 3, off_o, p4_read_order(r0 & 7));
        if constexpr (sl < kDouble) read_frag((r0 + 1) >> 3, off_o, p4_read_order((r0 + 1) & 7));
      }
      if constexpr (L == 63) {
        const uint32_t t = off_c;
        off_c = off_o;
        off_o = t;
      }
    };
    auto mfma_at = [&](auto l_c) {
      constexpr int L = decltype(l_c)::value, t = L & 15, i = t >> 2, j = t & 3, ks = L >> 4;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
      extras(l_c);
      P4_FENCE();
    };
#define P5_M(n) mfma_at(std::integral_constant<int, n>{})
#define P5_M8(n) P5_M(n); P5_M(n + 1); P5_M(n + 2); P5_M(n + 3); P5_M(n + 4); P5_M(n + 5); P5_M(n + 6); P5_M(n + 7)
    P5_M8(0); P5_M8(8); P5_M8(16); P5_M8(24); P5_M8(32); P5_M8(40); P5_M8(48); P5_M8(56);
#undef P5_M8
#undef P5_M
  };

  bool relax = false;
  constexpr uint32_t kEpiStage = 2 * kP5Buf;  // 4 x 8 KiB behind the two operand buffers
#pragma unroll 1
  for (int c_item = first_item; c_item < n_items; c_item += stride) {
    const P4Item cit = p4_item(p, c_item, ntiles);
    const int nt = cit.nk >> 1;  // 64-deep tiles of this item (>= 2)
    const int n1 = c_item + stride;
    const int next_item = n1 <= last_item ? n1 : last_item;  // past the last item the cursor re-reads it (never consumed)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // tile k stages tile k + 2: of this item while k + 2 < nt, else tile k + 2 - nt of the next item
    if (nt == 2) cur = src_of(next_item);
    run_tile(relax);
#pragma unroll 1
    for (int k = 1; k < nt; ++k) {
      if (k == nt - 2) cur = src_of(next_item);
      run_tile(false);
    }

#include "gemm_p5_epilogue.inc"
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

#ifndef CMB_P5_PAIR_TU
// Workgroups of problem 0 in a pair launch: the split g0 in [1, n_cu) that minimises max over the two sides of
// rounds x (K tiles per item + epilogue), rounds = ceil(items / workgroups) (whole XCD multiples are not needed: a side's items
// are numbered from its own first workgroup).
int p5_pair_split(int items0, int k0, int items1, int k1, int n_cu) {
  const double c0 = k0 / 64.0 + 6.0, c1 = k1 / 64.0 + 6.0;
  int best = n_cu / 2;
  double best_t = 1e30;
  static int step = 0;   // CMB_P5_PAIR_ALIGN (lab): candidate splits in multiples of this (8 keeps item % 8 == block % 8 on both sides)
  if (!step) {
    const char* e = getenv("CMB_P5_PAIR_ALIGN");
    step = e && atoi(e) > 0 ? atoi(e) : 1;
  }
  for (int g = 8; g <= n_cu - 8; g += step) {
    const double t0 = (double)((items0 + g - 1) / g) * c0, t1 = (double)((items1 + (n_cu - g) - 1) / (n_cu - g)) * c1;
    const double t = t0 > t1 ? t0 : t1;
    if (t < best_t - 1e-9) best_t = t, best = g;
  }
  return best;
}
double p5_pair_cost(int items0, int k0, int items1, int k1, int n_cu, bool paired) {
  const double c0 = k0 / 64.0 + 6.0, c1 = k1 / 64.0 + 6.0;
  if (!paired) return (double)((items0 + n_cu - 1) / n_cu) * c0 + (double)((items1 + n_cu - 1) / n_cu) * c1;
  const int g = p5_pair_split(items0, k0, items1, k1, n_cu);
  const double t0 = (double)((items0 + g - 1) / g) * c0, t1 = (double)((items1 + (n_cu - g) - 1) / (n_cu - g)) * c1;
  return t0 > t1 ? t0 : t1;
}

template <int ACT>
int launch_p5_act(GemmParams& p, int splits, hipStream_t s, GemmParams* q = nullptr) {
  constexpr int smem = 2 * kP5Buf + 4 * 8192;  // two operand buffers + the epilogue's staging area = all 160 KiB
  static CmbAttrOnce attr_once;
  static int n_cu = 0;
  // pair launches exist for the plain and the erf-GELU epilogue templates only (the residual linears of the ViT blocks are
  // plain; every further instantiation is 35 s of compile time): gemm_p5_pair_act_ok() tells cmb_gemm_pair
  constexpr bool kPairAct = ACT == CMB_ACT_NONE || ACT == CMB_ACT_GELU_ERF;
  auto kern = gemm_nt_p5_kernel<ACT, false>;
  if (q && !kPairAct) return CMB_ERR_BAD_ARG;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
            hipSuccess ||
        p5_pair_set_attr(smem) != CMB_OK)
      return CMB_ERR_LAUNCH;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      return CMB_ERR_LAUNCH;
    n_cu -= n_cu % 8;  // whole XCD rounds: item % 8 == block % 8 in every round
    if (n_cu <= 0) return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int n_items = p.tiles_m * p.tiles_n * splits;
  int g0 = n_items < n_cu ? n_items : n_cu;
  int grid = g0;
  P5Args args;
  args.prob[0] = p;
  args.n_items[0] = n_items;
  if (q) {   // pair launch: q on the workgroups [g0, n_cu)
    q->tiles_m = (q->M + 255) / 256;
    q->tiles_n = (q->N + 255) / 256;
    args.prob[1] = *q;
    args.n_items[1] = q->tiles_m * q->tiles_n;
    g0 = p5_pair_split(n_items, p.K, args.n_items[1], q->K, n_cu);
    if (const char* e = getenv("CMB_P5_PAIR_G0")) {   // lab: sweep the split
      const int v = atoi(e);
      if (v >= 8 && v <= n_cu - 8) g0 = v;
    }
    grid = n_cu;
  } else {
    args.prob[1] = p;
    args.n_items[1] = 0;
  }
  args.g0 = g0;
  if (q) return p5_pair_launch(ACT, (unsigned)grid, smem, s, args);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, s, args);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

}  // namespace

int launch_gemm_p5_bf16(GemmParams& p, int splits, hipStream_t s, GemmParams* q) {
  switch (p.slabs ? CMB_ACT_NONE : p.act) {
    case CMB_ACT_GELU_ERF: return launch_p5_act<CMB_ACT_GELU_ERF>(p, splits, s, q);
    case CMB_ACT_GELU_TANH: return launch_p5_act<CMB_ACT_GELU_TANH>(p, splits, s, q);
    case CMB_ACT_QUICK_GELU: return launch_p5_act<CMB_ACT_QUICK_GELU>(p, splits, s, q);
    case CMB_ACT_SILU: return launch_p5_act<CMB_ACT_SILU>(p, splits, s, q);
    case CMB_ACT_SWIGLU_PAIRS: return launch_p5_act<CMB_ACT_SWIGLU_PAIRS>(p, splits, s, q);
    default: return launch_p5_act<CMB_ACT_NONE>(p, splits, s, q);
  }
}

bool gemm_p5_pair_act_ok(int act) { return act == CMB_ACT_NONE || act == CMB_ACT_GELU_ERF; }

// cost model of a pair launch against the two single launches (K tiles + epilogue per item, whole rounds): > 0 = the pair wins
double gemm_p5_pair_gain(const GemmParams& a, const GemmParams& b, int n_cu) {
  const int ia = ((a.M + 255) / 256) * ((a.N + 255) / 256), ib = ((b.M + 255) / 256) * ((b.N + 255) / 256);
  const double single = p5_pair_cost(ia, a.K, ib, b.K, n_cu, false), pair = p5_pair_cost(ia, a.K, ib, b.K, n_cu, true);
  return (single - pair) / single;
}

#else   // CMB_P5_PAIR_TU: only the pair instantiations and their two entry points
}  // namespace

int p5_pair_set_attr(int smem) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_p5_kernel<CMB_ACT_NONE, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          smem) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_p5_kernel<CMB_ACT_GELU_ERF, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          smem) != hipSuccess)
    return CMB_ERR_LAUNCH;
  return CMB_OK;
}

int p5_pair_launch(int act, unsigned grid, int smem, hipStream_t s, const P5Args& args) {
  if (act == CMB_ACT_GELU_ERF) hipLaunchKernelGGL((gemm_nt_p5_kernel<CMB_ACT_GELU_ERF, true>), dim3(grid), dim3(256), smem, s, args);
  else if (act == CMB_ACT_NONE) hipLaunchKernelGGL((gemm_nt_p5_kernel<CMB_ACT_NONE, true>), dim3(grid), dim3(256), smem, s, args);
  else return CMB_ERR_BAD_ARG;
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}
#endif

}  // namespace cmb_gemm_detail
