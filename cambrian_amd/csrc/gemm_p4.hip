// gemm_p4.hip — persistent large-tile configuration of C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T), bf16, gfx950.
//
// Why a second 256 x 256 kernel (DESIGN.md §4 "gemm_nt_p4_kernel"): the 8-wave / 8-phase kernel of gemm256.hip keeps the
// matrix pipe 60 % busy — a wave owns 128 x 64 of the tile, so every K-step moves 24 KB of fragments per wave through
// the LDS, each phase is fenced by two workgroup barriers, and prologue + epilogue of every tile run with the pipe
// idle (one workgroup per CU).  This kernel removes all three:
//   * 4 waves (one per SIMD, 512 registers each), a wave owns 128 x 128 = 4 x 4 tiles of 32 x 32: 256 accumulator
//     registers, 16 KB of fragments per wave per 64-deep K-step (2/3 of the LDS traffic per MFMA), 32 MFMAs between
//     barriers, the other waves' instructions never contend for the SIMD;
//   * the LDS is a ring of NS stages of 32 k-values (A: 256 rows x 64 B, B: 256 rows x 64 B = 32 KiB per stage) filled
//     by LDS-DMA (global_load_lds_dwordx4) NS - 2 stages ahead with a counted vmcnt; ONE barrier per stage;
//   * the workgroup is PERSISTENT (grid = number of CUs): the DMA cursor runs ahead across tile boundaries, so the
//     next tile's first stages are in flight / landed while the current tile's epilogue runs, and the epilogue
//     leaves straight from the accumulators (v_permlane32_swap pairs two half-waves' column groups into 8 consecutive
//     columns: 16-byte bf16 stores, no LDS round trip, no barrier), its stores draining under the next tile's MFMAs.
//
// One wave per SIMD means every cycle an LDS / vector-memory instruction spends waiting to be ACCEPTED is a cycle the
// matrix pipe idles.  Measured (tools/gemm_lab.py, 8192^3, profiles/r02_gemm_lab.md): the MFMA stream alone runs 1.94-2.04
// PFLOP/s; with the fragment reads 1.55-1.6; with the LDS-DMA pieces 1.19-1.30.  A per-wave stagger of the issue slots
// (one wave reads while another stages) was tried and measured nothing; what rides behind each MFMA is therefore the
// same for all four waves: groups of four slots carrying two reads, the M0 write, one DMA piece, the cursor hand-over.
// This ring kernel ended within +-7 % of the 8-wave kernel and is opt-in (tile_hint 2570); gemm_nt_p5_kernel further
// down is the design that came out of its ablations and is the default.
//
// Stage g (ring slot g % NS; fragments of the two 16-deep sub-steps ks = 0, 1 live in register sets 0 / 1):
//     phase A:  16 MFMA on set 0 | reads of (g, ks 1) -> set 1   | DMA pieces 4..7 of stage g + NS - 2 -> slot (g - 2) % NS
//               s_waitcnt vmcnt(8 (NS - 3))     this wave's pieces of stage g + 1 have landed
//               s_barrier                       => stage g + 1 is visible to every wave
//     phase B:  16 MFMA on set 1 | reads of (g + 1, ks 0) -> set 0 | DMA pieces 0..3 of stage g + NS - 1 -> slot (g - 1) % NS
// RAW: stage g + 1 was issued in phase B (g + 2 - NS) / phase A (g + 3 - NS); at the wait the NS - 3 younger stages
//      (8 pieces per wave each) are all that may still be in flight; the wait precedes the barrier and the first read
//      of stage g + 1 follows it.  WAR: slot (g - 1) % NS is refilled after barrier g; its last reads (ks 1 of stage
//      g - 1) were issued in phase A (g - 1) and each feeds an MFMA of phase B (g - 1), so every wave has retired them
//      (the compiler's own counted lgkmcnt in front of those MFMAs) before it reaches barrier g.  No lgkmcnt(0)
//      drain is needed anywhere.  Global stores of an epilogue also count on vmcnt: they are younger than the stage
//      being waited for, so the counted wait can only wait longer, never shorter (a wave's memory operations retire
//      in order on gfx9).
//
// LDS stage layout (gemm_layout.h documents the 128-byte-row variant): rows of 64 B = 4 chunks of 16 B; chunk c of row
// r lives at r*64 + ((c ^ ((r >> 2) & 3)) << 4).  ds_read_b128 is serviced in 16-lane groups over a 256-byte bank row
// (4 tile rows); the lanes of a group read 16 different rows at one logical chunk, and (r & 3, (r >> 2) & 3) is distinct
// for them, so fragment reads are conflict-free.  The swizzle is applied to the DMA's per-lane SOURCE address.
#include <type_traits>
#include "gemm_common.h"

namespace cmb_gemm_detail {
namespace {

constexpr int kP4StageA = 256 * 64;          // bytes of one operand's stage (256 rows x 32 bf16)
constexpr int kP4Stage = 2 * kP4StageA;      // A then B
constexpr int kP4Pieces = 8;                 // LDS-DMA instructions per wave per stage (4 A + 4 B)

#define P4_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)

// One LDS-DMA piece: 64 lanes x 16 B, wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset -> LDS bytes
// [lds_base + OFF + 16*lane, +16).  M0 (the LDS destination) is written and used in one statement and not restored:
// nothing else in this kernel reads M0 (gfx9 DS instructions do not), which keeps a piece at 3 issue slots — with one
// wave per SIMD every slot beyond ~7 per MFMA is matrix-pipe idle time (5 slots measured 42 cycles per piece).
// A base that came out of v_readfirstlane needs 5 wait states before a VMEM instruction reads it: the callers'
// SALU arithmetic on the base after every hand-over provides them.
template <int OFF>
__device__ __forceinline__ void p4_glds(const char* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_base), "n"(OFF)
               : "memory", "m0", "scc");
}

// 16 consecutive floats at a wave-uniform address (scalar loads: they count on lgkmcnt, so unlike a vector load they
// do not have to wait behind the tile's stores and LDS-DMA on vmcnt) -> this lane's 8: lanes 0-31 take floats 0..7,
// lanes 32-63 floats 8..15 (the column halves of one 16-column group after the permlane32 swap).
typedef float f32x16s_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void p4_colvec(const float* base, bool upper, float (&v)[8]) {
  f32x16s_t o;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(o) : "s"(base) : "memory");
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float lo = o[e], hi = o[8 + e];
    asm volatile("" : "+s"(lo), "+s"(hi));  // keep two scalars: select-of-extract otherwise becomes a dynamic extract
    v[e] = upper ? hi : lo;
  }
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

// per-lane / per-wave DMA source description of one item
struct P4Src {
  uint32_t a_off[4], b_off[4];  // per-lane byte offsets of the wave's 4 + 4 pieces
  const char* a_base;           // wave-uniform: first stage of the item
  const char* b_base;
  int nk;
};

// order in which a phase reads the NEXT phase's fragments (0..3 = A row block i, 4..7 = B column block j): the MFMA
// order is (i, j) = (t >> 2, t & 3), so B0, A0 are needed first and A3 only by the 13th MFMA
__host__ __device__ constexpr int p4_read_order(int n) {
  return n == 0 ? 4 : n == 1 ? 0 : n == 2 ? 5 : n == 3 ? 6 : n == 4 ? 7 : n - 4;
}

// VAR: ablation bits (tools/gemm_lab.py, built with -DCMB_GEMM_LAB; all but 4 compute garbage — timing only):
// 1 = no in-loop DMA, 2 = no in-loop fragment reads, 4 = no stagger (every wave runs wave 0's schedule), 8 = no barrier.
template <int ACT, int NS, int VAR>
__global__ void __launch_bounds__(256) gemm_nt_p4_kernel(const GemmParams p, const int n_items) {
  constexpr bool kNoDma = VAR & 1, kNoRead = VAR & 2, kNoBar = VAR & 8;
  constexpr bool kPlainLoad = VAR & 16, kDummyWrite = VAR & 32;  // issue-cost probes (garbage results)
  constexpr bool kBufLds = VAR & 64, kBufVgpr = VAR & 128;         // buffer_load forms of the piece (timing probes)
  constexpr bool kFrozen = VAR & 256;  // the cursor never moves: every piece re-reads the tile's first stage (cache-hot probe)
  static_assert(NS >= 4, "the ring needs the stage being read, the next one and two being filled");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16x8_t frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int stride = (int)gridDim.x;
  const int first_item = (int)blockIdx.x;
  const int last_item = first_item + ((n_items - 1 - first_item) / stride) * stride;  // this workgroup's last item

  // ---- DMA side -------------------------------------------------------------------------------------------------
  // wave w stages rows [64 w, 64 w + 64) of A and of B: piece i covers rows 64 w + 16 i + (lane >> 2), logical chunk
  // (lane & 3) ^ swz(row) with swz(row) = (row >> 2) & 3 = (lane >> 4) & 3.
  const int d_row = wave * 64 + (lane >> 2);
  const int d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const uint32_t dma_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem)) + (uint32_t)wave * 4096u;
  auto src_of = [&](int item) -> P4Src {
    P4Src r;
    const P4Item it = p4_item(p, item, ntiles);
    const int64_t a_row0 = row_off(p.a_map, (uint32_t)it.m0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int gm = it.m0 + d_row + 16 * i;
      gm = gm < p.M ? gm : p.M - 1;
      r.a_off[i] = (uint32_t)((row_off(p.a_map, (uint32_t)gm) - a_row0) * 2 + d_chunk * 16);
      int gn = it.n0 + d_row + 16 * i;
      gn = gn < p.N ? gn : p.N - 1;
      r.b_off[i] = (uint32_t)((int64_t)(gn - it.n0) * p.ldb * 2 + d_chunk * 16);
    }
    r.a_base = p4_uniform_ptr(p.A + (a_row0 + it.kbeg) * 2);
    r.b_base = p4_uniform_ptr(p.B + ((int64_t)it.n0 * p.ldb + it.kbeg) * 2);
    r.nk = it.nk;
    return r;
  };
  // The cursor (cur) walks the workgroup's items ahead of the MFMAs; nxt describes the item after the cursor's.  The
  // hand-over cur <- nxt at the end of an item is a register select (no branch inside the stage loop); nxt is
  // refreshed once per tile, next to the epilogue.  Past the last item the cursor re-reads it: those stages land in
  // ring slots nobody reads any more and are drained before the workgroup ends.
  P4Src cur = src_of(first_item);
  P4Src nxt = src_of(first_item + stride <= last_item ? first_item + stride : last_item);
  // The A-row pieces (0..3) of a stage go out in one phase, its B-row pieces (4..7) in the next: the two halves of
  // the cursor advance separately (each right after its own last piece), with their own stage counters.
  int ka = 0, kb = 0, nka = cur.nk, nkb = cur.nk;
  asm volatile("s_nop 4" ::: "memory");  // v_readfirstlane -> SGPR base -> first LDS-DMA: 5 wait states
  // M0 <- LDS destination of piece pc in the ring slot whose first byte (for this wave) is `slot_lds`; issued one
  // MFMA slot ahead of the piece itself (SALU write of M0 -> LDS-DMA needs a wait state; an s_nop in the same slot as
  // the DMA is a slot the matrix pipe does not get)
  auto dma_m0 = [&](uint32_t slot_lds, auto pc_c) {
    constexpr int pc = decltype(pc_c)::value;
    constexpr int off = pc < 4 ? pc * 1024 : kP4StageA + (pc - 4) * 1024;
    if constexpr (!kPlainLoad) asm volatile("s_add_u32 m0, %0, %1" ::"s"(slot_lds), "n"(off) : "m0", "scc");
  };
  f32x4_t probe_buf[8];  // probes only: a "+v" chain pins each load's destination to one register for the whole kernel
  if constexpr (kPlainLoad || kBufVgpr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) probe_buf[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // piece pc (0..3 = A rows, 4..7 = B rows of this wave) of the cursor's stage -> LDS at M0.  The s_nop covers a base
  // that hipcc has just re-read from an SGPR spill (v_readlane writes the SGPR, and a VMEM instruction may read it
  // only 5 wait states later; hipcc pads its own instructions, not ours: without it the DMA fetches from a stale base).
  auto dma_piece = [&](auto pc_c) {
    constexpr int pc = decltype(pc_c)::value;
    P4Src& c = cur;  // (named here: a variable used only inside `if constexpr` arms of a generic lambda is not captured)
    auto& pb = probe_buf;
    if constexpr (kBufLds || kBufVgpr) {
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const uint64_t b = (uint64_t)(pc < 4 ? c.a_base : c.b_base);
      const u32x4_t srd = {(uint32_t)b, (uint32_t)(b >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
      const uint32_t vo = pc < 4 ? c.a_off[pc] : c.b_off[pc - 4];
      if constexpr (kBufLds) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(srd) : "memory");
      else asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(pb[pc]) : "v"(vo), "s"(srd) : "memory");
    } else if constexpr (kPlainLoad) {  // a VGPR-destination load of the same bytes (never waited for, never used)
      if constexpr (pc < 4)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(pb[pc]) : "v"(c.a_off[pc]), "s"(c.a_base) : "memory");
      else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(pb[pc]) : "v"(c.b_off[pc - 4]), "s"(c.b_base) : "memory");
    } else if constexpr (pc < 4) {
      asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(c.a_off[pc]), "s"(c.a_base) : "memory");
    } else {
      asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(c.b_off[pc - 4]), "s"(c.b_base) : "memory");
    }
  };
  // hand-over of one operand's half of the cursor, in two parts that each fit one MFMA slot
  bool sw_a = false, sw_b = false;
  auto adv_a = [&](int part) {
    if constexpr (kFrozen) return;
    if (part == 0) {
      ++ka;
      sw_a = (ka == nka);
      cur.a_base = sw_a ? nxt.a_base : cur.a_base + 64;
      nka = sw_a ? nxt.nk : nka;
      ka = sw_a ? 0 : ka;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cur.a_off[i] = sw_a ? nxt.a_off[i] : cur.a_off[i];
    }
  };
  auto adv_b = [&](int part) {
    if constexpr (kFrozen) return;
    if (part == 0) {
      ++kb;
      sw_b = (kb == nkb);
      cur.b_base = sw_b ? nxt.b_base : cur.b_base + 64;
      nkb = sw_b ? nxt.nk : nkb;
      kb = sw_b ? 0 : kb;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cur.b_off[i] = sw_b ? nxt.b_off[i] : cur.b_off[i];
    }
  };

  // ---- fragment read addresses (bytes inside a stage) ------------------------------------------------------------
  const int f_swz = (lane >> 2) & 3;
  uint32_t a_rd[2], b_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int koff = ((2 * ks + (lane >> 5)) ^ f_swz) << 4;
    a_rd[ks] = (uint32_t)((wm * 128 + (lane & 31)) * 64 + koff);
    b_rd[ks] = (uint32_t)(kP4StageA + (wn * 128 + (lane & 31)) * 64 + koff);
  }

  f32x16_t acc[4][4];
  frag_t fa[2][4], fb[2][4];  // [set][i or j]

  auto read_frag = [&](int set, uint32_t slot_off, int which) {  // which: 0..3 = A row block, 4..7 = B column block
    const char* base = smem + slot_off;
    if (which < 4) fa[set][which] = *reinterpret_cast<const frag_t*>(base + a_rd[set] + which * 2048);
    else fb[set][which - 4] = *reinterpret_cast<const frag_t*>(base + b_rd[set] + (which - 4) * 2048);
  };

  // ---- prologue: stages 0 .. NS - 3 whole, the first half of stage NS - 2 (its second half goes out in phase A of
  // the first stage, as in steady state) --------------------------------------------------------------------------
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  typedef std::integral_constant<int, 4> I4;
  typedef std::integral_constant<int, 5> I5;
  typedef std::integral_constant<int, 6> I6;
  typedef std::integral_constant<int, 7> I7;
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) {
    const uint32_t l = dma_lds + (uint32_t)st * (uint32_t)kP4Stage;
    dma_m0(l, I0{}); asm volatile("s_nop 0"); dma_piece(I0{});
    dma_m0(l, I1{}); asm volatile("s_nop 0"); dma_piece(I1{});
    dma_m0(l, I2{}); asm volatile("s_nop 0"); dma_piece(I2{});
    dma_m0(l, I3{}); asm volatile("s_nop 0"); dma_piece(I3{});
    if (st < NS - 2) {
      dma_m0(l, I4{}); asm volatile("s_nop 0"); dma_piece(I4{});
      dma_m0(l, I5{}); asm volatile("s_nop 0"); dma_piece(I5{});
      dma_m0(l, I6{}); asm volatile("s_nop 0"); dma_piece(I6{});
      dma_m0(l, I7{}); asm volatile("s_nop 0"); dma_piece(I7{});
      adv_a(0); adv_a(1); adv_b(0); adv_b(1);
    }
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP4Pieces * (NS - 3) + 4) : "memory");  // stage 0 has landed
  P4_BARRIER();
#pragma unroll
  for (int w = 0; w < 8; ++w) read_frag(0, 0u, p4_read_order(w));

  // ---- main loop: this workgroup's items, each a run of branch-free stages -----------------------------------------
  // LDS byte offsets of the ring slots of the running stage index g: g % NS, (g + 1) % NS, and this wave's DMA
  // destinations in slots (g - 1) % NS, (g - 2) % NS
  uint32_t off_cur = 0, off_next = kP4Stage;
  uint32_t dst_m1 = dma_lds + (NS - 1) * kP4Stage, dst_m2 = dma_lds + (NS - 2) * kP4Stage;
  // One stage = 32 MFMA slots L = 16 * phase + t.  What rides behind each MFMA is laid out so that no slot carries
  // more than one memory instruction or ~7 scalar / 4 vector instructions (with one wave per SIMD an overfull slot is
  // matrix-pipe idle time), in groups of four slots q = t / 4:
  //     t % 4 == 0   two ds_read_b128 of the next phase's fragments
  //     t % 4 == 1   M0 <- destination of the group's DMA piece
  //     t % 4 == 2   the DMA piece (phase A: B-row pieces 4..7 of the cursor's stage; phase B: A-row pieces 0..3 of the next)
  //     t % 4 == 3   cursor hand-over (phase A: the A half, whose pieces went out in the previous phase B; phase B: the
  //                  B half), ring rotation
  // RELAX: the stage runs right after an epilogue that issued 32 global stores behind the DMA pieces this stage waits
  // for, so 32 more operations may be outstanding at its counted wait.
  auto stage = [&](auto relax_c) {
    constexpr bool RELAX = decltype(relax_c)::value;
    auto extras = [&](auto l_c) {
      constexpr int L = decltype(l_c)::value, t = L & 15, q = t >> 2, r = t & 3;
      constexpr bool phase_b = L >= 16;
      if constexpr (r == 0 && !kNoRead) {
        read_frag(phase_b ? 0 : 1, phase_b ? off_next : off_cur, p4_read_order(2 * q));
        read_frag(phase_b ? 0 : 1, phase_b ? off_next : off_cur, p4_read_order(2 * q + 1));
      }
      if constexpr (r == 1 && !kNoDma) {
        if constexpr (phase_b) dma_m0(dst_m1, std::integral_constant<int, q>{});
        else dma_m0(dst_m2, std::integral_constant<int, 4 + q>{});
      }
      if constexpr (r == 2 && !kNoDma) {
        if constexpr (phase_b) dma_piece(std::integral_constant<int, q>{});
        else dma_piece(std::integral_constant<int, 4 + q>{});
      }
      if constexpr (L == 3) adv_a(0);
      if constexpr (L == 7) adv_a(1);
      if constexpr (L == 19) adv_b(0);
      if constexpr (L == 23) adv_b(1);
      if constexpr (L == 31) {  // ring rotation (every piece and read of this stage has been issued)
        dst_m2 = dst_m1;
        dst_m1 = dma_lds + off_cur;
        off_cur = off_next;
        off_next = (off_next + kP4Stage == NS * kP4Stage) ? 0u : off_next + kP4Stage;
      }
    };
    auto mfma_at = [&](auto l_c) {
      constexpr int L = decltype(l_c)::value, t = L & 15, i = t >> 2, j = t & 3, set = L >> 4;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
      extras(l_c);
      P4_FENCE();
    };
#define P4_M(n) mfma_at(std::integral_constant<int, n>{})
    // phase A: sub-step 0 | reads of sub-step 1 | B-row pieces of the cursor's stage into slot g - 2
    P4_M(0); P4_M(1); P4_M(2); P4_M(3); P4_M(4); P4_M(5); P4_M(6); P4_M(7);
    P4_M(8); P4_M(9); P4_M(10); P4_M(11); P4_M(12); P4_M(13); P4_M(14); P4_M(15);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP4Pieces * (NS - 3) + (RELAX ? 32 : 0)) : "memory");  // stage g + 1 landed
    P4_FENCE();
    if (!kNoBar) P4_BARRIER();
    P4_FENCE();
    // phase B: sub-step 1 | reads of (g + 1, sub-step 0) | A-row pieces of the next stage into slot g - 1
    P4_M(16); P4_M(17); P4_M(18); P4_M(19); P4_M(20); P4_M(21); P4_M(22); P4_M(23);
    P4_M(24); P4_M(25); P4_M(26); P4_M(27); P4_M(28); P4_M(29); P4_M(30); P4_M(31);
#undef P4_M
  };
  auto run_stages = [&](int nk, bool relax) {
    // the first two stages after an epilogue are peeled: their counted waits may leave that epilogue's stores in flight
    // (stage s is issued in phase B (s - 4) / phase A (s - 3): both stages' needs went out before the stores)
    if (relax) {
      stage(std::true_type{});
      stage(std::true_type{});
    } else {
      stage(std::false_type{});
      stage(std::false_type{});
    }
#pragma unroll 1
    for (int k = 2; k < nk; ++k) stage(std::false_type{});
  };

  bool relax = false;  // did this wave's previous epilogue issue exactly its 32 stores (and nothing it waited for after them)?
#pragma unroll 1
  for (int c_item = first_item; c_item < n_items; c_item += stride) {
    const P4Item cit = p4_item(p, c_item, ntiles);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    run_stages(cit.nk, relax);

    constexpr uint32_t kEpiStage = 0;  // the ring fills the whole LDS: direct stores
#include "gemm_p4_epilogue.inc"
    // the cursor is now inside item c_item + stride (nk >= NS is a launch condition): describe the one after it
    const int n2 = c_item + 2 * stride;
    nxt = src_of(n2 <= last_item ? n2 : last_item);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
  if constexpr (kPlainLoad || kBufVgpr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(probe_buf[i]));
  }
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

// VAR (lab builds, timing only — 1, 4, 8 compute garbage): 1 = no in-loop DMA, 2 = s_nop 4 in front of every piece,
// 4 = frozen cursor (every piece re-reads the item's first tile: cache-hot), 8 = unswizzled DMA source, 16 = B1 three
// slots later with a piece every 2 slots, 32 = a piece every 2 slots, 64 = buffer_load ... lds form of the piece.
template <int ACT, int VAR>
__global__ void __launch_bounds__(256) gemm_nt_p5_kernel(const GemmParams p, const int n_items) {
  constexpr bool kNoDma = VAR & 1, kNoNop = !(VAR & 2), kFrozen = VAR & 4, kLinear = VAR & 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16x8_t frag_t;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int stride = (int)gridDim.x;
  const int first_item = (int)blockIdx.x;
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
      const int chunk = kLinear ? (lane & 7) : ((lane & 7) ^ gl_swz(row));
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
    if constexpr ((VAR & 64) != 0) {  // the buffer form of the same piece (timing probe)
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const uint64_t b = (uint64_t)(pc < 8 ? c.a_base : c.b_base);
      const u32x4_t srd = {(uint32_t)b, (uint32_t)(b >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
      const uint32_t vo = pc < 8 ? c.a_off[pc] : c.b_off[pc - 8];
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                   :
                   : "v"(vo), "s"(srd), "s"(buf), "n"(pc < 8 ? pc * 1024 : kP5Op + (pc - 8) * 1024)
                   : "memory", "m0", "scc");
    } else if constexpr (kNoNop) {
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
    } else if constexpr (pc < 8)
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(c.a_off[pc]), "s"(c.a_base), "s"(buf), "n"(pc * 1024)
                   : "memory", "m0", "scc");
    else
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(c.b_off[pc - 8]), "s"(c.b_base), "s"(buf), "n"(kP5Op + (pc - 8) * 1024)
                   : "memory", "m0", "scc");
  };
  auto advance = [&]() {
    if constexpr (kFrozen) return;
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
  // kB1 / kB2: the MFMA slots the two barriers follow; PH: this wave issues its pieces on odd (1) or even (0) slots
  constexpr int kStep = (VAR & 32) ? 2 : 3;  // MFMA slots between two DMA pieces
  constexpr bool kSplit = (VAR & 16) == 0;   // the A half of a buffer is handed back before the B half has been read (16: lab, one hand-over)
  constexpr int kB1 = 17, kB2 = kB1 + 33, kTail = 63 - kB2, kDouble = 16 - kTail;
  // rl: the tile's counted wait may leave 32 more operations (an epilogue's stores) in flight; a two-instruction uniform
  // branch around the s_waitcnt, NOT a second copy of the tile body (two copies selected at run time cost the register
  // allocator hundreds of spills)
  auto tile = [&](bool rl, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value, P0 = kB1 + 1 + PH;
    auto extras = [&](auto l_c) {
      constexpr int L = decltype(l_c)::value;
      if constexpr (kSplit) {
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
        if constexpr (L >= 10 && L <= 31 && ((L - 10) % 3) == 0 && !kNoDma)   // A-row pieces 0..7
          piece(dma_lds + off_c, std::integral_constant<int, (L - 10) / 3>{});
        if constexpr (L >= 34 && L <= 55 && ((L - 34) % 3) == 0 && !kNoDma)   // B-row pieces 8..15
          piece(dma_lds + off_c, std::integral_constant<int, 8 + (L - 34) / 3>{});
        if constexpr (L == kB2) {  // 8 A + 6 B pieces of this tile may stay in flight
          if (rl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(14 + 32) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(14) : "memory");
          P4_BARRIER();
        }
      } else {
      if constexpr (L < 16) read_frag(2 + (L >> 3), off_c, p4_read_order(L & 7));
      if constexpr (L == kB1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P4_BARRIER();
      }
      if constexpr (L >= P0 && L <= P0 + 15 * kStep && ((L - P0) % kStep) == 0 && !kNoDma)
        piece(dma_lds + off_c, std::integral_constant<int, (L - P0) / kStep>{});
      if constexpr (L == kB2) {  // the pieces of this tile issued so far may stay in flight
        constexpr int mine = (kB2 - P0) / kStep + 1 > 16 ? 16 : (kB2 - P0) / kStep + 1;
        if (rl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(mine + 32) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(mine) : "memory");
        P4_BARRIER();
      }
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
  auto run_tile = [&](bool rl) __attribute__((always_inline)) { tile(rl, std::integral_constant<int, 0>{}); };

  if constexpr ((VAR & 1024) != 0) {  // lab probe: phase-shift the workgroups of an XCD by eighths of an item
    const int phase = ((int)blockIdx.x >> 3) & 7;
    const int n64 = phase * (p4_item(p, first_item, ntiles).nk >> 1) * 4;  // x 64 cycles: phase * (nt * 2048 / 8) cycles
    for (int q = 0; q < n64; ++q) asm volatile("s_sleep 1" ::: "memory");
  }
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

#include "gemm_p4_epilogue.inc"
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int ACT, int VAR>
int launch_p5_act(GemmParams& p, int splits, hipStream_t s) {
  constexpr int smem = 2 * kP5Buf + 4 * 8192;  // two operand buffers + the epilogue's staging area = all 160 KiB
  static bool attr_done = false;
  static int n_cu = 0;
  auto kern = gemm_nt_p5_kernel<ACT, VAR>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess)
      return CMB_ERR_LAUNCH;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      return CMB_ERR_LAUNCH;
    n_cu -= n_cu % 8;  // whole XCD rounds: item % 8 == block % 8 in every round
    if (n_cu <= 0) return CMB_ERR_LAUNCH;
    attr_done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int n_items = p.tiles_m * p.tiles_n * splits;
  const int grid = (VAR & 4096) ? n_items : (n_items < n_cu ? n_items : n_cu);  // (4096: lab probe — one item per workgroup)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, s, p, n_items);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

template <int ACT, int NS, int VAR>
int launch_p4_act(GemmParams& p, int splits, hipStream_t s) {
  constexpr int smem = NS * kP4Stage;
  static bool attr_done = false;
  static int n_cu = 0;
  auto kern = gemm_nt_p4_kernel<ACT, NS, VAR>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess)
      return CMB_ERR_LAUNCH;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      return CMB_ERR_LAUNCH;
    n_cu -= n_cu % 8;  // whole XCD rounds: item % 8 == block % 8 in every round
    if (n_cu <= 0) return CMB_ERR_LAUNCH;
    attr_done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int n_items = p.tiles_m * p.tiles_n * splits;
  const int grid = n_items < n_cu ? n_items : n_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, s, p, n_items);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}

template <int NS, int VAR>
int launch_p4_ns(GemmParams& p, int splits, hipStream_t s) {
  switch (p.slabs ? CMB_ACT_NONE : p.act) {
    case CMB_ACT_GELU_ERF: return launch_p4_act<CMB_ACT_GELU_ERF, NS, VAR>(p, splits, s);
    case CMB_ACT_GELU_TANH: return launch_p4_act<CMB_ACT_GELU_TANH, NS, VAR>(p, splits, s);
    case CMB_ACT_QUICK_GELU: return launch_p4_act<CMB_ACT_QUICK_GELU, NS, VAR>(p, splits, s);
    case CMB_ACT_SILU: return launch_p4_act<CMB_ACT_SILU, NS, VAR>(p, splits, s);
    default: return launch_p4_act<CMB_ACT_NONE, NS, VAR>(p, splits, s);
  }
}

}  // namespace

int launch_gemm_p5_bf16(GemmParams& p, int splits, int var, hipStream_t s) {
#ifdef CMB_GEMM_LAB
  switch (var) {
    case 1: return launch_p5_act<CMB_ACT_NONE, 1>(p, splits, s);
    case 2: return launch_p5_act<CMB_ACT_NONE, 2>(p, splits, s);
    case 4: return launch_p5_act<CMB_ACT_NONE, 4>(p, splits, s);
    case 6: return launch_p5_act<CMB_ACT_NONE, 6>(p, splits, s);
    case 8: return launch_p5_act<CMB_ACT_NONE, 8>(p, splits, s);
    case 10: return launch_p5_act<CMB_ACT_NONE, 10>(p, splits, s);
    case 12: return launch_p5_act<CMB_ACT_NONE, 12>(p, splits, s);
    case 16: return launch_p5_act<CMB_ACT_NONE, 16>(p, splits, s);
    case 17: return launch_p5_act<CMB_ACT_NONE, 17>(p, splits, s);
    case 18: return launch_p5_act<CMB_ACT_NONE, 18>(p, splits, s);
    case 4096: return p.act == CMB_ACT_GELU_ERF ? launch_p5_act<CMB_ACT_GELU_ERF, 4096>(p, splits, s) : launch_p5_act<CMB_ACT_NONE, 4096>(p, splits, s);
    case 1024: return p.act == CMB_ACT_GELU_ERF ? launch_p5_act<CMB_ACT_GELU_ERF, 1024>(p, splits, s) : launch_p5_act<CMB_ACT_NONE, 1024>(p, splits, s);
    case 2048: return p.act == CMB_ACT_GELU_ERF ? launch_p5_act<CMB_ACT_GELU_ERF, 2048>(p, splits, s) : launch_p5_act<CMB_ACT_NONE, 2048>(p, splits, s);
    case 512: return p.act == CMB_ACT_GELU_ERF ? launch_p5_act<CMB_ACT_GELU_ERF, 512>(p, splits, s) : launch_p5_act<CMB_ACT_NONE, 512>(p, splits, s);
    case 32: return launch_p5_act<CMB_ACT_NONE, 32>(p, splits, s);
    case 34: return launch_p5_act<CMB_ACT_NONE, 34>(p, splits, s);
    case 80: return launch_p5_act<CMB_ACT_NONE, 80>(p, splits, s);
    default: break;
  }
#endif
  (void)var;
  switch (p.slabs ? CMB_ACT_NONE : p.act) {
    case CMB_ACT_GELU_ERF: return launch_p5_act<CMB_ACT_GELU_ERF, 0>(p, splits, s);
    case CMB_ACT_GELU_TANH: return launch_p5_act<CMB_ACT_GELU_TANH, 0>(p, splits, s);
    case CMB_ACT_QUICK_GELU: return launch_p5_act<CMB_ACT_QUICK_GELU, 0>(p, splits, s);
    case CMB_ACT_SILU: return launch_p5_act<CMB_ACT_SILU, 0>(p, splits, s);
    default: return launch_p5_act<CMB_ACT_NONE, 0>(p, splits, s);
  }
}

// var: ablation bits of the kernel template (0 = production); ablations exist for act = none in lab builds only
int launch_gemm_p4_bf16(GemmParams& p, int splits, int ns, int var, hipStream_t s) {
#ifdef CMB_P4_QUICK  // compile-time experiments: one instantiation only
  return launch_p4_act<CMB_ACT_NONE, 5, 0>(p, splits, s);
#else
  switch (var) {
#ifdef CMB_GEMM_LAB
    case 1: return launch_p4_act<CMB_ACT_NONE, 5, 1>(p, splits, s);
    case 2: return launch_p4_act<CMB_ACT_NONE, 5, 2>(p, splits, s);
    case 3: return launch_p4_act<CMB_ACT_NONE, 5, 3>(p, splits, s);
    case 4: return launch_p4_act<CMB_ACT_NONE, 5, 4>(p, splits, s);
    case 5: return launch_p4_act<CMB_ACT_NONE, 5, 5>(p, splits, s);
    case 6: return launch_p4_act<CMB_ACT_NONE, 5, 6>(p, splits, s);
    case 8: return launch_p4_act<CMB_ACT_NONE, 5, 8>(p, splits, s);
    case 16: return launch_p4_act<CMB_ACT_NONE, 5, 16>(p, splits, s);
    case 256: return launch_p4_act<CMB_ACT_NONE, 5, 256>(p, splits, s);
    case 258: return launch_p4_act<CMB_ACT_NONE, 5, 258>(p, splits, s);
    case 64: return launch_p4_act<CMB_ACT_NONE, 5, 64>(p, splits, s);
#endif
    default: return launch_p4_ns<5, 0>(p, splits, s);
  }
#endif
}

}  // namespace cmb_gemm_detail
