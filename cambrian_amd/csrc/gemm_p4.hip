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
//     by LDS-DMA (global_load_lds_dwordx4) NS stages ahead with a counted vmcnt; ONE barrier per stage;
//   * the workgroup is PERSISTENT (grid = number of CUs): the DMA cursor runs ahead across tile boundaries, so the
//     next tile's first NS stages are in flight / landed while the current tile's epilogue runs, and the epilogue
//     leaves straight from the accumulators (v_permlane32_swap pairs two half-waves' column groups into 8 consecutive
//     columns: 16-byte bf16 stores, no LDS round trip, no barrier), its stores draining under the next tile's MFMAs.
//
// In-wave software pipeline of one stage g (stage slot g % NS; fragments of the two 16-deep sub-steps ks = 0, 1 live
// in register sets 0 / 1):
//     phase A:  16 MFMA on set 0   |  8 ds_read_b128 of (g, ks = 1) -> set 1   (interleaved with the first 8 MFMAs)
//               s_waitcnt lgkmcnt(0)            this wave's last reads of slot g % NS have returned
//               s_waitcnt vmcnt(8 (NS - 2))     this wave's DMA pieces of stage g + 1 have landed
//               s_barrier                       => stage g + 1 is visible to every wave, slot g % NS is free
//     phase B:  16 MFMA on set 1   |  8 ds_read_b128 of (g + 1, ks = 0) -> set 0  |  8 LDS-DMA of stage g + NS -> slot g % NS
//               s_waitcnt lgkmcnt(0)
// RAW: the DMA of stage g + 1 was issued NS - 1 iterations earlier; the counted wait leaves exactly the NS - 2 younger
//      stages (8 instructions per wave each) in flight, precedes the barrier, and the first read of stage g + 1 follows
//      the barrier.  WAR: slot g % NS is re-filled only after the barrier that every wave reaches with its reads of
//      that slot retired (lgkmcnt(0)).  Global stores of an epilogue also count on vmcnt: they are younger than the
//      stage being waited for, so the counted wait can only wait longer, never shorter (memory operations of a wave
//      retire in order on gfx9).
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
// [lds_addr + 16*lane, +16).  M0 is written and used in one statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void p4_glds(const char* sbase, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
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

template <int ACT, int NS>
__global__ void __launch_bounds__(256) gemm_nt_p4_kernel(const GemmParams p, const int n_items) {
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
  // The cursor (cur) walks the workgroup's items NS stages ahead of the MFMAs; nxt describes the item after the
  // cursor's.  The hand-over cur <- nxt at the end of an item is a register select (no branch inside the stage
  // loop); nxt is refreshed once per tile, next to the epilogue.  Past the last item the cursor re-reads it: those
  // stages land in ring slots nobody reads any more and are drained before the workgroup ends.
  P4Src cur = src_of(first_item);
  P4Src nxt = src_of(first_item + stride <= last_item ? first_item + stride : last_item);
  int d_k = 0;
  auto dma_piece = [&](int slot, int pc) {
    const uint32_t dst = dma_lds + (uint32_t)slot * (uint32_t)kP4Stage;
    if (pc < 4) p4_glds(cur.a_base, cur.a_off[pc], dst + (uint32_t)pc * 1024u);
    else p4_glds(cur.b_base, cur.b_off[pc - 4], dst + (uint32_t)kP4StageA + (uint32_t)(pc - 4) * 1024u);
  };
  auto dma_advance = [&]() {
    ++d_k;
    const bool sw = (d_k == cur.nk);
    cur.a_base = sw ? nxt.a_base : cur.a_base + 64;
    cur.b_base = sw ? nxt.b_base : cur.b_base + 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cur.a_off[i] = sw ? nxt.a_off[i] : cur.a_off[i];
      cur.b_off[i] = sw ? nxt.b_off[i] : cur.b_off[i];
    }
    cur.nk = sw ? nxt.nk : cur.nk;
    d_k = sw ? 0 : d_k;
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

  auto read_frag = [&](int set, int slot, int which) {  // which: 0..3 = A row block, 4..7 = B column block
    const char* base = smem + slot * kP4Stage;
    if (which < 4) fa[set][which] = *reinterpret_cast<const frag_t*>(base + a_rd[set] + which * 2048);
    else fb[set][which - 4] = *reinterpret_cast<const frag_t*>(base + b_rd[set] + (which - 4) * 2048);
  };

  // ---- prologue: fill the ring ------------------------------------------------------------------------------------
#pragma unroll
  for (int st = 0; st < NS; ++st) {
#pragma unroll
    for (int pc = 0; pc < kP4Pieces; ++pc) dma_piece(st, pc);
    dma_advance();
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP4Pieces * (NS - 1)) : "memory");
  P4_BARRIER();
#pragma unroll
  for (int w = 0; w < 8; ++w) read_frag(0, 0, w);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)

  // ---- main loop: this workgroup's items, each a run of branch-free stages -----------------------------------------
  int slot = 0, slot_next = 1;  // g % NS, (g + 1) % NS for the running stage index g
#pragma unroll 1
  for (int c_item = first_item; c_item < n_items; c_item += stride) {
    const P4Item cit = p4_item(p, c_item, ntiles);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#pragma unroll 1
    for (int k = 0; k < cit.nk; ++k) {
      // phase A: MFMAs of sub-step 0, fragment reads of sub-step 1
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = t >> 2, j = t & 3;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
        if (t < 8) read_frag(1, slot, t);
        P4_FENCE();
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): slot `slot` is no longer read by this wave
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP4Pieces * (NS - 2)) : "memory");  // stage g + 1 has landed
      P4_FENCE();
      P4_BARRIER();
      P4_FENCE();
      // phase B: MFMAs of sub-step 1, fragment reads of (g + 1, sub-step 0), DMA of stage g + NS into slot g % NS
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = t >> 2, j = t & 3;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
        if (t < 8) read_frag(0, slot_next, t);
        else dma_piece(slot, t - 8);
        P4_FENCE();
      }
      dma_advance();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      P4_FENCE();
      slot = slot_next;
      slot_next = (slot_next + 1 == NS) ? 0 : slot_next + 1;
    }

    // tile finished: epilogue straight from the accumulators --------------------------------------------------------
    int e_lane = lane;
    asm volatile("" : "+v"(e_lane));  // opaque: nothing below is hoisted above the stage loop (register pressure)
    const int half = e_lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = cit.m0 + wm * 128 + i * 32 + (e_lane & 31);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // vdst = column group 2 pr (n = 16 pr + 4 half + e), src = group 2 pr + 1 (n = 16 pr + 8 + 4 half + e)
            const uint32_t lo = __builtin_bit_cast(uint32_t, acc[i][j][8 * pr + e]);
            const uint32_t hi = __builtin_bit_cast(uint32_t, acc[i][j][8 * pr + 4 + e]);
            const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
            v[e] = __builtin_bit_cast(float, (uint32_t)r[0]);
            v[4 + e] = __builtin_bit_cast(float, (uint32_t)r[1]);
          }
          const int gn = cit.n0 + wn * 128 + j * 32 + 16 * pr + 8 * half;
          if (gm < p.M && gn < p.N)
            gemm_epilogue8<bf16_t, ACT>(p, cit.kz, gm, gn, v);  // split-K slabs are launched with ACT = none
        }
      }
    }
    // the cursor is now inside item c_item + stride (nk >= NS is a launch condition): describe the one after it
    const int n2 = c_item + 2 * stride;
    nxt = src_of(n2 <= last_item ? n2 : last_item);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int ACT, int NS>
int launch_p4_act(GemmParams& p, int splits, hipStream_t s) {
  constexpr int smem = NS * kP4Stage;
  static bool attr_done = false;
  static int n_cu = 0;
  auto kern = gemm_nt_p4_kernel<ACT, NS>;
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

template <int NS>
int launch_p4_ns(GemmParams& p, int splits, hipStream_t s) {
  switch (p.slabs ? CMB_ACT_NONE : p.act) {
    case CMB_ACT_GELU_ERF: return launch_p4_act<CMB_ACT_GELU_ERF, NS>(p, splits, s);
    case CMB_ACT_GELU_TANH: return launch_p4_act<CMB_ACT_GELU_TANH, NS>(p, splits, s);
    case CMB_ACT_QUICK_GELU: return launch_p4_act<CMB_ACT_QUICK_GELU, NS>(p, splits, s);
    case CMB_ACT_SILU: return launch_p4_act<CMB_ACT_SILU, NS>(p, splits, s);
    default: return launch_p4_act<CMB_ACT_NONE, NS>(p, splits, s);
  }
}

}  // namespace

int launch_gemm_p4_bf16(GemmParams& p, int splits, int ns, hipStream_t s) {
  return ns == 4 ? launch_p4_ns<4>(p, splits, s) : launch_p4_ns<5>(p, splits, s);
}

}  // namespace cmb_gemm_detail
