// gemm.hip — C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T) for gfx950 (MI355X).
//
// Design:
//   * operands stream HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS stages,
//     one barrier per 128-byte K-step; the next K-step's DMA is issued before the current MFMAs.
//   * LDS tiles are XOR-swizzled (gemm_layout.h) so every ds_read_b128 fragment read is
//     bank-conflict free; the swizzle lives in the per-lane global source address.
//   * v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32 parity path),
//     fp32 accumulation in 64 accumulator registers per wave (2x2 tiles of 32x32).
//   * operands are swapped at the MFMA so a lane owns one output row and 4 consecutive columns per
//     register quad; the tile is staged through LDS once and leaves as 16/32-byte row-contiguous
//     stores with the whole epilogue (bias, activation, LayerScale, residual, fp32 accumulate) fused.
//   * block id -> tile mapping is XCD-aware (8 XCDs, private L2s).
//   * rows of A / C / residual / pre_out go through a 3-level row map so gathers such as the in-LLM
//     slice hidden[:, 91:691].view(B,24,25,H)[:, :, :24] are folded into the loads.
#include <cstdio>
#include <stdlib.h>
#include "gemm_common.h"

using namespace cmb_gemm_detail;

namespace {

template <int BM, int BN>
constexpr int gemm_smem_bytes() {
  return (2 * (BM + BN) * 128) > (BM * (BN + 4) * 4) ? (2 * (BM + BN) * 128) : (BM * (BN + 4) * 4);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) gemm_nt_kernel(const GemmParams p_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmParams p = p_in;
  if (p.batch > 1) {  // batched launch: problem blockIdx.z of p.batch (operand / result bases advance by the batch strides)
    const int64_t bz = blockIdx.z;
    p.A += bz * p.a_bs * (int64_t)sizeof(T);
    p.B += bz * p.b_bs * (int64_t)sizeof(T);
    p.C += bz * p.c_bs * (int64_t)(p.out_f32 ? 4 : sizeof(typename Mfma<T>::out_t));
    if (p.R) p.R += bz * p.c_bs * (int64_t)sizeof(T);   // a batched residual is laid out as C (same batch stride)
  }
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int BK = 128 / (int)sizeof(T);
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  typedef typename Mfma<T>::frag_t frag_t;
  typedef typename Mfma<T>::out_t out_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nblk = p.tiles_m * p.tiles_n;
  const int id = gl_xcd_remap((int)blockIdx.x, nblk);
  int tile_m, tile_n;
  gl_group_tile(id, p.tiles_m, p.tiles_n, 8, &tile_m, &tile_n);  // 64 resident workgroups per XCD = 8 x 8 tiles
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz = blockIdx.y;
  const int kbeg = kz * p.k_per_split;
  const int kend = (kbeg + p.k_per_split < p.K) ? (kbeg + p.k_per_split) : p.K;
  const int nk = (kend > kbeg) ? (kend - kbeg) / BK : 0;

  // per-lane global source pointers of the LDS-DMA pieces this wave issues (row clamped at the edge:
  // out-of-range rows re-read the last valid row and are never stored)
  const char* a_src[A_IT];
  const char* b_src[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int grp = wave + i * NW;
    const int row = gl_dma_row(grp, lane), c = gl_dma_chunk(grp, lane);
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = p.A + (row_off(p.a_map, (uint32_t)gm) + kbeg) * (int64_t)sizeof(T) + c * 16;
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int grp = wave + i * NW;
    const int row = gl_dma_row(grp, lane), c = gl_dma_chunk(grp, lane);
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    b_src[i] = p.B + ((int64_t)gn * p.ldb + kbeg) * (int64_t)sizeof(T) + c * 16;
  }

  // fragment read offsets (bytes inside a stage)
  int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * WTM + t * 32 + gl_frag_row(lane);
    a_off[t] = row * 128;
    a_swz[t] = gl_swz(row);
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * WTN + t * 32 + gl_frag_row(lane);
    b_off[t] = A_BYTES + row * 128;
    b_swz[t] = gl_swz(row);
  }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto stage = [&](int s) {
    char* sa = smem + s * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      glds16(a_src[i], sa + (wave + i * NW) * 1024);
      a_src[i] += 128;
    }
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      glds16(b_src[i], sb + (wave + i * NW) * 1024);
      b_src[i] += 128;
    }
  };

  if (nk > 0) {
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1);  // async: lands while this K-step's MFMAs run
      const char* base = smem + cur * STAGE;
#pragma unroll
      for (int ks = 0; ks < Mfma<T>::KSTEPS; ++ks) {
        frag_t a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = Mfma<T>::load(base + a_off[t], a_swz[t], ks, lane);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = Mfma<T>::load(base + b_off[t], b_swz[t], ks, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mfma<T>::run(b[j], a[i], acc[i][j]);  // swapped: rows=n, cols=m
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32, row stride BN+4) -> row-contiguous global stores ----
  constexpr int CS = BN + 4;
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = wm * WTM + i * 32 + gl_acc_m(lane);
        const int n = wn * WTN + j * 32 + gl_acc_n(4 * q, lane);
        f32x4_t v;
        v[0] = acc[i][j][4 * q + 0];
        v[1] = acc[i][j][4 * q + 1];
        v[2] = acc[i][j][4 * q + 2];
        v[3] = acc[i][j][4 * q + 3];
        *reinterpret_cast<f32x4_t*>(cs + m * CS + n) = v;
      }
  __syncthreads();

  constexpr int GPR = BN / 8;  // 8-column groups per tile row
  constexpr int GROUPS = BM * GPR;
  dispatch_act(p.slabs ? CMB_ACT_NONE : p.act, [&](auto act_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
    for (int grp = tid; grp < GROUPS; grp += NT) {
      const int row = grp / GPR, c8 = grp - row * GPR;
      const int gm = m0 + row, gn = n0 + c8 * 8;
      if (gm >= p.M || gn >= p.N) continue;
      float v[8];
      {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8);
        const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cs + row * CS + c8 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      }
      gemm_epilogue8<out_t, ACT>(p, kz, gm, gn, v);
    }
  });
}

// out = alpha * sum_z slab[z] + beta * out   (fp32 slabs [Z][M][N]; out through the C row map)
template <typename TOut>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* slabs, int Z, int M, int N,
                                                            char* C, RowMap c_map, float alpha,
                                                            float beta) {
  const int64_t groups = (int64_t)M * (N / 8);
  for (int64_t gidx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gidx < groups;
       gidx += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(gidx / (N / 8)), n = (int)(gidx - (int64_t)m * (N / 8)) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int z = 0; z < Z; ++z) {
      float v[8];
      load8f(slabs + ((int64_t)z * M + m) * N + n, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    TOut* cp = reinterpret_cast<TOut*>(C) + row_off(c_map, (uint32_t)m) + n;
    if (beta != 0.0f) {
      float old[8];
      Vec8<TOut>::load(cp, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = alpha * acc[e] + beta * old[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= alpha;
    }
    Vec8<TOut>::store(cp, acc);
  }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_gemm(GemmParams& p, int splits, hipStream_t s) {
  constexpr int smem = gemm_smem_bytes<BM, BN>();
  static CmbAttrOnce attr_once;
  auto kern = gemm_nt_kernel<T, BM, BN, WM, WN>;
  if (const uint32_t attr_bit = attr_once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return CMB_ERR_LAUNCH;
    attr_once.done(attr_bit);
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits, (unsigned)(p.batch > 1 ? p.batch : 1));
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, s, p);
  CMB_CHECK_LAUNCH();
  return CMB_OK;
}


// Tile-configuration choice for bf16.  Cost unit = one "round" of the 256x256 kernel (256 workgroups, one per
// CU).  Measured on full grids the 256x256 / 8-phase kernel is ~1.33x the 128x128 one (tools/bench_kernels.py:
// 0.93-1.23 vs 0.70-0.90 PFLOP/s), and a round of 512 128x128 workgroups (two per CU) covers half the output of a
// 256x256 round, so it costs 2 / 1.33 / 2 ~ 0.667 units; a last round of <= 256 such workgroups (one per CU, no
// co-resident partner) ~0.6 of that.  cmb_gemm_desc.tile_hint / CMB_GEMM_TILE=128|256 override (tests, A-B).
static int tile_override() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CMB_GEMM_TILE");
    v = e ? atoi(e) : 0;
  }
  return v;
}
static bool use_tile256(int M, int N, int splits, int ov) {
  if (ov == 256 || ov == 2560 || ov == 2561 || ov == 2590) return true;
  if (ov == 128) return false;
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256) * splits;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splits;
  const double c256 = (double)((t256 + 255) / 256);
  const long rem = t128 % 512;
  const double c128 = 0.667 * ((double)(t128 / 512) + (rem == 0 ? 0.0 : (rem <= 256 ? 0.6 : 1.0)));
  return c256 <= c128;
}

// The 256 x 256 kernels address a tile's rows as a wave-uniform 64-bit base + a 32-bit per-lane byte offset: the span of
// 256 consecutive (row-mapped) rows of A and of B must fit (upper bound; negative strides never qualify).
static bool tile_span_fits_u32(const RowMap& m, int64_t ldb) {
  if (m.s0 < 0 || m.s1 < 0 || m.s2 < 0 || ldb < 0) return false;
  double span = 256.0 * (double)m.s2;
  if (m.n1) span += (256.0 / (double)m.n2 + 1.0) * (double)m.s1 + (256.0 / (double)m.n1 + 1.0) * (double)m.s0;
  return 2.0 * span + 256.0 < 4.0e9 && 2.0 * 256.0 * (double)ldb + 256.0 < 4.0e9;
}

// gemm_p5.hip stages two 64-deep tiles ahead across item boundaries: every item (tile x K slice) must be at least two
// tiles long.
static bool p5_ok(const GemmParams& p, int splits) {
  const int last = p.K - (splits - 1) * p.k_per_split;
  return last >= 128 && p.k_per_split >= 128;
}

static thread_local int g_last_kernel = 0;  // cmb_gemm_last_kernel()

// Per-shape dispatch policy (round 3, cmb_gemm_policy_set): which bf16 kernel a (M, N, K, act) problem takes when the
// caller gives no tile_hint.  Filled by the host's start-up calibration (cambrian_amd/ops.py::calibrate_gemm_dispatch
// times the candidates on THIS device — the 4-wave kernel's lead over the 8-wave one varies from box to box) and read
// by every launch: a handful of entries, linear scan.  Written only between steps (no launches in flight on other
// threads); the kernels it selects between are bit-identical in their results (tests/test_gemm256_gpu.py).
struct PolicyEntry { int64_t M, N, K; int act, kernel; };
constexpr int kMaxPolicy = 64;
static PolicyEntry g_policy[kMaxPolicy];
static int g_npolicy = 0;
static int policy_lookup(int64_t M, int64_t N, int64_t K, int act) {
  for (int i = 0; i < g_npolicy; ++i)
    if (g_policy[i].M == M && g_policy[i].N == N && g_policy[i].K == K && g_policy[i].act == act) return g_policy[i].kernel;
  return 0;
}

// Where the register-buffered 4-wave kernel is the default 256 x 256 kernel: whole 128-column halves (round 4: in a last
// column tile with only its lower half in range the upper waves skip the epilogue; a half that is itself ragged leaves
// through the generic epilogue and drains the DMA pipeline — N = 1152 used to run 20-40 % behind the 8-wave kernel for that
// reason), no pre-activation copy (generic epilogue again) and at least 64 tiles.  Rounds 2-3 required more than one round of
// items per CU (the kernel's gain is the overlap ACROSS items, and a single round with an activation epilogue was 25 %
// faster on the 8-wave kernel); with round 4's epilogue (compile-time bias / LayerScale / residual variants, packed math) the
// start-up calibration found it ahead on all 20 hottest shapes of the step including the single-round ones (13824 x 1024 x
// 1024: 33.0 vs 36.4 us; profiles/r04_lab.md).
static bool p5_default(const GemmParams& p, int splits) {
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * splits;
  static long min_tiles = -1;  // CMB_GEMM_P5_MIN_TILES: experiments only
  if (min_tiles < 0) {
    const char* e = getenv("CMB_GEMM_P5_MIN_TILES");
    min_tiles = e ? atol(e) : 64;
  }
  return p.N % 128 == 0 && !p.P && tiles >= min_tiles;
}

// Which bf16 kernel a problem takes.  tile_hint / CMB_GEMM_TILE: 0 = the per-shape policy if the host calibrated one for
// this (M, N, K, act), else the cost model (128x128 tile, or a 256x256 tile: the 4-wave register-buffered kernel
// gemm_nt_p5_kernel where p5_default() says, else the 8-wave kernel) | 128 | 256 (as the cost model's 256 branch) |
// 2560 / 2561 (8-wave kernel, schedule 0 / 1) | 2590 (gemm_nt_p5_kernel).
// Measured on the path's shapes (profiles/r02_gemm_lab.md): p5 is 3-11 % ahead of the 8-wave kernel when N is a
// multiple of 256 and up to 40 % behind when it is not (N = 384, 1152).
static int choose_bf16_kernel(const GemmParams& p, int splits, int hint, int* sched) {
  int ov = hint ? hint : tile_override();
  if (!ov && !p.slabs) ov = policy_lookup(p.M, p.N, p.K, p.act);
  *sched = ov == 2561 ? 1 : 0;
  if (!use_tile256(p.M, p.N, splits, ov) || !tile_span_fits_u32(p.a_map, p.ldb)) return 128;
  if ((ov == 2590 || ((ov == 0 || ov == 256) && p5_default(p, splits))) && p5_ok(p, splits)) return 2590;
  return 256;
}

// Tail split (round 3).  A 256 x 256 grid of T tiles takes ceil(T / CUs) rounds; when T is a little more than a whole
// number of rounds the last round runs a handful of tiles on an otherwise idle chip (DINOv2's 11680 x 1536 GEMMs: 46 x 6
// = 276 tiles = 2 rounds for 1.08 rounds of work; SigLIP's 11664 x 4352: 782 tiles = 4 rounds for 3.05).  Such a problem
// is launched as two row ranges: the first m1 row tiles (as many whole rounds as fit) on the 256-tile kernel the cost
// model picks for them, the remaining rows on the 128 x 128 kernel (two workgroups per CU, any epilogue), whose partial
// round is much shorter than a 256-tile round.  Cost model in 256-tile rounds: the 128-tile kernel runs ~1.6x longer per
// FLOP (profiles/r02_gemm_lab.md), 0.1 round for the extra launch; taken when it saves more than 7 %.  Row maps must be
// linear across the cut (identity, or the cut a multiple of the outer period).  Returns the rows of the first range or 0.
static int device_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    n -= n % 8;
    if (n <= 0) n = 256;
  }
  return n;
}
static bool tail_split_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CMB_GEMM_NO_TAIL_SPLIT");
    v = (e && atoi(e)) ? 0 : 1;
  }
  return v != 0;
}
static bool map_linear_at(const RowMap& m, int64_t row) { return m.n1 == 0 || row % m.n1 == 0; }
static int tail_split_rows_mnk(int64_t M, int64_t N) {
  if (!tail_split_enabled() || N % 128 != 0) return 0;
  const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256, T = tm * tn;
  const int64_t ncu = device_cus();
  if (T <= ncu) return 0;
  const int64_t m1 = ((T / ncu) * ncu) / tn;   // row tiles that fill whole rounds
  if (m1 <= 0 || m1 >= tm) return 0;
  const int64_t tail = T - m1 * tn;
  const double now = (double)((T + ncu - 1) / ncu);
  const double hyb = (double)((m1 * tn + ncu - 1) / ncu) + 1.6 * (double)tail / (double)ncu + 0.1;
  return hyb < 0.93 * now ? (int)(m1 * 256) : 0;
}
static int tail_split_rows(const GemmParams& p, int splits, int hint) {
  if (hint || tile_override() || splits > 1 || p.slabs || p.a_scale || p.b_scale) return 0;
  const int m1 = tail_split_rows_mnk(p.M, p.N);
  if (!m1) return 0;
  if (!map_linear_at(p.a_map, m1) || !map_linear_at(p.c_map, m1) || (p.R && !map_linear_at(p.r_map, m1)) ||
      (p.P && !map_linear_at(p.p_map, m1)))
    return 0;
  return m1;
}

// descriptor -> parameter block: the checks and the split-K geometry of one cmb_gemm call (shared by cmb_gemm and cmb_gemm_pair)
template <typename T>
int gemm_params_from_desc(const cmb_gemm_desc* d, GemmParams& p, int& splits) {
  constexpr int BK = 128 / (int)sizeof(T);
  if (d->K % BK != 0 || d->N % 8 != 0) return CMB_ERR_SHAPE;
  p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
  p.A = (const char*)d->A; p.a_map = make_rowmap(d->a_map);
  p.B = (const char*)d->B; p.ldb = d->ldb;
  p.C = (char*)d->C; p.c_map = make_rowmap(d->c_map);
  p.bias = d->bias; p.colscale = d->colscale;
  p.R = (const char*)d->residual; p.r_map = make_rowmap(d->r_map);
  p.P = (char*)d->pre_out; p.p_map = make_rowmap(d->p_map);
  p.act = d->act; p.alpha = d->alpha; p.beta = d->beta;
  p.out_f32 = (d->out_dtype == CMB_F32);
  p.slabs = nullptr;
  p.k_per_split = p.K;
  p.a_scale = nullptr; p.b_scale = nullptr;
  p.batch = d->batch > 1 ? d->batch : 1;
  p.a_bs = d->a_batch_stride; p.b_bs = d->b_batch_stride; p.c_bs = d->c_batch_stride;
  p.slab_rows = p.M;
  p.row_mean = d->row_mean; p.row_rstd = d->row_rstd; p.col_sum = d->col_sum;
  splits = d->split_k > 1 ? d->split_k : 1;
  if (d->row_mean && (!d->row_rstd || !d->col_sum || !d->bias || d->alpha != 1.0f || splits > 1 || p.batch > 1 || sizeof(T) != 2 ||
                      d->pre_out))
    return CMB_ERR_BAD_ARG;   // the folded-LayerNorm epilogue: bf16 operands, bias (b'), one launch over the whole K
  if (d->act == CMB_ACT_SWIGLU_PAIRS &&
      (splits > 1 || d->colscale || d->residual || d->pre_out || d->out_dtype != d->dtype || d->N % 16 != 0 || p.batch > 1 ||
       sizeof(T) == 1))
    return CMB_ERR_BAD_ARG;   // the gated epilogue writes an N / 2 wide C of the operand dtype and nothing else
  if (p.batch > 1) {
    // batched problems: plain epilogue (alpha / activation / out dtype; a residual laid out as C: its batch stride is C's), no
    // split-K, 16-byte aligned strides
    if (splits > 1 || d->bias || d->colscale || d->pre_out || sizeof(T) == 1) return CMB_ERR_BAD_ARG;
    if (d->residual && (d->out_dtype != d->dtype)) return CMB_ERR_BAD_ARG;
    if ((p.a_bs * (int64_t)sizeof(T)) % 16 || (p.b_bs * (int64_t)sizeof(T)) % 16 || (p.c_bs * 2) % 16) return CMB_ERR_ALIGNMENT;
  }
  if constexpr (sizeof(T) == 1) {
    if (splits > 1) return CMB_ERR_BAD_ARG;
    p.a_scale = d->a_scale; p.b_scale = d->b_scale;
  }
  // every row base and leading dimension must keep 16-byte chunks aligned
  const int64_t es = sizeof(T);
  if (!cmb_aligned16(d->A) || !cmb_aligned16(d->B) || (d->ldb * es) % 16 != 0 ||
      (d->a_map.s2 * es) % 16 != 0 || (d->a_map.n1 && ((d->a_map.s0 * es) % 16 || (d->a_map.s1 * es) % 16)))
    return CMB_ERR_ALIGNMENT;
  if (splits > 1) {
    int ksteps = p.K / BK;
    if (splits > ksteps) splits = ksteps;
    int per = (ksteps + splits - 1) / splits;
    splits = (ksteps + per - 1) / per;
    p.k_per_split = per * BK;
    if (splits > 1) {
      const int64_t need = (int64_t)splits * p.M * p.N * 4;
      if (!d->workspace || d->workspace_bytes < need) return CMB_ERR_WORKSPACE;
      p.slabs = (float*)d->workspace;
    } else {
      p.k_per_split = p.K;
    }
  }
  return CMB_OK;
}

template <typename T>
int gemm_dispatch(const cmb_gemm_desc* d, hipStream_t s) {
  GemmParams p;
  int splits = 1;
  {
    const int prc = gemm_params_from_desc<T>(d, p, splits);
    if (prc != CMB_OK) return prc;
  }
  int rc;
  if constexpr (sizeof(T) == 2) {
    int sched = 0;
    const int m1 = p.batch > 1 ? 0 : tail_split_rows(p, splits, d->tile_hint);
    if (p.batch > 1) {
      if (gemm_k64_eligible(p)) g_last_kernel = 64, rc = launch_gemm_k64_batched(p, s);   // (CMB_GEMM_K64=0: the tile kernel)
      else g_last_kernel = 128, rc = launch_gemm<T, 128, 128, 2, 2>(p, 1, s);
    } else if (!d->tile_hint && gemm_small_m_eligible(p, splits)) {
      g_last_kernel = 32, rc = launch_gemm_small_m(p, s);
    } else if (m1) {
      GemmParams head = p, tail = p;
      head.M = m1;
      tail.M = p.M - m1;
      tail.A += row_off(p.a_map, (uint32_t)m1) * 2;
      tail.C += row_off(p.c_map, (uint32_t)m1) * (p.out_f32 ? 4 : 2);
      if (p.R) tail.R += row_off(p.r_map, (uint32_t)m1) * 2;
      if (p.P) tail.P += row_off(p.p_map, (uint32_t)m1) * 2;
      if (p.row_mean) tail.row_mean += m1, tail.row_rstd += m1;
      const int kern = choose_bf16_kernel(head, 1, 256, &sched);   // (256: the 256-tile branch of the cost model, no policy)
      g_last_kernel = kern;
      if (kern == 128) {
        // the cost model refused the 256-tile kernels for the head (tile_span_fits_u32: their 32-bit per-lane offsets
        // cannot span this row stride): no split, the whole problem on the 128-tile kernel as before the tail split
        rc = launch_gemm<T, 128, 128, 2, 2>(p, 1, s);
      } else {
        rc = kern == 2590 ? launch_gemm_p5_bf16(head, 1, s) : launch_gemm256_bf16(head, 1, sched, s);
        if (rc == CMB_OK) rc = launch_gemm<T, 128, 128, 2, 2>(tail, 1, s);
      }
    } else {
      const int kern = choose_bf16_kernel(p, splits, d->tile_hint, &sched);
      g_last_kernel = kern;
      if (kern == 128) rc = launch_gemm<T, 128, 128, 2, 2>(p, splits, s);
      else if (kern == 2590) rc = launch_gemm_p5_bf16(p, splits, s);
      else rc = launch_gemm256_bf16(p, splits, sched, s);
    }
  } else {
    g_last_kernel = 128, rc = launch_gemm<T, 128, 128, 2, 2>(p, splits, s);
  }
  if (rc != CMB_OK) return rc;
  if (p.slabs) {
    const int64_t groups = (int64_t)p.M * (p.N / 8);
    int blocks = (int)((groups + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (p.out_f32)
      hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, s, p.slabs, splits,
                         p.M, p.N, p.C, p.c_map, p.alpha, p.beta);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<typename Mfma<T>::out_t>, dim3(blocks), dim3(256), 0, s, p.slabs, splits,
                         p.M, p.N, p.C, p.c_map, p.alpha, p.beta);
    CMB_CHECK_LAUNCH();
  }
  return CMB_OK;
}

// C[M,N] = alpha * At[K,M]^T Bt[K,N] (+ beta C): descriptor checks, split-K slabs and their reduction around gemm_tn.hip
int gemm_tn_dispatch(const cmb_gemm_desc* d, hipStream_t s) {
  if (d->dtype != CMB_BF16 || (d->out_dtype != CMB_F32 && d->out_dtype != CMB_BF16)) return CMB_ERR_BAD_ARG;
  if (d->bias || d->colscale || d->residual || d->pre_out || d->act != CMB_ACT_NONE || d->a_map.n1 != 0) return CMB_ERR_BAD_ARG;
  if (d->M % 8 != 0 || d->N % 8 != 0) return CMB_ERR_SHAPE;
  const int64_t lda = d->a_map.s2;
  if (!cmb_aligned16(d->A) || !cmb_aligned16(d->B) || (lda * 2) % 16 != 0 || (d->ldb * 2) % 16 != 0 || lda < d->M || d->ldb < d->N)
    return CMB_ERR_ALIGNMENT;
  GemmParams p;
  p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
  p.A = (const char*)d->A; p.a_map = make_rowmap(d->a_map);
  p.B = (const char*)d->B; p.ldb = d->ldb;
  p.C = (char*)d->C; p.c_map = make_rowmap(d->c_map);
  p.bias = nullptr; p.colscale = nullptr;
  p.R = nullptr; p.r_map = make_rowmap(d->r_map);
  p.P = nullptr; p.p_map = make_rowmap(d->p_map);
  p.act = CMB_ACT_NONE; p.alpha = d->alpha; p.beta = d->beta;
  p.out_f32 = (d->out_dtype == CMB_F32);
  p.slabs = nullptr;
  p.k_per_split = p.K;
  p.a_scale = nullptr; p.b_scale = nullptr;
  p.batch = d->batch > 1 ? d->batch : 1;
  p.a_bs = d->a_batch_stride; p.b_bs = d->b_batch_stride; p.c_bs = d->c_batch_stride;
  p.slab_rows = p.M;
  p.row_mean = p.row_rstd = p.col_sum = nullptr;
  if (d->row_mean) return CMB_ERR_BAD_ARG;
  int splits = d->split_k > 1 ? d->split_k : 1;
  if (p.batch > 1) {
    if ((p.a_bs * 2) % 16 || (p.b_bs * 2) % 16 || (p.c_bs * 2) % 16) return CMB_ERR_ALIGNMENT;
    // split-K of a batch: the slabs are [split][batch * M][N], reduced as ONE matrix — the results must be contiguous
    if (splits > 1 && (d->c_map.n1 != 0 || d->c_map.s2 != d->N || p.c_bs != (int64_t)p.M * p.N)) return CMB_ERR_BAD_ARG;
  }
  if (splits > 1) {
    const int ksteps = (p.K + 63) / 64;
    if (splits > ksteps) splits = ksteps;
    const int per = (ksteps + splits - 1) / splits;
    splits = (ksteps + per - 1) / per;
    if (splits > 1) {
      p.k_per_split = per * 64;
      p.slab_rows = p.batch * p.M;
      const int64_t need = (int64_t)splits * p.slab_rows * p.N * 4;
      if (!d->workspace || d->workspace_bytes < need) return CMB_ERR_WORKSPACE;
      p.slabs = (float*)d->workspace;
    }
  }
  g_last_kernel = 1281;
  const int rc = launch_gemm_tn_bf16(p, splits, s);
  if (rc != CMB_OK) return rc;
  if (p.slabs) {
    p.M = p.slab_rows;   // (a batch's results are one contiguous [batch * M, N] matrix)
    const int64_t groups = (int64_t)p.M * (p.N / 8);
    int blocks = (int)((groups + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (p.out_f32)
      hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, s, p.slabs, splits, p.M, p.N, p.C, p.c_map,
                         p.alpha, p.beta);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, p.slabs, splits, p.M, p.N, p.C, p.c_map,
                         p.alpha, p.beta);
    CMB_CHECK_LAUNCH();
  }
  return CMB_OK;
}

}  // namespace

extern "C" int cmb_gemm_tile(int dtype, int64_t M, int64_t N, int32_t split_k, int32_t tile_hint) {
  if (dtype != CMB_BF16) return 128;
  return use_tile256((int)M, (int)N, split_k > 1 ? split_k : 1, tile_hint ? tile_hint : tile_override()) ? 256 : 128;
}

extern "C" int cmb_gemm_policy_set(int64_t M, int64_t N, int64_t K, int32_t act, int32_t kernel) {
  if (kernel != 0 && kernel != 128 && kernel != 2560 && kernel != 2590) return CMB_ERR_BAD_ARG;
  for (int i = 0; i < g_npolicy; ++i)
    if (g_policy[i].M == M && g_policy[i].N == N && g_policy[i].K == K && g_policy[i].act == act) {
      if (kernel) { g_policy[i].kernel = kernel; return CMB_OK; }
      g_policy[i] = g_policy[--g_npolicy];
      return CMB_OK;
    }
  if (!kernel) return CMB_OK;
  if (g_npolicy == kMaxPolicy) return CMB_ERR_WORKSPACE;
  g_policy[g_npolicy++] = PolicyEntry{M, N, K, act, kernel};
  return CMB_OK;
}

extern "C" int cmb_gemm_policy_clear(void) {
  g_npolicy = 0;
  return CMB_OK;
}

extern "C" int cmb_gemm_last_kernel(void) { return g_last_kernel; }
extern "C" int cmb_gemm(const cmb_gemm_desc* d, void* stream);

extern "C" int64_t cmb_gemm_tail_rows(int64_t M, int64_t N) { return tail_split_rows_mnk(M, N); }

extern "C" int cmb_gemm_tn(const cmb_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return CMB_ERR_BAD_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K < 0) return CMB_ERR_BAD_ARG;
  return gemm_tn_dispatch(d, (hipStream_t)stream);
}

static int g_last_pair = 0;   // cmb_gemm_pair: did the last call take the one-launch path?

// Two independent bf16 GEMMs.  One launch of the persistent 256 x 256 kernel with the workgroups split between the problems
// (gemm_p5.hip, P5Args) when both would take that kernel on their own with their whole K, the same activation template and no
// tail split, and the round arithmetic says the pair saves at least 4 % (DINOv2's and SigLIP's 1.62- / 1.35-round linears
// side by side: 3.0 + 2.9 rounds on 138 + 118 workgroups instead of 2 + 2 on 256); otherwise exactly the two cmb_gemm calls.
// Results are bit-identical either way (same kernel, same item arithmetic).  CMB_GEMM_PAIR=0 forces the two calls (A/B runs).
extern "C" int cmb_gemm_pair(const cmb_gemm_desc* d0, const cmb_gemm_desc* d1, void* stream) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("CMB_GEMM_PAIR");
    enabled = (e && atoi(e) == 0) ? 0 : 1;
  }
  g_last_pair = 0;
  hipStream_t s = (hipStream_t)stream;
  auto simple = [](const cmb_gemm_desc* d) {
    return d && d->A && d->B && d->C && d->M > 0 && d->N > 0 && d->K > 0 && d->dtype == CMB_BF16 && d->out_dtype == CMB_BF16 &&
           d->split_k <= 1 && d->batch <= 1 && !d->pre_out && !d->tile_hint && !d->row_mean;
  };
  static int debug = -1;
  if (debug < 0) debug = getenv("CMB_GEMM_PAIR_DEBUG") ? 1 : 0;
  if (enabled && simple(d0) && simple(d1) && d0->act == d1->act && gemm_p5_pair_act_ok(d0->act)) {
    GemmParams p0, p1;
    int s0 = 1, s1 = 1;
    const int r0 = gemm_params_from_desc<bf16_t>(d0, p0, s0), r1 = gemm_params_from_desc<bf16_t>(d1, p1, s1);
    if (r0 == CMB_OK && r1 == CMB_OK && s0 == 1 && s1 == 1) {
      // legal on the persistent kernel (whatever the single-launch cost model would pick: a tail split or the 128-tile kernel
      // are answers to the same partly filled rounds the pair fills)
      auto legal = [](const GemmParams& p) {
        const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        return p.N % 128 == 0 && !p.P && tiles >= 64 && p5_ok(p, 1) && tile_span_fits_u32(p.a_map, p.ldb) && !tile_override();
      };
      const int t0 = 0, t1 = 0;
      const int k0 = legal(p0) ? 2590 : 0, k1 = legal(p1) ? 2590 : 0;
      const double gain = gemm_p5_pair_gain(p0, p1, device_cus());
      if (debug) fprintf(stderr, "cmb_gemm_pair: legal %d %d gain %.3f\n", k0, k1, gain);
      if (!t0 && !t1 && k0 == 2590 && k1 == 2590 && gain >= 0.04) {
        g_last_kernel = 2590;
        const int rc = launch_gemm_p5_bf16(p0, 1, s, &p1);
        if (rc == CMB_OK) g_last_pair = 1;
        return rc;
      }
    } else if (debug) {
      fprintf(stderr, "cmb_gemm_pair: params rc %d %d splits %d %d\n", r0, r1, s0, s1);
    }
  } else if (debug) {
    fprintf(stderr, "cmb_gemm_pair: not simple (enabled %d, %d %d, act %d %d)\n", enabled, (int)simple(d0), (int)simple(d1),
            d0 ? d0->act : -1, d1 ? d1->act : -1);
  }
  const int rc = cmb_gemm(d0, stream);
  return rc != CMB_OK ? rc : cmb_gemm(d1, stream);
}

extern "C" int cmb_gemm_pair_last(void) { return g_last_pair; }

extern "C" int cmb_gemm(const cmb_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return CMB_ERR_BAD_ARG;
  if (d->M < 0 || d->N <= 0 || d->K <= 0) return CMB_ERR_BAD_ARG;
  if (d->M == 0) return CMB_OK;
  if (d->dtype == CMB_F32 && d->out_dtype != CMB_F32) return CMB_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CMB_BF16) return gemm_dispatch<bf16_t>(d, s);
  if (d->dtype == CMB_F32) return gemm_dispatch<float>(d, s);
  if (d->dtype == CMB_FP8_E4M3) return gemm_dispatch<fp8e4m3_t>(d, s);
  return CMB_ERR_BAD_ARG;
}
