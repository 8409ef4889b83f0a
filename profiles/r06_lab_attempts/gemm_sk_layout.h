// gemm_sk_layout.h — the stream-K tail schedule of gemm_nt_p5_kernel (gemm_p5.hip), free of HIP types so that
// tests/csrc/gemm_sk_sim.cpp walks the very functions the kernel and its launch code use.
//
// A launch of T output tiles (256 x 256) on G persistent workgroups takes ceil(T / G) rounds.  The last `tiles` tiles — the
// partial round, plus one whole round when the partial one is under half full — are cut into 64-deep K tiles, `tiles * nt`
// units, and workgroup g takes the contiguous unit range [sk_bound(g), sk_bound(g + 1)): parts of at most three tiles.
// Unit u = K tile (u % nt) of region tile (u / nt).  A workgroup walks its range from the HIGHEST tile down:
//   * the piece in the highest tile starts at that tile's K tile 0 unless the range lies inside one tile; if it stops
//     short of the tile's last K tile ("part"), its accumulators go to workspace slot g;
//   * every other piece reaches its tile's last K tile: that workgroup OWNS the tile, adds the slots of the lower workgroups
//     c whose ranges end strictly inside the tile (sk_bound(c + 1) in (x * nt, (x + 1) * nt)) and runs the fused epilogue.
// So a workgroup only ever waits for lower-numbered workgroups, which ran the awaited piece FIRST in their lists.
// Boundaries are snapped so that no piece is a single K tile (the kernel's cursor hand-over needs two): sk_bound % nt is never
// 1 or nt - 1; with units / G >= 4 and nt >= 4 every range is non-empty and every piece has at least two K tiles.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CMB_SK_HD __host__ __device__ __forceinline__
#else
#define CMB_SK_HD static inline
#endif

CMB_SK_HD int sk_bound(int g, int G, int units, int nt) {
  int b = (int)(((uint32_t)g * (uint32_t)units) / (uint32_t)G);   // (g * units < 2^31: launch condition, sk_plan_tiles)
  const int r = b % nt;
  if (r == 1) b -= 1;
  else if (r == nt - 1) b += 1;
  return b;
}

// number of pieces of workgroup g and piece `idx` (0 = highest tile): region tile x, K tiles [kt0, kt0 + nkt), part flag
CMB_SK_HD int sk_pieces(int g, int G, int tiles, int nt) {
  const int b0 = sk_bound(g, G, tiles * nt, nt), b1 = sk_bound(g + 1, G, tiles * nt, nt);
  return b1 > b0 ? (b1 - 1) / nt - b0 / nt + 1 : 0;
}
CMB_SK_HD void sk_piece(int g, int G, int tiles, int nt, int idx, int* x, int* kt0, int* nkt, int* part) {
  const int b0 = sk_bound(g, G, tiles * nt, nt), b1 = sk_bound(g + 1, G, tiles * nt, nt);
  const int xx = (b1 - 1) / nt - idx;
  const int u0 = xx * nt, lo = b0 > u0 ? b0 : u0, hi = b1 < u0 + nt ? b1 : u0 + nt;
  *x = xx;
  *kt0 = lo - u0;
  *nkt = hi - lo;
  *part = hi < u0 + nt;
}

// The launch-side plan: how many of the T tiles form the stream-K region (0 = whole-tile rounds only).  Cost model in units
// of one K tile (~1.5 us): an item costs nt + e with e the epilogue (4: plain, 7: activation or residual), the hand-over of
// a split tile ~4.5 (slot write + drain on one side, ~5 us of slot reads on the other).  mode: 0 = never, 1 = when the model
// saves more than 5 %, 2 = whenever the schedule is legal.
CMB_SK_HD int sk_plan_tiles(int T, int G, int nt, int heavy_epilogue, int mode) {
  if (mode == 0 || nt < 4 || T <= G / 2 || (long long)(G + 1) * nt * (2LL * G) >= (1LL << 31)) return 0;
  const int rem = T % G;
  if (rem == 0) return 0;
  int tiles = rem;
  if (rem < G / 2 || (long long)rem * nt < 4LL * G) tiles = (T > G) ? rem + G : 0;
  if (tiles == 0 || (long long)tiles * nt < 4LL * G) return 0;
  if (mode >= 2) return tiles;
  const double e = heavy_epilogue ? 7.0 : 4.0, fix = 4.5;
  const double rounds_dp = (double)((T + G - 1) / G);
  const double cost_dp = rounds_dp * (nt + e);
  const double full = (double)((T - tiles) / G);
  const double share = (double)tiles * nt / G;
  const double owners = (double)((tiles + G - 1) / G);
  const double cost_sk = full * (nt + e) + share + owners * e + fix;
  return cost_sk < 0.95 * cost_dp ? tiles : 0;
}
