// gemm_layout.h — index arithmetic of the LDS-staged MFMA GEMM, kept free of HIP types so the
// same functions are exercised on the host by tests/test_gemm_layout_sim.py (a CPU simulation of
// the LDS-DMA staging, the fragment reads, the assumed MFMA operand layout and the epilogue).
//
// Geometry (bytes, independent of the element type):
//   an operand tile is R rows x 128 bytes (64 bf16 or 32 fp32 along K) = 8 chunks of 16 B per row;
//   chunk c of row r is stored at LDS byte  r*128 + ((c ^ ((r >> 1) & 7)) << 4).
// Why this swizzle: ds_read_b128 is serviced in 16-lane groups over a 256-byte bank row (two tile
// rows); lanes of a group read 16 different rows at the same logical chunk, and
// (r & 1, (r >> 1) & 7) is distinct for the rows of every group, so the read is conflict-free
// (MI355X_MICROARCH.md §LDS).
// Staging uses global_load_lds_dwordx4: a wave-instruction writes 64 lanes x 16 B = 1 KiB
// lane-linearly (= 8 tile rows); the swizzle is therefore applied to each lane's global SOURCE
// address and undone by the same XOR on the fragment read (cdna_hip_programming.md §5.4 rule 21).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CMB_HD __host__ __device__ __forceinline__
#else
#define CMB_HD static inline
#endif

#define CMB_TILE_ROW_BYTES 128
#define CMB_CHUNK_BYTES 16
#define CMB_ROWS_PER_DMA 8  // rows covered by one wave-wide 1 KiB LDS-DMA

// swizzle key of a tile row
CMB_HD int gl_swz(int row) { return (row >> 1) & 7; }

// LDS byte offset (inside an operand tile) of logical 16-B chunk `chunk` of row `row`.
CMB_HD int gl_lds_off(int row, int chunk) {
  return row * CMB_TILE_ROW_BYTES + ((chunk ^ gl_swz(row)) << 4);
}

// For LDS-DMA row-group g (8 rows) and lane l: which tile row / which logical chunk must the lane
// fetch from global memory so that, written at byte g*1024 + l*16, the tile obeys gl_lds_off().
CMB_HD int gl_dma_row(int g, int lane) { return g * CMB_ROWS_PER_DMA + (lane >> 3); }
CMB_HD int gl_dma_chunk(int g, int lane) { return (lane & 7) ^ gl_swz(gl_dma_row(g, lane)); }

// MFMA 32x32 operand fragment: lane l supplies row (l & 31) of its 32-row sub-tile and the 16-B
// chunk 2*ks + (l >> 5) of the 128-B K-step (ks = 0..3).  A and B use the same map, so whatever
// k-permutation the hardware applies inside a chunk pair is applied to both operands alike.
CMB_HD int gl_frag_row(int lane) { return lane & 31; }
CMB_HD int gl_frag_chunk(int ks, int lane) { return 2 * ks + (lane >> 5); }

// MFMA 32x32 accumulator: register reg (0..15) of lane l holds D[i][j] with
//   j = l & 31,  i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)   (cdna_hip_programming.md §3).
// The kernel issues mfma(Bfrag, Afrag): D[i][j] = sum_k Btile[i][k] * Atile[j][k], i.e. i walks N
// and j walks M, so a lane owns ONE output row m and 4 consecutive columns n per register quad.
CMB_HD int gl_acc_m(int lane) { return lane & 31; }
CMB_HD int gl_acc_n(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Grouped rasterisation of a linear tile id: tiles are walked column by column inside bands of `group_m` tile
// rows, so the ~32 (64) workgroups resident on one XCD at a time form a compact group_m x (32/group_m) patch of the
// output and share A row panels / B column panels in that XCD's 4 MiB L2 instead of streaming one whole row of
// tiles (which re-fetches all of B for every row panel).
CMB_HD void gl_group_tile(int id, int tiles_m, int tiles_n, int group_m, int* tm, int* tn) {
  const int width = group_m * tiles_n;
  const int group = id / width, rem = id - group * width;
  const int first = group * group_m;
  const int gsize = (tiles_m - first) < group_m ? (tiles_m - first) : group_m;
  *tm = first + rem % gsize;
  *tn = rem / gsize;
}

// XCD-aware bijective remap of the linear block id (cdna_hip_programming.md §5 template): hardware
// places block b on XCD b % 8; give every XCD a contiguous range of tiles so neighbouring tiles
// (same A row panel) share that XCD's L2.
CMB_HD int gl_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}
