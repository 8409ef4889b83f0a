// flash_layout.h — LDS image of a 64-row x 128-column bf16 operand tile of the round-5 flash kernels (flash2.hip), free of
// HIP types so that tests/csrc/flash_layout_sim.cpp runs the very same functions on the host under the documented semantics
// of ds_read_b64_tr_b16 (cdna_hip_programming.md T10; model in tests/csrc/gemm_tn_layout_sim.cpp).
//
// A tile is 64 rows (keys, or queries in the dK/dV kernel) of 256 bytes = 16 slots of 16 bytes, NO padding: it is written by
// 16 LDS-DMA pieces of 1 KiB (4 rows each, lane l -> row l >> 4, slot l & 15, lane-linear) straight from global memory — no
// registers, no vector instructions, no ds_write.  Physical slot x of row r holds LOGICAL slot x ^ fl_swz(r) (the XOR is
// applied to the lane's DMA source address).  fl_swz swaps the two low bit pairs of the row number, so that
//   * a ROW-MAJOR fragment read (ds_read_b128: lane j of 32 reads row j, one logical slot) finds the 16 rows of each of the
//     instruction's lane groups {0-3, 12-15, 20-27}, ... in 16 different slots: conflict-free;
//   * a TRANSPOSING read (ds_read_b64_tr_b16: 16 lanes read a [4 rows][16 columns] block, 32 bytes per row) finds the block's
//     four consecutive rows in four different 64-byte bank groups: conflict-free in both 32-lane halves.
// The same image therefore serves an operand both as rows (K in S^T = K Q^T) and as columns (V in O^T = V^T P^T, K^T in
// dQ^T = K^T dS^T, Q^T / dO^T in dK / dV) — the padded row-major + transposed image pair of flash_bwd.hip, their 16-bit
// shuffles and their register prefetch are gone.
#pragma once
#include "gemm_layout.h"   // CMB_HD

#define FL_ROW_BYTES 256
#define FL_TILE_BYTES (64 * FL_ROW_BYTES)
#define FL_PIECE_BYTES 1024

CMB_HD int fl_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// LDS-DMA piece `piece` (0..15) of a tile, lane l: the tile row it fetches, the LOGICAL 16-byte slot of that row, and where
// it lands (lane-linear inside the piece)
CMB_HD int fl_dma_row(int piece, int lane) { return 4 * piece + (lane >> 4); }
CMB_HD int fl_dma_src_slot(int piece, int lane) { return (lane & 15) ^ fl_swz(fl_dma_row(piece, lane)); }
CMB_HD int fl_dma_lds_off(int piece, int lane) { return piece * FL_PIECE_BYTES + lane * 16; }

// Row-major MFMA fragment (A operand of a [32 rows] x [16 k] product, k = columns): lane (j = lane & 31, g = lane >> 5) reads
// 8 consecutive columns 16 ks + 8 g .. + 7 of row row0 + j: byte offset inside the tile
CMB_HD int fl_row_frag_off(int row0, int ks, int lane) {
  const int row = row0 + (lane & 31), slot = 2 * ks + (lane >> 5);
  return row * FL_ROW_BYTES + ((slot ^ fl_swz(row)) << 4);
}

// Transposed MFMA fragment (A operand of a [32 columns] x [16 k] product, k = ROWS of the tile): the lane's 8 k values are rows
// r16 + 4 g + {0..3} (read 0) and r16 + 8 + 4 g + {0..3} (read 1) of column c32 + (lane & 31) — the k <-> register assignment
// of a P / dS operand converted straight out of a 32 x 32 accumulator (element e of lane half g <-> row (e & 3) + 8 (e >> 2)
// + 4 g of the 16).  Returns the address the LANE supplies (it is the address of another lane's data: see the model).
CMB_HD int fl_tr_frag_off(int r16, int c32, int read, int lane) {
  const int q = lane >> 4, i = lane & 15, g = q >> 1;
  const int row = r16 + 8 * read + 4 * g + (i >> 2);
  const int slot = (c32 >> 3) + 2 * (q & 1) + ((i & 3) >> 1);   // logical 16-byte slot of columns c32 + 16 (q & 1) + 4 (i & 3)
  return row * FL_ROW_BYTES + ((slot ^ fl_swz(row)) << 4) + (i & 1) * 8;
}
