// vit_layout.h — LDS image of a 64-key x 64-column bf16 K / V tile of vit_attn_dma_kernel<64> (vit_attn.hip), free of HIP
// types so that tests/csrc/vit_layout_sim.cpp runs the very same functions on the host under the documented semantics of
// ds_read_b128 / ds_read_b64_tr_b16 (MI355X_MICROARCH.md §LDS; model in tests/csrc/gemm_tn_layout_sim.cpp).
//
// head_dim 64: a tile is 64 rows (keys) of 128 bytes = 8 slots of 16 bytes, NO padding, written by 8 LDS-DMA pieces of 1 KiB
// (8 rows each: lane l -> row l >> 3, physical slot l & 7, lane-linear).  Physical slot x of row r holds LOGICAL slot
// x ^ vl_swz(r) (applied to the lane's DMA source address).  vl_swz was found by enumerating every GF(2)-linear map of the row
// bits (49152 of 262144 pass both tests below); the one kept is bits {1, 2} of the row in bits {0, 1} and bit 1 ^ bit 3 in bit 2:
//   * a ROW-MAJOR fragment read (ds_read_b128: lane j of 32 reads row j, one logical slot; serviced in the lane groups
//     {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of each half) finds a group's 16 rows on 16 different 16-byte bank positions
//     (two 128-byte rows share a 256-byte bank row: position = (row & 1) * 8 + physical slot);
//   * a TRANSPOSING read (ds_read_b64_tr_b16: a 32-lane half reads 4 consecutive rows x 4 consecutive logical slots) needs
//     rows r and r + 2 (same bank half) in different 64-byte groups: bit 2 of the swizzle differs between them.
// head_dim 96 (SigLIP's 72, zero-padded by the packer) uses flash_layout.h's 256-byte rows with logical slots 12..15 unused.
#pragma once
#include "gemm_layout.h"   // CMB_HD

#define VL_ROW_BYTES 128
#define VL_TILE_BYTES (64 * VL_ROW_BYTES)
#define VL_PIECE_BYTES 1024

CMB_HD int vl_swz(int row) { return ((row >> 1) & 3) | ((((row >> 1) ^ (row >> 3)) & 1) << 2); }

// LDS-DMA piece `piece` (0..7) of a tile, lane l: the tile row it fetches, the LOGICAL slot of that row, where it lands
CMB_HD int vl_dma_row(int piece, int lane) { return 8 * piece + (lane >> 3); }
CMB_HD int vl_dma_src_slot(int piece, int lane) { return (lane & 7) ^ vl_swz(vl_dma_row(piece, lane)); }
CMB_HD int vl_dma_lds_off(int piece, int lane) { return piece * VL_PIECE_BYTES + lane * 16; }

// Row-major MFMA fragment: lane (j = lane & 31, g = lane >> 5) reads columns 16 ks + 8 g .. + 7 of row row0 + j
CMB_HD int vl_row_frag_off(int row0, int ks, int lane) {
  const int row = row0 + (lane & 31), slot = 2 * ks + (lane >> 5);
  return row * VL_ROW_BYTES + ((slot ^ vl_swz(row)) << 4);
}

// Transposed MFMA fragment (k = ROWS of the tile): the lane's 8 k values are rows r16 + 4 g + {0..3} (read 0) and
// r16 + 8 + 4 g + {0..3} (read 1) of column c32 + (lane & 31) — the k <-> register assignment of a P operand converted
// straight out of a 32 x 32 accumulator.  Returns the address the LANE supplies (another lane's data: see the model).
CMB_HD int vl_tr_frag_off(int r16, int c32, int read, int lane) {
  const int q = lane >> 4, i = lane & 15, g = q >> 1;
  const int row = r16 + 8 * read + 4 * g + (i >> 2);
  const int slot = (c32 >> 3) + 2 * (q & 1) + ((i & 3) >> 1);
  return row * VL_ROW_BYTES + ((slot ^ vl_swz(row)) << 4) + (i & 1) * 8;
}
