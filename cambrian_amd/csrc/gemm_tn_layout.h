// gemm_tn_layout.h — index arithmetic of gemm_tn.hip's operand stage (LDS-DMA placement, transposing fragment reads), free
// of HIP types so that tests/csrc/gemm_tn_layout_sim.cpp runs the very same functions on the host under the documented
// semantics of ds_read_b64_tr_b16 (cdna_hip_programming.md §LDS, T10).
//
// An operand stage is 64 contraction rows (k) x 128 columns of bf16 = 64 rows of 256 bytes = 16 slots of 16 bytes per row,
// written by 16 LDS-DMA pieces of 1 KiB (4 rows each, lane l -> row l >> 4, slot l & 15, lane-linear).  Slot x of row k
// holds SOURCE slot x ^ 4 (k & 3): the four rows of a [4 k][16 columns] block of a transposing read lie 256 bytes apart
// (the same banks) and the rotation moves them to four different 64-byte bank groups.
#pragma once
#include "gemm_layout.h"   // CMB_HD

#define CMB_TN_ROW_BYTES 256
#define CMB_TN_PIECE_BYTES 1024

// LDS-DMA piece `piece` (0..15), lane l: contraction row inside the stage and the 8-column source slot the lane fetches
CMB_HD int tn_dma_row(int piece, int lane) { return 4 * piece + (lane >> 4); }
CMB_HD int tn_dma_src_slot(int lane) { return (lane & 15) ^ (4 * (lane >> 4)); }
CMB_HD int tn_dma_lds_off(int piece, int lane) { return piece * CMB_TN_PIECE_BYTES + lane * 16; }

// Transposing fragment read.  Lane (q = lane >> 4, i = lane & 15) supplies the address of row i >> 2, 8-byte piece i & 3
// of a [4 k][16 columns] block and receives column i of it (4 consecutive k).  An MFMA fragment is 32 columns = the
// 16-column subtiles sub0 + (q & 1); lanes 0-31 take the 4-row piece tn_frag_piece(s, r), lanes 32-63 the next one.
// Returns the byte offset inside the operand stage WITHOUT the tn_frag_piece(s, r) KiB.
CMB_HD int tn_frag_off(int sub0, int lane) {
  const int q = lane >> 4, i = lane & 15, brow = i >> 2, bp = i & 3;
  const int sub = sub0 + (q & 1);
  return (q >> 1) * CMB_TN_PIECE_BYTES + brow * CMB_TN_ROW_BYTES + (((2 * sub + (bp >> 1)) ^ (4 * brow)) << 4) + (bp & 1) * 8;
}
// k-step s (0..3) of a stage, read r (0..1): first of the two consecutive 4-row pieces the wave's halves read
CMB_HD int tn_frag_piece(int s, int r) { return 4 * s + 2 * r; }
