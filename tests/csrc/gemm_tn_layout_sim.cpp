// Host-side simulation of gemm_tn.hip's operand stage (gemm_tn_layout.h), compiled with g++ by
// tests/test_tn_layout_sim.py.  Model of ds_read_b64_tr_b16 (cdna_hip_programming.md §LDS / T10): inside a group of 16
// lanes, lane i' supplies the address of 4 consecutive bf16 = row i' >> 2, columns 4 (i' & 3) .. + 3 of a [4][16] block;
// lane i receives column i: element j = row j of the block, taken from the piece lane 4 j + (i >> 2) supplied, element
// i & 3.  The instruction is serviced in two halves of 32 lanes; bank of byte address a = (a / 4) % 64.
// Checks, for every wave position (fragment base), k-step and read:
//   1. lane l receives column base + (l & 31) of the tile and 4 contraction rows that depend only on (l >> 5, s, r);
//   2. over the 4 k-steps x 2 reads the lanes' halves see all 64 rows of the stage exactly once (so a dot product over
//      the stage is complete, whatever order the MFMA takes its k in, as At and Bt use the same functions);
//   3. every transposing read is conflict-free in both halves of the wave (each of the 64 banks at most once);
//   4. the DMA fills every slot of the stage exactly once.
#include <cstdio>
#include <set>
#include <vector>
#include "../../cambrian_amd/csrc/gemm_tn_layout.h"

int main() {
  const int ROWS = 64, COLS = 128;
  // LDS image in bf16 elements: value = k * 1024 + column
  std::vector<int> lds(ROWS * COLS, -1);
  for (int piece = 0; piece < 16; ++piece)
    for (int lane = 0; lane < 64; ++lane) {
      const int k = tn_dma_row(piece, lane), slot = tn_dma_src_slot(lane);
      const int dst = tn_dma_lds_off(piece, lane) / 2;
      for (int e = 0; e < 8; ++e) {
        if (lds[dst + e] != -1) { printf("DMA overlap\n"); return 1; }
        lds[dst + e] = k * 1024 + slot * 8 + e;
      }
    }
  for (int v : lds)
    if (v == -1) { printf("stage slot never written\n"); return 1; }
  for (int sub0 = 0; sub0 < 8; sub0 += 2) {          // fragment bases: (wave row or column * 64 + f * 32) / 16
    std::vector<int> seen[2] = {std::vector<int>(ROWS, 0), std::vector<int>(ROWS, 0)};
    for (int s = 0; s < 4; ++s)
      for (int r = 0; r < 2; ++r) {
        int addr[64];
        for (int lane = 0; lane < 64; ++lane) {
          addr[lane] = tn_frag_off(sub0, lane) + tn_frag_piece(s, r) * CMB_TN_PIECE_BYTES;
          if (addr[lane] % 8) { printf("transposing read not 8-byte aligned\n"); return 1; }
        }
        int krows[2][4];
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, i = lane & 15;
          for (int j = 0; j < 4; ++j) {
            const int supplier = g * 16 + 4 * j + (i >> 2);
            const int v = lds[addr[supplier] / 2 + (i & 3)];
            const int k = v / 1024, col = v % 1024;
            if (col != sub0 * 16 + (lane & 31)) { printf("lane %d got column %d, wants %d\n", lane, col, sub0 * 16 + (lane & 31)); return 1; }
            if ((lane & 31) == 0) krows[lane >> 5][j] = k;
            else if (krows[lane >> 5][j] != k) { printf("contraction rows differ inside a wave half\n"); return 1; }
          }
        }
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 4; ++j) seen[h][krows[h][j]]++;
        for (int h = 0; h < 2; ++h) {               // two service halves of 32 lanes
          std::set<int> banks;
          for (int lane = 32 * h; lane < 32 * h + 32; ++lane)
            for (int d = 0; d < 2; ++d)
              if (!banks.insert((addr[lane] / 4 + d) % 64).second) { printf("bank conflict sub0 %d s %d r %d half %d\n", sub0, s, r, h); return 1; }
        }
      }
    for (int k = 0; k < ROWS; ++k)
      if (seen[0][k] + seen[1][k] != 1) { printf("row %d of the stage is read %d times\n", k, seen[0][k] + seen[1][k]); return 1; }
  }
  printf("OK\n");
  return 0;
}
