// Host-side simulation of vit_attn_dma_kernel<64>'s K / V tile (cambrian_amd/csrc/vit_layout.h: 64 keys x 128 bytes), compiled with
// g++ by tests/test_tn_layout_sim.py — the model and the checks of flash_layout_sim.cpp on the 128-byte-row image: the DMA
// fills every slot once; a row-major fragment read hands lane (j, g) columns 16 ks + 8 g .. + 7 of row row0 + j, conflict-free
// in each ds_read_b128 lane group; a transposing fragment hands lane (j, g) column c32 + j and rows r16 + 8 read + 4 g + {0..3},
// conflict-free in each half.
#include <cstdio>
#include <set>
#include <vector>
#include "../../cambrian_amd/csrc/vit_layout.h"

int main() {
  const int ROWS = 64, COLS = 64;
  std::vector<int> lds(ROWS * COLS, -1);   // bf16 elements: value = row * 1024 + column
  for (int piece = 0; piece < 8; ++piece)
    for (int lane = 0; lane < 64; ++lane) {
      const int r = vl_dma_row(piece, lane), slot = vl_dma_src_slot(piece, lane);
      const int dst = vl_dma_lds_off(piece, lane) / 2;
      for (int e = 0; e < 8; ++e) {
        if (lds[dst + e] != -1) { printf("DMA overlap\n"); return 1; }
        lds[dst + e] = r * 1024 + slot * 8 + e;
      }
    }
  for (int v : lds)
    if (v == -1) { printf("tile slot never written\n"); return 1; }
  // ---- row-major fragments
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  for (int row0 = 0; row0 < 64; row0 += 32)
    for (int ks = 0; ks < 4; ++ks) {
      int addr[64];
      for (int lane = 0; lane < 64; ++lane) {
        addr[lane] = vl_row_frag_off(row0, ks, lane);
        if (addr[lane] % 16) { printf("row fragment not 16-byte aligned\n"); return 1; }
        for (int e = 0; e < 8; ++e) {
          const int v = lds[addr[lane] / 2 + e];
          if (v != (row0 + (lane & 31)) * 1024 + 16 * ks + 8 * (lane >> 5) + e) { printf("row fragment: lane %d element %d wrong\n", lane, e); return 1; }
        }
      }
      for (int gi = 0; gi < 4; ++gi) {
        std::set<int> banks;
        for (int k = 0; k < 16; ++k)
          for (int d = 0; d < 4; ++d)
            if (!banks.insert((addr[groups[gi][k]] / 4 + d) % 64).second) { printf("row fragment bank conflict row0 %d ks %d group %d\n", row0, ks, gi); return 1; }
      }
    }
  // ---- transposing fragments
  for (int r16 = 0; r16 < 64; r16 += 16)
    for (int c32 = 0; c32 < 64; c32 += 32)
      for (int read = 0; read < 2; ++read) {
        int addr[64];
        for (int lane = 0; lane < 64; ++lane) {
          addr[lane] = vl_tr_frag_off(r16, c32, read, lane);
          if (addr[lane] % 8) { printf("transposing read not 8-byte aligned\n"); return 1; }
        }
        for (int lane = 0; lane < 64; ++lane) {
          const int q = lane >> 4, i = lane & 15, g = lane >> 5;
          for (int j = 0; j < 4; ++j) {
            const int supplier = q * 16 + 4 * j + (i >> 2);
            const int v = lds[addr[supplier] / 2 + (i & 3)];
            const int row = v / 1024, col = v % 1024;
            if (col != c32 + (lane & 31)) { printf("tr: lane %d got column %d, wants %d\n", lane, col, c32 + (lane & 31)); return 1; }
            if (row != r16 + 8 * read + 4 * g + j) { printf("tr: lane %d element %d got row %d, wants %d\n", lane, j, row, r16 + 8 * read + 4 * g + j); return 1; }
          }
        }
        for (int h = 0; h < 2; ++h) {
          std::set<int> banks;
          for (int lane = 32 * h; lane < 32 * h + 32; ++lane)
            for (int d = 0; d < 2; ++d)
              if (!banks.insert((addr[lane] / 4 + d) % 64).second) { printf("tr bank conflict r16 %d c32 %d read %d half %d\n", r16, c32, read, h); return 1; }
        }
      }
  printf("OK\n");
  return 0;
}
