// Host-side simulation of sva_absorbed.hip's LDS images (sva_abs_layout.h), compiled with g++ by
// tests/test_tn_layout_sim.py.  Model of ds_read_b64_tr_b16 as in gemm_tn_layout_sim.cpp; ds_read_b128 / ds_write_b128 are
// serviced in the four 16-lane groups of gfx950 (MI355X_MICROARCH.md §LDS), bank of byte address a = (a / 4) % 64.
// Checks:
//   1. the forms the kernels evaluate equal the slot-rotation definitions (abs_win_off / abs_w_off);
//   2. window: after the LDS-DMA of 16 token rows, the score product's read of (lane, s) holds token lane & 15, channels
//      [32 s + 8 (lane >> 4), + 8) — the B operand U[h = lane & 15] is loaded for the same channels, so the 32 MFMAs sum
//      over all 1024 channels once;
//   3. token mix: for every 32-channel group and tile T the lane receives channel 32 cg + 8 (i >> 2) + 4 T + (i & 3)
//      — MFMA row i — of tokens 4 qd .. 4 qd + 3 — contraction slots 4 qd + j, the layout the probabilities have in the
//      accumulator of the score product; D rows 4 qd + r then are channels 32 cg + 8 qd + 4 T + r: 8 consecutive channels
//      per lane over T = 0, 1;
//   4. backward operand stage: what lane (i, qd) wrote for (row, s) is what the transposing reads deliver: channel
//      32 cg + 8 (i >> 2) + 4 T + (i & 3) of W rows 8 qd + 4 hi + j;
//   5. bank conflicts: at most 2 lanes per bank in any service group of any of these accesses (16-way without the rotation).
#include <cstdio>
#include <map>
#include <vector>
#include "../../cambrian_amd/csrc/sva_abs_layout.h"

static const int GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

// worst number of lanes on one bank; `bytes` per lane, service groups of 16 (b128) or halves of 32 (b64)
static int worst_b128(const int* addr) {
  int worst = 0;
  for (int g = 0; g < 4; ++g) {
    std::map<int, int> n;
    for (int t = 0; t < 16; ++t)
      for (int d = 0; d < 4; ++d) worst = std::max(worst, ++n[(addr[GROUPS[g][t]] / 4 + d) % 64]);
  }
  return worst;
}
static int worst_b64(const int* addr) {
  int worst = 0;
  for (int h = 0; h < 2; ++h) {
    std::map<int, int> n;
    for (int lane = 32 * h; lane < 32 * h + 32; ++lane)
      for (int d = 0; d < 2; ++d) worst = std::max(worst, ++n[(addr[lane] / 4 + d) % 64]);
  }
  return worst;
}
// the 4 elements lane receives from a transposing read with per-lane byte addresses addr[] over image img (bf16 elements)
static void tr_read(const std::vector<int>& img, const int* addr, int lane, int (&out)[4]) {
  const int g = lane >> 4, i = lane & 15;
  for (int j = 0; j < 4; ++j) out[j] = img[addr[g * 16 + 4 * j + (i >> 2)] / 2 + (i & 3)];
}

int main() {
  // ---- 1. cheap forms == definitions
  for (int lane = 0; lane < 64; ++lane) {
    const int i = lane & 15, qd = lane >> 4;
    for (int s = 0; s < 32; ++s)
      if (abs_rows_off(i, qd, s) != abs_win_off(i, s >> 4, 4 * (s & 15) + qd)) { printf("abs_rows_off\n"); return 1; }
    for (int cg = 0; cg < 32; ++cg)
      for (int T = 0; T < 2; ++T)
        if (abs_mix_off(i, qd, cg, T) != abs_win_off(4 * qd + (i >> 2), cg >> 4, 4 * (cg & 15) + (i & 3)) + 8 * T) { printf("abs_mix_off\n"); return 1; }
    for (int s = 0; s < 16; ++s)
      for (int row : {i, 16 + i})
        if (abs_w_write_off(row, qd, s) != abs_w_off(row, 4 * s + qd)) { printf("abs_w_write_off\n"); return 1; }
    for (int cg = 0; cg < 16; ++cg)
      for (int hi = 0; hi < 2; ++hi)
        for (int T = 0; T < 2; ++T)
          if (abs_w_read_off(i, qd, hi, cg, T) != abs_w_off(8 * qd + 4 * hi + (i >> 2), 4 * cg + (i & 3)) + 8 * T) { printf("abs_w_read_off\n"); return 1; }
  }
  // ---- window image: value = token * 1024 + channel
  std::vector<int> win(16 * 1024, -1);
  for (int t = 0; t < 16; ++t)
    for (int half = 0; half < 2; ++half)
      for (int lane = 0; lane < 64; ++lane) {
        const int dst = (t * 2048 + half * 1024 + lane * 16) / 2;      // LDS-DMA: lane-linear
        const int src = half * 512 + abs_win_src_slot(lane, t) * 8;
        for (int e = 0; e < 8; ++e) {
          if (win[dst + e] != -1) { printf("window DMA overlap\n"); return 1; }
          win[dst + e] = t * 1024 + src + e;
        }
      }
  int worst_rows = 0, worst_mix = 0, worst_ww = 0, worst_wr = 0;
  // ---- 2. score product
  std::vector<int> cover(1024, 0);
  for (int s = 0; s < 32; ++s) {
    int addr[64];
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane & 15, qd = lane >> 4;
      addr[lane] = abs_rows_off(i, qd, s);
      for (int e = 0; e < 8; ++e)
        if (win[addr[lane] / 2 + e] != i * 1024 + 32 * s + 8 * qd + e) { printf("score product operand mismatch\n"); return 1; }
      if (i == 0)
        for (int e = 0; e < 8; ++e) cover[32 * s + 8 * qd + e]++;
    }
    worst_rows = std::max(worst_rows, worst_b128(addr));
  }
  for (int c : cover)
    if (c != 1) { printf("score product does not cover every channel once\n"); return 1; }
  // ---- 3. token mix
  for (int cg = 0; cg < 32; ++cg)
    for (int T = 0; T < 2; ++T) {
      int addr[64];
      for (int lane = 0; lane < 64; ++lane) addr[lane] = abs_mix_off(lane & 15, lane >> 4, cg, T);
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, qd = lane >> 4;
        if (addr[lane] % 8) { printf("unaligned transposing read\n"); return 1; }
        int v[4];
        tr_read(win, addr, lane, v);
        for (int j = 0; j < 4; ++j)
          if (v[j] != (4 * qd + j) * 1024 + 32 * cg + 8 * (i >> 2) + 4 * T + (i & 3)) { printf("token mix operand mismatch\n"); return 1; }
      }
      worst_mix = std::max(worst_mix, worst_b64(addr));
    }
  // D rows 4 qd + r of tile T stand for MFMA rows i = 4 qd + r, i.e. channels 32 cg + 8 qd + 4 T + r: T = 0, 1 -> 8 in a row
  for (int qd = 0; qd < 4; ++qd)
    for (int T = 0; T < 2; ++T)
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * qd + r;
        if (8 * (i >> 2) + 4 * T + (i & 3) != 8 * qd + 4 * T + r) { printf("store mapping\n"); return 1; }
      }
  // ---- 4. backward operand stage, one half: value = w * 1024 + channel (of the half)
  std::vector<int> wst(32 * 512, -1);
  for (int s = 0; s < 16; ++s)
    for (int part = 0; part < 2; ++part) {
      int addr[64];
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, qd = lane >> 4, row = part * 16 + i;
        addr[lane] = abs_w_write_off(row, qd, s);
        for (int e = 0; e < 8; ++e) {
          if (wst[addr[lane] / 2 + e] != -1) { printf("operand stage write overlap\n"); return 1; }
          wst[addr[lane] / 2 + e] = row * 1024 + 32 * s + 8 * qd + e;
        }
      }
      worst_ww = std::max(worst_ww, worst_b128(addr));
    }
  for (int v : wst)
    if (v == -1) { printf("operand stage slot never written\n"); return 1; }
  for (int cg = 0; cg < 16; ++cg)
    for (int T = 0; T < 2; ++T)
      for (int hi = 0; hi < 2; ++hi) {
        int addr[64];
        for (int lane = 0; lane < 64; ++lane) addr[lane] = abs_w_read_off(lane & 15, lane >> 4, hi, cg, T);
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, qd = lane >> 4;
          int v[4];
          tr_read(wst, addr, lane, v);
          for (int j = 0; j < 4; ++j)
            if (v[j] != (8 * qd + 4 * hi + j) * 1024 + 32 * cg + 8 * (i >> 2) + 4 * T + (i & 3)) { printf("operand stage read mismatch\n"); return 1; }
        }
        worst_wr = std::max(worst_wr, worst_b64(addr));
      }
  printf("lanes per bank, worst service group: score product %d, token mix %d, stage write %d, stage read %d\n", worst_rows,
         worst_mix, worst_ww, worst_wr);
  if (worst_rows > 2 || worst_mix > 2 || worst_ww > 2 || worst_wr > 2) { printf("more than 2 lanes on a bank\n"); return 1; }
  printf("OK\n");
  return 0;
}
