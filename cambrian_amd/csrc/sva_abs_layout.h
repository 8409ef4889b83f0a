// sva_abs_layout.h — index arithmetic of sva_absorbed.hip's LDS images (token window, backward operand stage), free of HIP
// types so that tests/csrc/sva_abs_layout_sim.cpp runs the very same functions on the host under the documented semantics
// of ds_read_b64_tr_b16 and the gfx950 ds_read_b128 lane grouping.
//
// Token window: row t (token, 0..15) = two halves of 1 KiB = 64 slots of 16 bytes (8 channels) each; SOURCE slot c of a
// half is stored at slot c ^ 4 (t & 15).  Rows are 2 KiB apart (all on bank 0): without the rotation the 16 rows a wave
// instruction touches at one channel offset would be served one after the other; with it every service group of every
// access below puts at most two lanes on a bank (the simulation counts them).  LDS-DMA writes lane l's 16 bytes at + 16 l,
// so the rotation is applied to the lane's source slot.
// Operand stage of the backward dX product: W = (dXb; U), row w (0..31) = 1 KiB = 64 slots (one 512-channel half), SOURCE
// slot c stored at slot c ^ 4 (w & 15); written with ds_write_b128 from registers.
// Lane roles: i = lane & 15, qd = lane >> 4 (sva_absorbed.hip).
#pragma once
#include "gemm_layout.h"   // CMB_HD

CMB_HD int abs_win_src_slot(int lane, int t) { return (lane ^ (4 * t)) & 63; }   // DMA of (t, half): LDS slot `lane` <- this source slot
// byte offset of source slot `slot` (0..63) of half `half` of row t: the definition the cheaper forms below are checked against
CMB_HD int abs_win_off(int t, int half, int slot) { return t * 2048 + half * 1024 + (((slot ^ (4 * t)) & 63) << 4); }
CMB_HD int abs_w_off(int w, int slot) { return w * 1024 + (((slot ^ (4 * (w & 15))) & 63) << 4); }

// The forms the kernels evaluate (two or three VALU operations on a hoisted base; equal to the definitions above for
// i < 16, qd < 4 — the simulation compares them):
// score product (A operand of the 16x16x32 MFMA): row i, channels [32 s + 8 qd, + 8), s = 0..31
//   = abs_win_off(i, s >> 4, 4 (s & 15) + qd)
CMB_HD int abs_rows_off(int i, int qd, int s) { return i * 2048 + qd * 16 + (s >> 4) * 1024 + (((s & 15) ^ i) << 6); }
// token mix (transposing read): the lane supplies row tk = 4 qd + (i >> 2), 8-byte piece i & 3 of the 32-channel group cg,
// tile T: channels 32 cg + 8 (i & 3) + 4 T .. + 3; it receives channel 32 cg + 8 (i >> 2) + 4 T + (i & 3) of tokens 4 qd .. + 3
//   = abs_win_off(tk, cg >> 4, 4 (cg & 15) + (i & 3)) + 8 T
CMB_HD int abs_mix_off(int i, int qd, int cg, int T) {
  const int tk = 4 * qd + (i >> 2);
  return tk * 2048 + (i & 3) * 16 + (cg >> 4) * 1024 + (((cg & 15) ^ tk) << 6) + 8 * T;
}
// lane (i, qd) writes the s-th 8-channel group of its registers (channels 32 s + 8 qd of the half, s = 0..15) into row `row`
//   = abs_w_off(row, 4 s + qd)
CMB_HD int abs_w_write_off(int row, int qd, int s) { return row * 1024 + qd * 16 + ((s ^ (row & 15)) << 6); }
// transposing read of W: the lane supplies row w = 8 qd + 4 hi + (i >> 2) (hi = 0, 1: the two reads of one 8-deep operand),
// piece i & 3 of channel group cg (0..15 inside the half), tile T
//   = abs_w_off(w, 4 cg + (i & 3)) + 8 T
CMB_HD int abs_w_read_off(int i, int qd, int hi, int cg, int T) {
  const int w = 8 * qd + 4 * hi + (i >> 2);
  return w * 1024 + (i & 3) * 16 + ((cg ^ (w & 15)) << 6) + 8 * T;
}
