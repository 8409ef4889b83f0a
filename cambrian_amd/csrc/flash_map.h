// flash_map.h — block -> work mapping of the flash attention kernels (flash_bwd.hip), host + device so that
// tests/csrc/flash_map_sim.cpp can enumerate it on the CPU for arbitrary shapes.
//
// The hardware deals linear block ids round-robin to the 8 XCDs; with a (query block, head, batch) grid every XCD would
// get two fixed query blocks of each head — under the causal mask XCD 0 then carries 2.4x the tiles of XCD 7.  Instead
// the grid is 1-D and an XCD owns a contiguous range of (batch, KV head) groups (gl_xcd_remap): all work items and all
// query heads of a group run on one XCD, which balances the causal triangle and keeps the group's K / V in that XCD's
// L2.  Causal kernels additionally take the work items in PAIRS (i, n-1-i) inside one workgroup (uniform work, no tail):
// `per` below is then ceil(n / 2) and flash_pair_* give the two (or one, for the middle item of an odd n) block indices.
#pragma once
#include "gemm_layout.h"

struct FlashBlock {
  int b, hk, h, blk;
};

// one work item per (query-side item, query head): `n_items` items per head, H query heads in groups of H / HKV
CMB_HD FlashBlock flash_block_qh(int bid, int nblk, int n_items, int H, int HKV) {
  const int group = H / HKV, per = n_items * group;
  const int id = gl_xcd_remap(bid, nblk);
  const int gi = id / per, w = id - gi * per;
  FlashBlock f;
  f.b = gi / HKV; f.hk = gi - f.b * HKV;
  f.blk = w / group;
  f.h = f.hk * group + w % group;
  return f;
}

// one work item per (key-side item, KV head): `per` items per (batch, KV head)
CMB_HD FlashBlock flash_block_kv(int bid, int nblk, int per, int HKV) {
  const int id = gl_xcd_remap(bid, nblk);
  const int gi = id / per;
  FlashBlock f;
  f.b = gi / HKV; f.hk = gi - f.b * HKV; f.h = 0;
  f.blk = id - gi * per;
  return f;
}

// causal pairing of n blocks: item i in [0, ceil(n/2)) covers blocks i and n-1-i (once if they coincide)
CMB_HD int flash_items(int n, bool causal) { return causal ? (n + 1) / 2 : n; }
CMB_HD int flash_pair_count(int n, int item, bool causal) { return (causal && item != n - 1 - item) ? 2 : 1; }
// query-side kernels start with the heavier block n-1-i, key-side kernels with block i (block 0 is the heaviest there)
CMB_HD int flash_pair_q(int n, int item, int rep, bool causal) { return !causal ? item : (rep == 0 ? n - 1 - item : item); }
CMB_HD int flash_pair_k(int n, int item, int rep, bool causal) { return !causal ? item : (rep == 0 ? item : n - 1 - item); }
