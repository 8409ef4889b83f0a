// CPU enumeration of the flash attention kernels' block -> work mapping (cambrian_amd/csrc/flash_map.h, the header the
// kernels include): for a sweep of (B, H, HKV, blocks per sequence, causal) it walks every block id of the 1-D grids and
// checks that
//   1. every (batch, head, query block) is produced exactly once by the query-side grid (forward, dQ) and every
//      (batch, KV head, key block) exactly once by the key-side grid (dK/dV), pairs included;
//   2. all work items of a (batch, KV head) group land on ONE XCD (block id % 8), and the tile counts of the 8 XCDs under
//      the causal mask differ by at most one group's worth of work;
//   3. a causal pair always carries n + 1 block-units of work (uniform workgroups).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <vector>
#include "../../cambrian_amd/csrc/flash_map.h"

static int check(int B, int H, int HKV, int n, bool causal) {
  const int items = flash_items(n, causal);
  // ---- query side
  {
    const int nblk = items * H * B;
    std::set<std::tuple<int, int, int>> seen;
    std::map<std::pair<int, int>, std::set<int>> xcd_of_group;
    std::vector<long> load(8, 0);
    for (int bid = 0; bid < nblk; ++bid) {
      const FlashBlock f = flash_block_qh(bid, nblk, items, H, HKV);
      if (f.b < 0 || f.b >= B || f.h < 0 || f.h >= H || f.hk != f.h / (H / HKV) || f.blk < 0 || f.blk >= items) return 1;
      const int reps = flash_pair_count(n, f.blk, causal);
      long work = 0;
      for (int rep = 0; rep < reps; ++rep) {
        const int qb = flash_pair_q(n, f.blk, rep, causal);
        if (qb < 0 || qb >= n) return 2;
        if (!seen.insert({f.b, f.h, qb}).second) return 3;
        work += causal ? qb + 1 : n;
      }
      if (causal && reps == 2 && work != n + 1) return 4;
      xcd_of_group[{f.b, f.hk}].insert(bid & 7);
      load[bid & 7] += work;
    }
    if ((int)seen.size() != B * H * n) return 5;
    for (auto& kv : xcd_of_group)
      if (kv.second.size() != 1 && nblk % 8 == 0 && (B * HKV) % 8 == 0) return 6;
    if ((B * HKV) % 8 == 0) {
      long lo = load[0], hi = load[0];
      for (long v : load) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
      if (hi != lo) return 7;   // whole groups per XCD: perfectly even
    }
  }
  // ---- key side
  {
    const int nblk = items * HKV * B;
    std::set<std::tuple<int, int, int>> seen;
    for (int bid = 0; bid < nblk; ++bid) {
      const FlashBlock f = flash_block_kv(bid, nblk, items, HKV);
      if (f.b < 0 || f.b >= B || f.hk < 0 || f.hk >= HKV || f.blk < 0 || f.blk >= items) return 11;
      const int reps = flash_pair_count(n, f.blk, causal);
      long work = 0;
      for (int rep = 0; rep < reps; ++rep) {
        const int kb = flash_pair_k(n, f.blk, rep, causal);
        if (kb < 0 || kb >= n) return 12;
        if (!seen.insert({f.b, f.hk, kb}).second) return 13;
        work += causal ? n - kb : n;
      }
      if (causal && reps == 2 && work != n + 1) return 14;
    }
    if ((int)seen.size() != B * HKV * n) return 15;
  }
  return 0;
}

int main() {
  const int shapes[][3] = {{16, 32, 8}, {2, 8, 2}, {1, 4, 4}, {3, 6, 2}, {5, 3, 3}, {1, 1, 1}, {7, 24, 24}, {16, 40, 10}};
  for (auto& s : shapes)
    for (int n = 1; n <= 17; ++n)
      for (int causal = 0; causal < 2; ++causal) {
        const int rc = check(s[0], s[1], s[2], n, causal != 0);
        if (rc) {
          printf("FAIL rc=%d B=%d H=%d HKV=%d n=%d causal=%d\n", rc, s[0], s[1], s[2], n, causal);
          return 1;
        }
      }
  printf("OK\n");
  return 0;
}
