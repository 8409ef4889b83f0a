// flash_common.h — parameter block and mask helpers shared by flash_bwd.hip (round 1-4 kernels + their round-5 tile bodies)
// and flash2.hip (round-5 forward on LDS-DMA tiles).
#pragma once
#include "common.h"
#include "flash_map.h"

namespace cmb_flash {

constexpr int HD = 128;       // head dim
constexpr int KS = HD / 16;   // MFMA k-steps over the head dim
constexpr int DT = HD / 32;   // 32-wide d tiles
constexpr int LDR = HD + 8;   // row-major LDS tile row stride (elements): conflict-free ds_read_b128
constexpr int LDT = 68;       // transposed LDS tile row stride (elements): conflict-free ds_read_b64
constexpr float LOG2E = 1.4426950408889634f;

struct FlashParams {
  const bf16_t *q, *k, *v, *o, *dout;
  bf16_t *dq, *dk, *dv;
  const float* lse;  // [B, H, S]
  float* dvec;       // [2, B, H, S]: [0] D = rowsum(dO * O); [1] unused since round 6 (was lse * log2 e for the removed flash_dkdv2_kernel)
  int64_t q_sb, q_ss, q_sh;     // strides of q / o / do / dq (elements)
  int64_t kv_sb, kv_ss, kv_sh;  // strides of k / v / dk / dv
  int B, S, H, HKV;
  int kv_len;   // non-causal kernels: keys >= kv_len are padding (masked); S is kv_len rounded up to 128
  float scale;
  // causal kernels: optional key-padding mask, [B, S] bytes, non-zero = the key may be attended to.  A query may see key k
  // iff k <= q and (key_valid[b][k] or k == q): the collator's attention_mask (train_fsdp.py:1057-1085) AND the causal
  // triangle, with the diagonal kept open so that a padded query row is never empty (its loss is ignored).
  const uint8_t* key_valid;
};

// validity bits of the 64 keys of tile t (lane i contributes key 64 t + i); all ones without a mask
__device__ __forceinline__ uint8_t kv_byte(const FlashParams& p, int b, int t, int lane) {
  return p.key_valid ? p.key_valid[(int64_t)b * p.S + t * 64 + lane] : (uint8_t)1;
}

// Which of this lane's 16 keys of one 32-key half tile are open to its query: bit (r & 3) + 8 (r >> 2) <-> accumulator
// element r (key = half tile base + 4 g + that bit index).  `vw` is the tile's 64-bit key-validity ballot; `dq` = the
// lane's query index minus the half tile's first key: the query's own key stays open even when it is padding (the
// diagonal of the collator mask, train_fsdp.py:1057-1085) — folded into the word here so that the per-element test is
// one constant-bit test.
__device__ __forceinline__ uint32_t flash_open_bits(uint64_t vw, int kt, int g, int dq) {
  const uint32_t w = (uint32_t)(vw >> (kt * 32)) >> (4 * g);
  const uint32_t pos = (uint32_t)(dq - 4 * g);  // bit of the query's own key in w (if < 32 and in this lane's groups)
  return w | ((pos < 32u && !(pos & 4u)) ? (1u << pos) : 0u);
}


// flash2.hip: forward on LDS-DMA operand tiles (flash_layout.h), selected by knob CMB_KNOB_FLASH bit 0
int launch_flash_fwd2(const FlashParams& p, bf16_t* out, float* lse, bool causal, hipStream_t stream);
// flash2.hip: dQ on the same tiles (knob bit 1); it also writes dvec (D = rowsum(dO o O)) for the dK/dV kernel
int launch_flash_dq2(const FlashParams& p, bool causal, hipStream_t stream);

}  // namespace cmb_flash
