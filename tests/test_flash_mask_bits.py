"""CPU check of the key-padding bit arithmetic of the flash kernels (cambrian_amd/csrc/flash_bwd.hip::flash_open_bits),
restated lane by lane: for every (key tile, 32-key half, lane, accumulator element) the constant-bit test the kernel applies
on top of the causal test must equal the collator's rule
    allowed(q, k) = k <= q and (key_valid[k] or k == q)            (train_fsdp.py:1057-1085: padding closed, diagonal open)
for random validity patterns, including queries that are themselves padding."""
import random


def flash_open_bits(vw, kt, g, dq):
    w = ((vw >> (kt * 32)) & 0xFFFFFFFF) >> (4 * g)
    pos = (dq - 4 * g) & 0xFFFFFFFF                      # uint32 wrap of a negative difference
    return (w | ((1 << pos) if (pos < 32 and not (pos & 4)) else 0)) & 0xFFFFFFFF


def test_bit_test_equals_the_collator_rule():
    rng = random.Random(0)
    S = 256
    for trial in range(20):
        valid = [rng.random() < (0.5 if trial % 2 else 0.9) for _ in range(S)]
        for t in range(S // 64):                          # key tile
            vw = sum(1 << c for c in range(64) if valid[t * 64 + c])
            for q0 in range(0, S, 32):                    # a wave's 32 queries
                for lane in range(64):
                    j, g = lane & 31, lane >> 5
                    qi = q0 + j
                    for kt in range(2):
                        w = flash_open_bits(vw, kt, g, qi - (t * 64 + kt * 32))
                        for r in range(16):
                            bit = (r & 3) + 8 * (r >> 2)
                            key = t * 64 + kt * 32 + bit + 4 * g
                            causal_ok = key <= qi
                            kernel_open = causal_ok and bool(w & (1 << bit))
                            rule = key <= qi and (valid[key] or key == qi)
                            assert kernel_open == rule, (trial, t, q0, lane, kt, r)
