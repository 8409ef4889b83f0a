"""CPU model of the hand-counted vector-memory waits of `gemm_nt_p5_kernel` (cambrian_amd/csrc/gemm_p5.hip) and of its
epilogue's residual loads (gemm_p5_epilogue.inc).  A wave's vector-memory operations retire IN ORDER on gfx9 and
`s_waitcnt vmcnt(N)` returns when at most N are outstanding, so a wait is correct iff everything it needs is older than
the N youngest operations issued before it.  The model replays one wave's issue order over several items (persistent
cursor: the last two tiles of an item stage the first two of the next) with the slot numbers and wait constants READ
FROM THE SOURCE, and checks every wait:
  * B2 of tile t needs all 16 LDS-DMA pieces of tile t + 1;
  * the first B2 after an epilogue that issued exactly 32 stores is the relaxed one and must still be sufficient, every
    later one must not rely on the stores being allowed outstanding;
  * residual load n of a half tile is needed when row block i, column group g is processed (staged order)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "cambrian_amd", "csrc", "gemm_p5.hip")).read()
EPI = open(os.path.join(ROOT, "cambrian_amd", "csrc", "gemm_p5_epilogue.inc")).read()


def consts():
    m = re.search(r'if \(rl\) asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\((\d+) \+ (\d+)\)', SRC)
    relaxed = int(m.group(1)) + int(m.group(2))
    strict = int(re.search(r'else asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\((\d+)\) : "memory"\);\n\s+P4_BARRIER\(\);\n\s+\}\n\s+if constexpr \(L == 63\) advance', SRC).group(1))
    a = re.search(r"L >= (\d+) && L <= (\d+) && \(\(L - \d+\) % (\d+)\) == 0\)\s+// A-row pieces", SRC)
    b = re.search(r"L >= (\d+) && L <= (\d+) && \(\(L - \d+\) % (\d+)\) == 0\)\s+// B-row pieces", SRC)
    kb2 = int(re.search(r"constexpr int kB2 = (\d+),", SRC).group(1))
    pro = int(re.search(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)" ::: "memory"\);\n\s+P4_BARRIER\(\);\n#pragma unroll\n\s+for \(int w = 0; w < 8; \+\+w\) read_frag\(0, 0u', SRC).group(1))
    slots = [s for s in range(int(a.group(1)), int(a.group(2)) + 1, int(a.group(3)))] + \
            [s for s in range(int(b.group(1)), int(b.group(2)) + 1, int(b.group(3)))]
    return relaxed, strict, slots, kb2, pro


def test_constants_are_what_the_model_expects():
    relaxed, strict, slots, kb2, pro = consts()
    assert len(slots) == 16 and slots == sorted(slots) and max(slots) <= 63
    assert strict == sum(1 for s in slots if s <= kb2)          # the pieces of THIS tile issued before its wait
    assert relaxed == strict + 32 and relaxed <= 63               # vmcnt is a 6-bit counter
    assert pro == 16


def run_wave(items, fast_epilogue):
    """items: tiles per item.  Returns the list of (wait position in the op stream, allowed, needed op ids)."""
    relaxed, strict, slots, kb2, pro = consts()
    ops, waits = [], []                 # ops: (kind, tag)
    def piece(tile_id):
        ops.append(("piece", tile_id))
    # global tile ids across items: tile k of item n -> base[n] + k
    base, acc = [], 0
    for nt in items:
        base.append(acc)
        acc += nt
    total = acc
    for _ in range(16):
        piece(0)
    for _ in range(16):
        piece(1)
    waits.append((len(ops), pro, 0))                                   # prologue: tile 0 must have landed
    relax = False
    for n, nt in enumerate(items):
        for k in range(nt):
            t = base[n] + k
            stage = t + 2 if t + 2 < total else None                   # past the end: re-reads (never consumed)
            before = [s for s in slots if s <= kb2]
            for _ in before:
                piece(stage)
            waits.append((len(ops), relaxed if (k == 0 and relax) else strict, t + 1 if t + 1 < total else None))
            for _ in range(16 - len(before)):
                piece(stage)
        stores = 32 if fast_epilogue[n] else 7                          # a slow path issues something else and drains
        for _ in range(stores):
            ops.append(("store", n))
        if not fast_epilogue[n]:
            ops.append(("drain", n))                                    # s_waitcnt vmcnt(0) at the end of the slow path
        relax = fast_epilogue[n]
    return ops, waits


def check(ops, waits):
    drained = 0
    for pos, allowed, need in waits:
        # a drain (vmcnt(0)) retires everything before it
        for q in range(pos - 1, -1, -1):
            if ops[q][0] == "drain":
                drained = max(drained, q + 1)
                break
        outstanding_from = max(drained, pos - allowed)                  # ops [outstanding_from, pos) may be in flight
        if need is None:
            continue
        needed = [q for q in range(pos) if ops[q] == ("piece", need)]
        assert len(needed) == 16, (pos, need, len(needed))
        assert max(needed) < outstanding_from, (pos, allowed, need, max(needed), outstanding_from)


def test_every_tile_wait_covers_the_next_tile():
    for items in ([2, 2, 2], [3, 2, 5], [6, 6, 6, 6], [24, 24], [2], [128]):
        for pattern in range(1 << len(items)):
            fast = [bool(pattern >> n & 1) for n in range(len(items))]
            check(*run_wave(items, fast))


def test_relaxed_wait_is_needed_and_only_once():
    # with the strict constant the first wait after a fast epilogue would also be correct (it waits for MORE); with the
    # relaxed constant on the SECOND tile it would not — the model must reject that
    relaxed, strict, slots, kb2, pro = consts()
    ops, waits = run_wave([4, 4], [True, True])
    bad = list(waits)
    idx = [i for i, w in enumerate(bad) if w[2] == 6][0]                # B2 of item 1's tile 1 (needs global tile 6)
    bad[idx] = (bad[idx][0], relaxed, bad[idx][2])
    try:
        check(ops, bad)
    except AssertionError:
        return
    raise AssertionError("a relaxed wait on the second tile after an epilogue must be caught")


def test_residual_load_waits_of_the_staged_epilogue():
    m = re.search(r'"n"\((\d+) - \(4 \* i \+ g\) \+ (\d+) \* \(i >> 1\)\)', EPI)
    c0, c1 = int(m.group(1)), int(m.group(2))
    # staged order of one half tile (the only store form of the fast path since round 4: a row-mapped C leaves through the
    # generic path): 16 loads (index 4 i + g), then for ip in 0, 1: for g: for i in (2 ip, 2 ip + 1): wait; after each ip 8 stores
    ops = [("load", 4 * i + g) for i in range(4) for g in range(4)]
    for ip in range(2):
        for g in range(4):
            for i in (2 * ip, 2 * ip + 1):
                allowed = c0 - (4 * i + g) + c1 * (i >> 1)
                pos = len(ops)
                q = ops.index(("load", 4 * i + g))
                assert q == pos - allowed - 1, (i, g, allowed)       # exactly tight: load n is the oldest op it may not leave behind
        ops += [("store", ip)] * 8
    # the fast path is staged only, and a residual needs every row of the wave in range (the counted stores)
    assert "p.c_map.n1 == 0 && (!p.R || rows_all)" in EPI and "kEpiStage != 0 &&" in EPI
