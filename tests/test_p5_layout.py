"""CPU check of the address arithmetic `gemm_nt_p5_kernel` adds (cambrian_amd/csrc/gemm_p5.hip, gemm_p5_epilogue.inc), restated
lane by lane: (a) the LDS-DMA pieces of a tile cover every (row, 16-byte chunk) of an operand buffer exactly once and put
logical chunk c of row r where the fragment reads look for it; (b) the fragment reads of a wave are bank-conflict-free under
the ds_read_b128 service model of MI355X_MICROARCH.md (16 lanes per pass over 64 banks x 4 B); (c) the epilogue's staging
area: every 16-byte piece written in the accumulator layout is read back exactly once by the row-contiguous store layout,
both directions conflict-free, and the read-back lanes of a store instruction cover whole 128-byte lines."""
import itertools

ROW = 128                      # bytes per tile row (64 bf16)


def gl_swz(r):                 # gemm_layout.h
    return (r >> 1) & 7


def lds_off(r, c):             # chunk c of row r
    return r * ROW + ((c ^ gl_swz(r)) << 4)


def conflict_free(addrs16):
    """16 lanes x 16 B in one pass: all 64 banks (4 B) distinct."""
    banks = set()
    for a in addrs16:
        for b in range(4):
            banks.add(((a >> 2) + b) & 63)
    return len(banks) == 64


def test_dma_pieces_tile_the_buffer_and_match_the_fragment_reads():
    seen = {}
    for wave, piece, lane in itertools.product(range(4), range(8), range(64)):
        row = wave * 64 + 8 * piece + (lane >> 3)            # src_of(): d_row + 8 i
        logical = (lane & 7) ^ gl_swz(row)                    # the lane FETCHES this logical chunk of its row ...
        dst = wave * 8192 + piece * 1024 + lane * 16          # ... and the DMA puts it at M0 + 16 lane
        assert dst == lds_off(row, logical)                   # = where chunk `logical` of `row` is expected
        assert (row, logical) not in seen
        seen[(row, logical)] = dst
    assert len(seen) == 256 * 8 and sorted(seen.values()) == list(range(0, 256 * ROW, 16))


def test_fragment_reads_are_conflict_free():
    for wm, blk, ks in itertools.product(range(2), range(4), range(4)):
        addrs = []
        for lane in range(64):
            row = wm * 128 + blk * 32 + (lane & 31)
            koff = ((2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4   # a_rd[ks]: gl_swz(row) == (lane >> 1) & 7
            assert koff == ((2 * ks + (lane >> 5)) ^ gl_swz(row)) << 4
            addrs.append(row * ROW + koff)
        for g in range(4):
            assert conflict_free(addrs[16 * g:16 * g + 16]), (wm, blk, ks, g)


def test_epilogue_staging_round_trip():
    # phase 1 (accumulator layout): row block i & 1, column group g of the half, lane -> row rl, chunk 2 g + (lane >> 5)
    written = {}
    for ib, g, lane in itertools.product(range(2), range(4), range(64)):
        rl = 32 * ib + (lane & 31)
        chunk = 2 * g + (lane >> 5)
        off = rl * ROW + ((chunk ^ gl_swz(rl)) << 4)
        assert off not in written
        written[off] = (rl, chunk)
    assert len(written) == 64 * 8
    for ib, g in itertools.product(range(2), range(4)):       # one ds_write_b128 instruction
        addrs = [(32 * ib + (l & 31)) * ROW + (((2 * g + (l >> 5)) ^ gl_swz(32 * ib + (l & 31))) << 4) for l in range(64)]
        for q in range(4):
            assert conflict_free(addrs[16 * q:16 * q + 16])
    # phase 2 (store layout): instruction k, lane -> row 8 k + (lane >> 3), chunk lane & 7
    read = set()
    for k in range(8):
        addrs = []
        for lane in range(64):
            rl0 = lane >> 3
            base = rl0 * ROW + (((lane & 7) ^ gl_swz(rl0 + (8 if k & 1 else 0))) << 4)   # stage_r0 / stage_r1
            off = base + k * 1024
            rl, chunk = 8 * k + rl0, lane & 7
            assert written[off] == (rl, chunk)                # the piece read is the piece the store expects
            assert off not in read
            read.add(off)
            addrs.append(off)
        for q in range(4):
            assert conflict_free(addrs[16 * q:16 * q + 16])
        for line in range(8):                                 # 8 consecutive lanes = one row = one 128-byte line of C
            assert [(written[a][0], written[a][1]) for a in addrs[8 * line:8 * line + 8]] == [(8 * k + line, c) for c in range(8)]
    assert len(read) == 64 * 8
