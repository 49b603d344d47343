"""The LDS-DMA schedule of the GEMM core's K loop (csrc/gemm8p.hip: `k_loop`, `slot_pieces`, `wait_tiles_in_flight`), replayed on the host.

The kernel's correctness rests on two pieces of bookkeeping that no GPU test can see failing deterministically:
  * the counted `s_waitcnt vmcnt(N)` in front of every workgroup barrier -- the vector-memory counter retires in order, so "tile t+1 has
    landed" holds exactly when at least N pieces have been issued AFTER tile t+1's last piece (N too large = the barrier releases
    readers onto a tile still in flight; N too small only over-waits);
  * a buffer is refilled only behind the barrier that ends its last readers, and every piece of every tile is issued exactly once.
This test walks the same control flow as the kernel for every instantiated (buffers, k-steps per wave, MFMAs per k-step, pieces per
wave) combination, with and without the refill spread over the tile (`SPREAD`: three or more buffers), and checks those invariants."""
import pytest


def replay(NBUF, KPW, NMMA, NP, nk):
    spread = NBUF >= 3
    nslot = KPW * NMMA
    issued = []          # (tile, piece) in issue order
    barrier_of = {}      # issue index at which the barrier ending tile t's reads (its rendezvous) was passed
    waits = []           # (t, vmcnt argument, index into `issued` at the wait)

    def slot_pieces(j, i, tile):
        for q in range(NP):
            sl = q * nslot // NP
            if sl // NMMA == j and sl % NMMA == i:
                issued.append((tile, q))

    def vmcnt_arg(later):  # wait_tiles_in_flight
        return 0 if later <= 0 else min(later, 3) * NP

    for b in range(min(NBUF, nk)):           # prologue
        for q in range(NP):
            issued.append((b, q))
    waits.append((-1, vmcnt_arg(min(nk, NBUF) - 1), len(issued)))
    for t in range(nk):
        prev_more = spread and t >= 1 and t - 1 + NBUF < nk
        for i in range(KPW - 1):
            for m in range(NMMA):
                if prev_more:
                    slot_pieces(i + 1, m, t - 1 + NBUF)
        if t + 1 < nk:
            waits.append((t, vmcnt_arg(min(nk - 1, t + NBUF - 1) - (t + 1)), len(issued)))
            barrier_of[t] = len(issued)
        more = t + NBUF < nk
        for m in range(NMMA):
            if more:
                if spread:
                    slot_pieces(0, m, t + NBUF)
                else:
                    for q in range(m * NP // NMMA, (m + 1) * NP // NMMA):
                        issued.append((t + NBUF, q))
    return issued, barrier_of, waits


CONFIGS = [  # (NBUF, k-steps per wave and tile, MFMAs per k-step, pieces per moving wave): every instantiation of gemm_mfma_kernel
    (4, 4, 2, 4), (4, 4, 2, 8),      # 128 x 128 / eight waves of 64 x 32, all waves moving | four mover waves
    (4, 2, 4, 4), (4, 2, 4, 8),      # ... as two k-step groups of 64 x 64
    (4, 4, 4, 8),                    # ... as four waves of 64 x 64
    (3, 4, 4, 6), (3, 4, 4, 12),     # 256 x 128 / three buffers
    (2, 4, 2, 4), (2, 4, 4, 8),      # 128 x 128 / two buffers
    (2, 4, 8, 8),                    # 256 x 256 / eight waves of 128 x 64
    (2, 4, 4, 4),                    # 256 x 256 / sixteen waves of 64 x 64
    (2, 4, 6, 7), (2, 4, 6, 14),     # 256 x 192
    (2, 4, 10, 9),                   # 256 x 320 (convolution)
    (2, 4, 10, 11), (2, 4, 5, 6),    # 192 x 320 (convolution, round 5): six waves of 64 x 160 (4 + 7 pieces, the last one padded) / twelve of 32 x 160
]


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 6, 9, 20])
def test_every_piece_once_counted_waits_exact_and_refills_behind_their_barrier(cfg, nk):
    NBUF, KPW, NMMA, NP = cfg
    assert (NBUF - 1) * NP <= 63  # 6-bit vmcnt
    issued, barrier_of, waits = replay(NBUF, KPW, NMMA, NP, nk)
    assert sorted(issued) == [(t, q) for t in range(nk) for q in range(NP)]  # every piece of every K-tile, exactly once
    last = {}
    for idx, (t, _) in enumerate(issued):
        last[t] = idx
    for t, arg, at in waits:
        need = t + 1                                 # the tile the readers are released onto
        assert last[need] < at                       # all of it has been issued ...
        younger = at - 1 - last[need]                # ... and this many pieces were issued behind its last one
        assert arg <= younger, (t, arg, younger)     # in-order retirement: at most `arg` outstanding => tile `need` has landed
        if (NBUF - 1) * NP <= 3 * NP:                # (the kernel's wait helper counts up to three tiles: always the case)
            assert arg == younger, (t, arg, younger)  # ... and not a piece more is waited for than necessary
    for idx, (tile, _) in enumerate(issued):        # a refill of tile r's buffer comes behind the rendezvous that ended tile r - NBUF
        if tile >= NBUF:
            assert idx >= barrier_of[tile - NBUF], (tile, idx)
            assert idx < barrier_of.get(tile - 1, len(issued) + 1)  # and is in flight before the wait that needs it


# ---------------------------------------------------------------------------------------------------------------- LDS image of a K-tile
def chan_pos(i):  # gemm8p.hip: MFMA A-operand row i of a 32-channel block holds channel chan_pos(i)
    return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3)


B128_GROUPS = [  # lanes a ds_read_b128 serves in one LDS cycle (MI355X_MICROARCH.md, LDS table)
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


@pytest.mark.parametrize("waves", [4, 6, 8, 12, 16])
def test_k_tile_image_written_by_dma_is_what_the_fragment_reads_expect_and_bank_conflict_free(waves):
    """A K-tile buffer is rows of 128 bytes (64 bf16 of K), eight 16-byte chunks per row.  The DMA destination is lane-linear, so the
    16-byte XOR swizzle is applied on the SOURCE side (`schunk`); the fragment reads undo it (`xo` / `wo`).  Replay both address
    computations: every (row, K-chunk) must be found where the reader looks, and the sixteen lanes a ds_read_b128 serves per LDS
    cycle must fall on sixty-four distinct banks (SQ_LDS_BANK_CONFLICT = 0 in profiles/pmc_sq_r02)."""
    PR = 8 * waves                                    # rows per DMA piece of the whole workgroup
    rows = 2 * PR
    where = {}                                        # (row, source chunk) -> LDS byte offset
    for piece in range(rows // PR):
        for wave in range(waves):
            for lane in range(64):
                srow = wave * 8 + (lane >> 3)
                schunk = (lane & 7) ^ ((srow >> 1) & 7)
                row = piece * PR + srow               # piece i adds i * PR rows to source and destination alike
                where[(row, schunk)] = piece * PR * 128 + wave * 1024 + lane * 16
    assert len(where) == rows * 8 and sorted(where.values()) == list(range(0, rows * 128, 16))  # a bijection onto the buffer
    for operand_row in (lambda l31: l31, chan_pos):   # token rows in MFMA column order, channel rows through chan_pos
        for base in range(0, rows, 32):
            for ks in range(4):
                offs = {}
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    r = operand_row(l31)
                    off = (base + r) * 128 + (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4)
                    assert where[(base + r, 2 * ks + hh)] == off          # the reader finds K-chunk 2 ks + hh of its row
                    offs[lane] = off
                for grp in B128_GROUPS:
                    banks = [((offs[l] >> 2) + d) % 64 for l in grp for d in range(4)]
                    assert len(set(banks)) == 64                          # one LDS cycle per lane group
