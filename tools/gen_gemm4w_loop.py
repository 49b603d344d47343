#!/usr/bin/env python
"""Writes custom-diffusion360_amd/csrc/gemm4w_loop.inc: the hand-scheduled K loop of the 256 x 256 tile as FOUR waves of 128 x 128
(one per SIMD, 256 accumulator registers each) -- gemm8p.hip's arrangement <2, 2, 4, 4, 2, 1, 0, EPI 0 | 1>.

Why a generated instruction stream: with one wave per SIMD nothing hides a bubble, so the order of the 64 MFMAs, 32 fragment reads and
the operand traffic of a K-tile IS the performance; hipcc's schedule of the same loop (branches around every piece, reads sunk behind the
MFMA run) measured slower than the eight-wave arrangement.

What was measured on the way (tools/probe/gemm4w_ab.py whatif, 8192^3, one box, us): MFMAs alone 683 (= hipBLASLt's 681 on that box);
+ the 32 fragment reads of a K-tile 716; + operands global -> registers -> ds_write_b128 -> LDS (one tile of slack per piece, stores
staggered over the waves) 868.  A buffer_load_dwordx4 into registers costs ~17 cycles of the wave's MFMA stream however it is placed
(the 1 KiB return occupies the register file the matrix pipe lives on), a ds_write_b128 ~7: operands have to travel by LDS-DMA.

LDS-DMA with two 64 KB buffers leaves a piece between a quarter of a tile and one tile to cross L2 -> LDS (a buffer can only be refilled
once its tile has been multiplied).  Here the 160 KB are a ring of FIVE 32 KB slots holding operand tiles in the order X0 W0 X1 W1 X2 ...
(operand tile n in slot n % 5).  The barrier of K-tile t (behind k-step 2) releases the slots of X_t and W_t; W_{t+2} goes into the first
at once (eight pieces per wave behind the first MFMAs of k-step 3: it is needed one tile later), X_{t+3} into the second with two tiles
of slack (behind k-step 0 of the next tile).  The slot pattern repeats every five tiles: the loop is unrolled five times, every LDS
address an immediate.

Operands: %0..%15 accumulators acc[nb][mb] (index nb * 4 + mb, AGPRs); %16 / %17 per-lane source byte offsets of the token / channel
operand (row srow of the tile + swizzled chunk, as gemm8p's LDS-DMA uses them); %18 wave * 1024 (SGPR: LDS offset of the wave's share of
a piece); %19 / %20 fragment read offsets of k-step 0 inside a slot (token / channel operand); %21 / %22 buffer descriptors; %23 / %24 byte
step between pieces (32 rows); %25 number of K-tiles."""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "custom-diffusion360_amd", "csrc", "gemm4w_loop.inc")

SLOT = 32768
FW = (192, 224)             # channel fragments of set 0 / 1: 4 blocks x 4
FX = (208, 240)             # token fragments of set 0 / 1
VROWX, VROWW = 160, 168     # source offsets of the 8 pieces of a token / channel operand tile (row part; the K part is the SGPR offset)
VOOB = 176
VXA, VWA = 128, 140         # fragment addresses: [k-step 0..3][64 KB half 0..2] -> VXA + 3 * ks + h
S_CNT, S_K, S_KMAX, S_T, S_T1 = 88, 89, 90, 91, 92
CLOBBER_V = list(range(128, 256))
CLOBBER_S = [S_CNT, S_K, S_KMAX, S_T, S_T1]


def vr(a, n=4):
    return f"v[{a}:{a + n - 1}]"


VARIANT = 0


def dma(op, i, slot, sk):
    """piece i (0..7) of an operand tile (op 0 = FIRST operand of the ring order, 1 = second) at K byte offset s<sk> -> ring slot `slot`"""
    chan = (op == 1) != bool(SWAP)  # is this the channel (weight) operand?
    rows, rsrc = (VROWW if chan else VROWX) + i, ("%22" if chan else "%21")
    if VARIANT & 64:    # what-if: every lane out of range (zeros into the LDS, nothing fetched)
        return [f"s_add_u32 m0, %18, {hex(slot * SLOT + i * 4096)}", "s_nop 0", f"buffer_load_dwordx4 v{VOOB}, {rsrc}, s{sk} offen lds"]
    if VARIANT & 128:   # what-if: issued with no lane active
        return [f"s_add_u32 m0, %18, {hex(slot * SLOT + i * 4096)}", "s_mov_b64 exec, 0", f"buffer_load_dwordx4 v{rows}, {rsrc}, s{sk} offen lds", "s_mov_b64 exec, -1"]
    return [f"s_add_u32 m0, %18, {hex(slot * SLOT + i * 4096)}", "s_nop 0", f"buffer_load_dwordx4 v{rows}, {rsrc}, s{sk} offen lds"]


def frag_reads(ks, dst_set, fslot, sslot):
    """the eight reads of k-step ks (first / second operand tile of the ring order in slots fslot / sslot) into fragment set dst_set"""
    xslot, wslot = (sslot, fslot) if SWAP else (fslot, sslot)
    def addr(base, slot, blk):
        return f"v{base + 3 * ks + (slot >> 1)} offset:{(slot & 1) * SLOT + blk * 4096}"
    r = [f"ds_read_b128 {vr(FW[dst_set])}, {addr(VWA, wslot, 0)}", f"ds_read_b128 {vr(FX[dst_set])}, {addr(VXA, xslot, 0)}"]
    for nb in range(1, 4):
        r.append(f"ds_read_b128 {vr(FW[dst_set] + 4 * nb)}, {addr(VWA, wslot, nb)}")
    for mb in range(1, 4):
        r.append(f"ds_read_b128 {vr(FX[dst_set] + 4 * mb)}, {addr(VXA, xslot, mb)}")
    return r


# Where the 16 pieces a wave requests per K-tile go: phase (k-step) -> [(MFMA slot inside the phase, what)], what = "W1" (pieces 4..7 of
# W_{t+1}), "X2" (X_{t+2}), "W2" (pieces 0..3 of W_{t+2}); only slots without a fragment read (8..15), four per phase: the texture
# address path takes ~21 cycles per 1 KiB piece, and four waves that queue eight pieces each behind consecutive MFMAs stall at issue
# (measured: + 7 % per group of eight, whether anybody waits for the data or not).
SCHED = os.environ.get("CD360_G4_SCHED", "uniform")
SWAP = int(os.environ.get("CD360_G4_SWAP", "1"))  # 1: ring order W0 X0 W1 X1 ... (the channel operand = the weights, the colder one, gets the two tiles of slack)


def schedule():
    if SCHED == "burst":   # round-6 first cut: W_{t+2} behind the first MFMAs after the barrier, X_{t+2} behind k-step 0's reads
        return {0: [(8 + i, ("X2", i)) for i in range(8)], 3: [(i, ("W2", i)) for i in range(8)]}
    return {0: [(9 + 2 * i, ("W1", 4 + i)) for i in range(4)], 1: [(9 + 2 * i, ("X2", i)) for i in range(4)],
            2: [(9 + 2 * i, ("X2", 4 + i)) for i in range(4)], 3: [(9 + 2 * i, ("W2", i)) for i in range(4)]}


def tile(u, variant):
    """K-tile t = u (mod 5): X_t in slot 2u % 5, W_t in (2u + 1) % 5.  s_k = K byte offset of tile t + 2; s_t / s_t1: the offsets of
    tiles t + 2 / t + 1 clamped to the last tile."""
    xs, ws = (2 * u) % 5, (2 * u + 1) % 5
    xn, wn = (2 * u + 2) % 5, (2 * u + 3) % 5          # tile t + 1
    x2_slot = (2 * u + 4) % 5                          # X_{t+2}: the slot W_{t-1} left at the last barrier
    sched = schedule()
    burst = SCHED == "burst"
    t = [f"s_min_u32 s{S_T}, s{S_K}, s{S_KMAX}", f"s_sub_u32 s{S_T1}, s{S_K}, 0x80", f"s_min_u32 s{S_T1}, s{S_T1}, s{S_KMAX}", "s_waitcnt lgkmcnt(0)"]
    for p in range(4):
        s = p & 1
        reads = frag_reads(p + 1, s ^ 1, xs, ws) if p < 3 else frag_reads(0, s ^ 1, xn, wn)
        todo = dict(sched.get(p, []))
        for i in range(16):
            nb, mb = i % 4, i // 4
            t.append(f"v_mfma_f32_32x32x16_bf16 %{nb * 4 + mb}, {vr(FW[s] + 4 * nb)}, {vr(FX[s] + 4 * mb)}, %{nb * 4 + mb}")
            if i < 8 and not (variant & 2):
                t.append(reads[i])
            if i in todo and not (variant & 1):
                what, piece = todo[i]
                if what == "X2" and not (variant & 16):
                    t += dma(0, piece, x2_slot, S_T)
                elif what == "W2" and not (variant & 32):
                    t += dma(1, piece, xs, S_T)            # X_t's slot: free since this tile's barrier
                elif what == "W1" and not (variant & 32):
                    t += dma(1, piece, wn, S_T1)           # = X_{t-1}'s slot, where W_{t+1} lives
        if p == 2:
            # tile t + 1 complete (everything but the eight pieces of X_{t+2} requested in this tile), every wave past its reads of tile t
            if not (variant & (1 | 8 | 16 | 32 | 64 | 128)):
                t.append("s_waitcnt vmcnt(8)")
            t.append("s_waitcnt lgkmcnt(0)")
            if not (variant & 4):
                t.append("s_barrier")
        elif p < 2:
            t.append("s_waitcnt lgkmcnt(0)")
    t += [f"s_add_u32 s{S_K}, s{S_K}, 0x80"]
    return t


def body(variant=0):
    global VARIANT
    VARIANT = variant
    t = [f"v_mov_b32 v{VXA}, %19", f"v_mov_b32 v{VWA}, %20", f"v_mov_b32 v{VOOB}, 0x80000000"]
    # fragment addresses of the three 64 KB halves of the ring, per k-step (the chunk index of k-step ks is that of k-step 0 ^ 2 ks)
    for ks in range(1, 4):
        t += [f"v_xor_b32 v{VXA + 3 * ks}, {hex(32 * ks)}, v{VXA}", f"v_xor_b32 v{VWA + 3 * ks}, {hex(32 * ks)}, v{VWA}"]
    for ks in range(4):
        for h in (1, 2):
            t += [f"v_add_u32 v{VXA + 3 * ks + h}, {hex(65536 * h)}, v{VXA + 3 * ks}", f"v_add_u32 v{VWA + 3 * ks + h}, {hex(65536 * h)}, v{VWA + 3 * ks}"]
    t += [f"v_mov_b32 v{VROWX}, %16", f"v_mov_b32 v{VROWW}, %17"]
    for i in range(1, 8):
        t += [f"v_add_u32 v{VROWX + i}, %23, v{VROWX + i - 1}", f"v_add_u32 v{VROWW + i}, %24, v{VROWW + i - 1}"]
    # last K-tile's byte offset (requests past the end re-fetch it: never read, but never outside the operand either)
    t += [f"s_sub_u32 s{S_KMAX}, %25, 1", f"s_lshl_b32 s{S_KMAX}, s{S_KMAX}, 7"]
    # ---- prologue: X0 W0 X1 W1 -> slots 0..3 ----
    first_w1 = 8 if SCHED == "burst" else 4    # the uniform schedule requests pieces 4..7 of W1 in tile 0's k-step 0
    for n in range(4):
        t.append(f"s_min_u32 s{S_T}, {hex(128 * (n >> 1))}, s{S_KMAX}")
        for i in range(8 if n < 3 else first_w1):
            t += dma(n & 1, i, n, S_T)
    t += [f"s_waitcnt vmcnt({8 + first_w1})", "s_barrier"]
    t += frag_reads(0, 0, 0, 1)
    t += [f"s_movk_i32 s{S_K}, 0x100", f"s_mov_b32 s{S_CNT}, %25"]
    # ---- five K-tiles per trip, leaving after any of them ----
    t.append("1:")
    for u in range(5):
        t += tile(u, variant)
        t += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 9f" if u < 4 else "s_cbranch_scc0 1b"]
    t += ["9:", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15"]
    return t


def emit(name, lines):
    return f"#define {name} \\\n" + " \\\n".join(f'  "{ln}\\n"' for ln in lines) + "\n"


def render() -> str:
    txt = "// GENERATED by tools/gen_gemm4w_loop.py -- do not edit; see that file for the schedule and the operand list.\n"
    txt += emit("CD360_GEMM4W_LOOP", body(0))
    txt += ("#ifdef CD360_WHATIF  // probe builds: 1 = no operand traffic, 3 = no fragment reads either, 4 = no barrier, 8 = nobody waits for a piece,\n"
            "                     // 16 / 32 = without the relaxed / the tight operand's pieces, 64 = every piece out of range, 128 = pieces with no lane active\n")
    for v in (1, 3, 4, 8, 16, 32, 64, 128):
        txt += emit(f"CD360_GEMM4W_LOOP_V{v}", body(v))
    txt += "#endif\n"
    clob = ['"memory"', '"scc"'] + [f'"s{s}"' for s in CLOBBER_S] + [f'"v{v}"' for v in CLOBBER_V]
    txt += "#define CD360_GEMM4W_CLOBBERS " + ", ".join(clob) + "\n"
    return txt


def main():
    with open(OUT, "w") as f:
        f.write(render())
    print(OUT, len(body(0)), "instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
