#!/usr/bin/env python3
"""Generates radiosonde_auto_rx_amd/csrc/md_fast_gen.h: the hand-scheduled gfx950 instruction streams of the 2.4 Msps -> 48 kHz
decimator (D = 50 input samples per output, Q = 7 tap columns; one lane = one block of D input samples, 64 rows per tile).

MD_FAST_BODY_1   the walk over the 50 samples of one tile, registers of the accumulators / phase chosen by the compiler
                 (operands) — used where single tiles are processed from C++ (chunk ends, odd launch geometries).
MD50_LOOP_1      a wave's whole sequence of full tiles as ONE statement with hand-allocated registers: non-temporal loads of
                 tile t+2 into one of two staging sets (2 x 50 VGPRs), the sample walk of tile t out of LDS, the diagonal sum
                 with a one-value carry, the store of the 64 outputs, then vmcnt-counted wait + park of tile t+1 into LDS.

Why generated streams and not C++:
  * SMEM (taps) and LDS (raw samples) share lgkmcnt and SMEM returns out of order, so every wait is lgkmcnt(0): the loads of
    the NEXT pair of samples must be issued right AFTER the wait for the current pair, never before it, or the wave sleeps
    for a scalar-cache round trip per pair (the compiler's loop did exactly that);
  * v_pk_fma_f32 takes its wave-uniform tap from an aligned SGPR pair; op_sel picks the odd tap of a pair, so a tap row
    loaded with one s_load_dwordx8 is used in place (the compiler copied 16 SGPRs per pair of samples);
  * the mixer phase t = fl32(f0*n) advances by one v_add_f64 per sample (re-seeded with the exact product per row and tile)
    instead of index-add + v_mul_f64;
  * with one 12.8 KB tile per wave in flight (12 waves per CU) the launch is bound by bytes in flight, not by arithmetic
    (an EMPTY sample loop still took 0.9 ms per 4.9 GB): two tiles per wave must be in flight, which only fits the 168
    registers of a 3-waves-per-SIMD kernel when every register is placed by hand (the compiler spilled and — fatal for
    loads it cannot see — copied staging registers before their data had landed).

Per sample r (reference: demod_mod.c:484-493 IQ-DC, :737-750 mixer, :639-648/753 FIR):
  S1(r): x = cvt(raw); dcs += x; t = fl32(T); T += f0; (c, s) = cos/sin(2 pi fract(t))
  S2(r): z = x * (c + i s); acc[q] += (2^-15 W_q[r]) * z, q = 0..6
The 2^-15 of x = b/32768 sits in the tap table (scaling by a power of two commutes with every rounding).  The IQ-DC mean is
NOT subtracted per sample: the filter is linear, so y = sum W (x - avg) ex = sum W x ex - avg * E, where E[m] = sum_k w[k] ex[.]
is the filter's response to the bare mixer table — it depends on the channel's frequency only, has the table's period
(lut_len / D outputs) and is tabulated once per engine (k_md_etable); one complex multiply per OUTPUT replaces one packed
add per INPUT sample.  (avg is constant within a launch; the few outputs whose window straddles a change of avg between two
launches are corrected by md_dc_boundary.)  The difference to subtracting per sample is rounding noise of the order 1e-8.
Block B_r = S2(r-1) interleaved with S1(r), r = 0..D.  Before every even block: s_waitcnt lgkmcnt(0), then the loads
for the blocks after the next wait (raw pair p+1, tap rows 2p+1 and 2p+2).  Four tap-row register sets, two raw pairs.
"""
import os
import sys

D = 50
Q = 7
H = Q - 1
TAPSET = [36, 44, 52, 60]          # s[36:67]: four 8-dword tap rows


def pair(n):
    return f"v[{n}:{n + 1}]"


class Regs:
    """Register names of one instance of the sample walk.  v0 = first of 17 scratch VGPRs."""

    def __init__(self, v0, acc, dcs, T, row, f0, wt, msk=None):
        self.RAW = [pair(v0), pair(v0 + 2)]
        self.RAWC = [[f"v{v0}", f"v{v0 + 1}"], [f"v{v0 + 2}", f"v{v0 + 3}"]]
        self.X = [pair(v0 + 4), pair(v0 + 6)]
        self.XR = [f"v{v0 + 4}", f"v{v0 + 6}"]
        self.XI = [f"v{v0 + 5}", f"v{v0 + 7}"]
        self.CS = [pair(v0 + 8), pair(v0 + 10)]
        self.C = [f"v{v0 + 8}", f"v{v0 + 10}"]
        self.S = [f"v{v0 + 9}", f"v{v0 + 11}"]
        self.TT, self.Z, self.TP = pair(v0 + 12), pair(v0 + 14), f"v{v0 + 16}"
        self.acc, self.dcs, self.T, self.row, self.f0, self.wt, self.msk = acc, dcs, T, row, f0, wt, msk


RAW = False   # the scanner's front end in ONE pass over the input (round 4): double mixer phase like SCAN, NO mean anywhere in the loop — the outputs are the
              # raw sum W x ex, and every row's sum of raw samples (the IQ-DC sum of its D samples, exact integers) goes to a side array; the means of the
              # 1/32 s windows and y -= mean * E follow at the IF rate (k_dc_rows_to_segments, k_scan_dc_fold), where the input is 1/50 of the bytes
EXP = ""      # experiment variants (tools/ab_variants.sh): timing only, results are garbage
SCAN = False  # the scanner's front end (scan/dft_detect.c): mixer phase kept in DOUBLE (t = f0 * n, :1090-1093: fract in f64, then one rounding to f32)
              # and the IQ-DC mean of the row's 1/32 s window taken off every sample ((x - avg) ex, :579-588; the means come from a table, a
              # launch spans many windows) instead of folded out per output — same instruction count: the subtraction takes the slot of the
              # IQ-DC sum, which k_dc_seg_sums owns in that mode


def tap_fma(R, q, row, init=False):
    b = TAPSET[row % 4] + (q & ~1)
    if init:                                                  # first sample of a tile: acc = w * z (no zeroing of the accumulators)
        sel = "op_sel:[1,0] op_sel_hi:[1,1]" if q & 1 else "op_sel_hi:[0,1]"
        return f"v_pk_mul_f32 {R.acc[q]}, s[{b}:{b + 1}], {R.Z} {sel}"
    sel = "op_sel:[1,0,0] op_sel_hi:[1,1,1]" if q & 1 else "op_sel_hi:[0,1,1]"
    return f"v_pk_fma_f32 {R.acc[q]}, s[{b}:{b + 1}], {R.Z}, {R.acc[q]} {sel}"


def loads_after_wait(R, p):
    out = []
    if p + 1 <= (D - 1) // 2:
        out.append(f"ds_read_b64 {R.RAW[(p + 1) & 1]}, {R.row} offset:{8 * (p + 1)}")
    for k in (2 * p + 1, 2 * p + 2):
        if k <= D - 1:
            b = TAPSET[k % 4]
            out.append(f"s_load_dwordx8 s[{b}:{b + 7}], {R.wt}, 0x{32 * k:x}")
    return out


def block(R, r, init=False):
    s1 = r <= D - 1
    s2 = r >= 1
    e, o = r & 1, (r - 1) & 1
    raw = R.RAWC[(r >> 1) & 1][r & 1] if s1 else None
    I = {}                                                    # the instructions of the block by name
    if s1:
        I["cvt64"] = f"v_cvt_f32_f64 {R.TP}, {R.T}"
        I["add64"] = f"v_add_f64 {R.T}, {R.T}, {R.f0}"
        I["fract"] = f"v_fract_f32 {R.TP}, {R.TP}"
        if SCAN or RAW:      # Z is free between the last tap FMA of the previous block and this block's `z`
            I["cvt64"] = f"v_fract_f64 {R.Z}, {R.T}"
            I["fract"] = f"v_cvt_f32_f64 {R.TP}, {R.Z}"
        I["xr"] = f"v_cvt_f32_i32_sdwa {R.XR[e]}, sext({raw}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0"
        I["xi"] = f"v_cvt_f32_i32_sdwa {R.XI[e]}, sext({raw}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
        I["cos"] = f"v_cos_f32 {R.C[e]}, {R.TP}"
        I["sin"] = f"v_sin_f32 {R.S[e]}, {R.TP}"
        I["dcs"] = f"v_pk_fma_f32 {R.dcs}, {R.X[e]}, {R.msk}, {R.dcs}" if R.msk else f"v_pk_add_f32 {R.dcs}, {R.X[e]}, {R.dcs}"
        if init and r == 0: I["dcs"] = f"v_mov_b64 {R.dcs}, {R.X[e]}"
        if SCAN: I["dcs"] = f"v_pk_add_f32 {R.X[e]}, {R.X[e]}, {R.dcs}"      # R.dcs holds -32768 * mean of the row's window
    if s2:
        I["tt"] = f"v_pk_mul_f32 {R.TT}, {R.X[o]}, {R.CS[o]} op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        I["z"] = f"v_pk_fma_f32 {R.Z}, {R.X[o]}, {R.CS[o]}, {R.TT} op_sel_hi:[0,1,1]"
        for q in range(Q): I[f"a{q}"] = tap_fma(R, q, r - 1, init and r == 1)
    order = "cvt64 tt add64 fract z xr cos a0 xi sin a1 a2 dcs a3 a4 a5 a6"
    if "nodc" in EXP: order = order.replace(" dcs", "")
    if "nocmul" in EXP: order = order.replace(" tt ", " ").replace(" z ", " ")
    L = [I[k] for k in order.split() if k in I]
    if "nosincos" in EXP:
        L = [l.replace("v_cos_f32", "v_mov_b32").replace("v_sin_f32", "v_mov_b32") for l in L]
    if "nof64" in EXP:
        L = [l for l in L if not l.startswith("v_add_f64")]
        L = [f"v_mov_b32 {R.TP}, {R.XR[0]}" if l.startswith("v_cvt_f32_f64") else l for l in L]
    if "nofir" in EXP:
        L = [l for l in L if not any(l.startswith(f"v_pk_fma_f32 {R.acc[q]},") or l.startswith(f"v_pk_mul_f32 {R.acc[q]},") for q in range(1, Q))]
    return L


def walk(R, init=False, hook=()):
    """the 50 samples of one tile; ends with every LDS / SMEM access complete.  init: the accumulators and the DC sum start
    from the first sample instead of from their old values.  hook: instructions placed behind the first wait (by then every
    LDS operation issued before the walk has completed as well)"""
    if "empty" in EXP:
        return ["s_waitcnt lgkmcnt(0)"] + list(hook)
    L = [f"ds_read_b64 {R.RAW[0]}, {R.row}", f"s_load_dwordx8 s[{TAPSET[0]}:{TAPSET[0] + 7}], {R.wt}, 0x0"]
    for r in range(D + 1):
        if r % 2 == 0:
            L.append("s_waitcnt lgkmcnt(0)")
            L += loads_after_wait(R, r // 2)
            if r == 0: L += list(hook)
        L += block(R, r, init)
    if "noloads" in EXP:
        L = [l for l in L if not (l.startswith("ds_read") or l.startswith("s_load") or l.startswith("s_waitcnt"))]
    return L


# ---- the operand form (md_fast_tile in sonde_kernels.hip) -----------------------------------------------------------------
def body_operands():
    R = Regs(100, [f"%[a{q}]" for q in range(Q)], "%[dcs]", "%[T]", "%[row]", "%[f0]", "%[wt]", msk=None if (SCAN or RAW) else "%[msk]")
    return walk(R)


# ---- the whole tile loop of k_mix_decimate50 -------------------------------------------------------------------------------
# VGPRs v28..v167 belong to the statement (clobbers, MD50_CLOBBERS in sonde_kernels.hip); the compiler keeps its operands below.
ACC0 = 28                          # v[28:41]  P[row][0..6] (re, im)
CARRY = 42                         # v[42:43]  what earlier rows add to the first H outputs of the next tile
STAGE = [44, 94]                   # two staging sets: 12 x 4 + 2 VGPRs each
SCR = 144                          # v[144:160] scratch of the sample walk; reused by the last tile's diagonal sum
EREG = 162                         # v[162:163] E of the tile being walked (requested at its start, used with its deferred sum)
DCS = 164                          # v[164:165]
TREG = 166                         # v[166:167]
S_B1, S_B2, S_B3, S_TB = 68, 70, 72, 74      # tile base + 4096 / 8192 / 12288 bytes, tile base
S_T, S_TMP, S_OUT, S_JM = 76, 77, 78, 80     # tile counter, scratch, s[78:79] lanes whose outputs count, ring index of the tile
TILE_BYTES = 64 * D * 4
ACC = [pair(ACC0 + 2 * q) for q in range(Q)]


def fetch(st):
    """13 non-temporal loads of the tile at s[S_TB] into staging set st; advances s[S_TB] by one tile"""
    L = []
    for b, o in ((S_B1, 0x1000), (S_B2, 0x2000), (S_B3, 0x3000)):
        L += [f"s_add_u32 s{b}, s{S_TB}, 0x{o:x}", f"s_addc_u32 s{b + 1}, s{S_TB + 1}, 0"]
    for i in range(12):
        base = (S_TB, S_B1, S_B2)[i // 4]
        L.append(f"global_load_dwordx4 v[{st + 4 * i}:{st + 4 * i + 3}], %[voff16], s[{base}:{base + 1}] offset:{1024 * (i % 4)} nt")
    L.append(f"global_load_dwordx2 v[{st + 48}:{st + 49}], %[voff8], s[{S_B3}:{S_B3 + 1}] nt")
    L += [f"s_add_u32 s{S_TB}, s{S_TB}, 0x{TILE_BYTES:x}", f"s_addc_u32 s{S_TB + 1}, s{S_TB + 1}, 0"]
    return L


def park(st):
    L = [f"ds_write_b128 %[ldsw16], v[{st + 4 * i}:{st + 4 * i + 3}]" + (f" offset:{1024 * i}" if i else "") for i in range(12)]
    L.append(f"ds_write_b64 %[ldsw8], v[{st + 48}:{st + 49}] offset:12288")
    return L


def diag_issue(rb):
    """rotations of the P columns into v[rb..rb+11], y = P[.][H] + carry into v[rb+12:rb+13], byte offset of the lane's output in the
    ring into v[rb+14]; the carry restarts from zero.  y[j] = P[j][H] + sum_{q<H} P[j-(H-q)][q]: column q goes down by k = H-q lanes
    (bpermute address 4*lane + 256 - 4k; the LDS unit reads the source registers when the instruction issues)"""
    L = []
    for q in range(H):
        k = H - q
        L += [f"ds_bpermute_b32 v{rb + 2 * q}, %[lane4], v{ACC0 + 2 * q} offset:{256 - 4 * k}",
              f"ds_bpermute_b32 v{rb + 2 * q + 1}, %[lane4], v{ACC0 + 2 * q + 1} offset:{256 - 4 * k}"]
    if "nodiag" in EXP: L = []
    Y, T1 = rb + 12, rb + 14
    L += [f"v_pk_add_f32 {pair(Y)}, {ACC[H]}, {pair(CARRY)}", f"v_mov_b64 {pair(CARRY)}, 0",
          f"v_lshrrev_b32 v{T1}, 2, %[lane4]", f"v_add_u32 v{T1}, s{S_JM}, v{T1}", f"v_and_b32 v{T1}, %[rmask], v{T1}",
          f"v_lshlrev_b32 v{T1}, 3, v{T1}", f"s_add_i32 s{S_JM}, s{S_JM}, 64"]
    return L


def diag_finish(rb):
    """(after lgkmcnt(0), E landed) lanes >= k hold a row of the same tile -> its term of y; lanes < k hold rows 64-k+l -> the next
    tile's carry; then y -= avg * E (complex; %[navg] = (-avg.re, -avg.im)) and the store of the 64 outputs"""
    Y, T1, L = rb + 12, rb + 14, []
    for q in range(H):
        L += [f"s_mov_b32 exec_lo, 0x{(0xffffffff << (H - q)) & 0xffffffff:x}", f"v_pk_add_f32 {pair(Y)}, {pair(Y)}, {pair(rb + 2 * q)}"]
    for q in range(H):
        L += [f"s_mov_b64 exec, {(1 << (H - q)) - 1}", f"v_pk_add_f32 {pair(CARRY)}, {pair(CARRY)}, {pair(rb + 2 * q)}"]
    L += [f"s_mov_b64 exec, s[{S_OUT}:{S_OUT + 1}]"]
    if not (SCAN or RAW):
        L += [f"v_pk_fma_f32 {pair(Y)}, {pair(EREG)}, %[navg], {pair(Y)} op_sel_hi:[1,0,1]",
              f"v_pk_fma_f32 {pair(Y)}, {pair(EREG)}, %[navg], {pair(Y)} op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"]
    L += [f"global_store_dwordx2 v{T1}, {pair(Y)}, %[yout]",
          "s_mov_b64 exec, -1", f"s_mov_b64 s[{S_OUT}:{S_OUT + 1}], -1"]
    if "nostore" in EXP: L = [l for l in L if not l.startswith("global_store")]
    if "sc1store" in EXP: L = [l + " sc1" if l.startswith("global_store") else l for l in L]
    if "sc01store" in EXP: L = [l + " sc0 sc1" if l.startswith("global_store") else l for l in L]
    return L


def gen_loop():
    R = Regs(SCR, ACC, pair(DCS), pair(TREG), "%[row]", "%[f0]", "%[wt]")
    T1, A, B = SCR + 14, SCR + 15, SCR + 16
    L = [f"s_mov_b64 s[{S_TB}:{S_TB + 1}], %[tb]", f"s_mov_b64 s[{S_OUT}:{S_OUT + 1}], %[outmask]", f"s_mov_b32 s{S_JM}, %[jm]",
         f"s_mov_b32 s{S_T}, 0", f"v_mov_b64 {pair(CARRY)}, %[carry]"]
    if SCAN: L.append(f"v_mov_b64 {pair(DCS)}, %[navg0]")        # -32768 * mean of the window of the lane's row in the first tile
    # tile 0 -> LDS (the wave's only exposed load latency); tile 1 on its way
    L += fetch(STAGE[0])
    L += ["s_cmp_gt_i32 %[nfull], 1", "s_cbranch_scc0 10f"]
    L += fetch(STAGE[1])
    L += ["s_waitcnt vmcnt(13)", "s_branch 11f", "10:", "s_waitcnt vmcnt(0)", "11:"]
    L += park(STAGE[0])
    # Tile t lives in staging set t & 1 until it is parked.  Half h of the loop walks a tile with t & 1 == h: set h has just been
    # parked, so its registers take the rotated columns of the PREVIOUS tile; those sums and the store of its outputs sit behind
    # the walk's first wait (one LDS round trip per tile instead of three), then this tile's E and tile t+2 (into set h) are requested.
    # %[e] = the lane's block as an index into the mixer table's period (table index = 50 e): phase seed and E index.
    for h in (0, 1):
        lab = 20 + 10 * h
        hook = [f"s_cmp_eq_u32 s{S_T}, 0", f"s_cbranch_scc1 {lab + 1}f",
                # E of the previous tile: younger than it are only the 13 loads of tile t+1 — if those were requested
                f"s_add_i32 s{S_TMP}, s{S_T}, 1", f"s_cmp_lt_i32 s{S_TMP}, %[nfull]", f"s_cbranch_scc0 {lab + 5}f",
                "s_waitcnt vmcnt(13)", f"s_branch {lab + 6}f", f"{lab + 5}:", "s_waitcnt vmcnt(0)", f"{lab + 6}:"]
        hook += diag_finish(STAGE[h]) + [f"{lab + 1}:"]
        if SCAN:
            # the mean for the lane's row in the NEXT tile (row + 64): window k = (row + 64 + off) / B through a float reciprocal — row + off < 2^24 is
            # exact in f32 and (n + 0.5) / B is at least 0.5 / B away from an integer, far more than the rounding of the product — clamped to the table
            hook += [f"v_add_u32 v{T1}, %[segoff64], %[jrow]", f"v_cvt_f32_u32 v{T1}, v{T1}", f"v_add_f32 v{T1}, 0.5, v{T1}", f"v_mul_f32 v{T1}, %[rcpB], v{T1}",
                     f"v_cvt_u32_f32 v{T1}, v{T1}", f"v_min_u32 v{T1}, %[segmax], v{T1}", f"v_lshlrev_b32 v{T1}, 3, v{T1}",
                     f"global_load_dwordx2 {pair(EREG)}, v{T1}, %[dcseg]"]
        elif RAW: pass                                        # neither a mean nor E: the fold happens at the IF rate
        elif "noE" not in EXP: hook += [f"v_lshlrev_b32 v{T1}, 3, %[e]", f"global_load_dwordx2 {pair(EREG)}, v{T1}, %[etab]"]
        fe = [f"s_add_i32 s{S_TMP}, s{S_T}, 2", f"s_cmp_lt_i32 s{S_TMP}, %[nfull]", f"s_cbranch_scc0 {lab + 2}f"] + fetch(STAGE[h]) + [f"{lab + 2}:"]
        if "latefetch" not in EXP: hook += fe
        L += [f"{lab}:", f"v_mul_u32_u24 v{T1}, 50, %[e]", f"v_cvt_f64_u32 {pair(TREG)}, v{T1}", f"v_mul_f64 {pair(TREG)}, {pair(TREG)}, %[f0]"]
        L += walk(R, init=True, hook=hook)
        if "latefetch" in EXP: L += fe
        # the lane's block in the next tile: e = (e + 64) mod P; IQ-DC sums of the rows that count
        L += [f"v_add_u32 %[e], 64, %[e]", f"v_subrev_u32 v{T1}, %[P], %[e]", f"v_min_u32 %[e], v{T1}, %[e]"]
        if SCAN: L += [f"v_add_u32 %[jrow], 64, %[jrow]"]
        elif RAW:
            # the rows' sums (re, im; sums of 50 int16: exact in f32 and in i32) -> bsum[jrow], for the rows whose outputs count (a halo row is the
            # previous segment's).  One store more in flight than the vmcnt(13) waits were counted for: they then also wait for the first load of tile
            # t + 2 — harmless
            L += [f"v_cvt_i32_f32 v{SCR + 12}, v{DCS}", f"v_cvt_i32_f32 v{SCR + 13}, v{DCS + 1}", f"v_lshlrev_b32 v{T1}, 3, %[jrow]",
                  f"s_mov_b64 exec, s[{S_OUT}:{S_OUT + 1}]", f"global_store_dwordx2 v{T1}, {pair(SCR + 12)}, %[bsum]", "s_mov_b64 exec, -1",
                  f"v_add_u32 %[jrow], 64, %[jrow]"]
        else: L += [f"v_cvt_i32_f32 v{A}, v{DCS}", f"v_cvt_i32_f32 v{B}, v{DCS + 1}",
                    f"s_mov_b64 exec, s[{S_OUT}:{S_OUT + 1}]", f"v_add_u32 %[sx], %[sx], v{A}", f"v_add_u32 %[sy], %[sy], v{B}", "s_mov_b64 exec, -1"]
        L += [f"s_add_i32 s{S_T}, s{S_T}, 1", f"s_cmp_ge_i32 s{S_T}, %[nfull]", "s_cbranch_scc1 90f"]
        # tile t+1: its loads are followed by this tile's E load and 13 loads of tile t+2 (if requested)
        L += [f"s_add_i32 s{S_TMP}, s{S_T}, 1", f"s_cmp_lt_i32 s{S_TMP}, %[nfull]", f"s_cbranch_scc0 {lab + 3}f",
              "s_waitcnt vmcnt(13)", f"s_branch {lab + 4}f", f"{lab + 3}:", "s_waitcnt vmcnt(0)", f"{lab + 4}:"]
        L += park(STAGE[1 - h])
        L += diag_issue(STAGE[1 - h])
        if SCAN:      # the next tile's means have landed (they are older than the loads the last wait left outstanding)
            L += [f"v_mul_f32 v{DCS}, 0xc7000000, v{EREG}", f"v_mul_f32 v{DCS + 1}, 0xc7000000, v{EREG + 1}"]
    L += ["s_branch 20b", "90:"]
    # the wave's last full tile: nothing follows, the walk's scratch registers take the rotations
    L += diag_issue(SCR) + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + diag_finish(SCR)
    L += [f"v_mov_b64 %[o{q}], {ACC[q]}" for q in range(Q)] + [f"v_mov_b64 %[carry], {pair(CARRY)}", "s_nop 1"]
    return L


def as_macro(name, lines):
    return f"#define {name} \\\n" + " \\\n".join(f'    "{l}\\n\\t"' for l in lines) + "\n"


def main():
    """md_fast_gen.h: MD_FAST_BODY_1 / MD50_LOOP_1 = production; with --experiments <spec>... also _2.. (timing only)"""
    global EXP
    text = "// generated by tools/gen_md_fast.py — do not edit (tests/test_generated_sources.py checks it is in sync)\n"
    text += as_macro("MD_FAST_BODY_1", body_operands()) + as_macro("MD50_LOOP_1", gen_loop())
    global SCAN
    SCAN = True
    text += "// the scanner's front end: double mixer phase, IQ-DC mean of the row's window off every sample (SCAN in tools/gen_md_fast.py)\n"
    text += as_macro("MD_FAST_BODY_S", body_operands()) + as_macro("MD50_LOOP_S", gen_loop())
    SCAN = False
    global RAW
    RAW = True
    text += "// the scanner's front end in one pass: double mixer phase, no mean in the loop, the rows' raw sums to a side array (RAW in tools/gen_md_fast.py)\n"
    text += as_macro("MD_FAST_BODY_R", body_operands()) + as_macro("MD50_LOOP_R", gen_loop())
    RAW = False
    if len(sys.argv) > 1 and sys.argv[1] == "--experiments":
        for k, e in enumerate(sys.argv[2:], 2):
            EXP = e
            text += f"// experiment {k}: {e}\n" + as_macro(f"MD50_LOOP_{k}", gen_loop())
    if len(sys.argv) > 1 and sys.argv[1] == "--print":
        sys.stdout.write(text); return
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "radiosonde_auto_rx_amd", "csrc")
    open(os.path.join(base, "md_fast_gen.h"), "w").write(text)


if __name__ == "__main__":
    main()
