// sonde_kernels.hip — CDNA4 (gfx950) kernels of the radiosonde IQ demodulation engine.
//
//   k_mix_decimate   cs16 -> (x - dc) * exp-LUT -> decM:1 Blackman-sinc FIR -> cf32 @ IF rate
//                    (reference: f32read_cblock + LUT mixer + lowpass, demod_mod.c:463-508,737-754,639-648)
//                    With D = decM and Q = ceil(taps/D):  y[m] = sum_q P[m-(Q-1)+q][q],  P[j][q] = sum_r W_q[r] * z[D*j + r];
//                    one lane owns one block j and walks its D samples with packed f32 FMAs on (re, im); the 2.4 Msps -> 48 kHz
//                    case runs a generated, hand-scheduled instruction stream (md_fast_gen.h, tools/gen_md_fast.py).
//   k_if_chain       IF low-pass, conj-product FM discriminator, two-tone sliding correlator, FM low-pass
//                    (demod_mod.c:765-808,843-852)
//   k_sync_plan,     the header search of find_header with the reference's own 8192-point transform per window (getCorrDFT,
//   k_sync_window_fft demod_mod.c:148-225): the windows the sync will ask for are planned, then evaluated one workgroup each
//   k_header_corr    the same correlation in the time domain for every end sample — used by --dc engines only (zero-mean
//                    windows and the FM-stream fallback are evaluated from the correlation ring)
//   k_framesync      per-channel find_header / headcmp / read_softbit2p state machine + RS41 byte framing
//                    + RS(255,231) syndromes (demod_mod.c:1533-1617,870-938,1087-1175; rs41mod.c:2900-2962)
//   k_dc_update      running IQ-DC mean hand-over at segment boundaries (demod_mod.c:495-504)
//
// One wave = 64 lanes everywhere.  No CUDA compatibility paths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sonde_dev.h"
#include "md_fast_gen.h"
#include "sonde_rs_dev.h"
#include <cstdlib>

typedef short  short2v __attribute__((ext_vector_type(2)));

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// k_mix_decimate
// ------------------------------------------------------------------------------------------------
// y[m] = sum_k w[k] z[D(m+1)-T+k],  z[n] = (x[n]-dc) * ex[n].  With Q = ceil(T/D) and the taps front-padded to
// Q*D:  y[m] = sum_q P[m-(Q-1)+q][q],  P[j][q] = sum_{r<D} W_q[r] z[D j + r]   (block j = D input samples).
//
// One LANE owns one block j ("row"); a wave walks its 64 rows through r = 0..D-1 in lock step, so
//   * the tap W_q[r] is wave-uniform -> scalar operand, no LDS traffic, no cross-lane reduction;
//   * each lane keeps its 2*Q partial sums P[j][0..Q-1] (re, im) in registers: 2*Q FMAs per sample;
//   * the diagonal sum over q is Q-1 lane shifts (ds_bpermute) with a carry from the previous tile.
// (A first version ran P as an f32 MFMA block product; on gfx950 v_mfma_f32_16x16x4_f32 executes on the same
//  fp32 ALUs as the VALU — tools/probes/mfma_valu_overlap.hip: MFMA 0.64 ms + VALU 0.64 ms = 1.45 ms when
//  mixed — and only Q = 7 of its 16 columns carried work, so the plain FMA form is 2.3x cheaper.)
// The 64*D*4 bytes of a tile are fetched with fully coalesced 16-byte-per-lane loads one tile ahead (registers)
// and parked in the wave's private LDS slice (stride-D reads are at most 2-way bank conflicted); no workgroup
// barrier anywhere.  The mixer phasor ex[n] = cexp(2 pi i fl32(f0 n)) of the reference's float32-phase table
// (demod_mod.c:1292-1295) is evaluated on the fly: fl32(f0*n) bit-exactly in f64, sin/cos by the hardware
// revolutions-input units (max abs error 2e-7 vs the table, tools/probes/sincos_probe.hip).
#define MD_ROWS   64          // blocks per tile = lanes
#define MD_NVMAX  16          // 16-byte loads per lane per tile = ceil(D/4); D <= 64

struct __attribute__((packed, aligned(4))) u32x4_u { uint32_t x, y, z, w; };

typedef float float2v __attribute__((ext_vector_type(2)));

// One tile: every lane walks the D samples of its row.  Measured issue costs on gfx950 (tools/probes/valu_rates.hip,
// cycles per wave64 instruction): v_fma_f32 3.2, v_pk_fma_f32 5.2, v_mul/add_f64 4.7, v_sin/cos_f32 10.1 — the
// kernel is VALU-bound, so the FIR uses packed FMAs on (re, im) pairs and everything else is kept to the minimum
// number of instructions.
// z = u * (c + i s) as two packed ops: t = (-u.y s, u.y c); z = (u.x c, u.x s) + t.  op_sel / neg_lo pick and negate the
// halves, so no register shuffling is needed (the compiler's own lowering costs two extra moves per sample).
__device__ __forceinline__ float2v cmul_pk(float2v u, float2v cs) {
    float2v t, z;
    // early-clobber outputs: a packed op whose op_sel crosses halves must not write a register pair it still reads.
    // s_nop: cs comes straight from v_cos/v_sin, and gfx940+ needs one wait state between a transcendental result and
    // its first VALU use — the compiler's hazard recognizer does not look into inline asm (seen as a corrupted real part).
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(t) : "v"(u), "v"(cs));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=&v"(z) : "v"(u), "v"(cs), "v"(t));
    return z;
}

// one input sample of a row: IQ-DC sums, DC removal, mixer phasor, Q packed tap FMAs
template <int Q_T, bool WRAP, bool PH64>
__device__ __forceinline__ void md_step(const uint32_t raw, const int r, const float *w, const float2v navg, const float2v msk,
                                        const double f0, double &ndrun, const uint32_t rown, const uint32_t towrap, const uint32_t L,
                                        float2v (&acc)[Q_T], float2v &dcs) {
    const float2v scale = { 3.0517578125e-05f, 3.0517578125e-05f };
    const float2v xf = { (float)(int)(short)(raw & 0xffffu), (float)(((int)raw) >> 16) };
    // IQ-DC sums: int16 values are exact in f32 and a tile row sums to < 2^24, so the float sum is exact
    dcs = __builtin_elementwise_fma(xf, msk, dcs);
    // x = b/32768.0 is exact -> one rounding for (x - avg) as in the reference (demod_mod.c:484-493)
    const float2v u = __builtin_elementwise_fma(xf, scale, navg);
    // ex[n], n = rown + r (mod L): t = fl32(f0*n) exactly as the reference's table was built
    double nd;
    if (!WRAP) { nd = ndrun; ndrun += 1.0; }                  // running index: integers stay exact in f64, one add per sample
    else nd = (double)(rown + (uint32_t)r - (((uint32_t)r >= towrap) ? L : 0u)) + ndrun;      // (WRAP: ndrun carries nd_base)
    // demod_mod.c keeps the table phase in a float (t = fl32(f0*n)); dft_detect.c:1090-1093 keeps it in a double
    float fr;
    if (PH64) fr = (float)__builtin_amdgcn_fract(f0 * nd);
    else      fr = __builtin_amdgcn_fractf((float)(f0 * nd));
    const float2v cs = { __builtin_amdgcn_cosf(fr), __builtin_amdgcn_sinf(fr) };
    const float2v z = cmul_pk(u, cs);                         // z = u * ex[n]  (demod_mod.c:744)
#pragma unroll
    for (int q = 0; q < Q_T; q++) {                           // P[row][q] += W_q[r] * z: one packed FMA per tap
        const float2v wq = { w[q], w[q] };
        acc[q] = __builtin_elementwise_fma(wq, z, acc[q]);
    }
}

template <int Q_T, bool WRAP, bool PH64, int D_T>
__device__ __forceinline__ void md_rows(const uint32_t *row, int D, const float *wt, float2 avg, double f0,
                                        uint32_t rown, uint32_t L, float dcmask, double nd_base,
                                        float2v (&acc)[Q_T], float2v &dcs) {
    const uint32_t towrap = L - rown;
    double nd0 = WRAP ? nd_base : (double)rown + nd_base;
    const float2v navg = { -avg.x, -avg.y }, msk = { dcmask, dcmask };
    if (D_T > 0) D = D_T;                                     // compile-time trip count
    // two samples per iteration (written out: the inline asm of the complex multiply counts as convergent, which
    // forbids a compiler-generated remainder loop); the taps of the next pair are fetched one iteration ahead
    float wa[Q_T], wb[Q_T];
#pragma unroll
    for (int q = 0; q < Q_T; q++) { wa[q] = wt[q]; wb[q] = wt[8 * (D > 1 ? 1 : 0) + q]; }
    int r = 0;
    for (; r + 1 < D; r += 2) {
        float w0[Q_T], w1[Q_T];
#pragma unroll
        for (int q = 0; q < Q_T; q++) { w0[q] = wa[q]; w1[q] = wb[q]; }
        {
            const float *pa = wt + 8 * (r + 2 < D ? r + 2 : D - 1), *pb = wt + 8 * (r + 3 < D ? r + 3 : D - 1);
#pragma unroll
            for (int q = 0; q < Q_T; q++) { wa[q] = pa[q]; wb[q] = pb[q]; }
        }
        md_step<Q_T, WRAP, PH64>(row[r], r, w0, navg, msk, f0, nd0, rown, towrap, L, acc, dcs);
        md_step<Q_T, WRAP, PH64>(row[r + 1], r + 1, w1, navg, msk, f0, nd0, rown, towrap, L, acc, dcs);
    }
    if (r < D) md_step<Q_T, WRAP, PH64>(row[r], r, wa, navg, msk, f0, nd0, rown, towrap, L, acc, dcs);
}

// Hand-scheduled walk of one 64-row tile for D = 50, Q = 7 (tools/gen_md_fast.py has the schedule and the reasons).  Differences
// to md_step<7, false, false>:
//   * the mixer phase advances by T += f0 in double from the exact product f0*n of the row's first sample (<= 49 additions: the
//     float rounding of t differs from fl32(f0*n) for about one sample in 1e7, far inside the 1e-6 stream tolerance);
//   * the IQ-DC mean is not subtracted per sample: P'[j][q] = sum_r W_q[r] x ex, and the caller subtracts avg * E[m] per OUTPUT
//     (E = the filter's response to the bare mixer table, k_md_etable; md_dc_correct below).  The P tail between calls holds P'.
// row_lds: byte address of the lane's row in LDS; wt: [50][8] tap rows * 2^-15 in global memory (scalar loads).
#define MD_FAST_CLOBBERS "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
        "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
        "v116", "memory"
#define MD_FAST_ASM(BODY) asm volatile(BODY \
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), \
          [a6] "+v"(acc[6]), [dcs] "+v"(dcs), [T] "+v"(T) \
        : [row] "v"(row_lds), [f0] "s"(f0), [msk] "v"(msk), [wt] "s"(wt) \
        : MD_FAST_CLOBBERS)
__device__ __forceinline__ void md_fast_tile(uint32_t row_lds, const float *wt, double f0, double T, float2v msk, float2v (&acc)[7], float2v &dcs) {
    MD_FAST_ASM(MD_FAST_BODY_1);
}
// the scanner's form (MD_FAST_BODY_S): double mixer phase; navg = -32768 * the IQ-DC mean of the lane's row, taken off every sample
__device__ __forceinline__ void md_fast_tile_s(uint32_t row_lds, const float *wt, double f0, double T, float2v navg, float2v (&acc)[7]) {
    float2v dcs = navg;
    asm volatile(MD_FAST_BODY_S
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]),
          [a6] "+v"(acc[6]), [T] "+v"(T)
        : [row] "v"(row_lds), [f0] "s"(f0), [dcs] "v"(dcs), [wt] "s"(wt)
        : MD_FAST_CLOBBERS);
}
// the one-pass form (MD_FAST_BODY_R): double mixer phase, no mean; dcs += the row's raw samples
__device__ __forceinline__ void md_fast_tile_r(uint32_t row_lds, const float *wt, double f0, double T, float2v (&acc)[7], float2v &dcs) {
    asm volatile(MD_FAST_BODY_R
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]),
          [a6] "+v"(acc[6]), [dcs] "+v"(dcs), [T] "+v"(T)
        : [row] "v"(row_lds), [f0] "s"(f0), [wt] "s"(wt)
        : MD_FAST_CLOBBERS);
}
// y -= avg * E (complex)
__device__ __forceinline__ float2v md_dc_correct(float2v y, float2 avg, float2 E) {
    y.x = fmaf(-avg.x, E.x, y.x); y.y = fmaf(-avg.x, E.y, y.y);
    y.x = fmaf(avg.y, E.y, y.x);  y.y = fmaf(-avg.y, E.x, y.y);
    return y;
}
// -avg as an SGPR pair (avg is per channel, i.e. wave-uniform)
__device__ __forceinline__ uint64_t md_navg_sgpr(float2 avg) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(__float_as_uint(-avg.x));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(__float_as_uint(-avg.y));
    return ((uint64_t)hi << 32) | lo;
}

// mixer phasor of table index n as the decimator evaluates it: ex[n] = cexp(2 pi i fl32(f0 n))  (demod_mod.c:1290-1295)
__device__ __forceinline__ float2 md_phasor(double f0, uint32_t n) {
    const float fr = __builtin_amdgcn_fractf((float)(f0 * (double)n));
    return make_float2(__builtin_amdgcn_cosf(fr), __builtin_amdgcn_sinf(fr));
}
// the scanner's table: ex[n] = cexp(2 pi i f0 n) with the phase in double, reduced before the single rounding (dft_detect.c:1090-1093)
__device__ __forceinline__ float2 md_phasor64(double f0, uint32_t n) {
    const float fr = (float)__builtin_amdgcn_fract(f0 * (double)n);
    return make_float2(__builtin_amdgcn_cosf(fr), __builtin_amdgcn_sinf(fr));
}
// table index of the launch's first sample / blocks since the last change of the IQ-DC mean, for a channel that may have been restarted at run time
// mixed engines (sonde_engine_create_mixed) keep their channels grouped by sonde type: channel ch of the engine reads row in_row[ch] of the caller's buffer
__device__ __forceinline__ int md_in_row(const MixDecArgs &a, int ch) { return a.in_row ? a.in_row[ch] : ch; }
__device__ __forceinline__ uint32_t md_lut_phase(const MixDecArgs &a, int ch) {
    if (!a.epoch_phase) return a.lut_phase;
    const uint32_t L = (uint32_t)a.lut_len;
    return (a.lut_phase + L - a.epoch_phase[ch]) % L;
}
__device__ __forceinline__ int md_dc_since(const MixDecArgs &a, int ch) { return a.dc_since_ch ? a.dc_since_ch[ch] : a.dc_since; }

// k_md_etable: E[ch][i] = sum_{q<Q} sum_{r<D} W_q[r] ex[D ((i-(Q-1)+q) mod P) + r], i < P = lut_len / D: the decimator's output for
// the input x = 1 when the block that completes the output is block i of the mixer table's period.  Once per engine.
__global__ __launch_bounds__(256)
void k_md_etable(const double *chan_f0, const float *wtab, int D, int Q, int P, float2 *etab, int ph64) {
    const int ch = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const double f0 = chan_f0[ch];
    float er = 0.f, ei = 0.f;
    for (int q = 0; q < Q; q++) {
        const uint32_t n0 = (uint32_t)D * (uint32_t)((i - (Q - 1) + q + P) % P);
        for (int r = 0; r < D; r++) {
            const float2 e = ph64 ? md_phasor64(f0, n0 + (uint32_t)r) : md_phasor(f0, n0 + (uint32_t)r);
            const float w = wtab[8 * r + q];
            er = fmaf(w, e.x, er); ei = fmaf(w, e.y, ei);
        }
    }
    etab[(size_t)ch * P + i] = make_float2(er, ei);
}
// The launch that follows a change of the IQ-DC mean (avg_old -> avg_new) subtracts avg_new * E from all of its outputs, but the
// blocks BEFORE the launch — which reach into its first Q-1 outputs — were mixed while avg_old was in effect (demod_mod.c:495-504:
// the mean changes at a sample).  Output j < Q-1 of the launch therefore gets  + (avg_new - avg_old) * S_j,
// S_j = sum over q with j-(Q-1-q) < 0 of G_q[block j-(Q-1-q)],  G_q[b] = sum_r W_q[r] ex[D b + r]  (complex product).
// Evaluated by the first Q-1 lanes of the wave that owns the launch's first rows; a.dc_avg_prev == nullptr: no change within reach.
// (A launch shorter than Q-1 blocks leaves part of that reach to the next one: dc_since.)
__device__ __forceinline__ float2v md_dc_boundary(const MixDecArgs &a, int ch, double f0, int j) {
    float2v t = {0.f, 0.f};
    const int since = md_dc_since(a, ch);                     // blocks between the change of the mean and this launch
    if (a.dc_avg_prev && j < a.Q - 1 - since) {
        const int H = a.Q - 1, P = a.etab_len;
        const int i0 = (int)((md_lut_phase(a, ch) / (uint32_t)a.D) % (uint32_t)P);
        float sr = 0.f, si = 0.f;
        for (int q = 0; q < H - j - since; q++) {             // block b = j - (H - q) of the launch lies before the change
            const uint32_t n0 = (uint32_t)a.D * (uint32_t)((i0 + j - (H - q) + P) % P);
            for (int r = 0; r < a.D; r++) {
                const float2 e = md_phasor(f0, n0 + (uint32_t)r);
                const float w = a.wtab_g[8 * r + q];
                sr = fmaf(w, e.x, sr); si = fmaf(w, e.y, si);
            }
        }
        const float2 an = a.dc_avg[ch], ao = a.dc_avg_prev[ch];
        const float dx = an.x - ao.x, dy = an.y - ao.y;
        t = (float2v){dx * sr - dy * si, dx * si + dy * sr};
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// k_mix_decimate50: the 2.4 Msps -> 48 kHz decimator (D = 50, Q = 7, float table phase) with everything around the sample
// loop arranged for the memory system.  Measured on the plain kernel (tools/ab_variants.sh, empty sample loop): with ONE tile
// (12.8 KB) per wave in flight, 12 waves per CU, the launch takes 0.9 ms without any arithmetic — 154 KB per CU in flight
// against a loaded HBM latency of ~7 us caps the stream at 5.3 TB/s (a pure read of the same bytes in the same pattern:
// 6.3 TB/s, 6.8 with non-temporal loads; tools/probes/read_bw.hip).  So here
//   * TWO tiles per wave are in flight: two staging sets of 12 x 16 B + 8 B per lane (100 VGPRs), filled by non-temporal
//     loads whose address is an SGPR base + one lane offset (no address arithmetic per tile), waited for with vmcnt counts;
//   * the previous tile's contribution to the first Q-1 outputs of a tile is ONE carry value per lane (the rotation that
//     forms the diagonal sum delivers both this tile's terms and the next tile's carry) instead of a copy of all P rows;
//   * the IQ-DC mean is subtracted per output (avg * E, see md_fast_tile); the 2^-15 of the int16 scale sits in the tap table.
// That only fits 3 waves per SIMD (168 VGPRs) with every register placed by hand, so a wave's whole run of full tiles is ONE
// generated statement (MD50_LOOP_*, tools/gen_md_fast.py); C++ does the set-up, a tile that sticks out of the chunk, the P tail
// and the IQ-DC sums.  Requires nblocks even (every 16-byte piece of a tile is then either inside or outside the chunk) and
// >= 64, and rows aligned with the mixer table (lut_len % 50 == 0, lut_phase % 50 == 0: no row wraps around the table end).
// ------------------------------------------------------------------------------------------------
#define MD50_CLOBBERS MD50_S36_81, MD50_V28_167, "memory", "vcc", "scc"
#define MD50_S36_81 "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", \
        "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81"
#define MD50_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define MD50_V28_167 "v28", "v29", MD50_V10(3), MD50_V10(4), MD50_V10(5), MD50_V10(6), MD50_V10(7), MD50_V10(8), MD50_V10(9), MD50_V10(10), \
        MD50_V10(11), MD50_V10(12), MD50_V10(13), MD50_V10(14), MD50_V10(15), "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
#define MD50_ASM(BODY) asm volatile(BODY \
        : [o0] "=v"(acc[0]), [o1] "=v"(acc[1]), [o2] "=v"(acc[2]), [o3] "=v"(acc[3]), [o4] "=v"(acc[4]), [o5] "=v"(acc[5]), [o6] "=v"(acc[6]), \
          [carry] "+v"(carry), [e] "+v"(eidx), [sx] "+v"(sx), [sy] "+v"(sy) \
        : [row] "v"(row_lds), [voff16] "v"(voff16), [voff8] "v"(voff8), [ldsw16] "v"(ldsw16), [ldsw8] "v"(ldsw8), [lane4] "v"(lane4), \
          [tb] "s"(tb), [f0] "s"(f0), [navg] "s"(navg), [wt] "s"(wt_s), [yout] "s"(yout), [jm] "s"(jm), [rmask] "s"(rmask), [P] "s"(P), \
          [etab] "s"(etab), [nfull] "s"(nfull), [outmask] "s"(outmask) \
        : MD50_CLOBBERS)

template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_mix_decimate50(const MixDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_u[];
    constexpr int Q_T = 7, H = 6, D = 50, TILE_DW = MD_ROWS * D;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *sRaw = smem_u + wave * (TILE_DW + 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(sRaw);     // LDS byte address (low half of the flat address)
    const uint32_t row_lds = lds0 + 4u * D * lane, ldsw16 = lds0 + 16u * lane, ldsw8 = lds0 + 8u * lane;
    const uint32_t voff16 = 16u * lane, voff8 = 8u * lane, lane4 = 4u * lane;

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ch = (slot / a.wgs_per_ch) * 8 + xcd;
    const int wg = slot % a.wgs_per_ch;
    if (ch >= a.n_ch) return;
    const int seg = wg * 4 + wave;
    const int rows_per_seg = MD_ROWS * a.G - H;
    const int jb = seg * rows_per_seg;
    if (jb >= a.nblocks) return;
    const int je = min(a.nblocks, jb + rows_per_seg);
    const int jt0 = (seg == 0) ? jb : jb - H;
    const int ntiles = (je - jt0 + MD_ROWS - 1) / MD_ROWS;
    const int nfull = min(ntiles, (a.nblocks - jt0) / MD_ROWS);        // leading tiles that lie completely inside the chunk

    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)md_in_row(a, ch) * a.ch_stride;
    const float2 avg = a.dc_avg[ch];
    const uint64_t navg = md_navg_sgpr(avg);
    const double f0 = a.chan_f0[ch];
    float2 *yout = a.y + (size_t)ch * a.ring_len;
    const float *wt_s = a.wtab_g + 64 * 8;                     // tap rows * 2^-15
    const uint32_t rmask = (uint32_t)a.ring_len - 1;
    const uint32_t P = (uint32_t)a.etab_len;                   // blocks per period of the mixer table
    const float2 *etab = a.etab + (size_t)ch * P;

    // carry: what the P rows before this tile add to its first H outputs (lane l < H: sum over q of P[l-(H-q)][q], rows < 0)
    float2v carry = {0.f, 0.f};
    if (seg == 0 && lane < H) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int i = lane + q;                            // row l - (H-q) of the chunk = row l + q of the P tail (blocks -H .. -1)
            if (i < H) { const float2 v = a.ptail_in[((size_t)ch * 8 + i) * 8 + q]; carry += (float2v){v.x, v.y}; }
        }
    }
    if (seg == 0 && lane < H) carry += md_dc_boundary(a, ch, f0, lane);
    int sx = 0, sy = 0;
    // the lane's block as an index into the period of the mixer table (table index = D * eidx): phase seed and E index
    uint32_t eidx = (uint32_t)(((uint64_t)(md_lut_phase(a, ch) / D) + (uint64_t)(jt0 + lane)) % P);
    float2v acc[Q_T];

    if (nfull > 0) {
        const uint32_t *tb = iq + (size_t)jt0 * D;
        const uint32_t jm = a.m0 + (uint32_t)jt0;
        const uint64_t outmask = (seg == 0) ? ~0ull : ~0ull << H;           // the H halo rows of a later segment produce no output
        if constexpr (VAR == 1) { MD50_ASM(MD50_LOOP_1); }
        // timing experiments (tools/ab_variants.sh; streams from `gen_md_fast.py --experiments ...`): results may be garbage
#ifdef MD50_LOOP_2
        if constexpr (VAR == 2) { MD50_ASM(MD50_LOOP_2); }
#endif
#ifdef MD50_LOOP_3
        if constexpr (VAR == 3) { MD50_ASM(MD50_LOOP_3); }
#endif
#ifdef MD50_LOOP_4
        if constexpr (VAR == 4) { MD50_ASM(MD50_LOOP_4); }
#endif
#ifdef MD50_LOOP_5
        if constexpr (VAR == 5) { MD50_ASM(MD50_LOOP_5); }
#endif
        const int j = jt0 + (nfull - 1) * MD_ROWS + lane;      // rows of the last tile walked above
        if (j >= a.nblocks - H && j < a.nblocks) {             // P rows of the last Q-1 blocks go to the next call
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
    }
    if (ntiles > nfull) {                                     // the wave's last tile sticks out of the chunk: checked loads, no staging
        const int jt = jt0 + nfull * MD_ROWS, total_dw = a.nblocks * D;
#pragma unroll 1
        for (int v = 0; v < 13; v++) {
            const int c = 64 * v + lane, off = jt * D + 4 * c;
            u32x4_u w = {0u, 0u, 0u, 0u};
            if (4 * c < TILE_DW && off + 4 <= total_dw) w = *reinterpret_cast<const u32x4_u *>(iq + off);
            if (4 * c < TILE_DW) *reinterpret_cast<uint4 *>(sRaw + 4 * c) = make_uint4(w.x, w.y, w.z, w.w);
        }
        const int j = jt + lane;
        const bool outrow = j >= jb && j < je;
#pragma unroll
        for (int q = 0; q < Q_T; q++) acc[q] = (float2v){0.f, 0.f};
        float2v dcs = {0.f, 0.f};
        const float m = outrow ? 1.f : 0.f;
        md_fast_tile(row_lds, wt_s, f0, f0 * (double)(eidx * (uint32_t)D), (float2v){m, m}, acc, dcs);
        sx += (int)dcs.x; sy += (int)dcs.y;
        float2v y = acc[H] + carry;
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int k = H - q, src = (lane - k) & 63;
            const float2v r = { __shfl(acc[q].x, src), __shfl(acc[q].y, src) };
            if (lane >= k) y += r;
        }
        y = md_dc_correct(y, avg, etab[eidx]);
        if (outrow) yout[(a.m0 + (uint32_t)j) & rmask] = make_float2(y.x, y.y);
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
    }
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    if (lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch), (unsigned long long)(long long)sx);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch + 1), (unsigned long long)(long long)sy);
    }
}

// k_mix_decimate50s: the same generated tile loop for the SCANNER's front end (scan/dft_detect.c:1085-1101 mixer table from a double phase,
// :539-588 IQ-DC over 1/32 s windows): MD50_LOOP_S keeps the phase in double and takes the mean of the row's window off every sample (the
// launch spans many windows: MixDecArgs.dc_seg, filled by k_dc_seg_sums / k_dc_seg_means, which also own the sums).  The P rows therefore
// hold what the generic k_mix_decimate<7, true, 50, 0> holds, and the two can follow each other between calls (P tail).
#define MD50S_ASM(BODY) asm volatile(BODY \
        : [o0] "=v"(acc[0]), [o1] "=v"(acc[1]), [o2] "=v"(acc[2]), [o3] "=v"(acc[3]), [o4] "=v"(acc[4]), [o5] "=v"(acc[5]), [o6] "=v"(acc[6]), \
          [carry] "+v"(carry), [e] "+v"(eidx), [jrow] "+v"(jrow) \
        : [row] "v"(row_lds), [voff16] "v"(voff16), [voff8] "v"(voff8), [ldsw16] "v"(ldsw16), [ldsw8] "v"(ldsw8), [lane4] "v"(lane4), \
          [tb] "s"(tb), [f0] "s"(f0), [wt] "s"(wt_s), [yout] "s"(yout), [jm] "s"(jm), [rmask] "s"(rmask), [P] "s"(P), \
          [nfull] "s"(nfull), [outmask] "s"(outmask), [navg0] "v"(navg0), [dcseg] "s"(dcseg), [segoff64] "s"(segoff64), [rcpB] "s"(rcpB), [segmax] "s"(segmax) \
        : MD50_CLOBBERS)
__device__ __forceinline__ float2v md_seg_navg(const float2 *dcseg, int j, int off, int B, int nmax) {
    int k = (j + off) / B; k = k < nmax ? k : nmax;
    const float2 m = dcseg[k];
    return (float2v){-32768.f * m.x, -32768.f * m.y};
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_mix_decimate50s(const MixDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_u[];
    constexpr int Q_T = 7, H = 6, D = 50, TILE_DW = MD_ROWS * D;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *sRaw = smem_u + wave * (TILE_DW + 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(sRaw);
    const uint32_t row_lds = lds0 + 4u * D * lane, ldsw16 = lds0 + 16u * lane, ldsw8 = lds0 + 8u * lane;
    const uint32_t voff16 = 16u * lane, voff8 = 8u * lane, lane4 = 4u * lane;

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ch = (slot / a.wgs_per_ch) * 8 + xcd;
    const int wg = slot % a.wgs_per_ch;
    if (ch >= a.n_ch) return;
    const int seg = wg * 4 + wave;
    const int rows_per_seg = MD_ROWS * a.G - H;
    const int jb = seg * rows_per_seg;
    if (jb >= a.nblocks) return;
    const int je = min(a.nblocks, jb + rows_per_seg);
    const int jt0 = (seg == 0) ? jb : jb - H;
    const int ntiles = (je - jt0 + MD_ROWS - 1) / MD_ROWS;
    const int nfull = min(ntiles, (a.nblocks - jt0) / MD_ROWS);

    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)md_in_row(a, ch) * a.ch_stride;
    const double f0 = a.chan_f0[ch];
    float2 *yout = a.y + (size_t)ch * a.ring_len;
    const float *wt_s = a.wtab_g + 64 * 8;                     // tap rows * 2^-15
    const uint32_t rmask = (uint32_t)a.ring_len - 1;
    const uint32_t P = (uint32_t)(a.lut_len / D);              // blocks per period of the mixer table
    const float2 *dcseg = a.dc_seg + (size_t)ch * a.dc_seg_n;
    const uint32_t segoff64 = (uint32_t)a.dc_seg_off + 64u, segmax = (uint32_t)a.dc_seg_n - 1u;
    const float rcpB = 1.0f / (float)a.dc_seg_blocks;

    // carry: what the P rows before this tile add to its first H outputs (lane l < H: sum over q of P[l-(H-q)][q], rows < 0)
    float2v carry = {0.f, 0.f};
    if (seg == 0 && lane < H) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int i = lane + q;
            if (i < H) { const float2 v = a.ptail_in[((size_t)ch * 8 + i) * 8 + q]; carry += (float2v){v.x, v.y}; }
        }
    }
    uint32_t eidx = (uint32_t)(((uint64_t)(a.lut_phase / D) + (uint64_t)(jt0 + lane)) % P);
    uint32_t jrow = (uint32_t)(jt0 + lane);
    float2v acc[Q_T];

    if (nfull > 0) {
        const uint32_t *tb = iq + (size_t)jt0 * D;
        const uint32_t jm = a.m0 + (uint32_t)jt0;
        const uint64_t outmask = (seg == 0) ? ~0ull : ~0ull << H;
        const float2v navg0 = md_seg_navg(dcseg, jt0 + lane, a.dc_seg_off, a.dc_seg_blocks, (int)segmax);
        MD50S_ASM(MD50_LOOP_S);
        const int j = jt0 + (nfull - 1) * MD_ROWS + lane;
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
    }
    if (ntiles > nfull) {                                     // the wave's last tile sticks out of the chunk: checked loads, no staging
        const int jt = jt0 + nfull * MD_ROWS, total_dw = a.nblocks * D;
#pragma unroll 1
        for (int v = 0; v < 13; v++) {
            const int c = 64 * v + lane, off = jt * D + 4 * c;
            u32x4_u w = {0u, 0u, 0u, 0u};
            if (4 * c < TILE_DW && off + 4 <= total_dw) w = *reinterpret_cast<const u32x4_u *>(iq + off);
            if (4 * c < TILE_DW) *reinterpret_cast<uint4 *>(sRaw + 4 * c) = make_uint4(w.x, w.y, w.z, w.w);
        }
        const int j = jt + lane;
        const bool outrow = j >= jb && j < je;
#pragma unroll
        for (int q = 0; q < Q_T; q++) acc[q] = (float2v){0.f, 0.f};
        md_fast_tile_s(row_lds, wt_s, f0, f0 * (double)(eidx * (uint32_t)D), md_seg_navg(dcseg, j, a.dc_seg_off, a.dc_seg_blocks, (int)segmax), acc);
        float2v y = acc[H] + carry;
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int k = H - q, src = (lane - k) & 63;
            const float2v r = { __shfl(acc[q].x, src), __shfl(acc[q].y, src) };
            if (lane >= k) y += r;
        }
        if (outrow) yout[(a.m0 + (uint32_t)j) & rmask] = make_float2(y.x, y.y);
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
    }
}

// k_mix_decimate50r — the scanner's base-rate front end in ONE pass over the input (round 4).  dft_detect takes the mean of the PREVIOUS 1/32 s window off
// every sample (dft_detect.c:539-588): k_mix_decimate50s therefore runs behind a pass of its own that does nothing but add (k_dc_seg_sums — the input is
// read twice, and the two passes together are HBM-bound).  The filter is linear: y = sum W (x - mean) ex = sum W x ex - sum over the blocks of mean(block) *
// (the block's part of E).  So the sample loop here needs no mean at all (MD50_LOOP_R: the raw sum, and every block's sum of raw samples to bsum — 8 bytes
// per 200 of input), every window is mixed in parallel, and what depends on the means happens at the IF rate, on 1/50 of the bytes: k_dc_rows_to_segments adds
// the block sums up per window, k_dc_seg_means makes the table of means (unchanged), k_scan_dc_edges tabulates what the Q-1 outputs behind a change of the
// mean need on top, and k_scan_if subtracts mean * E from every output AS IT LOADS IT (ScanFold; a pass of its own over y cost 0.27 ms per 512 channel-seconds).
// The P tail between calls holds raw partial sums; the ring y holds raw outputs (k_scan_if keeps the folded history it needs).  Rounding: y - mean * E rounds relative to |mean|, not to |x - mean| — 1e-8 of the offset, as in the demodulator's decimator.
#define MD50R_ASM(BODY) asm volatile(BODY \
        : [o0] "=v"(acc[0]), [o1] "=v"(acc[1]), [o2] "=v"(acc[2]), [o3] "=v"(acc[3]), [o4] "=v"(acc[4]), [o5] "=v"(acc[5]), [o6] "=v"(acc[6]), \
          [carry] "+v"(carry), [e] "+v"(eidx), [jrow] "+v"(jrow) \
        : [row] "v"(row_lds), [voff16] "v"(voff16), [voff8] "v"(voff8), [ldsw16] "v"(ldsw16), [ldsw8] "v"(ldsw8), [lane4] "v"(lane4), \
          [tb] "s"(tb), [f0] "s"(f0), [wt] "s"(wt_s), [yout] "s"(yout), [jm] "s"(jm), [rmask] "s"(rmask), [P] "s"(P), \
          [nfull] "s"(nfull), [outmask] "s"(outmask), [bsum] "s"(bsum) \
        : MD50_CLOBBERS)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_mix_decimate50r(const MixDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_u[];
    constexpr int Q_T = 7, H = 6, D = 50, TILE_DW = MD_ROWS * D;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *sRaw = smem_u + wave * (TILE_DW + 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(sRaw);
    const uint32_t row_lds = lds0 + 4u * D * lane, ldsw16 = lds0 + 16u * lane, ldsw8 = lds0 + 8u * lane;
    const uint32_t voff16 = 16u * lane, voff8 = 8u * lane, lane4 = 4u * lane;

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ch = (slot / a.wgs_per_ch) * 8 + xcd;
    const int wg = slot % a.wgs_per_ch;
    if (ch >= a.n_ch) return;
    const int seg = wg * 4 + wave;
    const int rows_per_seg = MD_ROWS * a.G - H;
    const int jb = seg * rows_per_seg;
    if (jb >= a.nblocks) return;
    const int je = min(a.nblocks, jb + rows_per_seg);
    const int jt0 = (seg == 0) ? jb : jb - H;
    const int ntiles = (je - jt0 + MD_ROWS - 1) / MD_ROWS;
    const int nfull = (a.nblocks & 1) ? 0 : min(ntiles, (a.nblocks - jt0) / MD_ROWS);      // (odd launches: channel rows need not sit on the 16-byte grid)

    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)md_in_row(a, ch) * a.ch_stride;
    const double f0 = a.chan_f0[ch];
    float2 *yout = a.y + (size_t)ch * a.ring_len;
    const float *wt_s = a.wtab_g + 64 * 8;                     // tap rows * 2^-15
    const uint32_t rmask = (uint32_t)a.ring_len - 1;
    const uint32_t P = (uint32_t)(a.lut_len / D);              // blocks per period of the mixer table
    int2 *bsum = a.bsum + (size_t)ch * a.bsum_stride;

    // carry: what the P rows before this tile add to its first H outputs (lane l < H: sum over q of P[l-(H-q)][q], rows < 0)
    float2v carry = {0.f, 0.f};
    if (seg == 0 && lane < H) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int i = lane + q;
            if (i < H) { const float2 v = a.ptail_in[((size_t)ch * 8 + i) * 8 + q]; carry += (float2v){v.x, v.y}; }
        }
    }
    if (seg == 0 && a.nblocks < H && lane < H - a.nblocks) {  // a launch shorter than the history: the older tail rows move up
        for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + lane) * 8 + q] = a.ptail_in[((size_t)ch * 8 + lane + a.nblocks) * 8 + q];
    }
    uint32_t eidx = (uint32_t)(((uint64_t)(a.lut_phase / D) + (uint64_t)(jt0 + lane)) % P);
    uint32_t jrow = (uint32_t)(jt0 + lane);
    float2v acc[Q_T];

    if (nfull > 0) {
        const uint32_t *tb = iq + (size_t)jt0 * D;
        const uint32_t jm = a.m0 + (uint32_t)jt0;
        const uint64_t outmask = (seg == 0) ? ~0ull : ~0ull << H;
        MD50R_ASM(MD50_LOOP_R);
        const int j = jt0 + (nfull - 1) * MD_ROWS + lane;
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
    }
    // tiles the generated loop does not take — the one that sticks out of the chunk, every tile of a launch with an odd number of blocks (channel rows off the
    // 16-byte grid) or of fewer than 64: checked loads, one tile at a time, the carry handed on in C++
#pragma unroll 1
    for (int t = nfull; t < ntiles; t++) {
        const int jt = jt0 + t * MD_ROWS, total_dw = a.nblocks * D;
#pragma unroll 1
        for (int v = 0; v < 13; v++) {
            const int c = 64 * v + lane, off = jt * D + 4 * c;
            u32x4_u w = {0u, 0u, 0u, 0u};
            if (4 * c < TILE_DW && off + 4 <= total_dw) w = *reinterpret_cast<const u32x4_u *>(iq + off);
            else if (4 * c < TILE_DW) { if (off < total_dw) w.x = iq[off]; if (off + 1 < total_dw) w.y = iq[off + 1]; if (off + 2 < total_dw) w.z = iq[off + 2]; }      // (an odd launch ends inside a quad)
            if (4 * c < TILE_DW) *reinterpret_cast<uint4 *>(sRaw + 4 * c) = make_uint4(w.x, w.y, w.z, w.w);
        }
        const int j = jt + lane;
        const bool outrow = j >= jb && j < je;
#pragma unroll
        for (int q = 0; q < Q_T; q++) acc[q] = (float2v){0.f, 0.f};
        float2v dcs = {0.f, 0.f};
        md_fast_tile_r(row_lds, wt_s, f0, f0 * (double)(eidx * (uint32_t)D), acc, dcs);
        if (outrow) bsum[j] = make_int2((int)dcs.x, (int)dcs.y);
        float2v y = acc[H] + carry, cnext = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int k = H - q, src = (lane - k) & 63;
            const float2v r = { __shfl(acc[q].x, src), __shfl(acc[q].y, src) };
            if (lane >= k) y += r; else cnext += r;           // lanes < k hold rows 64 - k + lane of this tile: the next tile's carry
        }
        carry = cnext;
        if (outrow) yout[(a.m0 + (uint32_t)j) & rmask] = make_float2(y.x, y.y);
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
        eidx += 64u; eidx = eidx >= P ? eidx - P : eidx;
    }
}


// block sums -> window sums: window k of the launch covers blocks [k B - off, (k + 1) B - off) of it (k_dc_seg_sums' output, from 1/25 of its input)
__global__ __launch_bounds__(256)
void k_dc_rows_to_segments(const int2 *bsum, long long bsum_stride, int nblocks, int seg_off, int seg_blocks, long long *seg_sums, int nseg) {
    const int k = blockIdx.x, ch = blockIdx.y;
    const long long lo = (long long)k * seg_blocks - seg_off, hi = lo + seg_blocks;
    const int j0 = (int)(lo < 0 ? 0 : lo), j1 = (int)(hi > nblocks ? nblocks : hi);
    const int2 *p = bsum + (size_t)ch * bsum_stride;
    long long lx = 0, ly = 0;
    for (int j = j0 + (int)threadIdx.x; j < j1; j += 256) { const int2 v = p[j]; lx += v.x; ly += v.y; }
    for (int off = 32; off > 0; off >>= 1) { lx += __shfl_down(lx, off); ly += __shfl_down(ly, off); }
    __shared__ long long s_l[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_l[2 * wave] = lx; s_l[2 * wave + 1] = ly; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tx = 0, ty = 0;
        for (int w = 0; w < 4; w++) { tx += s_l[2 * w]; ty += s_l[2 * w + 1]; }
        seg_sums[((size_t)ch * nseg + k) * 2] = tx; seg_sums[((size_t)ch * nseg + k) * 2 + 1] = ty;
    }
}
// The Q-1 outputs behind a change of the IQ-DC mean: output m = (start of window k) + i, i < Q-1, sums Q blocks of which the first Q-1-i still ran under
// the mean of window k-1.  k_scan_if takes mean(k) * E[e] off every output (E = sum over the Q blocks' parts, k_md_etable); what is missing for these is
// corr[ch][k][i] = -(mean(k-1) - mean(k)) * sum_{q < Q-1-i} Eblk(e, q),  Eblk(e, q) = sum_r W_q[r] ex[D ((e - (Q-1) + q) mod P) + r]  (k_md_etable's order).
// A window that began before the launch (k = 0 with dc_seg_off > 0) has its first outputs in the launch before; mean(-1) travels in dc_prev.
// (Eblk(e + i, q) is the tap column q over table block e - (Q-1) + t, t = i + q: 21 (t, q) pairs of D terms for Q = 7.  A lane takes one pair and a third of
// its D terms — one lane per output walked 300 dependent sincos and the launch took 0.115 ms.)
__global__ __launch_bounds__(64)
void k_scan_dc_edges(const ScanEdgeArgs a) {
    const int k = blockIdx.x, ch = blockIdx.y, lane = threadIdx.x;
    const int H = a.Q - 1, P = a.etab_len;
    if (k >= a.nseg) return;
    __shared__ float2 s_part[64];
    const int m0w = k * a.dc_seg_blocks - a.dc_seg_off;                 // the window's first block, as a block of this launch (negative: it began before)
    const uint32_t e0w = (uint32_t)((((long long)a.e0 + (long long)m0w) % P + P) % P);      // its table block
    const double f0 = a.chan_f0[ch];
    // pair p < H (H + 1) / 2: t = row of the triangle, q <= t; lanes p, p + 21, p + 42 share its D terms
    const int np = H * (H + 1) / 2;
    const int p = lane % np, part = lane / np, nparts = 64 / np;
    float er = 0.f, ei = 0.f;
    if (part < nparts) {
        int t = 0, q = p; while (q > t) { q -= t + 1; t++; }             // p = t (t + 1) / 2 + q
        const uint32_t n0 = (uint32_t)a.D * (uint32_t)(((long long)e0w - H + t + 2LL * P) % P);
        const int per = (a.D + nparts - 1) / nparts, r0 = part * per, r1 = min(a.D, r0 + per);
        for (int r = r0; r < r1; r++) {
            const float2 ex = md_phasor64(f0, n0 + (uint32_t)r);
            const float w = a.wtab[8 * r + q];
            er = fmaf(w, ex.x, er); ei = fmaf(w, ex.y, ei);
        }
    }
    s_part[lane] = make_float2(er, ei);
    __syncthreads();
    if (lane < H) {
        const int i = lane, m = m0w + i;
        float2 *out = a.corr + ((size_t)ch * a.dc_seg_n + k) * 8 + i;
        if (m < 0 || m >= a.nblocks) { *out = make_float2(0.f, 0.f); return; }
        float sr = 0.f, si = 0.f;
        for (int q = 0; q < H - i; q++) {                               // blocks of output i that lie before the change: tap columns q < H - i, table row t = i + q
            const int t = i + q, pp = t * (t + 1) / 2 + q;
            for (int c = 0; c < nparts; c++) { const float2 v = s_part[pp + c * np]; sr += v.x; si += v.y; }
        }
        const float2 mn = a.dc_seg[(size_t)ch * a.dc_seg_n + k], mo = k > 0 ? a.dc_seg[(size_t)ch * a.dc_seg_n + k - 1] : a.dc_prev[ch];
        const float dx = mn.x - mo.x, dy = mn.y - mo.y;                 // -(mo - mn)
        *out = make_float2(dx * sr - dy * si, dx * si + dy * sr);
    }
}

template <int Q_T, bool PH64, int D_T, int FAST>
__global__ __launch_bounds__(256)
void k_mix_decimate(const MixDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_u[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = D_T > 0 ? D_T : a.D;
    constexpr int H = Q_T - 1;
    const int tile_dw = MD_ROWS * D;
    uint32_t *sRaw = smem_u + wave * (tile_dw + 4);
    // LDS byte address of the lane's row (the low half of a flat LDS address is the LDS offset)
    const uint32_t row_lds = (uint32_t)reinterpret_cast<uintptr_t>(sRaw + lane * D);

    // XCD-aware mapping: consecutive block ids round-robin over the 8 XCDs; a channel stays on one XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ch = (slot / a.wgs_per_ch) * 8 + xcd;
    const int wg = slot % a.wgs_per_ch;
    if (ch >= a.n_ch) return;
    const int seg = wg * 4 + wave;
    const int rows_per_seg = MD_ROWS * a.G - H;               // later segments start H rows early: exactly G full tiles
    const int jb = seg * rows_per_seg;
    if (jb >= a.nblocks) return;
    const int je = min(a.nblocks, jb + rows_per_seg);

    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)md_in_row(a, ch) * a.ch_stride;
    const float2 avg = a.dc_avg[ch];
    const double f0 = a.chan_f0[ch];
    const uint32_t L = (uint32_t)a.lut_len;
    float2 *yout = a.y + (size_t)ch * a.ring_len;
    const int total_dw = a.nblocks * D;
    const int nv = (D + 3) >> 2;                              // 16-byte loads per lane per tile
    constexpr int NV_T = D_T > 0 ? (D_T + 3) / 4 : MD_NVMAX;

    // P of the previous tile (carry for the diagonal sum).  Segment 0 continues the previous call (P tail),
    // later segments start Q-1 rows early instead (those rows produce no output).
    float2v pv[Q_T];
#pragma unroll
    for (int q = 0; q < Q_T; q++) pv[q] = (float2v){0.f, 0.f};
    if (seg == 0 && lane >= MD_ROWS - H) {
#pragma unroll
        for (int q = 0; q < Q_T; q++) {
            const float2 v = a.ptail_in[((size_t)ch * 8 + (lane - (MD_ROWS - H))) * 8 + q];
            pv[q] = (float2v){v.x, v.y};
        }
    }
    if (seg == 0 && a.nblocks < H && lane < H - a.nblocks) {  // chunk shorter than the history: old tail rows survive
        for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + lane) * 8 + q] = a.ptail_in[((size_t)ch * 8 + lane + a.nblocks) * 8 + q];
    }

    const int jt0 = (seg == 0) ? jb : jb - H;
    u32x4_u pre[NV_T];
    // tile bytes -> registers: 16 bytes per lane per load, 64 lanes = 1 KB per load instruction.  A tile that lies
    // completely inside the chunk takes the unchecked path (immediate offsets from one base address).
    auto fetch = [&](int jt) {
        if (D_T > 0 && jt + MD_ROWS <= a.nblocks) {
            const uint32_t *p = iq + (size_t)jt * D + 4 * lane;
#pragma unroll
            for (int v = 0; v < NV_T; v++) {
                if (4 * (64 * v) + 3 < MD_ROWS * D_T) {         // load v exists for lane 0; the last one only for the low lanes
                    const uint32_t *q = (4 * (64 * v + 63) < MD_ROWS * D_T || 4 * (64 * v + lane) < MD_ROWS * D_T) ? p + 256 * v : p;
                    // read-once stream: non-temporal (tools/probes/read_bw.hip: 6.3 -> 6.8 TB/s on this access pattern)
                    pre[v].x = __builtin_nontemporal_load(q); pre[v].y = __builtin_nontemporal_load(q + 1);
                    pre[v].z = __builtin_nontemporal_load(q + 2); pre[v].w = __builtin_nontemporal_load(q + 3);
                }
            }
            return;
        }
#pragma unroll
        for (int v = 0; v < NV_T; v++) {
            if (v < nv) {
                const int c = 64 * v + lane;                  // 16-byte chunk of the tile
                const int off = jt * D + 4 * c;               // dword offset in the chunk
                if (4 * c < tile_dw && off + 4 <= total_dw) pre[v] = *reinterpret_cast<const u32x4_u *>(iq + off);
                else if (4 * c < tile_dw && off < total_dw) { // the chunk ends inside this 16-byte piece: keep the samples that exist
                    pre[v].x = iq[off];
                    pre[v].y = off + 1 < total_dw ? iq[off + 1] : 0u;
                    pre[v].z = off + 2 < total_dw ? iq[off + 2] : 0u;
                    pre[v].w = 0u;
                }
                else pre[v] = u32x4_u{0u, 0u, 0u, 0u};          // beyond the tile / the chunk
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int v = 0; v < NV_T; v++) {
            if (v < nv) {
                const int c = 64 * v + lane;
                if (4 * c < tile_dw) *reinterpret_cast<uint4 *>(sRaw + 4 * c) = make_uint4(pre[v].x, pre[v].y, pre[v].z, pre[v].w);
            }
        }
    };
    fetch(jt0);
    park();

    int sx = 0, sy = 0;
    const uint32_t step = (uint32_t)(((uint64_t)MD_ROWS * (uint64_t)D) % L);
    uint32_t rown = (uint32_t)(((uint64_t)md_lut_phase(a, ch) + (uint64_t)(jt0 + lane) * (uint64_t)D) % L);

    for (int jt = jt0; jt < je; jt += MD_ROWS) {
        const bool more = jt + MD_ROWS < je;
        if (more) fetch(jt + MD_ROWS);                        // next tile's bytes fly while this one is computed
        const int j = jt + lane;
        const bool outrow = j >= jb && j < je;
        float2v acc[Q_T];
#pragma unroll
        for (int q = 0; q < Q_T; q++) acc[q] = (float2v){0.f, 0.f};
        float2v dcs = {0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool nowrap = __builtin_amdgcn_ballot_w64(L - rown < (uint32_t)D) == 0;     // wave-uniform
        if constexpr (FAST > 0) {                             // rows are aligned with the mixer table here (launcher): none wraps
            const float m = outrow ? 1.f : 0.f;
            md_fast_tile(row_lds, a.wtab_g + 64 * 8, f0, f0 * (double)rown, (float2v){m, m}, acc, dcs);
        } else {
            float2 ravg = avg;                                // the mean of the IQ-DC window the row lies in, when the launch spans several
            if (a.dc_seg) { int k = (j + a.dc_seg_off) / a.dc_seg_blocks; k = k < a.dc_seg_n ? k : a.dc_seg_n - 1; ravg = a.dc_seg[(size_t)ch * a.dc_seg_n + k]; }
            if (nowrap) md_rows<Q_T, false, PH64, D_T>(sRaw + lane * D, D, a.wtab, ravg, f0, rown, L, outrow ? 1.f : 0.f, a.nd_base, acc, dcs);
            else        md_rows<Q_T, true, PH64, 0>(sRaw + lane * D, D, a.wtab, ravg, f0, rown, L, outrow ? 1.f : 0.f, a.nd_base, acc, dcs);
        }
        sx += (int)dcs.x; sy += (int)dcs.y;

        // y[j] = sum_q P[j-(H-q)][q]: shift column q down by H-q lanes, the first lanes take the previous tile's rows
        float yr = acc[H].x, yi = acc[H].y;
#pragma unroll
        for (int q = 0; q < H; q++) {
            // lane l takes row l-k: this tile's for l >= k, the previous tile's row l-k+64 otherwise — one rotation of
            // the vector that holds the previous tile in its last k lanes
            const int k = H - q, src = (lane - k) & 63;
            const bool old = lane >= MD_ROWS - k;
            yr += __shfl(old ? pv[q].x : acc[q].x, src);
            yi += __shfl(old ? pv[q].y : acc[q].y, src);
        }
        if constexpr (FAST > 0) {                             // the IQ-DC mean, per output: y -= avg * E (md_fast_tile)
            if (outrow) {
                float2v yc = md_dc_correct((float2v){yr, yi}, avg, a.etab[(size_t)ch * a.etab_len + (rown / (uint32_t)D)]);
                if (j < H) yc += md_dc_boundary(a, ch, f0, j);
                yr = yc.x; yi = yc.y;
            }
        }
        if (outrow) yout[(a.m0 + (uint32_t)j) & (uint32_t)(a.ring_len - 1)] = make_float2(yr, yi);
        if (j >= a.nblocks - H && j < a.nblocks) {            // P rows of the last Q-1 blocks go to the next call
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
#pragma unroll
        for (int q = 0; q < Q_T; q++) pv[q] = acc[q];
        if (more) park();                                     // LDS reads of the old tile are done (in-order DS)
        rown += step; if (rown >= L) rown -= L;
    }

    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    if (lane == 0 && !a.dc_seg) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch), (unsigned long long)(long long)sx);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch + 1), (unsigned long long)(long long)sy);
    }
}

// IQ-DC windows of a launch that spans several (MixDecArgs.dc_seg): k_dc_seg_sums adds up the raw samples of every window the call touches,
// k_dc_seg_means turns them into the table of means — window k runs under the mean of window k-1 (the first under the one carried in dc_avg), the
// trailing incomplete window's sum is carried to the next call in dc_sums — exactly what one launch per window plus k_dc_update did.
__global__ __launch_bounds__(256)
void k_dc_seg_sums(const int16_t *iq, long long ch_stride, int n_samples, unsigned dc_cnt0, unsigned dc_max, long long *seg_sums, int nseg) {
    const int k = blockIdx.x, ch = blockIdx.y;
    const long long lo = (long long)k * dc_max - dc_cnt0, hi = lo + dc_max;
    const int s0 = (int)(lo < 0 ? 0 : lo), s1 = (int)(hi > n_samples ? n_samples : hi);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(iq) + (size_t)ch * ch_stride;
    long long lx = 0, ly = 0;
    // 16-byte loads over the part of the window that is 16-byte aligned (the channel rows are), four of them in flight per thread; the few
    // samples in front of and behind it one by one.  An int accumulates at most 2^15 samples of 16 bits before it is folded into the long sums.
    // (aligned by ADDRESS, not by sample index: a caller's rows need not start on 16 bytes — converted 8-bit input with a row stride of
    // n_samples, a d_in offset)
    const int mis = (int)((reinterpret_cast<uintptr_t>(p) >> 2) & 3u);                   // samples the row start lies behind a 16-byte boundary
    const int a0 = min(s1, ((s0 + mis + 3) & ~3) - mis), a1 = max(a0, ((s1 + mis) & ~3) - mis);
    for (int i = s0 + (int)threadIdx.x; i < a0; i += 256) { const uint32_t raw = p[i]; lx += (int)(short)(raw & 0xffffu); ly += ((int)raw) >> 16; }
    for (int i = a1 + (int)threadIdx.x; i < s1; i += 256) { const uint32_t raw = p[i]; lx += (int)(short)(raw & 0xffffu); ly += ((int)raw) >> 16; }
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(p + a0);
        const int nq = (a1 - a0) >> 2;
        int sx = 0, sy = 0, cnt = 0;
        auto add = [&](const uint4 v) {
            sx += (int)(short)(v.x & 0xffffu) + (int)(short)(v.y & 0xffffu) + (int)(short)(v.z & 0xffffu) + (int)(short)(v.w & 0xffffu);
            sy += (((int)v.x) >> 16) + (((int)v.y) >> 16) + (((int)v.z) >> 16) + (((int)v.w) >> 16);
        };
        int i = (int)threadIdx.x;
        for (; i + 3 * 256 < nq; i += 4 * 256) {
            const uint4 v0 = q[i], v1 = q[i + 256], v2 = q[i + 512], v3 = q[i + 768];
            add(v0); add(v1); add(v2); add(v3);
            if ((cnt += 16) >= 32768 - 16) { lx += sx; ly += sy; sx = sy = 0; cnt = 0; }
        }
        for (; i < nq; i += 256) add(q[i]);
        lx += sx; ly += sy;
    }
    for (int off = 32; off > 0; off >>= 1) { lx += __shfl_down(lx, off); ly += __shfl_down(ly, off); }
    __shared__ long long s_l[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_l[2 * wave] = lx; s_l[2 * wave + 1] = ly; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tx = 0, ty = 0;
        for (int w = 0; w < 4; w++) { tx += s_l[2 * w]; ty += s_l[2 * w + 1]; }
        seg_sums[((size_t)ch * nseg + k) * 2] = tx; seg_sums[((size_t)ch * nseg + k) * 2 + 1] = ty;
    }
}
__global__ void k_dc_seg_means(int n_ch, int nseg, int ncomplete, float maxcnt, const long long *seg_sums, long long *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n,
                               const float2 *dc_prev = nullptr, float2 *dc_prev_out = nullptr) {
    // dc_prev / dc_prev_out (optional): the mean of the window before the one in progress, at the start / at the end of this call — two arrays, because
    // k_scan_dc_edges, behind this kernel, needs the value of the call's START (the mean the blocks in front of the call's first window ran under)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    float2 mean = dc_avg[c], prev = dc_prev ? dc_prev[c] : mean;
    long long cx = dc_sums[2 * c], cy = dc_sums[2 * c + 1];   // what the window in progress had collected before this call
    for (int k = 0; k < dc_seg_n; k++) {
        dc_seg[(size_t)c * dc_seg_n + k] = mean;
        if (k >= nseg) continue;
        cx += seg_sums[((size_t)c * nseg + k) * 2]; cy += seg_sums[((size_t)c * nseg + k) * 2 + 1];
        if (k < ncomplete) {                                  // avg = (float)(sum / (float)maxcnt), sum = S / 32768 exact in double (k_dc_update)
            prev = mean;
            mean = make_float2((float)(((double)cx / 32768.0) / (double)maxcnt), (float)(((double)cy / 32768.0) / (double)maxcnt));
            cx = 0; cy = 0;
        }
    }
    dc_avg[c] = mean; dc_sums[2 * c] = cx; dc_sums[2 * c + 1] = cy;
    if (dc_prev_out) dc_prev_out[c] = prev;
}

// Decimation factors above 64 (input rates above ~3 Msps, e.g. a 10 Msps wideband stream): same lane-per-block scheme,
// but a block's D samples are walked in NS = D/DS pieces of DS <= 64 samples.  Each lane stages its own piece in LDS
// (16-byte loads along its row), so there is no cross-lane traffic at all; the taps come from a global table.  Not
// software-pipelined — this variant serves the wideband scanner, not the headline path.
template <int Q_T, bool PH64>
__global__ __launch_bounds__(256)
void k_mix_decimate_wide(const MixDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_u[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D, DS = a.DS, NS = D / DS, pitch = (DS + 3) & ~3;
    constexpr int H = Q_T - 1;
    uint32_t *sRow = smem_u + (wave * MD_ROWS + lane) * pitch;

    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ch = (slot / a.wgs_per_ch) * 8 + xcd;
    const int wg = slot % a.wgs_per_ch;
    if (ch >= a.n_ch) return;
    const int seg = wg * 4 + wave;
    const int rows_per_seg = MD_ROWS * a.G - H;
    const int jb = seg * rows_per_seg;
    if (jb >= a.nblocks) return;
    const int je = min(a.nblocks, jb + rows_per_seg);

    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)md_in_row(a, ch) * a.ch_stride;
    const float2 avg = a.dc_avg[ch];
    const double f0 = a.chan_f0[ch];
    const uint32_t L = (uint32_t)a.lut_len;
    float2 *yout = a.y + (size_t)ch * a.ring_len;
    const long long total_dw = (long long)a.nblocks * D;

    float2v pv[Q_T];
#pragma unroll
    for (int q = 0; q < Q_T; q++) pv[q] = (float2v){0.f, 0.f};
    if (seg == 0 && lane >= MD_ROWS - H) {
#pragma unroll
        for (int q = 0; q < Q_T; q++) {
            const float2 v = a.ptail_in[((size_t)ch * 8 + (lane - (MD_ROWS - H))) * 8 + q];
            pv[q] = (float2v){v.x, v.y};
        }
    }
    if (seg == 0 && a.nblocks < H && lane < H - a.nblocks) {
        for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + lane) * 8 + q] = a.ptail_in[((size_t)ch * 8 + lane + a.nblocks) * 8 + q];
    }
    const int jt0 = (seg == 0) ? jb : jb - H;
    int sx = 0, sy = 0;
    for (int jt = jt0; jt < je; jt += MD_ROWS) {
        const int j = jt + lane;
        const bool outrow = j >= jb && j < je;
        float2v acc[Q_T];
#pragma unroll
        for (int q = 0; q < Q_T; q++) acc[q] = (float2v){0.f, 0.f};
        float2v dcs = {0.f, 0.f};
        const uint32_t rown = (uint32_t)(((uint64_t)md_lut_phase(a, ch) + (uint64_t)j * (uint64_t)D) % L);
        for (int sub = 0; sub < NS; sub++) {
            const long long o0 = (long long)j * D + (long long)sub * DS;
            for (int v = 0; v < pitch / 4; v++) {
                const long long o = o0 + 4 * v;
                uint4 w = make_uint4(0u, 0u, 0u, 0u);
                if (j < a.nblocks) {
                    if (o + 4 <= total_dw) { const u32x4_u t = *reinterpret_cast<const u32x4_u *>(iq + o); w = make_uint4(t.x, t.y, t.z, t.w); }
                    else { if (o < total_dw) w.x = iq[o]; if (o + 1 < total_dw) w.y = iq[o + 1]; if (o + 2 < total_dw) w.z = iq[o + 2]; }
                }
                *reinterpret_cast<uint4 *>(sRow + 4 * v) = w;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            uint32_t rs = rown + (uint32_t)(sub * DS); if (rs >= L) rs -= L;
            md_rows<Q_T, true, PH64, 0>(sRow, DS, a.wtab_g + 8 * sub * DS, avg, f0, rs, L, outrow ? 1.f : 0.f, a.nd_base, acc, dcs);
            sx += (int)dcs.x; sy += (int)dcs.y; dcs = (float2v){0.f, 0.f};
        }
        float yr = acc[H].x, yi = acc[H].y;
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int k = H - q, src = (lane - k) & 63;
            const float cr = __shfl(acc[q].x, src), ci = __shfl(acc[q].y, src);
            const float or_ = __shfl(pv[q].x, src), oi = __shfl(pv[q].y, src);
            yr += (lane >= k) ? cr : or_;
            yi += (lane >= k) ? ci : oi;
        }
        if (outrow) yout[(a.m0 + (uint32_t)j) & (uint32_t)(a.ring_len - 1)] = make_float2(yr, yi);
        if (j >= a.nblocks - H && j < a.nblocks) {
#pragma unroll
            for (int q = 0; q < Q_T; q++) a.ptail_out[((size_t)ch * 8 + (j - (a.nblocks - H))) * 8 + q] = make_float2(acc[q].x, acc[q].y);
        }
#pragma unroll
        for (int q = 0; q < Q_T; q++) pv[q] = acc[q];
    }
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    if (lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch), (unsigned long long)(long long)sx);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch + 1), (unsigned long long)(long long)sy);
    }
}

// avg = (float)(sum / (float)maxcnt) with sum = S/32768 exact in double (demod_mod.c:498-503)
// dc_avg_prev (optional): keeps the mean that was in effect up to here (md_dc_boundary corrects the outputs whose taps straddle the change)
__global__ void k_dc_update(int n_ch, long long *dc_sums, float2 *dc_avg, float2 *dc_avg_prev, float maxcnt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    const double sx = (double)dc_sums[2 * c] / 32768.0, sy = (double)dc_sums[2 * c + 1] / 32768.0;
    if (dc_avg_prev) dc_avg_prev[c] = dc_avg[c];
    dc_avg[c] = make_float2((float)(sx / (double)maxcnt), (float)(sy / (double)maxcnt));
    dc_sums[2 * c] = 0; dc_sums[2 * c + 1] = 0;
}
__global__ void k_dc_update_pcs(int n_ch, long long *dc_sums, float2 *dc_avg, float2 *dc_avg_prev, uint32_t *cnt, uint32_t *mx, uint32_t lim, int32_t *since,
                                uint32_t n_samples, int nblocks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    const uint32_t k = cnt[c] + n_samples, m = mx[c];
    if (k >= m) {                                             // the host never lets a launch run past a channel's segment edge: k == m
        const double sx = (double)dc_sums[2 * c] / 32768.0, sy = (double)dc_sums[2 * c + 1] / 32768.0;
        if (dc_avg_prev) dc_avg_prev[c] = dc_avg[c];
        dc_avg[c] = make_float2((float)(sx / (double)(float)m), (float)(sy / (double)(float)m));
        dc_sums[2 * c] = 0; dc_sums[2 * c + 1] = 0;
        cnt[c] = 0; if (m < lim) mx[c] = 2 * m;
        since[c] = 0;
    } else {
        cnt[c] = k;
        const int sn = since[c];
        since[c] = sn < (1 << 20) ? sn + nblocks : sn;
    }
}
// one word of device memory -> pinned host memory (the frame counter after a call's frame sync): a one-lane kernel on the same queue instead of
// a copy-engine round trip at the end of every call
__global__ void k_publish_u32(const unsigned *src, unsigned *dst) { if (threadIdx.x == 0) *dst = *src; }

// ------------------------------------------------------------------------------------------------
// k_u8_to_s16: 8-bit unsigned input (rtl_sdr's native IQ format, 8-bit WAV).  The reference reads x = (u - 128) / 128.0
// (demod_mod.c:397-398,438-439,480-481, dft_detect.c:534-535,575-576,607-608); v = (u - 128) * 256 as int16 gives
// v / 32768 = exactly that x, so the 16-bit path then produces bit-identical samples and IQ-DC sums.
// Strides: bytes per channel in the input, int16 elements per channel in the output.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_u8_to_s16(const uint8_t *in, long long in_stride, int16_t *out, long long out_stride, int n_bytes) {
    const int ch = blockIdx.y;
    const uint8_t *srcb = in + (size_t)ch * in_stride;
    int16_t *dsts = out + (size_t)ch * out_stride;
    const int n_words = n_bytes >> 2;
    if (((reinterpret_cast<uintptr_t>(srcb) & 3) | (reinterpret_cast<uintptr_t>(dsts) & 7)) == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(srcb);
        uint2 *dst = reinterpret_cast<uint2 *>(dsts);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
            const uint32_t w = src[i] ^ 0x80808080u;                                                // u - 128 as signed bytes
            dst[i] = make_uint2(((w & 0xffu) << 8) | ((w & 0xff00u) << 16), ((w >> 8) & 0xff00u) | (w & 0xff000000u));
        }
    } else {                                                                                        // odd strides: bytewise
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n_words; i += gridDim.x * blockDim.x)
            dsts[i] = (int16_t)(((int)srcb[i] - 128) * 256);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_bytes & 3)) {
        const int i = 4 * n_words + threadIdx.x;
        dsts[i] = (int16_t)(((int)srcb[i] - 128) * 256);
    }
}

// ------------------------------------------------------------------------------------------------
// k_audio_chain: FM-audio input (dsp.opt_iq = 0): optional FM low-pass, then fm_buffer and bufs (demod_mod.c:836-852)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_audio_chain(const AudioChainArgs a) {
    const int ch = blockIdx.y;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const float *raw = a.raw + (size_t)ch * a.ring_len;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        const uint32_t m = a.m0 + (uint32_t)i;
        float s;
        if (a.taps > 0) {                       // s_fm = sum_k w[k] * raw[m-(T-1)+k], zero history before the stream (re_lowpass :711-719)
            s = 0.f;
            for (int k = 0; k < a.taps; k++) {
                const int64_t p = (int64_t)m - (a.taps - 1) + k;
                s = fmaf(a.w[k], p >= 0 ? raw[(uint32_t)p & mask] : 0.f, s);
            }
        } else s = raw[m & mask];
        a.fm[(size_t)ch * a.ring_len + (m & mask)] = s;
        a.bufs[(size_t)ch * a.ring_len + (m & mask)] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// float32 input: k_mix_f32 / k_decimate_f32 / k_dc_update_f64 (see MixF32Args).  Correct-by-construction, not tuned: cf32 is
// 8 B per sample and rare (the reference's usual sources are cs16 and cu8).
// ------------------------------------------------------------------------------------------------
// Round 6: a workgroup takes MF32_CHUNK consecutive samples of one channel (coalesced, 32 per thread), keeps the mixer-table index by addition instead of a 64-bit
// modulo per sample, and adds its IQ-DC sums with ONE pair of double atomics (round 1..5: a pair per wave, 8192 per channel and launch on two addresses — the kernel
// took 4 ms for 64 channels x 1 s, bench side_paths).
#define MF32_CHUNK 8192
__global__ __launch_bounds__(256)
void k_mix_f32(const MixF32Args a) {
    const int ch = blockIdx.y;
    const float2 *x = a.x + (size_t)ch * a.ch_stride;
    float2 *z = a.z + (size_t)ch * ((size_t)a.zmask + 1);
    float2 avg = a.dc_avg[ch];
    const float2 *seg = a.dc_seg ? a.dc_seg + (size_t)ch * a.dc_seg_n : nullptr;
    const double f0 = a.mix ? a.chan_f0[ch] : 0.0;
    const uint32_t L = (uint32_t)a.lut_len;
    const double nd0 = a.nd_base - (a.epoch ? (double)a.epoch[ch] : 0.0);
    double sx = 0.0, sy = 0.0;
    const int i_first = blockIdx.x * MF32_CHUNK + threadIdx.x, i_end = min(a.n, (int)(blockIdx.x + 1) * MF32_CHUNK);
    uint32_t k = i_first < a.n ? (uint32_t)(((uint64_t)a.lut_phase + (uint64_t)i_first) % L) : 0u;      // table index (demod_mod.c:746), advanced by 256 per step
    const uint32_t kstep = 256u % L;
    for (int i = i_first; i < i_end; i += 256) {
        const float2 v = x[i];
        sx += (double)v.x; sy += (double)v.y;
        if (seg) avg = seg[(a.dc_seg_off + (uint32_t)i) / a.dc_seg_len];
        float2 u = make_float2(v.x - avg.x, v.y - avg.y);
        if (a.mix) {
            const double nd = (double)k + nd0;
            float fr;
            if (a.phase_f64) fr = (float)__builtin_amdgcn_fract(f0 * nd);
            else             fr = __builtin_amdgcn_fractf((float)(f0 * nd));
            const float c = __builtin_amdgcn_cosf(fr), s = __builtin_amdgcn_sinf(fr);
            u = make_float2(u.x * c - u.y * s, u.x * s + u.y * c);
            k += kstep; if (k >= L) k -= L;
        }
        z[(uint32_t)(a.n0 + (uint64_t)i) & a.zmask] = u;
    }
    if (seg) return;                                           // the windows' sums were taken by k_dc_seg_sums_f32
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    __shared__ double s_d[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_d[2 * wave] = sx; s_d[2 * wave + 1] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0.0, ty = 0.0;
        for (int w = 0; w < 4; w++) { tx += s_d[2 * w]; ty += s_d[2 * w + 1]; }
        atomicAdd(a.dc_sums + 2 * (size_t)ch, tx); atomicAdd(a.dc_sums + 2 * (size_t)ch + 1, ty);
    }
}

// float32 form of k_dc_seg_sums / k_dc_seg_means: the IQ-DC windows a call touches, summed in double (one workgroup per window and channel),
// then per channel the table of means — window k under the mean of window k-1, the window in progress carried over in dc_sums.
__global__ __launch_bounds__(256)
void k_dc_seg_sums_f32(const float2 *x, long long ch_stride, int n_samples, unsigned dc_cnt0, unsigned dc_max, double *seg_sums, int nseg) {
    const int k = blockIdx.x, ch = blockIdx.y;
    const long long lo = (long long)k * dc_max - dc_cnt0, hi = lo + dc_max;
    const int s0 = (int)(lo < 0 ? 0 : lo), s1 = (int)(hi > n_samples ? n_samples : hi);
    const float2 *p = x + (size_t)ch * ch_stride;
    double sx = 0.0, sy = 0.0;
    for (int i = s0 + (int)threadIdx.x; i < s1; i += 256) { const float2 v = p[i]; sx += (double)v.x; sy += (double)v.y; }
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    __shared__ double s_d[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_d[2 * wave] = sx; s_d[2 * wave + 1] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0.0, ty = 0.0;
        for (int w = 0; w < 4; w++) { tx += s_d[2 * w]; ty += s_d[2 * w + 1]; }
        seg_sums[((size_t)ch * nseg + k) * 2] = tx; seg_sums[((size_t)ch * nseg + k) * 2 + 1] = ty;
    }
}

__global__ void k_dc_seg_means_f32(int n_ch, int nseg, int ncomplete, float maxcnt, const double *seg_sums, double *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    float2 mean = dc_avg[c];
    double cx = dc_sums[2 * c], cy = dc_sums[2 * c + 1];       // what the window in progress had collected before this call
    for (int k = 0; k < dc_seg_n; k++) {
        dc_seg[(size_t)c * dc_seg_n + k] = mean;
        if (k >= nseg) continue;
        cx += seg_sums[((size_t)c * nseg + k) * 2]; cy += seg_sums[((size_t)c * nseg + k) * 2 + 1];
        if (k < ncomplete) { mean = make_float2((float)(cx / (double)maxcnt), (float)(cy / (double)maxcnt)); cx = 0.0; cy = 0.0; }      // k_dc_update_f64
    }
    dc_avg[c] = mean; dc_sums[2 * c] = cx; dc_sums[2 * c + 1] = cy;
}

// Round 6: DF32_J consecutive outputs per workgroup, their D (DF32_J - 1) + T input samples staged in LDS with coalesced loads (round 1..5: every thread walked its T taps
// through global memory at a stride of D samples between lanes — 64 cache lines per load, 24 x the cs16 path per channel-second, bench side_paths).  One thread per
// output and the taps in ascending order, as before: the same fused multiply-adds on the same operands, results unchanged to the bit.
#define DF32_J 128
__global__ __launch_bounds__(256)
void k_decimate_f32_global(const DecF32Args a) {                          // (decimation factors whose tile does not fit the LDS: wide IF rates, rare)
    const int ch = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nblocks) return;
    const float2 *z = a.z + (size_t)ch * ((size_t)a.zmask + 1);
    const int64_t first = (int64_t)a.n0 + (int64_t)a.D * (j + 1) - a.T;
    float re = 0.f, im = 0.f;
    for (int k = 0; k < a.T; k++) {
        const int64_t n = first + k;
        if (n < 0) continue;
        const float2 v = z[(uint32_t)n & a.zmask];
        const float w = a.taps[k];
        re = fmaf(w, v.x, re); im = fmaf(w, v.y, im);
    }
    a.y[(size_t)ch * a.ring_len + ((a.m0 + (uint32_t)j) & ((uint32_t)a.ring_len - 1))] = make_float2(re, im);
}
__global__ __launch_bounds__(DF32_J)
void k_decimate_f32(const DecF32Args a, const int jper) {                 // jper <= DF32_J outputs per workgroup (what fits the LDS)
    extern __shared__ __attribute__((aligned(16))) float2 s_z[];          // [D (jper - 1) + T] samples first(j0) ..., then [T] taps as floats
    const int ch = blockIdx.y, j0 = blockIdx.x * jper, tid = threadIdx.x;
    const int nj = min(jper, a.nblocks - j0);
    if (nj <= 0) return;
    const float2 *z = a.z + (size_t)ch * ((size_t)a.zmask + 1);
    // y[m] = sum_k w[k] z[D (m+1) - T + k]; samples before the stream are zero (calloc'ed delay line)
    const int64_t first0 = (int64_t)a.n0 + (int64_t)a.D * (j0 + 1) - a.T;
    const int nz = a.D * (nj - 1) + a.T;
    float *s_w = reinterpret_cast<float *>(s_z + a.D * (jper - 1) + a.T);
    for (int i = tid; i < nz; i += DF32_J) { const int64_t n = first0 + i; s_z[i] = n < 0 ? make_float2(0.f, 0.f) : z[(uint32_t)n & a.zmask]; }
    for (int k = tid; k < a.T; k += DF32_J) s_w[k] = a.taps[k];
    __syncthreads();
    if (tid >= nj) return;
    const int64_t first = first0 + (int64_t)a.D * tid;
    const float2 *p = s_z + a.D * tid;
    float re = 0.f, im = 0.f;
    int k = first < 0 ? (int)min((int64_t)a.T, -first) : 0;            // (taps that would pair with samples before the stream are skipped, not multiplied by zero: as before)
    // eight taps at a time: their LDS reads are in flight together, the multiply-adds follow in tap order (a loop of one read and one dependent FMA pair per
    // iteration waited a full LDS round trip per tap)
    for (; k + 8 <= a.T; k += 8) {
        float2 v[8]; float w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { v[u] = p[k + u]; w[u] = s_w[k + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { re = fmaf(w[u], v[u].x, re); im = fmaf(w[u], v[u].y, im); }
    }
    for (; k < a.T; k++) {
        const float2 v = p[k];
        const float w = s_w[k];
        re = fmaf(w, v.x, re); im = fmaf(w, v.y, im);
    }
    a.y[(size_t)ch * a.ring_len + ((a.m0 + (uint32_t)(j0 + tid)) & ((uint32_t)a.ring_len - 1))] = make_float2(re, im);
}

__global__ void k_dc_update_f64(int n_ch, double *dc_sums, float2 *dc_avg, float maxcnt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    dc_avg[c] = make_float2((float)(dc_sums[2 * c] / (double)maxcnt), (float)(dc_sums[2 * c + 1] / (double)maxcnt));   // sumIQx/(float)maxcnt
    dc_sums[2 * c] = 0.0; dc_sums[2 * c + 1] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// k_afc_rotate (--dc): yrot[m] = y[m] * cexp(-t 2 pi Df), t = m / sr, for m in [start[ch], m_end)   (demod_mod.c:758-761)
// The reference multiplies the float sample by a double phasor and rounds once; Df is piecewise constant in time (it
// changes at header detections), so the rotated stream is kept in its own ring = the IF filter's delay line lpIQ_buf.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_afc_rotate(const AfcRotArgs a) {
    const int ch = blockIdx.y;
    const uint32_t start = a.start[ch];
    const int n = (int32_t)(a.m_end - start);
    if (n <= 0) return;
    const double Df = a.afc[ch].Df;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const float2 *y = a.y + (size_t)ch * a.ring_len;
    float2 *yr = a.yrot + (size_t)ch * a.ring_len;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t m = start + (uint32_t)i;
        const float2 z = y[m & mask];
        float2 r = z;
        if (Df != 0.0) {
            const double t = (double)m / (double)a.sr;
            const double ph = -t * 6.2831853071795864769 * Df;
            double sn, cs;
            sincos(ph, &sn, &cs);
            r = make_float2((float)((double)z.x * cs - (double)z.y * sn), (float)((double)z.x * sn + (double)z.y * cs));
        }
        yr[m & mask] = r;
    }
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// k_if_chain: one workgroup = IF_TILE output samples of one channel
// ------------------------------------------------------------------------------------------------
#ifndef IF_TILE
#define IF_TILE 960
#endif
#ifndef IF_THREADS
#define IF_THREADS 256
#endif
#ifndef IF_NB
#define IF_NB 4
#endif
#define IF_RUN 4
#ifndef IF_LD
#define IF_LD 5
#endif

// (the body serves two kernels: k_if_chain — one sonde type, arguments in the kernel argument segment — and k_if_chain_multi, the channel groups of a mixed engine in one launch)
__device__ __forceinline__ void if_chain_body(const IfArgs &a, const int ch, const int bx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t t0 = a.m0 + (uint32_t)bx * IF_TILE;        // first output sample (absolute)
    const int nout = min(IF_TILE, (int)(a.m0 + (uint32_t)a.n - t0));
    if (nout <= 0) return;
    const bool afc = a.afc != nullptr;
    const uint32_t start = afc ? a.start[ch] : t0;    // --dc restart: nothing below start[ch] is recomputed
    if (afc && (int32_t)(t0 + (uint32_t)nout - start) <= 0) return;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const int T1 = a.lpiq_on ? a.lpiq_taps : 1;       // IF low-pass taps
    const int T2 = a.lpfm_on ? a.lpfm_taps : 1;       // FM low-pass taps
    const int nwin = a.nwin;                          // tone window = (int)sps
    const int64_t ep = a.epoch ? (int64_t)a.epoch[ch] : 0;   // this channel's stream start: phase origin of the tone mixer (history before it is zeroed by the host)
    // sample ranges (relative to t0): s_fm needed on [-(T2-1), nout); z' on [-(T2-1)-max(1,nwin-1)..]
    const int hz = (T2 - 1) + max(1, nwin - 1);       // history of z' needed
    const int nz = hz + nout;                         // z' count
    const int ny = nz + (T1 - 1);                     // y count
    const int nyp = (ny + 2 * IF_NB + 1) & ~1;                 // padded: the IF_NB-output groups read a little past ny
    float2 *sy = reinterpret_cast<float2 *>(smem);            // [nyp]
    const int nzs = a.fm_on ? nz + (nz & 1) : 0;               // z' is kept for the discriminator only
    float2 *sz = sy + nyp;                                     // [nz]   z'[t0 - hz + k]
    float4 *sx4 = reinterpret_cast<float4 *>(sz + nzs);        // [nz]   (X1, X2): X1 = z' * e^{+i 2 pi m rho}, X2 = z' * e^{-i 2 pi m rho}
    float  *sf = reinterpret_cast<float *>(sx4 + (a.tone_on ? nz : 0));   // [T2-1+nout] raw s_fm (the tone products exist with the tone correlator only)
    const int nsf = a.fm_on ? T2 - 1 + nout : 0;
    float  *wf = sf + nsf;                                     // [T2]
    float  *wq = sf + ((nsf + T2 + 3) & ~3);                   // [T1] IF low-pass taps, 16-byte aligned (sf is): read 4 at a time

    const float2 *yr = a.y + (size_t)ch * a.ring_len;
    // the tile's y samples: IF_LD loads per thread in flight before the first is parked (a workgroup has nothing else to do until they are there)
    for (int kb = threadIdx.x; kb < nyp; kb += IF_LD * IF_THREADS) {
        float2 v[IF_LD];
#pragma unroll
        for (int u = 0; u < IF_LD; u++) {
            const int k = kb + u * IF_THREADS;
            const int64_t m = (int64_t)t0 - hz - (T1 - 1) + k;     // absolute IF index, may be < 0 at stream start
            v[u] = (m >= 0 && k < ny) ? yr[(uint32_t)m & mask] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < IF_LD; u++) { const int k = kb + u * IF_THREADS; if (k < nyp) sy[k] = v[u]; }
    }
    // acquisition / locked tap set (demod_mod.c:1577-1590); the choice is per channel = per workgroup: kept in a scalar register so that the taps load as scalars
    const int acq = __builtin_amdgcn_readfirstlane((afc && !a.afc[ch].locked) ? 1 : 0);
    const float *w_iq = acq ? a.w_iq0 : a.w_iq;
    for (int k = threadIdx.x; k < T2; k += IF_THREADS) wf[k] = a.lpfm_on ? a.w_fm[k] : 1.0f;
    for (int k = threadIdx.x; k < T1; k += IF_THREADS) wq[k] = a.lpiq_on ? w_iq[k] : 1.0f;
    typedef const float __attribute__((address_space(4))) *cfptr;
    const cfptr wtaps = (cfptr)(uintptr_t)w_iq;        // the taps of the 4-tap loop come through the scalar cache (that loop only runs with a filter: T1 >= 4); the tail loops read the LDS copy
    __syncthreads();

    // IF low-pass: z'[m] = sum_k w[k] * y[m-(T1-1)+k]   (oldest sample pairs with tap 0, demod_mod.c:639-648).
    // IF_NB consecutive outputs per thread with a sliding register window: one 16-byte LDS read per 2 taps and IF_NB outputs; the taps are
    // wave-uniform: four at a time as one broadcast LDS read; (re, im) pairs accumulate with packed FMAs — per component the same fused multiply-adds in the
    // same tap order as before.  The tone phasors e^{+-i 2 pi m rho} are applied once per sample here (X1, X2), not once per window term.
    typedef float v2f __attribute__((ext_vector_type(2)));
    for (int k0 = IF_NB * threadIdx.x; k0 < nz; k0 += IF_NB * IF_THREADS) {
        v2f acc[IF_NB];
#pragma unroll
        for (int j = 0; j < IF_NB; j++) acc[j] = v2f{0.f, 0.f};
        v2f win[IF_NB + 2];
#pragma unroll
        for (int j = 0; j < IF_NB; j += 2) {
            const float4 v0 = *reinterpret_cast<const float4 *>(sy + k0 + j);
            win[j] = v2f{v0.x, v0.y}; win[j + 1] = v2f{v0.z, v0.w};
        }
        int t = 0;
#ifdef IF_EXP_NOFIR
        t = T1 & ~3;
#endif
        for (; t + 3 < T1; t += 4) {                                   // 4 taps: two 16-byte sample reads, one 16-byte (broadcast) tap read
            const float4 n0 = *reinterpret_cast<const float4 *>(sy + k0 + t + IF_NB), n1 = *reinterpret_cast<const float4 *>(sy + k0 + t + IF_NB + 2);
            const float4 w4 = make_float4(wtaps[t], wtaps[t + 1], wtaps[t + 2], wtaps[t + 3]);      // wave-uniform, constant address space: one scalar load instead of a (broadcast) LDS read — round 6, -2.5 %
            win[IF_NB] = v2f{n0.x, n0.y}; win[IF_NB + 1] = v2f{n0.z, n0.w};
#pragma unroll
            for (int j = 0; j < IF_NB; j++) {
                acc[j] = __builtin_elementwise_fma(win[j], v2f{w4.x, w4.x}, acc[j]);
                acc[j] = __builtin_elementwise_fma(win[j + 1], v2f{w4.y, w4.y}, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < IF_NB; j++) win[j] = win[j + 2];
            win[IF_NB] = v2f{n1.x, n1.y}; win[IF_NB + 1] = v2f{n1.z, n1.w};
#pragma unroll
            for (int j = 0; j < IF_NB; j++) {
                acc[j] = __builtin_elementwise_fma(win[j], v2f{w4.z, w4.z}, acc[j]);
                acc[j] = __builtin_elementwise_fma(win[j + 1], v2f{w4.w, w4.w}, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < IF_NB; j++) win[j] = win[j + 2];
        }
        for (; t + 1 < T1; t += 2) {
            const float4 nv = *reinterpret_cast<const float4 *>(sy + k0 + t + IF_NB);
            win[IF_NB] = v2f{nv.x, nv.y}; win[IF_NB + 1] = v2f{nv.z, nv.w};
            const float w0 = wq[t], w1 = wq[t + 1];
#pragma unroll
            for (int j = 0; j < IF_NB; j++) {
                acc[j] = __builtin_elementwise_fma(win[j], v2f{w0, w0}, acc[j]);
                acc[j] = __builtin_elementwise_fma(win[j + 1], v2f{w1, w1}, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < IF_NB; j++) win[j] = win[j + 2];
        }
        if (T1 == 1) {
            // no IF low-pass (or a one-tap one): the sample times its weight, NOT 0 + sample * 1 — the sum would turn a -0 into +0, and the discriminator below
            // tells them apart (atan2(+-0, +-0) = 0 / pi / -pi / -0): exactly-zero samples (8-bit input at 128 / 128) behind the AFC rotation carry signed zeros, and
            // the reference's discriminator sees them (found by tests/fuzz/fuzz_chunks.py: 0.8 instead of 0 in 1.4 % of the noise samples moved a header-search arg-max)
            const float w0 = wq[0];
#pragma unroll
            for (int j = 0; j < IF_NB; j++) acc[j] = win[j] * v2f{w0, w0};
        } else if (T1 & 1) {
            const float w0 = wq[T1 - 1];
#pragma unroll
            for (int j = 0; j < IF_NB; j++) acc[j] = __builtin_elementwise_fma(win[j], v2f{w0, w0}, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < IF_NB; j++) {
            const int k = k0 + j;
            if (k >= nz) break;
            float re = acc[j].x, im = acc[j].y;
            const int64_t m = (int64_t)t0 - hz + k;
            if (m < 0) { re = 0.f; im = 0.f; }
            else if (afc && (int32_t)((uint32_t)m - start) < 0) {          // older than the restart: rot_iqbuf as it stands
                const float2 zo = a.tap_ifiq[(size_t)ch * a.ring_len + ((uint32_t)m & mask)];
                re = zo.x; im = zo.y;
            }
            if (a.fm_on) sz[k] = make_float2(re, im);
            if (a.tone_on) {
                // tone mixer e^{-i t w}, t = m/sr: phase in revolutions = m * rho (double), reduced before the f32 sincos
#ifdef IF_EXP_NOTONE
                const float sn = 0.f, cs = 1.f;
#else
                const float fr = (float)__builtin_amdgcn_fract((double)(m - ep) * a.rho);
                const float sn = __builtin_amdgcn_sinf(fr), cs = __builtin_amdgcn_cosf(fr);   // revolutions in, abs error ~2e-7
#endif
                // X1 = z * e^{+i 2pi fr}; X2 = z * e^{-i 2pi fr}  (iw1 = 2 pi i f1, f1 < 0, demod_mod.c:796-803,1467-1470); stored side by side
                // (which product of a*b + c*d is rounded before the fused multiply-add is the compiler's choice under contraction, and it chose differently in the
                // two kernels that share this body — one ulp apart; spelled out: the cosine products are rounded, the sine products fused)
                const float rc = re * cs, ic = im * cs;
                sx4[k] = make_float4(__builtin_fmaf(-sn, im, rc), __builtin_fmaf(sn, re, ic), __builtin_fmaf(sn, im, rc), __builtin_fmaf(-sn, re, ic));
            }
            if (a.tap_ifiq && m >= (int64_t)t0 && (int32_t)((uint32_t)m - start) >= 0)
                a.tap_ifiq[(size_t)ch * a.ring_len + ((uint32_t)m & mask)] = make_float2(re, im);
        }
    }
    __syncthreads();

    // FM discriminator on [-(T2-1), nout): s_fm = 0.8 * arg(z[m] * conj(z[m-1])) / pi   (demod_mod.c:771-773)
    if (a.fm_on)
    for (int k = threadIdx.x; k < T2 - 1 + nout; k += IF_THREADS) {
        const int zi = k + (hz - (T2 - 1));            // index into sz of sample m
        const float2 z1 = sz[zi], z0 = sz[zi - 1];
        const float wr = z1.x * z0.x + z1.y * z0.y, wi = z1.y * z0.x - z1.x * z0.y;
#ifdef IF_EXP_NOATAN
        float v = wi + wr;
#else
        float v = 0.8f * atan2f(wi, wr) * 0.31830988618379067f;
#endif
        if (afc) {                                     // raw FM samples older than the restart come from lpFM_buf's ring
            const int64_t m = (int64_t)t0 - (T2 - 1) + k;
            float *fr = a.fmraw + (size_t)ch * a.ring_len;
            if (m >= 0 && (int32_t)((uint32_t)m - start) < 0) v = fr[(uint32_t)m & mask];
            else if (m >= (int64_t)t0) fr[(uint32_t)m & mask] = v;
        }
        sf[k] = v;
    }
    __syncthreads();

    float *bufs = a.bufs + (size_t)ch * a.ring_len;
    float *fmb = a.fm + (size_t)ch * a.ring_len;
    // two-tone correlator: windowed sums over the last nwin samples (the reference keeps them as recursive sliding sums over the whole stream,
    // demod_mod.c:796-803 — same value up to its float drift).  Each thread takes IF_RUN consecutive outputs: a full window sum for the first,
    // then + newest - oldest for the next IF_RUN - 1 (the run is short, so no drift builds up: <= 2 (IF_RUN - 1) roundings on top of the sum's own).
    const float inv_sps = 1.0f / a.sps;
    for (int k0 = IF_RUN * threadIdx.x; k0 < nout; k0 += IF_RUN * IF_THREADS) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        float so[IF_RUN], sfm[IF_RUN];
        // a full run at a 16-byte boundary of the ring leaves as one store per stream (the ring length is a power of two >= IF_RUN: no wrap inside it)
        const bool vec = !afc && k0 + IF_RUN <= nout && ((t0 + (uint32_t)k0) & (IF_RUN - 1)) == 0;
#pragma unroll
        for (int r = 0; r < IF_RUN; r++) {
            const int k = k0 + r;
            so[r] = 0.f; sfm[r] = 0.f;
            if (k >= nout) break;
            const uint32_t m = t0 + (uint32_t)k;
            if (a.tone_on) {
#ifdef IF_EXP_NOWIN
                if (1) { f = sx4[hz + k]; } else
#endif
                if (r == 0) {
                    for (int j = nwin - 1; j >= 0; j--) { const float4 x = sx4[hz + k - j]; f.x += x.x; f.y += x.y; f.z += x.z; f.w += x.w; }
                } else {
                    const float4 xn = sx4[hz + k], xo = sx4[hz + k - nwin];
                    f.x += xn.x - xo.x; f.y += xn.y - xo.y; f.z += xn.z - xo.z; f.w += xn.w - xo.w;
                }
            }
            if (afc && (int32_t)(m - start) < 0) continue;
            float s_fm = 0.f;
            if (a.fm_on) {
                s_fm = sf[T2 - 1 + k];
                if (a.lpfm_on) {
                    float acc = 0.f;
                    for (int t = 0; t < T2; t++) acc = fmaf(sf[k + t], wf[t], acc);
                    s_fm = acc;
                }
                if (!vec) fmb[m & mask] = s_fm;
            }
            float s = s_fm;
            // |F2| - |F1| scaled by 1/sps: hardware square root and a reciprocal multiply (1 ulp each — the reference itself evaluates this in
            // double from drifting float sums; the tolerance of the stream is 1e-5 RMS, tests/test_gpu_parity.py)
            if (a.tone_on) s = (__builtin_amdgcn_sqrtf(f.z * f.z + f.w * f.w) - __builtin_amdgcn_sqrtf(f.x * f.x + f.y * f.y)) * inv_sps;
            if (!vec) bufs[m & mask] = s;
            so[r] = s; sfm[r] = s_fm;
        }
        if (vec) {
            const uint32_t mi = (t0 + (uint32_t)k0) & mask;
            static_assert(IF_RUN == 4, "the vector store below writes four outputs");
            *reinterpret_cast<float4 *>(bufs + mi) = make_float4(so[0], so[1], so[2], so[3]);
            if (a.fm_on) *reinterpret_cast<float4 *>(fmb + mi) = make_float4(sfm[0], sfm[1], sfm[2], sfm[3]);
        }
    }
}

__global__ __launch_bounds__(IF_THREADS)
void k_if_chain(const IfArgs a) { if_chain_body(a, blockIdx.y, blockIdx.x); }
// Mixed engines: the groups (one sonde type each: own taps, tone spacing, window) side by side in ONE launch; blockIdx.y = row of the engine, rows grouped by type
template <class A> struct MultiArgs { A g[SONDE_MAX_GROUPS]; int row0[SONDE_MAX_GROUPS + 1]; int n_groups; };
template <class A> __device__ __forceinline__ int multi_group(const MultiArgs<A> &m, const int row) {
    int g = 0;
    while (g + 1 < m.n_groups && row >= m.row0[g + 1]) g++;
    return __builtin_amdgcn_readfirstlane(g);
}
__global__ __launch_bounds__(IF_THREADS)
void k_if_chain_multi(const MultiArgs<IfArgs> m) {
    const int g = multi_group(m, (int)blockIdx.y);
    if_chain_body(m.g[g], (int)blockIdx.y - m.row0[g], blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_header_corr: c[p] = sum_u match[u] * bufs[p-(L-1)+u]   for p in [m0, m0+n)
// ------------------------------------------------------------------------------------------------
// The reference correlates only while find_header() runs: not during the nbits a framer slices after a hit, and a
// window only reaches back K+delay samples.  The channel's sync state (left by the previous k_framesync) gives the
// first end position any future window can examine; correlation tiles entirely below it are skipped.
// Two passes per call (sonde_engine.cpp): a sonde that is being received spends 85 % of the time inside a frame, where no window is
// examined — but WHERE the next frame starts is only known once the sync has run.  Pass 1 correlates the `limit` samples behind
// `first` (two windows) and the sync stops at that horizon (sync_corr_horizon); pass 2 repeats both with the state pass 1 left: for a
// channel that found its header, everything up to the end of the new frame is now skipped; a channel that is still searching gets
// the rest of the call.  Same decisions as one pass (the sync is resumable at any sample — it is how calls are chained anyway).
__device__ __forceinline__ uint32_t sync_first_pos(const SyncState &st, uint32_t frame_samples, int delay) {
    uint32_t first;                                           // earliest candidate end position of the next window
    if (st.mode == 1) first = st.mv_pos + 1 + frame_samples;                    // s_in_after - delay (frame in progress)
    else first = st.s_in - st.k;                                                // window start of the running search
    return first - (uint32_t)(delay + 16);
}
__device__ __forceinline__ bool corr_tile_unused(const CorrArgs &a, int ch, uint32_t tile_start, uint32_t tile_end) {
    if (a.start && (int32_t)(tile_end - a.start[ch]) <= 0) return true;        // --dc restart: bufs below start[ch] did not change
    if (!a.state) return false;
    const SyncState st = a.state[ch];
    if (st.mode != 0 && st.mode != 1) return true;                              // stream finished
    const uint32_t first = sync_first_pos(st, a.frame_samples, a.delay);
    if (a.limit && (int32_t)(tile_start - (first + a.limit)) >= 0) return true; // pass 1: beyond the horizon
    return (int32_t)(tile_end - first) <= 0;
}

#define HC_TILE 1024
#define HC_THREADS 256

__global__ __launch_bounds__(HC_THREADS)
void k_header_corr(const CorrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ch = blockIdx.y, L = a.L;
    const uint32_t p0 = a.m0 + (uint32_t)blockIdx.x * HC_TILE;
    const int nout = min(HC_TILE, (int)(a.m0 + (uint32_t)a.n - p0));
    if (nout <= 0) return;
    if (corr_tile_unused(a, ch, p0, p0 + (uint32_t)nout)) return;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    float *sx = smem;                 // [HC_TILE + L - 1]
    float *sm = smem + HC_TILE + L;   // [L]
    const float *bufs = a.bufs + (size_t)ch * a.ring_len;
    for (int k = threadIdx.x; k < HC_TILE + L - 1; k += HC_THREADS) {
        const int64_t m = (int64_t)p0 - (L - 1) + k;
        sx[k] = (m >= 0 && k < nout + L - 1) ? bufs[(uint32_t)m & mask] : 0.f;
    }
    for (int k = threadIdx.x; k < L; k += HC_THREADS) sm[k] = a.match[k];
    __syncthreads();
    float acc[HC_TILE / HC_THREADS];
#pragma unroll
    for (int j = 0; j < HC_TILE / HC_THREADS; j++) acc[j] = 0.f;
    for (int u = 0; u < L; u++) {
        const float mu = sm[u];
#pragma unroll
        for (int j = 0; j < HC_TILE / HC_THREADS; j++) acc[j] = fmaf(mu, sx[threadIdx.x + j * HC_THREADS + u], acc[j]);
    }
    float *corr = a.corr + (size_t)ch * a.ring_len;
#pragma unroll
    for (int j = 0; j < HC_TILE / HC_THREADS; j++) {
        const int k = threadIdx.x + j * HC_THREADS;
        if (k < nout) corr[(p0 + (uint32_t)k) & mask] = acc[j];
    }
}

// Factorised form for integer samples/symbol (sps): the template is, symbol by symbol, one of a few
// `shapes` (own bit +-1 and the two neighbour bits, demod_mod.c:1398-1416), so
//   c[p] = sum_k sign_k * F_{type_k}[p-(L-1)+sps*k],   F_t[n] = sum_{d<sps} shape_t[d] * bufs[n+d]
// ~ (types*sps + symbols) operations per sample instead of L.  The shapes are taken from the very same
// match[] floats, so the result differs from the direct sum only by float association.
#define HCF_TILE 1024
#define HCF_THREADS 512
#define HCF_MAXTYPES 9

__global__ __launch_bounds__(HCF_THREADS)
void k_header_corr_fact(const CorrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ch = blockIdx.y, L = a.L, sps = a.isps, nsym = a.nsym, nt = a.ntypes;
    const uint32_t p0 = a.m0 + (uint32_t)blockIdx.x * HCF_TILE;
    const int nout = min(HCF_TILE, (int)(a.m0 + (uint32_t)a.n - p0));
    if (nout <= 0) return;
    if (corr_tile_unused(a, ch, p0, p0 + (uint32_t)nout)) return;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const int nx = HCF_TILE + L - 1;                 // samples p0-(L-1) .. p0+HCF_TILE-1
    const int nf = (HCF_TILE + sps * (nsym - 1) + 3) & ~3;   // F entries per type (multiple of 4)
    const int nxp = (nf + sps + 3 + 3) & ~3;
    float *sx = smem;                                // [nxp]
    float *sF = smem + nxp;                          // [nt][nf]
    const float *bufs = a.bufs + (size_t)ch * a.ring_len;
    for (int k = threadIdx.x; k < nxp; k += HCF_THREADS) {
        const int64_t m = (int64_t)p0 - (L - 1) + k;
        sx[k] = (m >= 0 && k < nout + L - 1 && k < nx) ? bufs[(uint32_t)m & mask] : 0.f;
    }
    __syncthreads();
    // F_t[n] = sum_d shape_t[d] * x[n+d]: 4 consecutive n per thread from 16-byte LDS reads, all types at once
    for (int n4 = threadIdx.x * 4; n4 < nf; n4 += HCF_THREADS * 4) {
        float xv[4 + 16];                            // sps <= 16
        const int nload = (sps + 3 + 3) / 4;         // float4 chunks covering x[n4 .. n4+sps+2]
#pragma unroll
        for (int c = 0; c < 5; c++) {
            if (c < nload) {
                const float4 v = *reinterpret_cast<const float4 *>(sx + n4 + 4 * c);
                xv[4 * c] = v.x; xv[4 * c + 1] = v.y; xv[4 * c + 2] = v.z; xv[4 * c + 3] = v.w;
            }
        }
        for (int t = 0; t < nt; t++) {
            const float *sh = a.shapes + t * sps;    // uniform -> scalar loads
            float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
#pragma unroll
            for (int d = 0; d < 16; d++) {
                if (d < sps) {
                    const float w = sh[d];
                    f0 = fmaf(w, xv[d], f0); f1 = fmaf(w, xv[d + 1], f1); f2 = fmaf(w, xv[d + 2], f2); f3 = fmaf(w, xv[d + 3], f3);
                }
            }
            *reinterpret_cast<float4 *>(sF + t * nf + n4) = make_float4(f0, f1, f2, f3);
        }
    }
    __syncthreads();
    // c[o] = sum_k sign_k * F_{type_k}[o + sps*k]: 2 consecutive outputs per thread and pass — lanes then read
    // consecutive 8-byte words (no LDS bank conflicts; 4 outputs per thread made every read 2-way conflicted)
    float *corr = a.corr + (size_t)ch * a.ring_len;
#if HCF_TILE / HCF_THREADS >= 2
    for (int o2 = threadIdx.x * 2; o2 < HCF_TILE; o2 += 2 * HCF_THREADS) {
        float c0 = 0.f, c1 = 0.f;
        for (int k = 0; k < nsym; k++) {
            const int ty = a.sym_type[k];                // uniform: scalar loads
            const float sg = a.sym_sign[k];
            const float2 u = *reinterpret_cast<const float2 *>(sF + ty * nf + sps * k + o2);     // sps even -> 8-byte aligned
            c0 = fmaf(sg, u.x, c0); c1 = fmaf(sg, u.y, c1);
        }
        if (o2 < nout) corr[(p0 + (uint32_t)o2) & mask] = c0;
        if (o2 + 1 < nout) corr[(p0 + (uint32_t)(o2 + 1)) & mask] = c1;
    }
#else
    {
        const int o = threadIdx.x;
        float c0 = 0.f;
        for (int k = 0; k < nsym; k++) c0 = fmaf(a.sym_sign[k], sF[a.sym_type[k] * nf + sps * k + o], c0);
        if (o < nout) corr[(p0 + (uint32_t)o) & mask] = c0;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_framesync: one wave per channel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Consumed-sample window [q0, q1) and centre `mid` of one symbol half of bit `pos`
// (read_softbit2p, demod_mod.c:1098-1161).  The reference walks an integer counter sc against double
// edges: bg = (float)(pos*symlen)*sps (0 for pos 0); per half: mid = bg + (sps-1)/2, bg += sps,
// do {..; sc++} while (sc < bg).  With sps >= 1 every loop ends at sc = ceil(bg), so each half's
// window is a closed form of pos and the bits can be sliced in parallel.
__device__ __forceinline__ void bit_window(int pos, int half, int symlen, float sps,
                                           uint32_t &q0, uint32_t &q1, double &mid) {
    double bg = (pos == 0) ? 0.0 : (double)((float)(pos * symlen) * sps);
    double prev;                                    // edge at which the previous do-while stopped
    if (half == 0) {
        if (pos == 0) prev = 0.0;
        else {
            prev = (pos == 1) ? 0.0 : (double)((float)((pos - 1) * symlen) * sps);
            prev += (double)sps;
            if (symlen == 2) prev += (double)sps;
        }
    } else {
        bg += (double)sps;
        prev = bg;
    }
    q0 = (uint32_t)ceil(prev);
    mid = bg + (double)(sps - 1.0f) / 2.0;
    q1 = (uint32_t)ceil(bg + (double)sps);
    if (q1 <= q0) q1 = q0 + 1;
}

// Sum of the ring samples the slicer selects inside one symbol half, consumed counts q in [qa, qb), added in
// ascending q as doubles (read_softbit2p, demod_mod.c:1139-1161).  The ranges are position independent, so the
// host tabulates them once per engine with the reference's float/double edge arithmetic (sonde_design.cpp
// bit_window / slice_range); all loads of a range are issued up front.
#define SLICE_MAXW 24
// DCSUB (--dc with an FM-sliced stream): sample = (float)(sample - dc) before it is added (demod_mod.c:1150).
template <bool DCSUB>
__device__ __forceinline__ double window_sum(const float *bufs, uint32_t base, uint32_t mask, uint32_t qa, uint32_t qb, double dc) {
    float v[SLICE_MAXW];
#pragma unroll
    for (int j = 0; j < SLICE_MAXW; j++) v[j] = (qa + (uint32_t)j < qb) ? bufs[(base + qa + (uint32_t)j) & mask] : 0.f;
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < SLICE_MAXW; j++) if (qa + (uint32_t)j < qb) sum += DCSUB ? (double)(float)((double)v[j] - dc) : (double)v[j];
    for (uint32_t q = qa + SLICE_MAXW; q < qb; q++) {                                           // very wide symbols
        const float s = bufs[(base + q) & mask];
        sum += DCSUB ? (double)(float)((double)s - dc) : (double)s;
    }
    return sum;
}

#define FS_THREADS 1024           // default workgroup of k_framesync; FS_THREADS_SMALL = the form that fits the slot of one decimator workgroup
#define FS_THREADS_SMALL 256

// One correlation window of getCorrDFT (demod_mod.c:148-225) evaluated from the precomputed correlation ring: arg-max of
// c^2 over the K+1 end positions p = pos-K .. pos (first maximum wins), edge rejection, L-sample norm.
// DC (--dc, :174-188): the reference zeroes bin 0 of the zero-padded N-point transform, i.e. subtracts mu = sum(window)/N
// from every sample including the padding; the circular correlation then drops by mu * sum(match) at every lag and
// the norm runs over (x - mu).  Returns the peak index 0..K, or -4 (edge / empty window); mv, mpos only when >= 0.
template <bool DC, int NT>
__device__ __forceinline__ int fs_window(const float *x, const float *corr, uint32_t mask, uint32_t pos, int K, int L, int N,
                                         float match_sum, int tid, int lane, int wave, float *s_rf, int *s_ri,
                                         float &mv, uint32_t &mpos) {
    float mu = 0.f;
    if (DC) {
        float s = 0.f;
        for (int t = tid; t < K + L; t += NT) {
            const int64_t p = (int64_t)pos - (K + L - 1) + t;
            if (p >= 0) s += x[(uint32_t)p & mask];
        }
        s = wave_sum(s);
        if (lane == 0) s_rf[wave] = s;
        __syncthreads();
        s = 0.f;
        for (int w = 0; w < (NT / WAVE); w++) s += s_rf[w];
        __syncthreads();
        mu = s / (float)N;
    }
    const float off = mu * match_sum;
    float best = 0.f; int bidx = -1;
    {
        float cv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = tid + u * NT;
            const int64_t p = (int64_t)pos - K + t;
            cv[u] = (t <= K && p >= 0) ? corr[(uint32_t)p & mask] : 0.f;
            if (DC) cv[u] = (t <= K) ? cv[u] - off : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float c2 = cv[u] * cv[u];
            if (c2 > best) { best = c2; bidx = tid + u * NT; }
        }
        for (int t = tid + 8 * NT; t <= K; t += NT) {      // K > 8191 only
            const int64_t p = (int64_t)pos - K + t;
            float c = (p >= 0) ? corr[(uint32_t)p & mask] : 0.f;
            if (DC) c -= off;
            if (c * c > best) { best = c * c; bidx = t; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi >= 0 && (bidx < 0 || oi < bidx))) { best = ob; bidx = oi; }
    }
    if (lane == 0) { s_rf[wave] = best; s_ri[wave] = bidx; }
    __syncthreads();
    best = 0.f; bidx = -1;
    for (int w = 0; w < (NT / WAVE); w++) {
        const float ob = s_rf[w]; const int oi = s_ri[w];
        if (ob > best || (ob == best && oi >= 0 && (bidx < 0 || oi < bidx))) { best = ob; bidx = oi; }
    }
    __syncthreads();
    if (bidx < 0) { mpos = pos - (uint32_t)(K + L); mv = 0.f; return -5; }       // nothing above zero: the reference's mp = -1 (see k_sync_window_fft)
    if (bidx == 0 || bidx == K) return -4;                         // edge value -> -4 (mv stays 0)
    mpos = pos - (uint32_t)K + (uint32_t)bidx;
    float e = 0.f;
    for (int t = tid; t < L; t += NT) {
        const int64_t p = (int64_t)mpos - t;
        float v = (p >= 0) ? x[(uint32_t)p & mask] : 0.f;
        if (DC) v -= mu;
        e = fmaf(v, v, e);
    }
    e = wave_sum(e);
    if (lane == 0) s_rf[wave] = e;
    __syncthreads();
    e = 0.f;
    for (int w = 0; w < (NT / WAVE); w++) e += s_rf[w];
    __syncthreads();
    float c = corr[mpos & mask];
    if (DC) c -= off;
    mv = c / sqrtf(e);
    return bidx;
}

// One workgroup of 16 waves per channel.  The state machine is evaluated redundantly by every thread (all
// decisions depend only on workgroup-uniform values); the data-parallel parts — window arg-max (K+1 candidates),
// L-sample energy, header bit check, the nbits soft bits, RS syndromes — are spread over the 1024 threads so that
// each phase costs about one memory round trip instead of a chain of them.
template <bool DC, int NT>
__device__ __forceinline__ void framesync_body(const SyncArgs &a, const int ch) {
    __shared__ uint8_t s_frame[520];
    __shared__ uint8_t s_exp[512];
    __shared__ uint8_t s_log[256];
    __shared__ float s_rf[(NT / WAVE)];
    __shared__ int s_ri[(NT / WAVE)];
    __shared__ int s_cnt[2];
    __shared__ unsigned s_slot;
    __shared__ uint8_t s_syn[(NT / WAVE)][48];
    __shared__ uint8_t s_S[48];                // first-pass syndromes of the frame in hand
    __shared__ double s_rd[(NT / WAVE)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ch >= a.n_ch) return;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const float *bufs = a.bufs + (size_t)ch * a.ring_len;
    const float *corr = a.corr + (size_t)ch * a.ring_len;
    SyncState st = a.state[ch];
    const uint32_t ep = a.epoch ? a.epoch[ch] : 0u;      // this channel's stream start (0 unless the channel was restarted)
    uint32_t avail = a.avail;                  // IF samples [0, avail) exist (--dc: cut at an AFC event, the rest is recomputed)
    AfcState af{};                             // --dc only
    if (DC) af = a.afc[ch];
    bool afc_event = false;
    const int K = a.K, L = a.L;

    // profiling aid (SONDE_WF_PROF): thread 0 of channel 0 adds the shader-clock cycles since the previous mark to phase k
#define FS_MARK(k) do { if (a.prof && ch == 0 && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); a.prof[k] += t_ - t_prev; t_prev = t_; } } while (0)
    unsigned long long t_prev = a.prof ? __builtin_readcyclecounter() : 0ull;
    for (int i = tid; i < 512; i += NT) s_exp[i] = a.gf_exp[i];
    for (int i = tid; i < 256; i += NT) s_log[i] = a.gf_log[i];
    __syncthreads();
    FS_MARK(0);
    // pass 1 of two: the correlation ring is valid below this end position only (corr_tile_unused with the same state and limit)
    const bool horizon_on = a.corr_limit != 0 && (st.mode == 0 || st.mode == 1);
    const uint32_t horizon = horizon_on ? sync_first_pos(st, a.frame_samples, a.delay) + a.corr_limit : 0u;

    // every pass consumes a window (K-4 samples) or a frame, or ends at `avail`: the bound only guards against a corrupted state
    for (int guard = 0; guard < (1 << 20); guard++) {
        if (st.mode == 2) break;                       // stream finished
        if (st.mode == 0) {
            // ---- find_header: next correlation once K-4 new samples were consumed (demod_mod.c:1540-1548)
            const uint32_t need = (uint32_t)(K - 4) - st.k;
            const uint32_t s_in_w = st.s_in + need;
            if ((int32_t)(avail - s_in_w) < 0) { st.k += avail - st.s_in; st.s_in = avail; break; }
            if (horizon_on && (int32_t)(s_in_w - 1 - (uint32_t)a.delay - horizon) >= 0) break;      // this window is pass 2's
            const uint32_t pos = s_in_w - 1 - (uint32_t)a.delay;      // sample_out
            const WinItem *wi = nullptr;
            if (!DC && a.win && pos >= (uint32_t)L) {                  // precomputed with the reference's transform (k_sync_window_fft)
                for (int w = 0; w < a.win_W; w++) { const WinItem *c = a.win + (size_t)ch * a.win_W + w; if (c->state == 2 && c->pos == pos) { wi = c; break; } }
                if (!wi) break;                                        // planned windows used up: the next round continues here
            }
            st.s_in = s_in_w; st.k = 0; st.mv = 0.f;
            if (pos - ep < (uint32_t)L) continue;                      // getCorrDFT returns -2 (position counted from the channel's stream start)
            float mv; uint32_t mpos;
            if (wi) { if (wi->rc < 0) { if (wi->rc == -5) st.mv_pos = wi->mpos; continue; } mv = wi->mv; mpos = wi->mpos; }
            else {
                const int rc = fs_window<DC, NT>(bufs, corr, mask, pos, K, L, a.N, a.match_sum, tid, lane, wave, s_rf, s_ri, mv, mpos);
                if (rc == -5) { st.mv_pos = mpos; if (DC) af.dc = 0.0; }       // an all-zero window: position taken, nothing found (getCorrDFT with mp = -1)
                if (rc < 0) continue;
            }
            const uint32_t prev = st.mv_pos;
            st.mv = mv; st.mv_pos = mpos;
            if (DC) {
                // FM-stream fallback (opt_iq >= 2 and the tone stream missed, demod_mod.c:230-277), header dc (:280-292), dDf (:298)
                const float *fm = a.fm + (size_t)ch * a.ring_len;
                const float hofs = ((float)a.lpfm_taps - (a.sps - 1.0f)) / 2.0f;
                uint32_t dpos = mpos, mv2_pos = 0;
                if (a.opt_iq >= 2 && fabsf(mv) < a.thres) {
                    float mv2; uint32_t mpos2;
                    if (fs_window<true, NT>(fm, a.corr2 + (size_t)ch * a.ring_len, mask, pos, K, L, a.N, a.match_sum, tid, lane, wave, s_rf, s_ri, mv2, mpos2) < 0) continue;
                    mv2_pos = (uint32_t)((float)mpos2 - hofs);
                    dpos = mpos2;
                    if (mv2 > a.thres || mv2 < -a.thres) { st.mv = mv2; st.mv_pos = mv2_pos; }
                }
                const int mp_ofs = (a.opt_iq >= 2 && mv2_pos == 0) ? (int)hofs : 0;
                double dsum = 0.0;
                for (int t = tid; t < L; t += NT) dsum += (double)fm[((uint32_t)mp_ofs + dpos - (uint32_t)t) & mask];
                for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
                if (lane == 0) s_rd[wave] = dsum;
                __syncthreads();
                dsum = 0.0;
                for (int w = 0; w < (NT / WAVE); w++) dsum += s_rd[w];
                __syncthreads();
                af.dc = dsum / (double)(float)L;
                mv = st.mv; mpos = st.mv_pos;
            }
            if (!(mv > a.thres || mv < -a.thres)) continue;
            if (DC && a.opt_iq) {
                // AFC (find_header, demod_mod.c:1553-1600): 60 % of the remaining offset once it exceeds 100 Hz; IF filter
                // locked below 1 kHz.  Everything from sample_in = st.s_in on is recomputed by the host loop.
                const double dDf = (double)a.sr * af.dc / (2.0 * 0.8);
                bool ev = false;
                if (fabs(dDf) > 100.0) {
                    const double diff = dDf * 0.6;
                    const int nn = (int)a.sps;
                    if (a.opt_iq >= 2 && tid < nn) {               // rot_iqbuf tail: the tone window straddling the change (:1562-1575)
                        const uint32_t m = st.s_in - (uint32_t)(tid + 1);
                        const double tn = (double)m / (double)a.sr;
                        double sn, cs;
                        sincos(-tn * 6.2831853071795864769 * diff, &sn, &cs);
                        float2 *zp = a.ifiq + (size_t)ch * a.ring_len + (m & mask);
                        const float2 z = *zp;
                        *zp = make_float2((float)((double)z.x * cs - (double)z.y * sn), (float)((double)z.x * sn + (double)z.y * cs));
                    }
                    af.Df += diff; ev = true;
                }
                if (fabs(dDf) > 1e3) { if (af.locked) { af.locked = 0; ev |= (a.lpiq_on != 0); } }
                else { if (!af.locked) { af.locked = 1; ev |= (a.lpiq_on != 0); } }
                if (ev) { afc_event = true; avail = st.s_in; }
            }
            if (!(mpos - ep > prev - ep)) continue;                    // positions as the reference counts them: from the channel's stream start (an all-zero first window leaves a wrapped one)
            FS_MARK(1);
            // ---- headcmp (demod_mod.c:870-938): hard-slice the header from the ring, count mismatches
            int errs = 0;
            const int nsym = a.hdrlen / a.symhd;
            const uint32_t mvp = mpos + 1 - (uint32_t)L;
            const double hdc = (DC && a.opt_iq < 2) ? af.dc : 0.0;     // read_bufbit: bufs - dc for the FM-sliced forms (demod_mod.c:879)
            for (int p = tid; p < nsym; p += NT) {
                double edge = (double)((float)(p * a.symhd) * a.sps);
                uint32_t cnt = (uint32_t)ceil(edge);
                double sum = 0.0;
                edge += (double)a.sps;
                // the reference's ring holds M = N samples (bufs[sample_in % M], demod_mod.c:850): a header position taken from the FM-stream
                // fallback lies up to (lpFMtaps - sps + 1) / 2 samples before the window (:268), and what it reads there has already been
                // overwritten by the newest samples — index + M
                const uint32_t ring_first = st.s_in - (uint32_t)a.N;
                do { uint32_t ix = cnt + mvp; if ((int32_t)(ix - ring_first) < 0) ix += (uint32_t)a.N; sum += (double)bufs[ix & mask] - hdc; cnt++; } while ((double)cnt < edge);
                if (a.symhd == 2) {
                    edge += (double)a.sps;
                    do { uint32_t ix = cnt + mvp; if ((int32_t)(ix - ring_first) < 0) ix += (uint32_t)a.N; sum -= (double)bufs[ix & mask] - hdc; cnt++; } while ((double)cnt < edge);
                }
                const int sign = mv < 0 ? 1 : 0;
                if (a.symhd == 1) {
                    const int bit = (sum >= 0) ? 1 : 0;
                    errs += ((bit ^ sign) != (a.hdr[p] & 1));
                } else {
                    const int b0 = (sum >= 0) ? 1 : 0, b1 = 1 - b0;
                    errs += ((b0 ^ sign) != (a.hdr[2 * p] & 1)) + ((b1 ^ sign) != (a.hdr[2 * p + 1] & 1));
                }
            }
            for (int off = 32; off > 0; off >>= 1) errs += __shfl_xor(errs, off);
            if (lane == 0) s_ri[wave] = errs;
            __syncthreads();
            errs = 0;
            for (int w = 0; w < (NT / WAVE); w++) errs += s_ri[w];
            __syncthreads();
            FS_MARK(2);
            if (errs > a.hdmax) continue;
            if (mv * (0.5f - (float)st.inv) < 0.f) {    // polarity mismatch (rs41mod.c:2887-2891): skipped, or flips the channel with --auto
                if (!a.opt_auto) continue;
                st.inv ^= 1u;
            }
            st.mode = 1;
        } else {
            // ---- frame: nbits soft bits from sample mv_pos+1+ofs on (read_softbit2p, demod_mod.c:1087-1175)
            const uint32_t s_in_after = st.mv_pos + (uint32_t)a.delay + 1 + a.frame_samples;
            const bool enough = (int32_t)(avail - s_in_after) >= 0;
            if (!enough && !(a.eof && (a.eof_ch < 0 || a.eof_ch == ch))) break;      // wait for the next chunk
            // at end of stream the reference slices until f32buf_sample() hits EOF (rs41mod.c:2931): consumption q
            // needs IF sample mv_pos+delay+1+q, so only bits ending at q1 <= q_lim exist
            const int32_t q_lim = enough ? (int32_t)a.frame_samples : (int32_t)(avail - (st.mv_pos + (uint32_t)a.delay + 1));
            const uint32_t base = st.mv_pos + 1 + (uint32_t)a.bitofs;
            if (tid == 0) { s_slot = atomicAdd(a.frame_count, 1u) % (unsigned)a.max_frames; s_cnt[0] = 0; s_cnt[1] = 0; }
            for (int i = tid; i < 520; i += NT) s_frame[i] = (a.rs41 && i < 8) ? a.hdr_bytes[i] : 0;
            __syncthreads();
            const unsigned slot = s_slot;                              // monotonic counter, ring of records
            FrameRec *rec = a.frames + slot;
            FS_MARK(3);
            for (int p0 = 0; p0 < a.nbits; p0 += NT) {
                const int bp = p0 + tid;
                double sum = 0.0;
                bool valid = bp < a.nbits;
                if (valid) {
                    const uint4 w = a.bitwin[bp];                      // {qa-, qb-, qa+, qb+}
                    valid = (int32_t)a.bitend[bp] <= q_lim;
                    if (valid) {
                        if (DC && a.opt_iq < 2) {
                            if (w.y > w.x) sum = 0.0 - window_sum<true>(bufs, base, mask, w.x, w.y, af.dc);
                            sum += window_sum<true>(bufs, base, mask, w.z, w.w, af.dc);
                        } else {
                            if (w.y > w.x) sum = 0.0 - window_sum<false>(bufs, base, mask, w.x, w.y, 0.0);
                            sum += window_sum<false>(bufs, base, mask, w.z, w.w, 0.0);
                        }
                    }
                }
                if (st.inv) sum = -sum;                                // -i / --auto: bit ^= 1, sb = -sb (rs41mod.c:2933-2937)
                const int hb = valid && (st.inv ? !(-sum >= 0.0) : (sum >= 0.0));
                const unsigned long long bal = __ballot(hb), vm = __ballot(valid);
                if (a.soft && valid) a.soft[(size_t)slot * a.nbits + bp] = (float)sum;
                if (a.soft1 && valid) {                                // sample1 = bufs[.. + ofs - 1]: same windows, one sample earlier
                    const uint4 w = a.bitwin[bp];
                    const double d1 = (DC && a.opt_iq < 2) ? af.dc : 0.0;
                    double s1 = 0.0;
                    if (DC && a.opt_iq < 2) {
                        if (w.y > w.x) s1 = 0.0 - window_sum<true>(bufs, base - 1u, mask, w.x, w.y, d1);
                        s1 += window_sum<true>(bufs, base - 1u, mask, w.z, w.w, d1);
                    } else {
                        if (w.y > w.x) s1 = 0.0 - window_sum<false>(bufs, base - 1u, mask, w.x, w.y, 0.0);
                        s1 += window_sum<false>(bufs, base - 1u, mask, w.z, w.w, 0.0);
                    }
                    a.soft1[(size_t)slot * a.nbits + bp] = (float)(st.inv ? -s1 : s1);
                }
                const int it = (p0 >> 6) + wave;                       // 64-bit group index = 8 frame bytes
                if (lane < 8) {
                    if (a.rs41) {
                        const int bi = 8 + it * 8 + lane;              // frame byte index (LSB-first bits, rs41mod.c:224)
                        if (bi < 518 && ((vm >> (8 * lane + 7)) & 1ULL))
                            s_frame[bi] = (uint8_t)((bal >> (8 * lane)) & 0xff) ^ a.mask[bi & 63];
                    } else {
                        const int bi = it * 8 + lane;                  // other sondes: hard bits packed LSB-first, framed on the host
                        if (bi < 520) s_frame[bi] = (uint8_t)(((bal & vm) >> (8 * lane)) & 0xff);
                    }
                }
                if (lane == 0) { atomicAdd(&s_cnt[0], __popcll(vm & 0x8080808080808080ULL)); atomicAdd(&s_cnt[1], __popcll(vm)); }
            }
            __syncthreads();
            FS_MARK(4);
            const int nbytes_ok = s_cnt[0], nbits_ok = s_cnt[1];
            // frame length from the type byte (rs41mod.c:407-415,2488-2490)
            int ft = 0; { const uint8_t b = s_frame[0x38]; for (int q = 0; q < 4; q++) ft += ((b >> q) & 1) - ((b >> (q + 4)) & 1); }
            const int flen = (ft >= 0) ? 320 : 518;
            // RS(255,231) syndromes S_j = cw(alpha^j), j = 0..23, two interleaved codewords (rs41mod.c:1729-1732): wave c evaluates coefficients
            // 16c..16c+15 by Horner and scales by alpha^(16 c j); XOR over the waves.  Bytes from flen on count as zero (:1727).
            // Whole frames of an engine with --ecc / --ecc2: a clean frame is final here (tail zeroed like rs41_ecc leaves it); a damaged one goes on
            // the work list of k_rs41_ecc_frames (Euclid / Chien / Forney on a wavefront per codeword, 2nd pass; sonde_rs_dev.h), which runs on its
            // own stream beside the next call's decimator.  Without a list (end-of-stream launches, a frame cut short whose missing bytes the host
            // fills from the previous frame, rs41mod.c:2479-2490) the record carries the syndromes and the host decodes it when it is fetched.
            int ecc_done = 0;
            if (a.rs41) {
                if (lane < 48) {
                    const int cw = lane / 24, jx = lane % 24;
                    const uint8_t x = s_exp[jx];
                    uint8_t hsum = 0;
                    constexpr int CH = 256 / (NT / WAVE);                  // coefficients per wave
                    for (int i = CH - 1; i >= 0; i--) {
                        const int n = CH * wave + i;
                        uint8_t v = 0;
                        if (n < 255) {
                            const int fi = (n >= 24) ? 56 + 2 * (n - 24) + cw : 8 + 24 * cw + n;
                            v = (fi < flen) ? s_frame[fi] : 0;
                        }
                        const uint8_t prod = (hsum && x) ? s_exp[s_log[hsum] + s_log[x]] : 0;
                        hsum = prod ^ v;
                    }
                    const int sh = (jx * CH * wave) % 255;                 // alpha^(j * CH * c)
                    s_syn[wave][lane] = hsum ? s_exp[(s_log[hsum] + sh) % 255] : 0;
                }
                __syncthreads();
                if (tid < 48) { uint8_t syn = 0; for (int w = 0; w < (NT / WAVE); w++) syn ^= s_syn[w][tid]; s_S[tid] = syn; }
                __syncthreads();
                if (a.ecc_level > 0 && 8 + nbytes_ok >= 518) {
                    bool clean = true;
                    for (int k = 0; k < 48; k++) clean &= (s_S[k] == 0);
                    if (clean) { ecc_done = 1; for (int i = flen + tid; i < 518; i += NT) s_frame[i] = 0; }
                    else if (a.ecc_list) { ecc_done = 2; if (tid == 0) a.ecc_list[atomicAdd(a.ecc_count, 1u) % (unsigned)a.max_frames] = slot; }
                }
            }
            __syncthreads();
            for (int i = tid; i < 518; i += NT) rec->frame[i] = s_frame[i];
            if (a.rs41 && tid < 48) rec->synd[tid] = s_S[tid];
            if (tid == 0) {
                rec->channel = ch; rec->mv = st.mv; rec->mv_pos = st.mv_pos; rec->len = a.rs41 ? flen : a.nbits; rec->nbytes = a.rs41 ? 8 + nbytes_ok : nbits_ok;
                rec->ecc = 0; rec->ecc_done = ecc_done;
            }
            if (a.summary && tid == 0) {                              // per-channel detection summary, stays on the device
                bool clean = a.rs41 != 0;
                if (a.rs41) for (int k = 0; k < 48; k++) clean &= (s_S[k] == 0);
                const uint32_t gch = a.summary_map ? (uint32_t)a.summary_map[ch] : (uint32_t)ch;      // mixed engines: the caller's channel number
                sonde_summary_t *sm = a.summary + gch;
                sm->channel_id = a.summary_base + gch; sm->type = (uint8_t)a.summary_type; sm->inverted = (uint8_t)(st.mv < 0.f);
                sm->score = st.mv; sm->freq_offset_hz = DC ? (float)af.Df : 0.f;
                sm->sample_pos = a.summary_epoch - (uint64_t)(uint32_t)((uint32_t)a.summary_epoch - st.mv_pos);     // mv_pos is the low half of a 64-bit index
                sm->frames += 1; sm->frames_clean += clean ? 1u : 0u;
            }
            __syncthreads();
            FS_MARK(5);
            if (a.prof && ch == 0 && tid == 0) a.prof[14] += 1;
            if (!enough) { st.mode = 2; st.s_in = avail; break; }
            st.s_in = s_in_after; st.k = 0; st.mode = 0;
        }
    }
    if (tid == 0) a.state[ch] = st;
    FS_MARK(6);
    if (a.prof && ch == 0 && tid == 0) a.prof[15] += 1;
    if (DC && tid == 0) {
        a.afc[ch] = af;
        a.start[ch] = afc_event ? avail : a.avail;      // first IF sample the host loop has to recompute for this channel
        if (afc_event) atomicAdd(a.pending, 1u);
    }
}

template <bool DC, int NT>
__global__ __launch_bounds__(NT, (NT == 1024 ? 4 : 3))      // small form: at most 168 registers, a wave per SIMD
void k_framesync(const SyncArgs a) { framesync_body<DC, NT>(a, (int)blockIdx.x); }
// mixed engines: one workgroup per row of the engine, the row's group brings its preset (header, bit clock, frame length, thresholds, record queue)
__global__ __launch_bounds__(1024, 4)
void k_framesync_multi(const MultiArgs<SyncArgs> m) {
    const int g = multi_group(m, (int)blockIdx.x);
    framesync_body<false, 1024>(m.g[g], (int)blockIdx.x - m.row0[g]);
}

// ------------------------------------------------------------------------------------------------
// k_sync_plan / k_sync_window_fft: the header search of find_header with the reference's transform (WinItem in sonde_dev.h)
// ------------------------------------------------------------------------------------------------
// the windows the frame sync will ask for, in order, assuming none of them finds a header (those behind a hit are not consumed)
__device__ __forceinline__ void sync_plan_body(const WinPlanArgs &a, const int ch, const bool first_thread) {
    if (ch >= a.n_ch) return;
    const SyncState st = a.state[ch];
    WinItem *it = a.items + (size_t)ch * a.stride;
    uint32_t s_in = st.s_in, k = st.k;
    const uint32_t ep = a.epoch ? a.epoch[ch] : 0u;
    bool on = st.mode == 0;
    if (st.mode == 1) {                                       // frame in progress: the search resumes behind it, if it ends in this call
        const uint32_t s_after = st.mv_pos + (uint32_t)a.delay + 1 + a.frame_samples;
        if ((int32_t)(a.avail - s_after) >= 0) { s_in = s_after; k = 0; on = true; }
    }
    int n = 0;
    while (on && n < a.W) {
        const uint32_t s_in_w = s_in + ((uint32_t)(a.K - 4) - k);
        if ((int32_t)(a.avail - s_in_w) < 0) break;
        const uint32_t pos = s_in_w - 1 - (uint32_t)a.delay;
        if (pos - ep >= (uint32_t)a.L) { it[n].pos = pos; it[n].state = 1; it[n].rc = -1; it[n].mv = 0.f; it[n].mpos = 0; n++; }     // else getCorrDFT returns -2: nothing to evaluate
        s_in = s_in_w; k = 0;
    }
    if (a.work && n > 0) {                                      // compact list of the windows to evaluate this round (no empty workgroups)
        const uint32_t at = atomicAdd(a.work_count + a.round_parity, (uint32_t)n);
        for (int i = 0; i < n; i++) a.work[at + (uint32_t)i] = (uint32_t)(ch * a.stride + i);
    }
    if (a.work && first_thread) a.work_count[a.round_parity ^ 1] = 0;       // the other round's counter: its consumer ran before this kernel
    for (; n < a.stride; n++) { it[n].pos = 0xffffffffu; it[n].state = 0; }
}
__global__ void k_sync_plan(const WinPlanArgs a) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    sync_plan_body(a, ch, ch == 0);
}
__global__ void k_sync_plan_multi(const MultiArgs<WinPlanArgs> m) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m.row0[m.n_groups]) return;
    int g = 0;
    while (g + 1 < m.n_groups && row >= m.row0[g + 1]) g++;
    sync_plan_body(m.g[g], row - m.row0[g], row == m.row0[g]);
}

// The reference is plain C on x86-64: separately rounded multiplies and adds (no fused multiply-add) in the butterflies and products.
#pragma clang fp contract(off)
// threads per window workgroup: two workgroups of 512 share a CU (74 KB of LDS each, up to 128 registers per thread) — while one waits at a
// barrier or for memory the other one computes; a single 1024-thread workgroup per CU left the CU idle in every such wait
#ifndef WF_THREADS
#define WF_THREADS 512
#endif
#define FFT_THREADS WF_THREADS
#include "sonde_fft_dev.h"
// one planned window (getCorrDFT without --dc, demod_mod.c:148-225), evaluated by one workgroup
#ifndef WF_MB
#define WF_MB 4               // table entries fetched together in the conjugate-and-swap pass
#endif
#ifndef WF_LU
#define WF_LU 1
#endif
#ifndef WF_LB
#define WF_LB 4               // window samples a thread fetches together (8: no faster, more spills)
#endif
// profiling aid (SONDE_WF_PROF): thread 0 of workgroup 0 adds the shader-clock cycles since the previous mark to phase k
#define WF_MARK(k) do { if (a.prof && blockIdx.x == 0 && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); a.prof[k] += t_ - t_prev; t_prev = t_; } } while (0)
__device__ __forceinline__ void sync_eval_window(const WinFftArgs &a, const int ch, WinItem *it, float2 *x, float2 *tws, float *s_rf, int *s_ri) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (it->state != 1) return;
    unsigned long long t_prev = a.prof ? __builtin_readcyclecounter() : 0ull;
    const int K = a.K, L = a.L, N = SC_N, wl = K + L;
    const uint32_t pos = it->pos, mask = (uint32_t)a.ring_len - 1;
    const float *bufs = a.bufs + (size_t)ch * a.ring_len;
    const int64_t start = (int64_t)pos - (wl - 1);
    // xn[i] = bufs[pos - (K+L-1) + i], i < K+L, zero padded (:168-169); bit-reversed for the DIT network, natural order for the norm
    // (all loads of a thread are issued before the first store: one memory round trip for the window instead of one per element)
    {
        constexpr int NL = SC_N / WF_THREADS;
#pragma unroll 1
        for (int h = 0; h < NL; h += WF_LB) {
            float v[WF_LB];
#pragma unroll
            for (int u = 0; u < WF_LB; u++) {
                const int i = tid + (h + u) * WF_THREADS;
                const int64_t p = start + i;
                v[u] = (i < wl && p >= 0) ? bufs[(uint32_t)p & mask] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < WF_LB; u++) x[XI(brev13(tid + (h + u) * WF_THREADS))] = make_float2(v[u], 0.f);
        }
    }
    __syncthreads();
    WF_MARK(0);
    dft_ref(x, tws, a.tws, tid);                                         // X = rdft(xn)
    WF_MARK(1);
    // Z = X * Fm (:190); Nidft() transforms conj(Z) (:78-80): conjugate and swap into bit-reversed order for the same network.
    // Pairs (i, r = brev(i)), r >= i; Fm[r] comes from the bit-reversed copy of the table behind it (a.Fm + N) — coalesced like Fm[i]
    {
        const float2 *FmR = a.Fm + N;
#pragma unroll 1
        for (int h = 0; h < N / WF_THREADS; h += WF_MB) {
            float2 fi[WF_MB], fr[WF_MB];
#pragma unroll
            for (int u = 0; u < WF_MB; u++) {
                const int i = tid + (h + u) * WF_THREADS;
                const bool on = brev13(i) >= i;
                fi[u] = on ? a.Fm[i] : make_float2(0.f, 0.f);
                fr[u] = on ? FmR[i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < WF_MB; u++) {
                const int i = tid + (h + u) * WF_THREADS, r = brev13(i);
                if (r < i) continue;
                const float2 zi = cmul(x[XI(i)], fi[u]), zr = cmul(x[XI(r)], fr[u]);
                x[XI(r)] = make_float2(zi.x, -zi.y);
                x[XI(i)] = make_float2(zr.x, -zr.y);
            }
        }
    }
    __syncthreads();
    WF_MARK(2);
    dft_ref_head(x, tws, a.tws, tid);                                    // cx = Nidft(Z), real part used
    // last stage (dit_pass<1> at t = 12) in registers: only re(cx) is looked at, so nothing is stored — the arg-max of re(cx)^2 over
    // i in [L-1, K+L), first maximum wins (:200-207), is taken from the butterfly outputs as they come
    float best = 0.f, bestc = 0.f; int bidx = -1;
#pragma unroll WF_LU
    for (int u = 0; u < SC_N / 2 / WF_THREADS; u++) {
        const int g = tid + u * WF_THREADS;
        const float2 w = a.tws[((1 << 12) - 1) + g];
        const float2 p = x[XI(g)], r = cmul(x[XI(g + (1 << 12))], w);
        const float c0 = p.x + r.x, c1 = p.x - r.x;
        const int i0 = g, i1 = g + (1 << 12);
        if (i0 >= L - 1 && i0 < wl) { const float c2 = c0 * c0; if (c2 > best || (c2 == best && bidx >= 0 && i0 < bidx)) { best = c2; bidx = i0; bestc = c0; } }
        if (i1 >= L - 1 && i1 < wl) { const float c2 = c1 * c1; if (c2 > best || (c2 == best && bidx >= 0 && i1 < bidx)) { best = c2; bidx = i1; bestc = c1; } }
    }
    WF_MARK(3);
    {
        float rb = best; int ri = bidx;
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(rb, off); const int oi = __shfl_xor(ri, off);
            if (ob > rb || (ob == rb && oi >= 0 && (ri < 0 || oi < ri))) { rb = ob; ri = oi; }
        }
        if (lane == 0) { s_rf[wave] = rb; s_ri[wave] = ri; }
    }
    __syncthreads();
    int mp = -1;
    {
        float b = 0.f;
        for (int w = 0; w < WF_THREADS / WAVE; w++) {
            const float ob = s_rf[w]; const int oi = s_ri[w];
            if (ob > b || (ob == b && oi >= 0 && (mp < 0 || oi < mp))) { b = ob; mp = oi; }
        }
    }
    __syncthreads();
    WF_MARK(4);
    if (mp < 0) {
        // not one correlation value above zero — a stream that begins with digital silence.  The reference's loop leaves mp = -1 (:200-207), which is no edge value:
        // getCorrDFT runs on and sets mv = 0 / (a norm read in front of the array) and mv_pos = pos - (K + L - 1) - 1, which wraps in the first window of a stream —
        // and `mv_pos > mvpos0` (find_header, :1603) then fails for the header the NEXT window finds.  Handed on as rc = -5 (k_framesync takes the position)
        if (tid == 0) { it->rc = -5; it->mv = 0.f; it->mpos = pos - (uint32_t)(wl - 1) - 1u; __threadfence(); it->state = 2; }
        return;
    }
    if (mp == L - 1 || mp == wl - 1) {                                    // edge value: -4 (:208)
        if (tid == 0) { it->rc = -4; it->mv = 0.f; it->mpos = 0; __threadfence(); it->state = 2; }
        return;
    }
    // xnorm = sqrt(sum_{i<L} xn[mp-i]^2) (:215-217); mx /= xnorm * N
    float e = 0.f;
    for (int k = tid; k < L; k += WF_THREADS) {                          // xn[mp - k], read again from the ring (it is not kept in LDS)
        const int i = mp - k; const int64_t p2 = start + i;
        const float v = (i < wl && p2 >= 0) ? bufs[(uint32_t)p2 & mask] : 0.f;
        e += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if (lane == 0) s_rf[wave] = e;
    if (bidx == mp) s_rf[WF_THREADS / WAVE] = bestc;                     // re(cx[mp]) from the thread whose butterfly produced it
    __syncthreads();
    if (tid == 0) {
        float es = 0.f;
        for (int w = 0; w < WF_THREADS / WAVE; w++) es += s_rf[w];
        const float xnorm = sqrtf(es);
        it->rc = mp; it->mv = s_rf[WF_THREADS / WAVE] / (xnorm * (float)N); it->mpos = pos - (uint32_t)(wl - 1) + (uint32_t)mp;
        __threadfence(); it->state = 2;
    }
    WF_MARK(5);
    if (a.prof && blockIdx.x == 0 && tid == 0) a.prof[15] += 1;
}

// Workgroups walk the compact list k_sync_plan wrote (a.work): the grid does not depend on how many windows a round planned, and a round in
// which almost every channel is inside a frame costs a handful of workgroups instead of stride x channels empty ones.
// (the scanner's k_scan_corr runs two such workgroups per CU at 64 VGPRs; here the ~1000 windows of a round gain nothing from that and the
// spills cost 10 %, measured: one workgroup per CU with the registers the compiler wants)
#if WF_THREADS <= 512
#define WF_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2 * WF_THREADS / 256, 2 * WF_THREADS / 256)))      // two workgroups per CU
#else
#define WF_WAVES_ATTR
#endif
__device__ __forceinline__ void sync_window_fft_body(const WinFftArgs &a, const uint32_t b, const uint32_t nb) {      // workgroup b of the nb that share a's list
    extern __shared__ __attribute__((aligned(16))) float2 smem2[];
    float2 *x = smem2;                           // [SC_XN] padded (XI)
    float2 *tws = smem2 + SC_XN;       // [SC_TW_LDS + 1] twiddles of stages 0..8
    __shared__ float s_rf[WF_THREADS / WAVE + 1];
    __shared__ int s_ri[WF_THREADS / WAVE];
    const uint32_t count = a.work_count[a.round_parity];
    if (b >= count) return;
    for (int k = threadIdx.x; k < SC_TW_LDS; k += WF_THREADS) tws[k] = a.tws[k];
    for (uint32_t w = b; w < count; w += nb) {
        const uint32_t item = a.work[w];
        __syncthreads();                         // the previous window's last reads of x / s_rf are over
        sync_eval_window(a, (int)(item / (uint32_t)a.stride), a.items + item, x, tws, s_rf, s_ri);
    }
}
__global__ __launch_bounds__(WF_THREADS) WF_WAVES_ATTR
void k_sync_window_fft(const WinFftArgs a) { sync_window_fft_body(a, blockIdx.x, gridDim.x); }
// mixed engines: row0[] counts WORKGROUPS here — group g's windows (its own template spectrum, K, L, list) are walked by workgroups row0[g] .. row0[g + 1] - 1
__global__ __launch_bounds__(WF_THREADS) WF_WAVES_ATTR
void k_sync_window_fft_multi(const MultiArgs<WinFftArgs> m) {
    const int g = multi_group(m, (int)blockIdx.x);
    sync_window_fft_body(m.g[g], blockIdx.x - (uint32_t)m.row0[g], (uint32_t)(m.row0[g + 1] - m.row0[g]));
}
// ---- the same window transform in HALF the LDS (round 4): a decimation-in-time network on bit-reversed input never mixes the two halves of its
// array before the last stage (stage s pairs i with i + 2^s inside blocks of 2^(s+1) <= 4096 for s <= 11).  So the half [0, 4096) — the EVEN window
// samples — goes through stages 0..11 in a 4096-point array, its results wait in registers (8 per thread), the half [4096, 8192) — the odd samples —
// follows in the same array, and the last stage combines registers with LDS.  35 KB + 4 KB of twiddles instead of 74 KB; with 256 threads (sixteen butterflies of the last
// stage each, <= 168 registers) the workgroup fits the slot ONE decimator workgroup frees (51 KB, a wave per SIMD), which is what lets the header
// search run beside the next call's decimator instead of waiting for it to drain.  Every butterfly is the same cmul / add / sub on
// the same operands with the same twiddle as in dft_ref: scores and positions identical to the bit.
#define WFH_THREADS 256               // four waves, one per SIMD at <= 168 registers: exactly the slot of one decimator workgroup
#define SCH_N (SC_N / 2)
#define SCH_XN (SCH_N + SCH_N / 16 + SCH_N / 256)
template <int R>
__device__ __forceinline__ void dit_pass_h(float2 *x, const float2 *tws, const int t0, const int tid) {
    constexpr int E = 1 << R;
    const int p_lo = t0;
#pragma unroll 1
    for (int g = tid; g < (SCH_N >> R); g += WFH_THREADS) {
        const int low = g & ((1 << p_lo) - 1), high = g >> p_lo;
        const int base = (high << (p_lo + R)) | low;
        float2 v[E];
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = x[XI(base + (e << p_lo))];
#pragma unroll
        for (int s2 = 0; s2 < R; s2++) {
            const int t = t0 + s2, bit = 1 << s2;
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (e & bit) continue;
                const int idx = base + (e << p_lo);
                const float2 w = tws[((1 << t) - 1) + (idx & ((1 << t) - 1))];
                const float2 p = v[e], r = cmul(v[e | bit], w);
                v[e] = make_float2(p.x + r.x, p.y + r.y);
                v[e | bit] = make_float2(p.x - r.x, p.y - r.y);
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) x[XI(base + (e << p_lo))] = v[e];
    }
    __syncthreads();
}
__device__ __forceinline__ void dft_half(float2 *x, const float2 *tws, const float2 *tws_g, const int tid) {      // stages 0..11 of one half
    dit_pass_h<3>(x, tws, 0, tid);
    dit_pass_h<3>(x, tws, 3, tid);
    dit_pass_h<3>(x, tws, 6, tid);
    dit_pass_h<3>(x, tws_g, 9, tid);
}
__device__ __forceinline__ int brev12(int k) { return (int)(__brev((unsigned)k) >> 20); }

// the parking array is written by one thread and read by another of the same workgroup (and re-used window after window): device-scope accesses,
// so that a read never hits a line an earlier window left in the CU's L1
__device__ __forceinline__ float2 park_ld(const float2 *p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)(v & 0xffffffffull)), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void park_st(float2 *p, const float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_eval_window_h(const WinFftArgs &a, const int ch, WinItem *it, float2 *x, float2 *tws, float *s_rf, int *s_ri, float2 *park) {
    constexpr int NU = SCH_N / WFH_THREADS;      // 16: butterflies of the last stage per thread, g = tid + WFH_THREADS u
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (it->state != 1) return;
    const int K = a.K, L = a.L, N = SC_N, wl = K + L;
    const uint32_t pos = it->pos, mask = (uint32_t)a.ring_len - 1;
    const float *bufs = a.bufs + (size_t)ch * a.ring_len;
    const int64_t start = (int64_t)pos - (wl - 1);
    // xn[i] = bufs[pos - (K+L-1) + i], i < K+L, zero padded (:168-169).  Slot brev12(k) of the even half holds xn[2k], of the odd half xn[2k+1].
    for (int par = 0; par < 2; par++) {
        float v[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int i = 2 * (tid + WFH_THREADS * u) + par;
            const int64_t p = start + i;
            v[u] = (i < wl && p >= 0) ? bufs[(uint32_t)p & mask] : 0.f;
        }
        if (par == 1) {                                                  // the even half's results wait in the parking array while the odd half uses the LDS
#pragma unroll 4
            for (int u = 0; u < NU; u++) park_st(park + tid + WFH_THREADS * u, x[XI(tid + WFH_THREADS * u)]);
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < NU; u++) x[XI(brev12(tid + WFH_THREADS * u))] = make_float2(v[u], 0.f);
        __syncthreads();
        dft_half(x, tws, a.tws, tid);
    }
    // last stage of X = rdft(xn), then Z = X * Fm (:190) and conj(Z) as the natural-order input of Nidft's transform (:78-80), parked in natural order:
    // park[n] = conj(Z[n]); element n goes to slot brev13(n) of that transform: even n to its even half (slot brev12(n / 2)), odd n to its odd half
#pragma unroll 2
    for (int u = 0; u < NU; u++) {
        const int g = tid + WFH_THREADS * u;
        const float2 w = a.tws[((1 << 12) - 1) + g];
        const float2 p = park_ld(park + g), r = cmul(x[XI(g)], w);
        const float2 x0 = make_float2(p.x + r.x, p.y + r.y), x1 = make_float2(p.x - r.x, p.y - r.y);
        const float2 q0 = cmul(x0, a.Fm[g]), q1 = cmul(x1, a.Fm[g + SCH_N]);
        park_st(park + g, make_float2(q0.x, -q0.y)); park_st(park + g + SCH_N, make_float2(q1.x, -q1.y));       // (g is read and written by this thread only)
    }
    __syncthreads();
    for (int par = 0; par < 2; par++) {
        if (par == 1) {
            // the even half's results: into the (now free) even-n places of the parking array
            float2 t[4];
#pragma unroll 1
            for (int u0 = 0; u0 < NU; u0 += 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) t[q] = x[XI(tid + WFH_THREADS * (u0 + q))];
#pragma unroll
                for (int q = 0; q < 4; q++) park_st(park + 2 * (tid + WFH_THREADS * (u0 + q)), t[q]);
            }
            __syncthreads();
        }
        float2 v2[4];
#pragma unroll 1
        for (int u0 = 0; u0 < NU; u0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) v2[q] = park_ld(park + 2 * (tid + WFH_THREADS * (u0 + q)) + par);
#pragma unroll
            for (int q = 0; q < 4; q++) x[XI(brev12(tid + WFH_THREADS * (u0 + q)))] = v2[q];
        }
        __syncthreads();
        dft_half(x, tws, a.tws, tid);
    }
    // last stage in registers: only re(cx) is looked at — the arg-max of re(cx)^2 over i in [L-1, K+L), first maximum wins (:200-207)
    float best = 0.f, bestc = 0.f; int bidx = -1;
#pragma unroll 4
    for (int u = 0; u < NU; u++) {
        const int g = tid + u * WFH_THREADS;
        const float2 w = a.tws[((1 << 12) - 1) + g];
        const float2 p = park_ld(park + 2 * g), r = cmul(x[XI(g)], w);
        const float c0 = p.x + r.x, c1 = p.x - r.x;
        const int i0 = g, i1 = g + (1 << 12);
        if (i0 >= L - 1 && i0 < wl) { const float c2 = c0 * c0; if (c2 > best || (c2 == best && bidx >= 0 && i0 < bidx)) { best = c2; bidx = i0; bestc = c0; } }
        if (i1 >= L - 1 && i1 < wl) { const float c2 = c1 * c1; if (c2 > best || (c2 == best && bidx >= 0 && i1 < bidx)) { best = c2; bidx = i1; bestc = c1; } }
    }
    {
        float rb = best; int ri = bidx;
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(rb, off); const int oi = __shfl_xor(ri, off);
            if (ob > rb || (ob == rb && oi >= 0 && (ri < 0 || oi < ri))) { rb = ob; ri = oi; }
        }
        if (lane == 0) { s_rf[wave] = rb; s_ri[wave] = ri; }
    }
    __syncthreads();
    int mp = -1;
    {
        float b = 0.f;
        for (int w = 0; w < WFH_THREADS / WAVE; w++) {
            const float ob = s_rf[w]; const int oi = s_ri[w];
            if (ob > b || (ob == b && oi >= 0 && (mp < 0 || oi < mp))) { b = ob; mp = oi; }
        }
    }
    __syncthreads();
    if (mp < 0) {
        // not one correlation value above zero — a stream that begins with digital silence.  The reference's loop leaves mp = -1 (:200-207), which is no edge value:
        // getCorrDFT runs on and sets mv = 0 / (a norm read in front of the array) and mv_pos = pos - (K + L - 1) - 1, which wraps in the first window of a stream —
        // and `mv_pos > mvpos0` (find_header, :1603) then fails for the header the NEXT window finds.  Handed on as rc = -5 (k_framesync takes the position)
        if (tid == 0) { it->rc = -5; it->mv = 0.f; it->mpos = pos - (uint32_t)(wl - 1) - 1u; __threadfence(); it->state = 2; }
        return;
    }
    if (mp == L - 1 || mp == wl - 1) {                                    // edge value: -4 (:208)
        if (tid == 0) { it->rc = -4; it->mv = 0.f; it->mpos = 0; __threadfence(); it->state = 2; }
        return;
    }
    // xnorm = sqrt(sum_{i<L} xn[mp-i]^2) (:215-217), added up in k_sync_window_fft's order so that the header score is the same float: a thread keeps the
    // partial sums of the WF_THREADS / WFH_THREADS threads of that kernel it stands for (same lane, waves wave + 4 j) apart
    constexpr int NV = WF_THREADS / WFH_THREADS;
    static_assert(WF_THREADS % WFH_THREADS == 0, "the energy sum follows k_sync_window_fft's partition");
    float e[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        e[j] = 0.f;
        for (int k = tid + j * WFH_THREADS; k < L; k += WF_THREADS) {
            const int i = mp - k; const int64_t p2 = start + i;
            const float v = (i < wl && p2 >= 0) ? bufs[(uint32_t)p2 & mask] : 0.f;
            e[j] += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) e[j] += __shfl_xor(e[j], off);
        if (lane == 0) s_rf[wave + j * (WFH_THREADS / WAVE)] = e[j];
    }
    if (bidx == mp) s_rf[WF_THREADS / WAVE] = bestc;
    __syncthreads();
    if (tid == 0) {
        float es = 0.f;
        for (int w = 0; w < WF_THREADS / WAVE; w++) es += s_rf[w];
        const float xnorm = sqrtf(es);
        it->rc = mp; it->mv = s_rf[WF_THREADS / WAVE] / (xnorm * (float)N); it->mpos = pos - (uint32_t)(wl - 1) + (uint32_t)mp;
        __threadfence(); it->state = 2;
    }
}
__global__ __launch_bounds__(WFH_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_sync_window_fft_h(const WinFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem2[];
    float2 *x = smem2;                           // [SCH_XN] padded (XI): one half of the transform's array at a time
    float2 *tws = smem2 + SCH_XN;                // [SC_TW_LDS + 1] twiddles of stages 0..8
    __shared__ float s_rf[WF_THREADS / WAVE + 1];
    __shared__ int s_ri[WFH_THREADS / WAVE];
    const uint32_t count = a.work_count[a.round_parity];
    if (blockIdx.x >= count) return;
    for (int k = threadIdx.x; k < SC_TW_LDS; k += WFH_THREADS) tws[k] = a.tws[k];
    for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {
        const uint32_t item = a.work[w];
        __syncthreads();
        sync_eval_window_h(a, (int)(item / (uint32_t)a.stride), a.items + item, x, tws, s_rf, s_ri, a.park + (size_t)blockIdx.x * SC_N);
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from sonde_engine.cpp)
// ------------------------------------------------------------------------------------------------
extern "C" int sonde_launch_mix_decimate(const MixDecArgs *a, hipStream_t s) {
    if (a->D < 1 || a->Q < 1 || a->Q > 8) return -1;
    const int rows_per_wg = 4 * (MD_ROWS * a->G - (a->Q - 1));
    const int wgs_per_ch = (a->nblocks + rows_per_wg - 1) / rows_per_wg;
    MixDecArgs b = *a; b.wgs_per_ch = wgs_per_ch;
    const dim3 grid(((a->n_ch + 7) / 8) * 8 * wgs_per_ch), blk(256);
    if (a->D > 64) {                           // wide variant: D walked in pieces of DS samples, taps from a.wtab_g
        if (!a->wtab_g || a->DS < 1 || a->DS > 64 || a->D % a->DS) return -1;
        const size_t ldsw = (size_t)4 * MD_ROWS * ((a->DS + 3) & ~3) * sizeof(uint32_t);
#define MDW_LAUNCH(QT) do { if (a->phase_f64) hipLaunchKernelGGL((k_mix_decimate_wide<QT, true>), grid, blk, ldsw, s, b); \
                          else hipLaunchKernelGGL((k_mix_decimate_wide<QT, false>), grid, blk, ldsw, s, b); } while (0)
        switch (a->Q) {
            case 5: MDW_LAUNCH(5); break;
            case 6: MDW_LAUNCH(6); break;
            case 7: MDW_LAUNCH(7); break;
            case 8: MDW_LAUNCH(8); break;
            default: return -1;
        }
#undef MDW_LAUNCH
        return 0;
    }
    const size_t lds = (size_t)4 * (MD_ROWS * a->D + 4) * sizeof(uint32_t);
#define MD_LAUNCH(QT) do { if (a->phase_f64) hipLaunchKernelGGL((k_mix_decimate<QT, true, 0, 0>), grid, blk, lds, s, b); \
                         else hipLaunchKernelGGL((k_mix_decimate<QT, false, 0, 0>), grid, blk, lds, s, b); } while (0)
    static const bool no_dt = getenv("SONDE_NO_DT") != nullptr;      // debugging aid: force the runtime-D variant
    if (a->Q == 7 && a->D == 50 && !no_dt) {            // 2.4 Msps -> 48 kHz: decimation known at compile time
        // fast / fold mode is a property of the ENGINE (a->etab set): the P tail between calls then holds sums without the IQ-DC
        // term, so every launch of such an engine goes through one of the two kernels that subtract avg * E per output
        static const bool no_k50 = getenv("SONDE_MD_NO50") != nullptr;        // A/B aid: the kernel with one tile in flight
        if (a->etab && !a->phase_f64 && a->nd_base == 0.0 && a->lut_phase % 50 == 0) {
            if (a->nblocks >= 64 && a->nblocks % 2 == 0 && !no_k50) {
                static const int var = getenv("SONDE_MD_VARIANT") ? atoi(getenv("SONDE_MD_VARIANT")) : 1;
#ifdef MD50_LOOP_2
                if (var == 2) { hipLaunchKernelGGL((k_mix_decimate50<2>), grid, blk, lds, s, b); return 0; }
#endif
#ifdef MD50_LOOP_3
                if (var == 3) { hipLaunchKernelGGL((k_mix_decimate50<3>), grid, blk, lds, s, b); return 0; }
#endif
#ifdef MD50_LOOP_4
                if (var == 4) { hipLaunchKernelGGL((k_mix_decimate50<4>), grid, blk, lds, s, b); return 0; }
#endif
#ifdef MD50_LOOP_5
                if (var == 5) { hipLaunchKernelGGL((k_mix_decimate50<5>), grid, blk, lds, s, b); return 0; }
#endif
                (void)var;
                hipLaunchKernelGGL((k_mix_decimate50<1>), grid, blk, lds, s, b);
            }
            else hipLaunchKernelGGL((k_mix_decimate<7, false, 50, 1>), grid, blk, lds, s, b);
        }
        else if (a->etab) return -1;                          // a fold-mode engine must never fall back to sums with the IQ-DC term
        else if (a->phase_f64 && a->dc_seg && a->nd_base == 0.0 && !a->epoch_phase && a->lut_len % 50 == 0 && a->lut_phase % 50 == 0 && a->dc_seg_blocks > 0
                 && a->nblocks >= 64 && a->nblocks % 2 == 0 && !no_k50 && a->wtab_scaled)
            hipLaunchKernelGGL(k_mix_decimate50s, grid, blk, lds, s, b);      // the scanner's front end on the generated tile loop
        else if (a->phase_f64) hipLaunchKernelGGL((k_mix_decimate<7, true, 50, 0>), grid, blk, lds, s, b);
        else hipLaunchKernelGGL((k_mix_decimate<7, false, 50, 0>), grid, blk, lds, s, b);
        return 0;
    }
    switch (a->Q) {
        case 1: MD_LAUNCH(1); break;
        case 2: MD_LAUNCH(2); break;
        case 3: MD_LAUNCH(3); break;
        case 4: MD_LAUNCH(4); break;
        case 5: MD_LAUNCH(5); break;
        case 6: MD_LAUNCH(6); break;
        case 7: MD_LAUNCH(7); break;
        default: MD_LAUNCH(8); break;
    }
#undef MD_LAUNCH
    return 0;
}
extern "C" void sonde_launch_dc_segments(const int16_t *iq, long long ch_stride, int n_ch, int n_samples, unsigned dc_cnt0, unsigned dc_max,
                                         long long *seg_sums, long long *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n, hipStream_t s) {
    const int nseg = (int)(((unsigned long long)dc_cnt0 + (unsigned)n_samples + dc_max - 1) / dc_max);
    const int ncomplete = (int)(((unsigned long long)dc_cnt0 + (unsigned)n_samples) / dc_max);
    hipLaunchKernelGGL(k_dc_seg_sums, dim3(nseg, n_ch), dim3(256), 0, s, iq, ch_stride, n_samples, dc_cnt0, dc_max, seg_sums, nseg);
    hipLaunchKernelGGL(k_dc_seg_means, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, nseg, ncomplete, (float)dc_max, seg_sums, dc_sums, dc_avg, dc_seg, dc_seg_n);
}
extern "C" int sonde_launch_mix_decimate50r(const MixDecArgs *a, hipStream_t s) {
    if (a->D != 50 || a->Q != 7 || a->lut_len % 50 || a->lut_phase % 50 || !a->wtab_scaled || !a->bsum || a->bsum_stride < a->nblocks || a->dc_seg) return -1;
    MixDecArgs b = *a;
    const int H = b.Q - 1, rps = 64 * b.G - H;
    if (b.G < 1 || rps < 1) return -1;
    const int segs = (b.nblocks + rps - 1) / rps;
    b.wgs_per_ch = (segs + 3) / 4;
    const int ch8 = (b.n_ch + 7) / 8;
    const size_t lds = (size_t)4 * (64 * b.D + 4) * sizeof(uint32_t);
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mix_decimate50r), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(k_mix_decimate50r, dim3((unsigned)(ch8 * b.wgs_per_ch * 8)), dim3(256), lds, s, b);
    return 0;
}
extern "C" void sonde_launch_dc_rows_to_segments(const int2 *bsum, long long bsum_stride, int n_ch, int nblocks, int seg_off, int seg_blocks, float maxcnt,
                                                 long long *seg_sums, long long *dc_sums, float2 *dc_avg, const float2 *dc_prev, float2 *dc_prev_out, float2 *dc_seg, int dc_seg_n, hipStream_t s) {
    const int nseg = (seg_off + nblocks + seg_blocks - 1) / seg_blocks, ncomplete = (seg_off + nblocks) / seg_blocks;
    hipLaunchKernelGGL(k_dc_rows_to_segments, dim3(nseg, n_ch), dim3(256), 0, s, bsum, bsum_stride, nblocks, seg_off, seg_blocks, seg_sums, nseg);
    hipLaunchKernelGGL(k_dc_seg_means, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, nseg, ncomplete, maxcnt, seg_sums, dc_sums, dc_avg, dc_seg, dc_seg_n, dc_prev, dc_prev_out);
}
extern "C" void sonde_launch_scan_dc_edges(const ScanEdgeArgs *a, hipStream_t s) {
    if (a->nseg <= 0) return;
    hipLaunchKernelGGL(k_scan_dc_edges, dim3(a->nseg, a->n_ch), dim3(64), 0, s, *a);
}
extern "C" void sonde_launch_dc_segments_f32(const float2 *x, long long ch_stride, int n_ch, int n_samples, unsigned dc_cnt0, unsigned dc_max,
                                             double *seg_sums, double *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n, hipStream_t s) {
    const int nseg = (int)(((unsigned long long)dc_cnt0 + (unsigned)n_samples + dc_max - 1) / dc_max);
    const int ncomplete = (int)(((unsigned long long)dc_cnt0 + (unsigned)n_samples) / dc_max);
    hipLaunchKernelGGL(k_dc_seg_sums_f32, dim3(nseg, n_ch), dim3(256), 0, s, x, ch_stride, n_samples, dc_cnt0, dc_max, seg_sums, nseg);
    hipLaunchKernelGGL(k_dc_seg_means_f32, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, nseg, ncomplete, (float)dc_max, seg_sums, dc_sums, dc_avg, dc_seg, dc_seg_n);
}
extern "C" void sonde_launch_dc_update(int n_ch, long long *sums, float2 *avg, float maxcnt, hipStream_t s) {
    hipLaunchKernelGGL(k_dc_update, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, sums, avg, (float2 *)nullptr, maxcnt);
}
extern "C" void sonde_launch_dc_update_keep(int n_ch, long long *sums, float2 *avg, float2 *avg_prev, float maxcnt, hipStream_t s) {
    hipLaunchKernelGGL(k_dc_update, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, sums, avg, avg_prev, maxcnt);
}
extern "C" void sonde_launch_dc_update_pcs(int n_ch, long long *sums, float2 *avg, float2 *avg_prev, uint32_t *cnt, uint32_t *max, uint32_t lim, int32_t *since,
                                           uint32_t n_samples, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_dc_update_pcs, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, sums, avg, avg_prev, cnt, max, lim, since, n_samples, nblocks);
}
extern "C" void sonde_launch_publish_u32(const unsigned *src, unsigned *dst_mapped, hipStream_t s) {
    hipLaunchKernelGGL(k_publish_u32, dim3(1), dim3(64), 0, s, src, dst_mapped);
}
extern "C" void sonde_launch_md_etable(const double *chan_f0, const float *wtab, int D, int Q, int P, int n_ch, float2 *etab, hipStream_t s) {
    hipLaunchKernelGGL(k_md_etable, dim3((P + 255) / 256, n_ch), dim3(256), 0, s, chan_f0, wtab, D, Q, P, etab, 0);
}
extern "C" void sonde_launch_md_etable64(const double *chan_f0, const float *wtab, int D, int Q, int P, int n_ch, float2 *etab, hipStream_t s) {
    hipLaunchKernelGGL(k_md_etable, dim3((P + 255) / 256, n_ch), dim3(256), 0, s, chan_f0, wtab, D, Q, P, etab, 1);
}
extern "C" void sonde_launch_u8_to_s16(const uint8_t *in, long long in_stride, int16_t *out, long long out_stride, int n_ch, int n_bytes, hipStream_t s) {
    int gx = (n_bytes / 4 + 255) / 256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_u8_to_s16, dim3(gx, n_ch), dim3(256), 0, s, in, in_stride, out, out_stride, n_bytes);
}
extern "C" void sonde_launch_mix_f32(const MixF32Args *a, hipStream_t s) {
    const int gx = (a->n + MF32_CHUNK - 1) / MF32_CHUNK;
    if (gx < 1) return;
    hipLaunchKernelGGL(k_mix_f32, dim3(gx, a->n_ch), dim3(256), 0, s, *a);
}
extern "C" void sonde_launch_decimate_f32(const DecF32Args *a, hipStream_t s) {
    if (a->nblocks <= 0) return;
    int jper = DF32_J;
    auto lds_of = [&](int j) { return (size_t)(a->D * (j - 1) + a->T) * sizeof(float2) + (size_t)a->T * sizeof(float); };
    while (jper > 8 && lds_of(jper) > 64 * 1024) jper >>= 1;
    if (lds_of(jper) > 64 * 1024) { hipLaunchKernelGGL(k_decimate_f32_global, dim3((a->nblocks + 255) / 256, a->n_ch), dim3(256), 0, s, *a); return; }
    hipLaunchKernelGGL(k_decimate_f32, dim3((a->nblocks + jper - 1) / jper, a->n_ch), dim3(DF32_J), lds_of(jper), s, *a, jper);
}
extern "C" void sonde_launch_dc_update_f64(int n_ch, double *sums, float2 *avg, float maxcnt, hipStream_t s) {
    hipLaunchKernelGGL(k_dc_update_f64, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, sums, avg, maxcnt);
}
extern "C" void sonde_launch_afc_rotate(const AfcRotArgs *a, hipStream_t s, int n_max) {
    int gx = (n_max + 255) / 256; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_afc_rotate, dim3(gx, a->n_ch), dim3(256), 0, s, *a);
}
extern "C" void sonde_launch_fill_u32(uint32_t *p, uint32_t v, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_u32, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
}
extern "C" void sonde_launch_audio_chain(const AudioChainArgs *a, hipStream_t s) {
    int gx = (a->n + 255) / 256; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_audio_chain, dim3(gx, a->n_ch), dim3(256), 0, s, *a);
}
extern "C" void sonde_launch_if_chain(const IfArgs *a, hipStream_t s) {
    const int T1 = a->lpiq_on ? a->lpiq_taps : 1, T2 = a->lpfm_on ? a->lpfm_taps : 1;
    const int hz = (T2 - 1) + (a->nwin - 1 > 1 ? a->nwin - 1 : 1);
    const int nz = hz + IF_TILE, ny = nz + T1 - 1;
    const size_t lds = (size_t)((ny + 2 * IF_NB + 1) & ~1) * 8 + (size_t)(a->fm_on ? nz + 1 : 0) * 8 + (size_t)(a->tone_on ? nz : 0) * 16 + (size_t)(a->fm_on ? T2 - 1 + IF_TILE : 0) * 4 + (size_t)(T1 + T2) * 4 + 32;
    hipLaunchKernelGGL(k_if_chain, dim3((a->n + IF_TILE - 1) / IF_TILE, a->n_ch), dim3(IF_THREADS), lds, s, *a);
}
// the groups of a mixed engine in one launch each (rows = channels in the engine's order, group after group)
template <class A> static MultiArgs<A> multi_pack(const A *a, const int *rows, int n_groups) {
    MultiArgs<A> m{};
    m.n_groups = n_groups; m.row0[0] = 0;
    for (int g = 0; g < n_groups; g++) { m.g[g] = a[g]; m.row0[g + 1] = m.row0[g] + rows[g]; }
    return m;
}
extern "C" int sonde_launch_if_chain_multi(const IfArgs *a, int n_groups, hipStream_t s) {
    if (n_groups < 1 || n_groups > SONDE_MAX_GROUPS) return -1;
    size_t lds = 0; int rows[SONDE_MAX_GROUPS], n = a[0].n;
    for (int g = 0; g < n_groups; g++) {
        const int T1 = a[g].lpiq_on ? a[g].lpiq_taps : 1, T2 = a[g].lpfm_on ? a[g].lpfm_taps : 1;
        const int hz = (T2 - 1) + (a[g].nwin - 1 > 1 ? a[g].nwin - 1 : 1);
        const int nz = hz + IF_TILE, ny = nz + T1 - 1;
        const size_t l = (size_t)((ny + 2 * IF_NB + 1) & ~1) * 8 + (size_t)(a[g].fm_on ? nz + 1 : 0) * 8 + (size_t)(a[g].tone_on ? nz : 0) * 16 + (size_t)(a[g].fm_on ? T2 - 1 + IF_TILE : 0) * 4 + (size_t)(T1 + T2) * 4 + 32;
        if (l > lds) lds = l;
        rows[g] = a[g].n_ch;
        if (a[g].n != n) return -1;                              // (one call = the same samples for every channel)
    }
    const MultiArgs<IfArgs> m = multi_pack(a, rows, n_groups);
    hipLaunchKernelGGL(k_if_chain_multi, dim3((n + IF_TILE - 1) / IF_TILE, m.row0[n_groups]), dim3(IF_THREADS), lds, s, m);
    return 0;
}
extern "C" int sonde_launch_sync_plan_multi(const WinPlanArgs *a, int n_groups, hipStream_t s) {
    if (n_groups < 1 || n_groups > SONDE_MAX_GROUPS) return -1;
    int rows[SONDE_MAX_GROUPS];
    for (int g = 0; g < n_groups; g++) rows[g] = a[g].n_ch;
    const MultiArgs<WinPlanArgs> m = multi_pack(a, rows, n_groups);
    hipLaunchKernelGGL(k_sync_plan_multi, dim3((m.row0[n_groups] + 255) / 256), dim3(256), 0, s, m);
    return 0;
}
extern "C" int sonde_launch_sync_window_fft_multi(const WinFftArgs *a, int n_groups, hipStream_t s) {
    if (n_groups < 1 || n_groups > SONDE_MAX_GROUPS) return -1;
    // two waves of workgroups on 256 CUs at most, shared out by the windows a group can have planned (W per channel)
    long long tot = 0; int rows[SONDE_MAX_GROUPS];
    for (int g = 0; g < n_groups; g++) tot += (long long)a[g].W * a[g].n_ch;
    const long long cap = tot > 512 ? 512 : tot;
    for (int g = 0; g < n_groups; g++) { const long long w = (long long)a[g].W * a[g].n_ch; rows[g] = (int)((w * cap + tot - 1) / tot); if (rows[g] < 1) rows[g] = 1; }
    const MultiArgs<WinFftArgs> m = multi_pack(a, rows, n_groups);
    hipLaunchKernelGGL(k_sync_window_fft_multi, dim3(m.row0[n_groups]), dim3(WF_THREADS), (size_t)(SC_XN + SC_TW_LDS + 1) * sizeof(float2), s, m);
    return 0;
}
extern "C" void sonde_launch_header_corr(const CorrArgs *a, hipStream_t s) {
    if (a->ntypes > 0) {
        const int nf = (HCF_TILE + a->isps * (a->nsym - 1) + 3) & ~3, nxp = (nf + a->isps + 3 + 3) & ~3;
        const size_t lds = (size_t)(nxp + a->ntypes * nf) * sizeof(float);
        hipLaunchKernelGGL(k_header_corr_fact, dim3((a->n + HCF_TILE - 1) / HCF_TILE, a->n_ch), dim3(HCF_THREADS), lds, s, *a);
        return;
    }
    const size_t lds = (size_t)(HC_TILE + 2 * a->L + 8) * sizeof(float);
    hipLaunchKernelGGL(k_header_corr, dim3((a->n + HC_TILE - 1) / HC_TILE, a->n_ch), dim3(HC_THREADS), lds, s, *a);
}
extern "C" void sonde_launch_sync_plan(const WinPlanArgs *a, hipStream_t s) {
    hipLaunchKernelGGL(k_sync_plan, dim3((a->n_ch + 255) / 256), dim3(256), 0, s, *a);
}
extern "C" void sonde_launch_sync_window_fft(const WinFftArgs *a, hipStream_t s) {
    const size_t lds = (size_t)(SC_XN + SC_TW_LDS + 1) * sizeof(float2);
    int grid = a->W * a->n_ch;
    if (a->small_wg) {                                          // the half-array form: fits the slot of one decimator workgroup (three per CU)
        if (grid > SONDE_WFH_MAXGRID) grid = SONDE_WFH_MAXGRID;
        hipLaunchKernelGGL(k_sync_window_fft_h, dim3(grid), dim3(WFH_THREADS), (size_t)(SCH_XN + SC_TW_LDS + 1) * sizeof(float2), s, *a);
        return;
    }
    if (grid > 512) grid = 512;                                 // two waves of workgroups on 256 CUs at most; the kernel strides over the list
    hipLaunchKernelGGL(k_sync_window_fft, dim3(grid), dim3(WF_THREADS), lds, s, *a);
}
// rs41_ecc() of the frames k_framesync put on its work list (records with ecc_done == 2): one workgroup of four waves per frame — small enough
// to take the place of ONE decimator workgroup (51 KB of LDS, a wave per SIMD), so that on its own stream it runs in the slots the next call's
// decimator frees instead of waiting for a whole CU.  Waves 0 / 1 decode a codeword each from the first-pass syndromes in the record; a 2nd
// pass (--ecc2) recomputes them with all four waves.  The list counter is reset by the last workgroup to finish.
#define RSK_THREADS 256
struct RsEccLds { uint8_t frame[520], exp[512], log[256], cw[2][256], part[RSK_THREADS / 64][48], scr[2][64], S[48]; int res[4]; };
__device__ __forceinline__ void rs_ecc_tables(RsEccLds &L, const uint8_t *gf_exp, const uint8_t *gf_log, int tid) {
    for (int i = tid; i < 512; i += RSK_THREADS) L.exp[i] = gf_exp[i];
    if (tid < 256) L.log[tid] = gf_log[tid];
}
__global__ __launch_bounds__(RSK_THREADS)
void k_rs41_ecc_frames(FrameRec *frames, const uint32_t *list, unsigned *count, unsigned *done, int max_frames, int level,
                       const uint8_t *gf_exp, const uint8_t *gf_log) {
    __shared__ RsEccLds L;
    __shared__ unsigned s_last;
    const int tid = threadIdx.x;
    const unsigned n = min(*count, (unsigned)max_frames);
    if (n) rs_ecc_tables(L, gf_exp, gf_log, tid);
    for (unsigned w = blockIdx.x; w < n; w += gridDim.x) {
        FrameRec *rec = frames + list[w];
        __syncthreads();
        if (rec->ecc_done != 2) continue;                                  // (uniform: every thread reads the same word)
        const int flen = rec->len;
        for (int i = tid; i < 518; i += RSK_THREADS) L.frame[i] = i < flen ? rec->frame[i] : 0;      // rs41mod.c:1727
        if (tid < 48) L.S[tid] = rec->synd[tid];
        __syncthreads();
        const RsGf gf{L.exp, L.log};
        const int r = rs41_ecc_wg(L.frame, level, L.cw, L.part, L.res, L.scr, L.S, gf, tid, RSK_THREADS);
        for (int i = tid; i < 518; i += RSK_THREADS) rec->frame[i] = L.frame[i];
        __threadfence();
        __syncthreads();
        if (tid == 0) { rec->ecc = r; rec->ecc_done = 1; }
    }
    // the last workgroup out clears the list for the call after next (two lists alternate)
    __syncthreads();
    if (tid == 0) { __threadfence(); s_last = atomicAdd(done, 1u); }
    __syncthreads();
    if (s_last == gridDim.x - 1 && tid == 0) { *count = 0; *done = 0; }
}
extern "C" void sonde_launch_rs41_ecc_frames(FrameRec *frames, const uint32_t *list, unsigned *count, unsigned *done, int max_frames, int level,
                                             const uint8_t *gf_exp, const uint8_t *gf_log, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_rs41_ecc_frames, dim3(grid), dim3(RSK_THREADS), 0, s, frames, list, count, done, max_frames, level, gf_exp, gf_log);
}

// rs41_ecc() over a batch of de-whitened 518-byte frames, one workgroup per frame: the same workgroup function on frames that come from
// somewhere else (tests: word-by-word parity with the compiled reference; callers with frames from --softin or a file); syndromes computed here
__global__ __launch_bounds__(RSK_THREADS)
void k_rs41_ecc_batch(uint8_t *frames, const int32_t *flen, int level, int32_t *ecc, int32_t *codes, uint8_t *synd, const uint8_t *gf_exp, const uint8_t *gf_log, const unsigned *count) {
    __shared__ RsEccLds L;
    const int tid = threadIdx.x, f = blockIdx.x;
    if (count && (unsigned)f >= *count) return;                           // (a batch whose size only the device knows: the soft-bit framer's frames of this call)
    rs_ecc_tables(L, gf_exp, gf_log, tid);
    for (int i = tid; i < 518; i += RSK_THREADS) L.frame[i] = i < flen[f] ? frames[(size_t)f * 518 + i] : 0;      // rs41mod.c:1727
    __syncthreads();
    const RsGf gf{L.exp, L.log};
    // first-pass syndromes for the caller, then the decoder as k_rs41_ecc_frames runs it (syndromes handed in)
    for (int i = tid; i < 510; i += RSK_THREADS) { const int c = i / 255, n = i % 255; L.cw[c][n] = n < 24 ? L.frame[8 + 24 * c + n] : L.frame[56 + 2 * (n - 24) + c]; }
    __syncthreads();
    rs41_syndrome_partials(L.cw, L.part, gf, tid & 63, tid >> 6, RSK_THREADS / 64);
    __syncthreads();
    if (tid < 48) { uint8_t sy = 0; for (int w = 0; w < RSK_THREADS / 64; w++) sy ^= L.part[w][tid]; L.S[tid] = sy; synd[(size_t)f * 48 + tid] = sy; }
    __syncthreads();
    const int r = rs41_ecc_wg(L.frame, level, L.cw, L.part, L.res, L.scr, (f & 1) ? L.S : (const uint8_t *)nullptr, gf, tid, RSK_THREADS);
    for (int i = tid; i < 518; i += RSK_THREADS) frames[(size_t)f * 518 + i] = L.frame[i];
    if (tid == 0) { ecc[f] = r; codes[2 * f] = L.res[0]; codes[2 * f + 1] = L.res[1]; }
}
extern "C" void sonde_launch_rs41_ecc_batch(uint8_t *frames, const int32_t *flen, int n, int level, int32_t *ecc, int32_t *codes, uint8_t *synd,
                                            const uint8_t *gf_exp, const uint8_t *gf_log, hipStream_t s) {
    hipLaunchKernelGGL(k_rs41_ecc_batch, dim3(n), dim3(RSK_THREADS), 0, s, frames, flen, level, ecc, codes, synd, gf_exp, gf_log, (const unsigned *)nullptr);
}
extern "C" void sonde_launch_rs41_ecc_batch_n(uint8_t *frames, const int32_t *flen, const unsigned *count, int cap, int level, int32_t *ecc, int32_t *codes, uint8_t *synd,
                                              const uint8_t *gf_exp, const uint8_t *gf_log, hipStream_t s) {
    hipLaunchKernelGGL(k_rs41_ecc_batch, dim3(cap), dim3(RSK_THREADS), 0, s, frames, flen, level, ecc, codes, synd, gf_exp, gf_log, count);
}

extern "C" int sonde_launch_framesync_multi(const SyncArgs *a, int n_groups, hipStream_t s) {
    if (n_groups < 1 || n_groups > SONDE_MAX_GROUPS) return -1;
    int rows[SONDE_MAX_GROUPS];
    for (int g = 0; g < n_groups; g++) { rows[g] = a[g].n_ch; if (a[g].opt_dc) return -1; }
    const MultiArgs<SyncArgs> m = multi_pack(a, rows, n_groups);
    hipLaunchKernelGGL(k_framesync_multi, dim3(m.row0[n_groups]), dim3(1024), 0, s, m);
    return 0;
}
extern "C" void sonde_launch_framesync(const SyncArgs *a, hipStream_t s) {
    if (a->opt_dc) hipLaunchKernelGGL((k_framesync<true, FS_THREADS>), dim3(a->n_ch), dim3(FS_THREADS), 0, s, *a);
    else if (a->small_wg) hipLaunchKernelGGL((k_framesync<false, FS_THREADS_SMALL>), dim3(a->n_ch), dim3(FS_THREADS_SMALL), 0, s, *a);
    else hipLaunchKernelGGL((k_framesync<false, FS_THREADS>), dim3(a->n_ch), dim3(FS_THREADS), 0, s, *a);
}
