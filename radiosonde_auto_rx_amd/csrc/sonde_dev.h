// sonde_dev.h — structs shared by the HIP kernels (sonde_kernels.hip) and the host engine (sonde_engine.cpp).
#ifndef SONDE_DEV_H
#define SONDE_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sonde_hip.h"

struct MixDecArgs {
    const int16_t *iq;        // [n_ch][ch_stride] complex int16
    long long ch_stride;      // complex samples between channels
    int n_ch, nblocks;        // blocks (= IF samples) in this chunk
    int D, Q;                 // decM, ceil(taps/D)
    int G;                    // 64-row tiles per wave
    int wgs_per_ch;           // filled by the launcher
    float wtab[64 * 8];       // [D][8]: wtab[r][q] = front-padded tap D*q + r; in the kernarg segment -> scalar loads
    const double *chan_f0;    // [n_ch] snapped mixer frequency / sample rate (demod_mod.c:1288)
    int lut_len;              // period of the reference's mixer table (sr_base / d)
    uint32_t lut_phase;       // table index of the chunk's first sample
    const float2 *dc_avg;     // [n_ch] IQ-DC mean in effect
    long long *dc_sums;       // [n_ch][2] integer sums of the running DC segment
    const float2 *ptail_in;   // [n_ch][8][8] P rows of the Q-1 blocks before the chunk
    float2 *ptail_out;
    float2 *y;                // [n_ch][ring_len] decimated IQ ring
    int ring_len;
    uint32_t m0;              // IF index of the chunk's first output
    const float *wtab_g;      // D > 64: [D][8] tap table in global memory, and the piece length DS (divides D, <= 64)
    int wtab_scaled;          // wtab_g holds a second table behind the first 64 rows: the tap rows * 2^-15 (the generated D = 50 kernels read it)
    int DS;
    int phase_f64;            // mixer phase f0*n kept in double (dft_detect.c:1090) instead of the float of demod_mod.c:1290
    double nd_base;           // --noLUT: absolute index of the launch's first sample (phase = f0 * absolute index, no table period); else 0
    // fast / fold mode (D = 50, Q = 7, float table phase, lut_len % D == 0): E[ch][etab_len] = the decimator's response to the bare mixer
    // table, etab_len = lut_len / D.  Set for every launch of such an engine or for none: the P tail then holds sums without the IQ-DC term.
    const float2 *etab; int etab_len;
    const float2 *dc_avg_prev; // fold mode, launches within Q-1 blocks after a change of the IQ-DC mean: the mean before (md_dc_boundary); else nullptr
    int dc_since;             // blocks between that change and this launch
    // channels restarted at run time (sonde_engine_restart_channel on a base-rate engine) have their own sample clock: the mixer table phase
    // and the IQ-DC segment schedule count from the channel's start.  nullptr = all channels started with the engine.
    const uint32_t *epoch_phase;   // [n_ch] (base-rate index of the channel's first sample) mod lut_len
    const int32_t *dc_since_ch;    // [n_ch] blocks since the channel's mean last changed (replaces dc_since; dc_avg_prev is then always set)
    // a launch that spans several IQ-DC windows (the scanner's are 1/32 s, dft_detect.c:539-573: a launch per window would be 64 launches per second of
    // signal): dc_seg[ch][k] = the mean in effect for the k-th window the launch touches, block j lies in window (j + dc_seg_off) / dc_seg_blocks.
    // nullptr = one mean for the whole launch (dc_avg).  Generic kernel only; the kernel then leaves dc_sums alone (k_dc_seg_* own them).
    const float2 *dc_seg; int dc_seg_n, dc_seg_off, dc_seg_blocks;
    // k_mix_decimate50r (the scanner's front end in one pass over the input): no mean at all in the sample loop — y receives the raw sum W x ex,
    // bsum[ch][j] the sum of the raw samples of block j (re, im: exact integers); k_dc_rows_to_segments / k_dc_seg_means / k_scan_dc_edges follow at the IF rate, k_scan_if folds as it loads
    int2 *bsum; long long bsum_stride;
    const int32_t *in_row;    // nullable: [n_ch] row of `iq` channel ch reads (mixed engines: channels grouped by type inside, the caller's order outside)
};
// (ScanFold, what k_scan_if folds with, is in sonde_scan_dev.h)
struct ScanEdgeArgs {
    int n_ch, nblocks, D, Q, nseg;
    const float2 *dc_seg; int dc_seg_n, dc_seg_off, dc_seg_blocks;
    const float2 *dc_prev;                          // [n_ch] the mean of the window before the one dc_seg[ch][0] belongs to (its value at the START of the call)
    int etab_len; uint32_t e0;
    const double *chan_f0; const float *wtab;
    float2 *corr;
};
extern "C" int  sonde_launch_mix_decimate50r(const MixDecArgs *a, hipStream_t s);
extern "C" void sonde_launch_dc_rows_to_segments(const int2 *bsum, long long bsum_stride, int n_ch, int nblocks, int seg_off, int seg_blocks, float maxcnt,
                                                 long long *seg_sums, long long *dc_sums, float2 *dc_avg, const float2 *dc_prev, float2 *dc_prev_out, float2 *dc_seg, int dc_seg_n, hipStream_t s);
extern "C" void sonde_launch_scan_dc_edges(const ScanEdgeArgs *a, hipStream_t s);
extern "C" void sonde_launch_md_etable64(const double *chan_f0, const float *wtab, int D, int Q, int P, int n_ch, float2 *etab, hipStream_t s);
extern "C" void sonde_launch_dc_segments(const int16_t *iq, long long ch_stride, int n_ch, int n_samples, unsigned dc_cnt0, unsigned dc_max,
                                         long long *seg_sums, long long *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n, hipStream_t s);

// --dc (AFC) per-channel state: what find_header keeps in dsp.Df / dsp.locked / dsp.dc (demod_mod.c:1555-1600, 280-298)
struct AfcState {
    double Df;                // accumulated frequency correction, Hz
    double dc;                // FM mean under the last header (dsp.dc)
    int locked;               // 1: nominal IF filter, 0: 1.5x acquisition filter
    uint32_t pad;
};

// float32 input (`- <sr> 32`, cf32): no tuned path — two plain kernels.  k_mix_f32 writes z = (x - avg) * ex[n] into a ring of
// mixed base-rate samples (the decimator's delay line decXbuffer, demod_mod.c:737-750) and accumulates the IQ-DC sums in double
// like the reference; k_decimate_f32 runs the FIR over that ring, one output per thread, taps in time order.
struct MixF32Args {
    const float2 *x; long long ch_stride; int n_ch, n;         // n complex samples per channel in this launch
    const double *chan_f0; int lut_len; uint32_t lut_phase; int phase_f64; double nd_base;
    const float2 *dc_avg; double *dc_sums;                      // [n_ch][2]
    float2 *z; uint32_t zmask; uint64_t n0;                     // ring [n_ch][zmask+1], absolute base-rate index of the first sample
    int mix;                                                    // 0: no mixer (IF-rate input, --iq0/2/3): z = x - avg
    const uint32_t *epoch;                                      // phase_f64 with IF-rate input (if_tune): per-channel stream start the phase counts from; nullptr = 0
    // a launch that spans several IQ-DC windows (the scanner's 1/32 s windows): sample i runs under dc_seg[ch][(dc_seg_off + i) / dc_seg_len] and the
    // sums are not accumulated here (sonde_launch_dc_segments_f32 has done both); nullptr = one window per launch as above
    const float2 *dc_seg; int dc_seg_n; uint32_t dc_seg_off, dc_seg_len;
};
struct DecF32Args {
    const float2 *z; uint32_t zmask; uint64_t n0;               // ring and the absolute index of the first input sample of output 0
    const float *taps; int T, D, n_ch, nblocks;
    float2 *y; int ring_len; uint32_t m0;
};

struct AfcRotArgs {           // z *= cexp(-t 2 pi Df) at IF rate, t = m / sr in double (demod_mod.c:758-761)
    const float2 *y; float2 *yrot; const AfcState *afc; const uint32_t *start;
    int n_ch, ring_len, sr; uint32_t m_end;
};

struct IfArgs {
    const float2 *y; float2 *tap_ifiq; float *fm; float *bufs;
    // --dc: per-channel restart.  Outputs are produced for m >= start[ch] only; older z' / raw FM samples come from the
    // rings (tap_ifiq = rot_iqbuf, whose tail find_header may have rotated in place; fmraw = lpFM_buf).  nullptr = off.
    const AfcState *afc; const uint32_t *start; float *fmraw; const float *w_iq0;
    int n_ch, ring_len, n; uint32_t m0;
    int lpiq_on, lpiq_taps, lpfm_on, lpfm_taps, tone_on, nwin;
    int fm_on;                // FM discriminator stream wanted (sliced stream of FM modes, --lpFM, taps); off = tone path only
    const float *w_iq, *w_fm;
    double rho;               // tone phase advance per IF sample, revolutions
    float sps;
    const uint32_t *epoch;    // per-channel stream start: the tone phase counts from there; nullptr = 0
};

struct AudioChainArgs {       // FM-audio input: raw ring -> (FM low-pass) -> fm, bufs
    const float *raw; float *fm, *bufs; const float *w;
    int n_ch, ring_len, n, taps; uint32_t m0;
};

struct SyncState {
    uint32_t s_in, k, mv_pos, mode;
    float mv; uint32_t inv;   // inv: current polarity of the channel (-i, flipped by --auto)
    uint32_t pad[2];
};

// Header search with the reference's own transform (getCorrDFT, demod_mod.c:148-225) instead of a time-domain correlation ring:
// k_sync_plan lists, per channel, the next windows its sync state will examine; k_sync_window_fft evaluates each one exactly as
// the reference does (window -> dft_raw -> X * Fm -> dft_raw of the conjugate -> first maximum of re^2 -> norm); k_framesync
// consumes the results in order and stops where they end.  The twiddle recurrence of dft_raw drifts by ~1e-4, and which of two
// almost equal neighbouring maxima wins depends on that drift: only the same arithmetic reproduces the reference's mv_pos.
struct WinItem {
    uint32_t pos;             // sample_out of the window
    int32_t  state;           // 0 unused, 1 planned, 2 evaluated
    int32_t  rc;              // peak index (>= 0) or -4 (edge value)
    float    mv; uint32_t mpos;
    uint32_t pad[3];
};
struct WinPlanArgs {
    const SyncState *state; WinItem *items;
    int n_ch, stride, W, K, L, delay; uint32_t frame_samples, avail;    // table of `stride` slots per channel, this round fills the first W
    const uint32_t *epoch;    // per-channel stream start (sonde_engine_restart_channel), nullptr = 0 everywhere
    uint32_t *work, *work_count; int round_parity;     // compact list of planned items (ch * stride + slot) for k_sync_window_fft; counters [2], alternating per round
};
struct WinFftArgs {
    const float *bufs; WinItem *items; const float2 *Fm, *tws;
    int n_ch, stride, W, K, L, ring_len;
    const uint32_t *work, *work_count; int round_parity;
    unsigned long long *prof; // SONDE_WF_PROF: cycles per phase of workgroup 0 (nullptr = off)
    int small_wg;             // the half-array form of the transform (39 KB of LDS, 256 threads): fits the slot of one decimator workgroup
    float2 *park;             // small_wg: [SONDE_WFH_MAXGRID][8192] per-workgroup parking array in global memory (L2-resident): the half that waits
};
#define SONDE_WFH_MAXGRID 768

struct CorrArgs {
    const float *bufs; float *corr; const float *match;
    const SyncState *state;   // per-channel sync state for skipping tiles no window can reach (nullptr = compute all)
    const uint32_t *start;    // --dc restart: positions below start[ch] are unchanged (nullptr = off)
    int delay; uint32_t frame_samples;
    uint32_t limit;           // != 0: only end positions less than `limit` behind the first one the sync can examine (pass 1 of two)
    int n_ch, ring_len, n, L; uint32_t m0;
    // factorised form (integer samples/symbol): ntypes == 0 selects the direct L-tap kernel
    int ntypes, isps, nsym;
    const float *shapes;      // [ntypes][isps]
    const int *sym_type;      // [nsym]
    const float *sym_sign;    // [nsym]
};

struct FrameRec {
    int32_t channel, len, nbytes; uint32_t mv_pos; float mv;
    int32_t ecc, ecc_done;    // ecc_done 1: rs41_ecc() is done on the device (ecc = its value, frame[] what it leaves in gpx->frame); 2: on the work list of
                              // k_rs41_ecc_frames, not decoded yet; 0: syndromes only, the host decodes
    uint8_t synd[48];
    uint8_t frame[520];
};

struct SyncArgs {
    const float *bufs, *corr; SyncState *state; FrameRec *frames; unsigned *frame_count; float *soft;
    float *soft1;             // optional: the same bit sums one IF sample earlier (hsbit1 of read_softbit2p, demod_mod.c:1120,1145)
    const uint8_t *hdr, *hdr_bytes, *mask, *gf_exp, *gf_log;
    const uint4 *bitwin;      // [nbits] consumed-sample ranges of every bit: {first half (Manchester) qa,qb, main half qa,qb}
    const uint32_t *bitend;   // [nbits] consumed samples after the bit
    int n_ch, ring_len, max_frames; uint32_t avail;
    int K, L, delay, hdrlen, symhd, symlen, hdmax, bitofs, nbits; uint32_t frame_samples;
    float sps, thres, l_win;
    int rs41;                 // RS41 byte framing + syndromes on the device; else packed hard bits + soft bits
    int ecc_level;            // rs41: 1 / 2 = rs41_ecc() of whole frames on the device (--ecc / --ecc2); 0 = first-pass syndromes only
    int small_wg;             // 256-thread workgroups (fit the slot one decimator workgroup frees) instead of 1024
    uint32_t *ecc_list; unsigned *ecc_count;       // work list of k_rs41_ecc_frames (record slots), nullptr = none (damaged frames are decoded by the host)
    int eof;                  // end of stream: emit the frame in progress with the bits that exist
    int eof_ch;               // with eof: only this channel (-1 = all)
    const uint32_t *epoch;    // per-channel stream start, nullptr = 0
    int opt_auto;             // --auto: opposite-polarity header flips SyncState.inv instead of being skipped
    // --dc (demod_mod.c:174-188,227-298,1555-1600): zero-mean windows, FM-stream fallback correlation, header dc, AFC events
    int opt_dc, opt_iq, lpiq_on, lpfm_taps, N, sr;
    float match_sum;
    const float *fm, *corr2; float2 *ifiq;
    AfcState *afc; uint32_t *start; unsigned *pending;
    sonde_summary_t *summary; uint32_t summary_base; int summary_type; uint64_t summary_epoch;      // nullable: per-channel summary records
    const int32_t *summary_map;                               // nullable: [n_ch] record index / channel number of channel ch (groups of a mixed engine)
    const WinItem *win; int win_W;                             // != nullptr: header windows precomputed by k_sync_window_fft (W per channel)
    uint32_t corr_limit;      // != 0: pass 1 of two — `corr` holds CorrArgs.limit end positions behind the state's first one; stop there
    unsigned long long *prof; // SONDE_WF_PROF: cycles per phase of channel 0 (nullptr = off)
};

#define SONDE_MAX_GROUPS 6      // groups of a mixed engine whose IF-rate stages share a launch (the argument structs of all groups travel in the kernel argument segment: 4 KB)
extern "C" {
int  sonde_launch_if_chain_multi(const IfArgs *a, int n_groups, hipStream_t s);
int  sonde_launch_sync_plan_multi(const WinPlanArgs *a, int n_groups, hipStream_t s);
int  sonde_launch_sync_window_fft_multi(const WinFftArgs *a, int n_groups, hipStream_t s);
int  sonde_launch_framesync_multi(const SyncArgs *a, int n_groups, hipStream_t s);
int  sonde_launch_mix_decimate(const MixDecArgs *a, hipStream_t s);   // -1: decimation factor not instantiated
void sonde_launch_dc_update(int n_ch, long long *sums, float2 *avg, float maxcnt, hipStream_t s);
void sonde_launch_dc_update_keep(int n_ch, long long *sums, float2 *avg, float2 *avg_prev, float maxcnt, hipStream_t s);
void sonde_launch_publish_u32(const unsigned *src, unsigned *dst_mapped, hipStream_t s);
// per-channel IQ-DC schedule (restarted channels): cnt += n_samples; a channel that reaches its segment length hands its mean over
// (avg_prev = avg, avg = sums / max, sums = 0, since = 0, cnt = 0, max doubles up to lim); the others get since += nblocks
void sonde_launch_dc_update_pcs(int n_ch, long long *sums, float2 *avg, float2 *avg_prev, uint32_t *cnt, uint32_t *max, uint32_t lim, int32_t *since,
                                uint32_t n_samples, int nblocks, hipStream_t s);
void sonde_launch_md_etable(const double *chan_f0, const float *wtab, int D, int Q, int P, int n_ch, float2 *etab, hipStream_t s);
void sonde_launch_if_chain(const IfArgs *a, hipStream_t s);
void sonde_launch_mix_f32(const MixF32Args *a, hipStream_t s);
void sonde_launch_dc_segments_f32(const float2 *x, long long ch_stride, int n_ch, int n_samples, unsigned dc_cnt0, unsigned dc_max,
                                  double *seg_sums, double *dc_sums, float2 *dc_avg, float2 *dc_seg, int dc_seg_n, hipStream_t s);
void sonde_launch_decimate_f32(const DecF32Args *a, hipStream_t s);
void sonde_launch_dc_update_f64(int n_ch, double *sums, float2 *avg, float maxcnt, hipStream_t s);
void sonde_launch_afc_rotate(const AfcRotArgs *a, hipStream_t s, int n_max);
void sonde_launch_fill_u32(uint32_t *p, uint32_t v, int n, hipStream_t s);
void sonde_launch_audio_chain(const AudioChainArgs *a, hipStream_t s);
// 8-bit unsigned IQ -> the int16 form with identical sample values; n complex samples per channel, n even
void sonde_launch_u8_to_s16(const uint8_t *in, long long in_stride, int16_t *out, long long out_stride, int n_ch, int n_bytes, hipStream_t s);
void sonde_launch_header_corr(const CorrArgs *a, hipStream_t s);
void sonde_launch_sync_plan(const WinPlanArgs *a, hipStream_t s);
void sonde_launch_sync_window_fft(const WinFftArgs *a, hipStream_t s);
void sonde_launch_framesync(const SyncArgs *a, hipStream_t s);
}
#endif
