// sonde_scan_dev.h — structs shared by the scanner kernels (sonde_scan.hip) and the scanner engine (sonde_scan.cpp).
// Scanner = the reference's scan/dft_detect.c: one IF front-end, 4 FM streams, 16 header templates correlated per window.
#ifndef SONDE_SCAN_DEV_H
#define SONDE_SCAN_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SC_N       8192      // N_DFT (dft_detect.c:1207-1211; fixed for IF rates up to ~51 kHz)
#define SC_LOG2N   13
#define SC_NTPL    16        // templates 0..idxIMETafsk (dft_detect.c:172-191)
#define SC_THREADS 1024

struct ScanTpl {              // one header template (rs_hdr[] row + derived sizes, dft_detect.c:1166-1175)
    int   L;                  // samples of the header
    int   hLen;               // header bits
    int   lpfm;               // FM low-pass variant 0/1 (4 kHz / 10 kHz)
    int   stream;             // physical FM stream the template reads
    int   active;             // 0: compiled out in the reference build (NOC34C50, NOIMET1AB) or disabled
    int   is_m10;             // M10/M20 type bytes are sliced behind the header
    int   hdr_off;            // offset of the header bits in ScanCorrArgs.hdrbits
    int   bnd_off;            // offset of the bit-boundary table in ScanCorrArgs.bnd
    float spb;                // samples per bit
    float thres;
    int   herrs;              // headcmp limit (exclusive)
    int   pad;
};

struct ScanItem { int ch; uint32_t pos; };                    // one correlation window: channel, sample_out at the window
struct ScanRes  { int mp; float mv; uint32_t mpos; float dc; int herrs; uint32_t m10; };   // per (window, template)

struct ScanCorrArgs {
    const float *fm;          // [streams][n_ch][ring_len]
    int n_ch, ring_len;
    const ScanItem *items; int n_items;
    ScanTpl tpl[SC_NTPL];
    const float2 *G;          // [SC_NTPL][SC_N] reference-transform spectrum of (FM low-pass x) time-reversed template
    const float2 *WS;         // [2][SC_N] reference-transform spectrum of the FM low-pass taps
    int lpfm_taps;
    const float2 *tws;        // [SC_N-1] stage twiddles of the reference's dft_raw: stage t at 2^t - 1 + j
    int K, opt_dc, opt_iq;
    const uint8_t *hdrbits;
    const int *bnd;           // bit boundaries in samples (float accumulation of the reference tabulated on the host)
    ScanRes *out;             // [n_items][SC_NTPL]
    const struct ScanWork *work; int n_work;      // work list: one workgroup per listed (window, template) pair (nullptr: the full n_items x SC_NTPL grid)
    int N, log2n;             // N_DFT: 0 / SC_N, or 16384 / 32768 with the work-list form and `scratch`
    float2 *scratch;          // N_DFT > SC_N: [n_work][N] transform arrays in global memory
};
struct ScanWork { int item, tpl; };

// Prefilter (k_scan_pre, sonde_scan_pre.hip): every (window, template) is first scored in the time domain on the matrix cores (f16 inputs,
// f32 accumulation) — FM low-pass and header correlation as Toeplitz products, energy under the template from prefix sums.  smax is an UPPER
// bound (up to the f16 rounding, ~1e-4) of |mv| as getCorrDFT would return it: the reference normalises the correlation at its arg-max, smax
// takes the maximum of the normalised value over all positions.  Only pairs with smax > thres - margin go to the exact transform network.
struct ScanPre { float smax; float mv; int mp; uint32_t mpos; float dc; int pad; };      // smax: the maximum over all positions of (score + its rounding bound)
struct ScanPreArgs {
    const float *fm; int n_ch, ring_len;
    const ScanItem *items; int n_items;
    ScanTpl tpl[SC_NTPL];
    const uint16_t *a_match;  // f16 A fragments of the header templates: template j at a_off[j] (halves), nc2[j] steps of 64 lanes x 8 halves
    int a_off[SC_NTPL], nc2[SC_NTPL];
    const uint16_t *a_ws;     // f16 A fragments of the two FM low-passes: [2][nc1][64][8]
    int nc1, taps, ws_pad;    // ws_pad = taps - 1 + front padding of the tap vector, a multiple of 8 (16-byte aligned LDS reads)
    const float *ws_tail;     // [2][taps]: sum of the taps behind tap i (what a constant loses in the filter's first taps-1 outputs)
    int K, opt_dc, opt_iq, lpfm_taps;
    float kap[SC_NTPL];       // kappa_j = 4.02 u ||ws||_1 sqrt(L_j) (u = 2^-11; 0 for FM-audio input, which has no low-pass): the part of the score's rounding bound that scales with
                              // (window maximum) / (rms under the template) — added per position inside the kernel
    ScanPre *out;             // [n_items][SC_NTPL]
    unsigned long long *prof; // SONDE_SP_PROF: shader cycles per phase of the workgroups of template 1 (RS41), [8]; nullptr = off
};
// What k_scan_if needs to take the IQ-DC means off the raw outputs of k_mix_decimate50r as it loads them: y -= mean(window of the block) * E, and for
// the Q-1 outputs behind a change of the mean the correction k_scan_dc_edges tabulated (the blocks of the window before ran under the other mean).
struct ScanFold {
    const float2 *etab; int etab_len; uint32_t e0; // E[ch][etab_len] (double-phase mixer table, k_md_etable); table block of the launch's first block.  etab == nullptr: no fold
    uint32_t m0; int nblocks;                       // the launch's outputs are IF samples m0 .. m0 + nblocks - 1
    const float2 *dc_seg; int dc_seg_n, dc_seg_off, dc_seg_blocks;   // as in MixDecArgs: the mean in effect for the k-th window the launch touches
    const float2 *corr; int edge_n;                 // [n_ch][dc_seg_n][8]: what output i < edge_n = Q-1 of window k gets on top (k_scan_dc_edges)
    const float2 *hist_in; float2 *hist_out; int hist_n;             // [n_ch][hist_n]: the last hist_n folded outputs of the previous / of this launch (the FIR's history)
};
struct ScanIfArgs {
    const float2 *y;          // [n_ch][ring_len] IF-rate IQ
    float *fm;                // [streams][n_ch][ring_len]
    int n_ch, ring_len, n; uint32_t m0;
    int taps, nfilt;          // IF low-pass taps, number of distinct filters (3, or 1 with --bw)
    const float *w;           // [nfilt][taps]
    int filt_stream[3];       // physical stream of filter b
    int raw_stream;           // physical stream of the unfiltered discriminator
    ScanFold fold;            // y holds the raw outputs of k_mix_decimate50r: the IQ-DC means come off as the tile is loaded (fold.etab == nullptr: y is final)
};

struct IqConvArgs {           // --iq: IF-rate IQ in, IQ-DC removed (f32read_csample, dft_detect.c:539-573)
    const int16_t *iq; long long ch_stride; int n_ch, n;
    const float2 *dc_avg; long long *dc_sums;
    float2 *y; int ring_len; uint32_t m0;
};

struct AudioConvArgs {        // FM audio in (f32read_sample, dft_detect.c:505-533): int16, one of nch interleaved channels
    const int16_t *pcm; long long ch_stride; int n_ch, n, nch, sel;
    float *fm; int ring_len; uint32_t m0;
    int f32;                  // samples are float32 (32-bit WAV / `- sr 32`): taken as they are (*s = *f)
};

extern "C" {
void sonde_launch_scan_if(const ScanIfArgs *a, hipStream_t s);
int  sonde_launch_scan_corr(const ScanCorrArgs *a, hipStream_t s);
int  sonde_launch_scan_pre(const ScanPreArgs *a, hipStream_t s);
void sonde_launch_iq_convert(const IqConvArgs *a, hipStream_t s);
void sonde_launch_s16_to_f32(const int16_t *in, long long in_stride, float2 *out, long long out_stride, int n_ch, int n, hipStream_t s);
void sonde_launch_audio_convert(const AudioConvArgs *a, hipStream_t s);
}
#endif
