/*
 * sonde_scan.h — C ABI of the batched sonde-type scanner in libsonde_hip.so.
 *
 * Replaces the reference's scan/dft_detect.c for many channels at once: the same front-end (baseband mixer +
 * decimator, or IF-rate IQ, or FM audio), the same 4 FM streams and 16 header templates, the same decision
 * logic and exit code.  As for the demodulator there is no in-process API in the reference; auto_rx calls the
 * binary (auto_rx/autorx/scan.py:541-547,600,625-639).  host/dft_detect.c keeps that CLI on top of this ABI.
 * Conventions as sonde_hip.h (0 / count on success, negative SONDE_E_* on error).
 */
#ifndef SONDE_SCAN_H
#define SONDE_SCAN_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_SCAN_NTPL 16      /* templates correlated per window: rs_hdr[0..idxIMETafsk], dft_detect.c:172-191 */

/* input forms (option_iq of dft_detect.c:1375-1388) */
#define SONDE_SCAN_AUDIO 0      /* FM audio, real int16 (WAV payload or "- sr 16" without --iq)   */
#define SONDE_SCAN_IFIQ  1      /* --iq: IF-rate IQ, no mixer / decimator                          */
#define SONDE_SCAN_BBIQ  5      /* --IQ fq: baseband IQ, mix by -fq and decimate to the IF rate    */

typedef struct sonde_scan sonde_scan_t;

typedef struct {
    int32_t abi_version;     /* SONDE_ABI_VERSION                                                   */
    int32_t device;
    int32_t n_channels;
    int32_t sample_rate;     /* input rate ("- <sr> <bits>" or the WAV header)                      */
    int32_t bits;            /* 16, 8 (unsigned) or 32 (float32)                                    */
    int32_t iq_mode;         /* SONDE_SCAN_AUDIO / _IFIQ / _BBIQ                                    */
    int32_t opt_dc;          /* --dc  (dft_detect.c:1397)                                           */
    int32_t opt_min;         /* --min (IF 32 kHz, :1398)                                            */
    int32_t opt_cont;        /* -c    (keep scanning after a detection, :1408)                      */
    int32_t opt_d2;          /* -d2   (type must be seen twice, :1419)                              */
    int32_t opt_lband;       /* -L    (wide IF filters, :1401-1407)                                 */
    int32_t audio_channels;  /* interleaved channels of the FM-audio input (1 or 2)                 */
    int32_t audio_select;    /* which of them (--ch2 = 1)                                           */
    int32_t max_chunk;       /* largest n_samples per process call                                  */
    float   bw_khz;          /* --bw k (one IF filter of that bandwidth), 0 = the three defaults    */
    float   ths;             /* --ths x (all templates), 0 = per-type defaults                      */
    float   time_limit;      /* -t seconds, <= 0: none                                              */
    uint32_t disable_mask;   /* bit j: skip template j; 0 = as the reference build (C34C50, IMET1AB off, scan/Makefile:1) */
    int32_t opt_exact;       /* 0: every (window, template) is scored by the matrix-core prefilter, and those within 0.03 of their
                              *    threshold — plus the same template's window before them — by the reference's own transform network;
                              *    detections, scores, positions and exit codes are the same as with 1.
                              * 1: the transform network for every pair (testing tap: per-window parity of every template)  */
    int32_t reserved[3];
} sonde_scan_cfg_t;

/* One printed detection = one stdout line of dft_detect (dft_detect.c:1612-1634). */
typedef struct {
    int32_t  channel;
    int32_t  tpl;            /* template row 0..17                                                   */
    int32_t  tn;             /* type number (exit code magnitude): 2 DFM, 3 RS41, 4 RS92, 5 M10, 6 M20 ... */
    char     type[12];       /* "RS41", "DFM9", "M10", "M20", ...                                    */
    float    score;          /* normalised header correlation, sign = polarity                       */
    uint32_t sample;         /* IF-sample index of the header end ("sample:" of -v)                  */
    float    df;             /* frequency offset / input sample rate (only --dc with IQ input)       */
    float    freq_hz;        /* df * input sample rate                                               */
    uint32_t m10_bytes;      /* M10/M20: the two type bytes sliced behind the header                 */
    int32_t  printed;        /* 0 if -d2 suppressed the line                                         */
} sonde_detection_t;

typedef struct {
    int32_t if_sr, decM, dectaps, lpiq_taps, lpfm_taps;
    int32_t K, N, delay, L2;         /* window step is K-4 IF samples                                */
    int32_t L[SONDE_SCAN_NTPL];      /* header length in samples per template                        */
    int32_t ring_len;
    int32_t reserved[3];
} sonde_scan_info_t;

/* Per-window values of every template (what getCorrDFT/headcmp/frm_M10 return, dft_detect.c:357,866,932):
 * testing tap for parity against the reference's own functions. */
typedef struct {
    int32_t  channel;
    uint32_t pos;                            /* sample_out of the window                             */
    int32_t  mp[SONDE_SCAN_NTPL];            /* getCorrDFT return: peak index, -4 edge, 0 template off */
    float    mv[SONDE_SCAN_NTPL];
    uint32_t mpos[SONDE_SCAN_NTPL];
    float    dc[SONDE_SCAN_NTPL];
    int32_t  herrs[SONDE_SCAN_NTPL];         /* -1: threshold not reached, header not compared; -2: prefilter only — |mv| is at least
                                              *  0.03 below the threshold, mp / mv / mpos are the prefilter's (mv within ~1e-3)        */
    uint32_t m10[SONDE_SCAN_NTPL];
} sonde_scan_window_t;

/* replaces init_buffers() (dft_detect.c:995); fq[c] = --IQ argument per channel (ignored unless SONDE_SCAN_BBIQ) */
int  sonde_scan_create(const sonde_scan_cfg_t *cfg, const double *fq, sonde_scan_t **out);
/* replaces free_buffers() (dft_detect.c:1289) */
void sonde_scan_destroy(sonde_scan_t *s);
int  sonde_scan_info(const sonde_scan_t *s, sonde_scan_info_t *info);

/* Push n_samples per channel (complex int16 pairs for the IQ forms, audio frames for SONDE_SCAN_AUDIO); replaces the
 * `while (f32buf_sample(fp) != EOF)` loop of main (dft_detect.c:1483-1651).  Runs every correlation window that is
 * complete, then the reference's decision logic per channel.  Synchronous.  BBIQ: n_samples % decM == 0.
 * ch_stride == 0: one wideband stream shared by all channels (each mixes its own fq out of it) — the channelizer form. */
int  sonde_scan_process_host(sonde_scan_t *s, const void *h_in, int64_t ch_stride, int32_t n_samples);
int  sonde_scan_process_device(sonde_scan_t *s, const void *d_in, int64_t ch_stride, int32_t n_samples);
/* The scanner's stream waits for everything queued so far on `stream` (a hipStream_t: sonde_chan_stream(), an engine's, the caller's own) —
 * the producer of d_in then needs no host synchronisation in front of sonde_scan_process_device.  (The reference has no such seam: its
 * scanner reads a pipe; this is the device-side equivalent of the pipe's ordering.) */
int  sonde_scan_wait_stream(sonde_scan_t *s, void *stream);

/* End of input: a channel parked in the IMET AFSK check (one more second of samples, dft_detect.c:1533-1607) is decided
 * with the samples that exist, like the reference at EOF. */
int  sonde_scan_finish(sonde_scan_t *s);

/* Detections found since the last fetch, in the order the reference would print them per channel. */
int  sonde_scan_fetch(sonde_scan_t *s, sonde_detection_t *out, int32_t max);
/* 1 once a channel has stopped (detection without -c, -d2 satisfied, or -t exceeded) */
int  sonde_scan_channel_done(const sonde_scan_t *s, int32_t channel);
/* exit code of the reference for this channel so far: header_found * tn, negative for inverted RS41/DFM/RS92
 * (dft_detect.c:1656-1666); the CLI returns it modulo 256 like the reference's `return` from main */
int  sonde_scan_result(const sonde_scan_t *s, int32_t channel, int32_t *code);
/* text line of one detection as dft_detect prints it (dft_detect.c:1612-1634); returns strlen */
int  sonde_scan_line(const sonde_scan_t *s, const sonde_detection_t *d, int verbose, char *buf, size_t buflen);

/* testing taps: windows evaluated by the last process call; FM stream samples still in the ring */
int  sonde_scan_last_windows(const sonde_scan_t *s, sonde_scan_window_t *out, int32_t max);
int  sonde_scan_read_fm(sonde_scan_t *s, int32_t channel, int32_t stream, int64_t first, int32_t count, float *out);
int  sonde_scan_kernel_ms(sonde_scan_t *s, const char *kernel, double *avg_ms, int64_t *launches);
/* testing tap of the prefilter (host only, no GPU): out[i] = sum_u h[u] x[i+u], i < n_out, evaluated from the SAME f16 fragment tables and in the
 * same contraction the matrix-core kernel uses (x beyond n_x reads as 0); returns the number of 16x16x32 steps per output tile */
int  sonde_scan_toeplitz_model(const float *h, int32_t n_taps, const float *x, int32_t n_x, float *out, int32_t n_out);

#ifdef __cplusplus
}
#endif
#endif
