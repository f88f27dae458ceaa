/*
 * sonde_hip.h — C ABI of libsonde_hip.so, the MI355X-native radiosonde IQ demodulation engine.
 *
 * Plain C, plain pointers and sizes.  The reference has no in-process FFI for this path: its
 * "operator API" is (i) the per-sonde CLI contract and (ii) the function seam of
 * demod/mod/demod_mod.h:179-192 under it.  This header is the batched (many
 * channels per call) equivalent of that seam; each entry point names the reference code it replaces.
 * Host programs (host/rs41mod.c ...) keep the CLI contract on top of it; INTEGRATION.md shows the bindings.
 *
 * Threading: an engine is owned by one host thread.  No global state; several engines (one per GPU)
 * may live in one process.  All functions return 0 / a non-negative count on success and a negative
 * SONDE_E_* code on error (the reference convention: negative int, message on stderr by the caller —
 * sonde_strerror() supplies the text).
 */
#ifndef SONDE_HIP_H
#define SONDE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_ABI_VERSION 3          /* 2: sonde_cfg_t.if_tune, sonde_generic_t.slice_baud, sonde_fsk_frame_t.f_est[4]; 3: sonde_dfm_frame_t.rawbits */

/* error codes */
#define SONDE_E_ARG      (-1)   /* bad argument / unsupported option combination          */
#define SONDE_E_NOGPU    (-2)   /* no HIP device / HIP runtime failure (never falls back) */
#define SONDE_E_NOMEM    (-3)
#define SONDE_E_RANGE    (-4)   /* chunk larger than max_chunk / not a multiple of decM   */
#define SONDE_E_OVERFLOW (-5)   /* (reserved; queue overflow is reported by sonde_engine_overflowed()) */

/* sonde types (dsp.hdr / baud / BT / h presets of the reference callers) */
#define SONDE_RS41  41          /* rs41mod.c:2812-2836: 4800 Bd, BT 0.5, h 0.6, 64-bit header, thres 0.7, hdmax 4 */
#define SONDE_FRONTEND 0        /* no sonde: front-end only = the reference's demod/mod/iq_dec.c (mixer, decimator, optional
                                 * IF low-pass / FM discriminator / FM low-pass); results are read with sonde_engine_read_tap */
#define SONDE_DFM09  9          /* dfm09mod.c:1309,1560-1582: 2500 Bd Manchester, BT 0.5, h 1.8, 32-symbol raw header,
                                 * thres 0.65, hdmax 2, lpIQ 12 kHz, 8 x 280-bit frames sliced per header hit       */
#define SONDE_M10    10         /* m10mod.c:55,76,1370-1390,1436-1510: 9615 Bd Manchester, BT 1.8, h 0.9, 32-symbol raw header compared
                                 * per symbol, thres 0.76, hdmax 2; 968 differentially coded bits per frame, then the rest of the
                                 * second (5 x 808 bits) is skipped; either polarity */
#define SONDE_M20    20         /* m20mod.c:60,86,1034-1040,1238-1251,1321-1365: as M10 with 9600 Bd and up to 64 aux bytes (1320 bits) */

#define SONDE_MIXED  100        /* cfg.sonde_type of sonde_engine_create_mixed: the type is a property of the channel (its group), not of the engine */
#define SONDE_GENERIC 99        /* any other 2-FSK sonde of the reference's demod/mod family, described by a sonde_generic_t given to sonde_engine_create_generic;
                                 * header hits + soft bits only (sonde_engine_fetch_hits), framing stays with the caller */

/* input forms (dsp.opt_iq of demod_mod.h:62; rs41mod.c:2674-2687,2786-2803) */
#define SONDE_IN_IQ    0        /* baseband IQ, mixed by -fq and decimated to the IF rate (opt_iq = 5)        */
#define SONDE_IN_AUDIO 1        /* FM-demodulated audio, one real sample per input frame (opt_iq = 0)         */
#define SONDE_IN_IFIQ0 2        /* --iq0: IF-rate IQ (no mixer / decimator), FM discriminator sliced (opt_iq = 1) */
#define SONDE_IN_IFIQ2 3        /* --iq2: IF-rate IQ, two-tone correlator sliced over whole bits (opt_iq = 2) */
#define SONDE_IN_IFIQ3 4        /* --iq3: as --iq2 with the centre window of --IQ (opt_iq = 3, rs41mod.c:2921) */

/* opt_lp bits, as demod_mod.h:12-14 */
#define SONDE_LP_IQ 1
#define SONDE_LP_FM 2

/* stream taps for parity testing (what the reference keeps in rot_iqbuf / fm_buffer / bufs,
 * demod_mod.c:775,849,852; CORR is the un-normalised matched-filter output of getCorrDFT :190-192) */
#define SONDE_TAP_DECIM 0       /* decimated IQ before the IF low-pass  (cf32) */
#define SONDE_TAP_IFIQ  1       /* IF-filtered IQ = rot_iqbuf          (cf32) */
#define SONDE_TAP_FM    2       /* fm_buffer                            (f32)  */
#define SONDE_TAP_BUFS  3       /* bufs (sliced stream)                 (f32)  */
#define SONDE_TAP_CORR  4       /* header correlation per end-sample    (f32)  */

typedef struct sonde_engine sonde_engine_t;

/* Engine configuration: the fields the reference callers set in dsp_t (rs41mod.c:2816-2836) plus batching. */
typedef struct {
    int32_t abi_version;     /* SONDE_ABI_VERSION                                             */
    int32_t device;          /* HIP device ordinal                                            */
    int32_t n_channels;      /* independent channels (one reference process each)             */
    int32_t sample_rate;     /* input rate of every channel ("- <sr> <bs>" argv)              */
    int32_t bits;            /* 16 (cs16 / s16), 8 (unsigned: rtl_sdr cu8, 8-bit WAV) or 32 (float32: cf32 / float WAV) */
    int32_t sonde_type;      /* SONDE_RS41, SONDE_DFM09, SONDE_M10, SONDE_M20, SONDE_FRONTEND, SONDE_GENERIC */
    int32_t opt_lp;          /* SONDE_LP_IQ (--lpIQ) | SONDE_LP_FM (--lpFM)                    */
    int32_t opt_dc;          /* --dc: zero-mean correlation windows, header dc, AFC feedback; with input = SONDE_IN_IQ the engine turns SONDE_LP_FM on as every decoder does (rs41mod.c:2747) */
    int32_t opt_min;         /* --min (IF 32 kHz)                                             */
    int32_t lpiq_bw;         /* --lpbw in Hz, 0 = type default (7400 for RS41)                */
    int32_t ecc_level;       /* 0 none, 1 --ecc, 2 --ecc2                                     */
    float   thres;           /* --ths, header score threshold (0 = type default)              */
    int32_t max_chunk;       /* largest n_samples per process call (per channel)              */
    int32_t max_frames;      /* frame queue capacity between two fetches (0 = 4*n_channels)   */
    int32_t keep_soft;       /* keep per-frame soft bits (soft-bit fetch call) and the IFIQ / FM tap streams; 2: also the second soft bit (fetch_soft1) */
    int32_t pipeline;        /* 1: IF-rate kernels on a second HIP stream so that sonde_engine_fetch_frames_lagged(lag=1)
                              * overlaps them with the next call's decimator; 0: one in-order stream             */
    int32_t input;           /* SONDE_IN_IQ (--IQ fq, cs16), SONDE_IN_AUDIO (FM audio: WAV payload, real int16; dsp.opt_iq = 0,
                              * the reference's CPU-runnable configuration) or SONDE_IN_IFIQ0/2/3 (--iq0/2/3)     */
    int32_t audio_channels;  /* SONDE_IN_AUDIO: interleaved channels per frame (1 or 2) and which one (--ch2 = 1) */
    int32_t audio_select;
    int32_t if_rate;         /* SONDE_FRONTEND: designated IF rate in Hz (iq_dec --IFbw k -> 1000 k, default 48000;
                              * iq_dec.c:632-651); 0 elsewhere (48000, or 32000 with opt_min)                     */
    int32_t opt_iqdc;        /* --iqdc: running-mean IQ-DC removal for SONDE_IN_IFIQ* (f32read_csample, demod_mod.c:444-458);
                              * SONDE_IN_IQ always removes it (f32read_cblock :492)                              */
    int32_t opt_inv;         /* -i: inverted polarity expected — headers with a negative score are taken, bits flipped
                              * (rs41mod.c:2887-2891,2933-2937; dfm09mod.c:1642-1645,1702-1705)                  */
    int32_t opt_nolut;       /* --noLUT (SONDE_IN_IQ): mixer phasor from the exact fq and the absolute sample index in double instead
                              * of the periodic float-phase table of the snapped fq (demod_mod.c:738-742); not with opt_dc   */
    int32_t m10_noskip;      /* SONDE_M10 / M20 with -vvv: do not drop the rest of the second after a frame (m10mod.c:1493) */
    int32_t opt_auto;        /* --auto: a header of the opposite polarity flips the channel's polarity instead of being skipped */
    int32_t if_tune;         /* SONDE_IN_IFIQ* with float32 samples (bits = 32): rotate channel c by -fq[c] (cycles per IF sample, exact double phase
                              * from the channel's own sample count) before the IF low-pass — the fine tuning a channelizer output needs (the sonde sits
                              * anywhere inside its channel); 0 = the reference's --iq0/2/3, which ignore fq                               */
} sonde_cfg_t;

/* One decoded frame = what rs41mod's print_frame() sees (rs41mod.c:2472-2553). */
typedef struct {
    int32_t  channel;
    int32_t  len;            /* 320 or 518 (frametype, rs41mod.c:407-415)                      */
    int32_t  ecc;            /* rs41_ecc() return: >=0 corrected symbols, -1/-2/-3 failed cw   */
    uint32_t mv_pos;         /* IF-sample index of the last header sample (dsp.mv_pos)         */
    float    mv;             /* header correlation score (dsp.mv)                              */
    int32_t  nbytes;         /* bytes actually sliced (518 unless stream ended)                */
    uint8_t  frame[518];     /* de-whitened, ECC-corrected frame bytes                         */
    uint8_t  pad[2];
} sonde_frame_t;

/* One DFM frame = what dfm09mod's print_frame() sees after de-interleave + Hamming(8,4) (dfm09mod.c:1153-1236). */
typedef struct {
    int32_t  channel;
    int32_t  frame_in_hit;   /* 0..7: frames are sliced back-to-back after one header hit (nfrms = 8)   */
    int32_t  ecc[3];         /* hamming() return per block conf/dat1/dat2: 0 ok, >0 fixed-codeword mask, <0 uncorrectable */
    uint32_t mv_pos;
    float    mv;
    uint8_t  conf[7], dat1[13], dat2[13];   /* data nibbles */
    uint8_t  pad[3];
    float    frm_count;      /* gpx._frmcnt: mv_pos / (2 sps 280) + frame_in_hit, or headers seen * 8 + frame_in_hit for
                              * soft / hard bit input (dfm09mod.c:1658-1663) — the time stamp of the telemetry decoder */
    int32_t  inv;            /* polarity in effect for this frame (gpx.option.inv after -i / --auto)        */
    uint8_t  rawbits[35];    /* the 280 hard bits of the frame as sliced (bit i = rawbits[i >> 3] >> (i & 7)), before the Hamming decoder —
                              * what `--rawecc` prints (dfm09mod.c:1177-1196); bits 0..15 of frame 0 of a hit are not sliced (the header
                              * was consumed by the correlator): 0 here, the telemetry decoder keeps the previous frame's like the reference */
    uint8_t  pad2;
} sonde_dfm_frame_t;

/* Derived constants of init_buffers() (demod_mod.c:1208-1474), for callers and tests. */
typedef struct {
    int32_t if_sr, decM, dectaps, lut_len, lpiq_taps, lpfm_taps;
    int32_t L, M, K, N, delay;
    float   sps;
    int32_t ring_len;        /* per-channel IF ring length (samples)                           */
    int32_t reserved[3];
} sonde_info_t;

/* replaces init_buffers() (demod_mod.c:1208) for n_channels channels; fq[c] is the --IQ <fq> argument
 * of channel c (-0.5..0.5, rs41mod.c:2678-2687). */
int  sonde_engine_create(const sonde_cfg_t *cfg, const double *fq, sonde_engine_t **out);
/* cfg->sonde_type == SONDE_GENERIC: the fields a decoder of the reference puts into dsp_t (e.g. rs92mod.c:1924-1939: br, symlen, symhd, hdr,
 * BT, h, lpIQ_bw, lpFM_bw), the find_header() arguments hdmax / bitofs, and how its bit loop consumes a hit: nbits soft bits (<= 8192), then
 * — if skip_bits > nbits — bits up to skip_bits dropped before the header search resumes.  l_win: the centre window `l` it passes to
 * read_softbit*() for opt_iq > 2 (0 = whole bits).  Threshold: cfg->thres (0 = 0.7).  Needs cfg->keep_soft; results through sonde_engine_fetch_hits(). */
typedef struct {
    char    header[68];      /* '0' / '1' characters, 8..64, NUL-terminated                     */
    float   baud, bt, h;
    int32_t symlen, symhd;   /* symbols per bit in the frame / in the header (1 or 2)           */
    int32_t hdmax, bitofs;
    int32_t nbits, skip_bits;
    float   l_win;
    int32_t lpiq_bw, lpfm_bw;   /* Hz */
    float   slice_baud;      /* > 0: dsp.br the decoder sets AFTER init_buffers() (lms6Xmod.c:1343-1347): bit clock / slicers only */
    int32_t reserved[3];
} sonde_generic_t;
int  sonde_engine_create_generic(const sonde_cfg_t *cfg, const double *fq, const sonde_generic_t *gen, sonde_engine_t **out);
/* With a preset type in cfg (SONDE_RS41 / DFM09 / M10 / M20) a non-NULL `gen` carries one thing only: gen->baud replaces the preset's symbol
 * rate before the design — the decoders' --br option (dfm09mod.c:1436-1443,1590-1594; m20mod.c:1082-1089).  Everything else of `gen` is ignored. */
/* Mixed-type engine (BASELINE configs[3] / [4]: RS41, DFM09, M10 ... channels side by side on one GPU).  Nothing in front of the IF rate depends on the sonde
 * type — IF_sr, decM and the decimator taps follow from the sample rate and --min alone (demod_mod.c:1222-1249), the mixer table from fq (:1262-1296), the
 * IQ-DC from the samples (:495-504) — so ONE decimator launch per call serves every channel, whatever it carries; the IF-rate stages (IF low-pass, tone
 * correlator, header search, frame sync, block code) run per group with the group's preset, each group on streams of its own behind that launch.
 * A group = what one decoder command line of the reference fixes: the decoder (sonde_type: SONDE_RS41 / DFM09 / M10 / M20) and its options.
 * cfg: sonde_type = SONDE_MIXED; device, n_channels, sample_rate, bits (16 / 8), opt_lp, opt_min, max_chunk, max_frames (shared out among the groups by
 * channel count), input = SONDE_IN_IQ are common to all channels; ecc_level / thres / lpiq_bw / opt_inv / opt_auto / m10_noskip come from the group.
 * Not here: --dc, --noLUT, float32 input, keep_soft (SONDE_E_ARG).  group_of_channel[c] = index into groups[] of channel c; channel numbers in every
 * result are the caller's.  Results: sonde_engine_fetch_frames[_lagged] / _finish return the frames of the RS41 channels, sonde_engine_fetch_dfm[_lagged] /
 * _m10[_lagged] / _m20 those of their types; process / sync / set_summary[_snapshots] / set_device_ecc / overflowed / samples_to_dc_boundary / read_tap /
 * finish_channel work as on a single-type engine. */
typedef struct {
    int32_t sonde_type;      /* SONDE_RS41, SONDE_DFM09, SONDE_M10, SONDE_M20 */
    int32_t ecc_level;       /* as sonde_cfg_t.ecc_level */
    int32_t lpiq_bw, opt_inv, opt_auto, m10_noskip;
    float   thres;           /* 0 = type default */
    int32_t reserved;
} sonde_group_t;
int  sonde_engine_create_mixed(const sonde_cfg_t *cfg, const double *fq, const sonde_group_t *groups, int32_t n_groups, const int32_t *group_of_channel, sonde_engine_t **out);
/* the sonde type and the derived constants (init_buffers) of the group channel `channel` belongs to; a single-type engine answers with its own */
int  sonde_engine_group_info(const sonde_engine_t *e, int32_t channel, int32_t *sonde_type, sonde_info_t *info);
/* replaces free_buffers() (demod_mod.c:1476) */
void sonde_engine_destroy(sonde_engine_t *e);
int  sonde_engine_info(const sonde_engine_t *e, sonde_info_t *info);

/* Push n_samples new complex samples per channel.  Replaces the pull loop
 * find_header()/read_softbit2p() -> f32buf_sample() -> f32read_cblock() (demod_mod.c:1533,1087,722,463).
 * iq: interleaved I,Q little-endian int16, channel c at iq + 2*c*ch_stride (ch_stride in complex samples).
 * ch_stride == 0: ONE wideband stream shared by all channels — each channel mixes its own fq out of it (the channelizer
 * form: one SDR stream, N sondes, no N processes).  n_samples must be a multiple of decM and <= max_chunk.  The *_device form takes a device pointer and
 * only enqueues work on the engine's HIP stream; the *_host form copies first (PCIe-inclusive). */
int  sonde_engine_process_device(sonde_engine_t *e, const void *d_iq, int64_t ch_stride, int32_t n_samples);
int  sonde_engine_process_host(sonde_engine_t *e, const void *h_iq, int64_t ch_stride, int32_t n_samples);
/* Samples until the next IQ-DC segment boundary of the reference (demod_mod.c:498-504: 1/32 s doubling up to 1 s).
 * A process call that ends exactly there (and later calls of one full segment) needs a single decimator launch;
 * any other chunking is split internally at the boundary with identical results. */
int64_t sonde_engine_samples_to_dc_boundary(const sonde_engine_t *e);
/* wait for all enqueued work */
int  sonde_engine_sync(sonde_engine_t *e);

/* Collect frames completed so far (syncs).  The RS(255,231) passes of rs41_ecc() (rs41mod.c:1703-1769) have run on the device for
 * whole frames (k_framesync: syndromes, and for non-zero ones the Euclid / Chien / Forney decoder of bch_ecc_mod.c:877-960 on a wavefront
 * per codeword, the 2nd pass of --ecc2 included).  Decoded on the host when they are fetched: a frame cut short by the end of the stream
 * (its missing bytes come from the previous frame, rs41mod.c:2479-2490), and the damaged frames the end-of-stream frame syncs of
 * sonde_engine_finish / _finish_channel emit (those launches have no decoder work list) — sonde_engine_host_ecc_frames() counts both.
 * Returns the number of frames written (<= max).  Frames of one channel come in stream order; the order between
 * channels that completed a frame in the same process call is unspecified (sonde_frame_t.channel tells them apart). */
/* Per-channel detection summary: the ONLY data that crosses GPUs when channels are sharded over a node — 32 bytes per
 * channel, written by the frame-sync kernel at every frame it emits and left in device memory, so the all_gather (RCCL over xGMI) runs
 * on the device buffer without a host round trip.  sample_pos = IF-rate sample index of the header's last sample (64 bit).  frames /
 * frames_clean are cumulative: frames emitted and, of those, frames whose RS syndromes were all zero (nothing to
 * correct; RS41 only).  reference: the per-sonde process of auto_rx prints this information per frame (rs41mod.c:2530-2545). */
typedef struct {
    uint32_t channel_id;     /* global channel number: summary_base + channel within the engine                  */
    uint8_t  type;           /* SONDE_RS41 / SONDE_DFM09 / ... (cfg.sonde_type)                                   */
    uint8_t  inverted;       /* header found with negative polarity                                               */
    uint16_t reserved;
    float    score;          /* header correlation of the last frame (mv, demod_mod.c:222)                        */
    float    freq_offset_hz; /* --dc: accumulated AFC offset Df; else 0                                          */
    uint64_t sample_pos;
    uint32_t frames;
    uint32_t frames_clean;
} sonde_summary_t;
/* The engine writes its channels' records to `d_summary` (device memory, n_channels records, zeroed by the caller) from now on;
 * channel_id starts at `channel_base`.  NULL switches the records off. */
int  sonde_engine_set_summary(sonde_engine_t *e, void *d_summary, uint32_t channel_base);
/* Pipelined callers (frames fetched with a lag, the next call already running): `d_snap` = device memory for 2 x n_channels records.
 * Every process call ends with a device-to-device copy of the live records into half (call number & 1), ordered behind the call's frame
 * sync on the engine's stream; once sonde_engine_fetch_frames_lagged(.., lag = 1) has returned, the half of the call BEFORE the latest one
 * is complete and stable — that half is what an all_gather may read while the latest call still runs.  Returns the index (0 / 1) of the
 * half the NEXT process call will fill; NULL switches the copies off. */
int  sonde_engine_set_summary_snapshots(sonde_engine_t *e, void *d_snap);

int  sonde_engine_fetch_frames(sonde_engine_t *e, sonde_frame_t *out, int32_t max);
/* 1 if the device-side frame queue (cfg.max_frames) overflowed since the last call of this function — the oldest frames were then
 * overwritten before a fetch could read them; the fetch functions themselves return the number of frames they delivered. */
int  sonde_engine_overflowed(sonde_engine_t *e);
/* RS41 frames whose Reed-Solomon decoder ran on the host in sonde_engine_fetch_frames* so far: frames cut short by the end of the stream,
 * damaged frames emitted by the end-of-stream frame syncs (finish / finish_channel), and every frame with non-zero syndromes when SONDE_HOST_ECC=1 is set in the environment (the A/B switch).  Whole frames are decoded by
 * k_framesync on the device (rs41_ecc, rs41mod.c:1703-1769; rs_decode_ErrEra, bch_ecc_mod.c:877-960). */
long long sonde_engine_host_ecc_frames(sonde_engine_t *e);
/* on = 0: frames of the following calls leave k_framesync with their first-pass syndromes only and are decoded on the host when fetched (the
 * round-3 arrangement; A/B measurements and tests); on = 1 (default): device decoder.  DFM09 / M10 engines: the same switch for their block codes
 * (Hamming(8,4) of the sliced frames / differential decoding + checkM10): on the device behind the frame sync, or on the host inside the fetch; there the
 * switch takes effect between records only — everything queued must have been fetched (SONDE_E_ARG otherwise). */
int  sonde_engine_set_device_ecc(sonde_engine_t *e, int32_t on);
/* Pipelined variant: return only the frames of process calls issued at least `lag` calls ago and wait only for those.
 * With lag = 1 the IF-rate kernels of call k (stream B) overlap the decimator of call k+1 (stream A); lag = 0 is
 * sonde_engine_fetch_frames().  Frames are never lost: what is not returned stays queued. */
int  sonde_engine_fetch_frames_lagged(sonde_engine_t *e, sonde_frame_t *out, int32_t max, int32_t lag);

/* Function-level seam: one header hit the way the reference's own main() sees it between find_header() and
 * read_softbit2p() (demod_mod.h:179-192) — score, position and how many soft bits were sliced behind it.  Any sonde type; needs
 * cfg.keep_soft.  The soft bits (hsbit_t.sb, in the polarity in effect) of the hits returned by the last call come from
 * sonde_engine_fetch_soft().  No ECC / framing is run.  host/seam/demod_mod_hip.c builds the reference's pull API on this. */
typedef struct {
    int32_t  channel;
    int32_t  nbits;          /* soft bits sliced for this hit (type's frame length unless the stream ended) */
    uint32_t mv_pos;
    float    mv;             /* negative: header of inverted polarity (the stored soft bits are already flipped) */
} sonde_hit_t;
int  sonde_engine_fetch_hits(sonde_engine_t *e, sonde_hit_t *out, int32_t max, int32_t finish);
/* find_header()'s hdmax / bitofs arguments (accepted header bit errors; sample offset of the bit windows, e.g. the decoders' -d <shift>)
 * instead of the sonde type's defaults; before the first process call */
int  sonde_engine_set_sync(sonde_engine_t *e, int32_t hdmax, int32_t bitofs);
/* cfg.keep_soft == 2: the second soft bit of read_softbit2p() (hsbit1: the same bit sums taken one IF sample earlier, demod_mod.c:1120,1145 —
 * what --ecc3 adds to the first before slicing, rs41mod.c:2925) of the hits returned by the last sonde_engine_fetch_hits() */
int  sonde_engine_fetch_soft1(sonde_engine_t *e, float *soft, int32_t max_frames);
/* find_header()'s threshold argument (demod_mod.c:1533) for the following process calls */
int  sonde_engine_set_threshold(sonde_engine_t *e, float thres);

/* DFM engines: frames completed so far (syncs).  The hits are sliced into frames and Hamming-decoded (dfm09mod.c:240-345) on the device behind the frame sync
 * (k_dfm_hits; the host decode remains as the A/B path, sonde_engine_set_device_ecc(e, 0)); only decoded frames come to the host, the soft bits on request.  cfg.ecc_level 0/1/2
 * = none / --ecc / --ecc2 (soft 2-bit pass).  finish != 0: end of input, also emits the complete frames of a hit
 * in progress (a partial frame is dropped like dfm09mod.c:1713). */
int  sonde_engine_fetch_dfm(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t finish);
/* pipelined form (as sonde_engine_fetch_frames_lagged): only the frames of process calls issued at least `lag` calls ago, waiting only for those */
int  sonde_engine_fetch_dfm_lagged(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t lag);
/* One M10 / M10+ / M2K2 frame = what m10mod's print_frame() sees after differential decoding (m10mod.c:1049-1070,1484). */
typedef struct {
    int32_t  channel;
    int32_t  nbits;          /* bits sliced (968 unless the stream ended inside the frame)                  */
    int32_t  len;            /* 101 + aux length (frame byte 0 - 0x64 when that is 1..20)                   */
    int32_t  cs_ok;          /* transmitted checksum == computed one                                        */
    uint32_t cs_calc;        /* checkM10() over len - 2 bytes (m10mod.c:594-628)                            */
    uint32_t mv_pos;
    float    mv;
    uint8_t  frame[124];     /* 101 + 20 bytes, big-endian bits                                             */
} sonde_m10_frame_t;
/* SONDE_M10 engines: frames completed so far; finish != 0 = end of input (a frame in progress is emitted with the bits that
 * exist, m10mod.c:1486-1490). */
int  sonde_engine_fetch_m10(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t finish);
int  sonde_engine_fetch_m10_lagged(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t lag);
/* m10mod --chk3 (m10mod.c:1233,1476-1479; IQ input forms): every bit is re-decided from both soft values of read_softbit2p, (sb + 0.25 sb1) >= 0.
 * Needs cfg.keep_soft = 2 (the engine then keeps the second soft value per bit); SONDE_E_ARG otherwise. */
int  sonde_engine_set_m10_chk3(sonde_engine_t *e, int32_t on);
/* Raw text line of `m10mod -r [-v]` (m10mod.c:1112-1123): hex bytes, with verbose " # <checksum> [OK]|[NO]"; buf >= 2 * len + 96 (340 at most) */
#define SONDE_M10_COLOR 0x100    /* or'ed into `verbose`: -c, ANSI colours around the fields (m10mod.c:1078-1110; M10 / M10+ frames only); buf >= 4096 */
int  sonde_m10_rawline(const sonde_m10_frame_t *f, int verbose, char *buf, size_t buflen);

/* One M20 frame (m20mod.c:870-911): frame byte 0 = length; checksum over the first len-1 bytes; older firmware (< 0x07) also carries a
 * checksum of the essential block (bytes 0x02..0x17). */
typedef struct {
    int32_t  channel;
    int32_t  nbits;          /* bits sliced (1320 unless the stream ended inside the frame)                 */
    int32_t  len;            /* bytes printed: frame[0] + 1, at most 0x45 + 64 + 1                          */
    int32_t  cs_ok;
    uint32_t cs_calc;
    int32_t  blk_ok;         /* 1 block checksum good, -1 transmitted as 0, 0 bad                           */
    int32_t  fw;             /* firmware byte (0 when implausible); the block verdict is printed for fw < 7 */
    uint32_t mv_pos;
    float    mv;
    uint8_t  frame[172];
} sonde_m20_frame_t;
int  sonde_engine_fetch_m20(sonde_engine_t *e, sonde_m20_frame_t *out, int32_t max, int32_t finish);
/* Fill len / cs_ok / cs_calc (/ blk_ok / fw) from f->frame: what print_frame() derives before printing (m10mod.c:1049-1070, m20mod.c:875-907).
 * For frames that do not come from an engine or soft-symbol framer, e.g. --rawhex input. */
int  sonde_m10_frame_finish(sonde_m10_frame_t *f);
int  sonde_m20_frame_finish(sonde_m20_frame_t *f);
/* Raw text line of `m20mod -r [-v]` (m20mod.c:959-973); buf >= 2 * len + 96 (400 is enough), 4096 with SONDE_M20_COLOR */
#define SONDE_M20_COLOR 0x100    /* or'ed into `verbose`: -c, ANSI colours around the fields of the raw line (m20mod.c:918-958) */
int  sonde_m20_rawline(const sonde_m20_frame_t *f, int verbose, char *buf, size_t buflen);

/* Raw text line of `dfm09mod -r [--ecc]` (dfm09mod.c:1198-1236); returns strlen. buf >= 96 bytes */
int  sonde_dfm_rawline(const sonde_dfm_frame_t *f, int ecc_level, char *buf, size_t buflen);

/* End of input (stdin EOF of the reference): emit the frame each channel was in the middle of, with the bits that
 * exist (rs41mod.c:2931 breaks the bit loop on EOF and still calls print_frame :2965), then fetch as above. */
int  sonde_engine_finish(sonde_engine_t *e, sonde_frame_t *out, int32_t max);
/* Channels that come and go in a running engine (the resident broker: one channel per decoder process).
 * finish_channel: end of ONE channel's stream — the frame in progress on it is emitted with the bits that exist, like sonde_engine_finish()
 * does for all channels (rs41mod.c:2931,2965); fetch as usual afterwards.
 * restart_channel: a new stream starts on the channel with the next samples fed.  Everything the channel has seen is forgotten (its rings read
 * as silence, sync state as created) and header positions count from here, so the channel behaves like channel 0 of a fresh engine.
 * Base-rate engines (`--IQ fq`, int16 / uint8 input through the mixer table) give the channel its own sample clock as well: the mixer table
 * phase, the IQ-DC mean and its segment schedule (75000 * 2^k samples, demod_mod.c:495-504) and the decimator history start over; from then on
 * process calls are cut at every channel's own segment edges (sonde_engine_samples_to_dc_boundary() = the nearest one).
 * Not with --dc / --iqdc / --noLUT / float32 base-rate input / pipeline: SONDE_E_ARG. */
int  sonde_engine_finish_channel(sonde_engine_t *e, int32_t channel);
/* A new carrier for one channel, normally together with restart_channel when the channel is given to another signal of the stream.
 * cfg.if_tune engines: fine-tuning offset fq in cycles per IF sample.  Base-rate engines (`--IQ fq`): fq in cycles per input sample, snapped to
 * the mixer table's raster like the --IQ argument itself (demod_mod.c:1265-1288); the table period must not change (always true: it depends on
 * the sample rate only). */
int  sonde_engine_tune_channel(sonde_engine_t *e, int32_t channel, double fq);
int  sonde_engine_restart_channel(sonde_engine_t *e, int32_t channel);
/* soft bits (hsbit_t.sb of read_softbit2p) of the frames returned by the last fetch; soft: [n][4080] */
int  sonde_engine_fetch_soft(sonde_engine_t *e, float *soft, int32_t max_frames);

/* Testing tap: copy `count` IF-rate samples starting at absolute IF index `first` of one channel
 * (must still be inside the ring).  out: count floats (x2 for the cf32 taps). */
int  sonde_engine_read_tap(sonde_engine_t *e, int32_t channel, int32_t tap, int64_t first, int32_t count, float *out);

/* HIP stream the engine enqueues on (hipStream_t), for event timing by the caller */
void *sonde_engine_stream(sonde_engine_t *e);
/* average GPU time (ms) of the named kernel since the last reset, measured with HIP events on the
 * engine stream when profiling is enabled; names: "mix_decimate","if_chain","header_corr","framesync".
 * enable = 1 brackets only the dominant kernel (k_mix_decimate), 2 every kernel, 0 none */
int  sonde_engine_profile(sonde_engine_t *e, int enable);
int  sonde_engine_kernel_ms(sonde_engine_t *e, const char *kernel, double *avg_ms, int64_t *launches);

/* Soft-bit input (`rs41mod --softin [-i]` behind `fsk_demod -s`, decode.py:901-909): float32 soft bits in, frames out.
 * find_softbinhead / corr_softhdb (demod_mod.c:1692-1762) + the bit loop of rs41mod.c:2893-2968; host side (bit-rate work).
 * invert_stream = --softinv (f32soft_read inv), opt_inv = -i (gpx.option.inv), opt_auto = --auto. */
typedef struct sonde_softin sonde_softin_t;
int  sonde_softin_create(int32_t sonde_type, int32_t ecc_level, int32_t invert_stream, int32_t opt_inv, int32_t opt_auto, sonde_softin_t **out);
void sonde_softin_destroy(sonde_softin_t *s);
int  sonde_softin_push(sonde_softin_t *s, const float *soft, int32_t n);
/* --bin: one byte per hard bit (LSB used; `fsk_demod` without -s): find_binhead / cmp_hdb (demod_mod.c:1639-1690), header
 * accepted at <= 3 (RS41) / 2 (DFM) bit errors in either polarity; then the same framers */
int  sonde_softin_push_bits(sonde_softin_t *s, const uint8_t *bits, int32_t n);
/* --rawhex / --xorhex (rs41mod.c:2976-3002): one frame given as bytes (xorhex: still whitened), handed to print_frame() */
int  sonde_softin_push_frame(sonde_softin_t *s, const uint8_t *bytes, int32_t len, int32_t xorhex);
/* the same from the text line itself: cut at 2 * 518 characters and at the first blank, lines of 0x3D + 10 bytes or less ignored, pairs read with
 * sscanf "%2hhx" — a pair that is not hex keeps the previous byte (after its de-whitening with --xorhex), across lines too (:2980-2999) */
int  sonde_softin_push_hexline(sonde_softin_t *s, const char *line, int32_t xorhex);
int  sonde_softin_finish(sonde_softin_t *s);               /* EOF: emit the frame in progress (rs41mod.c:2931,2965) */
int  sonde_softin_fetch(sonde_softin_t *s, sonde_frame_t *out, int32_t max);
/* ecc_level 3 / 4 (rs41mod --softin --ecc3): frames come out uncorrected; the soft values of the frames of the last fetch
 * ([n][4080], nbits[i] of them, inv[i] = polarity in effect) go to sonde_rs41_dec_ecc() (sonde_rs41.h) with soft1 = NULL. */
int  sonde_softin_fetch_soft(sonde_softin_t *s, float *soft, int32_t *nbits, int32_t *inv, int32_t max);
/* SONDE_DFM09 framers (dfm09mod --softin, dfm09mod.c:1604-1720: two soft symbols per bit, 8 frames per header hit) */
int  sonde_softin_fetch_dfm(sonde_softin_t *s, sonde_dfm_frame_t *out, int32_t max);
/* dfm09mod --rawhex (dfm09mod.c:1730-1787): the text `--rawecc` wrote — `+|-<frame count>  hex nibbles` per line, blanks skipped, other
 * characters dropped, 66 nibbles = the 264 bits behind the header, LSB first — back into frames (hard bits, soft = +-1) */
int  sonde_softin_push_dfm_rawhex(sonde_softin_t *s, const char *text, int32_t n);
/* SONDE_M10 / SONDE_M20 framers (m10mod / m20mod --softin, m10mod.c:1405-1510: header threshold 0.8, two soft symbols per bit,
 * differential decoding, the rest of the second dropped) */
int  sonde_softin_fetch_m10(sonde_softin_t *s, sonde_m10_frame_t *out, int32_t max);
int  sonde_softin_fetch_m20(sonde_softin_t *s, sonde_m20_frame_t *out, int32_t max);
/* -vvv: the reference then searches the next header right behind a frame instead of dropping the rest of the second */
int  sonde_softin_set_m10_skip(sonde_softin_t *s, int32_t skip);

/* Raw text line of `rs41mod -r` for one frame (rs41mod.c:2530-2545); returns strlen. buf >= 1100 bytes */
int  sonde_rs41_rawline(const sonde_frame_t *f, char *buf, size_t buflen);

/* RS(255,231) codec of bch_ecc_mod.c (rs_encode :860, rs_decode :962) — exposed for tests/tools */
int  sonde_rs255_encode(uint8_t cw[255]);
int  sonde_rs255_decode(uint8_t cw[255]);
/* rs41_ecc() (rs41mod.c:1703-1769, ecc level 1 = --ecc, 2 = --ecc2) over n de-whitened RS41 frames of 518 bytes each ON THE DEVICE, one
 * workgroup per frame — the decoder k_framesync runs behind its slicer (syndromes on the workgroup's four wavefronts, rs_decode_ErrEra of
 * bch_ecc_mod.c:877-960 with no erasures on one wavefront per codeword, the 2nd pass with the known block ids).  Host pointers; frames are
 * repaired in place exactly as the reference leaves gpx->frame (bytes from flen[i] on count as zero, :1727).  ecc[i] = rs41_ecc's value;
 * codes (nullable) = [n][2] the two rs_decode() values of the last pass; synd (nullable) = [n][48] first-pass syndromes.
 * Returns 0 or a SONDE_E_* code (SONDE_E_NOGPU without a device: there is no CPU fallback). */
int  sonde_rs41_ecc_device(uint8_t *frames, const int32_t *flen, int32_t n, int32_t level, int32_t *ecc, int32_t *codes, uint8_t *synd);
/* CRC-16/CCITT-FALSE of rs41mod.c:284 */
int  sonde_crc16(const uint8_t *data, int len);

/* What this device's HBM delivers to a plain read stream right now: `bytes` of device memory at d_buf read once per repetition, 16 bytes per lane,
 * non-temporal (the decimator's access pattern without its arithmetic); *gbps = the best of `reps` passes in GB/s.  bench.py puts it beside the
 * roofline's nominal peak (the measured stream figure of the same run). */
int  sonde_probe_read_gbps(const void *d_buf, size_t bytes, int32_t reps, double *gbps);

const char *sonde_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
