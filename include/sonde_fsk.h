/*
 * sonde_fsk.h — C ABI of the batched 2-/4-FSK modem in libsonde_hip.so.
 *
 * Replaces the reference's utils/fsk.c demodulator (the codec2 "fsk_demod" auto_rx pipes IQ into,
 * auto_rx/autorx/decode.py:901,976,1067,1120) for many channels at once.  The reference seam is
 * fsk_create_hbr / fsk_set_freq_est_limits / fsk_set_freq_est_alg / fsk_nin / fsk_demod_sd / fsk_get_demod_stats /
 * fsk_destroy (utils/fsk.h:115-205) over struct FSK (fsk.h:47-95); every channel here is one such struct.
 * host/fsk_demod.c keeps the CLI (utils/fsk_demod.c) on top of it; host/seam/fsk_hip.c is the fsk.h function seam itself (the
 * reference's fsk_demod.c links against it unchanged).  Conventions as sonde_hip.h.
 */
#ifndef SONDE_FSK_H
#define SONDE_FSK_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* input sample formats (fsk_demod.c:103-110,283-311) */
#define SONDE_FSK_S16   1       /* real int16,  x/1000           */
#define SONDE_FSK_CS16  2       /* --cs16: complex int16, x/1000 */
#define SONDE_FSK_CU8   3       /* --cu8: complex uint8, (u-127)/128 */
#define SONDE_FSK_CF32  4       /* complex float32 as is: the COMP fsk_in[] of fsk_demod() / fsk_demod_sd() (fsk.h:176,186) */

typedef struct sonde_fsk sonde_fsk_t;

typedef struct {
    int32_t abi_version;     /* SONDE_ABI_VERSION                                            */
    int32_t device;
    int32_t n_channels;
    int32_t Fs, Rs;          /* sample / symbol rate; Fs % Rs == 0 (fsk.c:127)               */
    int32_t M;               /* 2 or 4 tones (fsk.c:130); 4-FSK yields two soft bits per symbol */
    int32_t P;               /* -p: timing oversampling, (Fs/Rs) % P == 0 (fsk.c:129)        */
    int32_t nsym;            /* --nsym: symbols per modem frame                              */
    int32_t format;          /* SONDE_FSK_*                                                  */
    int32_t fsk_lower, fsk_upper;   /* -b / -u estimator limits in Hz (fsk_set_freq_est_limits) */
    int32_t mask;            /* --mask given: mask estimator (fsk_set_freq_est_alg)          */
    int32_t tone_spacing;    /* --mask <Hz> (tx_tone_separation, default 100)                */
    int32_t max_chunk;       /* largest n_samples per process call                           */
    int32_t burst_mode;      /* fsk_enable_burst_mode: nin never adjusted (fsk.c:724,976)    */
    int32_t raw_eye;         /* fsk_stats_normalise_eye(fsk, 0): eye traces not normalised   */
    int32_t reserved[2];
} sonde_fsk_cfg_t;

/* struct FSK constants (fsk_create_core, fsk.c:114-201) */
typedef struct {
    int32_t Ts, N, Ndft, Nmem, Nbits;
    float   tc;
    int32_t reserved[4];
} sonde_fsk_info_t;

/* per modem frame: what fsk_demod_core leaves in struct FSK / MODEM_STATS (fsk.c:593-915) */
typedef struct {
    int32_t nin;             /* samples this frame consumed                                  */
    int32_t nin_next;        /* fsk_nin() after the frame                                    */
    float   f_est[4];        /* tone estimates used by the demod (peak or mask estimator), M of them */
    float   norm_rx_timing;
    float   ppm;
    float   EbNodB;
    float   snr_est;         /* MODEM_STATS.snr_est (the "EbNodB" of the stats JSON)         */
} sonde_fsk_frame_t;

int  sonde_fsk_create(const sonde_fsk_cfg_t *cfg, sonde_fsk_t **out);     /* fsk_create_hbr + limits + estimator */
void sonde_fsk_destroy(sonde_fsk_t *f);                                   /* fsk_destroy                          */
int  sonde_fsk_info(const sonde_fsk_t *f, sonde_fsk_info_t *info);

/* Push n_samples per channel (channel c at in + c*ch_stride samples); runs every modem frame for which fsk_nin()
 * samples are available (the `while (fread(.., fsk_nin(fsk), ..))` loop of fsk_demod.c:279) — samples left over stay
 * queued.  Synchronous. */
int  sonde_fsk_process_host(sonde_fsk_t *f, const void *h_in, int64_t ch_stride, int32_t n_samples);
int  sonde_fsk_process_device(sonde_fsk_t *f, const void *d_in, int64_t ch_stride, int32_t n_samples);
/* sonde_fsk_process_device in two halves: submit enqueues everything on the engine's stream and returns, wait blocks until the launch is through.
 * Between the two the host is free: the other engines of a mixed batch can be submitted (their launches overlap on the GPU without a host thread
 * each).  Any other call of the engine waits first.  d_in is read by a copy that is only ENQUEUED when submit returns: it must stay untouched until
 * sonde_fsk_wait (or any other call of the engine) has returned. */
int  sonde_fsk_submit_device(sonde_fsk_t *f, const void *d_in, int64_t ch_stride, int32_t n_samples);
int  sonde_fsk_wait(sonde_fsk_t *f);

/* Channels fed independently (the resident broker, host/sonde_broker.c: every client reads exactly fsk_nin() samples of its own
 * stream per frame, and nin differs between channels): channel c gets n_samples[c] samples from h_in[c] (0 = nothing this time).
 * An engine is fed either this way or through sonde_fsk_process_host / _device, not both. */
int  sonde_fsk_process_host_var(sonde_fsk_t *f, const void *const *h_in, const int32_t *n_samples);
/* Back to the state fsk_create_hbr() leaves (oscillators, timing, Sf, nin = N) for one channel; samples queued for it are dropped.
 * The next samples fed to the channel are the first of a new stream. */
int  sonde_fsk_reset_channel(sonde_fsk_t *f, int32_t channel);

/* Soft decisions (fsk_demod_sd: one float per bit — 2-FSK: >0 = the lower tone; 4-FSK: two per symbol, fsk.c:793-802) produced by the last process call for one
 * channel; returns the number of floats written (<= max). frames (optional, may be NULL): per-frame records,
 * max_frames entries; *n_frames receives the count. */
int  sonde_fsk_fetch(sonde_fsk_t *f, int32_t channel, float *sd, int32_t max, sonde_fsk_frame_t *frames, int32_t max_frames,
                     int32_t *n_frames);
/* Hard decisions of the same frames (rx_bits of fsk_demod(): the strictly largest tone, first wins; fsk.c:760-778), one byte per bit */
int  sonde_fsk_fetch_bits(sonde_fsk_t *f, int32_t channel, uint8_t *bits, int32_t max);
/* fsk_get_demod_stats + Sf: smoothed magnitude spectrum (Ndft floats, DC at Ndft/2) and samples consumed so far */
int  sonde_fsk_stats(sonde_fsk_t *f, int32_t channel, sonde_fsk_frame_t *last, float *Sf, int64_t *samples);
/* Eye diagram of the last modem frame as fsk_get_demod_stats() returns it (rx_eye, fsk.c:857-903; modem_stats.h:63-65):
 * neyetr = 8 traces (8/M per tone, interleaved by tone) of neyesamp = 2P/ceil(2P/160) integrator magnitudes, normalised
 * to the largest.  eye receives neyetr * neyesamp floats (row-major; at most 8 * 160); returns that count. */
int  sonde_fsk_eye(sonde_fsk_t *f, int32_t channel, float *eye, int32_t *neyetr, int32_t *neyesamp);
int  sonde_fsk_clear_estimators(sonde_fsk_t *f);                          /* fsk_clear_estimators (fsk.c:981)     */
int  sonde_fsk_kernel_ms(sonde_fsk_t *f, double *avg_ms, int64_t *launches);

/* ---- the consumer of the soft decisions on the device: `rs41mod --softin [-i] [--ecc|--ecc2]` for every channel of a modem engine
 * (auto_rx's pipe `fsk_demod ... | rs41mod --softin -i`, auto_rx/autorx/decode.py:901-909).  find_softbinhead / corr_softhdb
 * (demod/mod/demod_mod.c:1692-1762, threshold 0.7), the bit loop and de-whitening of rs41mod.c:2893-2962 and rs41_ecc() (:1703-1769) run in
 * device memory; only completed frames (518 bytes each) come to the host.  invert_stream = --softinv, opt_inv = -i, opt_auto = --auto.
 * sonde_type: SONDE_RS41; SONDE_DFM09 = `dfm09mod --softin [-i] [--ecc|--ecc2]` (dfm09mod.c:1604-1720: 32 raw header symbols, two soft symbols per bit, eight frames
 * per header hit, de-interleave + Hamming(8,4) incl. the soft 2-bit pass :231-345 on a lane per codeword); SONDE_M10 = `m10mod --softin` (m10mod.c:1405-1510: header
 * threshold 0.8 in either polarity, differential decoding, the rest of the second skipped, checkM10 :594-628).  No CPU fallback. */
typedef struct sonde_softin_dev sonde_softin_dev_t;
int  sonde_softin_dev_create(int32_t n_channels, int32_t sonde_type, int32_t ecc_level, int32_t invert_stream, int32_t opt_inv, int32_t opt_auto,
                             sonde_softin_dev_t **out);
void sonde_softin_dev_destroy(sonde_softin_dev_t *s);
/* consume the soft decisions the modem's last process call left in device memory (every channel; n_channels must match).  Synchronous. */
int  sonde_softin_dev_push_fsk(sonde_softin_dev_t *s, sonde_fsk_t *modem);
/* sonde_softin_dev_push_fsk in two halves: submit waits for the modem's launch (sonde_fsk_wait), then puts the consumer's kernels and the copies of its frames on the
 * consumer's OWN stream and returns; collect waits for them.  In between the modem can be given its next second (sonde_fsk_submit_device): the modem keeps the soft
 * decisions of its last two launches, so the consumer of second k runs beside the modem of second k + 1.  Order per second: sonde_fsk_wait(k - 1), collect (k - 2),
 * submit_fsk (k - 1), sonde_fsk_submit_device (k).  The modem's launch that overwrites a buffer of soft decisions waits (on the device) for the consumer that was
 * given that buffer, so another order costs overlap, never frames. */
int  sonde_softin_dev_submit_fsk(sonde_softin_dev_t *s, sonde_fsk_t *modem);
/* the same over the modem's launch BEFORE the one in flight — for the order sonde_fsk_wait (k - 1), sonde_fsk_submit_device (k), collect (k - 2), submit_fsk_behind (k - 1):
 * the modem gets its next second before the host does the decoder's bookkeeping, and nothing here waits for the launch in flight.  With no launch in flight:
 * sonde_softin_dev_submit_fsk. */
int  sonde_softin_dev_submit_fsk_behind(sonde_softin_dev_t *s, sonde_fsk_t *modem);
int  sonde_softin_dev_collect(sonde_softin_dev_t *s);
/* the same over any soft-bit streams in device memory: channel c at d_soft + c * ch_stride, n_bits each */
int  sonde_softin_dev_push_device(sonde_softin_dev_t *s, const float *d_soft, int64_t ch_stride, int32_t n_bits);
/* frames completed by the push calls since the last fetch (all channels, in completion order per call; channel / len / ecc / mv / mv_pos = the header's
 * bit index in the channel's stream); returns the count (<= max) */
int  sonde_softin_dev_fetch(sonde_softin_dev_t *s, sonde_frame_t *out, int32_t max);
/* SONDE_DFM09 / SONDE_M10 consumers: their frames (ecc[3] = hamming()'s value per block; cs_ok / cs_calc = the frame checksum) */
int  sonde_softin_dev_fetch_dfm(sonde_softin_dev_t *s, sonde_dfm_frame_t *out, int32_t max);
int  sonde_softin_dev_fetch_m10(sonde_softin_dev_t *s, sonde_m10_frame_t *out, int32_t max);
/* tallies since creation: frames completed, frames accepted (RS41: rs41_ecc() >= 0; DFM: no block uncorrectable; M10: checksum good), frames repaired, symbols / codewords repaired, frames lost to a full buffer */
int  sonde_softin_dev_counts(sonde_softin_dev_t *s, int64_t *frames, int64_t *ecc_ok, int64_t *repaired, int64_t *symbols, int64_t *dropped);

#ifdef __cplusplus
}
#endif
#endif
