/*
 * sonde_lms6.h — LMS6-403 / LMS-X bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/lms6Xmod.c does behind its demodulator: a header hit is followed by one block of raw channel bits (rate-1/2, K = 7
 * convolutional code, every second bit inverted); the block is decoded (algebraic `deconv` or Viterbi, hard or soft), cut into bytes,
 * RS(255,223)-corrected, and the 223-byte data frames inside it are checked (CRC-16) and printed as text / JSON.  The sample-rate part
 * (FM / tone demodulation, header correlation, bit slicing) is the engine's generic sonde description (sonde_hip.h: SONDE_GENERIC with
 * the LMS6 header, 4800 Bd, BT 1.2, h 0.9, 10 header errors accepted, slice_baud for LMS-X); host/lms6Xmod.c puts the two together.
 *
 * Mirrors: proc_frame lms6Xmod.c:829-989, deconv :343-374, viterbi :232-341, bits2bytes :415-441, frmsync_6 / frmsync_X :800-827,
 * field getters :464-697, print_frame :713-798, the block loop of main :1353-1465 (raw-bit sign alternation, --ecc3 soft-bit merge,
 * LMS6 <-> LMS-X auto detection) and, for soft input, find_softbinhead / corr_softhdb (demod_mod.c:1692-1762).
 */
#ifndef SONDE_LMS6_H
#define SONDE_LMS6_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_lms6_dec sonde_lms6_dec_t;

typedef struct {
    int32_t raw;             /* -r: frame bytes instead of the text line                                              */
    int32_t ecc;             /* 0, 1 = --ecc, 3 = --ecc3 (soft-bit merge before decoding, lms6Xmod.c:1395-1400)        */
    int32_t vit;             /* 0 = algebraic deconvolution, 1 = --vit (hard), 2 = --vit2 (soft)                       */
    int32_t json;            /* --json (implies ecc = 1, vit = 1, lms6Xmod.c:1153-1157)                                */
    int32_t typ;             /* 0 = auto detection (starts as LMS6), 6 = --lms6, 10 = --lmsX                           */
    int32_t gpsweek;         /* --gpsweek w (1024..3072), 0 = none                                                     */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                            */
    char    version[32];     /* "version" of the JSON (the reference compiles it in as VER_JSN_STR); "" = omit         */
    int32_t reserved[4];
} sonde_lms6_opts_t;

int  sonde_lms6_dec_create(const sonde_lms6_opts_t *opts, sonde_lms6_dec_t **out);
void sonde_lms6_dec_destroy(sonde_lms6_dec_t *d);

/* Raw bits the decoder reads behind a header for the type currently in effect: 4096 (LMS6) or 4720 (LMS-X). */
int  sonde_lms6_dec_block_bits(const sonde_lms6_dec_t *d);
/* 6 or 10 (| 0x0200 for the LMS6-403-2 frame sync): the type in effect; *symbol_rate_changed = 1 when the last block made the auto
 * detection switch between LMS6 (4800 Bd) and LMS-X (4797.8 Bd) — the caller's demodulator has to follow (lms6Xmod.c:1436-1462). */
int  sonde_lms6_dec_type(const sonde_lms6_dec_t *d, int32_t *symbol_rate_changed);

/* One header hit from a demodulator: nbits (<= block_bits) soft values of the bits behind the 64-bit header in RAW polarity, as
 * read_softbit2p() returns them (soft1: the second soft value per bit, may be NULL), mv = header score (its sign is the phase of the
 * (c0, inv(c1)) alternation), frm_rate = 4800 * sr / (mv_pos - previous mv_pos) (:1372), t_elapsed = seconds of input so far.
 * Writes what the reference prints for this block NUL-terminated into out; returns its length or a negative SONDE_E_* code. */
int  sonde_lms6_dec_block(sonde_lms6_dec_t *d, const float *soft0, const float *soft1, int32_t nbits, float mv, float frm_rate,
                          double t_elapsed, char *out, size_t outlen);

/* Soft-bit input (`lms6Xmod --softin`, the consumer of `fsk_demod -s`): n float32 soft bits in, header search and block assembly
 * inside; finish != 0 at end of input (a block in progress is decoded with the bits that exist).  Output as above. */
int  sonde_lms6_dec_push_soft(sonde_lms6_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
