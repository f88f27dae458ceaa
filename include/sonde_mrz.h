/*
 * sonde_mrz.h — Meteo-Radiy MRZ (MP3-H1) bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/mp3h1mod.c does behind its demodulator: a header hit (AA AA + 101111 as 44 Manchester half symbols at 2399 Bd) is followed
 * by the rest of a 51-byte (ECEF position) or 48-byte (lat / lon position) frame, MSB first: AA BF 35, sub-frame counter, time, position,
 * velocity, PTU, one of 16 configuration words per frame (calibration, serial numbers, date), CRC-16 (0xA001 reflected, init 0xFFFF).
 * Printed as the position line + [OK] / [NO], with --json the JSON object once date and both serial numbers are known.
 * The sample-rate part is the engine's generic sonde description (2399 Bd, two half symbols per bit, BT 1.0, h 2.0, 2 header errors, bit
 * offset 2, centre window 2 for IF-rate IQ, polarity per -i / --auto); host/mp3h1mod.c puts the two together.
 *
 * Mirrors print_frame mp3h1mod.c:784-862, print_gpx :629-782, bits2bytes :157-183, crc16rev / check_CRC :280-311, ecef2elli :325-341,
 * get_GPSkoord_ecef / _latlon :343-427, get_time :437-473 (datetime2GPSweek :187-205), get_ptu :482-531, get_cfg :533-622, the bit loop of
 * main :1173-1243, the --rawhex reader :1249-1276 and, for soft input, find_softbinhead / corr_softhdb (demod_mod.c:1692-1762; threshold
 * 0.82, :1164).
 */
#ifndef SONDE_MRZ_H
#define SONDE_MRZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_MRZ_MAX_BITS 386           /* (45 + 6) * 8 - 22 bits behind the header (mp3h1mod.c:62,1196); 362 for the lat / lon frame */

typedef struct sonde_mrz_dec sonde_mrz_dec_t;

typedef struct {
    int32_t raw;             /* 1 = -r (frame bytes as hex), 2 = -R (bits)                                               */
    int32_t verbose;         /* -v = 1, -vv = 2                                                                         */
    int32_t dbg;             /* --dbg: the configuration word of the frame                                              */
    int32_t ptu;             /* --ptu                                                                                   */
    int32_t uniq;            /* --uniq: one line per second (frames are sent six times)                                 */
    int32_t color;           /* -c                                                                                      */
    int32_t json;            /* --json                                                                                  */
    int32_t inv, aut;        /* -i, --auto: polarity handling of the soft-input framer (:1186-1189)                      */
    int32_t bits_ofs;        /* --ofs n: first frame bit (0..64); used when bits_ofs_given != 0, else the default 8      */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                             */
    char    version[32];     /* "version" of the JSON; "" = omit                                                        */
    int32_t bits_ofs_given;
    int32_t reserved[3];
} sonde_mrz_opts_t;

int  sonde_mrz_dec_create(const sonde_mrz_opts_t *opts, sonde_mrz_dec_t **out);
void sonde_mrz_dec_destroy(sonde_mrz_dec_t *d);

/* Bits the decoder reads behind a header for the frame type currently in effect: 386 (ECEF) or 362 (lat / lon).  It changes when a frame
 * of the other type passes its CRC (:812-814) — a demodulator slicing a fixed number of bits per hit has to follow. */
int  sonde_mrz_dec_frame_bits(const sonde_mrz_dec_t *d);

/* One header hit from a demodulator: n (<= frame_bits) soft values of the bits behind the header in the polarity in effect (what the
 * engine stores: second half symbol minus first, negated for an inverted signal).  Writes what the reference prints NUL-terminated into
 * out; returns its length or a negative SONDE_E_* code. */
int  sonde_mrz_dec_frame(sonde_mrz_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen);
/* --rawhex: one line of hex frame bytes separated by blanks (the output of -r); lines of 20 bytes or less are ignored */
int  sonde_mrz_dec_rawhex(sonde_mrz_dec_t *d, const char *line, char *out, size_t outlen);
/* Soft-symbol input (`mp3h1mod --softin`, decode.py:1293: two float32 half symbols per bit): header search, polarity and frame assembly
 * inside; finish != 0 at end of input. */
int  sonde_mrz_dec_push_soft(sonde_mrz_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
