/*
 * sonde_mts01.h — Meteosis MTS01 bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/mts01mod.c does behind its demodulator: a header hit (AA AA B4 2B, 32 bits at 1200 Bd) is followed by 1048 bits = 131 bytes
 * MSB first: one byte, 128 bytes of comma-separated ASCII telemetry, CRC-16 (poly 0x8005, init 0xFFFF, result bit-reversed, low byte
 * first).  Printed as the string + [OK] / [NO], with -v the parsed fields, with --json the JSON object of frames whose CRC holds.
 * The sample-rate part is the engine's generic sonde description (1200 Bd, BT 1.5, h 0.9, 2 header errors accepted, centre window 2 for
 * IF-rate IQ); host/mts01mod.c puts the two together.
 *
 * Mirrors print_frame mts01mod.c:151-286 (crc16_re :76-99, bits2bytes :101-127, fn :129-137, get_Temp :139-148), the bit loop of main
 * :569-618 and, for soft input, find_softbinhead / corr_softhdb (demod_mod.c:1692-1762; threshold 0.8, mts01mod.c:557).
 */
#ifndef SONDE_MTS01_H
#define SONDE_MTS01_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_MTS01_FRAME_BITS 1048      /* 8 * (130 + 1), mts01mod.c:52-53 */

typedef struct sonde_mts01_dec sonde_mts01_dec_t;

typedef struct {
    int32_t raw;             /* 1 = -r (bytes as hex + CRC), 2 = -R (bits)                                               */
    int32_t verbose;         /* -v                                                                                      */
    int32_t json;            /* --json                                                                                  */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                             */
    char    version[32];     /* "version" of the JSON; "" = omit                                                        */
    int32_t reserved[4];
} sonde_mts01_opts_t;

int  sonde_mts01_dec_create(const sonde_mts01_opts_t *opts, sonde_mts01_dec_t **out);
void sonde_mts01_dec_destroy(sonde_mts01_dec_t *d);

/* One header hit from a demodulator: n (<= SONDE_MTS01_FRAME_BITS) soft values of the bits behind the header in RAW polarity (the reference
 * reads them without regard to the header's sign, :604).  A short frame (stream ended) keeps the previous frame's bits behind it, as the
 * reference's buffer does; fewer than 129 bytes print nothing.  Writes what the reference prints NUL-terminated into out; returns its
 * length or a negative SONDE_E_* code. */
int  sonde_mts01_dec_frame(sonde_mts01_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen);

/* Soft-bit input (`mts01mod --softin`): header search and frame assembly inside; finish != 0 at end of input (a frame in progress is
 * printed with the bits that exist, :613-615). */
int  sonde_mts01_dec_push_soft(sonde_mts01_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
