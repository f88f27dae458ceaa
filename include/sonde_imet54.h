/*
 * sonde_imet54.h — InterMet iMet-54 / iMet-50 bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/imet54mod.c does behind its demodulator: a header hit (0x00 0xAA 0x24 0x24 as 8N1 characters, 4798 Bd) is followed by 2200
 * bits = 220 8N1 characters: start / stop bits removed, three sync characters skipped, 64-bit blocks de-interleaved (8 x 8 transpose),
 * Hamming(8,4) codewords -> nibbles -> 108 frame bytes; two check sums (the 32-bit polynomial check of the standard frame, CRC-32 of the
 * continuous frame), GPS time / position, PTU (Hyland-Wexler humidity correction), status, text line and JSON.
 * The sample-rate part is the engine's generic sonde description (4798 Bd, BT 1.0, h 0.8, 4 header errors accepted, bit offset 1, centre
 * window 2 for IF-rate IQ, polarity as -i / --auto say); host/imet54mod.c puts the two together.
 *
 * Mirrors print_frame imet54mod.c:618-707, print_position :494-616, de8n1 :107-118, deinter64 :120-133, check / hamming :162-227, crc32ok
 * :229-284, crc32_802 :286-303, crc32ok_cont :350-360, get_GPS :368-406, vaporSatP :409-422, get_PTU :424-475, the bit loop of main
 * :1008-1061, the --rawhex reader :1086-1112 and, for soft input, find_softbinhead / corr_softhdb (demod_mod.c:1692-1762; threshold 0.8, :998).
 * For a frame cut short by the end of the stream the reference sums Hamming results it never computed (locals that are not cleared, :623-624,:651):
 * in the compiled reference they hold what the previous frame left there, and so they do here (kept in the decoder object).
 */
#ifndef SONDE_IMET54_H
#define SONDE_IMET54_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_IMET54_FRAME_BITS 2200     /* 220 8N1 characters, imet54mod.c:59-62 */

typedef struct sonde_imet54_dec sonde_imet54_dec_t;

typedef struct {
    int32_t raw;             /* 1 = -r (frame bytes as hex), 4 = -r4 (grouped)                                           */
    int32_t verbose;         /* -v                                                                                      */
    int32_t ecc;             /* --ecc: Hamming single-error correction                                                  */
    int32_t ptu;             /* --ptu                                                                                   */
    int32_t silent;          /* --silent (or -r with --json, :900): no position line                                    */
    int32_t json;            /* --json (implies ecc)                                                                    */
    int32_t inv, aut;        /* -i, --auto: polarity handling of the soft-input framer (:1018-1021)                      */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                             */
    char    version[32];     /* "version" of the JSON; "" = omit                                                        */
    int32_t reserved[4];
} sonde_imet54_opts_t;

int  sonde_imet54_dec_create(const sonde_imet54_opts_t *opts, sonde_imet54_dec_t **out);
void sonde_imet54_dec_destroy(sonde_imet54_dec_t *d);

/* One header hit from a demodulator: n (<= SONDE_IMET54_FRAME_BITS) soft values of the bits behind the header in the polarity in effect
 * (what the engine stores).  Writes what the reference prints NUL-terminated into out; returns its length or a negative SONDE_E_* code. */
int  sonde_imet54_dec_frame(sonde_imet54_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen);
/* --rawhex: one line of hex frame bytes (the output of -r); lines of 20 bytes or less are ignored */
int  sonde_imet54_dec_rawhex(sonde_imet54_dec_t *d, const char *line, char *out, size_t outlen);
/* Soft-bit input (`imet54mod --softin`, decode.py:1250): header search, polarity and frame assembly inside; finish != 0 at end of input. */
int  sonde_imet54_dec_push_soft(sonde_imet54_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
