/*
 * sonde_meisei.h — Meisei iMS-100 / RS-11G bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/meisei100mod.c does behind its demodulator: a header hit (0x049DCE as 48 biphase-S half symbols at 2400 Bd) is followed
 * by 1152 hard half symbols = 576 bits: the rest of subframe 0 and subframe 1 (header 0xFB6230), each 6 blocks of 46 bits — BCH(63,51)
 * shortened to (46,34), message 16 + parity + 16 + parity.  Blocks are corrected (--ecc), the two variants (RS-11G / iMS-100) are told
 * apart by the type word and decoded to the reference's text lines / JSON; configuration words arrive one per frame (64-frame cycle)
 * and carry the serial number, the transmit frequency and the temperature / humidity calibration.
 * The sample-rate part is the engine's generic sonde description (48-symbol header, 2400 Bd, BT 1.2, h 2.4, 1 header error accepted);
 * host/meisei100mod.c puts the two together.
 *
 * Mirrors the frame loop of main, meisei100mod.c:681-1318 (biphi_s :213-229, BCH + parity check :735-776, RS-11G :779-1017, iMS-100
 * :1018-1283, raw :1284-1310), f32e2 :163-191, est_year_ims100 :330-342, the config sanity checks :254-300, and for soft input
 * find_softbinhead / corr_softhdb (demod_mod.c:1692-1762; threshold 0.8, meisei100mod.c:668).
 * Deviations: the reference reads `counter` before any frame has set it and `block_err[]` without --ecc (uninitialised stack, :386,:381);
 * both start as 0 here.
 */
#ifndef SONDE_MEISEI_H
#define SONDE_MEISEI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_MEISEI_FRAME_SYMBOLS 1152     /* half symbols behind the header: 2 * 600 - 48 (meisei100mod.c:707) */

typedef struct sonde_meisei_dec sonde_meisei_dec_t;

typedef struct {
    int32_t raw;             /* -r: the 16-bit words of both subframes as hex                                           */
    int32_t verbose;         /* -v                                                                                      */
    int32_t dbg;             /* --dbg: configuration word of each frame                                                 */
    int32_t ecc;             /* --ecc: BCH(63,51) per block                                                             */
    int32_t json;            /* --json (implies ecc)                                                                    */
    int32_t ptu;             /* --ptu                                                                                   */
    int32_t ims100;          /* start as iMS-100 (--ims100, or a file name given without -r / --rs11g, :552); else RS-11G; switches by itself */
    int32_t ref_year;        /* --year (2004..2099) for the one-digit year of the iMS-100; 0 = 2024                      */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                             */
    char    version[32];     /* "version" of the JSON; "" = omit                                                        */
    int32_t reserved[4];
} sonde_meisei_opts_t;

int  sonde_meisei_dec_create(const sonde_meisei_opts_t *opts, sonde_meisei_dec_t **out);
void sonde_meisei_dec_destroy(sonde_meisei_dec_t *d);

/* One header hit from a demodulator: the SONDE_MEISEI_FRAME_SYMBOLS soft values behind the header (sign = half symbol, either
 * polarity: biphase-S compares neighbours).  Fewer symbols (stream ended): nothing is printed, as in the reference.  Writes what the
 * reference prints for this frame NUL-terminated into out; returns its length or a negative SONDE_E_* code. */
int  sonde_meisei_dec_frame(sonde_meisei_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen);

/* Soft-bit input (`meisei100mod --softin`, decode.py:1379): n float32 soft half symbols in, header search and frame assembly inside;
 * finish != 0 at end of input appends the newline the reference prints before it exits (:1320). */
int  sonde_meisei_dec_push_soft(sonde_meisei_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
