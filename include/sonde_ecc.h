/*
 * sonde_ecc.h — C ABI of the block-code codec in libsonde_hip.so: the reference's demod/mod/bch_ecc_mod.{c,h} code set.
 *
 *   SONDE_ECC_RS255      RS(255,231) over GF(2^8)/0x11D, alpha = 2, b = 0, p = 1      bch_ecc_mod.h:98  (RS41, RS92, LMS6, Meisei)
 *   SONDE_ECC_RS255CCSDS RS(255,223) over GF(2^8)/0x187, alpha = 2, b = 112, p = 11   bch_ecc_mod.h:99  (LMS6 / CCSDS framing)
 *   SONDE_ECC_BCH64      binary BCH(63,51), GF(2^6)/0x43, t = 2, b = 1                bch_ecc_mod.h:100 (Meisei, MRZ)
 *   SONDE_ECC_RS15CCSDS  RS(15,11) over GF(2^4)/0x13, b = 6                           bch_ecc_mod.h:103
 *
 * Functions mirror rs_encode (:860), rs_decode (:962), rs_decode_ErrEra (:877: errors + erasures, 2 e + f <= 2 t) and
 * rs_decode_bch_gf2t2 (:968), with the reference's conventions: cw[0 .. R-1] = parity, cw[R .. N-1] = message (polynomial
 * coefficient order), the word is corrected in place on success, the return value is the number of corrected symbols or the
 * reference's negative failure code (-1 roots missing, -2 Lambda(0) = 0, -3 degree check, -4 too many erasures) — so that words
 * the code cannot repair, and the rare miscorrections, come out exactly as in the reference.  Host code (frames are a few hundred
 * bytes per second per sonde); the RS41 syndromes of the batched engine are computed on the GPU (k_framesync).
 */
#ifndef SONDE_ECC_H
#define SONDE_ECC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_ECC_RS255       1
#define SONDE_ECC_RS255CCSDS  2
#define SONDE_ECC_BCH64       3
#define SONDE_ECC_RS15CCSDS   4

typedef struct sonde_ecc sonde_ecc_t;

sonde_ecc_t *sonde_ecc_create(int code);                       /* NULL for an unknown code */
void sonde_ecc_destroy(sonde_ecc_t *c);
int  sonde_ecc_params(const sonde_ecc_t *c, int *N, int *t, int *R, int *K);
int  sonde_ecc_encode(const sonde_ecc_t *c, uint8_t *cw);
int  sonde_ecc_decode(const sonde_ecc_t *c, uint8_t *cw, uint8_t *err_pos, uint8_t *err_val);
int  sonde_ecc_decode_errera(const sonde_ecc_t *c, uint8_t *cw, int nera, const uint8_t *era_pos, uint8_t *err_pos, uint8_t *err_val);
int  sonde_ecc_decode_bch_gf2t2(const sonde_ecc_t *c, uint8_t *cw, uint8_t *err_pos, uint8_t *err_val);

#ifdef __cplusplus
}
#endif
#endif
