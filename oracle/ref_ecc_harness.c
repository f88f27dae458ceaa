/*
 * oracle/ref_ecc_harness.c — TEST INFRASTRUCTURE.  Thin shim over the reference's own
 * bch_ecc_mod.o (rs_init_RS255 bch_ecc_mod.c:742, rs_decode :962, rs_decode_ErrEra :877,
 * rs_encode :860) so tests can compare the restated / engine RS(255,231) decoder with it
 * codeword by codeword, including uncorrectable and miscorrected words.
 */
#include <string.h>
#include "bch_ecc_mod.h"

static RS_t g_rs;
static int g_init = 0;

static void ensure(void) { if (!g_init) { rs_init_RS255(&g_rs); g_init = 1; } }

int ref_rs255_encode(ui8_t cw[255]) { ensure(); return rs_encode(&g_rs, cw); }

/* returns the reference's error count (>=0) or its negative failure code; cw corrected in place */
int ref_rs255_decode(ui8_t cw[255], ui8_t err_pos[24], ui8_t err_val[24]) {
    ensure();
    return rs_decode(&g_rs, cw, err_pos, err_val);
}

int ref_rs255_decode_era(ui8_t cw[255], int nera, ui8_t *era_pos, ui8_t err_pos[24], ui8_t err_val[24]) {
    ensure();
    return rs_decode_ErrEra(&g_rs, cw, nera, era_pos, err_pos, err_val);
}

void ref_rs255_tables(ui8_t exp_a[256], ui8_t log_a[256], ui8_t g[25]) {
    ensure();
    memcpy(exp_a, g_rs.GF.exp_a, 256); memcpy(log_a, g_rs.GF.log_a, 256); memcpy(g, g_rs.g, 25);
}

/* every code of bch_ecc_mod.h:98-103 (1 RS(255,231), 2 RS(255,223) CCSDS, 3 BCH(63,51), 4 RS(15,11)) for tests/test_ecc_codes.py */
static RS_t g_codes[5];
static int g_codes_init[5];
static RS_t *code(int k) {
    if (k < 1 || k > 4) return 0;
    if (!g_codes_init[k]) {
        if (k == 1) rs_init_RS255(&g_codes[k]);
        if (k == 2) rs_init_RS255ccsds(&g_codes[k]);
        if (k == 3) rs_init_BCH64(&g_codes[k]);
        if (k == 4) rs_init_RS15ccsds(&g_codes[k]);
        g_codes_init[k] = 1;
    }
    return &g_codes[k];
}
int ref_ecc_params(int k, int *N, int *t, int *R, int *K) { RS_t *r = code(k); if (!r) return -1; *N = r->N; *t = r->t; *R = r->R; *K = r->K; return 0; }
int ref_ecc_encode(int k, ui8_t *cw) { RS_t *r = code(k); return r ? rs_encode(r, cw) : -9; }
int ref_ecc_decode(int k, ui8_t *cw, ui8_t *err_pos, ui8_t *err_val) { RS_t *r = code(k); return r ? rs_decode(r, cw, err_pos, err_val) : -9; }
int ref_ecc_decode_era(int k, ui8_t *cw, int nera, ui8_t *era_pos, ui8_t *err_pos, ui8_t *err_val) {
    RS_t *r = code(k); return r ? rs_decode_ErrEra(r, cw, nera, era_pos, err_pos, err_val) : -9;
}
int ref_ecc_decode_bch(int k, ui8_t *cw, ui8_t *err_pos, ui8_t *err_val) { RS_t *r = code(k); return r ? rs_decode_bch_gf2t2(r, cw, err_pos, err_val) : -9; }
