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
