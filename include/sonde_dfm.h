/*
 * sonde_dfm.h — DFM06 / DFM09 / DFM17 / PS-15 telemetry decode + text / JSON lines, C ABI of libsonde_hip.so.
 *
 * Bit-rate work behind the GPU path: the three nibble blocks of one 280-bit frame with their Hamming verdicts (what
 * sonde_engine_fetch_dfm / sonde_softin_fetch_dfm hand out) -> exactly the characters the reference's conf_out / dat_out /
 * print_gpx write to stdout (dfm09mod.c:347-505, 694-895, 897-1150): a DFM spreads one fix over nine data packets and its
 * serial number / sensor set over the configuration channels, so the decoder object carries the reference's gpx_t state:
 * packet time stamps and error counts, serial-number detection, measurement channels, sonde type guess.
 *
 * Options: -v, -vv, --ptu, --ecc / --ecc2, --dist, --json, --jsn_cfq, --sat, -r together with --json.  Not implemented
 * (create fails with SONDE_E_ARG): -vvv, -vx (xdata / ozone dump), --dbg, -R.
 */
#ifndef SONDE_DFM_H
#define SONDE_DFM_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_dfm_dec sonde_dfm_dec_t;

typedef struct {
    int32_t verbose;        /* 0, 1 (-v), 2 (-vv), 3 (-vvv: sensor type / polarity, battery, internal temperature, on-time) */
    int32_t ptu;            /* --ptu: temperature                                                                     */
    int32_t ecc;            /* 0, 1 (--ecc; forced by --dist / --json), 2 (--ecc2)                                     */
    int32_t dist;           /* --dist: output only when packets 0,1,2,3,4,8 of the last 6 frames are all good         */
    int32_t json;           /* --json                                                                                 */
    int32_t sat;            /* --sat: geoid separation / satellites line                                              */
    int32_t raw;            /* 1 = -r given as well: the caller prints the raw line (sonde_dfm_rawline), the text line is suppressed;
                             * 2 = -R: the nine data packets as hex instead of the text line; 9 = --rawecc: the frame bits as sliced, as hex  */
    int32_t opt_auto;       /* --auto (only changes the "<+> " / "<-> " prefix of -vv)                                */
    int32_t jsn_freq_khz;   /* (--jsn_cfq + 500) / 1000, 0 = none (dfm09mod.c:1516)                                   */
    char    version[32];    /* "version" of the JSON (VER_JSN_STR of the reference build); "" = omit                  */
    int32_t dbg;            /* --dbg: the measurement channels and two alternative thermistor evaluations (dfm09mod.c:1004-1026) */
    int32_t reserved[3];
} sonde_dfm_opts_t;

int  sonde_dfm_dec_create(const sonde_dfm_opts_t *opts, sonde_dfm_dec_t **out);
void sonde_dfm_dec_destroy(sonde_dfm_dec_t *d);

/* One frame: conf_out + 2 x dat_out (+ print_gpx after a packet 8) as print_frame() sequences them
 * (dfm09mod.c:1238-1262).  f->frm_count = the reference's gpx._frmcnt (frame time stamp), f->inv = polarity in effect.
 * Writes the text the reference prints (often empty: output happens once per nine packets) into out; returns its length or
 * SONDE_E_ARG if it does not fit. */
int  sonde_dfm_dec_frame(sonde_dfm_dec_t *d, const sonde_dfm_frame_t *f, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
