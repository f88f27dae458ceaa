/*
 * sonde_m10.h — M10 / M10+ telemetry decode + text / JSON lines, C ABI of libsonde_hip.so.
 *
 * One frame of bytes (sonde_engine_fetch_m10: differentially decoded, checksum evaluated) -> exactly the characters the
 * reference's print_pos() writes (m10mod.c:862-1047): Trimble (type 0x9F) or Gtop (0xAF) GPS fields, serial number, thermistor
 * temperature with its three measuring ranges, humidity from the capacitance counter ratio, battery voltage, JSON.
 * Options: -v, -vv, -vvv, --ptu, --json, --jsn_cfq, --silent.  Colour output (-c) is not implemented.
 */
#ifndef SONDE_M10_H
#define SONDE_M10_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_m10_dec sonde_m10_dec_t;

typedef struct {
    int32_t verbose;        /* 0..3 (-v, -vv, -vvv)                                                                    */
    int32_t ptu;            /* --ptu                                                                                   */
    int32_t json;           /* --json                                                                                  */
    int32_t silent;         /* --silent, or -r together with --json (m10mod.c:1328)                                    */
    int32_t raw;            /* -r: the caller prints the raw line; print_pos only runs when silent (m10mod.c:1123)     */
    int32_t jsn_freq_khz;   /* "freq" of the JSON when > 0                                                             */
    char    version[32];    /* "version" of the JSON (VER_JSN_STR of the reference build); "" = omit                   */
    int32_t color;          /* -c: ANSI colours around the fields of the position line (m10mod.c:253-277,886-923)       */
    int32_t reserved[3];
} sonde_m10_opts_t;

int  sonde_m10_dec_create(const sonde_m10_opts_t *opts, sonde_m10_dec_t **out);
void sonde_m10_dec_destroy(sonde_m10_dec_t *d);
/* what print_frame() prints for this frame besides the raw line (m10mod.c:1049-1140); returns strlen or SONDE_E_ARG */
int  sonde_m10_dec_frame(sonde_m10_dec_t *d, const sonde_m10_frame_t *f, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
