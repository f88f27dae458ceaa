/*
 * sonde_m20.h — M20 telemetry decode + text / JSON lines, C ABI of libsonde_hip.so.
 *
 * One frame of bytes (sonde_engine_fetch_m20 / sonde_softin_fetch_m20) -> exactly the characters the reference's m20mod
 * print_pos() writes (m20mod.c:729-868): GPS time of week (seconds) and week, position in 1e-6 degrees, 24-bit altitude,
 * velocities, serial number text, thermistor temperature (range from the ADC word), humidity-sensor temperature, humidity,
 * optional pressure, battery, JSON.  Options: -v, -vv, -vvv, --ptu, --json, --jsn_cfq, --silent.  -c colours the position line (opts.color) and the raw line (sonde_m20_rawline with SONDE_M20_COLOR in `verbose`).
 */
#ifndef SONDE_M20_H
#define SONDE_M20_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_m20_dec sonde_m20_dec_t;

typedef struct {
    int32_t verbose, ptu, json, silent, raw, jsn_freq_khz;      /* as sonde_m10_opts_t */
    char    version[32];
    int32_t color;           /* -c: ANSI colours around the fields of the position line (m20mod.c:239-267,772-822)                    */
    int32_t reserved[3];
} sonde_m20_opts_t;

int  sonde_m20_dec_create(const sonde_m20_opts_t *opts, sonde_m20_dec_t **out);
void sonde_m20_dec_destroy(sonde_m20_dec_t *d);
/* what print_frame() prints for this frame besides the raw line (m20mod.c:870-1008); returns strlen or SONDE_E_ARG */
int  sonde_m20_dec_frame(sonde_m20_dec_t *d, const sonde_m20_frame_t *f, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
