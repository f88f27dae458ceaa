/*
 * sonde_rs41.h — RS41 telemetry field decode + text / JSON lines, C ABI of libsonde_hip.so.
 *
 * Bit-rate work behind the GPU path: one frame of bytes (what sonde_engine_fetch_frames / sonde_softin_fetch hand out:
 * de-whitened, ECC-corrected, with rs41_ecc()'s return value) -> exactly the characters the reference's
 * print_position() writes to stdout for that frame (rs41mod.c:2126-2470): the text line and, with --json, the JSON
 * object auto_rx parses (auto_rx/autorx/decode.py:1602-1661).  The decoder object keeps what the reference keeps in
 * gpx_t between frames: sonde ID, the 51 x 16-byte calibration / configuration subframes and what is derived from
 * them (PTU coefficients, sub-type, frequency, firmware, burst-kill timers).
 *
 * Implemented: standard and aux (xdata) frames, blocks 0x7928 / 0x7A2A / 0x7C1E / 0x7D59 / 0x7B15 / 0x7Exx / 0x76xx and
 * the newer 0x8226 (position + UTC date/time), 0x8329 (GNSS satellites), 0x7F1B (SGM xTU) and 0x80A7 (encrypted) block
 * kinds; options -v, --ptu, --ptu2, --dewp, --json, --jsnsubfrm1/2, --jsn_cfq, --silent.  Not implemented (create
 * fails with SONDE_E_ARG): --sat, -vv / -vx / --aux (satellite tables, calibration dumps, OIF411 xdata decoding).
 */
#ifndef SONDE_RS41_H
#define SONDE_RS41_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_rs41_dec sonde_rs41_dec_t;

typedef struct {
    int32_t verbose;        /* 0, 1 = -v (frequency / firmware / sub-type / timers as their subframes arrive), 2 = -vx (+ xdata text), 3 = -vv (+ battery, week, sats, subframe bytes) */
    int32_t ptu;            /* 0, 1 = --ptu, 2 = --ptu2 (rs41mod.c:2650-2651)                                         */
    int32_t dewp;           /* --dewp: dew point next to the PTU values (rs41mod.c:2000-2010)                         */
    int32_t json;           /* --json                                                                                 */
    int32_t jsn_subfrm;     /* 0, 1 = --jsnsubfrm1, 2 = --jsnsubfrm2 (imply --json, rs41mod.c:2713-2714,2770-2774)    */
    int32_t silent;         /* --silent, or -r together with --json: no text line, JSON only (rs41mod.c:2754)         */
    int32_t jsn_freq_khz;   /* "freq" of the JSON when > 0: (--jsn_cfq Hz - xlt_fq * sr + 500) / 1000 (rs41mod.c:2806-2809) */
    char    version[32];    /* "version" of the JSON — the reference compiles it in (VER_JSN_STR); "" = omit          */
    int32_t sat;            /* --sat: raw GPS block contents behind the time / position pieces (prn_sat1/2/3, rs41mod.c:2052-2111) */
    int32_t aux;            /* --aux: ozone / frost-point instruments decoded from the xdata text (implies verbose >= 2, rs41mod.c:2763)   */
    int32_t reserved[2];
} sonde_rs41_opts_t;

int  sonde_rs41_dec_create(const sonde_rs41_opts_t *opts, sonde_rs41_dec_t **out);
void sonde_rs41_dec_destroy(sonde_rs41_dec_t *d);

/* print_position(gpx, ec) for one frame: f->frame = bytes after ECC, f->ecc = rs41_ecc() return (>= 0 corrected
 * symbols, -1 / -2 / -3 failed codeword(s)).  Writes the text the reference prints (possibly empty, possibly several
 * lines: text line, JSON line, blank line) NUL-terminated into out; returns its length, or SONDE_E_ARG if it does not
 * fit (4 KiB always do). */
int  sonde_rs41_dec_frame(sonde_rs41_dec_t *d, const sonde_frame_t *f, char *out, size_t outlen);

/* rs41_ecc() for --ecc3 / --ecc4 (levels 1 / 2 work too), from the soft bits of one frame: soft0 / soft1 = the two soft values
 * read_softbit2p() returns per bit (sonde_engine_fetch_frames(.. keep_soft = 2): soft and soft1 of the frame; soft1 == NULL = one
 * soft value per bit, the --softin case), nbits of them behind the 64 header bits, ts = mv_pos / sr of the frame.  The bits are re-sliced from both values
 * ((s0 + s1) >= 0, rs41mod.c:2930), the least reliable byte positions become erasure candidates and their weakest bit a toggle
 * candidate (:1861-1941), and --ecc4 first restores bytes that are known from earlier frames of the same sonde — ID, calibration
 * subframe, frame counter (:1764-1849) — which is why this lives in the decoder object: call it BEFORE sonde_rs41_dec_frame()
 * for every frame, in order.  Fills f->frame (518 bytes after ECC), f->len (320 / 518), f->nbytes (bytes read) and f->ecc. */
int  sonde_rs41_dec_ecc(sonde_rs41_dec_t *d, int level, int inv, const float *soft0, const float *soft1, int nbits, float ts, sonde_frame_t *f);

/* Fields of the last decoded frame, for callers that want numbers instead of text. */
typedef struct {
    int32_t frame_nr; char id[12];
    int32_t year, month, day, hour, minute; float second; int32_t is_utc;
    double  lat, lon, alt, vel_h, heading, vel_v;
    int32_t sats; float batt;
    float   temp, humidity, pressure;       /* -273.15 / -1 / -1 when not (yet) available                            */
    char    subtype[12]; int32_t tx_freq_khz; int32_t crc_fail_mask;
    int32_t have_id, have_time, have_pos;   /* block CRCs of the three groups were good                              */
} sonde_rs41_fields_t;
int  sonde_rs41_dec_fields(const sonde_rs41_dec_t *d, sonde_rs41_fields_t *out);

#ifdef __cplusplus
}
#endif
#endif
