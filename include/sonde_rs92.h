/*
 * sonde_rs92.h — Vaisala RS92-SGP / RS92-NGP bit-rate tier of libsonde_hip.so (C ABI, host code: no GPU involved).
 *
 * What demod/mod/rs92mod.c does behind its demodulator: a header hit (2A 2A 10 as 8N1 Manchester, 60 raw symbols at 4800 per second, two per bit) is followed by
 * 234 bytes of 10 bits (start, 8 data bits LSB first, stop; not checked).  Frame = 6 header bytes + config block (frame number, id, one of 32
 * calibration rows) + PTU block (8 x 24-bit period counts) + GPS block (time of week, 12 PRNs, status, pseudo-range chips and delta chips)
 * + aux block, each with CRC-16 (0x1021, 0xFFFF), and 24 Reed-Solomon parity bytes over bytes 6..215 (RS(255,231), the RS41 code).
 * The sonde carries no position: the decoder solves it from the raw ranges with satellite orbits from a RINEX navigation file (-e) or an
 * SEM almanac (-a): closed form over every 4-satellite subset picked by GDOP (default) or Bancroft over all satellites (-g2), one
 * linearised correction, velocity from the delta chips (--vel / --vel1 / --vel2).  Printed as one line per frame, with --json the object
 * of frames whose config and GPS CRCs hold.
 *
 * The sample-rate part is the engine's generic sonde description (4800 Bd, two symbols per bit, BT 0.5, h 0.8 — 3.8 / 32 kHz IF for the
 * 1680 MHz RS92-NGP — 3 header errors accepted, bit offset 2, centre window 4 for IF-rate IQ); host/rs92mod.c puts the two together.
 *
 * Mirrors rs92mod.c: print_frame :1546-1575, rs92_ecc :1360-1385, print_position :1389-1544, get_FrameNb / get_SondeID / xor_ptu /
 * chk_toggle_type :297-545, get_Meas / get_PTU :565-647, get_GPStime :652-693, get_Aux :695-715, get_Cal :717-771, prn12 :777-843,
 * calc_satpos_alm / _rnx2 :845-959, get_pseudorange :975-1105, get_GPSvel / get_GPSkoord :1107-1351, Gps2Date :217-234, the bit loop of
 * main :1985-2050, the --rawhex loop :2058-2084; nav_gps_vel.c: read_SEMalmanac :132-179, read_RNXpephs :299-436, satellite clock /
 * position / velocity :444-677,1401-1694, NAV_ClosedFormSolution_FromPseudorange :682-874, calc_DOPn :880-991, NAV_bancroft1 :1082-1182,
 * NAV_LinP :1717-1796, NAV_LinV :1798-1891; for soft input find_softbinhead / corr_softhdb (demod_mod.c:1692-1762; threshold 0.8,
 * rs92mod.c:1976).
 */
#ifndef SONDE_RS92_H
#define SONDE_RS92_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_RS92_FRAME_LEN   240
#define SONDE_RS92_FRAME_BITS  2340      /* (240 - 6) bytes of 10 bits behind the header, rs92mod.c:81-85,2010 */

typedef struct sonde_rs92_dec sonde_rs92_dec_t;

typedef struct {
    int32_t raw;             /* -r: frame bytes as hex                                                                    */
    int32_t verbose;         /* -v = 1, -vv = 4                                                                           */
    int32_t aux;             /* -vx                                                                                       */
    int32_t ecc;             /* 1 = --ecc (also the default), 2 = --ecc2 / --json: number of corrected bytes printed      */
    int32_t ptu;             /* --ptu                                                                                     */
    int32_t inv;             /* -i: soft input only (push_soft); frames from a demodulator arrive in their final polarity */
    int32_t ngp;             /* --ngp (the calibration rows switch it when they say otherwise, :342-365)                  */
    int32_t dbg;             /* --dbg                                                                                     */
    int32_t json;            /* --json                                                                                    */
    int32_t gps_verbose;     /* -g1 = 1, -g2 = 2, -gg = 8                                                                 */
    int32_t gps_iter;        /* --iter                                                                                    */
    int32_t gps_vel;         /* --vel = 4 (also --json), --vel1 = 1, --vel2 = 2                                           */
    int32_t exsat;           /* --exsat <prn>; <= 0: none                                                                 */
    int32_t gpsepoch;        /* --gpsepoch <n> for the almanac's 10-bit week; < 0: the reference's default 1              */
    float   dop_limit;       /* --dop; <= 0: 9.9                                                                          */
    float   d_err;           /* --der; <= 0: 10000 without orbit data, 4000 with an almanac, 1000 with ephemerides        */
    int32_t jsn_freq_khz;    /* "freq" of the JSON when > 0                                                               */
    char    version[32];     /* "version" of the JSON; "" = omit                                                          */
    int32_t reserved[4];
} sonde_rs92_opts_t;

int  sonde_rs92_dec_create(const sonde_rs92_opts_t *opts, sonde_rs92_dec_t **out);
void sonde_rs92_dec_destroy(sonde_rs92_dec_t *d);

/* Orbit data (rs92mod.c:1834-1855): an SEM almanac (text) and / or a RINEX 2 navigation file; ephemerides win when both are loaded.
 * 0, or SONDE_E_ARG when the file cannot be opened / read as such (the reference goes on without positions then). */
int  sonde_rs92_dec_load_almanac(sonde_rs92_dec_t *d, const char *path);
int  sonde_rs92_dec_load_ephemeris(sonde_rs92_dec_t *d, const char *path);

/* One header hit from a demodulator: n (<= SONDE_RS92_FRAME_BITS) soft values of the bits behind the header (one per Manchester pair,
 * >= 0 = 1) in their final polarity.  A short frame (stream ended) is printed with the bytes that exist, :2030,2045.  Writes what the
 * reference prints NUL-terminated into out; returns its length or a negative SONDE_E_* code (SONDE_E_ARG also when out is too small:
 * -gg prints up to 495 lines per frame). */
int  sonde_rs92_dec_frame(sonde_rs92_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen);

/* One frame as bytes (`rs92mod --rawhex`, :2075-2081): len bytes from the start of the frame (header included). */
int  sonde_rs92_dec_bytes(sonde_rs92_dec_t *d, const uint8_t *frame, int32_t len, char *out, size_t outlen);

/* Soft-symbol input (`rs92mod --softin`): raw symbols at 4800 per second (two per bit); header search, polarity check against opts.inv and the bit loop
 * inside; finish != 0 at end of input (a frame in progress is printed with the bytes that exist). */
int  sonde_rs92_dec_push_soft(sonde_rs92_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
