/*
 * host/sonde_wideband.c — one wideband IQ stream -> every RS41 / DFM / M10 / M20 in it, in one process on one GPU.  C.
 *
 * The reference handles a wideband source with one detector process per candidate peak (auto_rx/autorx/scan.py:413-656: rtl_power peaks ->
 * `dft_detect` per peak) and then one decoder pipeline per sonde (decode.py:869-913, sdr_wrappers.py:270-371).  Here both steps run batched on
 * the same stream:
 *   scanner   `dft_detect --IQ fq --dc -c` on a frequency raster: sonde_scan.h with one channel per raster point, every channel mixing its own
 *             fq out of the shared stream (channel stride 0);
 *   decoders  per sonde type ONE `--IQ fq` engine (sonde_hip.h) with --slots channels that are handed out at run time: a detection takes a free
 *             channel, which gets the detected carrier (sonde_engine_tune_channel) and its own sample clock (sonde_engine_restart_channel: mixer
 *             table phase, IQ-DC schedule, decimator history and the IF-rate chain start over) — from then on the channel behaves like the
 *             `rs41mod --IQ fq` process auto_rx would have started at that moment, fed from the same chunks;
 *   telemetry the decoders' own bit-rate tiers (sonde_rs41.h / sonde_dfm.h / sonde_m10.h / sonde_m20.h): one JSON line per decoded frame, the
 *             object `rs41mod --json` / `dfm09mod --json` / `m10mod --json` / `m20mod --json` prints ("freq" = channel frequency in kHz).
 * A channel that has not delivered a frame for --release-s seconds of stream is ended and free again.
 *
 *     sonde_wideband [--cfreq Hz] [--raster Hz] [--slots N] [--release-s S] [--device D] [-v] - <sr> 16     < cs16 stream
 *     sonde_wideband --channelize [--chan-M 256] [--chan-D 200] [--chan-P 16] [...] - <sr> 16                 the form for wide streams (BASELINE configs[2])
 *
 * --channelize: the stream is not mixed once per raster point and once per sonde; a polyphase channelizer (sonde_chan.h: M channels at sr / D, one pass)
 * feeds the scanner (`dft_detect --iq --dc` on every channel) and, per sonde type, ONE IF-rate decoder engine whose channels are handed out at run
 * time: a detection takes a free channel, which is fine-tuned to the offset the scanner measured inside its channelizer channel (cfg.if_tune /
 * sonde_engine_tune_channel) and fed that channel's samples (sonde_chan_gather) from then on.  Every block costs one channelizer launch, one scanner
 * step and one launch sequence per sonde type, whatever the number of sondes (python -m radiosonde_auto_rx_amd.wideband --channelize is the same loop).
 *
 * Same arguments as `python -m radiosonde_auto_rx_amd.wideband`; -v logs detections and releases on stderr.  Exit 0 at EOF, 255 on error
 * (no GPU, bad arguments).  The generic-family types (LMS6, iMet-54, Meisei, MRZ, MTS01) get a generic-description engine per type and their own
 * bit-rate tiers (sonde_lms6.h ...): header hits + soft bits -> the JSON `lms6Xmod --json --ecc --vit2` / `imet54mod --json --ecc --ptu` / ... print.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_hip.h"
#include "sonde_scan.h"
#include "sonde_chan.h"
#include "sonde_rs41.h"
#include "sonde_dfm.h"
#include "sonde_m10.h"
#include "sonde_m20.h"
#include "sonde_lms6.h"
#include "sonde_meisei.h"
#include "sonde_imet54.h"
#include "sonde_mrz.h"
#include "sonde_mts01.h"
#include "sonde_rs92.h"

enum { T_RS41, T_DFM, T_M10, T_M20, T_LMS6, T_MEISEI, T_IMET5, T_MRZ, T_MTS01, T_RS92, T_LMSX, T_N };
static const char *kTypeName[T_N] = { "RS41", "DFM", "M10", "M20", "LMS6", "MEISEI", "IMET5", "MRZ", "MTS01", "RS92", "LMSX" };
#define IS_FAMILY(t) ((t) >= T_LMS6)
/* The generic family: what each decoder of the reference puts into dsp_t and passes to find_header() (sonde_generic_t, the same numbers as the
 * stand-alone front ends host/lms6Xmod.c ... and radiosonde_auto_rx_amd/family.py), the header threshold, whether either polarity is taken, whether
 * the decoder wants the bits as sent (raw) or in the polarity in effect, and how far apart two sondes of the type must be. */
typedef struct { const char *header; float baud, bt, h; int symlen, symhd, hdmax, bitofs, nbits; float l_win; int lpiq_bw, lpfm_bw; float thres; int aut, raw_pol; double sep_hz; float slice_baud; } family_t;
static const family_t kFamily[T_N] = {
    [T_LMS6]   = { "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110", 4800.0f, 1.2f, 0.9f, 1, 1, 10, 0, 261 * 16 - 80, -1.0f, 16000, 6000, 0.65f, 1, 1, 8000.0 },
    [T_MEISEI] = { "101010101011010100101011001101001100101011001101", 2400.0f, 1.2f, 2.4f, 1, 1, 1, 0, 1152, -1.0f, 16000, 4000, 0.7f, 1, 0, 12000.0 },
    [T_IMET5]  = { "0000000001" "0101010101" "0001001001" "0001001001", 4798.0f, 1.0f, 0.8f, 1, 1, 4, 1, 2200, 2.0f, 7400, 6000, 0.7f, 0, 0, 8000.0 },
    [T_MRZ]    = { "100110011001100110011001100110011001" "10101010", 2399.0f, 1.0f, 2.0f, 2, 2, 2, 2, 386, 2.0f, 9000, 6000, 0.76f, 0, 0, 10000.0 },
    [T_MTS01]  = { "10101010" "10101010" "10110100" "00101011", 1200.0f, 1.5f, 0.9f, 1, 1, 2, 0, 1048, 2.0f, 4000, 4000, 0.76f, 1, 1, 6000.0 },
    /* not a scanner type: an LMS6 whose blocks turn out to be LMS-X (lms6Xmod.c:1436-1462) moves here — same filters and header, 4797.8 Bd bit clock, 4720 bits per block */
    [T_LMSX]   = { "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110", 4800.0f, 1.2f, 0.9f, 1, 1, 10, 0, 300 * 16 - 80, -1.0f, 16000, 6000, 0.65f, 1, 1, 8000.0, 4797.8f },
    [T_RS92]   = { "10100110011001101001" "1010011001100110100110101010100110101001", 4800.0f, 0.5f, 0.8f, 2, 2, 3, 2, SONDE_RS92_FRAME_BITS, 4.0f, 8000, 6000, 0.7f, 0, 0, 8000.0 },
};

typedef struct {
    int used, type, slot;
    int chan;                            /* --channelize: the channelizer channel it sits in */
    double fq;                           /* carrier / sample rate, snapped to the mixer raster (--channelize: carrier in Hz relative to the stream centre / stream rate) */
    int khz;
    void *dec;                           /* sonde_<type>_dec_t */
    long frames; int64_t last_frame_at;  /* stream position (samples) of the last frame delivered */
    uint32_t last_pos;                   /* LMS6: header position of the previous block (frame rate) */
    int move_to;                         /* LMS6 / LMSX: 1 + the type whose engine it moves to at the end of this block of samples; 0 = stays */
    int moved;                           /* its first block on the new engine: no header position to take the frame rate from */
    double df;                           /* --channelize: offset from the channel centre, cycles per IF sample */
} sonde_t;

typedef struct { sonde_engine_t *eng; int *owner; void *d_rows; int32_t *rows; long calls; } group_t;       /* owner[slot] = index into g_sondes or -1; --channelize: the engine's input rows */

static int g_sr = 0, g_slots = 8, g_device = 0, g_verbose = 0;
static long long g_cfreq = 0; static int g_raster = 10000; static double g_release_s = 20.0;
static group_t g_gr[T_N];
static sonde_t *g_sondes = NULL; static int g_nsondes = 0, g_capsondes = 0;
static int64_t g_pos = 0;                                            /* samples consumed */
static int g_chunk = 0;
static char g_version[32] = "sonde_hip";
static const char *g_rs92_eph = NULL, *g_rs92_alm = NULL;        /* orbit data for RS92 positions (rs92mod -e / -a) */
static int g_fam_if_sr = 48000;                                      /* IF rate of the base-rate engines (raster form) */
/* --channelize */
static int g_channelize = 0, g_M = 256, g_D = 200, g_P = 16, g_if_sr = 0, g_nmax = 0;
static sonde_chan_t *g_chan = NULL;
static double g_spacing = 0.0;

static double snap_fq(double fq, int sr) { return (double)(long long)llround(fq * sr / 16.0) * 16.0 / sr; }      /* demod_mod.c:1265-1288 where 16 divides sr */

static int group_engine(int type) {
    group_t *g = &g_gr[type];
    if (g->eng) return 0;
    sonde_cfg_t c; memset(&c, 0, sizeof c);
    c.abi_version = SONDE_ABI_VERSION; c.device = g_device; c.n_channels = g_slots; c.sample_rate = g_sr; c.bits = 16;
    c.sonde_type = type == T_RS41 ? SONDE_RS41 : type == T_DFM ? SONDE_DFM09 : type == T_M10 ? SONDE_M10 : SONDE_M20;
    c.opt_lp = SONDE_LP_IQ; c.ecc_level = type == T_DFM ? 1 : 2; c.opt_auto = type == T_DFM;
    c.max_chunk = g_chunk; c.max_frames = 16 * g_slots; c.input = SONDE_IN_IQ;
    if (g_channelize) {                                             /* IF-rate float32 IQ from the channelizer, fine-tuned per channel */
        c.sample_rate = g_if_sr; c.bits = 32; c.input = SONDE_IN_IFIQ3; c.if_tune = 1; c.max_chunk = g_nmax; c.max_frames = 8 * g_slots;
    }
    double *fq = (double *)calloc((size_t)g_slots, sizeof(double));
    if (!fq) return SONDE_E_NOMEM;
    int rc;
    if (IS_FAMILY(type)) {                                          /* header hits + soft bits; the type's bit-rate tier decodes them */
        const family_t *f = &kFamily[type];
        sonde_generic_t gd; memset(&gd, 0, sizeof gd);
        snprintf(gd.header, sizeof gd.header, "%s", f->header);
        gd.baud = f->baud; gd.bt = f->bt; gd.h = f->h; gd.symlen = f->symlen; gd.symhd = f->symhd; gd.hdmax = f->hdmax; gd.bitofs = f->bitofs;
        gd.nbits = f->nbits; gd.l_win = f->l_win; gd.lpiq_bw = f->lpiq_bw; gd.lpfm_bw = f->lpfm_bw; gd.slice_baud = f->slice_baud;
        c.sonde_type = SONDE_GENERIC; c.keep_soft = 1; c.thres = f->thres; c.opt_auto = f->aut; c.ecc_level = 0; c.max_frames = 8 * g_slots;
        rc = sonde_engine_create_generic(&c, fq, &gd, &g->eng);
    } else rc = sonde_engine_create(&c, fq, &g->eng);
    free(fq);
    if (rc < 0) return rc;
    g->owner = (int *)malloc(sizeof(int) * (size_t)g_slots);
    if (!g->owner) return SONDE_E_NOMEM;
    for (int s = 0; s < g_slots; s++) g->owner[s] = -1;
    if (g_channelize) {
        g->rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)g_slots);
        if (!g->rows) return SONDE_E_NOMEM;
        const int rc2 = sonde_chan_rows_alloc(g_chan, g_slots, &g->d_rows);
        if (rc2 < 0) return rc2;
    }
    return 0;
}

static void *make_decoder(int type, int khz) {
    void *d = NULL;
    if (type == T_RS41) { sonde_rs41_opts_t o; memset(&o, 0, sizeof o); o.ptu = 2; o.json = 1; o.silent = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
                          if (sonde_rs41_dec_create(&o, (sonde_rs41_dec_t **)&d) < 0) return NULL; }
    else if (type == T_DFM) { sonde_dfm_opts_t o; memset(&o, 0, sizeof o); o.verbose = 2; o.ptu = 1; o.ecc = 1; o.dist = 1; o.json = 1; o.opt_auto = 1; o.jsn_freq_khz = khz;
                              snprintf(o.version, sizeof o.version, "%s", g_version); if (sonde_dfm_dec_create(&o, (sonde_dfm_dec_t **)&d) < 0) return NULL; }
    else if (type == T_M10) { sonde_m10_opts_t o; memset(&o, 0, sizeof o); o.verbose = 1; o.ptu = 1; o.json = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
                              if (sonde_m10_dec_create(&o, (sonde_m10_dec_t **)&d) < 0) return NULL; }
    else if (type == T_M20) { sonde_m20_opts_t o; memset(&o, 0, sizeof o); o.verbose = 1; o.ptu = 1; o.json = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_m20_dec_create(&o, (sonde_m20_dec_t **)&d) < 0) return NULL; }
    else if (type == T_LMS6 || type == T_LMSX) { sonde_lms6_opts_t o; memset(&o, 0, sizeof o); o.ecc = 1; o.vit = 2; o.json = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_lms6_dec_create(&o, (sonde_lms6_dec_t **)&d) < 0) return NULL; }
    else if (type == T_MEISEI) { sonde_meisei_opts_t o; memset(&o, 0, sizeof o); o.ecc = 1; o.json = 1; o.ptu = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_meisei_dec_create(&o, (sonde_meisei_dec_t **)&d) < 0) return NULL; }
    else if (type == T_IMET5) { sonde_imet54_opts_t o; memset(&o, 0, sizeof o); o.ecc = 1; o.json = 1; o.ptu = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_imet54_dec_create(&o, (sonde_imet54_dec_t **)&d) < 0) return NULL; }
    else if (type == T_MRZ) { sonde_mrz_opts_t o; memset(&o, 0, sizeof o); o.json = 1; o.ptu = 1; o.uniq = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_mrz_dec_create(&o, (sonde_mrz_dec_t **)&d) < 0) return NULL; }
    else if (type == T_RS92) { sonde_rs92_opts_t o; memset(&o, 0, sizeof o); o.verbose = 1; o.aux = 1; o.ecc = 2; o.gps_vel = 4; o.json = 1; o.gpsepoch = -1; o.jsn_freq_khz = khz;     /* rs92mod -vx -v --crc --ecc --vel --json (decode.py:484) */
           snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_rs92_dec_create(&o, (sonde_rs92_dec_t **)&d) < 0) return NULL;
           if (g_rs92_alm && sonde_rs92_dec_load_almanac((sonde_rs92_dec_t *)d, g_rs92_alm) < 0) fprintf(stderr, "rs92: almanac %s not readable as such\n", g_rs92_alm);
           if (g_rs92_eph && sonde_rs92_dec_load_ephemeris((sonde_rs92_dec_t *)d, g_rs92_eph) < 0) fprintf(stderr, "rs92: ephemeris %s not readable as such\n", g_rs92_eph); }
    else { sonde_mts01_opts_t o; memset(&o, 0, sizeof o); o.json = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_mts01_dec_create(&o, (sonde_mts01_dec_t **)&d) < 0) return NULL; }
    return d;
}
static void free_decoder(int type, void *d) {
    if (!d) return;
    if (type == T_RS41) sonde_rs41_dec_destroy((sonde_rs41_dec_t *)d); else if (type == T_DFM) sonde_dfm_dec_destroy((sonde_dfm_dec_t *)d);
    else if (type == T_M10) sonde_m10_dec_destroy((sonde_m10_dec_t *)d); else if (type == T_M20) sonde_m20_dec_destroy((sonde_m20_dec_t *)d);
    else if (type == T_LMS6 || type == T_LMSX) sonde_lms6_dec_destroy((sonde_lms6_dec_t *)d); else if (type == T_MEISEI) sonde_meisei_dec_destroy((sonde_meisei_dec_t *)d);
    else if (type == T_IMET5) sonde_imet54_dec_destroy((sonde_imet54_dec_t *)d); else if (type == T_MRZ) sonde_mrz_dec_destroy((sonde_mrz_dec_t *)d);
    else if (type == T_RS92) sonde_rs92_dec_destroy((sonde_rs92_dec_t *)d);
    else sonde_mts01_dec_destroy((sonde_mts01_dec_t *)d);
}

/* the JSON lines of a decoder's output text (it may also hold the decoder's text line) */
static void print_json_lines(const char *tx) {
    for (const char *p = tx; *p;) {
        const char *e = strchr(p, '\n'); const size_t n = e ? (size_t)(e - p) : strlen(p);
        if (n && p[0] == '{') { fwrite(p, 1, n, stdout); fputc('\n', stdout); }
        p += n + (e ? 1 : 0);
    }
}

static void start_sonde(int type, double fq_found) {
    const double merge_hz = 6000.0 * ((type == T_M10 || type == T_M20) ? 3.0 : 1.0);       /* 9.6 kBd: seen from neighbouring raster points too */
    if (IS_FAMILY(type) && g_gr[type].eng == NULL && g_verbose) fprintf(stderr, "engine: %s (generic description)\n", kTypeName[type]);
    for (int i = 0; i < g_nsondes; i++) if (g_sondes[i].used && fabs(g_sondes[i].fq - fq_found) * g_sr < merge_hz) return;
    const double fq = snap_fq(fq_found, g_sr);
    if (group_engine(type) < 0) { fprintf(stderr, "sonde_wideband: no engine for %s\n", kTypeName[type]); return; }
    group_t *g = &g_gr[type];
    int slot = -1;
    for (int s = 0; s < g_slots; s++) if (g->owner[s] < 0) { slot = s; break; }
    if (slot < 0) { if (g_verbose) fprintf(stderr, "no free channel: %s %+.0f Hz\n", kTypeName[type], fq * g_sr); return; }
    if (sonde_engine_tune_channel(g->eng, slot, fq) < 0 || sonde_engine_restart_channel(g->eng, slot) < 0) { fprintf(stderr, "sonde_wideband: channel set-up failed\n"); return; }
    int idx = -1;
    for (int i = 0; i < g_nsondes; i++) if (!g_sondes[i].used) { idx = i; break; }
    if (idx < 0) {
        if (g_nsondes == g_capsondes) {
            const int cap = g_capsondes ? 2 * g_capsondes : 32;
            sonde_t *p = (sonde_t *)realloc(g_sondes, sizeof(sonde_t) * (size_t)cap);
            if (!p) return;
            g_sondes = p; g_capsondes = cap;
        }
        idx = g_nsondes++;
    }
    sonde_t *s = &g_sondes[idx];
    memset(s, 0, sizeof *s);
    s->used = 1; s->type = type; s->slot = slot; s->fq = fq; s->last_frame_at = g_pos;
    s->khz = g_cfreq ? (int)llround((g_cfreq + fq * g_sr) / 1000.0) : 0;
    s->dec = make_decoder(type, s->khz);
    if (!s->dec) { s->used = 0; return; }
    g->owner[slot] = idx;
    if (g_verbose) fprintf(stderr, "detected: %s %+.0f Hz (%d kHz) -> channel %d\n", kTypeName[type], fq * g_sr, s->khz, slot);
}

/* --channelize: a detection in channelizer channel k, df cycles per IF sample off its centre */
static void start_sonde_chan(int type, int k, double df) {
    const double f_hz = (double)(k < g_M / 2 ? k : k - g_M) * g_spacing + df * g_if_sr;
    const double sep = IS_FAMILY(type) ? kFamily[type].sep_hz : (type == T_M10 || type == T_M20) ? 20000.0 : 8000.0;     /* the neighbouring channel sees a strong signal too */
    for (int i = 0; i < g_nsondes; i++) if (g_sondes[i].used && (g_sondes[i].type == T_LMSX ? T_LMS6 : g_sondes[i].type) == type && fabs(g_sondes[i].fq * g_sr - f_hz) < sep) return;
    if (group_engine(type) < 0) { fprintf(stderr, "sonde_wideband: no engine for %s\n", kTypeName[type]); return; }
    group_t *g = &g_gr[type];
    int slot = -1;
    for (int s = 0; s < g_slots; s++) if (g->owner[s] < 0) { slot = s; break; }
    if (slot < 0) { if (g_verbose) fprintf(stderr, "no free channel: %s %+.0f Hz\n", kTypeName[type], f_hz); return; }
    if ((g->calls && sonde_engine_restart_channel(g->eng, slot) < 0) || sonde_engine_tune_channel(g->eng, slot, df) < 0) { fprintf(stderr, "sonde_wideband: channel set-up failed\n"); return; }
    int idx = -1;
    for (int i = 0; i < g_nsondes; i++) if (!g_sondes[i].used) { idx = i; break; }
    if (idx < 0) {
        if (g_nsondes == g_capsondes) {
            const int cap = g_capsondes ? 2 * g_capsondes : 32;
            sonde_t *p = (sonde_t *)realloc(g_sondes, sizeof(sonde_t) * (size_t)cap);
            if (!p) return;
            g_sondes = p; g_capsondes = cap;
        }
        idx = g_nsondes++;
    }
    sonde_t *s = &g_sondes[idx];
    memset(s, 0, sizeof *s);
    s->used = 1; s->type = type; s->slot = slot; s->chan = k; s->df = df; s->fq = f_hz / g_sr; s->last_frame_at = g_pos;
    s->khz = g_cfreq ? (int)llround((g_cfreq + f_hz) / 1000.0) : 0;
    s->dec = make_decoder(type, s->khz);
    if (!s->dec) { s->used = 0; return; }
    g->owner[slot] = idx;
    if (g_verbose) fprintf(stderr, "detected: %s %+.0f Hz (%d kHz) in channel %d -> decoder channel %d\n", kTypeName[type], f_hz, s->khz, k, slot);
}

static void release_sonde(int idx) {
    sonde_t *s = &g_sondes[idx];
    group_t *g = &g_gr[s->type];
    sonde_engine_finish_channel(g->eng, s->slot);          /* whatever it still emits is drained (and dropped) with the next fetch */
    g->owner[s->slot] = -1;
    if (g_verbose) fprintf(stderr, "released: %s %d kHz channel %d after %ld frames\n", kTypeName[s->type], s->khz, s->slot, s->frames);
    free_decoder(s->type, s->dec);
    s->used = 0; s->dec = NULL;
}

/* an LMS6 whose decoder found LMS-X blocks (or the reverse): the sonde moves to a channel of the engine of the other description, decoder object and all;
 * the next block of samples is the first it sees there (the stand-alone decoder replays its input from the end of the block instead, host/lms6Xmod.c) */
static void move_sondes(void) {
    for (int i = 0; i < g_nsondes; i++) {
        sonde_t *s = &g_sondes[i];
        if (!s->used || !s->move_to) continue;
        const int want = s->move_to - 1;
        s->move_to = 0;
        group_t *g0 = &g_gr[s->type];
        sonde_engine_finish_channel(g0->eng, s->slot);
        g0->owner[s->slot] = -1;
        int slot = -1;
        if (group_engine(want) == 0) for (int k = 0; k < g_slots; k++) if (g_gr[want].owner[k] < 0) { slot = k; break; }
        group_t *g = &g_gr[want];
        int ok = slot >= 0;
        if (ok && g_channelize) ok = !(g->calls && sonde_engine_restart_channel(g->eng, slot) < 0) && sonde_engine_tune_channel(g->eng, slot, s->df) >= 0;
        else if (ok) ok = sonde_engine_tune_channel(g->eng, slot, s->fq) >= 0 && sonde_engine_restart_channel(g->eng, slot) >= 0;
        if (!ok) {
            if (g_verbose) fprintf(stderr, "no free channel: %s %d kHz\n", kTypeName[want], s->khz);
            free_decoder(s->type, s->dec); s->used = 0; s->dec = NULL;
            continue;
        }
        if (g_verbose) fprintf(stderr, "retuned: %s -> %s %d kHz -> channel %d\n", kTypeName[s->type], kTypeName[want], s->khz, slot);
        s->type = want; s->slot = slot; s->moved = 1;
        g->owner[slot] = i;
    }
}

/* frames the engine of one type has ready -> the decoders of the sondes that own the channels */
static void drain(int type, int finish) {
    group_t *g = &g_gr[type];
    if (!g->eng) return;
    static char tx[1 << 16];
    if (IS_FAMILY(type)) {                                          /* header hits -> the type's bit-rate tier, one decoder object per sonde */
        const family_t *f = &kFamily[type];
        const int if_sr = g_channelize ? g_if_sr : g_fam_if_sr;
        static sonde_hit_t hits[64];
        static float *soft = NULL; static size_t soft_cap = 0;
        for (;;) {
            const int k = sonde_engine_fetch_hits(g->eng, hits, 64, finish);
            finish = 0;
            if (k <= 0) break;
            if ((size_t)k * (size_t)f->nbits > soft_cap) { float *p = (float *)realloc(soft, sizeof(float) * (size_t)k * (size_t)f->nbits); if (!p) return; soft = p; soft_cap = (size_t)k * (size_t)f->nbits; }
            if (sonde_engine_fetch_soft(g->eng, soft, k) < 0) return;
            for (int i = 0; i < k; i++) {
                const int o = (hits[i].channel >= 0 && hits[i].channel < g_slots) ? g->owner[hits[i].channel] : -1;
                if (o < 0) continue;
                sonde_t *sn = &g_sondes[o];
                float *b = soft + (size_t)i * (size_t)f->nbits;
                int n = hits[i].nbits;
                if (f->raw_pol && hits[i].mv < 0.f) for (int j = 0; j < n; j++) b[j] = -b[j];      /* stored in the polarity in effect; this decoder reads the bits as sent */
                int m = 0;
                if (type == T_LMS6 || type == T_LMSX) {
                    const int want = sonde_lms6_dec_block_bits((sonde_lms6_dec_t *)sn->dec); if (n > want) n = want;
                    const uint32_t d = hits[i].mv_pos - sn->last_pos;
                    float rate = d ? (float)(4800.0 * if_sr / (double)d) : INFINITY;
                    if (sn->moved) { rate = 4800.0f; sn->moved = 0; }                  /* first block after the change of engine: the nominal rate (a rate outside 4000..5000 would send the decoder back, lms6Xmod.c:959) */
                    sn->last_pos = hits[i].mv_pos;
                    m = sonde_lms6_dec_block((sonde_lms6_dec_t *)sn->dec, b, NULL, n, hits[i].mv, rate, ((double)hits[i].mv_pos + n * (double)if_sr / 4800.0) / if_sr, tx, sizeof tx);
                    int32_t changed = 0;
                    const int now = (sonde_lms6_dec_type((sonde_lms6_dec_t *)sn->dec, &changed) & 0xFF) == 10 ? T_LMSX : T_LMS6;
                    if (changed && now != type) sn->move_to = 1 + now;                              /* at the end of this block of samples (move_sondes) */
                }
                else if (type == T_MEISEI) m = sonde_meisei_dec_frame((sonde_meisei_dec_t *)sn->dec, b, n, tx, sizeof tx);
                else if (type == T_IMET5) m = sonde_imet54_dec_frame((sonde_imet54_dec_t *)sn->dec, b, n, tx, sizeof tx);
                else if (type == T_MRZ) { const int want = sonde_mrz_dec_frame_bits((sonde_mrz_dec_t *)sn->dec); if (n > want) n = want; m = sonde_mrz_dec_frame((sonde_mrz_dec_t *)sn->dec, b, n, tx, sizeof tx); }
                else if (type == T_RS92) m = sonde_rs92_dec_frame((sonde_rs92_dec_t *)sn->dec, b, n, tx, sizeof tx);
                else m = sonde_mts01_dec_frame((sonde_mts01_dec_t *)sn->dec, b, n, tx, sizeof tx);
                if (m > 0) { if ((size_t)m < sizeof tx) tx[m] = 0; else tx[sizeof tx - 1] = 0; print_json_lines(tx); }
                sn->frames++; if (m > 0) sn->last_frame_at = g_pos;       /* only a frame the decoder accepted counts as a sign of life */
            }
        }
        fflush(stdout);
        return;
    }
    for (;;) {
        int k = 0;
        if (type == T_RS41) {
            static sonde_frame_t fr[64];
            k = finish ? sonde_engine_finish(g->eng, fr, 64) : sonde_engine_fetch_frames(g->eng, fr, 64);
            for (int i = 0; i < k; i++) {
                const int o = (fr[i].channel >= 0 && fr[i].channel < g_slots) ? g->owner[fr[i].channel] : -1;
                if (o < 0) continue;
                fr[i].channel = 0;
                if (sonde_rs41_dec_frame((sonde_rs41_dec_t *)g_sondes[o].dec, &fr[i], tx, sizeof tx) > 0) print_json_lines(tx);
                g_sondes[o].frames++; if (fr[i].ecc >= 0) g_sondes[o].last_frame_at = g_pos;      /* the Reed-Solomon code accepts the frame */
            }
        } else if (type == T_DFM) {
            static sonde_dfm_frame_t fr[64];
            k = sonde_engine_fetch_dfm(g->eng, fr, 64, finish);
            for (int i = 0; i < k; i++) {
                const int o = (fr[i].channel >= 0 && fr[i].channel < g_slots) ? g->owner[fr[i].channel] : -1;
                if (o < 0) continue;
                fr[i].channel = 0;
                if (sonde_dfm_dec_frame((sonde_dfm_dec_t *)g_sondes[o].dec, &fr[i], tx, sizeof tx) > 0) print_json_lines(tx);
                g_sondes[o].frames++; if (fr[i].ecc[0] >= 0 && fr[i].ecc[1] >= 0 && fr[i].ecc[2] >= 0) g_sondes[o].last_frame_at = g_pos;   /* all three Hamming blocks */
            }
        } else if (type == T_M10) {
            static sonde_m10_frame_t fr[64];
            k = sonde_engine_fetch_m10(g->eng, fr, 64, finish);
            for (int i = 0; i < k; i++) {
                const int o = (fr[i].channel >= 0 && fr[i].channel < g_slots) ? g->owner[fr[i].channel] : -1;
                if (o < 0) continue;
                fr[i].channel = 0;
                if (sonde_m10_dec_frame((sonde_m10_dec_t *)g_sondes[o].dec, &fr[i], tx, sizeof tx) > 0) print_json_lines(tx);
                g_sondes[o].frames++; if (fr[i].cs_ok) g_sondes[o].last_frame_at = g_pos;
            }
        } else {
            static sonde_m20_frame_t fr[64];
            k = sonde_engine_fetch_m20(g->eng, fr, 64, finish);
            for (int i = 0; i < k; i++) {
                const int o = (fr[i].channel >= 0 && fr[i].channel < g_slots) ? g->owner[fr[i].channel] : -1;
                if (o < 0) continue;
                fr[i].channel = 0;
                if (sonde_m20_dec_frame((sonde_m20_dec_t *)g_sondes[o].dec, &fr[i], tx, sizeof tx) > 0) print_json_lines(tx);
                g_sondes[o].frames++; if (fr[i].cs_ok) g_sondes[o].last_frame_at = g_pos;
            }
        }
        finish = 0;
        if (k <= 0) break;
    }
    fflush(stdout);
}

static void on_detection(const sonde_detection_t *d, int channelized, const double *raster) {
    const double fq = channelized ? 0.0 : raster[d->channel] + d->df;
    int type = -1;
    if (!strcmp(d->type, "RS41")) { if (d->score > 0) type = T_RS41; else return; }
    else if (!strcmp(d->type, "DFM9")) type = T_DFM;                 /* either polarity: the decoder runs with --auto */
    else if (!strcmp(d->type, "M10")) type = T_M10;                  /* differential code: polarity does not matter */
    else if (!strcmp(d->type, "M20")) type = T_M20;
    else {                                                            /* the generic family: positive score, or either polarity where the decoder takes both */
        static const struct { const char *name; int t; } fam[] = { { "LMS6", T_LMS6 }, { "MEISEI", T_MEISEI }, { "IMET5", T_IMET5 }, { "MRZ", T_MRZ }, { "MTS01", T_MTS01 }, { "RS92", T_RS92 } };
        for (unsigned i = 0; i < sizeof fam / sizeof fam[0]; i++) if (!strcmp(d->type, fam[i].name)) { if (d->score > 0 || kFamily[fam[i].t].aut) type = fam[i].t; else return; }
    }
    if (type < 0) { if (g_verbose) fprintf(stderr, "seen: %s %.4f in %s %d (decoder: the type's stand-alone front end)\n", d->type, d->score, channelized ? "channel" : "raster point", d->channel); return; }
    if (channelized) start_sonde_chan(type, d->channel, d->df); else start_sonde(type, fq);
}

/* --channelize: channelizer -> scanner on every channel -> per type one IF-rate engine with run-time channels */
static int run_channelized(void) {
    sonde_chan_cfg_t cc; memset(&cc, 0, sizeof cc);
    cc.abi_version = SONDE_ABI_VERSION; cc.device = g_device; cc.sample_rate = g_sr; cc.M = g_M; cc.D = g_D; cc.P = g_P;
    g_chunk = g_sr / 4; g_chunk -= g_chunk % g_D;
    cc.max_chunk = g_chunk;
    int rc = sonde_chan_create(&cc, &g_chan);
    if (rc < 0) { fprintf(stderr, "sonde_wideband: channelizer: %s\n", sonde_strerror(rc)); return 255; }
    sonde_chan_info_t ci; sonde_chan_info(g_chan, &ci);
    if (ci.out_rate_den < 1 || ci.out_rate_num % ci.out_rate_den) { fprintf(stderr, "sonde_wideband: sr / D must be an integer rate\n"); return 255; }
    g_if_sr = ci.out_rate_num / ci.out_rate_den; g_nmax = ci.max_frames; g_spacing = (double)g_sr / g_M;
    void *d_out = NULL; int64_t stride = 0;
    if (sonde_chan_output(g_chan, &d_out, &stride) < 0) return 255;
    sonde_scan_cfg_t sc; memset(&sc, 0, sizeof sc);
    sc.abi_version = SONDE_ABI_VERSION; sc.device = g_device; sc.n_channels = g_M; sc.sample_rate = g_if_sr; sc.bits = 32; sc.iq_mode = SONDE_SCAN_IFIQ;
    sc.opt_dc = 1; sc.opt_cont = 1; sc.audio_channels = 1; sc.max_chunk = g_nmax;
    double *zero = (double *)calloc((size_t)g_M, sizeof(double));
    sonde_scan_t *scan = NULL;
    rc = zero ? sonde_scan_create(&sc, zero, &scan) : SONDE_E_NOMEM;
    free(zero);
    if (rc < 0) { fprintf(stderr, "sonde_wideband: scanner: %s\n", sonde_strerror(rc)); return 255; }
    int16_t *buf = (int16_t *)malloc((size_t)g_chunk * 4);
    if (!buf) return 255;
    size_t have = 0;
    int eof = 0;
    while (!eof) {
        while (have < (size_t)g_chunk * 4) {
            const size_t k = fread((char *)buf + have, 1, (size_t)g_chunk * 4 - have, stdin);
            if (!k) { eof = 1; break; }
            have += k;
        }
        int n = (int)(have / 4); n -= n % g_D;
        if (n > 0) {
            const int m = sonde_chan_process_host(g_chan, buf, n, d_out, stride);            /* m IF samples per channel */
            if (m < 0 || sonde_chan_sync(g_chan) < 0) { fprintf(stderr, "sonde_wideband: channelizer: %s\n", sonde_strerror(m < 0 ? m : SONDE_E_NOGPU)); return 255; }
            if (m > 0) {
                rc = sonde_scan_process_device(scan, d_out, stride, m);
                if (rc < 0) { fprintf(stderr, "sonde_wideband: scanner: %s\n", sonde_strerror(rc)); return 255; }
                sonde_detection_t det[64];
                for (;;) {
                    const int k = sonde_scan_fetch(scan, det, 64);
                    for (int i = 0; i < k; i++) on_detection(&det[i], 1, NULL);
                    if (k < 64) break;
                }
                for (int t = 0; t < T_N; t++) {                      /* every decoder channel gets its channelizer channel's samples of this block */
                    group_t *g = &g_gr[t];
                    if (!g->eng) continue;
                    for (int sl = 0; sl < g_slots; sl++) g->rows[sl] = g->owner[sl] >= 0 ? g_sondes[g->owner[sl]].chan : -1;
                    if (sonde_chan_gather(g_chan, d_out, stride, g->rows, g_slots, m, g->d_rows) < 0) { fprintf(stderr, "sonde_wideband: gather failed\n"); return 255; }
                }
                if (sonde_chan_sync(g_chan) < 0) return 255;           /* the copies ran on the channelizer's stream, the engines have their own */
                for (int t = 0; t < T_N; t++) {
                    group_t *g = &g_gr[t];
                    if (!g->eng) continue;
                    rc = sonde_engine_process_device(g->eng, g->d_rows, g_nmax, m);
                    if (rc < 0) { fprintf(stderr, "sonde_wideband: %s engine: %s\n", kTypeName[t], sonde_strerror(rc)); return 255; }
                    g->calls++;
                }
                g_pos += m;
                for (int t = 0; t < T_N; t++) drain(t, 0);
                move_sondes();
                for (int i = 0; i < g_nsondes; i++)
                    if (g_sondes[i].used && g_release_s > 0 && (double)(g_pos - g_sondes[i].last_frame_at) > g_release_s * g_if_sr) release_sonde(i);
            }
        }
        const size_t rest = have - (size_t)n * 4;
        memmove(buf, (char *)buf + (size_t)n * 4, rest);
        have = rest;
    }
    for (int t = 0; t < T_N; t++) drain(t, 1);
    for (int i = 0; i < g_nsondes; i++) if (g_sondes[i].used) free_decoder(g_sondes[i].type, g_sondes[i].dec);
    for (int t = 0; t < T_N; t++) { if (g_gr[t].eng) sonde_engine_destroy(g_gr[t].eng); free(g_gr[t].owner); free(g_gr[t].rows); }
    sonde_scan_destroy(scan);
    sonde_chan_destroy(g_chan);
    free(buf); free(g_sondes);
    return 0;
}

int main(int argc, char **argv) {
    int ai = 1;
    for (; ai < argc; ai++) {
        if (!strcmp(argv[ai], "--cfreq") && ai + 1 < argc) g_cfreq = atoll(argv[++ai]);
        else if (!strcmp(argv[ai], "--raster") && ai + 1 < argc) g_raster = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--slots") && ai + 1 < argc) g_slots = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--release-s") && ai + 1 < argc) g_release_s = atof(argv[++ai]);
        else if (!strcmp(argv[ai], "--device") && ai + 1 < argc) g_device = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "-v")) g_verbose = 1;
        else if (!strcmp(argv[ai], "--channelize")) g_channelize = 1;
        else if (!strcmp(argv[ai], "--rs92-ephem") && ai + 1 < argc) g_rs92_eph = argv[++ai];
        else if (!strcmp(argv[ai], "--rs92-alm") && ai + 1 < argc) g_rs92_alm = argv[++ai];
        else if (!strcmp(argv[ai], "--chan-M") && ai + 1 < argc) g_M = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--chan-D") && ai + 1 < argc) g_D = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--chan-P") && ai + 1 < argc) g_P = atoi(argv[++ai]);
        else break;
    }
    if (argc - ai != 3 || strcmp(argv[ai], "-") || atoi(argv[ai + 2]) != 16 || atoi(argv[ai + 1]) < 48000 || g_raster < 100 || g_slots < 1 || g_slots > 256) {
        fprintf(stderr, "usage: %s [--channelize [--chan-M 256] [--chan-D 200] [--chan-P 16]] [--cfreq Hz] [--raster Hz] [--slots N] [--release-s S] [--rs92-ephem rinex_nav] [--rs92-alm sem_almanac] [--device D] [-v] - <sr> 16   (cs16 on stdin)\n", argv[0]);
        return 255;
    }
    g_sr = atoi(argv[ai + 1]);
    if (getenv("SONDE_JSN_VERSION")) snprintf(g_version, sizeof g_version, "%s", getenv("SONDE_JSN_VERSION"));
    if (g_channelize) return run_channelized();
    /* raster of the scanner: every `raster` Hz over +-0.45 of the sample rate, snapped like an --IQ argument */
    const int kmax = (int)(0.45 * g_sr / g_raster), nr = 2 * kmax + 1;
    double *raster = (double *)malloc(sizeof(double) * (size_t)nr);
    if (!raster) return 255;
    for (int k = -kmax; k <= kmax; k++) raster[k + kmax] = snap_fq((double)k * g_raster / g_sr, g_sr);
    g_chunk = g_sr / 4;
    sonde_scan_cfg_t sc; memset(&sc, 0, sizeof sc);
    sc.abi_version = SONDE_ABI_VERSION; sc.device = g_device; sc.n_channels = nr; sc.sample_rate = g_sr; sc.bits = 16; sc.iq_mode = SONDE_SCAN_BBIQ;
    sc.opt_dc = 1; sc.opt_cont = 1; sc.audio_channels = 1; sc.max_chunk = g_chunk;
    sonde_scan_t *scan = NULL;
    int rc = sonde_scan_create(&sc, raster, &scan);
    if (rc < 0) { fprintf(stderr, "sonde_wideband: scanner: %s\n", sonde_strerror(rc)); return 255; }
    sonde_scan_info_t si; sonde_scan_info(scan, &si);
    /* calls are cut at multiples of the scanner's and the decoders' decimation factors (the decoders decimate to the reference's IF rate: 48 kHz,
     * raised until it divides the sample rate, demod_mod.c:1229-1236); what is left of a read waits for the next one */
    int if_sr = g_sr < 48000 ? g_sr : 48000; while (g_sr % if_sr) if_sr++;
    g_fam_if_sr = if_sr;
    const int d1 = si.decM, d2 = g_sr / if_sr;
    int a = d1, b = d2; while (b) { const int t = a % b; a = b; b = t; }
    const int align = d1 / a * d2;
    g_chunk -= g_chunk % align;
    if (g_chunk < align) { fprintf(stderr, "sonde_wideband: sample rate too low for this raster\n"); return 255; }
    int16_t *buf = (int16_t *)malloc((size_t)g_chunk * 4);
    if (!buf) return 255;
    size_t have = 0;                       /* bytes in buf */
    int eof = 0;
    while (!eof) {
        while (have < (size_t)g_chunk * 4) {
            const size_t k = fread((char *)buf + have, 1, (size_t)g_chunk * 4 - have, stdin);
            if (!k) { eof = 1; break; }
            have += k;
        }
        int n = (int)(have / 4); n -= n % align;
        if (n > 0) {
            rc = sonde_scan_process_host(scan, buf, 0, n);
            if (rc < 0) { fprintf(stderr, "sonde_wideband: scanner: %s\n", sonde_strerror(rc)); return 255; }
            sonde_detection_t det[64];
            for (;;) {
                const int k = sonde_scan_fetch(scan, det, 64);
                for (int i = 0; i < k; i++) on_detection(&det[i], 0, raster);
                if (k < 64) break;
            }
            for (int t = 0; t < T_N; t++) {
                if (!g_gr[t].eng) continue;
                rc = sonde_engine_process_host(g_gr[t].eng, buf, 0, n);
                if (rc < 0) { fprintf(stderr, "sonde_wideband: %s engine: %s\n", kTypeName[t], sonde_strerror(rc)); return 255; }
            }
            g_pos += n;
            for (int t = 0; t < T_N; t++) drain(t, 0);
            move_sondes();
            for (int i = 0; i < g_nsondes; i++)
                if (g_sondes[i].used && g_release_s > 0 && (double)(g_pos - g_sondes[i].last_frame_at) > g_release_s * g_sr) release_sonde(i);
        }
        const size_t rest = have - (size_t)n * 4;
        memmove(buf, (char *)buf + (size_t)n * 4, rest);
        have = rest;
        if (eof) break;
    }
    for (int t = 0; t < T_N; t++) drain(t, 1);            /* the frames in progress at EOF are still due */
    for (int i = 0; i < g_nsondes; i++) if (g_sondes[i].used) free_decoder(g_sondes[i].type, g_sondes[i].dec);
    for (int t = 0; t < T_N; t++) { if (g_gr[t].eng) sonde_engine_destroy(g_gr[t].eng); free(g_gr[t].owner); }
    sonde_scan_destroy(scan);
    free(buf); free(raster); free(g_sondes);
    return 0;
}
