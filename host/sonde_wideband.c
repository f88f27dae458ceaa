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
 *
 * Same arguments as `python -m radiosonde_auto_rx_amd.wideband`; -v logs detections and releases on stderr.  Exit 0 at EOF, 255 on error
 * (no GPU, bad arguments).  The generic-family types (LMS6, iMet-54, Meisei, MRZ, MTS01) are listed on stderr when detected; their decoders are
 * the stand-alone front ends (host/lms6Xmod.c ...).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_hip.h"
#include "sonde_scan.h"
#include "sonde_rs41.h"
#include "sonde_dfm.h"
#include "sonde_m10.h"
#include "sonde_m20.h"

enum { T_RS41, T_DFM, T_M10, T_M20, T_N };
static const char *kTypeName[T_N] = { "RS41", "DFM", "M10", "M20" };

typedef struct {
    int used, type, slot;
    double fq;                           /* carrier / sample rate, snapped to the mixer raster */
    int khz;
    void *dec;                           /* sonde_<type>_dec_t */
    long frames; int64_t last_frame_at;  /* stream position (samples) of the last frame delivered */
} sonde_t;

typedef struct { sonde_engine_t *eng; int *owner; } group_t;       /* owner[slot] = index into g_sondes or -1 */

static int g_sr = 0, g_slots = 8, g_device = 0, g_verbose = 0;
static long long g_cfreq = 0; static int g_raster = 10000; static double g_release_s = 20.0;
static group_t g_gr[T_N];
static sonde_t *g_sondes = NULL; static int g_nsondes = 0, g_capsondes = 0;
static int64_t g_pos = 0;                                            /* samples consumed */
static int g_chunk = 0;
static char g_version[32] = "sonde_hip";

static double snap_fq(double fq, int sr) { return (double)(long long)llround(fq * sr / 16.0) * 16.0 / sr; }      /* demod_mod.c:1265-1288 where 16 divides sr */

static int group_engine(int type) {
    group_t *g = &g_gr[type];
    if (g->eng) return 0;
    sonde_cfg_t c; memset(&c, 0, sizeof c);
    c.abi_version = SONDE_ABI_VERSION; c.device = g_device; c.n_channels = g_slots; c.sample_rate = g_sr; c.bits = 16;
    c.sonde_type = type == T_RS41 ? SONDE_RS41 : type == T_DFM ? SONDE_DFM09 : type == T_M10 ? SONDE_M10 : SONDE_M20;
    c.opt_lp = SONDE_LP_IQ; c.ecc_level = type == T_DFM ? 1 : 2; c.opt_auto = type == T_DFM;
    c.max_chunk = g_chunk; c.max_frames = 16 * g_slots; c.input = SONDE_IN_IQ;
    double *fq = (double *)calloc((size_t)g_slots, sizeof(double));
    if (!fq) return SONDE_E_NOMEM;
    const int rc = sonde_engine_create(&c, fq, &g->eng);
    free(fq);
    if (rc < 0) return rc;
    g->owner = (int *)malloc(sizeof(int) * (size_t)g_slots);
    if (!g->owner) return SONDE_E_NOMEM;
    for (int s = 0; s < g_slots; s++) g->owner[s] = -1;
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
    else { sonde_m20_opts_t o; memset(&o, 0, sizeof o); o.verbose = 1; o.ptu = 1; o.json = 1; o.jsn_freq_khz = khz; snprintf(o.version, sizeof o.version, "%s", g_version);
           if (sonde_m20_dec_create(&o, (sonde_m20_dec_t **)&d) < 0) return NULL; }
    return d;
}
static void free_decoder(int type, void *d) {
    if (!d) return;
    if (type == T_RS41) sonde_rs41_dec_destroy((sonde_rs41_dec_t *)d); else if (type == T_DFM) sonde_dfm_dec_destroy((sonde_dfm_dec_t *)d);
    else if (type == T_M10) sonde_m10_dec_destroy((sonde_m10_dec_t *)d); else sonde_m20_dec_destroy((sonde_m20_dec_t *)d);
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

static void release_sonde(int idx) {
    sonde_t *s = &g_sondes[idx];
    group_t *g = &g_gr[s->type];
    sonde_engine_finish_channel(g->eng, s->slot);          /* whatever it still emits is drained (and dropped) with the next fetch */
    g->owner[s->slot] = -1;
    if (g_verbose) fprintf(stderr, "released: %s %d kHz channel %d after %ld frames\n", kTypeName[s->type], s->khz, s->slot, s->frames);
    free_decoder(s->type, s->dec);
    s->used = 0; s->dec = NULL;
}

/* frames the engine of one type has ready -> the decoders of the sondes that own the channels */
static void drain(int type, int finish) {
    group_t *g = &g_gr[type];
    if (!g->eng) return;
    static char tx[8192];
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
                g_sondes[o].frames++; g_sondes[o].last_frame_at = g_pos;
            }
        } else if (type == T_DFM) {
            static sonde_dfm_frame_t fr[64];
            k = sonde_engine_fetch_dfm(g->eng, fr, 64, finish);
            for (int i = 0; i < k; i++) {
                const int o = (fr[i].channel >= 0 && fr[i].channel < g_slots) ? g->owner[fr[i].channel] : -1;
                if (o < 0) continue;
                fr[i].channel = 0;
                if (sonde_dfm_dec_frame((sonde_dfm_dec_t *)g_sondes[o].dec, &fr[i], tx, sizeof tx) > 0) print_json_lines(tx);
                g_sondes[o].frames++; g_sondes[o].last_frame_at = g_pos;
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

int main(int argc, char **argv) {
    int ai = 1;
    for (; ai < argc; ai++) {
        if (!strcmp(argv[ai], "--cfreq") && ai + 1 < argc) g_cfreq = atoll(argv[++ai]);
        else if (!strcmp(argv[ai], "--raster") && ai + 1 < argc) g_raster = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--slots") && ai + 1 < argc) g_slots = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--release-s") && ai + 1 < argc) g_release_s = atof(argv[++ai]);
        else if (!strcmp(argv[ai], "--device") && ai + 1 < argc) g_device = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "-v")) g_verbose = 1;
        else break;
    }
    if (argc - ai != 3 || strcmp(argv[ai], "-") || atoi(argv[ai + 2]) != 16 || atoi(argv[ai + 1]) < 48000 || g_raster < 100 || g_slots < 1 || g_slots > 256) {
        fprintf(stderr, "usage: %s [--cfreq Hz] [--raster Hz] [--slots N] [--release-s S] [--device D] [-v] - <sr> 16   (cs16 on stdin)\n", argv[0]);
        return 255;
    }
    g_sr = atoi(argv[ai + 1]);
    if (getenv("SONDE_JSN_VERSION")) snprintf(g_version, sizeof g_version, "%s", getenv("SONDE_JSN_VERSION"));
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
                for (int i = 0; i < k; i++) {
                    const double fq = raster[det[i].channel] + det[i].df;
                    if (!strcmp(det[i].type, "RS41")) { if (det[i].score > 0) start_sonde(T_RS41, fq); }
                    else if (!strcmp(det[i].type, "DFM9")) start_sonde(T_DFM, fq);                 /* either polarity: the decoder runs with --auto */
                    else if (!strcmp(det[i].type, "M10")) start_sonde(T_M10, fq);                  /* differential code: polarity does not matter */
                    else if (!strcmp(det[i].type, "M20")) start_sonde(T_M20, fq);
                    else if (g_verbose) fprintf(stderr, "seen: %s %.4f %+.0f Hz (decoder: the type's stand-alone front end)\n", det[i].type, det[i].score, fq * g_sr);
                }
                if (k < 64) break;
            }
            for (int t = 0; t < T_N; t++) {
                if (!g_gr[t].eng) continue;
                rc = sonde_engine_process_host(g_gr[t].eng, buf, 0, n);
                if (rc < 0) { fprintf(stderr, "sonde_wideband: %s engine: %s\n", kTypeName[t], sonde_strerror(rc)); return 255; }
            }
            g_pos += n;
            for (int t = 0; t < T_N; t++) drain(t, 0);
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
