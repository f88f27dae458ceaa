/*
 * host/m20mod.c — `m20mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract kept for the sample-input forms with raw output (reference demod/mod/m20mod.c:1055-1200 argv, :918-990
 * output, :1300-1420 frame loop):
 *     m20mod -r [-v] [--ths x] ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>
 *     m20mod -r [-v] [--ch2] [file.wav]                                  FM audio
 * stdout: one line of hex per frame (frame byte 0 + 1 bytes), with -v ` # <checksum> [(ok)|(oo)|(no)] [OK]|[NO]` (block check for firmware < 7)
 * stderr: `note: sample rate low`, `IF:` / `dec:`; exit 0 at EOF, 255 on argument / init errors.
 * Without -r: the position / PTU text line and, with --json, the JSON object (include/sonde_m20.h; -v, -vv, -vvv, --ptu, --json,
 * --jsn_cfq, --silent); --softin / --softinv take the soft symbols of fsk_demod -s.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "broker_client.h"
#include "sonde_m20.h"
#include "cli_common.h"

static int g_shift = 0;      /* -d <shift>: added to the bit offset of the slicer (m20mod.c:1040,1108-1114) */
static double g_baud = -1;                   /* --br */
static int g_verbose = 0, g_raw = 0, g_color = 0;
static sonde_m20_dec_t *g_dec = NULL;

/* print_frame() (m20mod.c:870-1003): raw line with -r, else (or with -r --json: silently) the decoded position line / JSON */
static void emit_frame(const sonde_m20_frame_t *f) {
    static char ln[4096], tx[4096];
    if (g_raw && sonde_m20_rawline(f, g_verbose | (g_color ? SONDE_M20_COLOR : 0), ln, sizeof ln) > 0) fprintf(stdout, "%s\n", ln);
    if (g_dec && sonde_m20_dec_frame(g_dec, f, tx, sizeof tx) > 0) fputs(tx, stdout);
}

static int make_decoder(sonde_m20_opts_t *o, int raw, int khz) {
    const char *ver = getenv("SONDE_JSN_VERSION");
    g_raw = raw;
    o->raw = raw; o->verbose = g_verbose; o->jsn_freq_khz = khz;
    if (raw && !o->json && !o->silent) return 0;
#ifdef VER_JSN_STR
    if (!ver) ver = VER_JSN_STR;
#endif
    if (ver) { strncpy(o->version, ver, sizeof o->version - 1); o->version[sizeof o->version - 1] = 0; }
    return sonde_m20_dec_create(o, &g_dec);
}

static void emit_rec(const void *r) { emit_frame((const sonde_m20_frame_t *)r); }      /* records from the resident broker */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    cli_in_t in;
    int raw = 0, softin = 0, rawhex = 0, cfreq = -1, oc;
    FILE *fp = stdin;
    sonde_m20_opts_t dopt;
    memset(&dopt, 0, sizeof dopt);
    cli_in_init(&in, 0, 48.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_M20;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -v, -vv, -vvv\n");
            fprintf(stderr, "       -r, --raw\n");
            fprintf(stderr, "       --ptu\n");
            fprintf(stderr, "       --json\n");
            return 0;
        }
        if (!strcmp(a, "-r") || !strcmp(a, "--raw")) raw = 1;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) g_verbose = 1;
        else if (!strcmp(a, "-vv")) g_verbose = 2;
        else if (!strcmp(a, "-vvv")) g_verbose = 3;
        else if (!strcmp(a, "--ptu")) dopt.ptu = 1;
        else if (!strcmp(a, "-c") || !strcmp(a, "--color")) { g_color = 1; dopt.color = 1; }
        else if (!strcmp(a, "--json")) dopt.json = 1;
        else if (!strcmp(a, "--silent")) dopt.silent = 1;
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--rawhex")) rawhex = 1;
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) { /* irrelevant for the differential code (m20mod.c:1447) */ }
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; cfg.thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; g_shift = atoi(argv[i]); if (g_shift > 4) g_shift = 4; if (g_shift < -4) g_shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (!strcmp(a, "--br")) {                   /* symbol rate; out of range = the default (m20mod.c) */
            if (++i >= argc) return -1;
            g_baud = atof(argv[i]);
            if (g_baud < 9000 || g_baud > 10000) g_baud = 9600.0;
        }
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
        }
        else { fprintf(stderr, "m20mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    if (rawhex) {                                    /* frames as hex lines, e.g. the output of -r (m20mod.c:1384-1412): bytes up to the first blank,
                                                      * lines longer than the week field; bytes not given keep the previous line's */
        if (make_decoder(&dopt, raw, cfreq > 0 ? (cfreq + 500) / 1000 : 0) < 0) return -1;
        static char lb[2 * 165 + 12];
        static sonde_m20_frame_t fr;
        while (fgets(lb, sizeof lb, fp)) {
            lb[2 * 165] = 0;
            char *sp = strchr(lb, ' ');
            if (sp) *sp = 0;
            const int len = (int)(strlen(lb) / 2);
            if (len <= 0x1A + 2) continue;
            static unsigned char frmbyte;                        /* keeps its value over pairs that are not hex, as the reference's does */
            for (int i = 0; i < len; i++) { sscanf(lb + 2 * i, "%2hhx", &frmbyte); fr.frame[i] = frmbyte; }
            fr.nbits = len * 8;
            sonde_m20_frame_finish(&fr);
            emit_frame(&fr);
        }
        return 0;
    }
    if (softin) {                                    /* float32 soft symbols on stdin (m20mod.c:1405-1510) */
        if (make_decoder(&dopt, raw, cfreq > 0 ? (cfreq + 500) / 1000 : 0) < 0) return -1;
        sonde_softin_t *si = NULL;
        if (sonde_softin_create(SONDE_M20, 0, softin == 2, 0, 1, &si) < 0) return -1;
        sonde_softin_set_m10_skip(si, g_verbose < 3);
        float sb[1024]; sonde_m20_frame_t fr[4]; size_t got;
        for (;;) {
            got = fread(sb, 4, 1024, fp);
            if (got) sonde_softin_push(si, sb, (int32_t)got);
            if (got < 1024) sonde_softin_finish(si);
            int k;
            while ((k = sonde_softin_fetch_m20(si, fr, 4)) > 0) for (int i = 0; i < k; i++) emit_frame(&fr[i]);
            if (got < 1024) break;
        }
        sonde_softin_destroy(si);
        return 0;
    }
    if (cli_input_setup("m20mod", fp, &cfg, &in) < 0) return -1;
    cfg.lpiq_bw = in.lpiq_bw;                         /* 0 = the sonde type's own */
    if ((float)cfg.sample_rate / 9600.0f < 8) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 9600.0f);   /* m20mod.c:1392 */
    {
        const double xlt = (in.iq_mode == 5) ? -in.fq : 0.0;
        if (make_decoder(&dopt, raw, cfreq > 0 ? (int)((cfreq - xlt * cfg.sample_rate + 500) / 1e3) : 0) < 0) return -1;
    }
    cfg.m10_noskip = g_verbose >= 3;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = 16;
    sonde_engine_t *eng = NULL;
    brk_demod_t brk; brk.fd = -1;
    const int use_broker = g_baud > 0 ? 0 : brk_demod_wanted(&cfg);     /* SONDE_BROKER: a channel of the resident engine instead of one of our own; --br: our own (the broker's groups run the preset rate) */
    int rc = 0;
    sonde_info_t info;
    if (g_baud > 0) fprintf(stderr, "sps corr: %.4f\n", (float)cfg.sample_rate / (float)g_baud);        /* before init_buffers()' own lines */
    if (use_broker) {
        if (brk_demod_open(&brk, &cfg, g_shift != 0, 2, 0 + g_shift) < 0) return -1;
        info = brk.info;
    } else {
        if (g_baud > 0) {                            /* --br: dsp.br / dsp.sps replaced before init_buffers() */
            sonde_generic_t gb; memset(&gb, 0, sizeof gb); gb.baud = (float)g_baud;
            rc = sonde_engine_create_generic(&cfg, &in.fq, &gb, &eng);
        } else rc = sonde_engine_create(&cfg, &in.fq, &eng);
        if (rc >= 0 && g_shift) rc = sonde_engine_set_sync(eng, 2, 0 + g_shift);
        if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
        sonde_engine_info(eng, &info);
    }
    if (in.iq_mode == 5) {                              /* init_buffers prints these first (demod_mod.c:1257-1258) */
        fprintf(stderr, "IF: %d\n", info.if_sr);
        fprintf(stderr, "dec: %d\n", info.decM);
    }
    const size_t unit = cli_sample_bytes(&cfg, &in);
    cli_reader_t rd;
    if (cli_reader_init(&rd, unit, cfg.sample_rate, info.decM) < 0) return -1;
    sonde_m20_frame_t frames[16];
    for (;;) {
        int n;
        const size_t got = cli_reader_fill(&rd, fp, &n);
        if (n > 0) {
            if (use_broker) {
                if (brk_demod_feed(&brk, rd.buf, n, unit, 0, sizeof frames[0], emit_rec) < 0) { fprintf(stderr, "error: broker\n"); return -1; }
            } else {
                rc = sonde_engine_process_host(eng, rd.buf, n, n);
                if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); return -1; }
                int k = sonde_engine_fetch_m20(eng, frames, 16, 0);
                for (int i = 0; i < k; i++) emit_frame(&frames[i]);
            }
            cli_reader_consume(&rd, n);
        }
        if (got == 0) break;
    }
    if (use_broker) brk_demod_feed(&brk, NULL, 0, unit, 1, sizeof frames[0], emit_rec);
    else {
        int k = sonde_engine_fetch_m20(eng, frames, 16, 1);
        for (int i = 0; i < k; i++) emit_frame(&frames[i]);
    }
    if (eng) sonde_engine_destroy(eng);
    brk_demod_close(&brk);
    cli_reader_free(&rd);
    return 0;
}
