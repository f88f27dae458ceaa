/*
 * host/dfm09mod.c — `dfm09mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract kept for the IQ + raw form (reference demod/mod/dfm09mod.c:1338-1500 argv, :1198-1236 output):
 *     dfm09mod -r [--ecc|--ecc2] [--ths x] --IQ <fq> [--lpIQ | --lpbw kHz] [--min] - <sr> 16
 * stdout: per frame `<7 nibbles> [OK]   <13 nibbles> [OK]   <13 nibbles> [OK] ` ([KO] = corrected, [NO] = uncorrectable)
 * stderr: `IF:` / `dec:`; exit 0 at EOF, 255 on argument / init errors.  Without -r: the telemetry text line once per nine
 * data packets and, with --json, the JSON object auto_rx parses (include/sonde_dfm.h; -v, -vv, --ptu, --dist, --json, --jsn_cfq, --sat).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "broker_client.h"
#include "sonde_dfm.h"
#include "cli_common.h"

static sonde_dfm_dec_t *g_dec = NULL;
static int g_raw = 0, g_ecc = 0;
static int g_shift = 0;      /* -d <shift>: added to the bit offset of the slicer (dfm09mod.c:1321,1398-1404) */
static double g_baud = -1;                   /* --br */

static int make_decoder(sonde_dfm_opts_t *o, int raw, int ecc, int opt_auto, int khz) {
    const char *ver = getenv("SONDE_JSN_VERSION");
    g_raw = raw == 1; g_ecc = ecc;
    if (raw == 1 && !o->json) return 0;
    o->raw = raw; o->ecc = ecc; o->opt_auto = opt_auto; o->jsn_freq_khz = khz;
#ifdef VER_JSN_STR
    if (!ver) ver = VER_JSN_STR;
#endif
    if (ver) { strncpy(o->version, ver, sizeof o->version - 1); o->version[sizeof o->version - 1] = 0; }
    return sonde_dfm_dec_create(o, &g_dec);
}

/* print_frame() (dfm09mod.c:1153-1262): raw line with -r, then what conf_out / dat_out / print_gpx print */
static void emit_frame(const sonde_dfm_frame_t *f) {
    static char ln[128], tx[4096];
    if (g_raw) { sonde_dfm_rawline(f, g_ecc, ln, sizeof ln); fprintf(stdout, "%s\n", ln); }
    if (g_dec && sonde_dfm_dec_frame(g_dec, f, tx, sizeof tx) > 0) fputs(tx, stdout);
}

static void emit_rec(const void *r) { emit_frame((const sonde_dfm_frame_t *)r); }      /* records from the resident broker */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    cli_in_t in;
    int raw = 0, softin = 0, opt_inv = 0, opt_auto = 0, opt_bin = 0, rawhex = 0, oc;
    FILE *fp = stdin;
    sonde_dfm_opts_t dopt;
    int force_ecc = 0, cfreq = -1;
    memset(&dopt, 0, sizeof dopt);
    memset(&cfg, 0, sizeof cfg);
    cli_in_init(&in, 0, 32.0);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_DFM09;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -v, -vv\n");
            fprintf(stderr, "       -r, --raw\n");
            fprintf(stderr, "       -i, --invert\n");
            fprintf(stderr, "       --ecc        (Hamming ECC)\n");
            fprintf(stderr, "       --ths <x>    (peak threshold; default=0.65)\n");
            fprintf(stderr, "       --json       (JSON output)\n");
            return 0;
        }
        if (!strcmp(a, "-r") || !strcmp(a, "--raw")) raw = 1;
        else if (!strcmp(a, "-R") || !strcmp(a, "--RAW")) raw = 2;          /* data packets as hex (dfm09mod.c:972-981) */
        else if (!strcmp(a, "-vvv")) dopt.verbose = 3;
        else if (!strcmp(a, "--dbg")) dopt.dbg = 1;
        else if (!strcmp(a, "--rawhex")) rawhex = 1;                        /* the lines of --rawecc as input (:1730-1787) */
        else if (!strcmp(a, "--rawecc")) raw = 9;                          /* frame bits before the Hamming decoder (:1177-1196; decode.py:1078) */
        else if (!strcmp(a, "--ecc")) cfg.ecc_level = 1;
        else if (!strcmp(a, "--ecc2")) cfg.ecc_level = 2;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) dopt.verbose = 1;
        else if (!strcmp(a, "-vv")) dopt.verbose = 2;
        else if (!strcmp(a, "--ptu")) dopt.ptu = 1;
        else if (!strcmp(a, "--sat")) dopt.sat = 1;
        else if (!strcmp(a, "--dist")) { dopt.dist = 1; force_ecc = 1; }
        else if (!strcmp(a, "--json")) { dopt.json = 1; force_ecc = 1; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; cfg.thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; g_shift = atoi(argv[i]); if (g_shift > 4) g_shift = 4; if (g_shift < -4) g_shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (!strcmp(a, "--br")) {                   /* symbol rate; out of range = the default (dfm09mod.c) */
            if (++i >= argc) return -1;
            g_baud = atof(argv[i]);
            if (g_baud < 2200 || g_baud > 2800) g_baud = 2500.0;
        }
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--bin")) opt_bin = 1;                       /* one byte per hard bit */
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) opt_inv = 1;
        else if (!strcmp(a, "--auto")) opt_auto = 1;
        else if (a[0] != '-') {                      /* WAV file instead of stdin (dfm09mod.c wavloaded) */
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
        }
        else { fprintf(stderr, "dfm09mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    if (force_ecc) cfg.ecc_level = 1;               /* --dist / --json: option_ecc = 1 (dfm09mod.c:1487) */
    if (rawhex) {
        if (make_decoder(&dopt, raw, cfg.ecc_level, opt_auto, cfreq > 0 ? (cfreq + 500) / 1000 : 0) < 0) return -1;
        sonde_softin_t *si = NULL;
        if (sonde_softin_create(SONDE_DFM09, cfg.ecc_level, 0, opt_inv, opt_auto, &si) < 0) return -1;
        char tb[1024]; sonde_dfm_frame_t fr[8]; size_t got;
        while ((got = fread(tb, 1, sizeof tb, fp)) > 0) {
            sonde_softin_push_dfm_rawhex(si, tb, (int32_t)got);
            int k;
            while ((k = sonde_softin_fetch_dfm(si, fr, 8)) > 0)
                for (int i = 0; i < k; i++) emit_frame(&fr[i]);
        }
        sonde_softin_destroy(si);
        return 0;
    }
    if (softin || opt_bin) {                                    /* float32 soft symbols on stdin (dfm09mod.c:1604-1720) */
        if (make_decoder(&dopt, raw, cfg.ecc_level, opt_auto, cfreq > 0 ? (cfreq + 500) / 1000 : 0) < 0) return -1;
        sonde_softin_t *si = NULL;
        if (sonde_softin_create(SONDE_DFM09, cfg.ecc_level, softin == 2, opt_inv, opt_auto, &si) < 0) return -1;
        float sb[1024]; sonde_dfm_frame_t fr[8]; size_t got;
        for (;;) {
            if (opt_bin && !softin) {                    /* --bin: one byte per bit (--softin wins if both are given) */
                got = fread(sb, 1, 1024, fp);
                if (got) sonde_softin_push_bits(si, (const uint8_t *)sb, (int32_t)got);
            } else {
                got = fread(sb, 4, 1024, fp);
                if (got) sonde_softin_push(si, sb, (int32_t)got);
            }
            if (got < 1024) sonde_softin_finish(si);
            int k;
            while ((k = sonde_softin_fetch_dfm(si, fr, 8)) > 0)
                for (int i = 0; i < k; i++) emit_frame(&fr[i]);
            if (got < 1024) break;
        }
        sonde_softin_destroy(si);
        return 0;
    }
    cfg.opt_inv = opt_inv; cfg.opt_auto = opt_auto;
    if (cli_input_setup("dfm09mod", fp, &cfg, &in) < 0) return -1;      /* raw data must be IQ; WAV header; --dc with --IQ implies --lpFM; cfg.input */
    cfg.lpiq_bw = in.lpiq_bw;                         /* 0 = the sonde type's own */
    {   /* "freq" of the JSON: (cfreq - xlt_fq * sr + 500) / 1e3 (dfm09mod.c:1554-1557) */
        const double xlt = (in.iq_mode == 5) ? -in.fq : 0.0;
        if (make_decoder(&dopt, raw, cfg.ecc_level, opt_auto, cfreq > 0 ? (int)((cfreq - xlt * cfg.sample_rate + 500) / 1e3) : 0) < 0) return -1;
    }

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
        if (brk_demod_open(&brk, &cfg, g_shift != 0, 2, 2 + g_shift) < 0) return -1;
        info = brk.info;
    } else {
        if (g_baud > 0) {                            /* --br: dsp.br / dsp.sps replaced before init_buffers() */
            sonde_generic_t gb; memset(&gb, 0, sizeof gb); gb.baud = (float)g_baud;
            rc = sonde_engine_create_generic(&cfg, &in.fq, &gb, &eng);
        } else rc = sonde_engine_create(&cfg, &in.fq, &eng);
        if (rc >= 0 && g_shift) rc = sonde_engine_set_sync(eng, 2, 2 + g_shift);
        if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
        sonde_engine_info(eng, &info);
    }
    if (in.iq_mode == 5) {
        fprintf(stderr, "IF: %d\n", info.if_sr);
        fprintf(stderr, "dec: %d\n", info.decM);
    }
    const size_t unit = cli_sample_bytes(&cfg, &in);  /* bytes per input sample / audio frame */
    cli_reader_t rd;
    if (cli_reader_init(&rd, unit, cfg.sample_rate, info.decM) < 0) return -1;
    sonde_dfm_frame_t frames[128];
    for (;;) {
        int n;
        const size_t got = cli_reader_fill(&rd, fp, &n);
        if (n > 0) {
            if (use_broker) {
                if (brk_demod_feed(&brk, rd.buf, n, unit, 0, sizeof frames[0], emit_rec) < 0) { fprintf(stderr, "error: broker\n"); return -1; }
            } else {
                rc = sonde_engine_process_host(eng, rd.buf, n, n);
                if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); return -1; }
                int k = sonde_engine_fetch_dfm(eng, frames, 128, 0);
                for (int i = 0; i < k; i++) emit_frame(&frames[i]);
            }
            cli_reader_consume(&rd, n);
        }
        if (got == 0) break;
    }
    if (use_broker) brk_demod_feed(&brk, NULL, 0, unit, 1, sizeof frames[0], emit_rec);
    else {
        int k = sonde_engine_fetch_dfm(eng, frames, 128, 1);
        for (int i = 0; i < k; i++) emit_frame(&frames[i]);
    }
    if (eng) sonde_engine_destroy(eng);
    brk_demod_close(&brk);
    cli_reader_free(&rd);
    return 0;
}
