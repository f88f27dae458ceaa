/*
 * host/rs41mod.c — `rs41mod` command-line front end on top of libsonde_hip (C, like the reference's tools).
 *
 * Keeps the reference's process contract for the IQ form (reference demod/mod/rs41mod.c:2617-2744):
 *     rs41mod [-r] [--ecc|--ecc2|--ecc3|--ecc4] [--crc] [--ths x] --IQ <fq> [--lpIQ | --lpbw kHz] [--min] - <sr> 16
 * stdin : interleaved little-endian int16 I/Q at <sr>            (rs41mod.c:2719-2734)
 * stdout: one raw line per frame, `<hex bytes> [OK]|[NO] (n)`     (rs41mod.c:2530-2545), unbuffered (:2612)
 * stderr: `IF: <rate>` / `dec: <M>`                               (demod_mod.c:1257-1258)
 * exit  : 0 on EOF, 255 on argument / init errors                 (rs41mod.c:2663,2739,2846)
 * Without -r the telemetry text line (and with --json the JSON object auto_rx parses) of print_position() is printed
 * (include/sonde_rs41.h; -v, --ptu, --ptu2, --dewp, --sat, --json, --jsnsubfrm1/2, --jsn_cfq, --silent).
 * The DSP runs on the GPU; there is no CPU fallback: without a HIP device the program exits 255.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "sonde_hip.h"
#include "sonde_rs41.h"
#include "broker_client.h"
#include "cli_common.h"

static sonde_rs41_dec_t *g_dec = NULL;
static int g_raw = 0;
static int g_shift = 0;      /* -d <shift>: added to the bit offset of the slicer (rs41mod.c:2597-2598,2664-2671,2850) */

/* print_frame() (rs41mod.c:2472-2553): raw line with -r (then JSON only, if asked for), else the decoded text */
/* the decoder behind the text / JSON output; version = what the reference compiles in as VER_JSN_STR */
static int make_decoder(sonde_rs41_opts_t *o, int raw, int khz) {
    const char *ver = getenv("SONDE_JSN_VERSION");
    g_raw = raw;
    if (raw && o->json) o->silent = 1;                  /* rs41mod.c:2754 */
    if (raw && !o->json) return 0;
    o->jsn_freq_khz = khz;
#ifdef VER_JSN_STR
    if (!ver) ver = VER_JSN_STR;
#endif
    if (ver) { strncpy(o->version, ver, sizeof o->version - 1); o->version[sizeof o->version - 1] = 0; }
    return sonde_rs41_dec_create(o, &g_dec);
}

static void emit_frame(const sonde_frame_t *f) {
    static char ln[1200], tx[8192];
    if (g_raw) { sonde_rs41_rawline(f, ln, sizeof ln); fprintf(stdout, "%s\n", ln); }
    if (g_dec && sonde_rs41_dec_frame(g_dec, f, tx, sizeof tx) > 0) fputs(tx, stdout);
}

static void emit_rec(const void *r) { emit_frame((const sonde_frame_t *)r); }      /* records from the resident broker */

/* --ecc3 / --ecc4: header hits with both soft bits of every bit (read_softbit2p's hsbit / hsbit1) -> sonde_rs41_dec_ecc() */
static sonde_rs41_dec_t *g_ecc_only = NULL;
static sonde_rs41_dec_t *ecc_ctx(void) {            /* -r without --json: the ECC state still lives in a decoder object (never printed from) */
    if (g_dec) return g_dec;
    if (!g_ecc_only) { sonde_rs41_opts_t q; memset(&q, 0, sizeof q); q.silent = 1; if (sonde_rs41_dec_create(&q, &g_ecc_only) < 0) return NULL; }
    return g_ecc_only;
}
static void emit_hits(sonde_engine_t *eng, int level, int if_sr, int finish) {
    static sonde_hit_t hits[8];
    static float s0[8 * 4080], s1[8 * 4080];
    int k = sonde_engine_fetch_hits(eng, hits, 8, finish);
    if (k <= 0) return;
    sonde_engine_fetch_soft(eng, s0, k); sonde_engine_fetch_soft1(eng, s1, k);
    for (int i = 0; i < k; i++) {
        sonde_frame_t f; memset(&f, 0, sizeof f);
        f.channel = hits[i].channel; f.mv = hits[i].mv; f.mv_pos = hits[i].mv_pos;
        /* the soft bits come in the polarity in effect: no further inversion here */
        sonde_rs41_dec_ecc(ecc_ctx(), level, 0, s0 + (size_t)i * 4080, s1 + (size_t)i * 4080, hits[i].nbits,
                           (float)hits[i].mv_pos / (float)if_sr, &f);
        emit_frame(&f);
    }
}

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    cli_in_t in;
    int raw = 0, softin = 0, opt_inv = 0, opt_auto = 0, opt_bin = 0, rawhex = 0, xorhex = 0, oc;
    FILE *fp = stdin;
    sonde_rs41_opts_t dopt;
    int json_ecc = 0, cfreq = -1;
    memset(&dopt, 0, sizeof dopt);
    memset(&cfg, 0, sizeof cfg);
    cli_in_init(&in, 0, 24.0);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_RS41;
    cfg.ecc_level = 1;                               /* rs41mod.c:2757: ecc < 2 -> 1 */
    setbuf(stdout, NULL);

    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -v, -vx, -vv  (info, aux, info/conf)\n");
            fprintf(stderr, "       -r, --raw\n");
            fprintf(stderr, "       -i, --invert\n");
            fprintf(stderr, "       --crc        (check CRC)\n");
            fprintf(stderr, "       --ecc        (Reed-Solomon)\n");
            fprintf(stderr, "       --ths <x>    (peak threshold; default=0.7)\n");
            fprintf(stderr, "       --iq0,2,3    (IQ data)\n");
            return 0;
        }
        if (!strcmp(a, "-r") || !strcmp(a, "--raw")) raw = 1;
        else if (!strcmp(a, "--ecc")) cfg.ecc_level = 1;
        else if (!strcmp(a, "--ecc2")) cfg.ecc_level = 2;
        else if (!strcmp(a, "--ecc3")) cfg.ecc_level = 3;               /* erasures + bit toggling from the soft bits (rs41mod.c:1861-1941) */
        else if (!strcmp(a, "--ecc4")) cfg.ecc_level = 4;               /* + bytes known from earlier frames (rs41mod.c:1764-1849) */
        else if (!strcmp(a, "--crc")) { /* block CRCs are always evaluated by the field decode */ }
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) dopt.verbose = 1;
        else if (!strcmp(a, "-vx")) dopt.verbose = 2;                   /* + the xdata text */
        else if (!strcmp(a, "-vv")) dopt.verbose = 3;                   /* + battery, week, sats, every subframe's bytes */
        else if (!strcmp(a, "--ptu")) dopt.ptu = 1;
        else if (!strcmp(a, "--ptu2")) dopt.ptu = 2;
        else if (!strcmp(a, "--dewp")) dopt.dewp = 1;
        else if (!strcmp(a, "--sat")) dopt.sat = 1;
        else if (!strcmp(a, "--aux")) dopt.aux = 1;
        else if (!strcmp(a, "--silent")) dopt.silent = 1;
        else if (!strcmp(a, "--json")) { dopt.json = 1; cfg.ecc_level = 2; }      /* at this point of the argument list: a later --ecc wins (rs41mod.c:2703-2707) */
        else if (!strcmp(a, "--jsnsubfrm1")) { dopt.jsn_subfrm = 1; dopt.json = 1; json_ecc = 1; }
        else if (!strcmp(a, "--jsnsubfrm2")) { dopt.jsn_subfrm = 2; dopt.json = 1; json_ecc = 1; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; cfg.thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; g_shift = atoi(argv[i]); if (g_shift > 4) g_shift = 4; if (g_shift < -4) g_shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--bin")) opt_bin = 1;                       /* one byte per hard bit */
        else if (!strcmp(a, "--rawhex")) rawhex = 1;                     /* frames as hex lines */
        else if (!strcmp(a, "--xorhex")) { rawhex = 1; xorhex = 1; }
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) opt_inv = 1;
        else if (!strcmp(a, "--auto")) opt_auto = 1;
        else if (a[0] != '-') {                      /* WAV file instead of stdin (rs41mod.c wavloaded) */
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
        }
        else { fprintf(stderr, "rs41mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    if (json_ecc) cfg.ecc_level = 2;                        /* --jsnsubfrm1/2: ecc = 2 after the arguments, whatever they said (rs41mod.c:2769-2773) */
    const int ecc34 = cfg.ecc_level >= 3 ? cfg.ecc_level : 0;
    if (ecc34 && (rawhex || (opt_bin && !softin))) { fprintf(stderr, "rs41mod (sonde_hip): --ecc3/--ecc4 need soft bits (samples or --softin)\n"); return -1; }
    if (ecc34 && !softin) { cfg.ecc_level = 2; cfg.keep_soft = 2; }      /* the engine hands out both soft bits per bit; the list decoding runs in sonde_rs41_dec_ecc() */
    if (rawhex || softin || opt_bin) {
        if (make_decoder(&dopt, raw, cfreq > 0 ? (cfreq + 500) / 1000 : 0) < 0) return -1;
        if (ecc34 && ecc_ctx() == NULL) return -1;
    }
    if (rawhex) {                                    /* rs41mod.c:2976-3002: hex up to the first blank, frames longer than the ID block */
        sonde_softin_t *si = NULL;
        if (sonde_softin_create(SONDE_RS41, cfg.ecc_level, 0, 0, 0, &si) < 0) return -1;
        char lb[2 * 518 + 12]; sonde_frame_t fr;
        while (fgets(lb, sizeof lb, fp)) {
            sonde_softin_push_hexline(si, lb, xorhex);
            while (sonde_softin_fetch(si, &fr, 1) > 0) emit_frame(&fr);
        }
        sonde_softin_destroy(si);
        return 0;
    }
    if (softin || opt_bin) {                                    /* float32 soft bits on stdin (rs41mod.c:2655-2656,2878-2917) */
        sonde_softin_t *si = NULL;
        if (sonde_softin_create(SONDE_RS41, cfg.ecc_level, softin == 2, opt_inv, opt_auto, &si) < 0) return -1;
        float sb[1024]; sonde_frame_t fr[4]; size_t got;
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
            while ((k = sonde_softin_fetch(si, fr, 4)) > 0) {
                if (ecc34) {                             /* ts = dsp.mv_pos / dsp.sr with an untouched dsp (rs41mod.c:2601,2963): 0 / 0 */
                    static float sv[4 * 4080]; int32_t nb[4], iv[4];
                    sonde_softin_fetch_soft(si, sv, nb, iv, k);
                    for (int i = 0; i < k; i++) sonde_rs41_dec_ecc(ecc_ctx(), ecc34, iv[i], sv + (size_t)i * 4080, NULL, nb[i], nanf(""), &fr[i]);
                }
                for (int i = 0; i < k; i++) emit_frame(&fr[i]);
            }
            if (got < 1024) break;
        }
        sonde_softin_destroy(si);
        return 0;
    }
    cfg.opt_inv = opt_inv; cfg.opt_auto = opt_auto;
    if (cli_input_setup("rs41mod", fp, &cfg, &in) < 0) return -1;      /* raw data must be IQ; WAV header; --dc with --IQ implies --lpFM; cfg.input */
    cfg.lpiq_bw = in.lpiq_bw;                         /* 0 = the sonde type's own */

    {   /* "freq" of the JSON: (cfreq - xlt_fq * sr + 500) / 1e3 with xlt_fq = -in.fq for --IQ (rs41mod.c:2806-2809) */
        const double xlt = (in.iq_mode == 5) ? -in.fq : 0.0;
        const int khz = cfreq > 0 ? (int)((cfreq - xlt * cfg.sample_rate + 500) / 1e3) : 0;
        if (make_decoder(&dopt, raw, khz) < 0) { fprintf(stderr, "error: telemetry options\n"); return -1; }
        if (ecc34 && ecc_ctx() == NULL) return -1;
    }
    /* 0.1 s of input per GPU call keeps latency well below one frame */
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    sonde_engine_t *eng = NULL;
    brk_demod_t brk; brk.fd = -1;
    const int use_broker = brk_demod_wanted(&cfg) && !ecc34;      /* SONDE_BROKER: a channel of the resident engine instead of one of our own */
    int rc = 0;
    sonde_info_t info;
    if (use_broker) {
        if (brk_demod_open(&brk, &cfg, g_shift != 0, 4, 2 + g_shift) < 0) return -1;
        info = brk.info;
    } else {
        rc = sonde_engine_create(&cfg, &in.fq, &eng);
        if (rc >= 0 && g_shift) rc = sonde_engine_set_sync(eng, 4, 2 + g_shift);
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
    sonde_frame_t frames[8];
    for (;;) {
        int n;
        const size_t got = cli_reader_fill(&rd, fp, &n);
        if (n > 0) {
            if (use_broker) {
                if (brk_demod_feed(&brk, rd.buf, n, unit, 0, sizeof frames[0], emit_rec) < 0) { fprintf(stderr, "error: broker\n"); return -1; }
            } else {
                rc = sonde_engine_process_host(eng, rd.buf, n, n);
                if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); return -1; }
                if (ecc34) emit_hits(eng, ecc34, info.if_sr, 0);
                else {
                    int k = sonde_engine_fetch_frames(eng, frames, 8);
                    for (int i = 0; i < k; i++) emit_frame(&frames[i]);
                }
            }
            cli_reader_consume(&rd, n);
        }
        if (got == 0) break;                        /* EOF */
    }
    {   /* EOF: the reference still prints a frame it was in the middle of (rs41mod.c:2931,2965) */
        if (use_broker) brk_demod_feed(&brk, NULL, 0, unit, 1, sizeof frames[0], emit_rec);
        else if (ecc34) emit_hits(eng, ecc34, info.if_sr, 1);
        else {
            int k = sonde_engine_finish(eng, frames, 8);
            for (int i = 0; i < k; i++) emit_frame(&frames[i]);
        }
    }
    if (eng) sonde_engine_destroy(eng);
    brk_demod_close(&brk);
    cli_reader_free(&rd);
    return 0;
}
