/*
 * host/imet54mod.c — `imet54mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/imet54mod.c:766-893 argv, :494-707 output, :1008-1061 frame loop):
 *     imet54mod [-r | -r4] [-v] [--ecc] [--ptu] [--silent] [--json] [--jsn_cfq hz] [-i] [--auto] [--br baud] [--ths x] [-d shift]
 *               ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     imet54mod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     imet54mod [...] --softin | --softinv                          float32 soft bits (decode.py:1250: `--ecc --json --softin -i --ptu`)
 *     imet54mod [...] --rawhex                                      frames as hex lines (the output of -r)
 * auto_rx: `imet54mod --ecc --IQ 0.0 --lp - 48000 16 --json --ptu` (decode.py:632).  As in the reference the argument list ends at the file name.
 * stdout: per frame the position line + [OK] / [ok] / [oo] / [NO] / [no]; -r the frame bytes; --json the JSON object of good frames.
 * Exit 0 at EOF, 255 on argument / init errors.
 *
 * The sample-rate part runs in the engine (generic sonde description: header 00 AA 24 24 as 8N1 characters, 4798 Bd, BT 1.0, h 0.8, 4 header
 * errors, bit offset 1, 2200 bits per hit, centre window 2 for IF-rate IQ, polarity per -i / --auto); everything behind a hit is sonde_imet54.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "sonde_imet54.h"
#include "cli_common.h"

#define MAXHITS 8
static const char kHeader[] = "0000000001" "0101010101" "0001001001" "0001001001";      /* imet54mod.c:91-98 */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_imet54_opts_t o;
    cli_in_t in;
    int rawhex = 0, softin = 0, cfreq = -1, shift = 0, oc;
    float thres = 0.7f, baudrate = -1.f;
    FILE *fp = stdin;
    static char out[1 << 16];
    memset(&o, 0, sizeof o);
    cli_in_init(&in, 7400, 24.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_GENERIC;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -v, -vx, -vv  (info, aux, info/conf)\n");
            fprintf(stderr, "       -r, --raw\n");
            fprintf(stderr, "       -i, --invert\n");
            fprintf(stderr, "       --ths <x>    (peak threshold; default=%.1f)\n", thres);
            fprintf(stderr, "       --iq0,2,3    (IQ data)\n");
            return 0;
        }
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) o.verbose = 1;
        else if (!strcmp(a, "-r") || !strcmp(a, "--raw")) o.raw = 1;
        else if (!strcmp(a, "-r4")) o.raw = 4;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) o.inv = 1;
        else if (!strcmp(a, "--ecc")) o.ecc = 1;
        else if (!strcmp(a, "--sat")) { /* accepted, no output depends on it */ }
        else if (!strcmp(a, "--ptu")) o.ptu = 1;
        else if (!strcmp(a, "--silent")) o.silent = 1;
        else if (!strcmp(a, "--auto")) o.aut = 1;
        else if (!strcmp(a, "--rawhex")) rawhex = 1;
        else if (!strcmp(a, "--br")) { if (++i >= argc) return -1; baudrate = (float)atof(argv[i]); if (baudrate < 4600 || baudrate > 5000) baudrate = 4798; }
        else if (!strcmp(a, "--json")) { o.json = 1; o.ecc = 1; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
            break;                                               /* the reference stops reading arguments here (:766) */
        }
        else { fprintf(stderr, "imet54mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    sonde_imet54_dec_t *dec = NULL;

    if (rawhex) {                                                /* :1086-1112 */
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_imet54_dec_create(&o, &dec) < 0) return -1;
        static char lb[2 * 220 + 12];
        while (fgets(lb, sizeof lb, fp)) {
            const int n = sonde_imet54_dec_rawhex(dec, lb, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
        }
        sonde_imet54_dec_destroy(dec);
        return 0;
    }
    if (softin) {
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_imet54_dec_create(&o, &dec) < 0) return -1;
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_imet54_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
            if (got < 1024) break;
        }
        sonde_imet54_dec_destroy(dec);
        return 0;
    }

    if (cli_input_setup("imet54mod", fp, &cfg, &in) < 0) return -1;
    if ((float)cfg.sample_rate / 4798.0f < 5) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 4798.0f);
    if (baudrate > 0) fprintf(stderr, "sps corr: %.4f\n", (float)cfg.sample_rate / baudrate);
    o.jsn_freq_khz = cfreq > 0 ? (int)((cfreq - (in.iq_mode == 5 ? -in.fq : 0.0) * cfg.sample_rate + 500) / 1e3) : 0;
    if (sonde_imet54_dec_create(&o, &dec) < 0) return -1;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_inv = o.inv; cfg.opt_auto = o.aut;                   /* a header of the other polarity is skipped, or flips the polarity with --auto (:1018-1021) */
    cfg.keep_soft = 1;

    sonde_generic_t g;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = baudrate > 0 ? baudrate : 4798.0f; g.bt = 1.0f; g.h = 0.8f; g.symlen = 1; g.symhd = 1;     /* imet54mod.c:945-962 */
    g.hdmax = 4; g.bitofs = 1 + shift;                                                                /* :748,:1013 */
    g.nbits = SONDE_IMET54_FRAME_BITS;
    g.l_win = 2.0f;                                                                                    /* bl = 2.0 for opt_iq > 2, whole bits else (:1042-1045) */
    g.lpiq_bw = in.lpiq_bw; g.lpfm_bw = 6000;
    sonde_engine_t *eng = NULL;
    int rc = sonde_engine_create_generic(&cfg, &in.fq, &g, &eng);
    if (rc >= 0) rc = sonde_engine_set_threshold(eng, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
    sonde_info_t info;
    sonde_engine_info(eng, &info);
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    cli_reader_t rd;
    static float s0[MAXHITS * SONDE_IMET54_FRAME_BITS];
    static sonde_hit_t hits[MAXHITS];
    int eof = 0;
    if (cli_reader_init(&rd, cli_sample_bytes(&cfg, &in), cfg.sample_rate, info.decM) < 0) return -1;
    while (!eof) {
        int n;
        if (cli_reader_fill(&rd, fp, &n) == 0) eof = 1;
        if (n > 0) {
            rc = sonde_engine_process_host(eng, rd.buf, n, n);
            if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); return -1; }
            cli_reader_consume(&rd, n);
        }
        if (n <= 0 && !eof) continue;
        const int k = sonde_engine_fetch_hits(eng, hits, MAXHITS, eof);
        if (k < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(k)); return -1; }
        if (k > 0) {
            sonde_engine_fetch_soft(eng, s0, k);
            for (int i = 0; i < k; i++) {
                float *b = s0 + (size_t)i * SONDE_IMET54_FRAME_BITS;
                const int m = sonde_imet54_dec_frame(dec, b, hits[i].nbits, out, sizeof out);
                if (m > 0) fwrite(out, 1, (size_t)m, stdout);
            }
        }
    }
    sonde_engine_destroy(eng);
    sonde_imet54_dec_destroy(dec);
    cli_reader_free(&rd);
    return 0;
}
