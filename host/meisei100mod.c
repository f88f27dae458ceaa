/*
 * host/meisei100mod.c — `meisei100mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/meisei100mod.c:434-571 argv, :779-1310 output, :681-1318 frame loop):
 *     meisei100mod [-r] [-v] [--dbg] [--ecc] [--ptu] [--json] [--jsn_cfq hz] [--ims100 | --rs11g] [--year y] [--br baud] [--ths x] [-d shift]
 *                  ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     meisei100mod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     meisei100mod [...] --softin | --softinv                       float32 soft half symbols (decode.py:1379)
 * auto_rx: `meisei100mod --IQ 0.0 --lpIQ --dc - <sr> 16 --json --ptu --ecc` (decode.py:756).
 * As in the reference the argument list ends at the file name, and a file name without -r / --rs11g starts the decoder as iMS-100 (:551-553).
 * stdout: per half-second frame the position / time line of the variant in effect, with --json the JSON object; one newline at the end of input.
 * Exit 0 at EOF, 255 on argument / init errors.
 *
 * The sample-rate part runs in the engine (generic sonde description: 48 half-symbol header of 0x049DCE, 2400 Bd, BT 1.2, h 2.4, 1 header
 * error, 1152 half symbols per hit); everything behind a hit is sonde_meisei.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "sonde_meisei.h"
#include "cli_common.h"

#define MAXHITS 8
static const char kHeader[] = "101010101011010100101011001101001100101011001101";      /* meisei100mod.c:200-201 */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_meisei_opts_t o;
    cli_in_t in;
    int softin = 0, cfreq = -1, shift = 0, rs11g = 0, oc;
    float thres = 0.7f, baudrate = -1.f;
    FILE *fp = stdin;
    static char out[1 << 16];
    memset(&o, 0, sizeof o);
    cli_in_init(&in, 16000, 32.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_GENERIC;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
        help_out:
            fprintf(stderr, "%s <-n> [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  n=1,2\n");
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -r, --raw\n");
            return 0;
        }
        else if (!strcmp(a, "-r")) o.raw = 1;
        else if (!strcmp(a, "--dbg")) o.dbg = 1;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) { /* biphase-S: polarity does not matter */ }
        else if (!strcmp(a, "--ims100")) o.ims100 = 1;
        else if (!strcmp(a, "--rs11g")) rs11g = 1;
        else if (!strcmp(a, "--ecc")) o.ecc = 1;
        else if (!strcmp(a, "--ptu")) o.ptu = 1;
        else if (!strcmp(a, "-v")) o.verbose = 1;
        else if (!strcmp(a, "--br")) { if (++i >= argc) return -1; baudrate = (float)atof(argv[i]); if (baudrate < 2200 || baudrate > 2600) baudrate = 2400; }
        else if (!strcmp(a, "--json")) { o.json = 1; o.ecc = 1; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--year")) { if (++i >= argc) return -1; int y = atoi(argv[i]); if (y > 2003 && y < 2100) o.ref_year = y; }
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            if (rs11g && o.ims100) goto help_out;
            if (!o.raw && !rs11g && !o.ims100) o.ims100 = 1;
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
            break;                                               /* the reference stops reading arguments here (:434) */
        }
        else { fprintf(stderr, "meisei100mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    sonde_meisei_dec_t *dec = NULL;

    if (softin) {
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_meisei_dec_create(&o, &dec) < 0) return -1;
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_meisei_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fputs(out, stdout);
            if (got < 1024) break;
        }
        sonde_meisei_dec_destroy(dec);
        return 0;
    }

    if (cli_input_setup("meisei100mod", fp, &cfg, &in) < 0) return -1;
    if ((float)cfg.sample_rate / 2400.0f < 8) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 2400.0f);
    if (baudrate > 0) fprintf(stderr, "sps corr: %.4f\n", (float)cfg.sample_rate / baudrate);
    o.jsn_freq_khz = cfreq > 0 ? (int)((cfreq - (in.iq_mode == 5 ? -in.fq : 0.0) * cfg.sample_rate + 500) / 1e3) : 0;
    if (sonde_meisei_dec_create(&o, &dec) < 0) return -1;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_auto = 1;                                            /* headers of both polarities */
    cfg.keep_soft = 1;

    sonde_generic_t g;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = baudrate > 0 ? baudrate : 2400.0f; g.bt = 1.2f; g.h = 2.4f; g.symlen = 1; g.symhd = 1;     /* meisei100mod.c:626-644 */
    g.hdmax = 1; g.bitofs = shift;                                                                    /* :690 */
    g.nbits = SONDE_MEISEI_FRAME_SYMBOLS;
    g.l_win = -1.0f;                                                                                   /* read_slbit(..., -1, 0) :713 */
    g.lpiq_bw = in.lpiq_bw; g.lpfm_bw = 4000;
    sonde_engine_t *eng = NULL;
    int rc = sonde_engine_create_generic(&cfg, &in.fq, &g, &eng);
    if (rc >= 0) rc = sonde_engine_set_threshold(eng, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
    sonde_info_t info;
    sonde_engine_info(eng, &info);
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    cli_reader_t rd;
    static float s0[MAXHITS * SONDE_MEISEI_FRAME_SYMBOLS];
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
                const int m = sonde_meisei_dec_frame(dec, s0 + (size_t)i * SONDE_MEISEI_FRAME_SYMBOLS, hits[i].nbits, out, sizeof out);
                if (m > 0) fputs(out, stdout);
            }
        }
    }
    printf("\n");                                                /* :1320 */
    sonde_engine_destroy(eng);
    sonde_meisei_dec_destroy(dec);
    cli_reader_free(&rd);
    return 0;
}
