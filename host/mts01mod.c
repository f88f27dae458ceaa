/*
 * host/mts01mod.c — `mts01mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/mts01mod.c:343-466 argv, :151-286 output, :569-618 frame loop):
 *     mts01mod [-r | -R] [-v] [--json] [--jsn_cfq hz] [--br baud] [--ths x] [-d shift]
 *              ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     mts01mod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     mts01mod [...] --softin | --softinv                           float32 soft bits
 * auto_rx: `mts01mod --json --IQ 0.0 --lpIQ --dc - <sr> 16` (decode.py:781).  As in the reference the argument list ends at the file name.
 * stdout: per frame the ASCII telemetry string + [OK] / [NO]; -v the parsed fields; --json the JSON object of frames whose CRC holds.
 * Exit 0 at EOF, 255 on argument / init errors.  --spike (clipping, only effective for FM audio / --iq0 in the reference) is refused: the
 * reference's clipping reads an uninitialised variable (demod_mod.c:1092,1121).
 *
 * The sample-rate part runs in the engine (generic sonde description: header AA AA B4 2B, 1200 Bd, BT 1.5, h 0.9, 2 header errors, 1048
 * bits per hit, centre window 2 for IF-rate IQ); everything behind a hit is sonde_mts01.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "sonde_mts01.h"
#include "cli_common.h"

#define MAXHITS 8
static const char kHeader[] = "10101010" "10101010" "10110100" "00101011";      /* mts01mod.c:47-48 */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_mts01_opts_t o;
    cli_in_t in;
    int spike = 0, softin = 0, cfreq = -1, shift = 0, oc;
    float thres = 0.76f, baudrate = -1.f;
    FILE *fp = stdin;
    static char out[1 << 16];
    memset(&o, 0, sizeof o);
    cli_in_init(&in, 4000, 48.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_GENERIC;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -r, --raw\n");
            return 0;
        }
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) o.verbose = 1;
        else if (!strcmp(a, "-r") || !strcmp(a, "--raw")) o.raw = 1;
        else if (!strcmp(a, "-R") || !strcmp(a, "--RAW")) o.raw = 2;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) { /* no effect in the reference either (:580-582) */ }
        else if (!strcmp(a, "--br")) { if (++i >= argc) return -1; baudrate = (float)atof(argv[i]); if (baudrate < 1000 || baudrate > 1400) baudrate = 1200; }
        else if (!strcmp(a, "--spike")) spike = 1;
        else if (!strcmp(a, "--json")) o.json = 1;
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
            break;                                               /* the reference stops reading arguments here (:343) */
        }
        else { fprintf(stderr, "mts01mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    sonde_mts01_dec_t *dec = NULL;

    if (softin) {
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_mts01_dec_create(&o, &dec) < 0) return -1;
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_mts01_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
            if (got < 1024) break;
        }
        sonde_mts01_dec_destroy(dec);
        return 0;
    }

    if (cli_input_setup("mts01mod", fp, &cfg, &in) < 0) return -1;
    if (spike && in.iq_mode < 2) { fprintf(stderr, "mts01mod (sonde_hip): --spike is not supported (undefined in the reference: demod_mod.c:1092,1121)\n"); return -1; }
    if ((float)cfg.sample_rate / 1200.0f < 8) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 1200.0f);
    if (baudrate > 0) fprintf(stderr, "sps corr: %.4f\n", (float)cfg.sample_rate / baudrate);
    o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;                /* no tuning offset added here (mts01mod.c:480) */
    if (sonde_mts01_dec_create(&o, &dec) < 0) return -1;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_auto = 1;                                            /* headers of both polarities */
    cfg.keep_soft = 1;

    sonde_generic_t g;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = baudrate > 0 ? baudrate : 1200.0f; g.bt = 1.5f; g.h = 0.9f; g.symlen = 1; g.symhd = 1;     /* mts01mod.c:514-533 */
    g.hdmax = 2; g.bitofs = shift;                                                                    /* :575 */
    g.nbits = SONDE_MTS01_FRAME_BITS;
    g.l_win = 2.0f;                                                                                    /* bl = 2.0 for opt_iq > 2, whole bits else (:601-604) */
    g.lpiq_bw = in.lpiq_bw; g.lpfm_bw = 4000;
    sonde_engine_t *eng = NULL;
    int rc = sonde_engine_create_generic(&cfg, &in.fq, &g, &eng);
    if (rc >= 0) rc = sonde_engine_set_threshold(eng, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
    sonde_info_t info;
    sonde_engine_info(eng, &info);
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    cli_reader_t rd;
    static float s0[MAXHITS * SONDE_MTS01_FRAME_BITS];
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
                float *b = s0 + (size_t)i * SONDE_MTS01_FRAME_BITS;
                if (hits[i].mv < 0.f) for (int j = 0; j < hits[i].nbits; j++) b[j] = -b[j];      /* stored in the polarity in effect; the reference reads raw bits */
                const int m = sonde_mts01_dec_frame(dec, b, hits[i].nbits, out, sizeof out);
                if (m > 0) fwrite(out, 1, (size_t)m, stdout);
            }
        }
    }
    sonde_engine_destroy(eng);
    sonde_mts01_dec_destroy(dec);
    cli_reader_free(&rd);
    return 0;
}
