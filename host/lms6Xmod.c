/*
 * host/lms6Xmod.c — `lms6Xmod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/lms6Xmod.c:1065-1192 argv, :713-798 / :919-928 output, :1353-1465 block loop):
 *     lms6Xmod [-r] [--ecc | --ecc3] [--vit | --vit2] [--json] [--jsn_cfq hz] [--lms6 | --lmsX] [--gpsweek w] [--ths x] [-d shift]
 *              ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     lms6Xmod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     lms6Xmod [...] --softin | --softinv                          float32 soft bits, what auto_rx pipes in from fsk_demod (decode.py:1209)
 * stdout: per data frame the position line (or, with -r, its 223 bytes as hex) + `[OK]` / `[NO]`, and with --json the JSON object of frames
 * whose CRC holds.  Exit 0 at EOF, 255 on argument / init errors.
 *
 * The sample-rate part runs in the engine (generic sonde description: the LMS6 header 58 f3 3f b8 in (c0, inv(c1)) form, 4800 Bd, BT 1.2,
 * h 0.9, 10 header errors, one block of raw bits per hit); everything behind a hit is sonde_lms6.h.  When the auto detection switches
 * between LMS6 (4800 Bd, 4096 bits per block) and LMS-X (4797.8 Bd, 4720 bits) — lms6Xmod.c:1436-1462 — the engine is set up again with the
 * other bit clock and fed from the end of the block that told; the reference keeps its filter state over that point, the engine starts
 * it fresh 64 bits earlier.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "sonde_hip.h"
#include "sonde_lms6.h"
#include "cli_common.h"

#define MAXHITS 8
#define MAXBITS 4720
static const char kHeader[] = "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110";      /* lms6Xmod.c:100 */

static sonde_engine_t *make_engine(const sonde_cfg_t *cfg, double fq, int lpiq_bw, int typ, float thres, int shift, sonde_info_t *info) {
    sonde_generic_t g;
    sonde_engine_t *e = NULL;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = 4800.0f; g.bt = 1.2f; g.h = 0.9f; g.symlen = 1; g.symhd = 1;                 /* lms6Xmod.c:1289-1299 */
    g.hdmax = 10; g.bitofs = shift;                                                     /* :1358, bitofs6 = bitofsX = 0 */
    g.nbits = typ == 10 ? 300 * 16 - 80 : 261 * 16 - 80;
    g.l_win = -1.0f;                                                                    /* read_softbit2p(..., -1, 0, ...) :1394 */
    g.lpiq_bw = lpiq_bw; g.lpfm_bw = 6000;
    if (typ == 10) g.slice_baud = 4797.8f;                                              /* dsp.br after init_buffers (:1343-1347) */
    int rc = sonde_engine_create_generic(cfg, &fq, &g, &e);
    if (rc >= 0) rc = sonde_engine_set_threshold(e, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return NULL; }
    sonde_engine_info(e, info);
    return e;
}

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_lms6_opts_t o;
    cli_in_t in;
    int softin = 0, cfreq = -1, shift = 0, oc;
    float thres = 0.65f;
    FILE *fp = stdin;
    static char out[1 << 16];
    memset(&o, 0, sizeof o);
    cli_in_init(&in, 16000, 24.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_GENERIC;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n");
            fprintf(stderr, "       -v, --verbose\n");
            fprintf(stderr, "       -r, --raw\n");
            fprintf(stderr, "       --vit        (Viterbi)\n");
            fprintf(stderr, "       --ecc        (Reed-Solomon)\n");
            return 0;
        }
        if (!strcmp(a, "-r") || !strcmp(a, "--raw")) o.raw = 1;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) { /* no output depends on it */ }
        else if (!strcmp(a, "--lms6")) o.typ = 6;
        else if (!strcmp(a, "--lmsX")) o.typ = 10;
        else if (!strcmp(a, "--ecc")) o.ecc = 1;
        else if (!strcmp(a, "--ecc3")) o.ecc = 3;
        else if (!strcmp(a, "--vit")) o.vit = 1;
        else if (!strcmp(a, "--vit2")) o.vit = 2;
        else if (!strcmp(a, "--json")) { o.json = 1; o.ecc = 1; o.vit = 1; }
        else if (!strcmp(a, "--gpsweek")) { if (++i >= argc) return -1; o.gpsweek = atoi(argv[i]); if (o.gpsweek < 1024 || o.gpsweek > 3072) o.gpsweek = 0; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) { /* irrelevant: the header's sign decides (lms6Xmod.c:1364-1367) */ }
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
        }
        else { fprintf(stderr, "lms6Xmod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    sonde_lms6_dec_t *dec = NULL;

    if (softin) {                                                /* float32 soft bits on stdin (lms6Xmod.c:1354-1356,1383-1390) */
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_lms6_dec_create(&o, &dec) < 0) return -1;
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_lms6_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fputs(out, stdout);
            if (got < 1024) break;
        }
        sonde_lms6_dec_destroy(dec);
        return 0;
    }

    if (!in.have_iq && !in.have_pcm && o.vit == 2) { o.vit = 1; fprintf(stderr, "info: soft decoding only for IQ\n"); }       /* lms6Xmod.c:1249-1252 (behind the raw-data check) */
    if (cli_input_setup("lms6Xmod", fp, &cfg, &in) < 0) return -1;
    if ((float)cfg.sample_rate / 4800.0f < 8) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 4800.0f);
    o.jsn_freq_khz = cfreq > 0 ? (int)((cfreq - (in.iq_mode == 5 ? -in.fq : 0.0) * cfg.sample_rate + 500) / 1e3) : 0;
    if (sonde_lms6_dec_create(&o, &dec) < 0) return -1;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_auto = 1;                                            /* headers of both polarities; the sign of the score is the code phase */
    cfg.keep_soft = 2;

    sonde_info_t info;
    int typ = sonde_lms6_dec_type(dec, NULL) & 0xFF;
    sonde_engine_t *eng = make_engine(&cfg, in.fq, in.lpiq_bw, typ, thres, shift, &info);
    if (!eng) return -1;
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    const size_t unit = cli_sample_bytes(&cfg, &in);
    int chunk = cfg.sample_rate / 10;
    chunk -= chunk % info.decM;
    if (chunk < info.decM) chunk = info.decM;
    /* input history: the samples since hist0 (an input sample index, multiple of decM) — enough to go back to the end of a block when the
     * symbol rate has to change */
    const int64_t keep = (int64_t)cfg.sample_rate * 3;
    size_t cap = (size_t)(keep + 2 * (int64_t)chunk) * unit;
    char *hist = (char *)malloc(cap);
    static float s0[MAXHITS * MAXBITS], s1[MAXHITS * MAXBITS];
    static sonde_hit_t hits[MAXHITS];
    if (!hist) return -1;
    int64_t hist0 = 0, hist_n = 0;            /* first sample held / samples held */
    int64_t fed = 0;                          /* next input sample for the engine */
    int64_t eng0 = 0;                         /* input sample the current engine started at */
    int64_t skip_before = 0;                  /* IF-rate position: hits before it belong to a block already decoded */
    uint32_t mpos0 = 0;
    int eof = 0;
    for (;;) {
        if (!eof && hist0 + hist_n - fed < chunk) {            /* (after a restart: first catch up with what is held) */
            if ((size_t)(hist_n + chunk) * unit > cap) {                                 /* drop what is older than `keep` */
                int64_t drop = hist_n - keep;
                drop -= drop % info.decM;
                if (drop > fed - hist0) drop = fed - hist0;
                if (drop > 0) { memmove(hist, hist + (size_t)drop * unit, (size_t)(hist_n - drop) * unit); hist0 += drop; hist_n -= drop; }
            }
            const size_t got = fread(hist + (size_t)hist_n * unit, unit, (size_t)chunk, fp);
            hist_n += (int64_t)got;
            if (got == 0) eof = 1;
        }
        int64_t n = hist0 + hist_n - fed;
        if (n > chunk) n = chunk;
        n -= n % info.decM;
        if (n > 0) {
            const int rc = sonde_engine_process_host(eng, hist + (size_t)(fed - hist0) * unit, n, (int32_t)n);
            if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); return -1; }
            fed += n;
        }
        const int at_end = eof && hist0 + hist_n - fed < info.decM;
        if (n <= 0 && !at_end) continue;
        const int k = sonde_engine_fetch_hits(eng, hits, MAXHITS, at_end);
        if (k < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(k)); return -1; }
        int restarted = 0;
        if (k > 0) {
            const int nb = sonde_lms6_dec_block_bits(dec);
            sonde_engine_fetch_soft(eng, s0, k); sonde_engine_fetch_soft1(eng, s1, k);
            for (int i = 0; i < k && !restarted; i++) {
                const int64_t pos = eng0 / info.decM + (int64_t)hits[i].mv_pos;           /* IF-rate position in the whole stream */
                if (pos < skip_before) continue;
                float *b0 = s0 + (size_t)i * nb, *b1 = s1 + (size_t)i * nb;
                if (hits[i].mv < 0.f) for (int j = 0; j < hits[i].nbits; j++) { b0[j] = -b0[j]; b1[j] = -b1[j]; }   /* stored in the polarity in effect; the decoder wants them raw */
                const uint32_t mv_pos = (uint32_t)pos;
                const float frm_rate = (float)(4800.0 * info.if_sr / (double)(uint32_t)(mv_pos - mpos0));           /* lms6Xmod.c:1372 */
                mpos0 = mv_pos;
                const double sps_if = (double)info.if_sr / (typ == 10 ? 4797.8 : 4800.0);
                const double block_end = (double)pos + hits[i].nbits * sps_if;
                const int m = sonde_lms6_dec_block(dec, b0, b1, hits[i].nbits, hits[i].mv, frm_rate, block_end / info.if_sr, out, sizeof out);
                if (m > 0) fputs(out, stdout);
                int32_t changed = 0;
                const int t = sonde_lms6_dec_type(dec, &changed) & 0xFF;
                if (changed && t != typ) {                       /* the other symbol rate from the end of this block on */
                    int64_t from = ((int64_t)block_end - (int64_t)(64 * sps_if)) * info.decM;
                    from -= from % info.decM;
                    if (from < hist0) from = hist0;
                    if (from > fed) from = fed;
                    sonde_engine_destroy(eng);
                    typ = t;
                    eng = make_engine(&cfg, in.fq, in.lpiq_bw, typ, thres, shift, &info);
                    if (!eng) return -1;
                    eng0 = from; fed = from; skip_before = (int64_t)block_end;
                    restarted = 1;
                }
            }
        }
        if (at_end && !restarted) break;
    }
    sonde_engine_destroy(eng);
    sonde_lms6_dec_destroy(dec);
    free(hist);
    return 0;
}
