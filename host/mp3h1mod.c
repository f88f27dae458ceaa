/*
 * host/mp3h1mod.c — `mp3h1mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/mp3h1mod.c:912-1062 argv, :629-862 output, :1173-1243 frame loop):
 *     mp3h1mod [-r | -R] [-v | -vv] [--dbg] [--ptu] [--uniq] [-c] [--json] [--jsn_cfq hz] [-i] [--auto] [--ofs n] [--br baud] [-d shift]
 *              ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     mp3h1mod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     mp3h1mod [...] --softin | --softinv                           float32 soft half symbols (decode.py:1293: `--auto --json --softin --ptu`)
 *     mp3h1mod [...] --rawhex                                       frames as hex lines (the output of -r)
 * auto_rx: `mp3h1mod --IQ 0.0 --lp - 48000 16 --json --ptu` (decode.py:659).  As in the reference the argument list ends at the file name.
 * stdout: per frame (each is sent six times a second; --uniq prints one) the position line + [OK] / [NO]; --json the JSON object once the
 * date and both serial numbers have been received.  Exit 0 at EOF, 255 on argument / init errors.
 *
 * The sample-rate part runs in the engine (generic sonde description: 44 half-symbol header, 2399 Bd, two half symbols per bit, BT 1.0, h 2.0,
 * 2 header errors, bit offset 2, centre window 2 for IF-rate IQ, polarity per -i / --auto); everything behind a hit is sonde_mrz.h.  A frame of
 * the other type (ECEF: 386 bits behind the header, lat / lon: 362) passing its CRC changes how many bits the reference reads per hit
 * (:812-814): the engine is set up again with that count and fed from the end of the frame that told.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "sonde_hip.h"
#include "sonde_mrz.h"
#include "cli_common.h"

#define MAXHITS 8
#define MAXBITS SONDE_MRZ_MAX_BITS
static const char kHeader[] = "100110011001100110011001100110011001" "10101010";      /* mp3h1mod.c:117 */

static sonde_engine_t *make_engine(const sonde_cfg_t *cfg, double fq, int lpiq_bw, int nbits, float baud, float thres, int shift, sonde_info_t *info) {
    sonde_generic_t g;
    sonde_engine_t *e = NULL;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = baud; g.bt = 1.0f; g.h = 2.0f; g.symlen = 2; g.symhd = 2;                    /* mp3h1mod.c:1112-1131 */
    g.hdmax = 2; g.bitofs = 2 + shift;                                                  /* :896,:1181 */
    g.nbits = nbits;
    g.l_win = 2.0f;                                                                     /* bl = 2.0 for opt_iq > 2, whole bits else (:1214-1217) */
    g.lpiq_bw = lpiq_bw; g.lpfm_bw = 6000;
    int rc = sonde_engine_create_generic(cfg, &fq, &g, &e);
    if (rc >= 0) rc = sonde_engine_set_threshold(e, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return NULL; }
    sonde_engine_info(e, info);
    return e;
}

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_mrz_opts_t o;
    cli_in_t in;
    int rawhex = 0, softin = 0, cfreq = -1, shift = 0, oc;
    float thres = 0.76f, baudrate = -1.f;
    FILE *fp = stdin;
    static char out[1 << 16];
    memset(&o, 0, sizeof o);
    cli_in_init(&in, 9000, 32.0);
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
            fprintf(stderr, "       -i, --invert\n");
            return 0;
        }
        else if (!strcmp(a, "--ofs")) { if (++i >= argc) return -1; o.bits_ofs = atoi(argv[i]); o.bits_ofs_given = 1; }
        else if (!strcmp(a, "--dbg")) o.dbg = 1;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) o.verbose = 1;
        else if (!strcmp(a, "-vv")) o.verbose = 2;
        else if (!strcmp(a, "-r") || !strcmp(a, "--raw")) o.raw = 1;
        else if (!strcmp(a, "-R") || !strcmp(a, "--RAW")) o.raw = 2;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) o.inv = 1;
        else if (!strcmp(a, "--auto")) o.aut = 1;
        else if (!strcmp(a, "--uniq")) o.uniq = 1;
        else if (!strcmp(a, "-c") || !strcmp(a, "--color")) o.color = 1;
        else if (!strcmp(a, "--br")) { if (++i >= argc) return -1; baudrate = (float)atof(argv[i]); if (baudrate < 2000 || baudrate > 3000) baudrate = 2399.0f; }
        else if (!strcmp(a, "--ecc")) { /* accepted; the frame has no error correction */ }
        else if (!strcmp(a, "--ptu")) o.ptu = 1;
        else if (!strcmp(a, "--json")) o.json = 1;
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--rawhex")) rawhex = 1;
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error open %s\n", a); return -1; }
            break;                                               /* the reference stops reading arguments here (:912) */
        }
        else { fprintf(stderr, "mp3h1mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    sonde_mrz_dec_t *dec = NULL;

    if (rawhex) {                                                /* :1249-1276 */
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_mrz_dec_create(&o, &dec) < 0) return -1;
        static char lb[3 * 51 + 12];
        while (fgets(lb, sizeof lb, fp)) {
            const int n = sonde_mrz_dec_rawhex(dec, lb, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
        }
        sonde_mrz_dec_destroy(dec);
        return 0;
    }
    if (softin) {                                                /* float32 soft half symbols on stdin (:1199-1211) */
        o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
        if (sonde_mrz_dec_create(&o, &dec) < 0) return -1;
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_mrz_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
            if (got < 1024) break;
        }
        sonde_mrz_dec_destroy(dec);
        return 0;
    }

    if (cli_input_setup("mp3h1mod", fp, &cfg, &in) < 0) return -1;
    if ((float)cfg.sample_rate / 2399.0f < 5) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 2399.0f);
    if (baudrate > 0) fprintf(stderr, "sps corr: %.4f\n", (float)cfg.sample_rate / baudrate);
    const float baud = baudrate > 0 ? baudrate : 2399.0f;
    o.jsn_freq_khz = cfreq > 0 ? (int)((cfreq - (in.iq_mode == 5 ? -in.fq : 0.0) * cfg.sample_rate + 500) / 1e3) : 0;
    if (sonde_mrz_dec_create(&o, &dec) < 0) return -1;
    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_inv = o.inv; cfg.opt_auto = o.aut;                   /* a header of the other polarity is skipped, or flips the polarity with --auto (:1186-1189) */
    cfg.keep_soft = 1;

    sonde_info_t info;
    int nb = sonde_mrz_dec_frame_bits(dec);
    sonde_engine_t *eng = make_engine(&cfg, in.fq, in.lpiq_bw, nb, baud, thres, shift, &info);
    if (!eng) return -1;
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    const size_t unit = cli_sample_bytes(&cfg, &in);
    int chunk = cfg.sample_rate / 10;
    chunk -= chunk % info.decM;
    if (chunk < info.decM) chunk = info.decM;
    /* input history: the samples since hist0 (an input sample index, multiple of decM) — enough to go back to the end of a block when the
     * frame length has to change */
    const int64_t keep = (int64_t)cfg.sample_rate * 3;
    size_t cap = (size_t)(keep + 2 * (int64_t)chunk) * unit;
    char *hist = (char *)malloc(cap);
    static float s0[MAXHITS * MAXBITS];
    static sonde_hit_t hits[MAXHITS];
    if (!hist) return -1;
    int64_t hist0 = 0, hist_n = 0;            /* first sample held / samples held */
    int64_t fed = 0;                          /* next input sample for the engine */
    int64_t eng0 = 0;                         /* input sample the current engine started at */
    int64_t skip_before = 0;                  /* IF-rate position: hits before it belong to a block already decoded */
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
            sonde_engine_fetch_soft(eng, s0, k);
            for (int i = 0; i < k && !restarted; i++) {
                const int64_t pos = eng0 / info.decM + (int64_t)hits[i].mv_pos;           /* IF-rate position in the whole stream */
                if (pos < skip_before) continue;
                const int m = sonde_mrz_dec_frame(dec, s0 + (size_t)i * nb, hits[i].nbits, out, sizeof out);
                if (m > 0) fwrite(out, 1, (size_t)m, stdout);
                const int want = sonde_mrz_dec_frame_bits(dec);
                if (want != nb) {                                /* the other frame type from the end of this frame on */
                    const double spb_if = 2.0 * (double)info.if_sr / baud;
                    const double frame_end = (double)pos + hits[i].nbits * spb_if;
                    int64_t from = ((int64_t)frame_end - (int64_t)(32 * spb_if)) * info.decM;
                    from -= from % info.decM;
                    if (from < hist0) from = hist0;
                    if (from > fed) from = fed;
                    sonde_engine_destroy(eng);
                    nb = want;
                    eng = make_engine(&cfg, in.fq, in.lpiq_bw, nb, baud, thres, shift, &info);
                    if (!eng) return -1;
                    eng0 = from; fed = from; skip_before = (int64_t)frame_end;
                    restarted = 1;
                }
            }
        }
        if (at_end && !restarted) break;
    }
    sonde_engine_destroy(eng);
    sonde_mrz_dec_destroy(dec);
    free(hist);
    return 0;
}
