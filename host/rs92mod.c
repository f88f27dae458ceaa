/*
 * host/rs92mod.c — `rs92mod` command-line front end on top of libsonde_hip (C).
 *
 * Reference contract (demod/mod/rs92mod.c:1642-1832 argv, :1389-1575 output, :1985-2050 frame loop, :2058-2084 --rawhex):
 *     rs92mod [-r] [-v | -vv] [-vx] [--crc] [--ecc | --ecc2] [--ptu] [-i] [--ngp] [--dbg] [--json] [--jsn_cfq hz] [--ths x] [-d shift]
 *             [-e rinex_nav | -a sem_almanac [--gpsepoch n]] [--vel | --vel1 | --vel2] [--iter] [-g1 | -g2 | -gg] [--dop x] [--der x] [--exsat prn]
 *             ( --IQ <fq> | --iq0 | --iq2 | --iq3 [--iqdc] ) [--lpIQ | --lpbw kHz] [--lpFM] [--dc] [--min] - <sr> <bits>      IQ samples
 *     rs92mod [...] [--ch2] [file.wav]                                                                                      FM audio
 *     rs92mod [...] --softin | --softinv                            float32 soft symbols (fsk_demod -s)
 *     rs92mod [...] --rawhex                                        one frame per line as hex
 * auto_rx: `rs92mod -vx -v --crc --ecc --vel --json -e <eph> [--softin -i]` (decode.py:484,985).  As in the reference the argument list ends
 * at the file name, and options are read in order (--json sets --ecc2 / --vel, a later --ecc takes the count back out).
 * stdout: per frame with a valid config block one line (frame number, id, date / time, position when orbit data was given and four
 * satellites are in view, CRC flags, the calibration row's frequency), --json the JSON object; -r the bytes.
 * Exit 0 at EOF, 255 on argument / init errors.  --spike is refused: the reference's clipping reads an uninitialised variable
 * (demod_mod.c:971,999 through read_slbit).
 *
 * The sample-rate part runs in the engine (generic sonde description: header 2A 2A 10 as 60 raw symbols, 4800 Bd, two symbols per bit,
 * BT 0.5, h 0.8, 3 header errors, 2340 bits per hit, centre window 4 for IF-rate IQ); everything behind a hit is sonde_rs92.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "sonde_rs92.h"
#include "cli_common.h"

#define MAXHITS 8
static const char kHeader[] = "10100110011001101001" "1010011001100110100110101010100110101001";      /* rs92mod.c:88-92 */

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    sonde_rs92_opts_t o;
    cli_in_t in;
    int spike = 0, softin = 0, rawhex = 0, cfreq = -1, shift = 0, oc;
    float thres = 0.7f;
    const char *eph_path = NULL, *alm_path = NULL;
    FILE *fp = stdin;
    static char out[1 << 18];
    memset(&o, 0, sizeof o);
    o.gpsepoch = -1;
    cli_in_init(&in, 8000, 48.0);
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_GENERIC;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] ( file.wav | --IQ <fq> - <sr> <bits> | --softin | --rawhex )\n", argv[0]);
            fprintf(stderr, "  orbit data:  -e <rinex nav file> | -a <SEM almanac> [--gpsepoch n]\n");
            fprintf(stderr, "  solution:    --vel | --vel1 | --vel2, --iter, -g1 | -g2 | -gg, --dop x, --der x, --exsat prn\n");
            fprintf(stderr, "  output:      -v | -vv, -vx, -r, --ptu, --ecc | --ecc2, --json [--jsn_cfq hz]\n");
            fprintf(stderr, "  signal:      -i, --ngp, --ths x (default %.1f), -d shift\n", thres);
            return 0;
        }
        else if (!strcmp(a, "--vel")) o.gps_vel = 4;
        else if (!strcmp(a, "--vel1")) { o.gps_vel = 1; if (o.gps_verbose < 1) o.gps_verbose = 2; }
        else if (!strcmp(a, "--vel2")) { o.gps_vel = 2; if (o.gps_verbose < 1) o.gps_verbose = 2; }
        else if (!strcmp(a, "--iter")) o.gps_iter = 1;
        else if (!strcmp(a, "-v")) o.verbose = 1;
        else if (!strcmp(a, "-vv")) o.verbose = 4;
        else if (!strcmp(a, "-vx")) o.aux = 1;
        else if (!strcmp(a, "--crc")) { /* always on (:1865) */ }
        else if (!strcmp(a, "--ecc")) o.ecc = 1;
        else if (!strcmp(a, "--ecc2")) o.ecc = 2;
        else if (!strcmp(a, "--ptu")) o.ptu = 1;
        else if (!strcmp(a, "-r") || !strcmp(a, "--raw")) o.raw = 1;
        else if (!strcmp(a, "-i") || !strcmp(a, "--invert")) o.inv = 1;
        else if (!strcmp(a, "-e") || !strncmp(a, "--ephem", 7)) {
            if (++i >= argc) return -1;
            eph_path = argv[i];
            FILE *t = fopen(eph_path, "rb");
            if (t == NULL) { fprintf(stderr, "[rinex] %s konnte nicht geoeffnet werden\n", eph_path); eph_path = NULL; } else fclose(t);
        }
        else if (!strcmp(a, "-a") || !strcmp(a, "--almanac")) {
            if (++i >= argc) return -1;
            alm_path = argv[i];
            FILE *t = fopen(alm_path, "r");
            if (t == NULL) { fprintf(stderr, "[almanac] %s konnte nicht geoeffnet werden\n", alm_path); alm_path = NULL; } else fclose(t);
        }
        else if (!strcmp(a, "--gpsepoch")) { if (++i >= argc) return -1; o.gpsepoch = atoi(argv[i]); if (o.gpsepoch < 0 || o.gpsepoch > 4) o.gpsepoch = 1; }
        else if (!strcmp(a, "--dop")) { if (++i >= argc) return -1; o.dop_limit = (float)atof(argv[i]); if (o.dop_limit <= 0 || o.dop_limit >= 100) o.dop_limit = 9.9f; }
        else if (!strcmp(a, "--der")) { if (++i >= argc) return -1; o.d_err = (float)atof(argv[i]); if (o.d_err <= 0 || o.d_err >= 100000) o.d_err = 0; }
        else if (!strcmp(a, "--exsat")) { if (++i >= argc) return -1; o.exsat = atoi(argv[i]); if (o.exsat < 1 || o.exsat > 32) o.exsat = -1; }
        else if (!strcmp(a, "-g1")) o.gps_verbose = 1;
        else if (!strcmp(a, "-g2")) o.gps_verbose = 2;
        else if (!strcmp(a, "-gg")) o.gps_verbose = 8;
        else if (!strcmp(a, "--json")) { o.json = 1; o.ecc = 2; o.gps_vel = 4; }
        else if (!strcmp(a, "--jsn_cfq")) { if (++i >= argc) return -1; cfreq = atoi(argv[i]); if (cfreq < 300000000) cfreq = -1; }
        else if (!strcmp(a, "--spike")) spike = 1;
        else if (!strcmp(a, "--softin")) softin = 1;
        else if (!strcmp(a, "--softinv")) softin = 2;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -1; thres = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d")) { if (++i >= argc) return -1; shift = atoi(argv[i]); if (shift > 4) shift = 4; if (shift < -4) shift = -4; }
        else if (!strcmp(a, "--ngp")) o.ngp = 1;
        else if (!strcmp(a, "--dbg")) o.dbg = 1;
        else if (!strcmp(a, "--rawhex")) rawhex = 2;
        else if ((oc = cli_input_option(argc, argv, &i, &cfg, &in)) != 0) { if (oc < 0) return -1; }      /* --IQ, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw, --min, --ch2, "- <sr> <bits>" */
        else if (a[0] != '-') {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
            break;                                               /* the reference stops reading arguments here (:1644) */
        }
        else { fprintf(stderr, "rs92mod (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    cli_json_version(o.version, sizeof o.version);
    o.jsn_freq_khz = cfreq > 0 ? (cfreq + 500) / 1000 : 0;
    const int ngp0 = o.ngp;

    if (!rawhex && !softin) {
        if (cli_input_setup("rs92mod", fp, &cfg, &in) < 0) return -1;
        if (spike && in.iq_mode < 2) { fprintf(stderr, "rs92mod (sonde_hip): --spike is not supported (undefined in the reference: demod_mod.c:971,999)\n"); return -1; }
        if (cfreq > 0) o.jsn_freq_khz = (int)((cfreq - (in.iq_mode == 5 ? -in.fq : 0.0) * cfg.sample_rate + 500) / 1e3);
        if ((float)cfg.sample_rate / 4800.0f < 8) fprintf(stderr, "note: sample rate low (%.1f sps)\n", (float)cfg.sample_rate / 4800.0f);
    }
    sonde_rs92_dec_t *dec = NULL;
    if (sonde_rs92_dec_create(&o, &dec) < 0) return -1;
    if (alm_path) sonde_rs92_dec_load_almanac(dec, alm_path);                  /* (:1834-1855) */
    if (eph_path) sonde_rs92_dec_load_ephemeris(dec, eph_path);

    if (rawhex) {                                                /* :2058-2084 */
        static char lb[2 * SONDE_RS92_FRAME_LEN + 12];
        static uint8_t fr[SONDE_RS92_FRAME_LEN];
        uint8_t b = 0;
        while (fgets(lb, 2 * SONDE_RS92_FRAME_LEN + 12, fp)) {
            lb[2 * SONDE_RS92_FRAME_LEN] = '\0';
            char *sp = strchr(lb, ' ');
            if (sp != NULL && sp - lb < 2 * SONDE_RS92_FRAME_LEN) *sp = '\0';
            const int len = (int)strlen(lb) / 2;
            if (len <= 0x48 + 4) continue;
            for (int i = 0; i < len; i++) { sscanf(lb + 2 * i, "%2hhx", &b); fr[i] = b; }
            const int n = sonde_rs92_dec_bytes(dec, fr, len, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
        }
        sonde_rs92_dec_destroy(dec);
        return 0;
    }
    if (softin) {
        float sb[1024];
        for (;;) {
            const size_t got = fread(sb, 4, 1024, fp);
            const int n = sonde_rs92_dec_push_soft(dec, sb, (int32_t)got, softin == 2, got < 1024, out, sizeof out);
            if (n > 0) fwrite(out, 1, (size_t)n, stdout);
            if (got < 1024) break;
        }
        sonde_rs92_dec_destroy(dec);
        return 0;
    }

    cfg.n_channels = 1;
    cfg.max_chunk = cfg.sample_rate;
    cfg.max_frames = MAXHITS;
    cfg.opt_inv = o.inv; cfg.opt_auto = 0;                       /* a header of the other polarity is skipped (:1999-2002; no --auto in this decoder) */
    cfg.keep_soft = 1;

    sonde_generic_t g;
    memset(&g, 0, sizeof g);
    strcpy(g.header, kHeader);
    g.baud = 4800.0f; g.bt = 0.5f; g.h = 0.8f; g.symlen = 2; g.symhd = 2;                                /* rs92mod.c:1914-1943 */
    if (ngp0) g.h = 3.8f;                                                                              /* 1680 MHz RS92-NGP: 4.2 times the deviation */
    g.hdmax = 3; g.bitofs = 2 + shift;                                                                /* :1619,1957,1992 */
    g.nbits = SONDE_RS92_FRAME_BITS;
    g.l_win = 4.0f;                                                                                    /* bl = 4.0 for opt_iq > 2, whole bits else (:2026-2028) */
    g.lpiq_bw = (!in.have_lpbw && ngp0) ? 32000 : in.lpiq_bw; g.lpfm_bw = 6000;      /* --ngp default 32 kHz, an explicit --lpbw wins (rs92mod.c:1940-1944) */
    sonde_engine_t *eng = NULL;
    int rc = sonde_engine_create_generic(&cfg, &in.fq, &g, &eng);
    if (rc >= 0) rc = sonde_engine_set_threshold(eng, thres);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
    sonde_info_t info;
    sonde_engine_info(eng, &info);
    if (in.iq_mode == 5) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }

    cli_reader_t rd;
    static float s0[MAXHITS * SONDE_RS92_FRAME_BITS];
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
                const int m = sonde_rs92_dec_frame(dec, s0 + (size_t)i * SONDE_RS92_FRAME_BITS, hits[i].nbits, out, sizeof out);
                if (m > 0) fwrite(out, 1, (size_t)m, stdout);
            }
        }
    }
    sonde_engine_destroy(eng);
    sonde_rs92_dec_destroy(dec);
    cli_reader_free(&rd);
    return 0;
}
