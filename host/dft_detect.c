/*
 * host/dft_detect.c — `dft_detect` command-line front end on top of libsonde_hip's scanner (C).
 *
 * Reference contract kept (reference scan/dft_detect.c:1368-1455 argv, :1612-1634 stdout, :1656-1666 exit code;
 * callers auto_rx/autorx/scan.py:541-547 (IQ) and :600 (FM audio)):
 *     dft_detect [-v] [-c] [-s] [-d2] [-t sec] [--ths x] [--dc] [--min] [-L] [--bw kHz] [--ch2]
 *                ( --iq | --IQ <fq> ) - <sr> <bits>          raw IQ on stdin
 *     dft_detect [...] [file.wav]                            FM audio, WAV on stdin or from a file
 * stdout: one line per detection `TYPE: %.4f[ , %+.1fHz]` (-v: `sample: n` before it, `[hhhh]` for M10/M20);
 * stderr: `IF:`/`dec:` (--IQ) or the WAV header summary; exit code = header_found * type number (negative for
 * inverted DFM/RS41/RS92), -50 on errors — all modulo 256 as seen by the shell.
 * 8-bit unsigned, 16-bit signed and 32-bit float input.
 *
 * Batch form (not in the reference; what auto_rx's detect_sonde() loop over the peaks of one scan step, scan.py:378-760, can call once):
 *     dft_detect [...] --IQ <fq1>,<fq2>,... - <sr> <bits>
 * scans every listed offset of the ONE stream on stdin in the same launches (scanner channels on a shared stream).  stdout: the reference's
 * line prefixed by `<index> <fq> ` per detection and, at the end, `# <index> <fq> <code>` with the exit code the single run would return;
 * the process exits 0.  A single <fq> behaves exactly like the reference.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_scan.h"

static int find4(FILE *fp, const char *tag) {       /* scan forward to a 4-char chunk tag (dft_detect.c:452-503) */
    char w[4] = { 0, 0, 0, 0 };
    int c;
    while ((c = fgetc(fp)) != EOF) {
        w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = (char)c;
        if (!memcmp(w, tag, 4)) return 0;
    }
    return -1;
}

static int read_wav_header(FILE *fp, int *sr, int *bits, int *nch) {
    unsigned char d[16];
    char t[4];
    if (fread(t, 1, 4, fp) < 4 || (strncmp(t, "RIFF", 4) && strncmp(t, "RF64", 4))) return -1;
    if (fread(t, 1, 4, fp) < 4) return -1;
    if (fread(t, 1, 4, fp) < 4 || strncmp(t, "WAVE", 4)) return -1;
    if (find4(fp, "fmt ") < 0) return -1;
    if (fread(d, 1, 4, fp) < 4) return -1;          /* chunk size  */
    if (fread(d, 1, 2, fp) < 2) return -1;          /* format tag  */
    if (fread(d, 1, 2, fp) < 2) return -1;
    *nch = d[0] + (d[1] << 8);
    if (fread(d, 1, 4, fp) < 4) return -1;
    *sr = (int)((uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
    if (fread(d, 1, 4, fp) < 4) return -1;          /* byte rate   */
    if (fread(d, 1, 2, fp) < 2) return -1;          /* block align */
    if (fread(d, 1, 2, fp) < 2) return -1;
    *bits = d[0] + (d[1] << 8);
    if (find4(fp, "data") < 0) return -1;
    if (fread(d, 1, 4, fp) < 4) return -1;
    fprintf(stderr, "sample_rate: %d\n", *sr);
    fprintf(stderr, "bits       : %d\n", *bits);
    fprintf(stderr, "channels   : %d\n", *nch);
    if (*bits != 8 && *bits != 16 && *bits != 32) return -1;
    if (*sr == 900001) *sr -= 1;
    return 0;
}

static void print_detections(sonde_scan_t *sc, int verbose, int silent, int batch, const double *fqs) {
    sonde_detection_t det[16];
    int n;
    while ((n = sonde_scan_fetch(sc, det, 16)) > 0)
        for (int k = 0; k < n; k++) {
            char line[256];
            if (silent || !det[k].printed) continue;
            sonde_scan_line(sc, &det[k], verbose, line, sizeof line);
            if (!batch) { fprintf(stdout, "%s\n", line); continue; }
            for (char *l = line; l; ) {                          /* -v: two physical lines per detection; each gets the prefix */
                char *nl = strchr(l, '\n');
                if (nl) *nl = 0;
                fprintf(stdout, "%d %.6f %s\n", det[k].channel, fqs[det[k].channel], l);
                l = nl ? nl + 1 : NULL;
            }
        }
}

int main(int argc, char **argv) {
    sonde_scan_cfg_t cfg;
    FILE *fp = stdin;
    double fq = 0.0;
    double fqs[256]; int nfq = 0;                        /* --IQ fq1,fq2,...: batch form */
    int verbose = 0, silent = 0, pcmraw = 0, wavloaded = 0, wav_channel = 0, channels = 0;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.iq_mode = SONDE_SCAN_AUDIO;
    cfg.time_limit = -1.0f;
    setbuf(stdout, NULL);
    for (int i = 1; i < argc && !wavloaded; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) {
            fprintf(stderr, "%s [options] audio.wav\n", argv[0]);
            fprintf(stderr, "  options:\n       -v          (verbose)\n       -c          (continuous)\n");
            fprintf(stderr, "       --iq        (IF iq-data)\n       --IQ <fq>   (baseband IQ at fq)\n       --bw <kHz>  (set IQ filter bw/kHz)\n");
            return 0;
        }
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) verbose = 1;
        else if (!strcmp(a, "--iq")) cfg.iq_mode = SONDE_SCAN_IFIQ;
        else if (!strcmp(a, "--IQ")) {
            if (++i >= argc) return -1;
            for (const char *q = argv[i]; q && *q && nfq < 256; ) {
                double v = atof(q);
                if (v < -0.5) v = -0.5;
                if (v > 0.5) v = 0.5;
                fqs[nfq++] = v;
                q = strchr(q, ',');
                if (q) q++;
            }
            if (nfq < 1) return -1;
            fq = fqs[0];
            cfg.iq_mode = SONDE_SCAN_BBIQ;
        }
        else if (!strcmp(a, "--bw")) {
            if (++i >= argc) return -1;
            double bw = atof(argv[i]);
            if (bw < 1.0) bw = 0.0;
            cfg.bw_khz = (float)bw;
        }
        else if (!strcmp(a, "--dc")) cfg.opt_dc = 1;
        else if (!strcmp(a, "--min")) cfg.opt_min = 1;
        else if (!strcmp(a, "-L")) cfg.opt_lband = 1;
        else if (!strcmp(a, "-c") || !strcmp(a, "--cnt")) cfg.opt_cont = 1;
        else if (!strcmp(a, "-s") || !strcmp(a, "--silent")) silent = 1;
        else if (!strcmp(a, "-t") || !strcmp(a, "--time")) { if (++i >= argc) return -50; cfg.time_limit = (float)atof(argv[i]); }
        else if (!strcmp(a, "-d2")) cfg.opt_d2 = 1;
        else if (!strcmp(a, "--ch2")) wav_channel = 1;
        else if (!strcmp(a, "--ths")) { if (++i >= argc) return -50; cfg.ths = (float)atof(argv[i]); }
        else if (!strcmp(a, "-")) {
            if (i + 2 >= argc) return -1;
            cfg.sample_rate = atoi(argv[++i]);
            cfg.bits = atoi(argv[++i]);
            channels = 2;
            if (cfg.sample_rate < 1 || (cfg.bits != 8 && cfg.bits != 16 && cfg.bits != 32)) { fprintf(stderr, "- <sr> <bs>\n"); return -1; }
            pcmraw = 1;
        }
        else {
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -50; }
            wavloaded = 1;
        }
    }
    if (cfg.opt_d2) cfg.opt_cont = 0;
    if (!pcmraw) {
        if (read_wav_header(fp, &cfg.sample_rate, &cfg.bits, &channels) < 0) { fclose(fp); fprintf(stderr, "error: wav header\n"); return -50; }
    }
    if (cfg.iq_mode != SONDE_SCAN_AUDIO && channels < 2) { fprintf(stderr, "error: iq channels < 2\n"); return -50; }
    if (channels < 1) channels = 1;
    cfg.audio_channels = channels;
    cfg.audio_select = (wav_channel >= 0 && wav_channel < channels) ? wav_channel : 0;
    cfg.n_channels = nfq > 1 ? nfq : 1;
    const int batch = nfq > 1;

    /* a quarter second per call keeps the reaction time of the blocking reference; multiples of decM for --IQ */
    int chunk = cfg.sample_rate / 4;
    if (chunk < 4096) chunk = 4096;
    chunk -= chunk % 64;
    cfg.max_chunk = chunk + 64;
    sonde_scan_t *sc = NULL;
    int rc = sonde_scan_create(&cfg, batch ? fqs : &fq, &sc);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -50; }
    sonde_scan_info_t info;
    sonde_scan_info(sc, &info);
    if (cfg.iq_mode == SONDE_SCAN_BBIQ) { fprintf(stderr, "IF: %d\n", info.if_sr); fprintf(stderr, "dec: %d\n", info.decM); }
    chunk -= chunk % info.decM;

    const size_t unit = (cfg.iq_mode == SONDE_SCAN_AUDIO ? (size_t)channels : 2) * (size_t)(cfg.bits / 8);
    unsigned char *buf = (unsigned char *)malloc((size_t)chunk * unit);
    if (!buf) return -50;
    for (;;) {
        size_t got = fread(buf, unit, (size_t)chunk, fp);
        got -= got % (size_t)info.decM;
        if (got == 0) break;
        rc = sonde_scan_process_host(sc, buf, batch ? 0 : (int64_t)got, (int32_t)got);      /* stride 0: all channels read the one stream */
        if (rc < 0) { fprintf(stderr, "error: %s\n", sonde_strerror(rc)); sonde_scan_destroy(sc); free(buf); return -50; }      /* the reference's error code (dft_detect.c:1446-1455), not a detection result */
        print_detections(sc, verbose, silent, batch, fqs);
        int all_done = 1;
        for (int c = 0; c < cfg.n_channels; c++) all_done &= sonde_scan_channel_done(sc, c) == 1;
        if (all_done) break;
        if (got < (size_t)chunk) break;
    }
    sonde_scan_finish(sc);
    print_detections(sc, verbose, silent, batch, fqs);
    if (batch) {
        for (int c = 0; c < cfg.n_channels; c++) { int32_t code = 0; sonde_scan_result(sc, c, &code); fprintf(stdout, "# %d %.6f %d\n", c, fqs[c], (int)code); }
        sonde_scan_destroy(sc); free(buf); fclose(fp);
        return 0;
    }
    int32_t code = 0;
    sonde_scan_result(sc, 0, &code);
    sonde_scan_destroy(sc);
    free(buf);
    fclose(fp);
    return code;
}
