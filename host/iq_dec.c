/*
 * host/iq_dec.c — `iq_dec` command-line front end on top of libsonde_hip's front-end-only engine (C).
 *
 * Reference contract kept (reference demod/mod/iq_dec.c:970-1157; callers auto_rx/autorx/sdr_wrappers.py:319-323
 * `iq_dec --bo 16 [--IFbw k] - <sr> 16` as DC block / decimator, and decode.py:808
 * `iq_dec --FM --IFbw k --lpFM --wav --iq 0.0 - <sr> 16` as FM demodulator):
 *     iq_dec [--iq <fq>] [--IFbw kHz] [--lpIQ | --lpbw kHz] [--FM] [--lpFM] [--decFM] [--dc] [--min] [--wav] [--bo 8|16|32] - <sr> 16
 * stdin : interleaved int16 I/Q;  stdout: decimated IQ (cf32 / cs16 / cu8) or the FM discriminator stream
 *         (f32 / s16 / u8), optionally behind a streaming WAV header (iq_dec.c:206-248);  stderr: `IF:` / `dec:`.
 * Sample conversion on output as iq_dec.c:798-936 (x*128[*256], C truncation).  Input: raw IQ (`- <sr> <8|16|32>`) or a 2-channel WAV,
 * on stdin or from a file (iq_dec.c:1051-1080).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_hip.h"
#include "wav_header.h"

static void write_wav_header(int sr, int bps, int nch) {
    uint32_t data = 0;
    fwrite("RIFF", 1, 4, stdout); data = 0; fwrite(&data, 1, 4, stdout); fwrite("WAVE", 1, 4, stdout);
    fwrite("fmt ", 1, 4, stdout); data = 16; if (bps == 32) data += 2; fwrite(&data, 1, 4, stdout);
    data = (bps == 32) ? 3 : 1; fwrite(&data, 1, 2, stdout);
    data = (uint32_t)nch; fwrite(&data, 1, 2, stdout);
    data = (uint32_t)sr; fwrite(&data, 1, 4, stdout);
    data = (uint32_t)sr * (uint32_t)bps / 8; fwrite(&data, 1, 4, stdout);
    data = ((uint32_t)bps + 7) / 8; fwrite(&data, 1, 2, stdout);
    data = (uint32_t)bps; fwrite(&data, 1, 2, stdout);
    if (bps == 32) { data = 0; fwrite(&data, 1, 2, stdout); }
    fwrite("data", 1, 4, stdout); data = 0xFFFFFFFF; fwrite(&data, 1, 4, stdout);
}

static void put_samples(const float *x, int n, int bps) {            /* fwrite_cpx_blk / fwrite_fm_blk */
    if (bps == 32) { fwrite(x, 4, (size_t)n, stdout); return; }
    for (int j = 0; j < n; j++) {
        float v = x[j] * 128.0f;
        if (bps == 8) { v += 128.0f; uint8_t u = (uint8_t)v; fwrite(&u, 1, 1, stdout); }
        else { v *= 256.0f; int16_t b = (int16_t)v; fwrite(&b, 2, 1, stdout); }
    }
}

int main(int argc, char **argv) {
    sonde_cfg_t cfg;
    double fq = 0.0;
    FILE *fp = stdin;
    int have_pcm = 0, opt_fm = 0, opt_decfm = 0, opt_wav = 0, bps_out = 32, if_min = 48000;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.sonde_type = SONDE_FRONTEND;
    cfg.lpiq_bw = 10000;                                          /* iq_dec.c:990 */
    setbuf(stdout, NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "--iqdc")) { /* DC removal is always on for baseband input (iq_dec.c:359) */ }
        else if (!strcmp(a, "--iq")) { if (++i >= argc) return -1; fq = atof(argv[i]); if (fq < -0.5) fq = -0.5; if (fq > 0.5) fq = 0.5; }
        else if (!strcmp(a, "--IFbw")) { if (++i >= argc) return -1; int k = atoi(argv[i]); if (k * 1000 >= 32000) if_min = k * 1000; }
        else if (!strcmp(a, "--lpIQ")) cfg.opt_lp |= SONDE_LP_IQ;
        else if (!strcmp(a, "--noLUT")) cfg.opt_nolut = 1;
        else if (!strcmp(a, "--lpbw")) { if (++i >= argc) return -1; double bw = atof(argv[i]); if (bw > 1.0) cfg.lpiq_bw = (int)(bw * 1e3); cfg.opt_lp |= SONDE_LP_IQ; }
        else if (!strcmp(a, "--FM")) opt_fm = 1;
        else if (!strcmp(a, "--lpFM")) { cfg.opt_lp |= SONDE_LP_FM; opt_fm = 1; }
        else if (!strcmp(a, "--decFM")) { opt_decfm = 1; cfg.opt_lp |= SONDE_LP_FM; opt_fm = 1; }
        else if (!strcmp(a, "--dc")) cfg.opt_lp |= SONDE_LP_FM;          /* iq_dec.c:1073 */
        else if (!strcmp(a, "--min")) cfg.opt_min = 1;
        else if (!strcmp(a, "--wav")) opt_wav = 1;
        else if (!strcmp(a, "--bo")) { if (++i >= argc) return -1; bps_out = atoi(argv[i]); if (bps_out != 8 && bps_out != 16 && bps_out != 32) bps_out = 0; }
        else if (!strcmp(a, "-")) {
            if (i + 2 >= argc) return -1;
            cfg.sample_rate = atoi(argv[++i]); cfg.bits = atoi(argv[++i]);
            if (cfg.sample_rate < 1 || (cfg.bits != 8 && cfg.bits != 16 && cfg.bits != 32)) { fprintf(stderr, "- <sr> <bs>\n"); return -1; }
            have_pcm = 1;
        }
        else if (a[0] != '-') {                                       /* input file instead of stdin (iq_dec.c:1051-1058) */
            fp = fopen(a, "rb");
            if (fp == NULL) { fprintf(stderr, "error: open %s\n", a); return -1; }
        }
        else { fprintf(stderr, "iq_dec (sonde_hip): option %s not supported by this build\n", a); return -1; }
    }
    if (!have_pcm) {                                                  /* WAV: IQ as 2 channels (iq_dec.c:1072-1079, :761) */
        int nch = 0;
        if (wav_read_header(fp, &cfg.sample_rate, &cfg.bits, &nch) < 0) { fprintf(stderr, "error: wav header\n"); return -1; }
        if (nch < 2) { fprintf(stderr, "error: init buffers\n"); return -1; }
        if (nch != 2) { fprintf(stderr, "iq_dec (sonde_hip): WAV input needs 2 channels\n"); return -1; }
    }
    const size_t unit = 2 * (size_t)(cfg.bits / 8);
    cfg.n_channels = 1; cfg.if_rate = if_min;
    cfg.max_chunk = cfg.sample_rate / 4 + 4096;
    sonde_engine_t *eng = NULL;
    int rc = sonde_engine_create(&cfg, &fq, &eng);
    if (rc < 0) { fprintf(stderr, "error: init buffers (%s)\n", sonde_strerror(rc)); return -1; }
    sonde_info_t info;
    sonde_engine_info(eng, &info);
    fprintf(stderr, "IF: %d\n", info.if_sr);
    fprintf(stderr, "dec: %d\n", info.decM);
    int decFM = 1;
    if (opt_decfm) { int fm_sr = info.if_sr; while (fm_sr % 2 == 0 && fm_sr / 2 >= 48000) { fm_sr /= 2; decFM *= 2; } }
    if (opt_wav) write_wav_header(opt_fm ? info.if_sr / decFM : info.if_sr, bps_out, opt_fm ? 1 : 2);

    int chunk = cfg.sample_rate / 4; chunk -= chunk % (info.decM * decFM); if (chunk < info.decM * decFM) chunk = info.decM * decFM;
    int16_t *buf = (int16_t *)malloc((size_t)chunk * unit);
    float *out = (float *)malloc((size_t)(chunk / info.decM + 8) * 8);
    int64_t m_done = 0;                                   /* IF samples written so far */
    int failed = 0;
    const int tap = opt_fm ? SONDE_TAP_FM : ((cfg.opt_lp & SONDE_LP_IQ) ? SONDE_TAP_IFIQ : SONDE_TAP_DECIM);
    for (;;) {
        size_t got = fread(buf, unit, (size_t)chunk, fp);
        got -= got % (size_t)(info.decM * decFM);                      /* whole output samples only (if_fm returns EOF mid-block) */
        if (got == 0) break;
        int erc = sonde_engine_process_host(eng, buf, (int64_t)got, (int32_t)got);
        const int n_if = (int)(got / (size_t)info.decM);
        if (erc >= 0) erc = sonde_engine_read_tap(eng, 0, tap, m_done, n_if, out);
        if (erc < 0) { fprintf(stderr, "iq_dec: the engine failed (%s)\n", sonde_strerror(erc)); failed = 1; break; }     /* never a silent 0 */
        if (opt_fm) {
            int k = 0;
            for (int m = decFM - 1; m < n_if; m += decFM) out[k++] = out[m];       /* s_fm of the last sub-sample (iq_dec.c:551-618) */
            put_samples(out, k, bps_out);
        } else put_samples(out, 2 * n_if, bps_out);
        m_done += n_if;
        if (got < (size_t)chunk) break;
    }
    sonde_engine_destroy(eng);
    free(buf); free(out);
    return failed ? -1 : 0;
}
