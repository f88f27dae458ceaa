/*
 * host/cli_common.h — what the decoder front ends share.
 *
 * Every <sonde>mod.c of the reference repeats the same strcmp chain for its sample input (rs41mod.c:2617-2744, dfm09mod.c:1357-1500,
 * m10mod.c:1180-1330, ... : --IQ fq, --iq0/2/3, --iqdc, --noLUT, --dc, --lpIQ, --lpFM, --lpbw kHz, --min, --ch2, "- <sr> <bits>"), the same
 * checks behind it (raw data must be IQ, WAV header, two channels for IQ in a WAV) and the same block-wise read of stdin; so did the front
 * ends here.  This header holds that part once: cli_input_option() for the argument loop, cli_input_setup() behind it,
 * cli_json_version(), and the block reader (whole multiples of the decimation per engine call, the remainder kept for the next read).
 * What differs between the decoders stays in their files: defaults, the --lpbw range, everything behind a header hit.
 */
#ifndef SONDE_CLI_COMMON_H
#define SONDE_CLI_COMMON_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_hip.h"
#include "wav_header.h"

typedef struct {
    double fq;            /* --IQ <fq>, clipped to +-0.5 like the reference */
    int have_iq;          /* one of --IQ / --iq0 / --iq2 / --iq3 was given */
    int iq_mode;          /* the reference's option_iq: 5 = --IQ, 1 / 2 / 3 = --iq0 / --iq2 / --iq3, 0 = FM audio */
    int have_pcm;         /* "- <sr> <bits>": headerless samples */
    int wav_ch;           /* --ch2 */
    int nch;              /* channels of the WAV file */
    int lpiq_bw;          /* IF low-pass bandwidth in Hz (--lpbw kHz inside (4.6, lpbw_max)), the decoder's default otherwise */
    double lpbw_max;
    int have_lpbw;        /* --lpbw was given with a value inside its range (the reference's set_lpIQbw > 0) */
} cli_in_t;

static inline void cli_in_init(cli_in_t *in, int lpiq_bw_default, double lpbw_max) {
    memset(in, 0, sizeof *in);
    in->nch = 1; in->lpiq_bw = lpiq_bw_default; in->lpbw_max = lpbw_max;
}

/* argv[*pi] against the shared input options: 1 = it was one (arguments consumed, *pi on the last of them), 0 = not one of these,
 * -1 = one of these with a missing / bad argument (the caller returns -1 as the reference does) */
static inline int cli_input_option(int argc, char **argv, int *pi, sonde_cfg_t *cfg, cli_in_t *in) {
    const char *a = argv[*pi];
    if (!strcmp(a, "--IQ")) {
        if (++*pi >= argc) return -1;
        in->fq = atof(argv[*pi]);
        if (in->fq < -0.5) in->fq = -0.5;
        if (in->fq > 0.5) in->fq = 0.5;
        in->have_iq = 1; in->iq_mode = 5;
    }
    else if (!strcmp(a, "--iq0")) { in->have_iq = 1; in->iq_mode = 1; }      /* IF-rate IQ, FM discriminator */
    else if (!strcmp(a, "--iq2")) { in->have_iq = 1; in->iq_mode = 2; }
    else if (!strcmp(a, "--iq3")) { in->have_iq = 1; in->iq_mode = 3; }
    else if (!strcmp(a, "--iqdc")) cfg->opt_iqdc = 1;
    else if (!strcmp(a, "--noLUT")) cfg->opt_nolut = 1;                        /* --IQ only, like the reference */
    else if (!strcmp(a, "--dc")) cfg->opt_dc = 1;
    else if (!strcmp(a, "--lpIQ")) cfg->opt_lp |= SONDE_LP_IQ;
    else if (!strcmp(a, "--lpFM")) cfg->opt_lp |= SONDE_LP_FM;
    else if (!strcmp(a, "--lpbw")) {
        if (++*pi >= argc) return -1;
        const double bw = atof(argv[*pi]);
        if (bw > 4.6 && bw < in->lpbw_max) { in->lpiq_bw = (int)(bw * 1e3); in->have_lpbw = 1; }
        cfg->opt_lp |= SONDE_LP_IQ;
    }
    else if (!strcmp(a, "--min")) cfg->opt_min = 1;
    else if (!strcmp(a, "--ch2")) in->wav_ch = 1;
    else if (!strcmp(a, "-")) {
        if (*pi + 2 >= argc) return -1;
        cfg->sample_rate = atoi(argv[++*pi]);
        cfg->bits = atoi(argv[++*pi]);
        if (cfg->sample_rate < 1 || (cfg->bits != 8 && cfg->bits != 16 && cfg->bits != 32)) { fprintf(stderr, "- <sr> <bs>\n"); return -1; }
        in->have_pcm = 1;
    }
    else return 0;
    return 1;
}

/* behind the argument loop, for the sample forms: headerless input must be IQ, WAV header otherwise (two channels for IQ), --dc with --IQ
 * implies the FM low-pass, --noLUT only with --IQ, cfg->input / audio channel from the form chosen.  0, or -1 with the message printed. */
static inline int cli_input_setup(const char *prog, FILE *fp, sonde_cfg_t *cfg, cli_in_t *in) {
    if (!in->have_iq && in->have_pcm) { fprintf(stderr, "error: raw data not IQ\n"); return -1; }
    if (!in->have_pcm && wav_read_header(fp, &cfg->sample_rate, &cfg->bits, &in->nch) < 0) { fprintf(stderr, "error: wav header\n"); return -1; }
    if (in->have_iq && !in->have_pcm && in->nch != 2) { fprintf(stderr, "%s (sonde_hip): IQ input needs 2 channels\n", prog); return -1; }
    if (in->iq_mode == 5 && cfg->opt_dc) cfg->opt_lp |= SONDE_LP_FM;
    if (in->iq_mode != 5) cfg->opt_nolut = 0;
    if (in->have_iq) cfg->input = in->iq_mode == 5 ? SONDE_IN_IQ : in->iq_mode == 1 ? SONDE_IN_IFIQ0 : in->iq_mode == 2 ? SONDE_IN_IFIQ2 : SONDE_IN_IFIQ3;
    else {
        cfg->input = SONDE_IN_AUDIO; cfg->audio_channels = in->nch < 1 ? 1 : in->nch;
        cfg->audio_select = (in->wav_ch < cfg->audio_channels) ? in->wav_ch : 0;
    }
    return 0;
}

/* bytes of one sample of the chosen form (IQ pair, or one frame of the WAV's channels) */
static inline size_t cli_sample_bytes(const sonde_cfg_t *cfg, const cli_in_t *in) {
    return (in->have_iq ? 2 : (size_t)cfg->audio_channels) * (size_t)(cfg->bits / 8);
}

/* "version" of the JSON output: SONDE_JSN_VERSION in the environment, else what the build defines (the reference: -DVER_JSN_STR) */
static inline void cli_json_version(char *dst, size_t cap) {
    const char *ver = getenv("SONDE_JSN_VERSION");
#ifdef VER_JSN_STR
    if (!ver) ver = VER_JSN_STR;
#endif
    if (ver && cap) { strncpy(dst, ver, cap - 1); dst[cap - 1] = 0; }
}

/* Block reader: ~0.1 s of input per engine call, always a whole multiple of the decimation; what does not fill a multiple stays in the
 * buffer for the next read. */
typedef struct { char *buf; size_t have, unit; int chunk, decM; } cli_reader_t;

static inline int cli_reader_init(cli_reader_t *r, size_t unit, int sample_rate, int decM) {
    r->unit = unit; r->decM = decM < 1 ? 1 : decM; r->have = 0;
    r->chunk = sample_rate / 10;
    r->chunk -= r->chunk % r->decM;
    if (r->chunk < r->decM) r->chunk = r->decM;
    r->buf = (char *)malloc((size_t)r->chunk * unit);
    return r->buf ? 0 : -1;
}
/* one read: returns the bytes it got (0 = end of input); *n = samples ready at r->buf (a multiple of decM, possibly 0) */
static inline size_t cli_reader_fill(cli_reader_t *r, FILE *fp, int *n) {
    const size_t got = fread(r->buf + r->have, 1, (size_t)r->chunk * r->unit - r->have, fp);
    r->have += got;
    int k = (int)(r->have / r->unit);
    k -= k % r->decM;
    *n = k;
    return got;
}
static inline void cli_reader_consume(cli_reader_t *r, int n) {
    memmove(r->buf, r->buf + (size_t)n * r->unit, r->have - (size_t)n * r->unit);
    r->have -= (size_t)n * r->unit;
}
static inline void cli_reader_free(cli_reader_t *r) { free(r->buf); r->buf = NULL; }
#endif
