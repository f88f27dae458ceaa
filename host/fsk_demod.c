/*
 * host/fsk_demod.c — `fsk_demod` command-line front end on top of libsonde_hip's 2-/4-FSK modem (C).
 *
 * Reference contract kept (utils/fsk_demod.c; callers auto_rx/autorx/decode.py:901,976,1067,1120):
 *     fsk_demod [--cs16|--cu8] [-s] [-i] [-b lo] [-u hi] [--mask S] [--nsym=N] [-p P] [--stats[=r]] [--testframes] (2|4) <Fs> <Rs> <in|-> <out|->
 * stdin : complex int16 (--cs16), complex uint8 (--cu8) or real int16;
 * stdout: Nbits float32 soft decisions per modem frame (-s; -i negates) or one byte per bit, flushed per frame;
 * stderr: `Setting estimator limits to a to b Hz.`; with --stats one JSON object per line every 1/(r*loop_time) frames carrying
 *         samples / EbNodB / ppm / f1_est.. / eye_diagram / samp_fft (what auto_rx/autorx/fsk_demod.py:23 parses); with --testframes
 *         the bit-error count against the 100-bit test frame of fsk_get_test_bits (srand(158324), rand() & 1).
 * exit 0 at EOF / SIGTERM, 1 on usage errors.  Each iteration reads exactly fsk_nin() samples, like the reference.
 *
 * Structure: parse() -> job_t; open_modem(); then one pass of frame_io() per modem frame, which hands the frame to the optional
 * test-frame matcher (tf_push) and the optional stats reporter (report).
 */
#include <getopt.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_fsk.h"

#define TF_BITS 100                              /* TEST_FRAME_SIZE (fsk_demod.c:30) */

typedef struct {
    sonde_fsk_cfg_t cfg;
    int soft, soft_negate, stats, stats_rate, testframes;
    int sample_bytes;                            /* bytes of one input sample (all components) */
    int real_input;
    int lower_given, upper_given;
    const char *in_path, *out_path;
} job_t;

/* ---- options ------------------------------------------------------------------------------------------------------------- */

static void usage(const char *prog) {
    fprintf(stderr, "usage: %s [options] (2|4) SampleRate SymbolRate InputModemRawFile OutputFile\n", prog);
    fprintf(stderr, " -c --cs16  -d --cu8  -t[r] --stats=[r]  -s --soft-dec  -i --softinv  -p P  --fsk_lower f  --fsk_upper f  --nsym N  --mask S  -f --testframes\n");
    exit(1);
}

static const struct option kLong[] = {
    {"help", no_argument, 0, 'h'}, {"softinv", no_argument, 0, 'i'}, {"conv", required_argument, 0, 'p'}, {"cs16", no_argument, 0, 'c'},
    {"cu8", no_argument, 0, 'd'}, {"fsk_lower", required_argument, 0, 'b'}, {"fsk_upper", required_argument, 0, 'u'},
    {"stats", optional_argument, 0, 't'}, {"soft-dec", no_argument, 0, 's'}, {"testframes", no_argument, 0, 'f'},
    {"nsym", required_argument, 0, 'n'}, {"mask", required_argument, 0, 'm'}, {0, 0, 0, 0}
};

static void parse(int argc, char **argv, job_t *j) {
    memset(j, 0, sizeof *j);
    j->cfg.abi_version = SONDE_ABI_VERSION; j->cfg.n_channels = 1;
    j->cfg.P = 10; j->cfg.nsym = 50; j->cfg.tone_spacing = 100;           /* fsk_demod.c:63-80 defaults */
    j->cfg.format = SONDE_FSK_S16; j->sample_bytes = 2; j->real_input = 1; j->stats_rate = 8;
    int o;
    while ((o = getopt_long(argc, argv, "fhilp:cdt::sb:u:m", kLong, NULL)) != -1) {
        switch (o) {
        case 'c': j->cfg.format = SONDE_FSK_CS16; j->sample_bytes = 4; j->real_input = 0; break;
        case 'd': j->cfg.format = SONDE_FSK_CU8; j->sample_bytes = 2; j->real_input = 0; break;
        case 'f': j->testframes = 1; break;
        case 'i': j->soft_negate = 1; break;
        case 's': j->soft = 1; break;
        case 't': j->stats = 1; if (optarg) { j->stats_rate = atoi(optarg); if (!j->stats_rate) j->stats_rate = 8; } break;
        case 'p': j->cfg.P = atoi(optarg); break;
        case 'b': if (optarg) { j->cfg.fsk_lower = atoi(optarg); j->lower_given = 1; } break;
        case 'u': if (optarg) { j->cfg.fsk_upper = atoi(optarg); j->upper_given = 1; } break;
        case 'n': if (optarg) j->cfg.nsym = atoi(optarg); break;
        case 'm': j->cfg.mask = 1; j->cfg.tone_spacing = optarg ? atoi(optarg) : 100; break;
        default: usage(argv[0]);
        }
    }
    const int left = argc - optind;
    if (left != 5) { fprintf(stderr, left < 5 ? "Too few arguments\n" : "Too many arguments\n"); usage(argv[0]); }
    j->cfg.M = atoi(argv[optind]); j->cfg.Fs = atoi(argv[optind + 1]); j->cfg.Rs = atoi(argv[optind + 2]);
    j->in_path = argv[optind + 3]; j->out_path = argv[optind + 4];
    if (j->cfg.M != 2 && j->cfg.M != 4) { fprintf(stderr, "Mode %d is not valid. Mode must be 2 or 4.\n", j->cfg.M); usage(argv[0]); }
    if (!j->lower_given) j->cfg.fsk_lower = j->real_input ? 0 : -j->cfg.Fs / 2;
    if (!j->upper_given) j->cfg.fsk_upper = j->cfg.Fs / 2;
}

/* ---- test frames (fsk_demod.c:239-256,319-357): the last 100 bits against the known frame, every bit ----------------------- */

typedef struct { uint64_t rx[2], tx[2]; int frames, bits, errs; } tf_t;

static void tf_shift(uint64_t r[2], int bit) {   /* newest bit at position 0, the oldest of the 100 at position 99 */
    r[1] = ((r[1] << 1) | (r[0] >> 63)) & ((1ull << (TF_BITS - 64)) - 1);
    r[0] = (r[0] << 1) | (uint64_t)(bit & 1);
}

static void tf_init(tf_t *t) {
    memset(t, 0, sizeof *t);
    srand(158324);
    for (int i = 0; i < TF_BITS; i++) tf_shift(t->tx, rand() & 0x1);
}

/* returns 1 when the window lines up with the test frame (fewer than 10 % errors); *errs = mismatches of this position */
static int tf_push(tf_t *t, int bit, int *errs) {
    tf_shift(t->rx, bit);
    *errs = __builtin_popcountll(t->rx[0] ^ t->tx[0]) + __builtin_popcountll(t->rx[1] ^ t->tx[1]);
    if (!(*errs < 0.1 * TF_BITS)) return 0;
    t->frames++; t->bits += TF_BITS; t->errs += *errs;
    return 1;
}

/* ---- stats line (fsk_demod.c:362-411) ---------------------------------------------------------------------------------------- */

static void put_floats(const float *v, int n) {
    for (int i = 0; i < n; i++) { fprintf(stderr, "%f ", v[i]); if (i < n - 1) fprintf(stderr, ","); }
}

static void report(sonde_fsk_t *fsk, const job_t *j, const sonde_fsk_info_t *info, const tf_t *tf) {
    static float eye[8 * 160];
    static float *Sf = NULL;
    sonde_fsk_frame_t last; int64_t samples = 0;
    if (!Sf) Sf = (float *)malloc(sizeof(float) * (size_t)info->Ndft);
    sonde_fsk_stats(fsk, 0, &last, Sf, &samples);
    fprintf(stderr, "{\"samples\": %ld, \"EbNodB\": %5.1f, \"ppm\": %4d,", (long)samples, last.snr_est, (int)last.ppm);
    fprintf(stderr, " \"f1_est\":%.1f, \"f2_est\":%.1f", last.f_est[0], last.f_est[1]);
    if (j->cfg.M == 4) fprintf(stderr, ", \"f3_est\":%.1f, \"f4_est\":%.1f", last.f_est[2], last.f_est[3]);
    if (!j->testframes) {
        int32_t ntr = 0, nes = 0;
        fprintf(stderr, ",\t\"eye_diagram\":[");
        if (sonde_fsk_eye(fsk, 0, eye, &ntr, &nes) > 0)
            for (int i = 0; i < ntr; i++) { fprintf(stderr, "["); put_floats(eye + i * nes, nes); fprintf(stderr, "]"); if (i < ntr - 1) fprintf(stderr, ","); }
        fprintf(stderr, "],\"samp_fft\":[");
        put_floats(Sf, info->Ndft / 2);
        fprintf(stderr, "]");
    } else fprintf(stderr, ", \"frames\":%d, \"bits\":%d, \"errs\":%d", tf->frames, tf->bits, tf->errs);
    fprintf(stderr, "}\n");
}

/* ---- main loop ---------------------------------------------------------------------------------------------------------------- */

static void on_term(int signo) { if (signo == SIGTERM) exit(0); }

int main(int argc, char *argv[]) {
    job_t job;
    parse(argc, argv, &job);
    FILE *fin = strcmp(job.in_path, "-") ? fopen(job.in_path, "r") : stdin;
    FILE *fout = strcmp(job.out_path, "-") ? fopen(job.out_path, "w") : stdout;
    fprintf(stderr, "Setting estimator limits to %d to %d Hz.\n", job.cfg.fsk_lower, job.cfg.fsk_upper);
    const sonde_fsk_cfg_t *c = &job.cfg;
    if (c->Fs < 1 || c->Rs < 1 || c->Fs % c->Rs || c->P < 1 || (c->Fs / c->Rs) % c->P || c->nsym < 1) {
        fprintf(stderr, "fsk_demod: Fs/Rs and (Fs/Rs)/P must be integers\n");          /* the reference asserts (fsk.c:119-129) */
        abort();
    }
    job.cfg.max_chunk = (c->Fs / c->Rs) * (c->nsym + 2);
    sonde_fsk_t *fsk = NULL;
    const int rc = sonde_fsk_create(&job.cfg, &fsk);
    if (!fin || !fout || rc < 0) { fprintf(stderr, "Couldn't open files (%s)\n", rc < 0 ? sonde_strerror(rc) : "io"); exit(1); }
    sonde_fsk_info_t info;
    sonde_fsk_info(fsk, &info);
    size_t nin = (size_t)info.N;                                /* fsk_nin() before the first frame */
    int every = 0, countdown = 0;                               /* stats line every `every` frames */
    if (job.stats) every = (int)(1 / (job.stats_rate * ((float)nin / (float)c->Fs)));
    unsigned char *raw = (unsigned char *)malloc((size_t)job.sample_bytes * (size_t)(info.N + 2 * info.Ts));
    float *soft = (float *)malloc(sizeof(float) * (size_t)info.Nbits);
    uint8_t *hard = (uint8_t *)malloc((size_t)info.Nbits);
    tf_t tf;
    if (job.testframes) tf_init(&tf);
    signal(SIGTERM, on_term);

    while (fread(raw, (size_t)job.sample_bytes, nin, fin) == nin) {
        sonde_fsk_frame_t fr; int32_t nf = 0;
        if (sonde_fsk_process_host(fsk, raw, (int64_t)nin, (int32_t)nin) < 0) break;
        if (sonde_fsk_fetch(fsk, 0, soft, info.Nbits, &fr, 1, &nf) != info.Nbits || nf != 1) break;
        if (sonde_fsk_fetch_bits(fsk, 0, hard, info.Nbits) != info.Nbits) break;
        nin = (size_t)fr.nin_next;
        int aligned = 0;
        if (job.testframes)
            for (int k = 0; k < info.Nbits; k++) {
                int errs;
                if (!tf_push(&tf, job.soft ? soft[k] < 0.0 : hard[k], &errs)) continue;
                aligned = 1;
                if (!job.stats) fprintf(stderr, "errs: %d FSK BER %f, bits tested %d, bit errors %d\n", errs, (float)tf.errs / (float)tf.bits, tf.bits, tf.errs);
            }
        if (job.stats) {
            if (countdown < 0 || aligned) { report(fsk, &job, &info, &tf); if (countdown < 0) countdown = every; }
            if (!job.testframes) countdown--;
        }
        if (job.soft) {
            if (job.soft_negate) for (int k = 0; k < info.Nbits; k++) soft[k] = soft[k] * -1.0f;
            fwrite(soft, sizeof(float), (size_t)info.Nbits, fout);
        } else fwrite(hard, 1, (size_t)info.Nbits, fout);
        if (fout == stdout) fflush(stdout);
    }
    free(raw); free(soft); free(hard);
    fclose(fin); fclose(fout);
    sonde_fsk_destroy(fsk);
    return 0;
}
