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
 * With SONDE_BROKER=<socket path> in the environment the modem is not opened here: the frames go to the resident broker
 * (host/sonde_broker.c, host/broker_proto.h), which batches the frames of all such processes into one launch.  Output is the same.
 *
 * Structure: parse() -> job_t; modem_open() (local engine or broker connection); one modem_frame() per fsk_nin() samples, handed to
 * the optional test-frame matcher (tf_push) and the optional stats reporter (report).
 */
#include <getopt.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <sys/socket.h>
#include <sys/un.h>
#include "sonde_fsk.h"
#include "broker_proto.h"

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

/* ---- the modem behind one of two doors: an engine of our own, or a channel of the resident broker --------------------------------- */

typedef struct {
    sonde_fsk_t *eng;                            /* local */
    int fd;                                      /* broker connection, -1 when local */
    sonde_fsk_info_t info;
} modem_t;

typedef struct {                                 /* what a stats line needs, valid when asked for with the frame */
    int64_t samples;
    float *Sf; float eye[8 * 160]; int32_t neyetr, neyesamp;
} frame_stats_t;

static int io_all(int fd, void *p, size_t n, int out) {
    char *c = (char *)p;
    while (n) {
        const ssize_t k = out ? send(fd, c, n, MSG_NOSIGNAL) : recv(fd, c, n, 0);
        if (k <= 0) return -1;
        c += k; n -= (size_t)k;
    }
    return 0;
}

/* reply header; an ERROR reply is printed and ends the program like a failed sonde_fsk_create() would */
static uint32_t broker_reply(int fd, uint32_t want) {
    brk_hdr_t h;
    if (io_all(fd, &h, sizeof h, 0) || h.magic != BRK_MAGIC) { fprintf(stderr, "fsk_demod: broker connection lost\n"); exit(1); }
    if (h.type == BRK_ERROR) {
        char msg[256]; const size_t n = h.length < sizeof msg ? h.length : sizeof msg - 1;
        if (io_all(fd, msg, n, 0)) msg[0] = 0;
        msg[n] = 0;
        fprintf(stderr, "Couldn't open files (%s)\n", msg); exit(1);
    }
    if (h.type != want) { fprintf(stderr, "fsk_demod: unexpected broker reply %u\n", h.type); exit(1); }
    return h.length;
}

static int modem_open(modem_t *m, const sonde_fsk_cfg_t *cfg) {
    const char *path = getenv("SONDE_BROKER");
    m->eng = NULL; m->fd = -1;
    if (!path || !*path) {
        const int rc = sonde_fsk_create(cfg, &m->eng);
        if (rc < 0) return rc;
        return sonde_fsk_info(m->eng, &m->info);
    }
    struct sockaddr_un addr; memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX; strncpy(addr.sun_path, path, sizeof addr.sun_path - 1);
    m->fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (m->fd < 0 || connect(m->fd, (struct sockaddr *)&addr, sizeof addr) < 0) { fprintf(stderr, "fsk_demod: cannot reach the broker at %s\n", path); return SONDE_E_NOGPU; }
    brk_hdr_t h = { BRK_MAGIC, BRK_HELLO, sizeof(brk_hello_t) };
    brk_hello_t hello; memset(&hello, 0, sizeof hello);
    hello.kind = BRK_KIND_FSK; hello.fsk = *cfg;
    if (io_all(m->fd, &h, sizeof h, 1) || io_all(m->fd, &hello, sizeof hello, 1)) return SONDE_E_NOGPU;
    if (broker_reply(m->fd, BRK_INFO) != sizeof m->info || io_all(m->fd, &m->info, sizeof m->info, 0)) return SONDE_E_NOGPU;
    return 0;
}

/* one modem frame from exactly fsk_nin() samples; 0 on success */
static int modem_frame(modem_t *m, const void *raw, size_t nin, size_t sample_bytes, int want_stats, float *soft, uint8_t *hard,
                       sonde_fsk_frame_t *fr, frame_stats_t *st) {
    const int nbits = m->info.Nbits;
    if (m->fd < 0) {
        int32_t nf = 0;
        if (sonde_fsk_process_host(m->eng, raw, (int64_t)nin, (int32_t)nin) < 0) return -1;
        if (sonde_fsk_fetch(m->eng, 0, soft, nbits, fr, 1, &nf) != nbits || nf != 1) return -1;
        if (sonde_fsk_fetch_bits(m->eng, 0, hard, nbits) != nbits) return -1;
        sonde_fsk_stats(m->eng, 0, NULL, want_stats ? st->Sf : NULL, &st->samples);
        if (want_stats) sonde_fsk_eye(m->eng, 0, st->eye, &st->neyetr, &st->neyesamp);
        return 0;
    }
    brk_hdr_t h = { BRK_MAGIC, BRK_DATA, (uint32_t)(sizeof(brk_data_t) + nin * sample_bytes) };
    brk_data_t d = { (uint32_t)nin, (uint32_t)(want_stats != 0) };
    if (io_all(m->fd, &h, sizeof h, 1) || io_all(m->fd, &d, sizeof d, 1) || io_all(m->fd, (void *)raw, nin * sample_bytes, 1)) return -1;
    const uint32_t len = broker_reply(m->fd, BRK_RESULT);
    brk_result_t r;
    if (len < sizeof r || io_all(m->fd, &r, sizeof r, 0) || (int)r.nbits != nbits) return -1;
    if (io_all(m->fd, soft, sizeof(float) * (size_t)nbits, 0) || io_all(m->fd, hard, (size_t)nbits, 0)) return -1;
    *fr = r.frame; st->samples = r.samples;
    if (r.has_stats) {
        brk_eye_t eye;
        if (io_all(m->fd, st->Sf, sizeof(float) * (size_t)m->info.Ndft, 0) || io_all(m->fd, &eye, sizeof eye, 0)) return -1;
        memcpy(st->eye, eye.eye, sizeof st->eye); st->neyetr = eye.neyetr; st->neyesamp = eye.neyesamp;
    }
    return 0;
}

static void modem_close(modem_t *m) { if (m->eng) sonde_fsk_destroy(m->eng); if (m->fd >= 0) close(m->fd); }

/* ---- stats line (fsk_demod.c:362-411) ---------------------------------------------------------------------------------------- */

static void put_floats(const float *v, int n) {
    for (int i = 0; i < n; i++) { fprintf(stderr, "%f ", v[i]); if (i < n - 1) fprintf(stderr, ","); }
}

static void report(const job_t *j, const sonde_fsk_info_t *info, const sonde_fsk_frame_t *last, const frame_stats_t *st, const tf_t *tf) {
    fprintf(stderr, "{\"samples\": %ld, \"EbNodB\": %5.1f, \"ppm\": %4d,", (long)st->samples, last->snr_est, (int)last->ppm);
    fprintf(stderr, " \"f1_est\":%.1f, \"f2_est\":%.1f", last->f_est[0], last->f_est[1]);
    if (j->cfg.M == 4) fprintf(stderr, ", \"f3_est\":%.1f, \"f4_est\":%.1f", last->f_est[2], last->f_est[3]);
    if (!j->testframes) {
        const int ntr = st->neyetr, nes = st->neyesamp;
        fprintf(stderr, ",\t\"eye_diagram\":[");
        for (int i = 0; i < ntr; i++) { fprintf(stderr, "["); put_floats(st->eye + i * nes, nes); fprintf(stderr, "]"); if (i < ntr - 1) fprintf(stderr, ","); }
        fprintf(stderr, "],\"samp_fft\":[");
        put_floats(st->Sf, info->Ndft / 2);
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
    modem_t modem;
    const int rc = modem_open(&modem, &job.cfg);
    if (!fin || !fout || rc < 0) { fprintf(stderr, "Couldn't open files (%s)\n", rc < 0 ? sonde_strerror(rc) : "io"); exit(1); }
    const sonde_fsk_info_t info = modem.info;
    size_t nin = (size_t)info.N;                                /* fsk_nin() before the first frame */
    int every = 0, countdown = 0;                               /* stats line every `every` frames */
    int failed = 0;                                             /* the engine refused or lost a frame: exit 255, never a silent 0 */
    if (job.stats) every = (int)(1 / (job.stats_rate * ((float)nin / (float)c->Fs)));
    unsigned char *raw = (unsigned char *)malloc((size_t)job.sample_bytes * (size_t)(info.N + 2 * info.Ts));
    float *soft = (float *)malloc(sizeof(float) * (size_t)info.Nbits);
    uint8_t *hard = (uint8_t *)malloc((size_t)info.Nbits);
    frame_stats_t fst; memset(&fst, 0, sizeof fst);
    fst.Sf = (float *)calloc((size_t)info.Ndft, sizeof(float));
    tf_t tf;
    if (job.testframes) tf_init(&tf);
    signal(SIGTERM, on_term);

    while (fread(raw, (size_t)job.sample_bytes, nin, fin) == nin) {
        sonde_fsk_frame_t fr;
        /* the eye and the spectrum are only printed without --testframes, and then exactly when the countdown has run out */
        const int want = job.stats && !job.testframes && countdown < 0;
        if (modem_frame(&modem, raw, nin, (size_t)job.sample_bytes, want, soft, hard, &fr, &fst)) {
            fprintf(stderr, "fsk_demod: the modem engine failed on a frame of %zu samples (unsupported geometry or device error)\n", nin);
            failed = 1; break;
        }
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
            if (countdown < 0 || aligned) { report(&job, &info, &fr, &fst, &tf); if (countdown < 0) countdown = every; }
            if (!job.testframes) countdown--;
        }
        if (job.soft) {
            if (job.soft_negate) for (int k = 0; k < info.Nbits; k++) soft[k] = soft[k] * -1.0f;
            fwrite(soft, sizeof(float), (size_t)info.Nbits, fout);
        } else fwrite(hard, 1, (size_t)info.Nbits, fout);
        if (fout == stdout) fflush(stdout);
    }
    free(raw); free(soft); free(hard); free(fst.Sf);
    fclose(fin); fclose(fout);
    modem_close(&modem);
    return failed ? -1 : 0;
}
