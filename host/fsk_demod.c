/*
 * host/fsk_demod.c — `fsk_demod` command-line front end on top of libsonde_hip's 2-FSK modem (C).
 *
 * Reference contract kept (reference utils/fsk_demod.c:53-457; callers auto_rx/autorx/decode.py:901,976,1067,1120):
 *     fsk_demod [--cs16|--cu8] [-s] [-i] [-b lo] [-u hi] [--mask S] [--nsym=N] [-p P] [--stats[=r]] 2 <Fs> <Rs> <in|-> <out|->
 * stdin : complex int16 (--cs16), complex uint8 (--cu8) or real int16;
 * stdout: Nbits float32 soft decisions per modem frame (-s; -i negates) or one byte per bit, flushed per frame (:430-435);
 * stderr: `Setting estimator limits to a to b Hz.` and, with --stats, one JSON line every 1/(r*loop_time) frames carrying
 *         samples / EbNodB / ppm / f1_est / f2_est / samp_fft (:365-411; what auto_rx/autorx/fsk_demod.py:23 requires).
 * exit 0 at EOF / SIGTERM, 1 on usage errors.  Like the reference, each iteration reads exactly fsk_nin() samples.
 * Not implemented: 4-FSK, --testframes.
 */
#include <getopt.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "sonde_fsk.h"

static void sig_handler(int signo) { if (signo == SIGTERM) exit(0); }

int main(int argc, char *argv[]) {
    sonde_fsk_cfg_t cfg;
    int enable_stats = 0, stats_rate = 8, soft_dec_mode = 0, softinv = 0, complex_input = 1, bytes_per_sample = 2;
    int user_lower = 0, user_upper = 0, o = 0, opt_idx = 0;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.P = 10; cfg.nsym = 50; cfg.tone_spacing = 100; cfg.n_channels = 1;
    while (o != -1) {
        static struct option long_opts[] = {
            {"help", no_argument, 0, 'h'}, {"softinv", no_argument, 0, 'i'}, {"conv", required_argument, 0, 'p'},
            {"cs16", no_argument, 0, 'c'}, {"cu8", no_argument, 0, 'd'}, {"fsk_lower", required_argument, 0, 'b'},
            {"fsk_upper", required_argument, 0, 'u'}, {"stats", optional_argument, 0, 't'}, {"soft-dec", no_argument, 0, 's'},
            {"testframes", no_argument, 0, 'f'}, {"nsym", required_argument, 0, 'n'}, {"mask", required_argument, 0, 'm'}, {0, 0, 0, 0}
        };
        o = getopt_long(argc, argv, "fhilp:cdt::sb:u:m", long_opts, &opt_idx);
        switch (o) {
        case 'c': complex_input = 2; bytes_per_sample = 2; break;
        case 'd': complex_input = 2; bytes_per_sample = 1; break;
        case 'f': fprintf(stderr, "fsk_demod (sonde_hip): --testframes is not implemented\n"); return 1;
        case 'i': softinv = 1; break;
        case 't': enable_stats = 1; if (optarg != NULL) { stats_rate = atoi(optarg); if (stats_rate == 0) stats_rate = 8; } break;
        case 's': soft_dec_mode = 1; break;
        case 'p': cfg.P = atoi(optarg); break;
        case 'b': if (optarg != NULL) { cfg.fsk_lower = atoi(optarg); user_lower = 1; } break;
        case 'u': if (optarg != NULL) { cfg.fsk_upper = atoi(optarg); user_upper = 1; } break;
        case 'n': if (optarg != NULL) cfg.nsym = atoi(optarg); break;
        case 'm': cfg.mask = 1; cfg.tone_spacing = optarg ? atoi(optarg) : 100; break;
        case 'h': case '?': goto helpmsg;
        }
    }
    int dx = optind;
    if ((argc - dx) < 5) { fprintf(stderr, "Too few arguments\n"); goto helpmsg; }
    if ((argc - dx) > 5) {
        fprintf(stderr, "Too many arguments\n");
    helpmsg:
        fprintf(stderr, "usage: %s [options] (2|4) SampleRate SymbolRate InputModemRawFile OutputFile\n", argv[0]);
        fprintf(stderr, " -c --cs16  -d --cu8  -t[r] --stats=[r]  -s --soft-dec  -i --softinv  -p P  --fsk_lower f  --fsk_upper f  --nsym N  --mask S\n");
        exit(1);
    }
    cfg.M = atoi(argv[dx]); cfg.Fs = atoi(argv[dx + 1]); cfg.Rs = atoi(argv[dx + 2]);
    if (cfg.M != 2 && cfg.M != 4) { fprintf(stderr, "Mode %d is not valid. Mode must be 2 or 4.\n", cfg.M); goto helpmsg; }
    if (cfg.M == 4) { fprintf(stderr, "fsk_demod (sonde_hip): 4-FSK is not implemented\n"); return 1; }
    FILE *fin = strcmp(argv[dx + 3], "-") == 0 ? stdin : fopen(argv[dx + 3], "r");
    FILE *fout = strcmp(argv[dx + 4], "-") == 0 ? stdout : fopen(argv[dx + 4], "w");
    if (!user_lower) cfg.fsk_lower = (complex_input == 1) ? 0 : -cfg.Fs / 2;
    if (!user_upper) cfg.fsk_upper = cfg.Fs / 2;
    fprintf(stderr, "Setting estimator limits to %d to %d Hz.\n", cfg.fsk_lower, cfg.fsk_upper);
    cfg.format = complex_input == 1 ? SONDE_FSK_S16 : (bytes_per_sample == 1 ? SONDE_FSK_CU8 : SONDE_FSK_CS16);
    if (cfg.Fs < 1 || cfg.Rs < 1 || cfg.Fs % cfg.Rs || cfg.P < 1 || (cfg.Fs / cfg.Rs) % cfg.P || cfg.nsym < 1) {
        fprintf(stderr, "fsk_demod: Fs/Rs and (Fs/Rs)/P must be integers\n");          /* the reference asserts (fsk.c:127-129) */
        abort();
    }
    cfg.max_chunk = (cfg.Fs / cfg.Rs) * (cfg.nsym + 2);
    sonde_fsk_t *fsk = NULL;
    int rc = sonde_fsk_create(&cfg, &fsk);
    if (fin == NULL || fout == NULL || rc < 0) { fprintf(stderr, "Couldn't open files (%s)\n", rc < 0 ? sonde_strerror(rc) : "io"); exit(1); }
    sonde_fsk_info_t info;
    sonde_fsk_info(fsk, &info);
    sonde_fsk_frame_t last;
    sonde_fsk_stats(fsk, 0, &last, NULL, NULL);
    int stats_loop = 0, stats_ctr = 0;
    if (enable_stats) { const float loop_time = ((float)last.nin_next) / ((float)cfg.Fs); stats_loop = (int)(1 / (stats_rate * loop_time)); stats_ctr = 0; }
    const size_t unit = (size_t)bytes_per_sample * complex_input;
    unsigned char *rawbuf = (unsigned char *)malloc(unit * (size_t)(info.N + info.Ts * 2));
    float *sdbuf = (float *)malloc(sizeof(float) * info.Nbits);
    float *Sf = (float *)malloc(sizeof(float) * info.Ndft);
    uint8_t *bitbuf = (uint8_t *)malloc(info.Nbits);
    signal(SIGTERM, sig_handler);

    for (;;) {
        const size_t nin = (size_t)last.nin_next;
        if (fread(rawbuf, unit, nin, fin) != nin) break;
        if (sonde_fsk_process_host(fsk, rawbuf, (int64_t)nin, (int32_t)nin) < 0) break;
        int32_t nf = 0;
        sonde_fsk_frame_t fr;
        if (sonde_fsk_fetch(fsk, 0, sdbuf, info.Nbits, &fr, 1, &nf) != info.Nbits) break;
        int64_t samples = 0;
        sonde_fsk_stats(fsk, 0, &last, enable_stats && stats_ctr < 0 ? Sf : NULL, &samples);
        if (enable_stats) {
            if (stats_ctr < 0) {
                fprintf(stderr, "{");
                fprintf(stderr, "\"samples\": %ld, \"EbNodB\": %5.1f, \"ppm\": %4d,", (long)samples, last.snr_est, (int)last.ppm);
                fprintf(stderr, " \"f1_est\":%.1f, \"f2_est\":%.1f", last.f_est[0], last.f_est[1]);
                {   /* eye diagram (fsk_demod.c:387-398) */
                    static float eye[8 * 160];
                    int32_t ntr = 0, nes = 0;
                    fprintf(stderr, ",\t\"eye_diagram\":[");
                    if (sonde_fsk_eye(fsk, 0, eye, &ntr, &nes) > 0)
                        for (int i = 0; i < ntr; i++) {
                            fprintf(stderr, "[");
                            for (int j = 0; j < nes; j++) { fprintf(stderr, "%f ", eye[i * nes + j]); if (j < nes - 1) fprintf(stderr, ","); }
                            fprintf(stderr, "]");
                            if (i < ntr - 1) fprintf(stderr, ",");
                        }
                    fprintf(stderr, "],");
                }
                fprintf(stderr, "\"samp_fft\":[");
                for (int i = 0; i < info.Ndft / 2; i++) { fprintf(stderr, "%f ", Sf[i]); if (i < info.Ndft / 2 - 1) fprintf(stderr, ","); }
                fprintf(stderr, "]}\n");
                stats_ctr = stats_loop;
            }
            stats_ctr--;
        }
        if (soft_dec_mode) {
            if (softinv) for (int j = 0; j < info.Nbits; j++) sdbuf[j] = sdbuf[j] * -1.0f;
            fwrite(sdbuf, sizeof(float), info.Nbits, fout);
        } else {
            for (int j = 0; j < info.Nbits; j++) bitbuf[j] = sdbuf[j] < 0.0f;      /* sym == 1: the upper tone is the larger */
            fwrite(bitbuf, 1, info.Nbits, fout);
        }
        if (fout == stdout) fflush(stdout);
    }
    free(rawbuf); free(sdbuf); free(Sf); free(bitbuf);
    fclose(fin); fclose(fout);
    sonde_fsk_destroy(fsk);
    return 0;
}
