/*
 * oracle/ref_fsk_harness.c — TEST INFRASTRUCTURE.  Our code, linked against the *reference's own* utils/fsk.c,
 * modem_stats.c, kiss_fft.c (compiled where they lie by oracle/Makefile).  Drives the public seam of fsk.h:115-205
 * the way utils/fsk_demod.c:228-311 does (fsk_create_hbr, limits, estimator, `fread fsk_nin` loop, cs16 / 1000) over an
 * in-memory capture and records what the CLI never prints: per modem frame nin, tone estimates, timing, ppm, Eb/N0 —
 * next to the soft decisions.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "fsk.h"
#include "codec2_fdmdv.h"

typedef struct { int Fs, Rs, P, nsym, format, lower, upper, mask, tone_spacing; } ref_fsk_cfg_t;
typedef struct { int nin, nin_next; float f_est[2]; float norm_rx_timing, ppm, EbNodB, snr_est; } ref_fsk_frame_t;

/* format: 1 real s16, 2 cs16, 3 cu8.  Returns frames; sd: [frames*nsym]; Sf_out: Ndft floats after the last frame */
int ref_fsk_run(const ref_fsk_cfg_t *c, const void *raw, size_t nsamples, int max_frames, float *sd, ref_fsk_frame_t *fr,
                float *Sf_out, int *consts /* Ts,N,Ndft,Nmem */)
{
    struct FSK *fsk = fsk_create_hbr(c->Fs, c->Rs, 2, c->P, c->nsym, 1000, c->mask ? c->tone_spacing : 100);
    struct MODEM_STATS stats;
    size_t pos = 0;
    int nf = 0, i;
    COMP *modbuf;
    if (!fsk) return -1;
    fsk_set_freq_est_limits(fsk, c->lower, c->upper);
    fsk_set_freq_est_alg(fsk, c->mask);
    consts[0] = fsk->Ts; consts[1] = fsk->N; consts[2] = fsk->Ndft; consts[3] = fsk->Nmem;
    modbuf = (COMP *)malloc(sizeof(COMP) * (fsk->N + fsk->Ts * 2));
    while (nf < max_frames && pos + fsk_nin(fsk) <= nsamples) {
        const int nin = (int)fsk_nin(fsk);
        for (i = 0; i < nin; i++) {
            if (c->format == 2) {
                const int16_t *p = (const int16_t *)raw + 2 * (pos + i);
                modbuf[i].real = ((float)p[0]) / FDMDV_SCALE; modbuf[i].imag = ((float)p[1] / FDMDV_SCALE);
            } else if (c->format == 1) {
                modbuf[i].real = ((float)((const int16_t *)raw)[pos + i]) / FDMDV_SCALE; modbuf[i].imag = 0.0;
            } else {
                const uint8_t *p = (const uint8_t *)raw + 2 * (pos + i);
                modbuf[i].real = ((float)p[0] - 127.0) / 128.0; modbuf[i].imag = ((float)p[1] - 127.0) / 128.0;
            }
        }
        pos += nin;
        fsk_demod_sd(fsk, sd + (size_t)nf * fsk->Nbits, modbuf);
        fsk_get_demod_stats(fsk, &stats);
        fr[nf].nin = nin; fr[nf].nin_next = (int)fsk_nin(fsk);
        fr[nf].f_est[0] = c->mask ? fsk->f2_est[0] : fsk->f_est[0];
        fr[nf].f_est[1] = c->mask ? fsk->f2_est[1] : fsk->f_est[1];
        fr[nf].norm_rx_timing = fsk->norm_rx_timing; fr[nf].ppm = fsk->ppm; fr[nf].EbNodB = fsk->EbNodB; fr[nf].snr_est = stats.snr_est;
        nf++;
    }
    memcpy(Sf_out, fsk->Sf, sizeof(float) * fsk->Ndft);
    free(modbuf);
    fsk_destroy(fsk);
    return nf;
}
