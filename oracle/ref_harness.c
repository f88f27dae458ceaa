/*
 * oracle/ref_harness.c — TEST INFRASTRUCTURE.  Our code, linked against the *reference's own*
 * demod_mod.o (built by oracle/Makefile from /root/reference/demod/mod/demod_mod.c).
 *
 * The reference has no CLI flag that prints per-sample streams or soft bits, so this shim drives the
 * public seam of demod_mod.h:179-192 (init_buffers / f32buf_sample / find_header / read_softbit2p)
 * over an in-memory capture and copies out what the engine is compared against:
 *   - rot_iqbuf  (decimated + IF-filtered IQ,      demod_mod.c:775)
 *   - fm_buffer  (FM discriminator [+FM low-pass], demod_mod.c:849)
 *   - bufs       (sliced stream: tone correlator for opt_iq>=2, FM otherwise, demod_mod.c:852)
 *   - per header hit: mv, mv_pos and the soft bits of read_softbit2p
 * dsp_t is filled the way the reference callers do it (rs41mod.c:2816-2836, dfm09mod.c main).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "demod_mod.h"

typedef struct {
    int    sr;          /* input sample rate                                  */
    int    bps;         /* 8 / 16 / 32                                        */
    int    opt_iq;      /* 0 FM audio (mono), 1..3 IF-rate IQ, 5 baseband IQ  */
    int    opt_lp;      /* LP_IQ | LP_FM                                      */
    int    opt_dc;
    int    opt_iqdc;
    int    opt_min;
    int    opt_nolut;
    double xlt_fq;      /* already negated like rs41mod.c:2685                */
    float  baud;
    int    symlen;
    int    symhd;
    float  BT;
    float  h;
    int    lpIQ_bw;
    int    lpFM_bw;
    const char *hdr;
} ref_cfg_t;

static void fill(dsp_t *d, const ref_cfg_t *c, FILE *fp) {
    memset(d, 0, sizeof(*d));
    d->fp = fp;
    d->sr = c->sr;
    d->bps = c->bps;
    d->nch = c->opt_iq ? 2 : 1;
    d->ch = 0;
    d->br = c->baud;
    d->sps = (float)d->sr / d->br;
    d->symlen = c->symlen;
    d->symhd = c->symhd;
    d->_spb = d->sps * c->symlen;
    d->hdr = (char *)c->hdr;
    d->hdrlen = (int)strlen(c->hdr);
    d->BT = c->BT;
    d->h = c->h;
    d->opt_iq = c->opt_iq;
    d->opt_iqdc = c->opt_iqdc;
    d->opt_lp = c->opt_lp;
    d->lpIQ_bw = c->lpIQ_bw;
    d->lpFM_bw = c->lpFM_bw;
    d->opt_dc = c->opt_dc;
    d->opt_IFmin = c->opt_min;
    d->opt_nolut = c->opt_nolut;
    d->xlt_fq = c->xlt_fq;
}

/* consts[]: N, M, L, K, delay, dectaps, decM, lut_len, lpIQtaps, lpFMtaps, IF sr */
static void consts_of(const dsp_t *d, int *consts) {
    consts[0] = d->DFT.N; consts[1] = d->M; consts[2] = d->L; consts[3] = d->K;
    consts[4] = (int)d->delay; consts[5] = (int)d->dectaps; consts[6] = d->decM;
    consts[7] = (int)d->lut_len; consts[8] = d->lpIQtaps; consts[9] = d->lpFMtaps;
    consts[10] = d->sr;
}

/* Run the sample front-end only.  Returns number of IF samples produced (<= max_if). */
int ref_streams(const ref_cfg_t *cfg, const void *data, size_t nbytes, int max_if,
                float *iq_out /*2*max_if or NULL*/, float *fm_out, float *bufs_out, int *consts) {
    dsp_t d;
    FILE *fp = fmemopen((void *)data, nbytes, "rb");
    int n = 0;
    if (!fp) return -1;
    fill(&d, cfg, fp);
    if (init_buffers(&d) < 0) { fclose(fp); return -2; }
    if (consts) consts_of(&d, consts);
    while (n < max_if && f32buf_sample(&d, 0) != EOF) {
        unsigned s = d.sample_in - 1;
        if (iq_out && d.opt_iq) {
            float complex z = d.rot_iqbuf[s % d.N_IQBUF];
            iq_out[2 * n] = crealf(z); iq_out[2 * n + 1] = cimagf(z);
        }
        if (fm_out)   fm_out[n]   = d.fm_buffer[s % d.M];
        if (bufs_out) bufs_out[n] = d.bufs[s % d.M];
        n++;
    }
    free_buffers(&d);
    fclose(fp);
    return n;
}

/*
 * Header search + soft-bit slicing exactly as the reference framers drive it
 * (rs41mod.c:2873-2968): find_header(thres, hdmax, bitofs, opt_dc); polarity check; nbits x
 * read_softbit2p(inv=0, ofs=bitofs, pos=bitpos, l, spike=0).  option_inv = 0, no --auto.
 * Per hit h: hits[4h..] = {mv, mv_pos, n_bits_read, sample_in_after}; sb[h*nbits + i], sb1[...].
 */
int ref_softframes(const ref_cfg_t *cfg, const void *data, size_t nbytes,
                   float thres, int hdmax, int bitofs, float l, int nbits, int max_hits,
                   double *hits, float *sb, float *sb1, int *consts) {
    dsp_t d;
    FILE *fp = fmemopen((void *)data, nbytes, "rb");
    int nh = 0;
    if (!fp) return -1;
    fill(&d, cfg, fp);
    if (init_buffers(&d) < 0) { fclose(fp); return -2; }
    if (consts) consts_of(&d, consts);
    while (nh < max_hits) {
        int hf = find_header(&d, thres, hdmax, bitofs, d.opt_dc);
        int i, q = 0;
        if (hf == EOF) break;
        if (d.mv * 0.5f < 0) continue;               /* rs41mod.c:2888-2891, inv=0 aut=0 */
        hits[4 * nh] = d.mv; hits[4 * nh + 1] = d.mv_pos;
        for (i = 0; i < nbits; i++) {
            hsbit_t a, b;
            q = read_softbit2p(&d, &a, 0, bitofs, i, l, 0, &b);
            if (q == EOF) break;
            sb[(size_t)nh * nbits + i] = a.sb;
            sb1[(size_t)nh * nbits + i] = b.sb;
        }
        hits[4 * nh + 2] = i; hits[4 * nh + 3] = d.sample_in;
        nh++;
        if (q == EOF) break;
    }
    free_buffers(&d);
    fclose(fp);
    return nh;
}
