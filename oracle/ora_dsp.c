/*
 * oracle/ora_dsp.c — TEST INFRASTRUCTURE: CPU restatement of the reference sample front-end,
 * header correlator and bit slicer (reference: demod/mod/demod_mod.c).  Written from the reference's
 * behaviour, not from its text; every routine cites the lines it follows.  Arithmetic types
 * (float vs double, int truncations) are kept exactly where the reference has them so that the
 * only difference to the compiled reference is -Ofast's freedom to re-associate.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ora_dsp.h"

#define TWO_PI 6.2831853071795864769252867665590
#define FM_GAIN 0.8

static inline ora_cf cf(float re, float im) { ora_cf z = { re, im }; return z; }
static inline ora_cf cmulf(ora_cf a, ora_cf b) {
    return cf(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
/* float-complex times double-complex, rounded once to float (C: float complex * double complex) */
static inline ora_cf cmul_fd(ora_cf a, double br, double bi) {
    double ar = a.re, ai = a.im;
    return cf((float)(ar * br - ai * bi), (float)(ar * bi + ai * br));
}

/* ---- filter design: Blackman-windowed sinc, 1-norm (demod_mod.c:555-587) ---- */
int ora_lowpass_design(float f, int taps, float **out) {
    if (taps % 2 == 0) taps++;
    if (taps < 1) taps = 1;
    float *ws = (float *)calloc((size_t)2 * taps + 1, sizeof(float));
    if (!ws) return -1;
    double norm = 0.0;
    const int c = (taps - 1) / 2;
    for (int n = 0; n < taps; n++) {
        double win = 7938 / 18608.0 - 9240 / 18608.0 * cos(TWO_PI * n / (taps - 1))
                   + 1430 / 18608.0 * cos(4 * M_PI * n / (taps - 1));
        float f2 = 2 * f;                       /* float product, as in the reference */
        double x = (double)(f2 * (float)(n - c)); /* float product feeding sinc()        */
        double sinc = (x == 0) ? 1.0 : sin(M_PI * x) / (M_PI * x);
        double h = f2 * sinc;
        ws[n] = (float)(win * h);
        norm += ws[n];
    }
    for (int n = 0; n < taps; n++) ws[n] = (float)(ws[n] / norm);
    for (int n = 0; n < taps; n++) ws[taps + n] = ws[n];
    *out = ws;
    return taps;
}

/* ---- mixer table (demod_mod.c:1262-1296): frequency snapped to a multiple of d Hz, float phase ---- */
int ora_lut_design(double xlt_fq, int sr_base, ora_cf **out) {
    const int W = 16;
    int d;
    int freq = (int)(xlt_fq * (double)sr_base + 0.5);
    int freq0 = freq;
    for (d = W; d > 0; d--) if (sr_base % d == 0) break;
    if (d == 0) d = 1;
    for (int k = 0; k < W / 2; k++) {
        if ((freq + k) % d == 0) { freq0 = freq + k; break; }
        if ((freq - k) % d == 0) { freq0 = freq - k; break; }
    }
    int len = sr_base / d;
    double f0 = freq0 / (double)sr_base;
    ora_cf *ex = (ora_cf *)calloc((size_t)len + 1, sizeof(ora_cf));
    if (!ex) return -1;
    for (int n = 0; n < len; n++) {
        float t = (float)(f0 * (double)n);
        double ph = t * TWO_PI;
        ex[n] = cf((float)cos(ph), (float)sin(ph));
    }
    *out = ex;
    return len;
}

/* ---- header template: Gaussian-shaped FM pulses incl. neighbour ISI (demod_mod.c:1190-1195,1398-1421) ---- */
static double gq(double x) { return 0.5 - 0.5 * erf(x / 1.4142135624); }
static double gpulse(double t, double sigma) { return gq((t - 0.5) / sigma) - gq((t + 0.5) / sigma); }

int ora_match_design(const char *hdr, int hdrlen, float sps, float bt, float **out) {
    int L = (int)(hdrlen * sps + 0.5);
    float *m = (float *)calloc((size_t)L + 1, sizeof(float));
    if (!m) return -1;
    double sigma = sqrt(log(2)) / (TWO_PI * bt);
    for (int i = 0; i < L; i++) {
        int pos = (int)(i / sps);
        float t = (float)((i - pos * sps) / sps - 0.5);
        float b1 = (float)(((hdr[pos] & 1) - 0.5) * 2.0);
        float b = (float)(b1 * gpulse(t, sigma));
        if (pos > 0) {
            float b0 = (float)(((hdr[pos - 1] & 1) - 0.5) * 2.0);
            b = (float)(b + b0 * gpulse((double)(t + 1), sigma));
        }
        if (pos < hdrlen - 1) {
            float b2 = (float)(((hdr[pos + 1] & 1) - 0.5) * 2.0);
            b = (float)(b + b2 * gpulse((double)(t - 1), sigma));
        }
        m[i] = b;
    }
    double n2 = 0.0;
    for (int i = 0; i < L; i++) { double x = m[i]; n2 += x * x; }
    float nm = (float)sqrt(n2);
    for (int i = 0; i < L; i++) m[i] /= nm;
    *out = m;
    return L;
}

/* ---- radix-2 DIT with per-stage twiddle recurrence (demod_mod.c:29-83) ---- */
static void fft_inplace(const ora_dsp *d, ora_cf *Z) {
    const int N = d->N;
    int j = 1;
    for (int i = 1; i < N; i++) {
        if (i < j) { ora_cf T = Z[j - 1]; Z[j - 1] = Z[i - 1]; Z[i - 1] = T; }
        int k = N / 2;
        while (k < j) { j -= k; k /= 2; }
        j += k;
    }
    for (int s = 0; s < d->log2n; s++) {
        int half = 1 << s, span = half << 1;
        ora_cf w = cf(1.0f, 0.0f), step = d->ew[s];
        for (int jj = 1; jj <= half; jj++) {
            for (int i = jj; i <= N; i += span) {
                int k = i + half;
                ora_cf T = cmulf(Z[k - 1], w);
                Z[k - 1] = cf(Z[i - 1].re - T.re, Z[i - 1].im - T.im);
                Z[i - 1] = cf(Z[i - 1].re + T.re, Z[i - 1].im + T.im);
            }
            w = cmulf(w, step);
        }
    }
}
static void fft_real(const ora_dsp *d, const float *x, ora_cf *Z) {
    for (int i = 0; i < d->N; i++) Z[i] = cf(x[i], 0.0f);
    fft_inplace(d, Z);
}
/* N * inverse DFT via conjugation; result is real for our inputs (demod_mod.c:77-83) */
static void ifft_scaled(const ora_dsp *d, const ora_cf *Z, ora_cf *z) {
    for (int i = 0; i < d->N; i++) z[i] = cf(Z[i].re, -Z[i].im);
    fft_inplace(d, z);
}

/* ---- init (demod_mod.c:1208-1474) ---- */
int ora_init(ora_dsp *d) {
    d->sr = d->sr_in;
    d->sps = (float)d->sr / d->baud;
    d->decM = 1; d->dectaps = 0; d->lut_len = 0;

    if (d->iq_mode == 5) {
        int IF_sr = d->if_min ? 32000 : 48000;
        int sr_base = d->sr_in, decM = 1;
        if (IF_sr > sr_base) IF_sr = sr_base;
        if (IF_sr < sr_base) {
            while (sr_base % IF_sr) IF_sr += 1;
            decM = sr_base / IF_sr;
        }
        float f_lp = (float)((IF_sr + 20e3) / (4.0 * sr_base));
        float t_bw = (float)(IF_sr - 20e3);
        if (d->if_min) t_bw = (float)(IF_sr - 12e3);
        if (t_bw < 0) t_bw = 10e3f;
        t_bw /= sr_base;
        int taps = (int)(4.0 / t_bw);
        if (taps % 2 == 0) taps++;
        taps = ora_lowpass_design(f_lp, taps, &d->w_dec);
        if (taps < 0) return -1;
        d->dectaps = taps;
        d->sr = IF_sr;
        d->sps /= (float)decM;
        d->decM = decM;
        d->lut_len = ora_lut_design(d->xlt_fq, sr_base, &d->lut);
        if (d->lut_len < 0) return -1;
        d->dec_ring = (ora_cf *)calloc((size_t)d->dectaps + 1, sizeof(ora_cf));
        d->dec_pos = 0; d->lut_pos = 0;
    }

    if (d->iq_mode && (d->lp_mask & ORA_LP_IQ)) {
        float f_lp = (float)(24e3 / (float)d->sr / 2.0);
        if (d->lpiq_bw) f_lp = (float)(d->lpiq_bw / (float)d->sr / 2.0);
        int taps = (int)(4 * d->sr / 4e3);
        if (taps % 2 == 0) taps++;
        taps = ora_lowpass_design((float)(1.5 * f_lp), taps, &d->w_iq_acq);
        taps = ora_lowpass_design(f_lp, taps, &d->w_iq_lock);
        if (taps < 0) return -1;
        d->lpiq_taps = taps;
        d->iq_ring = (ora_cf *)calloc((size_t)taps + 3, sizeof(ora_cf));
        d->w_iq = d->w_iq_lock;
        if (d->afc) { d->locked = 0; d->w_iq = d->w_iq_acq; }
    }

    if (d->lp_mask & ORA_LP_FM) {
        float f_lp = (float)(10e3 / (float)d->sr);
        if (d->lpfm_bw > 0) f_lp = d->lpfm_bw / (float)d->sr;
        int taps = (int)(4 * d->sr / 2e3);
        if (taps % 2 == 0) taps++;
        taps = ora_lowpass_design(f_lp, taps, &d->w_fm);
        if (taps < 0) return -1;
        d->lpfm_taps = taps;
        d->fm_ring = (float *)calloc((size_t)taps + 3, sizeof(float));
    }

    d->dc_sx = d->dc_sy = 0; d->dc_ax = d->dc_ay = 0; d->dc_cnt = 0;
    d->dc_lim = (uint32_t)d->sr;
    d->dc_max = d->dc_lim / 32;
    if (d->decM > 1) { d->dc_lim *= d->decM; d->dc_max *= d->decM; }

    int L = (int)(d->hdrlen * d->sps + 0.5);
    int M = 3 * L, p2 = 1;
    d->delay = L / 16;
    d->s_in = 0;
    while (p2 < M) p2 <<= 1;
    while (p2 < 0x2000) p2 <<= 1;
    M = p2;
    d->N = p2;
    d->log2n = (int)(log(d->N) / log(2) + 0.1);
    d->K = M - L - d->delay;
    d->L = L; d->M = M;

    d->bufs = (float *)calloc((size_t)M + 1, sizeof(float));
    d->fmb  = (float *)calloc((size_t)M + 1, sizeof(float));
    d->rawbits = (char *)calloc((size_t)2 * d->hdrlen + 1, 1);
    if (ora_match_design(d->hdr, d->hdrlen, d->sps, d->bt, &d->match) != L) return -1;

    d->xn = (float *)calloc((size_t)d->N + 1, sizeof(float));
    d->Fm = (ora_cf *)calloc((size_t)d->N + 1, sizeof(ora_cf));
    d->X  = (ora_cf *)calloc((size_t)d->N + 1, sizeof(ora_cf));
    d->Z  = (ora_cf *)calloc((size_t)d->N + 1, sizeof(ora_cf));
    d->cx = (ora_cf *)calloc((size_t)d->N + 1, sizeof(ora_cf));
    d->ew = (ora_cf *)calloc((size_t)d->log2n + 1, sizeof(ora_cf));
    for (int n = 0; n < d->log2n; n++) {
        int k = 1 << n;
        double a = M_PI / (float)k;
        d->ew[n] = cf((float)cos(a), (float)-sin(a));
    }
    float *m = (float *)calloc((size_t)d->N + 1, sizeof(float));
    for (int i = 0; i < L; i++) m[L - 1 - i] = d->match[i];
    fft_real(d, m, d->Fm);
    free(m);

    if (d->iq_mode) {
        d->ziq = (ora_cf *)calloc((size_t)d->N + 1, sizeof(ora_cf));
        float nh = -d->h;
        float hs = nh * d->sr;
        double f1 = hs / (2.0 * d->sps);
        d->w1 = TWO_PI * f1;
        d->w2 = TWO_PI * (-f1);
    }
    d->F1 = d->F2 = cf(0, 0);
    d->mv = 0; d->mv_pos = 0; d->Df = 0; d->dDf = 0; d->dc = 0; d->buffered = 0; d->sc = 0;
    return d->K;
}

void ora_free(ora_dsp *d) {
    free(d->w_dec); free(d->dec_ring); free(d->lut);
    free(d->w_iq_acq); free(d->w_iq_lock); free(d->iq_ring);
    free(d->w_fm); free(d->fm_ring);
    free(d->ziq); free(d->bufs); free(d->fmb); free(d->match);
    free(d->Fm); free(d->X); free(d->Z); free(d->cx); free(d->ew); free(d->xn); free(d->rawbits);
}

/* ---- circular FIRs (demod_mod.c:639-648, 711-719): oldest sample pairs with tap 0 ---- */
static ora_cf fir_c(const ora_cf *ring, uint32_t oldest, int taps, const float *ws) {
    float re = 0, im = 0;
    int S = taps - (int)(oldest % (uint32_t)taps);
    for (int n = 0; n < taps; n++) { re += ring[n].re * ws[S + n]; im += ring[n].im * ws[S + n]; }
    return cf(re, im);
}
static float fir_r(const float *ring, uint32_t oldest, int taps, const float *ws) {
    float a = 0;
    int S = taps - (int)(oldest % (uint32_t)taps);
    for (int n = 0; n < taps; n++) a += ring[n] * ws[S + n];
    return a;
}

/* ---- input conversion + running IQ-DC (demod_mod.c:419-508) ---- */
static int read_xy(ora_dsp *d, float *x, float *y) {
    size_t need = (size_t)2 * d->bps / 8;
    if (d->src_pos + need > d->src_len) return -1;
    const uint8_t *p = d->src + d->src_pos;
    d->src_pos += need;
    if (d->bps == 32) { float f[2]; memcpy(f, p, 8); *x = f[0]; *y = f[1]; }
    else if (d->bps == 16) { int16_t b[2]; memcpy(b, p, 4); *x = (float)(b[0] / 32768.0); *y = (float)(b[1] / 32768.0); }
    else { *x = (float)((p[0] - 128) / 128.0); *y = (float)((p[1] - 128) / 128.0); }
    return 0;
}
static void dc_track(ora_dsp *d, float x, float y) {
    d->dc_sx += x; d->dc_sy += y; d->dc_cnt += 1;
    if (d->dc_cnt == d->dc_max) {
        d->dc_ax = (float)(d->dc_sx / (float)d->dc_max);
        d->dc_ay = (float)(d->dc_sy / (float)d->dc_max);
        d->dc_sx = d->dc_sy = 0; d->dc_cnt = 0;
        if (d->dc_max < d->dc_lim) d->dc_max *= 2;
    }
}
static int read_real(ora_dsp *d, float *s) {   /* demod_mod.c:379-405, mono */
    size_t need = (size_t)d->bps / 8;
    if (d->src_pos + need > d->src_len) return -1;
    const uint8_t *p = d->src + d->src_pos;
    d->src_pos += need;
    if (d->bps == 32) { memcpy(s, p, 4); }
    else if (d->bps == 16) { int16_t b; memcpy(&b, p, 2); float v = (float)(b / 128.0); *s = (float)(v / 256.0); }
    else { int16_t b = (int16_t)(p[0] - 128); *s = (float)(b / 128.0); }
    return 0;
}

/* ---- one IF-rate sample (demod_mod.c:722-868) ---- */
int ora_sample(ora_dsp *d, int inv) {
    float s = 0.0f, s_fm = 0.0f;
    double t = d->s_in / (double)d->sr;
    const uint32_t Nm = (uint32_t)d->N, Mm = (uint32_t)d->M;

    if (d->iq_mode) {
        ora_cf z = cf(0, 0);
        if (d->iq_mode == 5) {
            if (d->src_pos + (size_t)d->decM * 2 * d->bps / 8 > d->src_len) return -1; /* short block = EOF */
            for (int j = 0; j < d->decM; j++) {
                float x, y;
                read_xy(d, &x, &y);
                ora_cf u = cf(x - d->dc_ax, y - d->dc_ay);
                dc_track(d, x, y);
                z = cmulf(u, d->lut[d->lut_pos]);
                if (++d->lut_pos >= (uint32_t)d->lut_len) d->lut_pos = 0;
                d->dec_ring[d->dec_pos] = z;
                if (++d->dec_pos >= (uint32_t)d->dectaps) d->dec_pos = 0;
            }
            if (d->decM > 1) z = fir_c(d->dec_ring, d->dec_pos, d->dectaps, d->w_dec);
        } else {
            float x, y;
            if (read_xy(d, &x, &y) < 0) return -1;
            z = cf(x, y);
            if (d->iqdc) { z.re -= d->dc_ax; z.im -= d->dc_ay; dc_track(d, x, y); }
        }
        if (d->afc) {                                  /* :758-761 */
            double a = -t * TWO_PI * d->Df;
            z = cmul_fd(z, cos(a), sin(a));
        }
        if (d->lp_mask & ORA_LP_IQ) {                  /* :765-768 */
            d->iq_ring[d->s_in % (uint32_t)d->lpiq_taps] = z;
            z = fir_c(d->iq_ring, d->s_in + 1, d->lpiq_taps, d->w_iq);
        }
        ora_cf z0 = d->ziq[(d->s_in - 1 + Nm) % Nm];  /* :771-775 */
        ora_cf w = cmulf(z, cf(z0.re, -z0.im));
        s_fm = (float)(FM_GAIN * atan2((double)w.im, (double)w.re) / M_PI);
        d->ziq[d->s_in % Nm] = z;

        if (d->iq_mode >= 2) {                         /* :778-808 two-tone sliding correlator */
            int n = (int)d->sps;
            double tn = (uint32_t)(d->s_in - (uint32_t)n) / (double)d->sr;
            ora_cf zo = d->ziq[(d->s_in - (uint32_t)n + Nm) % Nm];
            ora_cf Xo, Xn;
            Xo = cmul_fd(zo, cos(-tn * d->w1), sin(-tn * d->w1));
            Xn = cmul_fd(z,  cos(-t  * d->w1), sin(-t  * d->w1));
            d->F1.re += Xn.re - Xo.re; d->F1.im += Xn.im - Xo.im;
            Xo = cmul_fd(zo, cos(-tn * d->w2), sin(-tn * d->w2));
            Xn = cmul_fd(z,  cos(-t  * d->w2), sin(-t  * d->w2));
            d->F2.re += Xn.re - Xo.re; d->F2.im += Xn.im - Xo.im;
            double xbit = hypot((double)d->F2.re, (double)d->F2.im) - hypot((double)d->F1.re, (double)d->F1.im);
            s = (float)(xbit / d->sps);
        } else {
            s = s_fm;
        }
    } else {
        if (read_real(d, &s) < 0) return -1;
        s_fm = s;
    }

    if (d->lp_mask & ORA_LP_FM) {                      /* :843-847 */
        d->fm_ring[d->s_in % (uint32_t)d->lpfm_taps] = s_fm;
        s_fm = fir_r(d->fm_ring, d->s_in + 1, d->lpfm_taps, d->w_fm);
        if (d->iq_mode < 2) s = s_fm;
    }
    d->fmb[d->s_in % Mm] = s_fm;
    if (inv) s = -s;
    d->bufs[d->s_in % Mm] = s;

    d->s_out = d->s_in - (uint32_t)d->delay;
    d->s_in += 1;
    return 0;
}

/* ---- windowed matched-filter correlation (demod_mod.c:148-301) ---- */
static int peak_of(const ora_dsp *d, float *mx_out) {
    float mx = 0, mx2 = 0; int mp = -1;
    for (int i = d->L - 1; i < d->K + d->L; i++) {
        float re = d->cx[i].re;
        if (re * re > mx2) { mx = re; mx2 = mx * mx; mp = i; }
    }
    *mx_out = mx;
    return mp;
}
static float norm_at(const ora_dsp *d, int mp) {
    float a = 0;
    /* mp = -1 (no correlation value above zero: a window of digital silence): the reference reads xn[-1 - i] here — in front of its array — and divides 0 by it;
       whatever it finds, the score stays below every threshold, and the wrapped mv_pos = pos - (K + L - 1) - 1 it goes on to store is what matters (find_header's
       `mv_pos > mvpos0` fails for the next window's header).  Restated with a norm of 1 instead of the out-of-bounds read. */
    if (mp < 0) return 1.0f;
    for (int i = 0; i < d->L; i++) a += d->xn[mp - i] * d->xn[mp - i];
    return (float)sqrt(a);
}
static void window_load(ora_dsp *d, const float *ring, uint32_t pos) {
    int W = d->K + d->L, i;
    for (i = 0; i < W; i++) d->xn[i] = ring[(pos + (uint32_t)d->M - (uint32_t)(W - 1) + (uint32_t)i) % (uint32_t)d->M];
    for (; i < d->N; i++) d->xn[i] = 0.0f;
    fft_real(d, d->xn, d->X);
}
static void remove_mean(ora_dsp *d) {
    d->X[0] = cf(0, 0);
    ifft_scaled(d, d->X, d->cx);
    for (int i = 0; i < d->N; i++) d->xn[i] = d->cx[i].re / (float)d->N;
}
static void correlate(ora_dsp *d) {
    for (int i = 0; i < d->N; i++) d->Z[i] = cmulf(d->X[i], d->Fm[i]);
    ifft_scaled(d, d->Z, d->cx);
}

static int corr_window(ora_dsp *d, float thres) {
    int mp; float mx; uint32_t mpos, pos = d->s_out;
    d->mv = 0.0f; d->dc = 0.0;
    if (d->K + d->L > d->N) return -1;
    if (d->s_out < (uint32_t)d->L) return -2;

    window_load(d, d->bufs, pos);
    if (d->afc) remove_mean(d);
    correlate(d);
    mp = peak_of(d, &mx);
    if (mp == d->L - 1 || mp == d->K + d->L - 1) return -4;
    mpos = pos - (uint32_t)(d->K + d->L - 1) + (uint32_t)mp;
    mx /= norm_at(d, mp) * d->N;
    d->mv = mx; d->mv_pos = mpos;
    d->buffered = (int)(d->s_out - d->mv_pos);

    d->mv2 = 0.0f; d->mv2_pos = 0;
    if (d->afc) {
        if (d->iq_mode >= 2 && fabs(mx) < thres) {     /* fallback on the FM stream (:229-277) */
            window_load(d, d->fmb, pos);
            remove_mean(d);
            correlate(d);
            mp = peak_of(d, &mx);
            if (mp == d->L - 1 || mp == d->K + d->L - 1) return -4;
            mpos = pos - (uint32_t)(d->K + d->L - 1) + (uint32_t)mp;
            mx /= norm_at(d, mp) * d->N;
            d->mv2 = mx;
            d->mv2_pos = (uint32_t)(mpos - (d->lpfm_taps - (d->sps - 1)) / 2);   /* float arithmetic, :268 */
            if (d->mv2 > thres || d->mv2 < -thres) {
                d->mv = d->mv2; d->mv_pos = d->mv2_pos;
                d->buffered = (int)(d->s_out - d->mv2_pos);
            }
        }
        double dc = 0.0;
        int ofs = 0;
        if (d->iq_mode >= 2 && d->mv2_pos == 0) ofs = (int)((d->lpfm_taps - (d->sps - 1)) / 2);
        for (int i = 0; i < d->L; i++) dc += d->fmb[((uint32_t)ofs + mpos - (uint32_t)i + (uint32_t)d->M) % (uint32_t)d->M];
        dc /= (float)d->L;
        d->dc = dc;
    }
    d->dDf = d->sr * d->dc / (2.0 * FM_GAIN);
    return mp;
}

/* ---- header bit check (demod_mod.c:870-938) ---- */
static void hdr_bit(const ora_dsp *d, int symlen, char *out, uint32_t mvp, int pos) {
    double edge = pos * symlen * d->sps;
    uint32_t cnt = (uint32_t)ceil(edge);
    double sum = 0.0, dc = 0.0;
    if (d->afc && d->iq_mode < 2) dc = d->dc;
    edge += d->sps;
    do { sum += d->bufs[(cnt + mvp + (uint32_t)d->M) % (uint32_t)d->M] - dc; cnt++; } while (cnt < edge);
    if (symlen == 2) {
        edge += d->sps;
        do { sum -= d->bufs[(cnt + mvp + (uint32_t)d->M) % (uint32_t)d->M] - dc; cnt++; } while (cnt < edge);
        if (sum >= 0) { out[0] = '1'; out[1] = '0'; } else { out[0] = '0'; out[1] = '1'; }
    } else {
        out[0] = (sum >= 0) ? '1' : '0';
    }
}
static int hdr_errors(ora_dsp *d) {
    int step = (d->symhd != 1) ? 2 : 1;
    int len = d->hdrlen / d->symhd, errs = 0;
    char sign = d->mv < 0 ? 1 : 0;
    for (int p = 0; p < len; p++) hdr_bit(d, d->symhd, d->rawbits + p * step, d->mv_pos + 1 - (uint32_t)d->L, p);
    for (int p = len * step; p > 0; p--) if ((d->rawbits[p - 1] ^ sign) != d->hdr[p - 1]) errs++;
    return errs;
}

/* ---- header search (demod_mod.c:1533-1617) ---- */
int ora_find_header(ora_dsp *d, float thres, int hdmax) {
    uint32_t k = 0, prev = 0;
    while (ora_sample(d, 0) == 0) {
        k++;
        if (k < (uint32_t)(d->K - 4)) { d->mv = 0.0f; continue; }
        prev = d->mv_pos;
        corr_window(d, thres);
        k = 0;
        if (!(d->mv > thres || d->mv < -thres)) continue;

        if (d->afc && d->iq_mode) {
            if (fabs(d->dDf) > 100.0) {
                double dd = d->dDf * 0.6;
                if (d->iq_mode >= 2) {                 /* retro-rotate last sps samples, rebuild tone sums */
                    ora_cf X1 = cf(0, 0), X2 = cf(0, 0);
                    for (int n = (int)d->sps; n > 0; n--) {
                        uint32_t idx = (d->s_in - (uint32_t)n + (uint32_t)d->N) % (uint32_t)d->N;
                        double tn = (uint32_t)(d->s_in - (uint32_t)n) / (double)d->sr;
                        double a = -tn * TWO_PI * dd;
                        d->ziq[idx] = cmul_fd(d->ziq[idx], cos(a), sin(a));
                        ora_cf zz = d->ziq[idx];
                        /* float complex += float complex * double complex: sum formed in double (:1578-1579) */
                        double c1 = cos(-tn * d->w1), s1 = sin(-tn * d->w1), c2 = cos(-tn * d->w2), s2 = sin(-tn * d->w2);
                        X1 = cf((float)(X1.re + ((double)zz.re * c1 - (double)zz.im * s1)),
                                (float)(X1.im + ((double)zz.re * s1 + (double)zz.im * c1)));
                        X2 = cf((float)(X2.re + ((double)zz.re * c2 - (double)zz.im * s2)),
                                (float)(X2.im + ((double)zz.re * s2 + (double)zz.im * c2)));
                    }
                    d->F1 = X1; d->F2 = X2;
                }
                d->Df += dd;
            }
            if (fabs(d->dDf) > 1e3) { if (d->locked) { d->locked = 0; d->w_iq = d->w_iq_acq; } }
            else if (!d->locked)    { d->locked = 1; d->w_iq = d->w_iq_lock; }
        }
        if (d->mv_pos > prev && hdr_errors(d) <= hdmax) return 1;
    }
    return -1;
}

/* ---- soft-bit slicer, current and one-sample-early sums (demod_mod.c:1087-1175) ---- */
static int slice_half(ora_dsp *d, int inv, int ofs, float l, int spike, double *edge,
                      double sign, double dc, double *sum, double *sum1, float *avg) {
    const uint32_t Mm = (uint32_t)d->M;
    double mid = *edge + (d->sps - 1) / 2.0;
    *edge += d->sps;
    do {
        if (d->buffered > 0) d->buffered -= 1;
        else if (ora_sample(d, inv) < 0) return -1;
        uint32_t at = d->s_out - (uint32_t)d->buffered + (uint32_t)ofs + Mm;
        float smp = d->bufs[at % Mm], smp1 = d->bufs[(at - 1) % Mm];
        if (spike && fabs(smp - *avg) > 0.5f) {
            *avg = (float)(0.5 * (d->bufs[(at - 1) % Mm] + d->bufs[(at + 1) % Mm]));
            smp = *avg + 0.27f * (smp - *avg);
        }
        smp = (float)(smp - dc); smp1 = (float)(smp1 - dc);
        if (l < 0 || (mid - l < d->sc && d->sc < mid + l)) { *sum += sign * smp; *sum1 += sign * smp1; }
        d->sc++;
    } while (d->sc < *edge);
    return 0;
}
int ora_softbit2p(ora_dsp *d, ora_bit *b, int inv, int ofs, int pos, float l, int spike, ora_bit *b1) {
    double sum = 0, sum1 = 0, dc = 0, edge = (double)(pos * d->symlen * d->sps);
    float avg = 0;
    if (d->afc && d->iq_mode < 2) dc = d->dc;
    if (pos == 0) { edge = 0; d->sc = 0; }
    if (d->symlen == 2 && slice_half(d, inv, ofs, l, spike, &edge, -1.0, dc, &sum, &sum1, &avg) < 0) return -1;
    if (slice_half(d, inv, ofs, l, spike, &edge, +1.0, dc, &sum, &sum1, &avg) < 0) return -1;
    b->hb = sum >= 0;  b->sb = (float)sum;
    b1->hb = sum1 >= 0; b1->sb = (float)sum1;
    return 0;
}
