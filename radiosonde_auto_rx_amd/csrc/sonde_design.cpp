// sonde_design.cpp — host-side filter / table design of the engine (product code, no oracle dependency).
//
// Reproduces the numbers init_buffers() derives (reference demod/mod/demod_mod.c:1208-1474) with the same
// float/double evaluation order, because the taps, the mixer table and the header template are *data* the
// GPU kernels must share with the reference to stay inside the soft-bit tolerance:
//   design_lowpass   Blackman x sinc, 1-norm, float accumulate            demod_mod.c:555-587
//   design_decimator IF rate / decM / tap count                            demod_mod.c:1222-1259
//   design_mixer     16-Hz-snapped mixer frequency + table period           demod_mod.c:1262-1296
//   design_match     Gaussian-pulse header template, 2-norm                demod_mod.c:1190-1195,1398-1421
#include "sonde_host.h"
#include <cmath>

namespace sonde {

static const double kTwoPi = 6.2831853071795864769252867665590;

std::vector<float> design_lowpass(float f, int taps) {
    if (taps % 2 == 0) taps++;
    if (taps < 1) taps = 1;
    std::vector<float> ws(taps);
    double norm = 0.0;
    const int centre = (taps - 1) / 2;
    const float twof = 2 * f;
    for (int n = 0; n < taps; n++) {
        const double w = 7938 / 18608.0 - 9240 / 18608.0 * std::cos(kTwoPi * n / (taps - 1))
                       + 1430 / 18608.0 * std::cos(4 * M_PI * n / (taps - 1));
        const double x = (double)(twof * (float)(n - centre));
        const double sinc = (x == 0) ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        ws[n] = (float)(w * (twof * sinc));
        norm += ws[n];
    }
    for (int n = 0; n < taps; n++) ws[n] = (float)(ws[n] / norm);
    return ws;
}

Decimator design_decimator(int sr_base, bool if_min) {
    Decimator d;
    int if_sr = if_min ? 32000 : 48000;
    d.decM = 1;
    if (if_sr > sr_base) if_sr = sr_base;
    if (if_sr < sr_base) {
        while (sr_base % if_sr) if_sr += 1;
        d.decM = sr_base / if_sr;
    }
    const float f_lp = (float)((if_sr + 20e3) / (4.0 * sr_base));
    float t_bw = (float)(if_sr - 20e3);
    if (if_min) t_bw = (float)(if_sr - 12e3);
    if (t_bw < 0) t_bw = 10e3f;
    t_bw /= sr_base;
    int taps = (int)(4.0 / t_bw);
    if (taps % 2 == 0) taps++;
    d.if_sr = if_sr;
    d.taps = design_lowpass(f_lp, taps);
    return d;
}

// dft_detect's variant (dft_detect.c:1021-1067): --bw above 48 kHz raises the IF rate to that bandwidth; an IF above 60 kHz ("wideIF") gets a wider
// pass band and transition (IF + 60 kHz, IF - 60 kHz)
Decimator design_decimator_scan(int sr_base, bool if_min, float set_lpIQ) {
    Decimator d;
    int if_sr = 48000;
    if (set_lpIQ > (float)if_sr) if_sr = (int)set_lpIQ;
    const bool wide = if_sr > 60e3;
    if (if_min) if_sr = 32000;
    d.decM = 1;
    if (if_sr > sr_base) if_sr = sr_base;
    if (if_sr < sr_base) {
        while (sr_base % if_sr) if_sr += 1;
        d.decM = sr_base / if_sr;
    }
    float f_lp = (float)((if_sr + 20e3) / (4.0 * sr_base));
    float t_bw = (float)(if_sr - 20e3);
    if (wide) { f_lp = (float)((if_sr + 60e3) / (4.0 * sr_base)); t_bw = (float)(if_sr - 60e3); }
    else if (if_min) t_bw = (float)(if_sr - 12e3);
    if (t_bw < 0) t_bw = 10e3f;
    t_bw /= sr_base;
    int taps = (int)(4.0 / t_bw);
    if (taps % 2 == 0) taps++;
    d.if_sr = if_sr;
    d.taps = design_lowpass(f_lp, taps);
    return d;
}

// iq_dec's variant (iq_dec.c:632-666): the IF rate is a parameter (--IFbw), --min only narrows the transition band
Decimator design_decimator_if(int sr_base, int if_target, bool narrow) {
    Decimator d;
    int if_sr = if_target;
    d.decM = 1;
    if (if_sr > sr_base) if_sr = sr_base;
    if (if_sr < sr_base) {
        while (sr_base % if_sr) if_sr += 1;
        d.decM = sr_base / if_sr;
    }
    const float f_lp = (float)((if_sr + 20e3) / (4.0 * sr_base));
    float t_bw = (float)(if_sr - 20e3);
    if (narrow) t_bw = (float)(if_sr - 12e3);
    if (t_bw < 0) t_bw = 10e3f;
    t_bw /= sr_base;
    int taps = (int)(4.0 / t_bw);
    if (taps % 2 == 0) taps++;
    d.if_sr = if_sr;
    d.taps = design_lowpass(f_lp, taps);
    return d;
}

// Mixer: the reference snaps the frequency to a multiple of d Hz (d = largest divisor <= 16 of the sample rate)
// and tabulates ex[n] = cexp(2 pi i * fl32(f0*n)) over one period lut_len = sr/d.  The kernels evaluate the
// same expression on the fly, so only (f0, lut_len) are needed.
Mixer design_mixer(double xlt_fq, int sr_base) {
    const int W = 16;
    int d;
    const int freq = (int)(xlt_fq * (double)sr_base + 0.5);
    int freq0 = freq;
    for (d = W; d > 0; d--) if (sr_base % d == 0) break;
    if (d == 0) d = 1;
    for (int k = 0; k < W / 2; k++) {
        if ((freq + k) % d == 0) { freq0 = freq + k; break; }
        if ((freq - k) % d == 0) { freq0 = freq - k; break; }
    }
    Mixer m;
    m.lut_len = sr_base / d;
    m.f0 = freq0 / (double)sr_base;
    return m;
}

static double gauss_q(double x) { return 0.5 - 0.5 * std::erf(x / 1.4142135624); }
static double gauss_pulse(double t, double sigma) { return gauss_q((t - 0.5) / sigma) - gauss_q((t + 0.5) / sigma); }

std::vector<float> design_match(const std::string &hdr, float sps, float bt) {
    const int hdrlen = (int)hdr.size();
    const int L = (int)(hdrlen * sps + 0.5);
    std::vector<float> m(L);
    const double sigma = std::sqrt(std::log(2)) / (kTwoPi * bt);
    for (int i = 0; i < L; i++) {
        const int pos = (int)(i / sps);
        const float t = (float)((i - pos * sps) / sps - 0.5);
        const float b1 = (float)(((hdr[pos] & 1) - 0.5) * 2.0);
        float b = (float)(b1 * gauss_pulse(t, sigma));
        if (pos > 0) {
            const float b0 = (float)(((hdr[pos - 1] & 1) - 0.5) * 2.0);
            b = (float)(b + b0 * gauss_pulse((double)(t + 1), sigma));
        }
        if (pos < hdrlen - 1) {
            const float b2 = (float)(((hdr[pos + 1] & 1) - 0.5) * 2.0);
            b = (float)(b + b2 * gauss_pulse((double)(t - 1), sigma));
        }
        m[i] = b;
    }
    double n2 = 0.0;
    for (int i = 0; i < L; i++) { const double x = m[i]; n2 += x * x; }
    const float nm = (float)std::sqrt(n2);
    for (int i = 0; i < L; i++) m[i] /= nm;
    return m;
}

// Consumed-sample window of one bit half; host twin of bit_window() in sonde_kernels.hip
// (read_softbit2p, demod_mod.c:1098-1161).
void bit_window(int pos, int half, int symlen, float sps, uint32_t &q0, uint32_t &q1, double &mid) {
    double bg = (pos == 0) ? 0.0 : (double)((float)(pos * symlen) * sps);
    double prev;
    if (half == 0) {
        if (pos == 0) prev = 0.0;
        else {
            prev = (pos == 1) ? 0.0 : (double)((float)((pos - 1) * symlen) * sps);
            prev += (double)sps;
            if (symlen == 2) prev += (double)sps;
        }
    } else {
        bg += (double)sps;
        prev = bg;
    }
    q0 = (uint32_t)std::ceil(prev);
    mid = bg + (double)(sps - 1.0f) / 2.0;
    q1 = (uint32_t)std::ceil(bg + (double)sps);
    if (q1 <= q0) q1 = q0 + 1;
}

// Range [qa, qb) of consumed-sample counts a symbol half actually sums: q in [q0, q1) with mid-l < q < mid+l
// (everything for l < 0) — the `if (l < 0 || (mid-l < dsp->sc && dsp->sc < mid+l))` of read_softbit2p.
void slice_range(uint32_t q0, uint32_t q1, double mid, float l, uint32_t &qa, uint32_t &qb) {
    qa = q0; qb = q1;
    if (!(l < 0.f)) {
        const double lo = mid - (double)l, hi = mid + (double)l;
        const double fa = std::floor(lo) + 1.0, fb = std::ceil(hi);      // smallest q > lo, smallest q >= hi
        if (fa > (double)qa) qa = (uint32_t)fa;
        if (fb < (double)qb) qb = (fb > 0.0) ? (uint32_t)fb : 0u;
    }
    if (qb < qa) qb = qa;
}

}  // namespace sonde
