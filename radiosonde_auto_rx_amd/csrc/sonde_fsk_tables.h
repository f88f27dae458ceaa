// sonde_fsk_tables.h — fsk_create_core()'s constants (fsk.c:114-201) and every data-independent table of the modem, computed with the
// host libm — the same cosf / sinf the reference binary calls — so that the device never evaluates a transcendental the reference
// evaluates on the CPU.  No HIP in here: sonde_fsk.cpp uploads the tables, tests/emu/fsk_wave_emu.cpp (test infrastructure) uses them as
// they are.
#ifndef SONDE_FSK_TABLES_H
#define SONDE_FSK_TABLES_H
#include "../../include/sonde_fsk.h"
#include "sonde_fsk_dev.h"
#include <cmath>
#include <vector>

struct FskTables {
    int Ndft = 0, lg = 0, Ts = 0, N = 0, Nmem = 0, NT = 0;
    float tc = 0;
    int st = 0, en = 0, f_zero = 0, len_mask = 0, n_mask = 0, mask_idx[12] = {0}, fs_tx = 0, est_type = 0, max_fft = 1;
    int n_stage = 0, st_p[8] = {0}, st_m[8] = {0}, st_fs[8] = {0};
    std::vector<float> hann, fmask;                       // [Ndft], [Ndft][M]
    std::vector<float2> tw, dpeak, dmask, phift;          // [Ndft], [Ndft], [Ndft][M], [(nsym + 1) P]
    std::vector<uint16_t> perm, iperm;                    // [Ndft] each
};

static inline float2 fsk_exp_j(float phi) { return make_float2(cosf(phi), sinf(phi)); }       // comp_exp_j (comp_prim.h:95)

// 0, or SONDE_E_ARG for a configuration the reference's asserts (or this implementation's table sizes) reject
static inline int fsk_build_tables(const sonde_fsk_cfg_t &cfg, FskTables &T) {
    const int Fs = cfg.Fs, Rs = cfg.Rs, P = cfg.P, nsym = cfg.nsym, M = cfg.M;
    // ---- fsk_create_core (fsk.c:114-201)
    const float bin_width_Hz = 0.1 * Rs;
    float Ndft_f = (float)Fs / bin_width_Hz;
    Ndft_f = pow(2.0, ceil(log2(Ndft_f)));
    const int Ndft = (int)Ndft_f, Ts = Fs / Rs, N = Ts * nsym, Nmem = N + 2 * Ts;
    int lg = 0; while ((1 << lg) < Ndft) lg++;
    if (Ndft > 1024 || Ndft < 8 || (1 << lg) != Ndft) return SONDE_E_ARG;
    T.Ndft = Ndft; T.lg = lg; T.Ts = Ts; T.N = N; T.Nmem = Nmem; T.NT = 2 * Ts + Ts / 2;
    T.tc = 0.95 * Ndft_f / Fs;
    const int est_space = 0.75 * Rs, fs_tx = cfg.mask ? cfg.tone_spacing : 100;
    T.fs_tx = fs_tx; T.est_type = cfg.mask ? 1 : 0;
    // fsk_demod_freq_est's bin limits (fsk.c:464-469), integer arithmetic
    T.st = (cfg.fsk_lower * Ndft) / Fs + Ndft / 2; if (T.st < 0) T.st = 0;
    T.en = (cfg.fsk_upper * Ndft) / Fs + Ndft / 2; if (T.en > Ndft) T.en = Ndft;
    T.f_zero = (est_space * Ndft) / Fs;
    {   // mask of the second estimator (fsk.c:553-560): ones at 0..2 and at bin_m..bin_m+2, bin_m = round(m fs_tx Ndft / Fs) - 1, m = 1..M-1
        std::vector<char> mask(Ndft + 8, 0);
        for (int i = 0; i < 3; i++) mask[i] = 1;
        int bin = 0; bool fits = true;
        for (int m = 1; m <= M - 1; m++) {
            bin = (int)round((float)m * fs_tx * Ndft / Fs) - 1;
            if (bin < 0 || bin + 2 >= Ndft) { fits = false; break; }
            for (int i = bin; i <= bin + 2; i++) mask[i] = 1;
        }
        if (!fits && cfg.mask) return SONDE_E_ARG;
        T.len_mask = bin + 2 + 1; T.n_mask = 0;
        for (int i = 0; i < Ndft && T.n_mask < 12; i++) if (mask[i]) T.mask_idx[T.n_mask++] = i;
    }
    T.max_fft = (N + Ts / 2) / (Ndft / 2) - 1; if (T.max_fft < 1) T.max_fft = 1;

    // ---- tables
    T.hann.resize(Ndft); T.fmask.resize((size_t)Ndft * M);
    T.tw.resize(Ndft); T.dpeak.resize(Ndft); T.dmask.resize((size_t)Ndft * M); T.phift.resize((size_t)(nsym + 1) * P);
    T.perm.resize(Ndft); T.iperm.resize(Ndft);
    for (int i = 0; i < Ndft; i++) T.hann[i] = 0.5 - 0.5 * cosf(2.0 * M_PI * (float)i / (float)(Ndft - 1));
    {   // kiss_fft_alloc / kf_factor / kf_work (kiss_fft.c:340-366, :304-331, :238-300): twiddles from cosf / sinf of the float phase,
        // factors 4,4,..(,2); output slot sum_s k_s m_s holds input sum_s k_s fstride_s; stages run innermost first
        for (int k = 0; k < Ndft; k++) {
            const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
            const double phase = -2 * pi * k / Ndft;
            T.tw[k] = make_float2(cosf(phase), sinf(phase));
        }
        int fp[8], fm[8], ffs[8], L = 0, n = Ndft, stride = 1;
        while (n > 1) { const int p = (n % 4 == 0) ? 4 : 2; n /= p; fp[L] = p; fm[L] = n; ffs[L] = stride; stride *= p; L++; }
        for (int o = 0; o < Ndft; o++) {
            int rem = o, in = 0;
            for (int s = 0; s < L; s++) { const int k = rem / fm[s]; rem -= k * fm[s]; in += k * ffs[s]; }
            T.perm[in] = (uint16_t)o; T.iperm[o] = (uint16_t)in;
        }
        T.n_stage = L;
        for (int s = 0; s < L; s++) { T.st_p[s] = fp[L - 1 - s]; T.st_m[s] = fm[L - 1 - s]; T.st_fs[s] = ffs[L - 1 - s]; }
    }
    for (int k = 0; k < Ndft; k++) {
        const float fp = (float)(k - Ndft / 2) * ((float)Fs / (float)Ndft);             // peak estimator (fsk.c:544-546)
        T.dpeak[k] = fsk_exp_j(2 * M_PI * ((fp) / (float)(Fs)));
        const float foff = (k - Ndft / 2) * Fs / Ndft;                                  // mask estimator (fsk.c:575-578), integer division
        for (int m = 0; m < M; m++) { const float fm = foff + m * fs_tx; T.fmask[M * k + m] = fm; T.dmask[M * k + m] = fsk_exp_j(2 * M_PI * ((fm) / (float)(Fs))); }
    }
    {   // timing oscillator: phi_ft = 1; used, then phi_ft *= dphift (fsk.c:682-703)
        const float2 d = fsk_exp_j(2 * M_PI * ((float)(Rs) / (float)(P * Rs)));
        float2 ph = make_float2(1.f, 0.f);
        for (size_t i = 0; i < T.phift.size(); i++) {
            T.phift[i] = ph;
            const float nr = ph.x * d.x - ph.y * d.y, ni = ph.x * d.y + ph.y * d.x;
            ph = make_float2(nr, ni);
        }
    }
    return 0;
}

// the scalar part of the kernel arguments (pointers are the caller's)
static inline void fsk_tables_to_args(const sonde_fsk_cfg_t &cfg, const FskTables &T, FskArgs &a) {
    a.format = cfg.format; a.M = cfg.M; a.burst = cfg.burst_mode ? 1 : 0; a.Fs = cfg.Fs; a.Rs = cfg.Rs; a.Ts = T.Ts; a.P = cfg.P; a.nsym = cfg.nsym;
    a.N = T.N; a.Ndft = T.Ndft; a.log2Ndft = T.lg; a.Nmem = T.Nmem; a.NT = T.NT; a.tc = T.tc;
    a.fs_tx = T.fs_tx; a.est_type = T.est_type; a.st = T.st; a.en = T.en; a.f_zero = T.f_zero; a.len_mask = T.len_mask; a.n_mask = T.n_mask;
    for (int i = 0; i < 12; i++) a.mask_idx[i] = T.mask_idx[i];
    a.max_fft = T.max_fft; a.n_stage = T.n_stage;
    for (int s = 0; s < 8; s++) { a.st_p[s] = T.st_p[s]; a.st_m[s] = T.st_m[s]; a.st_fs[s] = T.st_fs[s]; }
}
#endif
