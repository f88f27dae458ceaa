// sonde_fsk.cpp — host side of the batched 2-FSK modem behind include/sonde_fsk.h (the reference's utils/fsk.c).
//
// fsk_create_core()'s constants and every data-independent table are computed here with the host libm — the same
// cosf/sinf the reference binary calls — so the device never evaluates a transcendental the reference evaluates on
// the CPU: Hann window (fsk.c:91-98), the per-sample oscillator step comp_exp_j(2 pi f / Fs) for every frequency the
// two estimators can return (fsk.c:643), and the timing oscillator's float recurrence (fsk.c:682-703).
#include "../../include/sonde_fsk.h"
#include "sonde_fsk_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

struct sonde_fsk {
    sonde_fsk_cfg_t cfg{};
    sonde_fsk_info_t info{};
    FskArgs args{};
    hipStream_t stream = nullptr;
    void *d_in = nullptr; float *d_hann = nullptr, *d_fmask = nullptr, *d_Sf = nullptr, *d_sd = nullptr, *d_eye = nullptr;
    unsigned long long *d_prof = nullptr;          // SONDE_FSK_PROF
    uint16_t *d_perm = nullptr;
    float2 *d_tw = nullptr, *d_dpeak = nullptr, *d_dmask = nullptr, *d_phift = nullptr, *d_tail = nullptr;
    FskChan *d_chan = nullptr; FskFrameRec *d_recs = nullptr; uint8_t *d_hb = nullptr; std::vector<uint8_t> h_hb;
    std::vector<FskChan> h_chan; std::vector<float> h_sd; std::vector<FskFrameRec> h_recs;
    bool pinned = false;
    size_t unit = 4;
    uint32_t wr = 0;
    std::vector<uint32_t> wr_ch; uint32_t *d_wr = nullptr;    // per-channel write positions once sonde_fsk_process_host_var is used
    double ms = 0; int64_t launches = 0;
    bool hb_on_host = false;                       // h_hb holds the last launch's hard bits (copied on the first sonde_fsk_fetch_bits behind a launch)
    // what a repeat of single channels needs (a pipeline that gave up, launch_and_collect): Sf and the tone tails as they were before the launch, the list
    float *d_Sf_bak = nullptr; float2 *d_tail_bak = nullptr; int *d_chlist = nullptr; std::vector<FskChan> h_chan_prev; int64_t repeats = 0;
};

template <class T> static int dalloc(T **p, size_t n, bool zero = true) {
    HIPCHK(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    if (zero) HIPCHK(hipMemset(*p, 0, (n ? n : 1) * sizeof(T)));
    return 0;
}
template <class T> static int dupload(T **p, const std::vector<T> &v) {
    if (dalloc(p, v.size(), false)) return SONDE_E_NOMEM;
    if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static float2 exp_j(float phi) { return make_float2(cosf(phi), sinf(phi)); }       // comp_exp_j (comp_prim.h:95)

extern "C" {

int sonde_fsk_create(const sonde_fsk_cfg_t *cfg, sonde_fsk_t **out) {
    if (!cfg || !out || cfg->abi_version != SONDE_ABI_VERSION) return SONDE_E_ARG;
    if (cfg->n_channels < 1 || cfg->Fs < 1 || cfg->Rs < 1 || cfg->P < 1 || cfg->nsym < 1 || cfg->max_chunk < 1) return SONDE_E_ARG;
    if (cfg->M != 2 && cfg->M != 4) return SONDE_E_ARG;                               // fsk.c:130
    if (cfg->Fs % cfg->Rs || (cfg->Fs / cfg->Rs) % cfg->P) return SONDE_E_ARG;          // the reference asserts (fsk.c:127-129)
    if (cfg->format != SONDE_FSK_S16 && cfg->format != SONDE_FSK_CS16 && cfg->format != SONDE_FSK_CU8 && cfg->format != SONDE_FSK_CF32) return SONDE_E_ARG;
    if (cfg->fsk_lower < -cfg->Fs / 2 || cfg->fsk_upper > cfg->Fs / 2 || cfg->fsk_upper <= cfg->fsk_lower) return SONDE_E_ARG;   // fsk.c:1019-1022
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device >= ndev) {
        fprintf(stderr, "libsonde_hip: no usable HIP device (the modem has no CPU fallback)\n");
        return SONDE_E_NOGPU;
    }
    HIPCHK(hipSetDevice(cfg->device));
    sonde_fsk *f = new sonde_fsk();
    f->cfg = *cfg;
    const int C = cfg->n_channels, Fs = cfg->Fs, Rs = cfg->Rs, P = cfg->P, nsym = cfg->nsym, M = cfg->M;

    // ---- fsk_create_core (fsk.c:114-201)
    const float bin_width_Hz = 0.1 * Rs;
    float Ndft_f = (float)Fs / bin_width_Hz;
    Ndft_f = pow(2.0, ceil(log2(Ndft_f)));
    const int Ndft = (int)Ndft_f, Ts = Fs / Rs, N = Ts * nsym, Nmem = N + 2 * Ts;
    int lg = 0; while ((1 << lg) < Ndft) lg++;
    if (Ndft > 1024 || Ndft < 8 || (1 << lg) != Ndft) { delete f; return SONDE_E_ARG; }
    FskArgs &a = f->args;
    a.format = cfg->format; a.M = M; a.burst = cfg->burst_mode ? 1 : 0; a.n_ch = C; a.Fs = Fs; a.Rs = Rs; a.Ts = Ts; a.P = P; a.nsym = nsym; a.N = N; a.Ndft = Ndft; a.log2Ndft = lg;
    a.Nmem = Nmem; a.NT = 2 * Ts + Ts / 2;
    a.tc = 0.95 * Ndft_f / Fs;
    const int est_space = 0.75 * Rs, fs_tx = cfg->mask ? cfg->tone_spacing : 100;
    a.fs_tx = fs_tx; a.est_type = cfg->mask ? 1 : 0;
    // fsk_demod_freq_est's bin limits (fsk.c:464-469), integer arithmetic
    a.st = (cfg->fsk_lower * Ndft) / Fs + Ndft / 2; if (a.st < 0) a.st = 0;
    a.en = (cfg->fsk_upper * Ndft) / Fs + Ndft / 2; if (a.en > Ndft) a.en = Ndft;
    a.f_zero = (est_space * Ndft) / Fs;
    {   // mask of the second estimator (fsk.c:553-560): ones at 0..2 and at bin_m..bin_m+2, bin_m = round(m fs_tx Ndft / Fs) - 1, m = 1..M-1
        std::vector<char> mask(Ndft + 8, 0);
        for (int i = 0; i < 3; i++) mask[i] = 1;
        int bin = 0; bool fits = true;
        for (int m = 1; m <= M - 1; m++) {
            bin = (int)round((float)m * fs_tx * Ndft / Fs) - 1;
            if (bin < 0 || bin + 2 >= Ndft) { fits = false; break; }
            for (int i = bin; i <= bin + 2; i++) mask[i] = 1;
        }
        if (!fits && cfg->mask) { delete f; return SONDE_E_ARG; }
        a.len_mask = bin + 2 + 1; a.n_mask = 0;
        for (int i = 0; i < Ndft && a.n_mask < 12; i++) if (mask[i]) a.mask_idx[a.n_mask++] = i;
    }
    f->info.Ts = Ts; f->info.N = N; f->info.Ndft = Ndft; f->info.Nmem = Nmem; f->info.Nbits = nsym * (M / 2); f->info.tc = a.tc;
    a.max_fft = (N + Ts / 2) / (Ndft / 2) - 1; if (a.max_fft < 1) a.max_fft = 1;

    // ---- tables
    std::vector<float> hann(Ndft), fmask((size_t)Ndft * M);
    std::vector<float2> tw(Ndft), dpeak(Ndft), dmask((size_t)Ndft * M), phift((size_t)(nsym + 1) * P);
    for (int i = 0; i < Ndft; i++) hann[i] = 0.5 - 0.5 * cosf(2.0 * M_PI * (float)i / (float)(Ndft - 1));
    std::vector<uint16_t> perm(Ndft);
    {   // kiss_fft_alloc / kf_factor / kf_work (kiss_fft.c:340-366, :304-331, :238-300): twiddles from cosf / sinf of the float phase,
        // factors 4,4,..(,2); output slot sum_s k_s m_s holds input sum_s k_s fstride_s; stages run innermost first
        for (int k = 0; k < Ndft; k++) {
            const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
            const double phase = -2 * pi * k / Ndft;
            tw[k] = make_float2(cosf(phase), sinf(phase));
        }
        int fp[8], fm[8], ffs[8], L = 0, n = Ndft, stride = 1;
        while (n > 1) { const int p = (n % 4 == 0) ? 4 : 2; n /= p; fp[L] = p; fm[L] = n; ffs[L] = stride; stride *= p; L++; }
        for (int o = 0; o < Ndft; o++) {
            int rem = o, in = 0;
            for (int s = 0; s < L; s++) { const int k = rem / fm[s]; rem -= k * fm[s]; in += k * ffs[s]; }
            perm[in] = (uint16_t)o;
        }
        a.n_stage = L;
        for (int s = 0; s < L; s++) { a.st_p[s] = fp[L - 1 - s]; a.st_m[s] = fm[L - 1 - s]; a.st_fs[s] = ffs[L - 1 - s]; }
    }
    for (int k = 0; k < Ndft; k++) {
        const float fp = (float)(k - Ndft / 2) * ((float)Fs / (float)Ndft);             // peak estimator (fsk.c:544-546)
        dpeak[k] = exp_j(2 * M_PI * ((fp) / (float)(Fs)));
        const float foff = (k - Ndft / 2) * Fs / Ndft;                                  // mask estimator (fsk.c:575-578), integer division
        for (int m = 0; m < M; m++) { const float fm = foff + m * fs_tx; fmask[M * k + m] = fm; dmask[M * k + m] = exp_j(2 * M_PI * ((fm) / (float)(Fs))); }
    }
    {   // timing oscillator: phi_ft = 1; used, then phi_ft *= dphift (fsk.c:682-703)
        const float2 d = exp_j(2 * M_PI * ((float)(Rs) / (float)(P * Rs)));
        float2 ph = make_float2(1.f, 0.f);
        for (size_t i = 0; i < phift.size(); i++) {
            phift[i] = ph;
            const float nr = ph.x * d.x - ph.y * d.y, ni = ph.x * d.y + ph.y * d.x;
            ph = make_float2(nr, ni);
        }
    }
    const int max_frames = cfg->max_chunk / std::max(1, N - Ts / 2) + 2;
    a.rec_cap = max_frames; a.sd_cap = max_frames * nsym * (M / 2);
    uint32_t ring = 1; while (ring < (uint32_t)(cfg->max_chunk + N + Ts + 16)) ring <<= 1;
    a.ring = ring;
    f->unit = cfg->format == SONDE_FSK_CF32 ? 8 : cfg->format == SONDE_FSK_CS16 ? 4 : 2;
    int bad = 0;
    bad |= dalloc((char **)&f->d_in, (size_t)C * ring * f->unit);
    bad |= dupload(&f->d_hann, hann); bad |= dupload(&f->d_tw, tw); bad |= dupload(&f->d_perm, perm); bad |= dupload(&f->d_dpeak, dpeak); bad |= dupload(&f->d_dmask, dmask);
    bad |= dupload(&f->d_fmask, fmask); bad |= dupload(&f->d_phift, phift);
    bad |= dalloc(&f->d_eye, (size_t)C * 8 * 160); bad |= dalloc(&f->d_Sf, (size_t)C * Ndft); bad |= dalloc(&f->d_tail, (size_t)C * M * a.NT);
    bad |= dalloc(&f->d_sd, (size_t)C * a.sd_cap); bad |= dalloc(&f->d_hb, (size_t)C * a.sd_cap); bad |= dalloc(&f->d_recs, (size_t)C * a.rec_cap); bad |= dalloc(&f->d_chan, (size_t)C, false);
    if (bad) { sonde_fsk_destroy(f); return SONDE_E_NOMEM; }
    f->h_chan.resize(C);
    for (auto &c : f->h_chan) { memset(&c, 0, sizeof c); for (int m = 0; m < 4; m++) c.phi_c[m] = exp_j(0); c.nin = N; }
    HIPCHK(hipMemcpy(f->d_chan, f->h_chan.data(), (size_t)C * sizeof(FskChan), hipMemcpyHostToDevice));
    a.in = f->d_in; a.hann = f->d_hann; a.tw = f->d_tw; a.perm = f->d_perm; a.dphi_peak = f->d_dpeak; a.dphi_mask = f->d_dmask; a.f_mask = f->d_fmask;
    a.phi_ft = f->d_phift; a.chan = f->d_chan; a.Sf = f->d_Sf; a.eye = f->d_eye; a.tail = f->d_tail; a.sd = f->d_sd; a.hb = f->d_hb; a.recs = f->d_recs;
    f->h_sd.resize((size_t)C * a.sd_cap); f->h_hb.resize((size_t)C * a.sd_cap); f->h_recs.resize((size_t)C * a.rec_cap);
    // the per-launch results come back into these (never resized again): page-locked, so that the copies are real asynchronous DMA and not staged through
    // a bounce buffer — with the pipelined kernel the copies of a thousand channels were a fifth of a step
    f->pinned = hipHostRegister(f->h_sd.data(), f->h_sd.size() * sizeof(float), hipHostRegisterDefault) == hipSuccess
             && hipHostRegister(f->h_hb.data(), f->h_hb.size(), hipHostRegisterDefault) == hipSuccess
             && hipHostRegister(f->h_recs.data(), f->h_recs.size() * sizeof(FskFrameRec), hipHostRegisterDefault) == hipSuccess
             && hipHostRegister(f->h_chan.data(), f->h_chan.size() * sizeof(FskChan), hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();                          // (registration is an optimisation: pageable buffers work too)
    HIPCHK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    *out = f;
    return 0;
}

void sonde_fsk_destroy(sonde_fsk_t *f) {
    if (!f) return;
    if (f->stream) { hipStreamSynchronize(f->stream); hipStreamDestroy(f->stream); }
    if (f->d_prof) {
        unsigned long long h[16];
        if (hipMemcpy(h, f->d_prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *nm[9] = { "input", "fft", "Sf+estimators", "oscillator", "downconv+tail", "integrate", "timing", "soft", "ebno+record" };
            if (h[15]) {          // the pipelined kernel: cycles of channel 0's waves, all of it and what they spent waiting
                fprintf(stderr, "fsk prof (Rs %d, nsym %d, channel 0, pipelined kernel): producer %.0f kcycles (waits: estimate %.1f%% ring %.1f%% length %.1f%%), "
                                "consumer %.0f (waits for samples %.1f%%, frame tail %.1f%%), estimators %.0f (waits %.1f%%)\n", f->cfg.Rs, f->cfg.nsym,
                        h[0] / 1e3, 100.0 * h[1] / std::max(1ull, h[0]), 100.0 * h[2] / std::max(1ull, h[0]), 100.0 * h[3] / std::max(1ull, h[0]),
                        h[4] / 1e3, 100.0 * h[5] / std::max(1ull, h[4]), 100.0 * h[6] / std::max(1ull, h[4]), h[8] / 1e3, 100.0 * h[9] / std::max(1ull, h[8]));
                fprintf(stderr, "fsk prof   estimator wave 0, kcycles: window %.0f stages %.0f magnitudes+Sf %.0f group barriers %.0f searches %.0f\n",
                        h[10] / 1e3, h[11] / 1e3, h[12] / 1e3, h[13] / 1e3, h[14] / 1e3);
            } else {
                unsigned long long tot = 0; for (int k = 0; k < 9; k++) tot += h[k];
                fprintf(stderr, "fsk prof (Rs %d, nsym %d, channel 0, %% of %llu cycles):", f->cfg.Rs, f->cfg.nsym, tot);
                for (int k = 0; k < 9; k++) fprintf(stderr, " %s %.1f", nm[k], tot ? 100.0 * (double)h[k] / (double)tot : 0.0);
                fprintf(stderr, "\n");
            }
        }
        hipFree(f->d_prof);
    }
    if (!f->h_sd.empty()) { hipHostUnregister(f->h_sd.data()); hipHostUnregister(f->h_hb.data()); hipHostUnregister(f->h_recs.data()); hipHostUnregister(f->h_chan.data()); (void)hipGetLastError(); }
    void *ptrs[] = { f->d_in, f->d_hann, f->d_fmask, f->d_Sf, f->d_sd, f->d_tw, f->d_dpeak, f->d_dmask, f->d_phift, f->d_tail, f->d_chan, f->d_recs, f->d_eye, f->d_hb, f->d_perm, f->d_wr, f->d_Sf_bak, f->d_tail_bak, f->d_chlist };
    for (void *p : ptrs) if (p) hipFree(p);
    delete f;
}

int sonde_fsk_info(const sonde_fsk_t *f, sonde_fsk_info_t *info) {
    if (!f || !info) return SONDE_E_ARG;
    *info = f->info;
    return 0;
}

static int collect(sonde_fsk_t *f) {
    const int C = f->cfg.n_channels;
    HIPCHK(hipMemcpyAsync(f->h_chan.data(), f->d_chan, (size_t)C * sizeof(FskChan), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(f->h_sd.data(), f->d_sd, f->h_sd.size() * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    f->hb_on_host = false;                                    // the hard bits follow when somebody asks for them (sonde_fsk_fetch_bits): auto_rx's pipelines read the soft decisions
    HIPCHK(hipMemcpyAsync(f->h_recs.data(), f->d_recs, f->h_recs.size() * sizeof(FskFrameRec), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return 0;
}
static int launch_and_collect(sonde_fsk_t *f) {
    const int C = f->cfg.n_channels;
    FskArgs &a = f->args;
    // The pipelined kernel's waves wait for each other with a bound (FSK_SPIN_MAX); a wait that runs out — a bug, or a device slowed to a crawl under a profiler —
    // ends that channel's launch with frames = -1.  Such channels are repeated with the frame-at-a-time kernel (same arithmetic, no waits between waves) from the
    // state they had before the launch: the channel records are still on the host, Sf and the tone tails are copied aside first (two small device copies).
    const int Ndft = f->info.Ndft;
    if (!f->d_Sf_bak) {
        if (dalloc(&f->d_Sf_bak, (size_t)C * Ndft, false) || dalloc(&f->d_tail_bak, (size_t)C * a.M * a.NT, false) || dalloc(&f->d_chlist, (size_t)C, false)) return SONDE_E_NOMEM;
    }
    f->h_chan_prev = f->h_chan;
    HIPCHK(hipMemcpyAsync(f->d_Sf_bak, f->d_Sf, (size_t)C * Ndft * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_tail_bak, f->d_tail, (size_t)C * a.M * a.NT * sizeof(float2), hipMemcpyDeviceToDevice, f->stream));
    { const char *t = getenv("SONDE_FSK_TEST_ABORT"); a.test_abort_ch = t ? atoi(t) : -1; }
    a.ch_list = nullptr; a.force_demod = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, f->stream);
    static const bool want_prof = getenv("SONDE_FSK_PROF") != nullptr;            // profiling aid: cycles per phase of channel 0, printed when the modem is destroyed
    if (want_prof && !f->d_prof) { if (hipMalloc((void **)&f->d_prof, 16 * sizeof(unsigned long long)) == hipSuccess) hipMemset(f->d_prof, 0, 16 * sizeof(unsigned long long)); }
    a.prof = f->d_prof;
    const int lrc = sonde_launch_fsk(&a, f->stream);
    hipEventRecord(e1, f->stream);
    if (lrc < 0) { hipEventDestroy(e0); hipEventDestroy(e1); return lrc == -1 ? SONDE_E_ARG : SONDE_E_NOGPU; }
    { const int rc = collect(f); if (rc) { hipEventDestroy(e0); hipEventDestroy(e1); return rc; } }
    float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { f->ms += ms; f->launches++; }
    hipEventDestroy(e0); hipEventDestroy(e1);
    std::vector<int> bad;
    for (int c = 0; c < C; c++) if (f->h_chan[c].frames < 0) bad.push_back(c);
    if (bad.empty()) return 0;
    fprintf(stderr, "libsonde_hip: fsk modem pipeline gave up on %zu channel(s) (first: %d): repeating them frame by frame\n", bad.size(), bad[0]);
    for (int c : bad) {
        HIPCHK(hipMemcpyAsync(f->d_chan + c, &f->h_chan_prev[c], sizeof(FskChan), hipMemcpyHostToDevice, f->stream));
        HIPCHK(hipMemcpyAsync(f->d_Sf + (size_t)c * Ndft, f->d_Sf_bak + (size_t)c * Ndft, (size_t)Ndft * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
        HIPCHK(hipMemcpyAsync(f->d_tail + (size_t)c * a.M * a.NT, f->d_tail_bak + (size_t)c * a.M * a.NT, (size_t)a.M * a.NT * sizeof(float2), hipMemcpyDeviceToDevice, f->stream));
    }
    HIPCHK(hipMemcpyAsync(f->d_chlist, bad.data(), bad.size() * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));               // (the copies read host vectors)
    FskArgs b = a;
    b.ch_list = f->d_chlist; b.n_ch = (int)bad.size(); b.force_demod = 1; b.test_abort_ch = -1; b.prof = nullptr;
    const int lrc2 = sonde_launch_fsk(&b, f->stream);
    if (lrc2 < 0) return lrc2 == -1 ? SONDE_E_ARG : SONDE_E_NOGPU;
    { const int rc = collect(f); if (rc) return rc; }
    f->repeats += (int64_t)bad.size();
    for (int c = 0; c < C; c++) if (f->h_chan[c].frames < 0) { fprintf(stderr, "libsonde_hip: fsk modem: channel %d failed again\n", c); return SONDE_E_NOGPU; }
    return 0;
}

// append n samples of one channel to its ring at absolute position w (two pieces when it wraps)
static int ring_write(sonde_fsk_t *f, int ch, uint32_t w, const char *src, int32_t n, hipMemcpyKind kind) {
    const FskArgs &a = f->args;
    const uint32_t w0 = w & (a.ring - 1), first = std::min<uint32_t>((uint32_t)n, a.ring - w0);
    char *dst = (char *)f->d_in + (size_t)ch * a.ring * f->unit;
    HIPCHK(hipMemcpyAsync(dst + (size_t)w0 * f->unit, src, (size_t)first * f->unit, kind, f->stream));
    if (first < (uint32_t)n) HIPCHK(hipMemcpyAsync(dst, src + (size_t)first * f->unit, (size_t)(n - first) * f->unit, kind, f->stream));
    return 0;
}

static int run(sonde_fsk_t *f, const void *src, int64_t ch_stride, int32_t n, hipMemcpyKind kind) {
    const int C = f->cfg.n_channels;
    if (n <= 0 || n > f->cfg.max_chunk || ch_stride < n) return SONDE_E_RANGE;
    if (!f->wr_ch.empty()) return SONDE_E_ARG;                  // the engine was switched to per-channel feeding
    FskArgs &a = f->args;
    // append to the per-channel rings (two pieces when the write position wraps)
    const uint32_t w0 = f->wr & (a.ring - 1);
    const uint32_t first = std::min<uint32_t>((uint32_t)n, a.ring - w0);
    char *dst = (char *)f->d_in;
    HIPCHK(hipMemcpy2DAsync(dst + (size_t)w0 * f->unit, (size_t)a.ring * f->unit, src, (size_t)ch_stride * f->unit, (size_t)first * f->unit, C, kind, f->stream));
    if (first < (uint32_t)n)
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)a.ring * f->unit, (const char *)src + (size_t)first * f->unit, (size_t)ch_stride * f->unit,
                                (size_t)(n - first) * f->unit, C, kind, f->stream));
    f->wr += (uint32_t)n;
    a.wr = f->wr;
    return launch_and_collect(f);
}

int sonde_fsk_process_host(sonde_fsk_t *f, const void *h_in, int64_t ch_stride, int32_t n_samples) {
    if (!f || !h_in) return SONDE_E_ARG;
    return run(f, h_in, ch_stride, n_samples, hipMemcpyHostToDevice);
}
int sonde_fsk_process_device(sonde_fsk_t *f, const void *d_in, int64_t ch_stride, int32_t n_samples) {
    if (!f || !d_in) return SONDE_E_ARG;
    return run(f, d_in, ch_stride, n_samples, hipMemcpyDeviceToDevice);
}

int sonde_fsk_process_host_var(sonde_fsk_t *f, const void *const *h_in, const int32_t *n_samples) {
    if (!f || !h_in || !n_samples) return SONDE_E_ARG;
    const int C = f->cfg.n_channels;
    if (f->wr_ch.empty()) {
        if (f->wr != 0) return SONDE_E_ARG;                     // one feeding mode per engine
        f->wr_ch.assign(C, 0);
        if (dalloc(&f->d_wr, (size_t)C)) return SONDE_E_NOMEM;
        f->args.wr_ch = f->d_wr;
    }
    for (int c = 0; c < C; c++) if (n_samples[c] < 0 || n_samples[c] > f->cfg.max_chunk || (n_samples[c] > 0 && !h_in[c])) return SONDE_E_RANGE;
    for (int c = 0; c < C; c++) {
        if (n_samples[c] == 0) continue;
        if (ring_write(f, c, f->wr_ch[c], (const char *)h_in[c], n_samples[c], hipMemcpyHostToDevice)) return SONDE_E_NOGPU;
        f->wr_ch[c] += (uint32_t)n_samples[c];
    }
    HIPCHK(hipMemcpyAsync(f->d_wr, f->wr_ch.data(), (size_t)C * sizeof(uint32_t), hipMemcpyHostToDevice, f->stream));
    return launch_and_collect(f);
}

int sonde_fsk_reset_channel(sonde_fsk_t *f, int32_t channel) {
    if (!f || channel < 0 || channel >= f->cfg.n_channels) return SONDE_E_ARG;
    HIPCHK(hipStreamSynchronize(f->stream));
    const FskArgs &a = f->args;
    FskChan c; memset(&c, 0, sizeof c);
    for (int m = 0; m < 4; m++) c.phi_c[m] = exp_j(0);
    c.nin = f->info.N;
    c.rd = f->wr_ch.empty() ? f->wr : f->wr_ch[channel];        // nothing queued: the next sample fed is this stream's first
    f->h_chan[channel] = c;
    HIPCHK(hipMemcpy(f->d_chan + channel, &c, sizeof c, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(f->d_Sf + (size_t)channel * f->info.Ndft, 0, (size_t)f->info.Ndft * sizeof(float)));
    HIPCHK(hipMemset(f->d_tail + (size_t)channel * a.M * a.NT, 0, (size_t)a.M * a.NT * sizeof(float2)));
    HIPCHK(hipMemset(f->d_eye + (size_t)channel * 8 * 160, 0, 8 * 160 * sizeof(float)));
    return 0;
}

int sonde_fsk_fetch(sonde_fsk_t *f, int32_t channel, float *sd, int32_t max, sonde_fsk_frame_t *frames, int32_t max_frames, int32_t *n_frames) {
    if (!f || channel < 0 || channel >= f->cfg.n_channels || (!sd && max > 0)) return SONDE_E_ARG;
    const FskChan &c = f->h_chan[channel];
    const int nf = c.frames, nb = std::min<int>(nf * f->info.Nbits, max);
    if (nb > 0) memcpy(sd, f->h_sd.data() + (size_t)channel * f->args.sd_cap, (size_t)nb * sizeof(float));
    if (frames) for (int i = 0; i < std::min(nf, max_frames); i++) {
        const FskFrameRec &r = f->h_recs[(size_t)channel * f->args.rec_cap + i];
        sonde_fsk_frame_t &o = frames[i];
        o.nin = r.nin; o.nin_next = r.nin_next; for (int m = 0; m < 4; m++) o.f_est[m] = r.f_est[m];
        o.norm_rx_timing = r.norm_rx_timing; o.ppm = r.ppm; o.EbNodB = r.EbNodB; o.snr_est = r.snr_est;
    }
    if (n_frames) *n_frames = nf;
    return nb;
}

int sonde_fsk_fetch_bits(sonde_fsk_t *f, int32_t channel, uint8_t *bits, int32_t max) {
    if (!f || channel < 0 || channel >= f->cfg.n_channels || (!bits && max > 0)) return SONDE_E_ARG;
    if (!f->hb_on_host) {
        HIPCHK(hipMemcpyAsync(f->h_hb.data(), f->d_hb, f->h_hb.size(), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        f->hb_on_host = true;
    }
    const int nb = std::min<int>(f->h_chan[channel].frames * f->info.Nbits, max);
    if (nb > 0) memcpy(bits, f->h_hb.data() + (size_t)channel * f->args.sd_cap, (size_t)nb);
    return nb;
}

int sonde_fsk_stats(sonde_fsk_t *f, int32_t channel, sonde_fsk_frame_t *last, float *Sf, int64_t *samples) {
    if (!f || channel < 0 || channel >= f->cfg.n_channels) return SONDE_E_ARG;
    const FskChan &c = f->h_chan[channel];
    if (last) {
        memset(last, 0, sizeof *last);
        last->nin_next = c.nin; for (int m = 0; m < f->cfg.M; m++) last->f_est[m] = c.f_est[m]; last->norm_rx_timing = c.norm_rx_timing;
        last->ppm = c.ppm; last->EbNodB = c.EbNodB; last->snr_est = c.snr_est;
    }
    if (Sf) HIPCHK(hipMemcpy(Sf, f->d_Sf + (size_t)channel * f->info.Ndft, (size_t)f->info.Ndft * sizeof(float), hipMemcpyDeviceToHost));
    if (samples) *samples = c.samples;
    return 0;
}

int sonde_fsk_eye(sonde_fsk_t *f, int32_t channel, float *eye, int32_t *neyetr, int32_t *neyesamp) {
    if (!f || !eye || channel < 0 || channel >= f->cfg.n_channels) return SONDE_E_ARG;
    const int P = f->cfg.P;
    const int dec = (int)ceil(((float)P * 2) / 160.0f), nes = (P * 2) / dec, ntr = 8;      // MODEM_STATS_EYE_IND_MAX 160, ET_MAX 8
    std::vector<float> raw(8 * 160);
    HIPCHK(hipMemcpy(raw.data(), f->d_eye + (size_t)channel * 8 * 160, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
    float eye_max = 1.f;
    if (!f->cfg.raw_eye) {                                    // normalise_eye = 1 unless fsk_stats_normalise_eye(.., 0) (fsk.c:198,892-903)
        eye_max = 0;
        for (int i = 0; i < ntr; i++) for (int j = 0; j < nes; j++) if (fabsf(raw[i * 160 + j]) > eye_max) eye_max = fabsf(raw[i * 160 + j]);
    }
    for (int i = 0; i < ntr; i++) for (int j = 0; j < nes; j++) eye[i * nes + j] = raw[i * 160 + j] / eye_max;
    if (neyetr) *neyetr = ntr;
    if (neyesamp) *neyesamp = nes;
    return ntr * nes;
}

int sonde_fsk_clear_estimators(sonde_fsk_t *f) {               // fsk_clear_estimators (fsk.c:981-989): Sf = 0, nin = N
    if (!f) return SONDE_E_ARG;
    const int C = f->cfg.n_channels;
    HIPCHK(hipStreamSynchronize(f->stream));
    HIPCHK(hipMemset(f->d_Sf, 0, (size_t)C * f->info.Ndft * sizeof(float)));
    HIPCHK(hipMemcpy(f->h_chan.data(), f->d_chan, (size_t)C * sizeof(FskChan), hipMemcpyDeviceToHost));
    for (auto &c : f->h_chan) c.nin = f->info.N;
    HIPCHK(hipMemcpy(f->d_chan, f->h_chan.data(), (size_t)C * sizeof(FskChan), hipMemcpyHostToDevice));
    return 0;
}

int sonde_fsk_kernel_ms(sonde_fsk_t *f, double *avg_ms, int64_t *launches) {
    if (!f) return SONDE_E_ARG;
    if (avg_ms) *avg_ms = f->launches ? f->ms / (double)f->launches : 0.0;
    if (launches) *launches = f->launches;
    return 0;
}

}  // extern "C"
