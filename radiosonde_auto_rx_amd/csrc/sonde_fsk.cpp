// sonde_fsk.cpp — host side of the batched 2-FSK modem behind include/sonde_fsk.h (the reference's utils/fsk.c).
//
// fsk_create_core()'s constants and every data-independent table are computed here with the host libm — the same
// cosf/sinf the reference binary calls — so the device never evaluates a transcendental the reference evaluates on
// the CPU: Hann window (fsk.c:91-98), the per-sample oscillator step comp_exp_j(2 pi f / Fs) for every frequency the
// two estimators can return (fsk.c:643), and the timing oscillator's float recurrence (fsk.c:682-703).
#include "../../include/sonde_fsk.h"
#include "sonde_fsk_dev.h"
#include "sonde_fsk_tables.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "sonde_pinned.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

struct sonde_fsk {
    sonde_fsk_cfg_t cfg{};
    sonde_fsk_info_t info{};
    FskArgs args{};
    hipStream_t stream = nullptr;
    void *d_in = nullptr; float *d_hann = nullptr, *d_fmask = nullptr, *d_Sf = nullptr, *d_sd = nullptr, *d_eye = nullptr;
    float *d_sd_alt = nullptr;                     // the soft decisions of the launch before the last: d_sd and d_sd_alt swap at every launch, so a consumer on the device
                                                   // (sonde_softin_dev_submit_fsk) can still read launch k's while launch k + 1 runs
    unsigned long long *d_prof = nullptr;          // SONDE_FSK_PROF
    uint16_t *d_perm = nullptr, *d_iperm = nullptr;
    float2 *d_tw = nullptr, *d_dpeak = nullptr, *d_dmask = nullptr, *d_phift = nullptr, *d_tail = nullptr;
    FskChan *d_chan = nullptr; FskFrameRec *d_recs = nullptr; uint8_t *d_hb = nullptr; Pinned<uint8_t> h_hb;
    Pinned<FskChan> h_chan; Pinned<float> h_sd; Pinned<FskFrameRec> h_recs;          // page-locked landing buffers (sonde_pinned.h)
    bool pinned = false;
    size_t unit = 4;
    uint32_t wr = 0;
    std::vector<uint32_t> wr_ch; uint32_t *d_wr = nullptr;    // per-channel write positions once sonde_fsk_process_host_var is used
    double ms = 0; int64_t launches = 0;
    bool sd_on_host = false;                       // h_sd holds the last launch's soft decisions (copied on the first sonde_fsk_fetch behind a launch)
    bool recs_on_host = false;                     // h_recs holds the last launch's frame records (copied with the soft decisions)
    bool hb_on_host = false;                       // h_hb holds the last launch's hard bits (copied on the first sonde_fsk_fetch_bits behind a launch)
    // what a repeat of single channels needs (a pipeline that gave up, launch_wait): Sf and the tone tails as they were before the launch, the list
    float *d_Sf_bak = nullptr; float2 *d_tail_bak = nullptr; int *d_chlist = nullptr; float *d_scratch = nullptr; std::vector<FskChan> h_chan_prev; int64_t repeats = 0;
    // a launch that was submitted and not yet waited for (sonde_fsk_submit_device / sonde_fsk_wait); the channels the last wait had to repeat
    bool pending = false; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // consumers on the device read d_sd / d_sd_alt on streams of their own (sonde_softin_dev_submit_fsk): each of the two buffers remembers the last reader's end
    // (sonde_fsk_dev_reader_done) and the launch that is about to overwrite it waits for that event — whatever order the caller submits things in
    hipEvent_t ev_rd[2] = { nullptr, nullptr }; bool rd_set[2] = { false, false }; int sd_idx = 0;
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

static float2 exp_j(float phi) { return fsk_exp_j(phi); }

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
    const int C = cfg->n_channels, nsym = cfg->nsym, M = cfg->M;

    // ---- fsk_create_core's constants and every data-independent table (sonde_fsk_tables.h)
    FskTables T;
    if (fsk_build_tables(*cfg, T)) { delete f; return SONDE_E_ARG; }
    const int Ndft = T.Ndft, Ts = T.Ts, N = T.N;
    FskArgs &a = f->args;
    fsk_tables_to_args(*cfg, T, a);
    a.n_ch = C;
    f->info.Ts = Ts; f->info.N = N; f->info.Ndft = Ndft; f->info.Nmem = T.Nmem; f->info.Nbits = nsym * (M / 2); f->info.tc = a.tc;
    const std::vector<float> &hann = T.hann, &fmask = T.fmask;
    const std::vector<float2> &tw = T.tw, &dpeak = T.dpeak, &dmask = T.dmask, &phift = T.phift;
    const std::vector<uint16_t> &perm = T.perm, &iperm = T.iperm;
    const int max_frames = cfg->max_chunk / std::max(1, N - Ts / 2) + 2;
    a.rec_cap = max_frames; a.sd_cap = max_frames * nsym * (M / 2);
    uint32_t ring = 1; while (ring < (uint32_t)(cfg->max_chunk + N + Ts + 16)) ring <<= 1;
    a.ring = ring;
    f->unit = cfg->format == SONDE_FSK_CF32 ? 8 : cfg->format == SONDE_FSK_CS16 ? 4 : 2;
    int bad = 0;
    bad |= dalloc((char **)&f->d_in, (size_t)C * ring * f->unit);
    bad |= dupload(&f->d_hann, hann); bad |= dupload(&f->d_tw, tw); bad |= dupload(&f->d_perm, perm); bad |= dupload(&f->d_iperm, iperm); bad |= dupload(&f->d_dpeak, dpeak); bad |= dupload(&f->d_dmask, dmask);
    bad |= dupload(&f->d_fmask, fmask); bad |= dupload(&f->d_phift, phift);
    bad |= dalloc(&f->d_eye, (size_t)C * 8 * 160); bad |= dalloc(&f->d_Sf, (size_t)C * Ndft); bad |= dalloc(&f->d_tail, (size_t)C * M * a.NT);
    bad |= dalloc(&f->d_sd, (size_t)C * a.sd_cap); bad |= dalloc(&f->d_sd_alt, (size_t)C * a.sd_cap); bad |= dalloc(&f->d_hb, (size_t)C * a.sd_cap); bad |= dalloc(&f->d_recs, (size_t)C * a.rec_cap); bad |= dalloc(&f->d_chan, (size_t)C, false);
    if (bad) { sonde_fsk_destroy(f); return SONDE_E_NOMEM; }
    if (!f->h_chan.alloc(C)) { sonde_fsk_destroy(f); return SONDE_E_NOMEM; }
    for (auto &c : f->h_chan) { memset(&c, 0, sizeof c); for (int m = 0; m < 4; m++) c.phi_c[m] = exp_j(0); c.nin = N; }
    HIPCHK(hipMemcpy(f->d_chan, f->h_chan.data(), (size_t)C * sizeof(FskChan), hipMemcpyHostToDevice));
    a.in = f->d_in; a.hann = f->d_hann; a.tw = f->d_tw; a.perm = f->d_perm; a.iperm = f->d_iperm; a.dphi_peak = f->d_dpeak; a.dphi_mask = f->d_dmask; a.f_mask = f->d_fmask;
    a.phi_ft = f->d_phift; a.chan = f->d_chan; a.Sf = f->d_Sf; a.eye = f->d_eye; a.tail = f->d_tail; a.sd = f->d_sd; a.hb = f->d_hb; a.recs = f->d_recs;
    if (!f->h_sd.alloc((size_t)C * a.sd_cap) || !f->h_hb.alloc((size_t)C * a.sd_cap) || !f->h_recs.alloc((size_t)C * a.rec_cap)) { sonde_fsk_destroy(f); return SONDE_E_NOMEM; }
    // the per-launch results come back into these (never resized again): page-locked, so that the copies are real asynchronous DMA and not staged through
    // a bounce buffer — with the pipelined kernel the copies of a thousand channels were a fifth of a step
    f->pinned = true;
    HIPCHK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    *out = f;
    return 0;
}

void sonde_fsk_destroy(sonde_fsk_t *f) {
    if (!f) return;
    if (f->stream) { hipStreamSynchronize(f->stream); hipStreamDestroy(f->stream); }
    if (f->ev0) { hipEventDestroy(f->ev0); hipEventDestroy(f->ev1); }
    for (hipEvent_t ev : f->ev_rd) if (ev) hipEventDestroy(ev);
    if (f->d_prof) {
        unsigned long long h[32];
        if (hipMemcpy(h, f->d_prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *nm[9] = { "input", "fft", "Sf+estimators", "oscillator", "downconv+tail", "integrate", "timing", "soft", "ebno+record" };
            if (h[15] >= 2) {     // the wave form (sonde_fsk_wave.h): cycles of channel 0's worker (and walker) by what they were doing
                const double tw = (double)(h[0] + h[1] + h[2] + h[3] + h[4]), tk = (double)(h[5] + h[6] + h[7]);
                fprintf(stderr, "fsk prof (Rs %d, nsym %d, channel 0, wave form, %s): worker %.0f kcycles: barriers / waiting %.1f%% estimator %.1f%% down-conversion %.1f%% "
                                "integrators + timing sum %.1f%% frame end %.1f%%", f->cfg.Rs, f->cfg.nsym, h[15] == 3 ? "walker + worker" : "one wave", tw / 1e3,
                        100.0 * h[0] / std::max(1.0, tw), 100.0 * h[1] / std::max(1.0, tw), 100.0 * h[2] / std::max(1.0, tw), 100.0 * h[3] / std::max(1.0, tw), 100.0 * h[4] / std::max(1.0, tw));
                if (h[15] == 3) fprintf(stderr, "; walker %.0f kcycles: oscillator %.1f%% barriers / waiting %.1f%% other %.1f%%; estimator: transforms + searches %.1f%% barriers / waiting %.1f%%", tk / 1e3, 100.0 * h[5] / std::max(1.0, tk), 100.0 * h[7] / std::max(1.0, tk), 100.0 * h[6] / std::max(1.0, tk),
                                        100.0 * h[8] / std::max(1.0, (double)(h[8] + h[9])), 100.0 * h[9] / std::max(1.0, (double)(h[8] + h[9])));
                else fprintf(stderr, " (oscillator %.1f%% of it)", 100.0 * h[5] / std::max(1.0, tw + (double)h[5]));
                fprintf(stderr, "\n");
                fprintf(stderr, "fsk prof   raw kcycles:"); for (int k = 0; k < 32; k++) if (k != 15 && k != 16 && k != 17) fprintf(stderr, " [%d] %.0f", k, h[k] / 1e3);
                fprintf(stderr, "  slots %llu frames %llu\n", h[16], h[17]);
            } else if (h[15]) {   // the pipelined kernel: cycles of channel 0's waves, all of it and what they spent waiting
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
    void *ptrs[] = { f->d_in, f->d_hann, f->d_fmask, f->d_Sf, f->d_sd, f->d_tw, f->d_dpeak, f->d_dmask, f->d_phift, f->d_tail, f->d_chan, f->d_recs, f->d_eye, f->d_hb, f->d_perm, f->d_iperm, f->d_wr, f->d_Sf_bak, f->d_tail_bak, f->d_chlist, f->d_sd_alt, f->d_scratch };
    for (void *p : ptrs) if (p) hipFree(p);
    delete f;
}

int sonde_fsk_info(const sonde_fsk_t *f, sonde_fsk_info_t *info) {
    if (!f || !info) return SONDE_E_ARG;
    *info = f->info;
    return 0;
}

// enqueue: the channels' records back to the host behind the launch (the soft decisions, hard bits and frame records stay on the device until somebody asks)
static int collect_enqueue(sonde_fsk_t *f) {
    const int C = f->cfg.n_channels;
    HIPCHK(hipMemcpyAsync(f->h_chan.data(), f->d_chan, (size_t)C * sizeof(FskChan), hipMemcpyDeviceToHost, f->stream));
    f->sd_on_host = false;                                    // the soft decisions follow when somebody asks for them (sonde_fsk_fetch): a consumer on the device (sonde_softin_dev.h) never does
    f->hb_on_host = false;                                    // the hard bits follow when somebody asks for them (sonde_fsk_fetch_bits): auto_rx's pipelines read the soft decisions
    f->recs_on_host = false;                                  // the frame records (timing, Eb/N0, tone estimates per frame) too: up to 40 bytes x 200 frames x channels a second
    return 0;
}
static int launch_enqueue(sonde_fsk_t *f) {
    const int C = f->cfg.n_channels;
    FskArgs &a = f->args;
    // A launch whose waves run out of slots — a bug, or a device slowed to a crawl under a profiler — ends that channel's launch with frames = -1.  Such
    // channels are repeated with the frame-at-a-time kernel (same arithmetic, no waits between waves) from the state they had before the launch: the channel
    // records are still on the host, Sf and the tone tails are copied aside first (two small device copies).
    const int Ndft = f->info.Ndft;
    if (!f->d_Sf_bak) {
        if (dalloc(&f->d_Sf_bak, (size_t)C * Ndft, false) || dalloc(&f->d_tail_bak, (size_t)C * a.M * a.NT, false) || dalloc(&f->d_chlist, (size_t)C, false)) return SONDE_E_NOMEM;
    }
    if (!f->d_scratch) {                                       // frames longer than a CU's LDS holds: k_fsk_demod's regions in global memory (sonde_fsk.hip)
        const long long nf = sonde_fsk_scratch_floats(&a);
        if (nf > 0) { if (dalloc(&f->d_scratch, (size_t)C * (size_t)nf, false)) return SONDE_E_NOMEM; a.scratch = f->d_scratch; a.scratch_stride = nf; }
    }
    if (!f->ev0) { HIPCHK(hipEventCreate(&f->ev0)); HIPCHK(hipEventCreate(&f->ev1)); }
    f->h_chan_prev.assign(f->h_chan.begin(), f->h_chan.end());
    // (the wave form of the modem keeps these two copies itself as it loads the state; the older kernels — SONDE_FSK_KERNEL, an A/B aid — get them from here)
    // (asked of the launcher itself: it also falls back to them where the wave form does not fit — Ndft < 64 or > 256, Ts % P != 0, more LDS than a CU has)
    a.ch_list = nullptr; a.force_demod = 0;
    const bool old_kernel = !sonde_fsk_wave_selected(&a);
    if (old_kernel) {
        HIPCHK(hipMemcpyAsync(f->d_Sf_bak, f->d_Sf, (size_t)C * Ndft * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
        HIPCHK(hipMemcpyAsync(f->d_tail_bak, f->d_tail, (size_t)C * a.M * a.NT * sizeof(float2), hipMemcpyDeviceToDevice, f->stream));
        a.Sf_bak = nullptr; a.tail_bak = nullptr;
    } else { a.Sf_bak = f->d_Sf_bak; a.tail_bak = f->d_tail_bak; }
    { const char *t = getenv("SONDE_FSK_TEST_ABORT"); a.test_abort_ch = t ? atoi(t) : -1; }
    a.ch_list = nullptr; a.force_demod = 0;
    std::swap(f->d_sd, f->d_sd_alt); a.sd = f->d_sd; f->sd_idx ^= 1;
    if (f->rd_set[f->sd_idx]) { HIPCHK(hipStreamWaitEvent(f->stream, f->ev_rd[f->sd_idx], 0)); f->rd_set[f->sd_idx] = false; }     // a consumer of the launch before last may still be reading this buffer
    hipEventRecord(f->ev0, f->stream);
    static const bool want_prof = getenv("SONDE_FSK_PROF") != nullptr;            // profiling aid: cycles per phase of channel 0, printed when the modem is destroyed
    if (want_prof && !f->d_prof) { if (hipMalloc((void **)&f->d_prof, 32 * sizeof(unsigned long long)) == hipSuccess) hipMemset(f->d_prof, 0, 32 * sizeof(unsigned long long)); }
    a.prof = f->d_prof;
    const int lrc = sonde_launch_fsk(&a, f->stream);
    hipEventRecord(f->ev1, f->stream);
    if (lrc < 0) return lrc == -1 ? SONDE_E_ARG : SONDE_E_NOGPU;
    f->pending = true;
    return collect_enqueue(f);
}
// the other half: wait for the launch, repeat the channels it gave up on
static int launch_wait(sonde_fsk_t *f) {
    if (!f->pending) return 0;
    const int C = f->cfg.n_channels;
    FskArgs &a = f->args;
    const int Ndft = f->info.Ndft;
    f->pending = false;
    HIPCHK(hipStreamSynchronize(f->stream));
    float ms = 0; if (hipEventElapsedTime(&ms, f->ev0, f->ev1) == hipSuccess) { f->ms += ms; f->launches++; }
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
    b.ch_list = f->d_chlist; b.n_ch = (int)bad.size(); b.force_demod = 1; b.test_abort_ch = -1; b.prof = nullptr; b.Sf_bak = nullptr; b.tail_bak = nullptr;
    const int lrc2 = sonde_launch_fsk(&b, f->stream);
    if (lrc2 < 0) return lrc2 == -1 ? SONDE_E_ARG : SONDE_E_NOGPU;
    { const int rc = collect_enqueue(f); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(f->stream));
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

static int submit(sonde_fsk_t *f, const void *src, int64_t ch_stride, int32_t n, hipMemcpyKind kind) {
    const int C = f->cfg.n_channels;
    if (f->pending) { const int rc = launch_wait(f); if (rc) return rc; }
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
    return launch_enqueue(f);
}
static int run(sonde_fsk_t *f, const void *src, int64_t ch_stride, int32_t n, hipMemcpyKind kind) {
    const int rc = submit(f, src, ch_stride, n, kind);
    return rc ? rc : launch_wait(f);
}

int sonde_fsk_process_host(sonde_fsk_t *f, const void *h_in, int64_t ch_stride, int32_t n_samples) {
    if (!f || !h_in) return SONDE_E_ARG;
    return run(f, h_in, ch_stride, n_samples, hipMemcpyHostToDevice);
}
int sonde_fsk_process_device(sonde_fsk_t *f, const void *d_in, int64_t ch_stride, int32_t n_samples) {
    if (!f || !d_in) return SONDE_E_ARG;
    return run(f, d_in, ch_stride, n_samples, hipMemcpyDeviceToDevice);
}
// the same in two halves: everything of sonde_fsk_process_device is enqueued on the engine's stream and the call returns; sonde_fsk_wait blocks until it is
// through (and repeats what has to be repeated).  Between the two the host is free — e.g. to submit the other engines of a mixed batch, or to run a consumer on the
// device over the launch BEFORE this one on its own stream (sonde_softin_dev_submit_fsk: the soft decisions of the last two launches are kept, d_sd / d_sd_alt).
// Every other call of this engine waits first.
int sonde_fsk_submit_device(sonde_fsk_t *f, const void *d_in, int64_t ch_stride, int32_t n_samples) {
    if (!f || !d_in) return SONDE_E_ARG;
    return submit(f, d_in, ch_stride, n_samples, hipMemcpyDeviceToDevice);
}
int sonde_fsk_wait(sonde_fsk_t *f) {
    if (!f) return SONDE_E_ARG;
    return launch_wait(f);
}
// (for sonde_softin_dev: frames per channel of the last launch, from the host's copy of the channel records — waits for a launch in flight)
int sonde_fsk_host_frames(sonde_fsk_t *f, int32_t *out) {
    if (!f || !out) return SONDE_E_ARG;
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
    for (int c = 0; c < f->cfg.n_channels; c++) out[c] = f->h_chan[c].frames;
    return 0;
}

int sonde_fsk_process_host_var(sonde_fsk_t *f, const void *const *h_in, const int32_t *n_samples) {
    if (!f || !h_in || !n_samples) return SONDE_E_ARG;
    if (f->pending) { const int rc = launch_wait(f); if (rc) return rc; }
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
    { const int rc = launch_enqueue(f); return rc ? rc : launch_wait(f); }
}

int sonde_fsk_reset_channel(sonde_fsk_t *f, int32_t channel) {
    if (!f || channel < 0 || channel >= f->cfg.n_channels) return SONDE_E_ARG;
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
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
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
    if (!f->sd_on_host) {
        HIPCHK(hipMemcpyAsync(f->h_sd.data(), f->d_sd, f->h_sd.size() * sizeof(float), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        f->sd_on_host = true;
    }
    if (frames && !f->recs_on_host) {
        HIPCHK(hipMemcpyAsync(f->h_recs.data(), f->d_recs, f->h_recs.size() * sizeof(FskFrameRec), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        f->recs_on_host = true;
    }
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
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
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
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
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
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
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
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
    const int C = f->cfg.n_channels;
    HIPCHK(hipStreamSynchronize(f->stream));
    HIPCHK(hipMemset(f->d_Sf, 0, (size_t)C * f->info.Ndft * sizeof(float)));
    HIPCHK(hipMemcpy(f->h_chan.data(), f->d_chan, (size_t)C * sizeof(FskChan), hipMemcpyDeviceToHost));
    for (auto &c : f->h_chan) c.nin = f->info.N;
    HIPCHK(hipMemcpy(f->d_chan, f->h_chan.data(), (size_t)C * sizeof(FskChan), hipMemcpyHostToDevice));
    return 0;
}

// what a consumer on the device needs of the last launch (sonde_softin_dev.hip): the soft decisions where they lie, the per-channel frame counts, the stream they were made on
int sonde_fsk_dev_view(sonde_fsk_t *f, const float **d_sd, long long *sd_cap, const FskChan **d_chan, int *bits_per_frame, int *n_ch, hipStream_t *stream) {
    if (!f) return SONDE_E_ARG;
    *d_sd = f->d_sd; *sd_cap = f->args.sd_cap; *d_chan = f->d_chan; *bits_per_frame = f->info.Nbits; *n_ch = f->cfg.n_channels; *stream = f->stream;
    return 0;
}

// the same for the launch BEFORE the one in flight (sonde_softin_dev_submit_fsk_behind): its soft decisions sit in the other buffer, its frame counts are the host's copy
// from before the launch in flight was submitted; 1 = there is such a pair (a launch in flight and one before it), 0 = no launch in flight (use sonde_fsk_dev_view)
int sonde_fsk_dev_view_prev(sonde_fsk_t *f, const float **d_sd, long long *sd_cap, int *bits_per_frame, int *n_ch, int32_t *frames_out) {
    if (!f) return SONDE_E_ARG;
    if (!f->pending || f->h_chan_prev.size() != (size_t)f->cfg.n_channels) return 0;
    *d_sd = f->d_sd_alt; *sd_cap = f->args.sd_cap; *bits_per_frame = f->info.Nbits; *n_ch = f->cfg.n_channels;
    for (int c = 0; c < f->cfg.n_channels; c++) frames_out[c] = f->h_chan_prev[(size_t)c].frames;
    return 1;
}
int sonde_fsk_dev_reader_done_prev(sonde_fsk_t *f, hipStream_t consumer_stream) {
    if (!f) return SONDE_E_ARG;
    const int i = f->sd_idx ^ 1;
    if (!f->ev_rd[i]) HIPCHK(hipEventCreateWithFlags(&f->ev_rd[i], hipEventDisableTiming));
    HIPCHK(hipEventRecord(f->ev_rd[i], consumer_stream));
    f->rd_set[i] = true;
    return 0;
}

// a consumer that has enqueued its reads of the buffer sonde_fsk_dev_view showed (the last launch's soft decisions) on `consumer_stream` says so: the launch that
// will overwrite that buffer — the next but one — waits for this point of the consumer's stream
int sonde_fsk_dev_reader_done(sonde_fsk_t *f, hipStream_t consumer_stream) {
    if (!f) return SONDE_E_ARG;
    const int i = f->sd_idx;
    if (!f->ev_rd[i]) HIPCHK(hipEventCreateWithFlags(&f->ev_rd[i], hipEventDisableTiming));
    HIPCHK(hipEventRecord(f->ev_rd[i], consumer_stream));
    f->rd_set[i] = true;
    return 0;
}

int sonde_fsk_kernel_ms(sonde_fsk_t *f, double *avg_ms, int64_t *launches) {
    if (!f) return SONDE_E_ARG;
    if (f->pending) { const int rc_ = launch_wait(f); if (rc_) return rc_; }
    if (avg_ms) *avg_ms = f->launches ? f->ms / (double)f->launches : 0.0;
    if (launches) *launches = f->launches;
    return 0;
}

}  // extern "C"
