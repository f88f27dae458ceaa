// fsk_wave_emu.cpp — test infrastructure: the wave form of the 2-/4-FSK modem (radiosonde_auto_rx_amd/csrc/sonde_fsk_wave.h, the source hipcc
// compiles into k_fsk_wave) compiled for the CPU under wave_emu.h, with the host engine's part — the per-channel sample ring, the state that
// survives a launch (FskChan, Sf, the f_dc tail), the collection of soft decisions and frame records — restated around it for ONE channel.
//   emu_fsk_run(...)  feeds a capture in calls of `chunk` samples, one emulated launch per call (64 threads, or 128 for the walker + worker form)
// Not used by the product.  Built by tests/test_fsk_wave_emu.py:  g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC
#define SONDE_FSK_EMU 1
#include "wave_emu.h"
#include "../../radiosonde_auto_rx_amd/csrc/sonde_fsk_tables.h"
#include "../../radiosonde_auto_rx_amd/csrc/sonde_fsk_wave.h"
#include <cstring>
#include <vector>

struct EmuRec { int nin, nin_next; float f_est[4]; float norm_rx_timing, ppm, EbNodB, snr_est; };

template <int M, int LOG2N>
static int run_launch(const FskArgs &a, bool split, std::vector<float> &lds, FwCtl &ctl) {
    const int nthreads = split ? (a.fin ? 256 : 192) : 64;
    emu::run_workgroup(nthreads, [&](int tid) {
        if (split) fsk_wave_channel<M, LOG2N, true, 0>(a, 0, tid, lds.data(), ctl);
        else       fsk_wave_channel<M, LOG2N, false, 0>(a, 0, tid, lds.data(), ctl);
    });
    return 0;
}

extern "C" int emu_fsk_run(int Fs, int Rs, int M, int P, int nsym, int format, int lower, int upper, int mask, int tone_spacing, int burst, int split, int fin,
                           const void *samples, int n_samples, int chunk, float *sd_out, int sd_max, EmuRec *recs_out, int rec_max, float *Sf_out,
                           long long *samples_out) {
    sonde_fsk_cfg_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION; cfg.n_channels = 1; cfg.Fs = Fs; cfg.Rs = Rs; cfg.M = M; cfg.P = P; cfg.nsym = nsym; cfg.format = format;
    cfg.fsk_lower = lower; cfg.fsk_upper = upper; cfg.mask = mask ? 1 : 0; cfg.tone_spacing = tone_spacing; cfg.max_chunk = chunk; cfg.burst_mode = burst;
    FskTables T;
    if (fsk_build_tables(cfg, T)) return -1;
    if (T.Ndft != 64 && T.Ndft != 128 && T.Ndft != 256) return -2;
    FskArgs a; memset(&a, 0, sizeof a);
    fsk_tables_to_args(cfg, T, a);
    a.n_ch = 1;
    const int N = T.N, Ts = T.Ts;
    const int max_frames = chunk / std::max(1, N - Ts / 2) + 2;
    a.rec_cap = max_frames; a.sd_cap = max_frames * nsym * (M / 2);
    uint32_t ring = 1; while (ring < (uint32_t)(chunk + N + Ts + 16)) ring <<= 1;
    a.ring = ring;
    const size_t unit = format == SONDE_FSK_CF32 ? 8 : format == SONDE_FSK_CS16 ? 4 : 2;
    std::vector<char> in(ring * unit, 0);
    std::vector<float> Sf(T.Ndft, 0.f), sd(a.sd_cap), eye(8 * 160, 0.f);
    std::vector<uint8_t> hb(a.sd_cap);
    std::vector<float2> tail((size_t)M * T.NT, make_float2(0.f, 0.f));
    std::vector<FskFrameRec> recs(a.rec_cap);
    FskChan chan; memset(&chan, 0, sizeof chan);
    for (int m = 0; m < 4; m++) chan.phi_c[m] = fsk_exp_j(0);
    chan.nin = N;
    a.in = in.data(); a.hann = T.hann.data(); a.tw = T.tw.data(); a.perm = T.perm.data(); a.iperm = T.iperm.data(); a.dphi_peak = T.dpeak.data();
    a.dphi_mask = T.dmask.data(); a.f_mask = T.fmask.data(); a.phi_ft = T.phift.data(); a.chan = &chan; a.Sf = Sf.data(); a.eye = eye.data();
    a.tail = tail.data(); a.sd = sd.data(); a.hb = hb.data(); a.recs = recs.data();
    a.R = fw_ring_len(T.NT, T.Ts / P); a.test_abort_ch = -1; a.fin = fin;
    { const char *e = getenv("EMU_ROLE_ROT"); a.role_rot = e ? atoi(e) : 0; }       // (which wavefront plays which role, sonde_fsk_wave.h)
    std::vector<float> lds(fw_lds_floats(M, nsym, P, a.R, T.Ndft, fin) + 64, 0.f);
    FwCtl ctl; memset(&ctl, 0, sizeof ctl);
    uint32_t wr = 0;
    int n_sd = 0, n_rec = 0;
    const char *src = reinterpret_cast<const char *>(samples);
    for (int s0 = 0; s0 < n_samples; s0 += chunk) {
        const int n = std::min(chunk, n_samples - s0);
        for (int i = 0; i < n; i++) memcpy(&in[(size_t)((wr + (uint32_t)i) & (ring - 1)) * unit], src + (size_t)(s0 + i) * unit, unit);
        wr += (uint32_t)n;
        a.wr = wr;
        if (M == 2) {
            if (T.Ndft == 64) run_launch<2, 6>(a, split, lds, ctl); else if (T.Ndft == 128) run_launch<2, 7>(a, split, lds, ctl); else run_launch<2, 8>(a, split, lds, ctl);
        } else {
            if (T.Ndft == 64) run_launch<4, 6>(a, split, lds, ctl); else if (T.Ndft == 128) run_launch<4, 7>(a, split, lds, ctl); else run_launch<4, 8>(a, split, lds, ctl);
        }
        if (chan.frames < 0) return -3;
        const int nb = nsym * (M / 2);
        for (int f = 0; f < chan.frames; f++) {
            if (n_sd + nb <= sd_max) memcpy(sd_out + n_sd, &sd[(size_t)f * nb], nb * sizeof(float));
            n_sd += nb;
            if (n_rec < rec_max) { const FskFrameRec &r = recs[f]; EmuRec &o = recs_out[n_rec]; o.nin = r.nin; o.nin_next = r.nin_next; for (int m = 0; m < 4; m++) o.f_est[m] = r.f_est[m];
                                   o.norm_rx_timing = r.norm_rx_timing; o.ppm = r.ppm; o.EbNodB = r.EbNodB; o.snr_est = r.snr_est; }
            n_rec++;
        }
    }
    if (Sf_out) memcpy(Sf_out, Sf.data(), T.Ndft * sizeof(float));
    if (samples_out) *samples_out = chan.samples;
    return n_rec;
}
