// sonde_scan.cpp — host side of the batched scanner behind include/sonde_scan.h (the reference's scan/dft_detect.c).
//
// Per process call, on one HIP stream:
//   front-end (k_mix_decimate per IQ-DC segment | k_iq_convert | k_audio_convert)  ->  k_scan_if (4 FM streams)
//   ->  k_scan_corr over the list of correlation windows that became complete  ->  decision logic on the host.
// Sequential state of the reference and where it lives here:
//   IQ-DC mean, fixed window sr_if/32 [* decM]  (dft_detect.c:1152-1156)   host schedule + k_dc_update
//   FIR histories, z0 of the discriminators                                  IF-rate rings in HBM (absolute indices)
//   k / sample_in / sample_out window counter (main, :1483-1505)             next_sin per channel
//   mv[], mv_pos[], mv0_pos[], mv_max, j_max, rs_detect2[], mutable type/tn   Chan (host)
// N_DFT is 8192 up to an IF rate of ~51 kHz; above it (--IQ with --bw > 48, --iq input at 96 / 192 kHz) 16384 / 32768: those windows skip the
// prefilter and every pair runs the reference's transform network on an array in global memory (k_scan_corr_t<true>) — rare, exactness over speed.
#include "../../include/sonde_scan.h"
#include "sonde_dev.h"
#include "sonde_host.h"
#include "sonde_scan_dev.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace sonde;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

namespace {

struct TplDef { int baud; const char *hdr; float bt; float thres; int herrs; const char *type; int tn; int lpfm; int lpiq; };

// rs_hdr[] of the reference (dft_detect.c:172-191); the header strings are the on-air sync patterns (:43-134)
const int kNrs = 18, kIdxImetAfsk = 15;      // rows 16/17 (IMET1RS, IMET4) are only reached from the IMETafsk post-processing
const TplDef kTpl[kNrs] = {
    { 2500, "10011010100110010101101001010101", 1.0f, 0.65f, 2, "DFM9", 2, 0, 1 },
    { 4800, "0000100001101101010100111000100001000100011010010100100000011111", 0.5f, 0.70f, 2, "RS41", 3, 0, 1 },
    { 4800, "10100110011001101001" "10100110011001101001" "1010011001100110100110101010100110101001", 0.5f, 0.70f, 3, "RS92", 4, 0, 1 },
    { 4800, "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110", 1.0f, 0.60f, 8, "LMS6", 8, 0, 1 },
    { 4800, "0000000001" "0101010101" "0001001001" "0001001001", 0.5f, 0.80f, 2, "IMET5", 24, 0, 1 },
    { 9616, "0010100111" "0010100111" "0001001001" "0010010101", 1.0f, 0.70f, 2, "MK2LMS", 18, 1, 2 },
    { 9608, "1001100110010100110010011001" "1010", 1.0f, 0.76f, 2, "M10", 5, 1, 2 },
    { 2400, "110011001101001101001101010100101010110010101010", 1.0f, 0.70f, 2, "MEISEI", 9, 0, 2 },
    { 4800, "10100110010110101001" "10010101011010010101" "10101001010101010101" "10011001010110101001", 1.0f, 0.70f, 2, "RD94RD41", 10, 0, 1 },
    { 2400, "1001100110011001" "1001101010101010", 1.5f, 0.80f, 2, "MRZ", 12, 0, 1 },
    { 1200, "10101010" "10101010" "10110100" "00101011", 1.0f, 0.65f, 2, "MTS01", 13, 0, 0 },
    { 5800, "01010101010101010101010101010101", 1.5f, 0.80f, 2, "C34C50", 15, 0, 2 },
    { 4800, "10101010" "10101010" "10101010" "00101101" "11010100", 1.0f, 0.65f, 2, "WXR301", 16, 0, 3 },
    { 5000, "10101010" "10101010" "10101010" "11000001" "10010100", 1.0f, 0.65f, 2, "WXRPN9", 17, 0, 3 },
    { 9600, "0000" "11110000111100001111000011110000" "1111" "0000" "10101100110010101100101010101100" "1111", 1.0f, 0.80f, 2, "IMET1AB", 29, 1, 3 },
    { 9600, "11110000111100001111000011110000" "11110000111100001111000011110000", 0.5f, 0.80f, 4, "IMETafsk", 25, 1, 1 },
    { 9600, "0000" "1111" "0000" "1111" "0000" "1111" "0000" "1111", 0.5f, 0.80f, 2, "IMET1RS", 28, 0, 3 },
    { 9600, "0000" "1111" "0000" "1111" "0000" "1111" "0000" "1111", 0.5f, 0.80f, 2, "IMET4", 26, 1, 1 },
};
const uint32_t kDefaultDisable = (1u << 11) | (1u << 14);       // -DNOC34C50 -DNOIMET1AB (scan/Makefile:1)

// The reference's own transform (dft_raw, dft_detect.c:285-322): radix-2 decimation in time with the stage twiddle
// advanced by a float recurrence w1 *= cexp(-i pi/2^s).  Its drift (up to ~2e-4 in the last stage) is part of every
// score the reference prints, so the template / low-pass spectra and the device kernels use the very same table.
std::vector<float2> ref_twiddles(int log2n = SC_LOG2N) {
    std::vector<float2> tws(((size_t)1 << log2n) - 1);
    for (int s = 0; s < log2n; s++) {
        const int l2 = 1 << s;
        const std::complex<double> e = std::exp(std::complex<double>(0.0, -M_PI / (double)(float)l2));
        const float w2r = (float)e.real(), w2i = (float)e.imag();
        float w1r = 1.0f, w1i = 0.0f;
        for (int j = 0; j < l2; j++) {
            tws[(size_t)l2 - 1 + j] = make_float2(w1r, w1i);
            const float nr = w1r * w2r - w1i * w2i, ni = w1r * w2i + w1i * w2r;
            w1r = nr; w1i = ni;
        }
    }
    return tws;
}

void dft_ref_host(std::vector<float2> &z, const std::vector<float2> &tws) {
    const int n = (int)z.size();                 // a power of two; tws = ref_twiddles(log2 n)
    int log2n = 0; while ((1 << log2n) < n) log2n++;
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(z[i], z[j]);
    }
    for (int s = 0; s < log2n; s++) {
        const int l2 = 1 << s, l = l2 << 1;
        for (int j = 0; j < l2; j++) {
            const float2 w = tws[(size_t)l2 - 1 + j];
            for (int i = j; i < n; i += l) {
                const int k = i + l2;
                const float tr = z[k].x * w.x - z[k].y * w.y, ti = z[k].x * w.y + z[k].y * w.x;
                z[k] = make_float2(z[i].x - tr, z[i].y - ti);
                z[i] = make_float2(z[i].x + tr, z[i].y + ti);
            }
        }
    }
}

double gq(double x) { return 0.5 - 0.5 * std::erf(x / 1.4142135624); }
double gpulse(double t, double sigma) { return gq((t - 0.5) / sigma) - gq((t + 0.5) / sigma); }

// header template of dft_detect.c:1227-1258 (double t, float storage, 2-norm).  hLen is the *longest* header of all
// active templates there (:1166-1173), so a shorter header sees its string terminator as a following 0 bit.
std::vector<float> scan_match(const char *bits, int hLen, float spb, float bt, int L) {
    std::vector<float> m(L);
    const double sigma = std::sqrt(std::log(2)) / (2 * M_PI * bt);
    for (int i = 0; i < L; i++) {
        const int pos = (int)(i / spb);
        const double t = (i - pos * spb) / spb - 0.5;
        const double b1 = ((bits[pos] & 1) - 0.5) * 2.0;
        double b = b1 * gpulse(t, sigma);
        if (pos > 0) b += ((bits[pos - 1] & 1) - 0.5) * 2.0 * gpulse(t + 1, sigma);
        if (pos < hLen - 1) b += ((bits[pos + 1] & 1) - 0.5) * 2.0 * gpulse(t - 1, sigma);
        m[i] = (float)b;
    }
    double n2 = 0; for (int i = 0; i < L; i++) { const double x = m[i]; n2 += x * x; }
    const float nm = (float)std::sqrt(n2);
    for (int i = 0; i < L; i++) m[i] /= nm;
    return m;
}

// read_bufbit's float bit clock (dft_detect.c:821-864): rbitgrenze += spb; do { ...; rcount++ } while (rcount < rbitgrenze)
void bit_boundaries(float spb, int nhalf, std::vector<int> &out) {
    unsigned rcount = 0; float grenze = 0.f;
    for (int k = 0; k < nhalf; k++) {
        grenze += spb;
        do { rcount++; } while ((float)rcount < grenze);
        out.push_back((int)rcount);
    }
}

struct Chan {
    float mv[kNrs]; uint32_t mv_pos[kNrs], mv0_pos[kNrs]; int mp[kNrs]; float dc[kNrs], df[kNrs];
    const char *type[kNrs]; int tn[kNrs]; int detect2[kNrs];
    int j_max = 0; float mv_max = 0.f; int d2_tn = kNrs; bool done = false;
    uint32_t next_sin = 0;
    // prefilter bookkeeping: the most recent decided window and which of its templates were evaluated by the exact kernel (a candidate in the
    // NEXT window needs this one's exact peak position for `mv_pos > mv0_pos`, dft_detect.c:1521)
    bool last_valid = false; uint32_t last_pos = 0; uint32_t last_exact = 0;
    // IMET AFSK post-processing in progress (dft_detect.c:1533-1607): the window's decision waits for one more second of the FM stream
    bool imet_hold = false; uint32_t imet_sin = 0; int imet_hf = 0;
};

struct KStat { double ms = 0; int64_t n = 0; };

}  // namespace

// the same transform for the demodulators' header search (sonde_engine.cpp: match spectrum Fm, twiddle table), plain float pairs
namespace sonde {
std::vector<float> ref_twiddle_table() {
    const std::vector<float2> t = ref_twiddles();
    std::vector<float> o(2 * t.size());
    for (size_t k = 0; k < t.size(); k++) { o[2 * k] = t[k].x; o[2 * k + 1] = t[k].y; }
    return o;
}
void ref_dft_8192(std::vector<float> &re_im) {
    static const std::vector<float2> tws = ref_twiddles();
    std::vector<float2> z(SC_N);
    for (int k = 0; k < SC_N; k++) z[k] = make_float2(re_im[2 * k], re_im[2 * k + 1]);
    dft_ref_host(z, tws);
    for (int k = 0; k < SC_N; k++) { re_im[2 * k] = z[k].x; re_im[2 * k + 1] = z[k].y; }
}
}  // namespace sonde

struct sonde_scan {
    sonde_scan_cfg_t cfg{};
    sonde_scan_info_t info{};
    hipStream_t stream = nullptr;
    hipEvent_t ev_fe[3] = { nullptr, nullptr, nullptr }; bool fe_pending = false, fe_timed = false;      // front end / scan_if timing of the call in flight (read at its first host wait)
    // one-pass front end (k_mix_decimate50r): decided when the scanner is made — the P tail between calls holds raw sums in this form
    bool front_raw = false; float2 *d_etab64 = nullptr; int etab_len = 0; int2 *d_bsum = nullptr; long long bsum_stride = 0; float2 *d_corr = nullptr, *d_dcprev[2] = { nullptr, nullptr }, *d_hist[2] = { nullptr, nullptr }; int hist_cur = 0; bool fold_pending = false; ScanFold fold{};
    hipEvent_t ev_rw[3] = { nullptr, nullptr, nullptr };                  // run_windows' timing events
    hipEvent_t ev_wait = nullptr;                                       // sonde_scan_wait_stream
    // design
    Decimator dec; int Q = 0; int lut_len = 1; int DS = 0; float *d_wtab = nullptr;
    std::vector<float> wtab;
    ScanTpl tpl[SC_NTPL]; float thres[kNrs]; uint32_t disabled = 0;
    int K = 0, delay = 0, nstreams = 0, nfilt = 0, filt_stream[3] = {0, 0, 0}, raw_stream = 0, lpiq_taps = 0, lpfm_taps = 0;
    int N = SC_N, log2n = SC_LOG2N;                // N_DFT (dft_detect.c:1196-1202): 8192, or 16384 / 32768 for IF rates above ~51 kHz
    float2 *d_scratch = nullptr; int scratch_pairs = 0;       // N_DFT > 8192: transform arrays of the exact kernel in global memory
    bool wide_fe = false; float2 *d_f32in = nullptr;           // decimator longer than 8 blocks (wide IF): int16 input goes through the float32 mixer / FIR kernels
    // device
    double *d_chanf0 = nullptr; float2 *d_dcavg = nullptr; long long *d_dcsums = nullptr; float2 *d_ptail[2] = {nullptr, nullptr}; int ptail_cur = 0;
    float2 *d_y = nullptr; float *d_fm = nullptr; float *d_wiq = nullptr; float2 *d_G = nullptr, *d_tw = nullptr, *d_WS = nullptr;
    uint8_t *d_hdr = nullptr; int *d_bnd = nullptr; ScanItem *d_items = nullptr; ScanRes *d_res = nullptr;
    ScanItem *h_items = nullptr; ScanRes *h_res = nullptr; int item_cap = 0;
    // prefilter (k_scan_pre): A fragments of the templates / FM low-passes, per-pair results, work list of the exact kernel
    uint16_t *d_amatch = nullptr, *d_aws = nullptr; float *d_wstail = nullptr; int a_off[SC_NTPL] = {0}, nc2[SC_NTPL] = {0}, nc1 = 0, ws_pad = 0;
    float kap[SC_NTPL] = {0};
    ScanPre *d_pre = nullptr, *h_pre = nullptr; ScanWork *d_work = nullptr, *h_work = nullptr; bool use_pre = false;
    void *d_stage = nullptr; size_t stage_bytes = 0;
    int16_t *d_conv = nullptr;                     // cu8 input: converted int16 copy
    double *d_dcsums_f = nullptr; float2 *d_zring = nullptr; float *d_taps_f = nullptr; uint32_t zmask = 0;   // float32 input (MixF32Args)
    int ring_len = 0;
    // stream position
    uint64_t samples_in = 0; uint32_t m_out = 0; uint32_t dc_cnt = 0, dc_max = 0;
    unsigned long long *d_spprof = nullptr;
    long long *d_segsums = nullptr; float2 *d_dcseg = nullptr;      // IQ-DC windows of a call: sums, table of means (MixDecArgs.dc_seg)
    std::vector<Chan> chan;
    std::vector<sonde_detection_t> queue;
    std::vector<sonde_scan_window_t> last_windows;
    std::map<std::string, KStat> stats;
};

static void timed(sonde_scan *s, const char *name, hipEvent_t a, hipEvent_t b) {
    float ms = 0; if (hipEventElapsedTime(&ms, a, b) == hipSuccess) { auto &k = s->stats[name]; k.ms += ms; k.n += 1; }
}

// the events around the front end and k_scan_if are read at the call's first host wait (behind the prefilter): the call does not stop for them
static void flush_front_timing(sonde_scan *s) {
    if (!s->fe_pending) return;
    s->fe_pending = false;
    if (s->fe_timed) { timed(s, "front_end", s->ev_fe[0], s->ev_fe[1]); timed(s, "scan_if", s->ev_fe[1], s->ev_fe[2]); }
}

template <class T> static int dalloc(T **p, size_t n, bool zero = true) {
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    if (zero) HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
    return 0;
}
template <class T> static int dupload(T **p, const std::vector<T> &v) {
    if (dalloc(p, v.size() ? v.size() : 1, false)) return SONDE_E_NOMEM;
    if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// ---- prefilter tables (k_scan_pre, sonde_scan_pre.hip)
static uint16_t f16_bits(float x) { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, sizeof u); return u; }
// A fragments of the Toeplitz product out[i] = sum_u h[u] x[i+u].  With i = 16 a + b and u = 32 c + e - b (e < 32, so e - b in (-16, 32)):
//     out[16 a + b] = sum_c sum_{e<32} A_c[b][e] x[16 a + 32 c + e],   A_c[b][e] = h[32 c + e - b]   (0 outside [0, U))
// For a fixed row b a step covers 32 consecutive taps and the next step the next 32: every tap in exactly one step and no structural zeros
// in a fragment (round 3 advanced by 16 taps per step with e - b restricted to [0, 16): half of every fragment was zero, twice the MFMAs and
// twice the LDS reads).  Fragment element: step c, lane (b = lane & 15, g = lane >> 4), r < 8  <-  A_c[b][8 g + r].
static int toeplitz_frags(const std::vector<float> &h, std::vector<uint16_t> &out) {
    const int U = (int)h.size(), nc = ((U + 15 + 31) / 32 + 1) & ~1;   // whole blocks of 2 steps (k_scan_pre prefetches the fragments block-wise); the padding is zeros
    for (int c = 0; c < nc; c++)
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 8; r++) {
                const int d = 8 * (lane >> 4) + r - (lane & 15), u = 32 * c + d;          // d in (-16, 32): every element of the fragment is a tap
                out.push_back((u >= 0 && u < U) ? f16_bits(h[u]) : (uint16_t)0);
            }
    return nc;
}
// candidates: smax > thres - margin, where smax already carries the part of the rounding bound that scales with the signal (ScanPreArgs.kap, added per position by the
// kernel) and the margin is what does not: 3 u + L 2^-24 + the reference's own distance from exact arithmetic (3e-4, its drifting twiddles) — DESIGN.md §4.6b,
// tests/test_scan_prefilter_model.py::test_prefilter_error_stays_inside_the_derived_bound.  (Rounds 3-5: a flat 0.03 on the bare score, argued not derived.)
static const float kPreMargin = 0.003f;

extern "C" {

int sonde_scan_create(const sonde_scan_cfg_t *cfg, const double *fq, sonde_scan_t **out) {
    if (!cfg || !out || cfg->abi_version != SONDE_ABI_VERSION) return SONDE_E_ARG;
    if (cfg->n_channels < 1 || cfg->sample_rate < 1 || (cfg->bits != 16 && cfg->bits != 8 && cfg->bits != 32) || cfg->max_chunk < 1) return SONDE_E_ARG;
    if (cfg->iq_mode != SONDE_SCAN_AUDIO && cfg->iq_mode != SONDE_SCAN_IFIQ && cfg->iq_mode != SONDE_SCAN_BBIQ) return SONDE_E_ARG;
    if (cfg->iq_mode == SONDE_SCAN_BBIQ && !fq) return SONDE_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device >= ndev) {
        fprintf(stderr, "libsonde_hip: no usable HIP device (the scanner has no CPU fallback)\n");
        return SONDE_E_NOGPU;
    }
    HIPCHK(hipSetDevice(cfg->device));
    sonde_scan *s = new sonde_scan();
    s->cfg = *cfg;
    const int C = cfg->n_channels;
    const bool iq = cfg->iq_mode != SONDE_SCAN_AUDIO;
    const float set_lpIQ = cfg->bw_khz < 1.0f ? 0.f : (float)((double)cfg->bw_khz * 1e3);
    s->disabled = cfg->disable_mask ? cfg->disable_mask : kDefaultDisable;

    // ---- init_buffers() (dft_detect.c:995-1285)
    int sr = cfg->sample_rate, D = 1;
    if (cfg->iq_mode == SONDE_SCAN_BBIQ) {
        s->dec = design_decimator_scan(cfg->sample_rate, cfg->opt_min != 0, set_lpIQ);     // dft_detect.c:1021-1067 (= demod_mod.c:1222-1259 up to 48 kHz)
        D = s->dec.decM; sr = s->dec.if_sr;
        if (D == 1) s->dec.taps.assign(1, 1.0f);
        const int T = (int)s->dec.taps.size();
        s->Q = (T + D - 1) / D;
        if (D > 1024) { delete s; return SONDE_E_ARG; }
        if (s->Q > 8) { s->wide_fe = true; s->Q = 8; }         // (wide IF: 267 taps over D = 25) -> the plain float32 mixer / FIR kernels, any tap count
        s->wtab.assign((size_t)std::max(64, D) * 8, 0.f);
        if (!s->wide_fe) {
            const int pad = s->Q * D - T;
            std::vector<float> wpad((size_t)s->Q * D, 0.f);
            for (int k = 0; k < T; k++) wpad[pad + k] = s->dec.taps[k];
            for (int r = 0; r < D; r++) for (int q = 0; q < s->Q; q++) s->wtab[(size_t)r * 8 + q] = wpad[(size_t)D * q + r];
        }
        if (D > 64 && !s->wide_fe) {                                         // wide decimation: taps in global memory, D walked in pieces of DS
            for (int k = 64; k >= 4; k--) if (D % k == 0) { s->DS = k; break; }
            if (!s->DS || s->Q < 5) { delete s; return SONDE_E_ARG; }
            if (dupload(&s->d_wtab, s->wtab)) { sonde_scan_destroy(s); return SONDE_E_NOMEM; }
        } else if (!s->wide_fe) {                                            // the generated D = 50 kernel reads the tap rows * 2^-15 behind the table
            std::vector<float> both(s->wtab);
            for (size_t i = 0; i < s->wtab.size(); i++) both.push_back(s->wtab[i] * 3.0517578125e-05f);
            if (dupload(&s->d_wtab, both)) { sonde_scan_destroy(s); return SONDE_E_NOMEM; }
        }
        std::vector<double> f0s(C);
        for (int c = 0; c < C; c++) {
            const Mixer m = design_mixer(-std::max(-0.5, std::min(0.5, fq[c])), cfg->sample_rate);
            f0s[c] = m.f0; s->lut_len = m.lut_len;
        }
        if (dupload(&s->d_chanf0, f0s)) { sonde_scan_destroy(s); return SONDE_E_NOMEM; }
    }
    std::vector<float> w_iq, w_lp;
    if (iq) {
        int taps = (int)(4 * sr / 2e3); if (taps % 2 == 0) taps++;                 // FM low-pass, 2 kHz transition (:1106-1117)
        float lpfm_bw[2] = { 4e3f, 10e3f };
        for (int j = 0; j < 2; j++) { const std::vector<float> w = design_lowpass(lpfm_bw[j] / (float)sr, taps); w_lp.insert(w_lp.end(), w.begin(), w.end()); s->lpfm_taps = (int)w.size(); }
        float lpiq_bw[3] = { 6e3f, 12e3f, 22e3f };
        if (cfg->opt_lband) { lpiq_bw[0] = 20e3f; lpiq_bw[1] = 32e3f; lpiq_bw[2] = 200e3f; }
        s->nfilt = 3;
        if (set_lpIQ > 100.0f) { lpiq_bw[0] = set_lpIQ; s->nfilt = 1; }             // --bw: option_singleLpIQ (:1121-1126)
        taps = (int)(4 * sr / 4e3); if (taps % 2 == 0) taps++;
        for (int b = 0; b < s->nfilt; b++) {
            const float f_lp = (float)(lpiq_bw[b] / (float)sr / 2.0);
            const std::vector<float> w = design_lowpass(f_lp, taps); w_iq.insert(w_iq.end(), w.begin(), w.end()); s->lpiq_taps = (int)w.size();
        }
        if (s->nfilt == 3) { s->filt_stream[0] = 0; s->filt_stream[1] = 1; s->filt_stream[2] = 2; s->raw_stream = 3; s->nstreams = 4; }
        else { s->filt_stream[0] = 0; s->raw_stream = 1; s->nstreams = 2; }
    } else s->nstreams = 1;
    auto phys_stream = [&](int lpiq) { if (!iq) return 0; if (lpiq == 3) return s->raw_stream; return s->nfilt == 3 ? lpiq : 0; };

    int Lmax = 0, hLenMax = 0;
    std::vector<int> Ls(kNrs); std::vector<float> spbs(kNrs);
    for (int j = 0; j < kNrs; j++) {
        spbs[j] = sr / (float)kTpl[j].baud;
        const int hLen = (int)strlen(kTpl[j].hdr);
        Ls[j] = (int)(hLen * spbs[j] + 0.5);
        if (!((s->disabled >> j) & 1u) && Ls[j] > Lmax) Lmax = Ls[j];
        if (!((s->disabled >> j) & 1u) && hLen > hLenMax) hLenMax = hLen;
        s->thres[j] = cfg->ths > 0.f ? cfg->ths : kTpl[j].thres;
    }
    const int L2 = 2 * Lmax;
    int p2 = 1; while (p2 < 3 * L2) p2 <<= 1; while (p2 < 0x2000) p2 <<= 1;
    if (p2 != 8192 && p2 != 16384 && p2 != 32768) { delete s; return SONDE_E_ARG; }
    s->N = p2; s->log2n = 13 + (p2 > 8192) + (p2 > 16384);
    const int N = s->N;
    s->K = N - L2; s->delay = L2 / 16;

    const std::vector<float2> tws = ref_twiddles(s->log2n);
    std::vector<float2> G((size_t)SC_NTPL * N, make_float2(0.f, 0.f));
    std::vector<uint8_t> hdrbits; std::vector<int> bnd;
    std::vector<std::vector<float2>> WS(2);
    if (iq) for (int j = 0; j < 2; j++) {                     // WS[j] = dft(FM low-pass taps) (dft_detect.c:1269-1278)
        WS[j].assign(N, make_float2(0.f, 0.f));
        for (int i = 0; i < s->lpfm_taps; i++) WS[j][i].x = w_lp[(size_t)j * s->lpfm_taps + i];
        dft_ref_host(WS[j], tws);
    }
    std::vector<uint16_t> a_match, a_ws;
    for (int j = 0; j < SC_NTPL; j++) {
        ScanTpl &t = s->tpl[j];
        memset(&t, 0, sizeof t);
        t.L = Ls[j]; t.hLen = (int)strlen(kTpl[j].hdr); t.lpfm = kTpl[j].lpfm; t.stream = phys_stream(kTpl[j].lpiq);
        t.active = !((s->disabled >> j) & 1u) && (s->K + t.L <= N);
        t.is_m10 = strncmp(kTpl[j].type, "M10", 3) == 0;
        t.spb = spbs[j]; t.thres = s->thres[j]; t.herrs = kTpl[j].herrs;
        t.hdr_off = (int)hdrbits.size(); hdrbits.insert(hdrbits.end(), kTpl[j].hdr, kTpl[j].hdr + t.hLen);
        t.bnd_off = (int)bnd.size(); bit_boundaries(t.spb, t.hLen, bnd);
        if (t.is_m10) bit_boundaries(t.spb, 28, bnd);
        s->info.L[j] = t.L;
        // Fm = dft of the time-reversed template (m[L-1-i] = match[i], dft_detect.c:1260-1262); G = WS[lpFM] * Fm
        const std::vector<float> match = scan_match(kTpl[j].hdr, hLenMax, t.spb, kTpl[j].bt, t.L);
        s->a_off[j] = (int)a_match.size(); s->nc2[j] = toeplitz_frags(match, a_match);        // c'[p'] = sum_k match[k] xf[p' + k]
        std::vector<float2> F(N, make_float2(0.f, 0.f));
        for (int i = 0; i < t.L; i++) F[t.L - 1 - i].x = match[i];
        dft_ref_host(F, tws);
        for (int k = 0; k < N; k++) {
            float2 g = F[k];
            if (iq) { const float2 w = WS[t.lpfm][k]; g = make_float2(w.x * F[k].x - w.y * F[k].y, w.x * F[k].y + w.y * F[k].x); }
            G[(size_t)j * N + k] = g;
        }
    }

    // FM low-pass as a Toeplitz product on the window stored behind ws_pad zeros: xf[i] = sum_t ws[t] xn[i-t] = sum_u h[u] xh[i+u],
    // h[u] = ws[taps-1-(u-front)], front = padding that makes ws_pad = taps-1+front a multiple of 8 (aligned 16-byte LDS reads)
    std::vector<float> ws_tail;
    if (iq) {
        const int taps = s->lpfm_taps, front = (8 - (taps - 1) % 8) % 8;
        s->ws_pad = taps - 1 + front;
        for (int lp = 0; lp < 2; lp++) {
            std::vector<float> h((size_t)front + taps, 0.f);
            for (int t = 0; t < taps; t++) h[(size_t)front + (taps - 1 - t)] = w_lp[(size_t)lp * taps + t];
            s->nc1 = toeplitz_frags(h, a_ws);
            for (int i = 0; i < taps; i++) { double acc = 0; for (int t = i + 1; t < taps; t++) acc += (double)w_lp[(size_t)lp * taps + t]; ws_tail.push_back((float)acc); }
        }
    }
    for (int j = 0; j < SC_NTPL; j++) {        // ScanPreArgs.kap: 4.02 u ||ws||_1 sqrt(L) with the taps as the kernel has them (f16)
        s->kap[j] = 0.f;
        if (!iq || !s->tpl[j].active) continue;
        double w1 = 0; for (int t = 0; t < s->lpfm_taps; t++) { const uint16_t hb = f16_bits(w_lp[(size_t)s->tpl[j].lpfm * s->lpfm_taps + t]); _Float16 hv; memcpy(&hv, &hb, sizeof hv); w1 += fabs((double)(float)hv); }
        s->kap[j] = (float)(4.02 * 4.8828125e-4 * w1 * sqrt((double)s->tpl[j].L));
    }
    s->use_pre = cfg->opt_exact == 0 && N == SC_N;             // windows beyond 8192 samples: the exact kernel for every pair (rare: wide --bw / wide --iq input)

    const int max_if = (cfg->max_chunk + D - 1) / D;
    int ring = 1; while (ring < max_if + sr + 2 * N + 4096) ring <<= 1;        // + one second: the IMET check re-reads it (imet_resolve)
    s->ring_len = ring;
    sonde_scan_info_t &I = s->info;
    I.if_sr = sr; I.decM = D; I.dectaps = (D == 1) ? 0 : (int)s->dec.taps.size(); I.lpiq_taps = s->lpiq_taps; I.lpfm_taps = s->lpfm_taps;
    I.K = s->K; I.N = N; I.delay = s->delay; I.L2 = L2; I.ring_len = ring;

    int bad = 0;
    bad |= dalloc(&s->d_dcavg, C); bad |= dalloc(&s->d_dcsums, 2 * (size_t)C);
    if (cfg->iq_mode == SONDE_SCAN_BBIQ) { bad |= dalloc(&s->d_ptail[0], (size_t)C * 64); bad |= dalloc(&s->d_ptail[1], (size_t)C * 64); }
    if (iq) bad |= dalloc(&s->d_y, (size_t)C * ring);
    bad |= dalloc(&s->d_fm, (size_t)s->nstreams * C * ring);
    bad |= dupload(&s->d_G, G); bad |= dupload(&s->d_tw, tws); bad |= dupload(&s->d_hdr, hdrbits); bad |= dupload(&s->d_bnd, bnd);
    if (iq) { bad |= dupload(&s->d_wiq, w_iq); std::vector<float2> ws2(WS[0]); ws2.insert(ws2.end(), WS[1].begin(), WS[1].end()); bad |= dupload(&s->d_WS, ws2); }
    s->item_cap = C * (max_if / (s->K - 4) + 2);
    const size_t icap = (size_t)s->item_cap + C;           // + one re-evaluated window of the previous call per channel (prefilter)
    bad |= dalloc(&s->d_items, icap, false); bad |= dalloc(&s->d_res, icap * SC_NTPL, false);
    bad |= dupload(&s->d_amatch, a_match); bad |= dupload(&s->d_aws, a_ws); bad |= dupload(&s->d_wstail, ws_tail);
    bad |= dalloc(&s->d_pre, icap * SC_NTPL, false); bad |= dalloc(&s->d_work, icap * SC_NTPL, false);
    if (bad) { sonde_scan_destroy(s); return SONDE_E_NOMEM; }
    HIPCHK(hipHostMalloc((void **)&s->h_items, icap * sizeof(ScanItem), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s->h_res, icap * SC_NTPL * sizeof(ScanRes), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s->h_pre, icap * SC_NTPL * sizeof(ScanPre), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s->h_work, icap * SC_NTPL * sizeof(ScanWork), hipHostMallocDefault));
    {   // the scanner's kernels are short and its call waits for them between stages: on a device that also runs a demodulator engine they must not
        // queue behind a millisecond of decimator workgroups — highest stream priority (the engine's streams have the default one)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
        HIPCHK(hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, hi));
    }

    // IQ-DC: always on, fixed window (dft_detect.c:1152-1156)
    s->dc_max = (uint32_t)(sr / 32); if (D > 1) s->dc_max *= (uint32_t)D;
    if (iq && (s->dc_max == 0 || s->dc_max % D)) { sonde_scan_destroy(s); return SONDE_E_ARG; }
    {   // int16 / 8-bit base-rate input at the generated decimator's geometry (D = 50, Q = 7, a mixer table whose period is whole blocks): the front end reads its
        // input ONCE (k_mix_decimate50r + the fold at the IF rate) instead of a summing pass and a mixing pass.  SONDE_SCAN_TWO_PASS=1: the two-pass form (A/B)
        const bool two_pass = getenv("SONDE_SCAN_TWO_PASS") != nullptr;
        if (cfg->iq_mode == SONDE_SCAN_BBIQ && cfg->bits != 32 && !s->wide_fe && D == 50 && s->Q == 7 && s->lut_len % D == 0 && s->d_wtab && !two_pass) {
            s->etab_len = s->lut_len / D;
            s->bsum_stride = (cfg->max_chunk + D - 1) / D;
            const int nseg_cap = cfg->max_chunk / (int)s->dc_max + 2;
            if (dalloc(&s->d_etab64, (size_t)C * s->etab_len, false) || dalloc(&s->d_bsum, (size_t)C * s->bsum_stride, false)
                || dalloc(&s->d_corr, (size_t)C * (nseg_cap + 1) * 8) || dalloc(&s->d_dcprev[0], (size_t)C) || dalloc(&s->d_dcprev[1], (size_t)C)
                || dalloc(&s->d_hist[0], (size_t)C * s->lpiq_taps) || dalloc(&s->d_hist[1], (size_t)C * s->lpiq_taps)) { sonde_scan_destroy(s); return SONDE_E_NOMEM; }
            sonde_launch_md_etable64(s->d_chanf0, s->d_wtab, D, s->Q, s->etab_len, C, s->d_etab64, nullptr);      // E of the double-phase mixer table, once
            HIPCHK(hipDeviceSynchronize());
            s->front_raw = true;
        }
    }
    s->chan.resize(C);
    for (auto &c : s->chan) {
        for (int j = 0; j < kNrs; j++) { c.mv[j] = 0; c.mv_pos[j] = 0; c.mv0_pos[j] = 0; c.mp[j] = 0; c.dc[j] = 0; c.df[j] = 0; c.type[j] = kTpl[j].type; c.tn[j] = kTpl[j].tn; c.detect2[j] = 0; }
        c.next_sin = (uint32_t)(s->K - 4);
    }
    *out = s;
    return 0;
}

void sonde_scan_destroy(sonde_scan_t *s) {
    if (!s) return;
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    for (auto &e : s->ev_fe) if (e) hipEventDestroy(e);
    for (auto &e : s->ev_rw) if (e) hipEventDestroy(e);
    if (s->ev_wait) hipEventDestroy(s->ev_wait);
    if (s->d_spprof) {
        unsigned long long h[8] = {0};
        if (hipMemcpy(h, s->d_spprof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[7])
            fprintf(stderr, "scan_pre prof (template 1, %llu workgroups, mean shader cycles): load+dc %llu  convert+stage %llu  fm-lowpass %llu  prefix %llu  correlation %llu  reduce %llu\n",
                    h[7], h[0] / h[7], h[1] / h[7], h[2] / h[7], h[3] / h[7], h[4] / h[7], h[5] / h[7]);
        hipFree(s->d_spprof);
    }
    if (s->h_items) hipHostFree(s->h_items);
    if (s->h_res) hipHostFree(s->h_res);
    if (s->h_pre) hipHostFree(s->h_pre);
    if (s->h_work) hipHostFree(s->h_work);
    void *ptrs[] = { s->d_amatch, s->d_aws, s->d_wstail, s->d_pre, s->d_work, s->d_scratch, s->d_f32in, s->d_chanf0, s->d_dcavg, s->d_dcsums, s->d_ptail[0], s->d_ptail[1], s->d_y, s->d_fm, s->d_wiq, s->d_WS, s->d_G, s->d_tw,
                     s->d_hdr, s->d_bnd, s->d_items, s->d_res, s->d_stage, s->d_wtab, s->d_conv, s->d_dcsums_f, s->d_zring, s->d_taps_f, s->d_segsums, s->d_dcseg, s->d_etab64, s->d_bsum, s->d_corr, s->d_dcprev[0], s->d_dcprev[1], s->d_hist[0], s->d_hist[1] };
    for (void *p : ptrs) if (p) hipFree(p);
    delete s;
}

int sonde_scan_info(const sonde_scan_t *s, sonde_scan_info_t *info) {
    if (!s || !info) return SONDE_E_ARG;
    *info = s->info;
    return 0;
}

// frm_M10 (dft_detect.c:932-977): differential Manchester over 2 header symbols + the 14 symbols sliced on the device
static uint32_t m10_bytes(const char *hdr, uint32_t mask, int inv) {
    const int ofs = ((int)strlen(hdr) - 28) / 2;
    char bit0 = (char)(0x30 + inv), frmbit[17];
    for (int p = 0; p < 16; p++) {
        char mb0;
        if (p < ofs) mb0 = (char)(hdr[28 + 2 * p] ^ inv);
        else mb0 = ((mask >> (p - ofs)) & 1u) ? '1' : '0';
        frmbit[p] = (char)(0x31 ^ (bit0 ^ mb0));
        bit0 = mb0;
    }
    uint32_t bytes = 0;
    for (int p = 0; p < 16; p++) bytes = (bytes << 1) | (uint32_t)(frmbit[p] == '1');
    return bytes;
}

// decision logic of main() for one window of one channel (dft_detect.c:1494-1649)
// print / bookkeeping of one accepted header (the `if (header_found)` block of main, dft_detect.c:1609-1643)
static void report(sonde_scan *s, int ch, int j, int &header_found, uint32_t &frm2) {
    Chan &c = s->chan[ch];
    const bool iq = s->cfg.iq_mode != SONDE_SCAN_AUDIO;
    if (!header_found) return;
    int printed = 0;
    if (c.mv[j] > s->thres[j] || c.mv[j] < -s->thres[j]) {
        if (s->cfg.opt_d2) {
            c.detect2[j] += 1;
            int tn = 0; for (tn = 0; tn < kNrs; tn++) if (c.detect2[tn] > 1) break;
            c.d2_tn = tn;
            if (c.d2_tn == kNrs) header_found = 0;
        }
        if (!s->cfg.opt_d2 || j == c.d2_tn) printed = 1;
        sonde_detection_t d; memset(&d, 0, sizeof d);
        d.channel = ch; d.tpl = j; d.tn = c.tn[j]; snprintf(d.type, sizeof d.type, "%s", c.type[j]);
        d.score = c.mv[j]; d.sample = c.mv_pos[j]; d.printed = printed;
        if (j < SC_NTPL && s->tpl[j].is_m10) { d.m10_bytes = frm2 & 0xFFFF; frm2 = 0; }
        if (s->cfg.opt_dc && iq) { d.df = c.df[j]; d.freq_hz = c.df[j] * (float)s->cfg.sample_rate; }
        s->queue.push_back(d);
    }
    if (std::fabs(c.mv_max) < std::fabs(c.mv[j])) { c.mv_max = c.mv[j]; c.j_max = j; }
}

static void end_window(sonde_scan *s, int ch, int header_found) {       // dft_detect.c:1647-1650
    Chan &c = s->chan[ch];
    if ((header_found && !s->cfg.opt_cont) || c.d2_tn < kNrs) c.done = true;
    for (int j = 0; j < kNrs; j++) c.mv[j] = 0.0f;
}

static void decide(sonde_scan *s, int ch, const ScanRes *res, uint32_t win_sin) {
    Chan &c = s->chan[ch];
    for (int j = 0; j <= kIdxImetAfsk; j++) {
        if (!s->tpl[j].active) continue;
        c.mv0_pos[j] = c.mv_pos[j];
        c.mp[j] = res[j].mp;
        c.dc[j] = res[j].dc;
        if (res[j].mp > 0 || res[j].mp == -1) {                // getCorrDFT ran to its end (-1: an all-zero window, its wrapped position blocks the next window's hit — k_scan_corr)
            c.mv[j] = res[j].mv; c.mv_pos[j] = res[j].mpos;
            if (s->cfg.opt_dc) c.df[j] = (float)(c.dc[j] / (2.0 * 0.8 * s->info.decM));
        }
    }
    int header_found = 0;
    uint32_t frm2 = 0;
    for (int j0 = 0; j0 <= kIdxImetAfsk; j0++) {
        int j = j0;
        if (!s->tpl[j].active) continue;
        if (!(c.mp[j] > 0 && (c.mv[j] > s->thres[j] || c.mv[j] < -s->thres[j]))) continue;
        if (!(c.mv_pos[j] > c.mv0_pos[j])) continue;
        if (!(res[j].herrs >= 0 && res[j].herrs < kTpl[j].herrs)) continue;
        if (s->tpl[j].is_m10) {
            const uint32_t bytes = m10_bytes(kTpl[j].hdr, res[j].m10, c.mv[j] < 0);
            int h = 0; for (int q = 0; q < 4; q++) h += (bytes >> q) & 1u;
            if (h < 2 || (h == 2 && (bytes & 0xF0) == 0x20)) { c.type[j] = "M20"; c.tn[j] = 6; }
            else { c.type[j] = "M10"; c.tn[j] = 5; }
            frm2 = bytes;
        }
        if (j == kIdxImetAfsk) {
            // the reference now reads one more second of samples inside this window's decision; park the channel until
            // they exist (imet_resolve), everything decided so far in this window is kept
            c.imet_hold = true; c.imet_sin = win_sin; c.imet_hf = header_found;
            return;
        }
        header_found = 1;
        report(s, ch, j, header_found, frm2);
    }
    end_window(s, ch, header_found);
}

static int read_fm_phys(sonde_scan *s, int channel, int phys, int64_t first, int32_t count, float *out) {
    const float *base = s->d_fm + ((size_t)phys * s->cfg.n_channels + channel) * s->ring_len;
    for (int32_t done = 0; done < count;) {
        const uint32_t idx = (uint32_t)(first + done) & (uint32_t)(s->ring_len - 1);
        const int run = (int)std::min<int64_t>(count - done, s->ring_len - idx);
        HIPCHK(hipMemcpy(out + done, base + idx, (size_t)run * sizeof(float), hipMemcpyDeviceToHost));
        done += run;
    }
    return 0;
}

// IMET AFSK check (dft_detect.c:1533-1607): after an IMET preamble hit the reference reads one more second, sums the
// magnitude spectra of 4093-sample blocks of the FM stream and decides between IMET4 / IMET1RS (2200 Hz space tone present,
// stronger than 2400 Hz and 800 Hz) and nothing.  Rare and bit-rate-scale work: host side, with the reference's transform.
// Returns 1 when the held window was completed, 0 if the second of samples is not there yet.
static int imet_resolve(sonde_scan *s, int ch, bool eof) {
    Chan &c = s->chan[ch];
    const int sr = s->info.if_sr, N = s->N, Dn = N / 2 - 3, j0 = kIdxImetAfsk;
    const uint32_t S = c.imet_sin;
    const uint32_t avail = s->m_out - S;
    if (!eof && avail < (uint32_t)sr) return 0;
    const int n_read = (int)std::min<uint32_t>((uint32_t)sr, avail);
    const int nb = n_read / Dn;
    std::vector<float> blk((size_t)std::max(1, nb) * Dn), db(N, 0.f);
    if (nb > 0 && read_fm_phys(s, ch, s->tpl[j0].stream, (int64_t)S - s->delay, nb * Dn, blk.data()) < 0) return SONDE_E_NOGPU;
    const std::vector<float2> tws = ref_twiddles(s->log2n);
    for (int b = 0; b < nb; b++) {
        std::vector<float2> X(N, make_float2(0.f, 0.f));
        for (int i = 0; i < Dn; i++) X[i].x = blk[(size_t)b * Dn + i];
        dft_ref_host(X, tws);
        for (int m = 0; m < N; m++) db[m] = (float)((double)db[m] + std::hypot((double)X[m].x, (double)X[m].y));
    }
    int header_found = c.imet_hf, j = j0;
    uint32_t frm2 = 0;
    const float df = (1 / (float)N) * sr;                                   // bin2freq(1)
    int m = (int)(50.0 / df); if (m < 1) m = 1;
    auto f2b = [&](int f) { return (float)(f * N) / (float)sr; };
    if (f2b(2500) > N / 2) { c.imet_hold = false; c.done = true; return 1; } // `goto ende`: the reference stops here
    auto band = [&](int f) { const int bin = (int)f2b(f); float p = 0.f; for (int n = 0; n < m; n++) p += db[bin - m / 4 + n]; return p; };
    const float pow2200 = band(2200), pow2400 = band(2400);
    c.mv[j0] = std::fabs(c.mv[j0]);
    if (pow2200 > pow2400) {
        const float pow800 = band(800);
        if (pow2200 > pow800) {                                             // IMET -> IMET1RS / IMET4
            const bool iq = s->cfg.iq_mode != SONDE_SCAN_AUDIO;
            j = (iq && (double)s->cfg.bw_khz * 1e3 > 50e3) ? 16 : 17;
            c.mv[j] = c.mv[j0]; c.mv_pos[j] = c.mv_pos[j0]; c.dc[j] = c.dc[j0]; c.df[j] = c.df[j0];
            c.mv[j0] = 0.0f;
            header_found = 1;
        } else c.mv[j0] = 0.0f;
    } else c.mv[j0] = 0.0f;
    report(s, ch, j, header_found, frm2);
    end_window(s, ch, header_found);
    c.imet_hold = false;
    c.next_sin = S + (uint32_t)n_read + (uint32_t)(s->K - 4);               // k restarts after the extra second (:1505,1544)
    return 1;
}

// Evaluate every correlation window that is complete (sample_in = k (K-4), pos = sample_out = sample_in - 1 - delay;
// dft_detect.c:1483-1505,811-813) and apply the decision logic, channel by channel in stream order.  A channel parked by
// the IMET check resumes with a shifted window phase, so the evaluation runs in rounds until nothing new is due.
static int run_windows(sonde_scan *s) {
    const int C = s->cfg.n_channels, mode = s->cfg.iq_mode;
    const float tl = s->cfg.time_limit;
    const float limit = (tl + 1.0f) * (float)s->info.if_sr;
    for (int round = 0; round < 64; round++) {
        for (int c = 0; c < C; c++) if (s->chan[c].imet_hold && !s->chan[c].done) { const int r = imet_resolve(s, c, false); if (r < 0) return r; }
        int n_items = 0;
        std::vector<int> first_item(C + 1, 0);
        for (int c = 0; c < C; c++) {
            first_item[c] = n_items;
            Chan &cs = s->chan[c];
            if (!cs.done && !cs.imet_hold && tl > 0 && (float)cs.next_sin > limit && (float)s->m_out > limit) cs.done = true;   // -t: the sample loop broke
            uint32_t sin = cs.next_sin;
            while (!cs.done && !cs.imet_hold && sin <= s->m_out && n_items < s->item_cap) {
                if (tl > 0 && (float)sin > limit) break;
                s->h_items[n_items].ch = c; s->h_items[n_items].pos = sin - 1u - (uint32_t)s->delay; n_items++;
                sin += (uint32_t)(s->K - 4);
            }
        }
        first_item[C] = n_items;
        if (!n_items) break;
        for (auto &e : s->ev_rw) if (!e) HIPCHK(hipEventCreate(&e));
        hipEvent_t e0 = s->ev_rw[0], e1 = s->ev_rw[1], e2 = s->ev_rw[2];
        ScanCorrArgs a{};
        a.fm = s->d_fm; a.n_ch = C; a.ring_len = s->ring_len; a.items = s->d_items; a.n_items = n_items;
        memcpy(a.tpl, s->tpl, sizeof a.tpl);
        a.G = s->d_G; a.WS = s->d_WS; a.lpfm_taps = s->lpfm_taps; a.tws = s->d_tw; a.K = s->K; a.opt_dc = s->cfg.opt_dc;
        a.opt_iq = (mode != SONDE_SCAN_AUDIO); a.hdrbits = s->d_hdr; a.bnd = s->d_bnd; a.out = s->d_res;
        std::vector<uint32_t> exact((size_t)n_items + C, 0u);                 // per window: templates evaluated by the exact kernel
        int n_all = n_items;                                                  // + re-evaluated last windows of the previous call
        std::vector<int> aux_of(C, -1);
        if (!s->use_pre && s->N > SC_N) {
            // N_DFT 16384 / 32768: every pair through the global-memory form of the exact kernel, in batches that share the scratch arrays
            hipEventRecord(e0, s->stream);
            HIPCHK(hipMemcpyAsync(s->d_items, s->h_items, (size_t)n_items * sizeof(ScanItem), hipMemcpyHostToDevice, s->stream));
            hipEventRecord(e1, s->stream);
            int n_work = 0;
            for (int i = 0; i < n_items; i++) for (int j = 0; j < SC_NTPL; j++) {
                if (s->tpl[j].active) { s->h_work[n_work].item = i; s->h_work[n_work].tpl = j; n_work++; }
                else s->h_res[(size_t)i * SC_NTPL + j] = ScanRes{ 0, 0.f, 0u, 0.f, -1, 0u };
            }
            const int batch = 256;
            if (!s->d_scratch) { HIPCHK(hipMalloc((void **)&s->d_scratch, (size_t)batch * s->N * sizeof(float2))); s->scratch_pairs = batch; }
            HIPCHK(hipMemcpyAsync(s->d_work, s->h_work, (size_t)n_work * sizeof(ScanWork), hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipMemsetAsync(s->d_res, 0, (size_t)n_items * SC_NTPL * sizeof(ScanRes), s->stream));
            a.N = s->N; a.log2n = s->log2n; a.scratch = s->d_scratch;
            for (int w0 = 0; w0 < n_work; w0 += batch) {
                a.work = s->d_work + w0; a.n_work = std::min(batch, n_work - w0);
                if (sonde_launch_scan_corr(&a, s->stream) < 0) return SONDE_E_NOGPU;
            }
            std::vector<ScanRes> keep(s->h_res, s->h_res + (size_t)n_items * SC_NTPL);
            HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, (size_t)n_items * SC_NTPL * sizeof(ScanRes), hipMemcpyDeviceToHost, s->stream));
            hipEventRecord(e2, s->stream);
            HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s);
            for (int i = 0; i < n_items; i++) for (int j = 0; j < SC_NTPL; j++) if (!s->tpl[j].active) s->h_res[(size_t)i * SC_NTPL + j] = keep[(size_t)i * SC_NTPL + j];
            timed(s, "scan_corr", e1, e2);
            for (int i = 0; i < n_items; i++) exact[i] = 0xffffu;
        } else if (!s->use_pre) {
            hipEventRecord(e0, s->stream);
            HIPCHK(hipMemcpyAsync(s->d_items, s->h_items, (size_t)n_items * sizeof(ScanItem), hipMemcpyHostToDevice, s->stream));
            hipEventRecord(e1, s->stream);
            if (sonde_launch_scan_corr(&a, s->stream) < 0) return SONDE_E_NOGPU;
            HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, (size_t)n_items * SC_NTPL * sizeof(ScanRes), hipMemcpyDeviceToHost, s->stream));
            hipEventRecord(e2, s->stream);
            HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s);
            timed(s, "scan_corr", e1, e2);
            for (int i = 0; i < n_items; i++) exact[i] = 0xffffu;
        } else {
            // pass 1: the prefilter over every (window, template)
            ScanPreArgs pa{};
            pa.fm = s->d_fm; pa.n_ch = C; pa.ring_len = s->ring_len; pa.items = s->d_items; pa.n_items = n_items;
            memcpy(pa.tpl, s->tpl, sizeof pa.tpl);
            pa.a_match = s->d_amatch; memcpy(pa.a_off, s->a_off, sizeof pa.a_off); memcpy(pa.nc2, s->nc2, sizeof pa.nc2);
            pa.a_ws = s->d_aws; pa.nc1 = s->nc1; pa.taps = s->lpfm_taps; pa.ws_pad = s->ws_pad; pa.ws_tail = s->d_wstail;
            pa.K = s->K; pa.opt_dc = s->cfg.opt_dc; pa.opt_iq = a.opt_iq; pa.lpfm_taps = s->lpfm_taps; pa.out = s->d_pre;
            memcpy(pa.kap, s->kap, sizeof pa.kap);
            hipEventRecord(e0, s->stream);
            HIPCHK(hipMemcpyAsync(s->d_items, s->h_items, (size_t)n_items * sizeof(ScanItem), hipMemcpyHostToDevice, s->stream));
            {   // profiling aid: cycles per phase of k_scan_pre, printed when the scanner is destroyed
                static const bool want = getenv("SONDE_SP_PROF") != nullptr;
                if (want && !s->d_spprof) { if (hipMalloc((void **)&s->d_spprof, 8 * sizeof(unsigned long long)) == hipSuccess) hipMemset(s->d_spprof, 0, 8 * sizeof(unsigned long long)); }
                pa.prof = s->d_spprof;
            }
            if (sonde_launch_scan_pre(&pa, s->stream) < 0) return SONDE_E_NOGPU;
            HIPCHK(hipMemcpyAsync(s->h_pre, s->d_pre, (size_t)n_items * SC_NTPL * sizeof(ScanPre), hipMemcpyDeviceToHost, s->stream));
            hipEventRecord(e1, s->stream);
            HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s);
            timed(s, "scan_pre", e0, e1);
            // candidates, and for each the same template in the window before it (this call's, or the last one of the previous call)
            int n_work = 0;
            auto add = [&](int item, int j) { if (!((exact[item] >> j) & 1u)) { exact[item] |= 1u << j; s->h_work[n_work].item = item; s->h_work[n_work].tpl = j; n_work++; } };
            for (int c = 0; c < C; c++) {
                Chan &cs = s->chan[c];
                for (int i = first_item[c]; i < first_item[c + 1]; i++)
                    for (int j = 0; j < SC_NTPL; j++) {
                        if (!s->tpl[j].active || !(s->h_pre[(size_t)i * SC_NTPL + j].smax > s->thres[j] - kPreMargin)) continue;
                        add(i, j);
                        if (i > first_item[c]) add(i - 1, j);
                        else if (cs.last_valid && !((cs.last_exact >> j) & 1u)) {
                            if (aux_of[c] < 0) { aux_of[c] = n_all; s->h_items[n_all].ch = c; s->h_items[n_all].pos = cs.last_pos; n_all++; }
                            add(aux_of[c], j);
                        }
                    }
            }
            { auto &k = s->stats["exact_pairs"]; k.ms += n_work; k.n += 1; }          // counters read through sonde_scan_kernel_ms: average pairs per round
            { int act = 0; for (int j = 0; j < SC_NTPL; j++) act += s->tpl[j].active; auto &k = s->stats["pre_pairs"]; k.ms += (double)n_items * act; k.n += 1; }
            // pass 2: the reference's transform network for the listed pairs
            if (n_work > 0) {
                hipEventRecord(e1, s->stream);
                a.n_items = n_all; a.work = s->d_work; a.n_work = n_work;
                if (n_all > n_items) HIPCHK(hipMemcpyAsync(s->d_items + n_items, s->h_items + n_items, (size_t)(n_all - n_items) * sizeof(ScanItem), hipMemcpyHostToDevice, s->stream));
                HIPCHK(hipMemcpyAsync(s->d_work, s->h_work, (size_t)n_work * sizeof(ScanWork), hipMemcpyHostToDevice, s->stream));
                if (sonde_launch_scan_corr(&a, s->stream) < 0) return SONDE_E_NOGPU;
                HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, (size_t)n_all * SC_NTPL * sizeof(ScanRes), hipMemcpyDeviceToHost, s->stream));
                hipEventRecord(e2, s->stream);
                HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s);
                timed(s, "scan_corr", e1, e2);
            }
            // everything else keeps the prefilter's values: below the threshold by more than the margin, header not compared (herrs = -2)
            for (int i = 0; i < n_items; i++)
                for (int j = 0; j < SC_NTPL; j++) {
                    if ((exact[i] >> j) & 1u) continue;
                    const ScanPre &q = s->h_pre[(size_t)i * SC_NTPL + j];
                    s->h_res[(size_t)i * SC_NTPL + j] = ScanRes{ s->tpl[j].active ? q.mp : 0, q.mv, q.mpos, q.dc, s->tpl[j].active ? -2 : -1, 0u };
                }
            // a re-evaluated last window of the previous call: its exact verdict replaces what the prefilter left in mv_pos (decide() kept the
            // value from before that window in mv0_pos)
            for (int c = 0; c < C; c++) {
                if (aux_of[c] < 0) continue;
                Chan &cs = s->chan[c];
                for (int j = 0; j < SC_NTPL; j++) {
                    if (!((exact[aux_of[c]] >> j) & 1u)) continue;
                    const ScanRes &r = s->h_res[(size_t)aux_of[c] * SC_NTPL + j];
                    cs.mv_pos[j] = (r.mp > 0 || r.mp == -1) ? r.mpos : cs.mv0_pos[j];
                }
            }
        }
        for (int c = 0; c < C; c++) {
            Chan &cs = s->chan[c];
            for (int i = first_item[c]; i < first_item[c + 1]; i++) {
                if (cs.done || cs.imet_hold) break;
                const ScanRes *r = s->h_res + (size_t)i * SC_NTPL;
                sonde_scan_window_t w; memset(&w, 0, sizeof w);
                w.channel = c; w.pos = s->h_items[i].pos;
                for (int j = 0; j < SC_NTPL; j++) { w.mp[j] = r[j].mp; w.mv[j] = r[j].mv; w.mpos[j] = r[j].mpos; w.dc[j] = r[j].dc; w.herrs[j] = r[j].herrs; w.m10[j] = (s->tpl[j].is_m10 && r[j].herrs >= 0) ? m10_bytes(kTpl[j].hdr, r[j].m10, r[j].mv < 0) : 0u; }
                s->last_windows.push_back(w);
                const uint32_t win_sin = cs.next_sin;
                cs.next_sin += (uint32_t)(s->K - 4);
                cs.last_valid = true; cs.last_pos = s->h_items[i].pos; cs.last_exact = exact[i];
                decide(s, c, r, win_sin);
            }
        }
    }
    return 0;
}

int sonde_scan_process_device(sonde_scan_t *s, const void *d_in, int64_t ch_stride, int32_t n_samples) {
    if (!s || !d_in) return SONDE_E_ARG;
    const int C = s->cfg.n_channels, D = s->info.decM, mode = s->cfg.iq_mode;
    if (n_samples <= 0 || n_samples > s->cfg.max_chunk || n_samples % D || (ch_stride != 0 && ch_stride < n_samples)) return SONDE_E_RANGE;   // stride 0: one wideband stream shared by all channels
    if (s->cfg.bits == 8) {                        // cu8: (u-128)/128 == ((u-128)*256)/32768 -> feed the 16-bit path
        const int nc = ch_stride == 0 ? 1 : C;
        const int epf = mode == SONDE_SCAN_AUDIO ? std::max(1, s->cfg.audio_channels) : 2;          // bytes per sample / audio frame
        if (!s->d_conv) HIPCHK(hipMalloc((void **)&s->d_conv, (size_t)C * s->cfg.max_chunk * epf * 2));
        sonde_launch_u8_to_s16((const uint8_t *)d_in, ch_stride * epf, s->d_conv, (long long)n_samples * epf, nc, n_samples * epf, s->stream);
        d_in = s->d_conv; if (ch_stride != 0) ch_stride = n_samples;
    }
    bool f32in = s->cfg.bits == 32;
    if (s->wide_fe && !f32in) {                    // wide IF: the decimator has more than 8 blocks of taps -> float32 copy (x / 32768, exact) for the plain kernels
        const int nc = ch_stride == 0 ? 1 : C;
        if (!s->d_f32in) HIPCHK(hipMalloc((void **)&s->d_f32in, (size_t)C * s->cfg.max_chunk * sizeof(float2)));
        sonde_launch_s16_to_f32((const int16_t *)d_in, ch_stride, s->d_f32in, n_samples, nc, n_samples, s->stream);
        d_in = s->d_f32in; if (ch_stride != 0) ch_stride = n_samples;
        f32in = true;
    }
    hipEvent_t *ev = s->ev_fe;
    for (int k = 0; k < 3; k++) if (!ev[k]) HIPCHK(hipEventCreate(&ev[k]));
    const uint32_t m_first = s->m_out;
    hipEventRecord(ev[0], s->stream);
    if (mode == SONDE_SCAN_AUDIO) {
        AudioConvArgs a{}; a.pcm = (const int16_t *)d_in; a.ch_stride = ch_stride; a.n_ch = C; a.n = n_samples;
        a.nch = std::max(1, s->cfg.audio_channels); a.sel = std::min(std::max(0, s->cfg.audio_select), a.nch - 1);
        a.fm = s->d_fm; a.ring_len = s->ring_len; a.m0 = s->m_out; a.f32 = (s->cfg.bits == 32);
        sonde_launch_audio_convert(&a, s->stream);
        s->m_out += (uint32_t)n_samples; s->samples_in += (uint64_t)n_samples;
    } else {
        int done = 0;
        // int16 base-rate input with a decimation the lane-per-block kernel takes in one piece: ONE decimator launch for the whole call, the IQ-DC
        // windows it spans (1/32 s each: 32 per second of signal) handled by a table of means (MixDecArgs.dc_seg) that two small kernels fill first —
        // a launch per window made the front end launch-bound (64 launches of ~9 us per second of signal)
        static const bool no_segtab = getenv("SONDE_SCAN_NO_SEGTAB") != nullptr;      // A/B aid
        if (s->front_raw) {
            // one pass over the input: raw mix + block sums, then window sums -> means -> y -= mean * E at the IF rate
            if (mode != SONDE_SCAN_BBIQ || f32in || n_samples % D || s->dc_cnt % (uint32_t)D) return SONDE_E_ARG;      // (cannot happen: whole blocks per call)
            const int nseg_cap = s->cfg.max_chunk / (int)s->dc_max + 2;
            if (!s->d_segsums) {
                HIPCHK(hipMalloc((void **)&s->d_segsums, (size_t)C * nseg_cap * 2 * sizeof(long long)));
                HIPCHK(hipMalloc((void **)&s->d_dcseg, (size_t)C * (nseg_cap + 1) * sizeof(float2)));
            }
            const int nb = n_samples / D, seg_off = (int)(s->dc_cnt / (uint32_t)D), seg_blocks = (int)(s->dc_max / (uint32_t)D);
            MixDecArgs a{};
            a.iq = (const int16_t *)d_in; a.ch_stride = ch_stride; a.n_ch = C; a.nblocks = nb;
            a.D = D; a.Q = s->Q; memcpy(a.wtab, s->wtab.data(), sizeof a.wtab); a.wtab_g = s->d_wtab; a.wtab_scaled = 1; a.chan_f0 = s->d_chanf0; a.lut_len = s->lut_len;
            a.lut_phase = (uint32_t)(s->samples_in % (uint64_t)s->lut_len);
            a.ptail_in = s->d_ptail[s->ptail_cur]; a.ptail_out = s->d_ptail[s->ptail_cur ^ 1];
            a.y = s->d_y; a.ring_len = s->ring_len; a.m0 = s->m_out; a.phase_f64 = 1;
            a.bsum = s->d_bsum; a.bsum_stride = s->bsum_stride;
            { long long tiles = (long long)C * ((a.nblocks + 63) / 64); int G = (int)(tiles / 12288); a.G = G < 1 ? 1 : (G > 16 ? 16 : G); }
            if (sonde_launch_mix_decimate50r(&a, s->stream) < 0) return SONDE_E_ARG;
            sonde_launch_dc_rows_to_segments(s->d_bsum, s->bsum_stride, C, nb, seg_off, seg_blocks, (float)s->dc_max, s->d_segsums, s->d_dcsums, s->d_dcavg,
                                             s->d_dcprev[s->hist_cur], s->d_dcprev[s->hist_cur ^ 1], s->d_dcseg, nseg_cap + 1, s->stream);
            ScanEdgeArgs g{};
            g.n_ch = C; g.nblocks = nb; g.D = D; g.Q = s->Q; g.nseg = (seg_off + nb + seg_blocks - 1) / seg_blocks;
            g.dc_seg = s->d_dcseg; g.dc_seg_n = nseg_cap + 1; g.dc_seg_off = seg_off; g.dc_seg_blocks = seg_blocks; g.dc_prev = s->d_dcprev[s->hist_cur];      // (the call-start value: k_dc_seg_means wrote the end-of-call one to the other array)
            g.etab_len = s->etab_len; g.e0 = (uint32_t)((a.lut_phase / (uint32_t)D) % (uint32_t)s->etab_len);
            g.chan_f0 = s->d_chanf0; g.wtab = s->d_wtab; g.corr = s->d_corr;
            sonde_launch_scan_dc_edges(&g, s->stream);
            ScanFold &f = s->fold;                              // what k_scan_if (below) folds with
            f.etab = s->d_etab64; f.etab_len = s->etab_len; f.e0 = g.e0; f.m0 = s->m_out; f.nblocks = nb;
            f.dc_seg = s->d_dcseg; f.dc_seg_n = nseg_cap + 1; f.dc_seg_off = seg_off; f.dc_seg_blocks = seg_blocks;
            f.corr = s->d_corr; f.edge_n = s->Q - 1;
            f.hist_in = s->d_hist[s->hist_cur]; f.hist_out = s->d_hist[s->hist_cur ^ 1]; f.hist_n = s->lpiq_taps;
            s->fold_pending = true;
            s->ptail_cur ^= 1; s->hist_cur ^= 1;
            s->samples_in += (uint64_t)n_samples; s->m_out += (uint32_t)nb;
            s->dc_cnt = (uint32_t)(((uint64_t)s->dc_cnt + (uint64_t)n_samples) % s->dc_max);
            done = n_samples;
        }
        if (done < n_samples && mode == SONDE_SCAN_BBIQ && !f32in && D <= 64 && n_samples % D == 0 && s->dc_cnt % (uint32_t)D == 0 && !no_segtab) {
            const int nseg_cap = s->cfg.max_chunk / (int)s->dc_max + 2;
            if (!s->d_segsums) {
                HIPCHK(hipMalloc((void **)&s->d_segsums, (size_t)C * nseg_cap * 2 * sizeof(long long)));
                HIPCHK(hipMalloc((void **)&s->d_dcseg, (size_t)C * (nseg_cap + 1) * sizeof(float2)));
            }
            sonde_launch_dc_segments((const int16_t *)d_in, ch_stride, C, n_samples, s->dc_cnt, s->dc_max, s->d_segsums, s->d_dcsums, s->d_dcavg,
                                     s->d_dcseg, nseg_cap + 1, s->stream);
            MixDecArgs a{};
            a.iq = (const int16_t *)d_in; a.ch_stride = ch_stride; a.n_ch = C; a.nblocks = n_samples / D;
            a.D = D; a.Q = s->Q; memcpy(a.wtab, s->wtab.data(), sizeof a.wtab); a.wtab_g = s->d_wtab; a.wtab_scaled = (D <= 64 && !s->wide_fe); a.DS = s->DS; a.chan_f0 = s->d_chanf0; a.lut_len = s->lut_len;
            a.lut_phase = (uint32_t)(s->samples_in % (uint64_t)s->lut_len);
            a.dc_avg = s->d_dcavg; a.dc_sums = s->d_dcsums;
            a.dc_seg = s->d_dcseg; a.dc_seg_n = nseg_cap + 1; a.dc_seg_off = (int)(s->dc_cnt / (uint32_t)D); a.dc_seg_blocks = (int)(s->dc_max / (uint32_t)D);
            a.ptail_in = s->d_ptail[s->ptail_cur]; a.ptail_out = s->d_ptail[s->ptail_cur ^ 1];
            a.y = s->d_y; a.ring_len = s->ring_len; a.m0 = s->m_out; a.phase_f64 = 1;
            { long long tiles = (long long)C * ((a.nblocks + 63) / 64); int G = (int)(tiles / 12288); a.G = G < 1 ? 1 : (G > 16 ? 16 : G); }
            if (sonde_launch_mix_decimate(&a, s->stream) < 0) return SONDE_E_ARG;
            s->ptail_cur ^= 1;
            s->samples_in += (uint64_t)n_samples; s->m_out += (uint32_t)(n_samples / D);
            s->dc_cnt = (uint32_t)(((uint64_t)s->dc_cnt + (uint64_t)n_samples) % s->dc_max);
            done = n_samples;
        }
        // float32 IF-rate input (the channelizer's output: 256 channels at 50 kHz): the same table of window means, one k_mix_f32 launch for the call
        // instead of a mixer launch and a mean update per 1/32 s window (63 launches of 4-9 us per second of signal: 0.4 of scan_wide's 2.5 ms)
        if (mode != SONDE_SCAN_BBIQ && f32in && !no_segtab && n_samples > 0) {
            const int nseg_cap = s->cfg.max_chunk / (int)s->dc_max + 2;
            if (!s->d_dcsums_f) { HIPCHK(hipMalloc((void **)&s->d_dcsums_f, 2 * (size_t)C * sizeof(double))); HIPCHK(hipMemset(s->d_dcsums_f, 0, 2 * (size_t)C * sizeof(double))); }
            if (!s->d_segsums) {
                HIPCHK(hipMalloc((void **)&s->d_segsums, (size_t)C * nseg_cap * 2 * sizeof(double)));
                HIPCHK(hipMalloc((void **)&s->d_dcseg, (size_t)C * (nseg_cap + 1) * sizeof(float2)));
            }
            sonde_launch_dc_segments_f32((const float2 *)d_in, ch_stride, C, n_samples, s->dc_cnt, s->dc_max, (double *)s->d_segsums, s->d_dcsums_f, s->d_dcavg,
                                         s->d_dcseg, nseg_cap + 1, s->stream);
            MixF32Args a{}; a.x = (const float2 *)d_in; a.ch_stride = ch_stride; a.n_ch = C; a.n = n_samples;
            a.chan_f0 = s->d_chanf0; a.lut_len = s->lut_len; a.lut_phase = 0; a.phase_f64 = 1; a.dc_avg = s->d_dcavg; a.dc_sums = s->d_dcsums_f;
            a.mix = 0; a.z = s->d_y; a.zmask = (uint32_t)s->ring_len - 1; a.n0 = s->m_out;
            a.dc_seg = s->d_dcseg; a.dc_seg_n = nseg_cap + 1; a.dc_seg_off = s->dc_cnt; a.dc_seg_len = s->dc_max;
            sonde_launch_mix_f32(&a, s->stream);
            s->samples_in += (uint64_t)n_samples; s->m_out += (uint32_t)(n_samples / D);
            s->dc_cnt = (uint32_t)(((uint64_t)s->dc_cnt + (uint64_t)n_samples) % s->dc_max);
            done = n_samples;
        }
        while (done < n_samples) {
            const int take = (int)std::min<uint32_t>((uint32_t)(n_samples - done), s->dc_max - s->dc_cnt);
            if (f32in) {                                     // float32 IQ (or the wide-IF copy): plain mixer / FIR kernels, IQ-DC sums in double
                if (!s->d_dcsums_f) {
                    HIPCHK(hipMalloc((void **)&s->d_dcsums_f, 2 * (size_t)C * sizeof(double))); HIPCHK(hipMemset(s->d_dcsums_f, 0, 2 * (size_t)C * sizeof(double)));
                    if (mode == SONDE_SCAN_BBIQ) {
                        uint32_t zl = 1; while (zl < (uint32_t)s->cfg.max_chunk + (uint32_t)s->dec.taps.size() + (uint32_t)D + 64u) zl <<= 1;
                        s->zmask = zl - 1;
                        HIPCHK(hipMalloc((void **)&s->d_zring, (size_t)C * zl * sizeof(float2))); HIPCHK(hipMemset(s->d_zring, 0, (size_t)C * zl * sizeof(float2)));
                        HIPCHK(hipMalloc((void **)&s->d_taps_f, s->dec.taps.size() * sizeof(float)));
                        HIPCHK(hipMemcpy(s->d_taps_f, s->dec.taps.data(), s->dec.taps.size() * sizeof(float), hipMemcpyHostToDevice));
                    }
                }
                MixF32Args a{}; a.x = (const float2 *)d_in + (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.n = take;
                a.chan_f0 = s->d_chanf0; a.lut_len = s->lut_len; a.lut_phase = (uint32_t)(s->samples_in % (uint64_t)std::max(1, s->lut_len));
                a.phase_f64 = 1; a.dc_avg = s->d_dcavg; a.dc_sums = s->d_dcsums_f;
                if (mode == SONDE_SCAN_BBIQ) { a.mix = 1; a.z = s->d_zring; a.zmask = s->zmask; a.n0 = s->samples_in; }
                else { a.mix = 0; a.z = s->d_y; a.zmask = (uint32_t)s->ring_len - 1; a.n0 = s->m_out; }
                sonde_launch_mix_f32(&a, s->stream);
                if (mode == SONDE_SCAN_BBIQ) {
                    DecF32Args d{}; d.z = s->d_zring; d.zmask = s->zmask; d.n0 = s->samples_in; d.taps = s->d_taps_f; d.T = (int)s->dec.taps.size(); d.D = D;
                    d.n_ch = C; d.nblocks = take / D; d.y = s->d_y; d.ring_len = s->ring_len; d.m0 = s->m_out;
                    sonde_launch_decimate_f32(&d, s->stream);
                }
                s->samples_in += (uint64_t)take; s->m_out += (uint32_t)(take / D); s->dc_cnt += (uint32_t)take; done += take;
                if (s->dc_cnt == s->dc_max) { sonde_launch_dc_update_f64(C, s->d_dcsums_f, s->d_dcavg, (float)s->dc_max, s->stream); s->dc_cnt = 0; }
                continue;
            }
            if (mode == SONDE_SCAN_BBIQ) {
                MixDecArgs a{};
                a.iq = (const int16_t *)d_in + 2 * (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.nblocks = take / D;
                a.D = D; a.Q = s->Q; memcpy(a.wtab, s->wtab.data(), sizeof a.wtab); a.wtab_g = s->d_wtab; a.wtab_scaled = (D <= 64 && !s->wide_fe); a.DS = s->DS; a.chan_f0 = s->d_chanf0; a.lut_len = s->lut_len;
                a.lut_phase = (uint32_t)(s->samples_in % (uint64_t)s->lut_len);
                a.dc_avg = s->d_dcavg; a.dc_sums = s->d_dcsums;
                a.ptail_in = s->d_ptail[s->ptail_cur]; a.ptail_out = s->d_ptail[s->ptail_cur ^ 1];
                a.y = s->d_y; a.ring_len = s->ring_len; a.m0 = s->m_out; a.phase_f64 = 1;
                { long long tiles = (long long)C * ((a.nblocks + 63) / 64); int G = (int)(tiles / 12288); a.G = G < 1 ? 1 : (G > 16 ? 16 : G); }
                if (sonde_launch_mix_decimate(&a, s->stream) < 0) return SONDE_E_ARG;
                s->ptail_cur ^= 1;
            } else {
                IqConvArgs a{}; a.iq = (const int16_t *)d_in + 2 * (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.n = take;
                a.dc_avg = s->d_dcavg; a.dc_sums = s->d_dcsums; a.y = s->d_y; a.ring_len = s->ring_len; a.m0 = s->m_out;
                sonde_launch_iq_convert(&a, s->stream);
            }
            s->samples_in += (uint64_t)take; s->m_out += (uint32_t)(take / D); s->dc_cnt += (uint32_t)take; done += take;
            if (s->dc_cnt == s->dc_max) { sonde_launch_dc_update(C, s->d_dcsums, s->d_dcavg, (float)s->dc_max, s->stream); s->dc_cnt = 0; }
        }
        hipEventRecord(ev[1], s->stream);
        ScanIfArgs b{};
        b.y = s->d_y; b.fm = s->d_fm; b.n_ch = C; b.ring_len = s->ring_len; b.n = n_samples / D; b.m0 = m_first;
        b.taps = s->lpiq_taps; b.nfilt = s->nfilt; b.w = s->d_wiq;
        for (int k = 0; k < 3; k++) b.filt_stream[k] = s->filt_stream[k];
        b.raw_stream = s->raw_stream;
        if (s->fold_pending) { b.fold = s->fold; s->fold_pending = false; }      // one-pass front end: y holds raw sums, the means come off as k_scan_if loads
        sonde_launch_scan_if(&b, s->stream);
    }
    hipEventRecord(ev[2], s->stream);
    // no host wait here: the windows that are due follow from the sample counts alone, so the prefilter is queued straight behind k_scan_if
    // (round 3 stopped for the stream at this point: one host round trip per call for two timing figures).  A channel parked by the IMET check
    // re-reads its ring on the host: then the stream is drained first.
    s->fe_pending = true; s->fe_timed = mode != SONDE_SCAN_AUDIO;
    bool parked = false;
    for (int c = 0; c < C; c++) parked |= s->chan[c].imet_hold && !s->chan[c].done;
    if (parked) { HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s); }
    s->last_windows.clear();
    const int rc = run_windows(s);
    if (s->fe_pending) { HIPCHK(hipStreamSynchronize(s->stream)); flush_front_timing(s); }      // (no window was due: nothing waited yet)
    return rc;
}

/* The scanner's stream waits for everything queued so far on `stream` (a channelizer's, an engine's): the producer of d_in needs no host
 * synchronisation in front of sonde_scan_process_device. */
int sonde_scan_wait_stream(sonde_scan_t *s, void *stream) {
    if (!s) return SONDE_E_ARG;
    if (!s->ev_wait) HIPCHK(hipEventCreateWithFlags(&s->ev_wait, hipEventDisableTiming));
    HIPCHK(hipEventRecord(s->ev_wait, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent(s->stream, s->ev_wait, 0));
    return 0;
}

int sonde_scan_process_host(sonde_scan_t *s, const void *h_in, int64_t ch_stride, int32_t n_samples) {
    if (!s || !h_in) return SONDE_E_ARG;
    const int C = s->cfg.n_channels;
    if (n_samples <= 0 || n_samples > s->cfg.max_chunk || (ch_stride != 0 && ch_stride < n_samples)) return SONDE_E_RANGE;
    const size_t unit = (s->cfg.iq_mode == SONDE_SCAN_AUDIO ? (size_t)std::max(1, s->cfg.audio_channels) : 2) * (size_t)(s->cfg.bits / 8);
    if (ch_stride == 0) {                                  // one wideband stream: staged once, every channel mixes its own fq out of it
        const size_t need1 = (size_t)n_samples * unit;
        if (need1 > s->stage_bytes) {
            if (s->d_stage) { hipStreamSynchronize(s->stream); hipFree(s->d_stage); s->d_stage = nullptr; }
            HIPCHK(hipMalloc(&s->d_stage, need1)); s->stage_bytes = need1;
        }
        HIPCHK(hipMemcpyAsync(s->d_stage, h_in, need1, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));     // h_in belongs to the caller again when this returns
        return sonde_scan_process_device(s, s->d_stage, 0, n_samples);
    }
    const size_t need = (size_t)C * n_samples * unit;
    if (need > s->stage_bytes) {
        if (s->d_stage) { hipStreamSynchronize(s->stream); hipFree(s->d_stage); s->d_stage = nullptr; }
        HIPCHK(hipMalloc(&s->d_stage, need)); s->stage_bytes = need;
    }
    HIPCHK(hipMemcpy2DAsync(s->d_stage, (size_t)n_samples * unit, h_in, (size_t)ch_stride * unit, (size_t)n_samples * unit, C,
                            hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));         // h_in belongs to the caller again when this returns
    return sonde_scan_process_device(s, s->d_stage, n_samples, n_samples);
}

int sonde_scan_finish(sonde_scan_t *s) {                     // end of input: complete a pending IMET check with the samples that exist
    if (!s) return SONDE_E_ARG;
    for (int c = 0; c < s->cfg.n_channels; c++)
        if (s->chan[c].imet_hold && !s->chan[c].done) { const int r = imet_resolve(s, c, true); if (r < 0) return r; s->chan[c].done = true; }
    return 0;
}

int sonde_scan_fetch(sonde_scan_t *s, sonde_detection_t *out, int32_t max) {
    if (!s || !out || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->queue.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->queue[i];
    s->queue.erase(s->queue.begin(), s->queue.begin() + n);
    return n;
}

int sonde_scan_channel_done(const sonde_scan_t *s, int32_t channel) {
    if (!s || channel < 0 || channel >= s->cfg.n_channels) return SONDE_E_ARG;
    return s->chan[channel].done ? 1 : 0;
}

int sonde_scan_result(const sonde_scan_t *s, int32_t channel, int32_t *code) {
    if (!s || !code || channel < 0 || channel >= s->cfg.n_channels) return SONDE_E_ARG;
    const Chan &c = s->chan[channel];
    int header_found = 0;
    if (c.mv_max != 0.f) header_found = (c.mv_max < 0 && c.j_max < 3) ? -1 : 1;        // dft_detect.c:1656-1666
    *code = header_found * c.tn[c.j_max];
    return 0;
}

int sonde_scan_line(const sonde_scan_t *s, const sonde_detection_t *d, int verbose, char *buf, size_t buflen) {
    if (!s || !d || !buf) return SONDE_E_ARG;
    std::string o;
    char t[96];
    if (verbose) { snprintf(t, sizeof t, "sample: %d\n", (int)d->sample); o += t; }
    snprintf(t, sizeof t, "%s: %.4f", d->type, d->score); o += t;
    if (strncmp(d->type, "M10", 3) == 0 || strncmp(d->type, "M20", 3) == 0) { if (verbose) { snprintf(t, sizeof t, " [%04X]", d->m10_bytes & 0xFFFF); o += t; } }
    if (s->cfg.opt_dc && s->cfg.iq_mode != SONDE_SCAN_AUDIO) {
        snprintf(t, sizeof t, " , %+.1fHz", d->df * (float)s->cfg.sample_rate); o += t;
        if (verbose) { snprintf(t, sizeof t, "   [ fq-ofs: %+.6f", d->df); o += t; snprintf(t, sizeof t, " = %+.1fHz ]", d->df * (float)s->cfg.sample_rate); o += t; }
    }
    snprintf(buf, buflen, "%s", o.c_str());
    return (int)std::min(o.size(), buflen ? buflen - 1 : 0);
}

int sonde_scan_toeplitz_model(const float *h, int32_t n_taps, const float *x, int32_t n_x, float *out, int32_t n_out) {
    if (!h || !x || !out || n_taps < 1 || n_out < 0) return SONDE_E_ARG;
    std::vector<uint16_t> fr;
    const int nc = toeplitz_frags(std::vector<float>(h, h + n_taps), fr);
    auto f16 = [](uint16_t u) { _Float16 v; memcpy(&v, &u, sizeof v); return (float)v; };
    for (int i = 0; i < n_out; i++) {
        const int a = i >> 4, b = i & 15;
        float acc = 0.f;
        for (int c = 0; c < nc; c++)
            for (int g = 0; g < 4; g++)
                for (int r = 0; r < 8; r++) {                       // D[b][n] += A[b][k] B[k][n], k = 8 g + r, lane of A = b + 16 g
                    const int idx = 16 * a + 32 * c + 8 * g + r;             // A_c[b][e] = h[32 c + e - b], e = 8 g + r: x index = 16 a + b + u
                    const float xv = idx < n_x ? (float)(_Float16)x[idx] : 0.f;
                    acc += f16(fr[((size_t)c * 64 + (size_t)(b + 16 * g)) * 8 + r]) * xv;
                }
        out[i] = acc;
    }
    return nc;
}

int sonde_scan_last_windows(const sonde_scan_t *s, sonde_scan_window_t *out, int32_t max) {
    if (!s || (!out && max > 0)) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->last_windows.size(), (size_t)std::max(0, max));
    for (int i = 0; i < n; i++) out[i] = s->last_windows[i];
    return (int)s->last_windows.size();
}

int sonde_scan_read_fm(sonde_scan_t *s, int32_t channel, int32_t stream, int64_t first, int32_t count, float *out) {
    if (!s || !out || channel < 0 || channel >= s->cfg.n_channels || stream < 0 || stream > 3 || count < 0) return SONDE_E_ARG;
    int phys = 0;
    if (s->cfg.iq_mode != SONDE_SCAN_AUDIO) phys = (stream == 3) ? s->raw_stream : (s->nfilt == 3 ? stream : 0);
    if (first < 0 || first + count > (int64_t)s->m_out || (int64_t)s->m_out - first > s->ring_len) return SONDE_E_RANGE;
    const float *base = s->d_fm + ((size_t)phys * s->cfg.n_channels + channel) * s->ring_len;
    for (int32_t done = 0; done < count;) {
        const uint32_t idx = (uint32_t)(first + done) & (uint32_t)(s->ring_len - 1);
        const int run = (int)std::min<int64_t>(count - done, s->ring_len - idx);
        HIPCHK(hipMemcpy(out + done, base + idx, (size_t)run * sizeof(float), hipMemcpyDeviceToHost));
        done += run;
    }
    return 0;
}

int sonde_scan_kernel_ms(sonde_scan_t *s, const char *kernel, double *avg_ms, int64_t *launches) {
    if (!s || !kernel) return SONDE_E_ARG;
    auto it = s->stats.find(kernel);
    if (it == s->stats.end() || it->second.n == 0) { if (avg_ms) *avg_ms = 0; if (launches) *launches = 0; return 0; }
    if (avg_ms) *avg_ms = it->second.ms / (double)it->second.n;
    if (launches) *launches = it->second.n;
    return 0;
}

}  // extern "C"
