// sonde_engine.cpp — host side of libsonde_hip: engine object behind the C ABI of include/sonde_hip.h.
//
// Push model of the reference's pull loop: every sonde_engine_process_* call hands
// n_samples new complex samples of every channel to the GPU and enqueues, on one HIP stream,
//   k_mix_decimate (per IQ-DC segment)  ->  k_if_chain  ->  k_header_corr  ->  k_framesync
// Sequential per-channel state of the reference and where it lives here:
//   IQ-DC running mean (demod_mod.c:407-417,495-504)   host schedule + k_dc_update, exact integer sums
//   FIR histories (decXbuffer, lpIQ_buf, lpFM_buf)     P-tail of the decimator + IF-rate rings in HBM
//   mixer table position (sample_decM)                 lut_phase (host counter, same for all channels)
//   find_header / read_softbit2p counters              SyncState per channel in HBM
//   AFC of --dc (dsp.Df / locked / dc, demod_mod.c:1553-1600)   AfcState per channel; optimistic chunk + per-channel restart loop (process_device)
// Returns SONDE_E_ARG for what is not mirrored: --ecc3/4, --noLUT together with --dc, decimators of more than 8 tap columns on a mixed engine (single-type engines run
// those through the float32 mixer / FIR kernels, create_impl).
#include "../../include/sonde_hip.h"
#include "sonde_dev.h"
#include "sonde_host.h"
#include "sonde_scan_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace sonde;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

struct KernelStat { double ms = 0; int64_t n = 0; };
struct PendingEvt { hipEvent_t a, b; const char *name; };

struct sonde_engine {
    sonde_cfg_t cfg{};
    sonde_info_t info{};
    hipStream_t stream = nullptr;      // A: input staging, k_mix_decimate, k_dc_update
    hipStream_t stream_b = nullptr;    // B: IF chain, header correlation, framesync (may overlap the next call's A work)
    unsigned long long *d_wfprof = nullptr;        // SONDE_WF_PROF
    hipStream_t stream_c = nullptr;    // C: record copies of a lagged fetch (on B they would queue behind the call that is still running)
    hipStream_t stream_e = nullptr;    // E: k_rs41_ecc_frames + the frame-counter publish of a call, beside the next call's decimator (RS41 engines with ECC)
    hipEvent_t ev_s = nullptr;         // B -> E hand-over
    // DFM / M10 block codes of the hits on the device (k_dfm_hits / k_m10_hits, sonde_softin_dev.hip): decoded frames per ring slot, {done, ticket}, M10's bit characters per channel
    sonde_dfm_frame_t *d_dfm_out = nullptr; sonde_m10_frame_t *d_m10_out = nullptr; unsigned *d_blk_done = nullptr; char *d_m10_bits = nullptr;
    std::vector<sonde_dfm_frame_t> h_dfm; std::vector<sonde_m10_frame_t> h_m10; bool soft_lazy = false; unsigned soft_lazy_start = 0;
    uint32_t *d_ecc_list = nullptr; unsigned *d_ecc_cnt = nullptr;     // two work lists of max_frames record slots, alternating per call; {count[2], done[2]}
    hipEvent_t ev_a[4] = {}, ev_b[4] = {}, ev_if[4] = {};      // per call: decimator done (A), records complete (B / E), y ring read (B: IF chain done)
    unsigned *h_count = nullptr;       // pinned: frame counter snapshot after each call's framesync
    unsigned *h_count_dev = nullptr;   // the same words as the device addresses them (k_publish_u32 writes them)
    FrameRec *h_recs = nullptr;        // pinned staging for record fetches
    int64_t call = 0;                  // process calls issued
    unsigned read_idx = 0;             // frames already handed to the caller (monotonic)
    bool eof_pending = false;          // an end-of-stream framesync ran after the last counter snapshot
    // design
    Decimator dec; int Q = 0, G = 8, DS = 0; float *d_wtab = nullptr;
    float2 *d_etab = nullptr, *d_dcavg_prev = nullptr; int etab_len = 0; int dc_since = 1 << 20;      // fold mode of the decimator (MixDecArgs.etab)
    std::vector<float> w_iq, w_fm, match, wtab;
    float sps = 0, sps_design = 0, baud = 0, bt = 0, hmod = 0, thres = 0, l_win = -1;   // sps: what the slicers / header bit clock use; sps_design: what init_buffers() saw
    int symlen = 1, symhd = 1, hdmax = 0, bitofs = 0, nbits = 0, hdrlen = 0;
    uint32_t frame_samples = 0;
    double rho = 0;
    // device
    float *d_Bop = nullptr; double *d_chanf0 = nullptr; int lut_len = 0;
    float2 *d_dcavg = nullptr; long long *d_dcsums = nullptr;
    float2 *d_ptail[2] = { nullptr, nullptr }; int ptail_cur = 0;
    float2 *d_y = nullptr, *d_ifiq = nullptr; float *d_fm = nullptr, *d_bufs = nullptr, *d_corr = nullptr, *d_raw = nullptr;
    float *d_wiq = nullptr, *d_wfm = nullptr, *d_match = nullptr;
    uint32_t corr_limit = 0;                       // pass 1 of the two correlation / sync passes of a call
    // header search with the reference's transform (k_sync_plan / k_sync_window_fft): per channel win_W planned windows
    WinItem *d_win = nullptr; float2 *d_Fm = nullptr, *d_tws = nullptr; int win_W = 0;
    uint32_t *d_work = nullptr, *d_work_count = nullptr; int sync_rounds = 0;      // compact window list of the round, counters [2]
    sonde_summary_t *d_summary = nullptr; uint32_t summary_base = 0;      // caller-owned device buffer (sonde_engine_set_summary)
    sonde_summary_t *d_summary_snap = nullptr;                            // caller-owned, 2 x n_channels records (sonde_engine_set_summary_snapshots)
    int corr_types = 0, corr_isps = 0; float *d_shapes = nullptr, *d_symsign = nullptr; int *d_symtype = nullptr;
    SyncState *d_state = nullptr; FrameRec *d_frames = nullptr; unsigned *d_fcount = nullptr; float *d_soft = nullptr, *d_soft1 = nullptr;
    uint4 *d_bitwin = nullptr; uint32_t *d_bitend = nullptr;
    uint8_t *d_consts = nullptr;   // hdr[64] | hdr_bytes[8] | mask[64] | gf_exp[512] | gf_log[256]
    int16_t *d_stage = nullptr; size_t stage_bytes = 0;
    bool ifiq = false;                             // --iq0/2/3 input
    // --dc: AFC state and the rings a restart needs (rotated pre-filter stream, raw FM, FM-stream correlation)
    int opt_iq = 5; float match_sum = 0.f;
    std::vector<float> w_iq0; float *d_wiq0 = nullptr;
    float2 *d_yrot = nullptr; float *d_fmraw = nullptr, *d_corr2 = nullptr;
    AfcState *d_afc = nullptr; uint32_t *d_start = nullptr; unsigned *d_pending = nullptr, *h_pending = nullptr;
    // float32 input: IQ-DC sums in double, ring of mixed base-rate samples, taps in time order
    double *d_dcsums_f = nullptr; float2 *d_zring = nullptr; float *d_taps_f = nullptr; uint32_t zmask = 0;
    hipEvent_t ev_copy = nullptr;                  // end of the host -> staging copy of process_host
    int16_t *d_conv = nullptr;                     // cu8 input: converted int16 copy [C][max_chunk]
    float2 *d_conv32 = nullptr; bool f32_path = false, conv32 = false;   // f32_path: the plain float32 mixer / FIR kernels (cf32 input, or a decimator of more than 8 tap columns); conv32: 16- / 8-bit input converted for them [C][max_chunk] float2
    int ring_len = 0, max_frames = 0;
    // stream position
    uint64_t samples_in = 0;       // base-rate samples consumed per channel
    uint32_t m_out = 0;            // IF samples produced per channel
    uint32_t dc_cnt = 0, dc_max = 0, dc_lim = 0;
    // base-rate engines whose channels were restarted at run time: every channel has its own sample clock (pcs = per-channel schedule):
    // mixer table phase origin, IQ-DC segment counters (host mirror + device copy kept by k_dc_update_pcs)
    bool pcs = false; std::vector<uint32_t> pcs_cnt, pcs_max; uint32_t dc_max0 = 0;
    uint32_t *d_epoch_phase = nullptr, *d_pcs_cnt = nullptr, *d_pcs_max = nullptr; int32_t *d_pcs_since = nullptr;
    // results of the last fetch
    std::vector<float> last_soft, last_soft1; int last_n = 0;
    std::vector<uint8_t> last_frame;
    std::vector<char> m10_bits;                    // M10: gpx.frame_bits per channel (persists between frames like the reference's)   // [n_ch][518] gpx.frame of the reference persists across frames
    bool overflow = false;
    bool in_call = false, ecc_listed = false;   // inside sonde_engine_process_device; a frame sync of this call was given a work list
    float2 *d_park = nullptr;          // parking arrays of k_sync_window_fft_h
    bool small_tail = false;           // header search and frame sync in workgroups that fit the slot of one decimator workgroup (two-stream engines)
    bool dev_ecc = true;               // rs41_ecc() of whole frames in k_framesync (SONDE_HOST_ECC=1: on the host from the device syndromes, the A/B switch)
    long long host_ecc_frames = 0;     // frames whose RS decoder ran on the host (fetch_rs41)
    bool m10_chk3 = false;                         // m10mod --chk3 (sonde_engine_set_m10_chk3)
    // channels restarted in mid-stream (sonde_engine_restart_channel): per-channel stream start in IF samples
    std::vector<uint32_t> epoch; uint32_t *d_epoch = nullptr; int eof_ch = -1;
    // profiling
    bool prof = false, prof_skip = false; int prof_level = 2; std::map<std::string, KernelStat> stats; std::vector<PendingEvt> pend;
    // Mixed engines (sonde_engine_create_mixed).  Nothing in front of the IF rate depends on the sonde type (demod_mod.c:1222-1249: IF_sr, decM and the
    // decimator taps follow from the sample rate and --min only; mixer table and IQ-DC likewise), so ONE object owns mixer / decimator / IQ-DC of all
    // channels — `front_only`: no IF-rate state of its own — and keeps the channels grouped by type inside: `groups` are engines of one type each whose
    // IF chain reads their rows of the owner's y ring.  grp_of_ch / row_of_ch: caller's channel -> group, row; ch_of_grp[g][i]: channel i of group g ->
    // caller's channel; d_in_row: row -> caller's channel (what the decimator reads for that row).
    std::vector<sonde_engine *> groups; std::vector<int32_t> grp_of_ch, row_of_ch; std::vector<std::vector<int32_t>> ch_of_grp; int32_t *d_in_row = nullptr;
    bool front_only = false; int64_t snap_calls = 0;       // snap_calls: calls up to which the owner has enqueued a summary snapshot copy (ev_b of the owner)
    // a group of a mixed engine: who owns its front end, the first of the owner's rows that is this group's, its channels' numbers at the caller (summary records)
    sonde_engine *front = nullptr; int front_row = 0; int32_t *d_sum_map = nullptr; bool is_group = false;
    bool merged = false, borrowed_b = false;     // merged (owner): one launch per IF-rate stage for all groups, on the one stream B they share; borrowed_b (group): stream_b is the owner's
};
// how sonde_engine_create_mixed creates its parts
struct CreateLink { bool is_group, front_only; int min_ring; hipStream_t shared_b; };     // shared_b: the ONE stream B of a mixed engine whose groups' IF-rate stages share launches (nullptr: a stream B per part)

template <class T> static int dalloc(T **p, size_t n, bool zero = true) {
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    if (zero) HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
    return 0;
}

static void prof_begin(sonde_engine *e, const char *name, hipStream_t s) {
    if (!e->prof || (e->prof_level == 1 && strcmp(name, "mix_decimate") != 0)) { e->prof_skip = true; return; }
    e->prof_skip = false;
    PendingEvt p; p.name = name;
    hipEventCreate(&p.a); hipEventCreate(&p.b);
    hipEventRecord(p.a, s);
    e->pend.push_back(p);
}
static void prof_end(sonde_engine *e, hipStream_t s) { if (e->prof && !e->prof_skip) hipEventRecord(e->pend.back().b, s); }
static void prof_collect(sonde_engine *e) {
    for (auto &p : e->pend) {
        float ms = 0; hipEventSynchronize(p.b); hipEventElapsedTime(&ms, p.a, p.b);
        auto &s = e->stats[p.name]; s.ms += ms; s.n += 1;
        hipEventDestroy(p.a); hipEventDestroy(p.b);
    }
    e->pend.clear();
}

// Hit records completed up to `lag` process calls ago (0 = everything enqueued so far; syncs stream B fully).
// The device frame counter is monotonic; records live in a ring of max_frames entries.
static int collect_records(sonde_engine *e, int lag, std::vector<FrameRec> &recs, std::vector<float> *soft, int max_take) {
    unsigned count = 0;
    if (lag > 0) {
        const int64_t target = e->call - 1 - lag;
        if (target < 0) { recs.clear(); if (soft) soft->clear(); return 0; }
        if (hipEventSynchronize(e->ev_b[target & 3]) != hipSuccess) return SONDE_E_NOGPU;
        count = e->h_count[target & 3];
    } else {
        if (hipStreamSynchronize(e->stream_b) != hipSuccess) return SONDE_E_NOGPU;
        if (e->stream_e && hipStreamSynchronize(e->stream_e) != hipSuccess) return SONDE_E_NOGPU;
        if (e->call > 0 && !e->eof_pending) count = e->h_count[(e->call - 1) & 3];       // snapshot taken by the last process call
        else if (hipMemcpy(&count, e->d_fcount, sizeof count, hipMemcpyDeviceToHost) != hipSuccess) return SONDE_E_NOGPU;
        e->eof_pending = false;
        prof_collect(e);
    }
    // the per-call snapshot is older than records an end-of-stream frame sync (finish / finish_channel) has added and a fetch has already
    // taken: never step back behind what has been read
    if ((int32_t)(count - e->read_idx) < 0) count = e->read_idx;
    // a lagged fetch copies on its own stream: on B the copy would sit behind the kernels of the latest call and waiting for it would wait for
    // that call, so the host could never run ahead of the GPU.  Records up to `count` are complete (their call's event has been waited for).
    hipStream_t cs = (lag > 0 && e->stream_c) ? e->stream_c : e->stream_b;
    unsigned n = count - e->read_idx;
    if (n > (unsigned)e->max_frames) { e->overflow = true; e->read_idx = count - (unsigned)e->max_frames; n = (unsigned)e->max_frames; }
    if (max_take >= 0 && n > (unsigned)max_take) n = (unsigned)max_take;     // the rest stays queued for the next fetch
    recs.resize(n);
    if (soft) soft->resize((size_t)n * e->nbits);
    unsigned done = 0;
    while (done < n) {
        const unsigned idx = (e->read_idx + done) % (unsigned)e->max_frames;
        const unsigned run = std::min<unsigned>(n - done, (unsigned)e->max_frames - idx);
        if (hipMemcpyAsync(e->h_recs + done, e->d_frames + idx, (size_t)run * sizeof(FrameRec), hipMemcpyDeviceToHost, cs) != hipSuccess) return SONDE_E_NOGPU;
        if (soft && e->d_soft && hipMemcpyAsync(soft->data() + (size_t)done * e->nbits, e->d_soft + (size_t)idx * e->nbits,
                                                (size_t)run * e->nbits * sizeof(float), hipMemcpyDeviceToHost, cs) != hipSuccess) return SONDE_E_NOGPU;
        done += run;
    }
    if (n && hipStreamSynchronize(cs) != hipSuccess) return SONDE_E_NOGPU;
    if (n) memcpy(recs.data(), e->h_recs, (size_t)n * sizeof(FrameRec));
    e->read_idx += n;
    return (int)n;
}

static void launch_framesync_impl(sonde_engine *e, int eof);
static SyncArgs fill_sync(sonde_engine *e, int eof);
static void fill_round(sonde_engine *e, int W, WinPlanArgs &p, WinFftArgs &f);
// header position as the caller's stream counts it: from the channel's own start
static inline uint32_t rel_pos(const sonde_engine *e, const FrameRec &r) { return e->epoch.empty() ? r.mv_pos : r.mv_pos - e->epoch[r.channel]; }
extern "C" void sonde_launch_dfm_hits(const FrameRec *frames, const float *soft, int nbits, int max_frames, const unsigned *fcount, unsigned *done, sonde_dfm_frame_t *out,
                                      int ecc_level, int grid, hipStream_t s);
extern "C" void sonde_launch_m10_hits(const FrameRec *frames, const float *soft, const float *soft1, int chk3, int nbits, int max_frames, const unsigned *fcount, unsigned *done,
                                      char *chan_bits, sonde_m10_frame_t *out, int grid, hipStream_t s);
static bool blockcodes_on_device(const sonde_engine *e);
// behind every frame-sync launch: the block codes of the hits it has queued (DFM Hamming(8,4), M10 checksum), on the same stream
static void launch_blockcodes(sonde_engine *e) {
    if (!blockcodes_on_device(e)) return;
    const int grid = std::max(1, std::min(e->cfg.n_channels, 1024));
    if (e->cfg.sonde_type == SONDE_DFM09)
        sonde_launch_dfm_hits(e->d_frames, e->d_soft, e->nbits, e->max_frames, e->d_fcount, e->d_blk_done, e->d_dfm_out, e->cfg.ecc_level, grid, e->stream_b);
    else
        sonde_launch_m10_hits(e->d_frames, e->d_soft, e->d_soft1, e->m10_chk3 ? 1 : 0, e->nbits, e->max_frames, e->d_fcount, e->d_blk_done, e->d_m10_bits, e->d_m10_out, grid, e->stream_b);
}
static void launch_framesync(sonde_engine *e, int eof) { launch_framesync_impl(e, eof); launch_blockcodes(e); if (eof) e->eof_pending = true; }
static void sync_round(sonde_engine *e, int W);

static bool blockcodes_on_device(const sonde_engine *e) {
    return e->dev_ecc && e->d_soft && e->d_blk_done && ((e->cfg.sonde_type == SONDE_DFM09 && e->d_dfm_out) || (e->cfg.sonde_type == SONDE_M10 && e->d_m10_out));
}
// the decoded frames of the n records a fetch has just taken (ring slots from `start` on) -> host
template <class T> static int copy_decoded(sonde_engine *e, const T *d_out, std::vector<T> &h, unsigned start, int n, int per, int lag = 0) {
    h.resize((size_t)n * per);
    hipStream_t cs = (lag > 0 && e->stream_c) ? e->stream_c : e->stream_b;      // a lagged fetch must not queue behind the call that is still running (collect_records)
    unsigned done = 0;
    while (done < (unsigned)n) {
        const unsigned idx = (start + done) % (unsigned)e->max_frames;
        const unsigned run = std::min<unsigned>((unsigned)n - done, (unsigned)e->max_frames - idx);
        if (hipMemcpyAsync(h.data() + (size_t)done * per, d_out + (size_t)idx * per, (size_t)run * per * sizeof(T), hipMemcpyDeviceToHost, cs) != hipSuccess) return SONDE_E_NOGPU;
        done += run;
    }
    if (n && hipStreamSynchronize(cs) != hipSuccess) return SONDE_E_NOGPU;
    return 0;
}

// mixed engines: the same fetch on every group of the wanted type, channel numbers translated to the caller's
template <class F, class FN> static int mixed_fetch(sonde_engine *e, int type, F *out, int32_t max, int lag, FN fn) {
    if (!e || !out || max < 0) return SONDE_E_ARG;
    if (lag > 0 && e->snap_calls > 0) {                        // the summary snapshot of the call the fetch reaches back to is complete as well
        const int64_t target = e->call - 1 - lag;
        if (target >= 0 && target < e->snap_calls && hipEventSynchronize(e->ev_b[target & 3]) != hipSuccess) return SONDE_E_NOGPU;
    }
    int n = 0;
    for (size_t gi = 0; gi < e->groups.size(); gi++) {
        sonde_engine *g = e->groups[gi];
        if (g->cfg.sonde_type != type) continue;
        const int k = fn(g, out + n, max - n);
        if (k < 0) return k;
        for (int i = 0; i < k; i++) out[n + i].channel = e->ch_of_grp[gi][(size_t)out[n + i].channel];
        n += k;
    }
    return n;
}

extern "C" {

const char *sonde_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case SONDE_E_ARG: return "bad argument or unsupported option";
        case SONDE_E_NOGPU: return "HIP device/runtime error";
        case SONDE_E_NOMEM: return "out of memory";
        case SONDE_E_RANGE: return "chunk size out of range or not a multiple of the decimation";
        case SONDE_E_OVERFLOW: return "frame queue overflow";
        default: return "unknown error";
    }
}

extern "C" void sonde_launch_rs41_ecc_frames(FrameRec *frames, const uint32_t *list, unsigned *count, unsigned *done, int max_frames, int level,
                                             const uint8_t *gf_exp, const uint8_t *gf_log, int grid, hipStream_t s);
extern "C" void sonde_launch_rs41_ecc_batch(uint8_t *frames, const int32_t *flen, int n, int level, int32_t *ecc, int32_t *codes, uint8_t *synd,
                                            const uint8_t *gf_exp, const uint8_t *gf_log, hipStream_t s);

int sonde_rs41_ecc_device(uint8_t *frames, const int32_t *flen, int32_t n, int32_t level, int32_t *ecc, int32_t *codes, uint8_t *synd) {
    if (!frames || !flen || !ecc || n < 0 || level < 1 || level > 2) return SONDE_E_ARG;
    if (n == 0) return 0;
    for (int i = 0; i < n; i++) if (flen[i] < 0 || flen[i] > 518) return SONDE_E_ARG;
    uint8_t *d_fr = nullptr, *d_syn = nullptr, *d_gf = nullptr; int32_t *d_len = nullptr, *d_ecc = nullptr, *d_codes = nullptr;
    int rc = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == 0) { fprintf(stderr, "libsonde_hip: sonde_rs41_ecc_device: %s\n", hipGetErrorString(e)); rc = SONDE_E_NOGPU; } return rc == 0; };
    if (ok(hipMalloc((void **)&d_fr, (size_t)n * 518)) && ok(hipMalloc((void **)&d_syn, (size_t)n * 48)) && ok(hipMalloc((void **)&d_gf, 768))
        && ok(hipMalloc((void **)&d_len, (size_t)n * 4)) && ok(hipMalloc((void **)&d_ecc, (size_t)n * 4)) && ok(hipMalloc((void **)&d_codes, (size_t)n * 8))
        && ok(hipMemcpy(d_fr, frames, (size_t)n * 518, hipMemcpyHostToDevice)) && ok(hipMemcpy(d_len, flen, (size_t)n * 4, hipMemcpyHostToDevice))
        && ok(hipMemcpy(d_gf, gf_exp_table(), 512, hipMemcpyHostToDevice)) && ok(hipMemcpy(d_gf + 512, gf_log_table(), 256, hipMemcpyHostToDevice))) {
        sonde_launch_rs41_ecc_batch(d_fr, d_len, n, level, d_ecc, d_codes, d_syn, d_gf, d_gf + 512, nullptr);
        ok(hipGetLastError());
        ok(hipMemcpy(frames, d_fr, (size_t)n * 518, hipMemcpyDeviceToHost));
        ok(hipMemcpy(ecc, d_ecc, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (codes) ok(hipMemcpy(codes, d_codes, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (synd) ok(hipMemcpy(synd, d_syn, (size_t)n * 48, hipMemcpyDeviceToHost));
    }
    hipFree(d_fr); hipFree(d_syn); hipFree(d_gf); hipFree(d_len); hipFree(d_ecc); hipFree(d_codes);
    return rc;
}

int sonde_engine_create(const sonde_cfg_t *cfg, const double *fq, sonde_engine_t **out) { return sonde_engine_create_generic(cfg, fq, nullptr, out); }

static int create_impl(const sonde_cfg_t *cfg, const double *fq, const sonde_generic_t *gen, const CreateLink *lk, sonde_engine_t **out);
static int fetch_dfm_impl(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t finish, int lag);
static int fetch_m10_impl(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t finish, int lag);
int sonde_engine_create_generic(const sonde_cfg_t *cfg, const double *fq, const sonde_generic_t *gen, sonde_engine_t **out) { return create_impl(cfg, fq, gen, nullptr, out); }

}  // extern "C"

static int create_impl(const sonde_cfg_t *cfg, const double *fq, const sonde_generic_t *gen, const CreateLink *lk, sonde_engine_t **out) {
    if (!cfg || !fq || !out || cfg->abi_version != SONDE_ABI_VERSION) return SONDE_E_ARG;
    if (cfg->sonde_type == SONDE_GENERIC && !gen) return SONDE_E_ARG;
    // a description next to a PRESET type (DFM09 / M10 / M20 / RS41): only its baud is read — the decoders' --br option (dfm09mod.c:1436-1443,
    // 1590-1594; m20mod.c:1082-1089): dsp.br / dsp.sps are replaced before init_buffers(), so filters, template and slicers all follow
    float baud_override = 0.f;
    if (gen && cfg->sonde_type != SONDE_GENERIC) { if (!(gen->baud > 0.f)) return SONDE_E_ARG; baud_override = gen->baud; gen = nullptr; }
    if (gen) {
        const size_t hl = strnlen(gen->header, sizeof gen->header);
        if (hl < 8 || hl > 64 || !(gen->baud > 0.f) || !(gen->bt > 0.f) || !(gen->h > 0.f) || gen->symlen < 1 || gen->symlen > 2 || gen->symhd < 1 || gen->symhd > gen->symlen ||
            hl % gen->symhd || gen->hdmax < 0 || gen->bitofs < -8 || gen->bitofs > 64 || gen->nbits < 1 || gen->nbits > 8192 || gen->skip_bits < 0 || gen->slice_baud < 0.f) return SONDE_E_ARG;
    }
    if (cfg->n_channels < 1 || cfg->sample_rate < 1 || (cfg->bits != 16 && cfg->bits != 8 && cfg->bits != 32)) return SONDE_E_ARG;
    // every decoder of the reference turns the FM low-pass on when it runs `--IQ fq` with `--dc` (rs41mod.c:2747, dfm09mod.c:1475, m10mod.c:1320, ... — all eleven
    // have the line): an engine that stands for such a decoder does the same, whether its caller remembered or not
    sonde_cfg_t cfg_own = *cfg;
    if (cfg->input == SONDE_IN_IQ && cfg->opt_dc && cfg->sonde_type != SONDE_FRONTEND) cfg_own.opt_lp |= SONDE_LP_FM;
    cfg = &cfg_own;
    if ((cfg->sonde_type != SONDE_RS41 && cfg->sonde_type != SONDE_DFM09 && cfg->sonde_type != SONDE_M10 && cfg->sonde_type != SONDE_M20 && cfg->sonde_type != SONDE_FRONTEND && cfg->sonde_type != SONDE_GENERIC) ) return SONDE_E_ARG;
    if (cfg->opt_dc && cfg->sonde_type == SONDE_FRONTEND) return SONDE_E_ARG;
    if (cfg->opt_nolut && (cfg->opt_dc || cfg->input != SONDE_IN_IQ)) return SONDE_E_ARG;     // --noLUT folds Df into the base-rate mixer: not with --dc here
    if (cfg->if_tune && (cfg->input < SONDE_IN_IFIQ0 || cfg->bits != 32 || cfg->opt_dc)) return SONDE_E_ARG;   // fine tuning: float32 IF-rate IQ only
    if (cfg->sonde_type == SONDE_FRONTEND && cfg->input != SONDE_IN_IQ) return SONDE_E_ARG;
    if (cfg->input < SONDE_IN_IQ || cfg->input > SONDE_IN_IFIQ3) return SONDE_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device >= ndev) {
        fprintf(stderr, "libsonde_hip: no usable HIP device (the engine has no CPU fallback)\n");
        return SONDE_E_NOGPU;
    }
    HIPCHK(hipSetDevice(cfg->device));
    sonde_engine *e = new sonde_engine();
    e->cfg = *cfg;
    const int C = cfg->n_channels;
    const bool is_group = lk && lk->is_group, front_only = lk && lk->front_only;
    e->is_group = is_group; e->front_only = front_only;

    // ---- sonde preset (rs41mod.c:2591-2597,2812-2836,2882,2920-2923)
    std::string header;
    int lpiq_def, lpfm_bw;
    int skip_last = -1;
    if (gen) {                                   // any other 2-FSK sonde of the reference's demod/mod family: what its main() puts into dsp_t / passes to find_header()
        e->baud = gen->baud; e->bt = gen->bt; e->hmod = gen->h; e->symlen = gen->symlen; e->symhd = gen->symhd; e->hdmax = gen->hdmax; e->bitofs = gen->bitofs;
        e->nbits = gen->nbits; e->l_win = gen->l_win > 0.f ? gen->l_win : -1.0f; e->thres = cfg->thres > 0 ? cfg->thres : 0.7f;
        header = std::string(gen->header, strnlen(gen->header, sizeof gen->header)); lpiq_def = gen->lpiq_bw; lpfm_bw = gen->lpfm_bw;
        if (gen->skip_bits > gen->nbits) skip_last = gen->skip_bits - 1;
    } else if (cfg->sonde_type == SONDE_RS41 || cfg->sonde_type == SONDE_FRONTEND) {   // front-end only: the sync preset is never used
        e->baud = 4800.f; e->bt = 0.5f; e->hmod = 0.6f; e->symlen = 1; e->symhd = 1; e->hdmax = 4; e->bitofs = 2;
        e->nbits = 510 * 8; e->l_win = 2.0f; e->thres = cfg->thres > 0 ? cfg->thres : 0.7f;
        header = kRs41Header; lpiq_def = 7400; lpfm_bw = 6000;
    } else if (cfg->sonde_type == SONDE_M10) {   // m10mod.c:55,76,1178-1181,1370-1390,1454-1476: Manchester bits, header compared per symbol
        e->baud = 9615.f; e->bt = 1.8f; e->hmod = 0.9f; e->symlen = 2; e->symhd = 1; e->hdmax = 2; e->bitofs = 0;      // m10mod.c:1184: bitofs 0
        e->nbits = (101 + 20) * 8; e->l_win = 4.0f; e->thres = cfg->thres > 0 ? cfg->thres : 0.76f;
        header = kM10RawHeader; lpiq_def = 24000; lpfm_bw = 10000;
    } else if (cfg->sonde_type == SONDE_M20) {   // m20mod.c:60,81,86,1034-1040,1238-1251: the M10 scheme at 9600 Bd with a longer aux part
        e->baud = 9600.f; e->bt = 1.8f; e->hmod = 0.9f; e->symlen = 2; e->symhd = 1; e->hdmax = 2; e->bitofs = 0;
        e->nbits = (101 + 64) * 8; e->l_win = 4.0f; e->thres = cfg->thres > 0 ? cfg->thres : 0.76f;
        header = kM10RawHeader; lpiq_def = 24000; lpfm_bw = 10000;
    } else {   // DFM06/09 (dfm09mod.c:1309-1312,1560-1582,1690-1694): 264 + 7*280 Manchester bits per header hit
        e->baud = 2500.f; e->bt = 0.5f; e->hmod = 1.8f; e->symlen = 2; e->symhd = 2; e->hdmax = 2; e->bitofs = 2;
        e->nbits = 264 + 7 * 280; e->l_win = 4.0f; e->thres = cfg->thres > 0 ? cfg->thres : 0.65f;
        header = kDfmRawHeader; lpiq_def = 12000; lpfm_bw = 4000;
    }
    if (baud_override > 0.f) e->baud = baud_override;
    e->hdrlen = (int)header.size();
    const int lpiq_bw = cfg->lpiq_bw > 0 ? cfg->lpiq_bw : lpiq_def;

    // ---- init_buffers() arithmetic (demod_mod.c:1208-1474)
    const bool audio = cfg->input == SONDE_IN_AUDIO;
    const bool ifiq = cfg->input >= SONDE_IN_IFIQ0;           // --iq0/2/3: the input already is the IF-rate stream (f32read_csample)
    if (audio) { e->dec.if_sr = cfg->sample_rate; e->dec.decM = 1; e->l_win = -1.0f; }     // opt_iq = 0: no front-end, whole-bit slicing (rs41mod.c:2920)
    else if (ifiq) { e->dec.if_sr = cfg->sample_rate; e->dec.decM = 1; if (cfg->input != SONDE_IN_IFIQ3) e->l_win = -1.0f; }   // centre window only for opt_iq > 2
    else if (cfg->sonde_type == SONDE_FRONTEND) e->dec = design_decimator_if(cfg->sample_rate, cfg->if_rate > 0 ? cfg->if_rate : 48000, cfg->opt_min != 0);
    else e->dec = design_decimator(cfg->sample_rate, cfg->opt_min != 0);
    const int D = e->dec.decM, sr = e->dec.if_sr;
    if (D == 1) e->dec.taps.assign(1, 1.0f);                   // reference bypasses the FIR for decM == 1 (:751)
    const int T = (int)e->dec.taps.size();
    e->sps = (float)cfg->sample_rate / e->baud;
    e->sps /= (float)D;
    e->Q = (T + D - 1) / D;
    // more tap columns than the packed kernels hold (a narrow transition band at a high decimation, e.g. `iq_dec --IFbw 32` at 960 kHz: 321 taps, D = 30): the plain
    // float32 mixer / FIR kernels take any length; 16- / 8-bit samples are converted for them first (x / 32768, what f32read_cblock does)
    const bool wide = e->Q > 8;
    if (D > 1024 || (wide && (audio || ifiq || lk || cfg->input != SONDE_IN_IQ))) { delete e; return SONDE_E_ARG; }
    e->f32_path = cfg->bits == 32 || wide; e->conv32 = wide && cfg->bits != 32;
    if ((cfg->opt_lp & SONDE_LP_IQ) && !audio) {
        float f_lp = (float)(24e3 / (float)sr / 2.0);
        if (lpiq_bw) f_lp = (float)(lpiq_bw / (float)sr / 2.0);
        int taps = (int)(4 * sr / 4e3); if (taps % 2 == 0) taps++;
        e->w_iq = design_lowpass(f_lp, taps);                  // locked filter
        if (cfg->opt_dc) e->w_iq0 = design_lowpass((float)(1.5 * f_lp), taps);       // coarse acquisition (demod_mod.c:1312)
    }
    if (cfg->opt_lp & SONDE_LP_FM) {
        float f_lp = (float)(10e3 / (float)sr);
        if (lpfm_bw > 0) f_lp = lpfm_bw / (float)sr;
        int taps = (int)(4 * sr / 2e3); if (taps % 2 == 0) taps++;
        e->w_fm = design_lowpass(f_lp, taps);
    }
    e->match = design_match(header, e->sps, e->bt);
    const int L = (int)e->match.size();
    int M = 3 * L, p2 = 1;
    const int delay = L / 16;
    while (p2 < M) p2 <<= 1;
    while (p2 < 0x2000) p2 <<= 1;
    M = p2;
    const int K = M - L - delay;
    { const float nh = -e->hmod; const float hs = nh * sr; const double f1 = hs / (2.0 * e->sps); e->rho = -f1 / (double)sr; }
    e->sps_design = e->sps;
    if (gen && gen->slice_baud > 0.f) {          // a decoder that changes dsp.br / dsp.sps after init_buffers() (lms6Xmod.c:1343-1347: LMS-X at 4797.8 Bd behind
        e->sps = (float)sr / gen->slice_baud;    // filters and a header template designed for 4800): `dsp.sps = (float)dsp.sr / dsp.br` with the IF rate — bit
    }                                            // clock, slicer windows and the tone correlator's 1/sps scale follow it, everything designed above stays
    {   // samples the framer consumes behind a header before the search resumes: all nbits — M10: the rest of the second as well
        // (bits up to 5 x 808 are read and dropped, m10mod.c:1494-1507)
        const int last = skip_last >= 0 ? skip_last : ((cfg->sonde_type == SONDE_M10 || cfg->sonde_type == SONDE_M20) && !cfg->m10_noskip) ? 5 * 808 - 1 : e->nbits - 1;
        uint32_t q0, q1; double mid; bit_window(last, e->symlen - 1, e->symlen, e->sps, q0, q1, mid); e->frame_samples = q1;
    }

    const int max_if = (cfg->max_chunk + D - 1) / D;
    int ring = 1; while (ring < max_if + (int)e->frame_samples + 2 * M + 4096 || ring < 4 * M || (cfg->pipeline && ring < 2 * max_if + 4096)) ring <<= 1;
    if (lk && ring < lk->min_ring) ring = lk->min_ring;        // the parts of a mixed engine share one ring length (the y ring is the owner's)
    e->ring_len = ring;
    e->max_frames = cfg->max_frames > 0 ? cfg->max_frames : 4 * C;

    sonde_info_t &I = e->info;
    I.if_sr = sr; I.decM = D; I.dectaps = (D == 1) ? 0 : T; I.lpiq_taps = (int)e->w_iq.size(); I.lpfm_taps = (int)e->w_fm.size();
    I.L = L; I.M = M; I.K = K; I.N = M; I.delay = delay; I.sps = e->sps_design; I.ring_len = ring;

    // ---- decimator taps, front-padded to Q*D and laid out [r][q] so that step r loads its Q taps with one scalar load
    if (wide) e->wtab.assign((size_t)std::max(64, D) * 8, 0.f);      // (not used: the float32 kernels read the taps as they are, d_taps_f)
    else {
        const int pad = e->Q * D - T;
        std::vector<float> wpad((size_t)e->Q * D, 0.f);
        for (int k = 0; k < T; k++) wpad[pad + k] = e->dec.taps[k];
        e->wtab.assign((size_t)std::max(64, D) * 8, 0.f);
        for (int r = 0; r < D; r++) for (int q = 0; q < e->Q; q++) e->wtab[(size_t)r * 8 + q] = wpad[(size_t)D * q + r];
        if (D > 64) {                                         // wide decimation (k_mix_decimate_wide): taps in global memory
            for (int k = 64; k >= 4; k--) if (D % k == 0) { e->DS = k; break; }
            if (!e->DS || e->Q < 5) { delete e; return SONDE_E_ARG; }
        }
        // the tap rows in global memory too: the wide variant and the hand-scheduled D = 50 stream fetch them with scalar loads
        // (for D <= 64 a second copy follows, scaled by 2^-15 — exact — which takes over the 1/32768 of the int16 samples)
        {
            std::vector<float> both(e->wtab);
            if (D <= 64) for (size_t i = 0; i < e->wtab.size(); i++) both.push_back(e->wtab[i] * 3.0517578125e-05f);
            if (dalloc(&e->d_wtab, both.size(), false)) { delete e; return SONDE_E_NOMEM; }
            HIPCHK(hipMemcpy(e->d_wtab, both.data(), both.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    // ---- mixer: snapped frequency per channel (xlt_fq = -fq, rs41mod.c:2685); the table period is common
    {
        std::vector<double> f0s(C);
        for (int c = 0; c < C; c++) {
            const Mixer m = design_mixer(-std::max(-0.5, std::min(0.5, fq[c])), cfg->sample_rate);
            f0s[c] = m.f0; e->lut_len = m.lut_len;
            if (cfg->opt_nolut || cfg->if_tune) f0s[c] = -std::max(-0.5, std::min(0.5, fq[c]));   // xlt_fq itself, not the table's snapped value
        }
        I.lut_len = e->lut_len;
        if (dalloc(&e->d_chanf0, (size_t)C, false)) { delete e; return SONDE_E_NOMEM; }
        HIPCHK(hipMemcpy(e->d_chanf0, f0s.data(), f0s.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // ---- 2.4 Msps -> 48 kHz class (D = 50, Q = 7, float table phase, rows aligned with the mixer table): the hand-scheduled
    // decimator, which subtracts the IQ-DC mean per output as avg * E — E tabulated here, once (k_md_etable)
    {
        static const bool no_fast = getenv("SONDE_MD_NOFAST") != nullptr;       // A/B aid: the compiler-scheduled kernels
        if (D == 50 && e->Q == 7 && !audio && cfg->bits != 32 && cfg->sonde_type != SONDE_FRONTEND && !cfg->opt_nolut && e->lut_len % D == 0 && !no_fast
            && cfg->input == SONDE_IN_IQ && !is_group) {
            e->etab_len = e->lut_len / D;
            if (dalloc(&e->d_etab, (size_t)C * e->etab_len, false) || dalloc(&e->d_dcavg_prev, C)) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
            sonde_launch_md_etable(e->d_chanf0, e->d_wtab, D, e->Q, e->etab_len, C, e->d_etab, nullptr);
            HIPCHK(hipDeviceSynchronize());
        }
    }
    int bad = 0;
    bad |= dalloc(&e->d_dcavg, C); bad |= dalloc(&e->d_dcsums, 2 * (size_t)C);
    bad |= dalloc(&e->d_ptail[0], (size_t)C * 64); bad |= dalloc(&e->d_ptail[1], (size_t)C * 64);
    if (audio) bad |= dalloc(&e->d_raw, (size_t)C * ring);
    else { if (!is_group) bad |= dalloc(&e->d_y, (size_t)C * ring); if (!front_only) bad |= dalloc(&e->d_ifiq, (size_t)C * ring); }
    if (!front_only) { bad |= dalloc(&e->d_fm, (size_t)C * ring); bad |= dalloc(&e->d_bufs, (size_t)C * ring); bad |= dalloc(&e->d_corr, (size_t)C * ring); }
    bad |= dalloc(&e->d_state, C); bad |= dalloc(&e->d_frames, e->max_frames); bad |= dalloc(&e->d_fcount, 1 + 4);      // the counter, and its value behind each of the last four calls (what a call publishes)
    if (cfg->keep_soft || cfg->sonde_type != SONDE_RS41) bad |= dalloc(&e->d_soft, (size_t)e->max_frames * e->nbits);
    if (cfg->keep_soft == 2) bad |= dalloc(&e->d_soft1, (size_t)e->max_frames * e->nbits);
    if (cfg->sonde_type == SONDE_DFM09) { bad |= dalloc(&e->d_dfm_out, (size_t)e->max_frames * 8); bad |= dalloc(&e->d_blk_done, 2); }
    if (cfg->sonde_type == SONDE_M10) { bad |= dalloc(&e->d_m10_out, (size_t)e->max_frames); bad |= dalloc(&e->d_blk_done, 2); bad |= dalloc(&e->d_m10_bits, (size_t)C * ((101 + 20) * 8 + 8)); }
    bad |= dalloc(&e->d_match, L, false);
    if (!e->w_iq.empty()) bad |= dalloc(&e->d_wiq, e->w_iq.size(), false);
    if (!e->w_fm.empty()) bad |= dalloc(&e->d_wfm, e->w_fm.size(), false);
    bad |= dalloc(&e->d_consts, 1024, true);
    if (bad) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
    HIPCHK(hipMemcpy(e->d_match, e->match.data(), L * sizeof(float), hipMemcpyHostToDevice));
    e->dev_ecc = getenv("SONDE_HOST_ECC") == nullptr;
    {   // SONDE_SMALL_TAIL=1: header search and frame sync in workgroups of the size of one decimator workgroup (k_sync_window_fft_h, k_framesync<.., 256>).
        // With them the whole IF-rate tail does run beside the next call's decimator — and the decimator slows down by as much as the tail takes
        // (profiles/r4k_*: 1.35 ms per step either way), so the plain forms stay the default; kept as the measurement's other arm, parity-tested
        const char *st = getenv("SONDE_SMALL_TAIL");
        e->small_tail = st && atoi(st) != 0;
    }
    {   // Fm = rdft(time-reversed match) with the reference's transform (init_buffers, demod_mod.c:1446-1449); N = 8192 only
        static const bool no_fft = getenv("SONDE_NO_FFTSYNC") != nullptr;        // A/B aid: time-domain correlation ring
        if (!cfg->opt_dc && M == 8192 && K + L <= M && !no_fft) {
            std::vector<float> m(2 * (size_t)M, 0.f);
            for (int i = 0; i < L; i++) m[2 * (size_t)(L - 1 - i)] = e->match[i];
            ref_dft_8192(m);
            const std::vector<float> tw = ref_twiddle_table();
            e->win_W = 8;
            m.resize(4 * (size_t)M);                          // behind the table its bit-reversed copy (the conjugate-and-swap pass reads both coalesced)
            for (int i = 0; i < M; i++) {
                int r = 0; for (int b = 0; b < 13; b++) if (i >> b & 1) r |= 1 << (12 - b);
                m[2 * (size_t)(M + i)] = m[2 * (size_t)r]; m[2 * (size_t)(M + i) + 1] = m[2 * (size_t)r + 1];
            }
            if (dalloc(&e->d_Fm, 2 * (size_t)M, false) || dalloc(&e->d_tws, tw.size() / 2, false) || dalloc(&e->d_win, (size_t)C * e->win_W) ||
                dalloc(&e->d_work, (size_t)C * e->win_W) || dalloc(&e->d_work_count, 2)) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
            HIPCHK(hipMemcpy(e->d_Fm, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_tws, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (e->d_wiq) HIPCHK(hipMemcpy(e->d_wiq, e->w_iq.data(), e->w_iq.size() * sizeof(float), hipMemcpyHostToDevice));
    if (e->d_wfm) HIPCHK(hipMemcpy(e->d_wfm, e->w_fm.data(), e->w_fm.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        uint8_t cb[1024]; memset(cb, 0, sizeof cb);
        memcpy(cb, header.data(), header.size()); memcpy(cb + 64, kRs41HeaderBytes, 8); memcpy(cb + 72, kRs41Mask, 64);
        memcpy(cb + 136, gf_exp_table(), 512); memcpy(cb + 648, gf_log_table(), 256);
        HIPCHK(hipMemcpy(e->d_consts, cb, sizeof cb, hipMemcpyHostToDevice));
    }
    // ---- factorised header correlation tables (integer samples/symbol only; else the direct L-tap kernel runs)
    {
        const int isps = (int)e->sps_design, nsym = e->hdrlen / e->symhd;
        if ((float)isps == e->sps_design && e->symhd == 1 && isps * nsym == L && isps >= 2 && isps <= 16 && isps % 2 == 0) {
            std::vector<float> shapes; std::vector<int> type(nsym); std::vector<float> sign(nsym);
            std::vector<int> key;                                     // (left, right) neighbour relative to the own bit
            bool ok = true;
            for (int k = 0; k < nsym && ok; k++) {
                const int b = (header[k] & 1) ? 1 : -1;
                const int l = (k > 0) ? ((header[k - 1] & 1) ? 1 : -1) * b : 0;
                const int r = (k < nsym - 1) ? ((header[k + 1] & 1) ? 1 : -1) * b : 0;
                const int ky = (l + 1) * 3 + (r + 1);
                int t = -1;
                for (size_t q = 0; q < key.size(); q++) if (key[q] == ky) t = (int)q;
                if (t < 0) {
                    t = (int)key.size(); key.push_back(ky);
                    for (int d = 0; d < isps; d++) shapes.push_back((float)b * e->match[(size_t)isps * k + d]);
                } else {
                    for (int d = 0; d < isps; d++) ok &= (shapes[(size_t)t * isps + d] == (float)b * e->match[(size_t)isps * k + d]);
                }
                type[k] = t; sign[k] = (float)b;
            }
            if (ok && key.size() <= 9) {
                e->corr_types = (int)key.size(); e->corr_isps = isps;
                bad = 0;
                bad |= dalloc(&e->d_shapes, shapes.size(), false); bad |= dalloc(&e->d_symtype, nsym, false); bad |= dalloc(&e->d_symsign, nsym, false);
                if (bad) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
                HIPCHK(hipMemcpy(e->d_shapes, shapes.data(), shapes.size() * sizeof(float), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_symtype, type.data(), nsym * sizeof(int), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_symsign, sign.data(), nsym * sizeof(float), hipMemcpyHostToDevice));
            }
        }
    }
    // ---- per-bit slicing ranges (position independent): read_softbit2p's double-edge / integer-counter walk, tabulated
    {
        std::vector<uint32_t> win((size_t)e->nbits * 4), end(e->nbits);
        for (int bp = 0; bp < e->nbits; bp++) {
            uint32_t q0, q1, qa, qb; double mid;
            if (e->symlen == 2) { bit_window(bp, 0, 2, e->sps, q0, q1, mid); slice_range(q0, q1, mid, e->l_win, qa, qb); }
            else { qa = qb = 0; }
            win[4 * bp] = qa; win[4 * bp + 1] = qb;
            bit_window(bp, e->symlen - 1, e->symlen, e->sps, q0, q1, mid);
            slice_range(q0, q1, mid, e->l_win, qa, qb);
            win[4 * bp + 2] = qa; win[4 * bp + 3] = qb; end[bp] = q1;
        }
        bad = 0;
        bad |= dalloc(&e->d_bitwin, (size_t)e->nbits, false); bad |= dalloc(&e->d_bitend, (size_t)e->nbits, false);
        if (bad) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
        HIPCHK(hipMemcpy(e->d_bitwin, win.data(), win.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_bitend, end.data(), end.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // IQ-DC segment schedule (demod_mod.c:1351-1357)
    e->dc_lim = (uint32_t)sr; e->dc_max = e->dc_lim / 32;
    if (D > 1) { e->dc_lim *= D; e->dc_max *= D; }
    if (e->dc_max == 0 || e->dc_max % D) { sonde_engine_destroy(e); return SONDE_E_ARG; }
    e->dc_max0 = e->dc_max;
    e->ifiq = ifiq;
    if (e->f32_path) {
        bad = dalloc(&e->d_dcsums_f, 2 * (size_t)C);
        if (e->conv32) bad |= dalloc(&e->d_conv32, (size_t)C * (size_t)cfg->max_chunk, false);
        if (!audio && !ifiq) {
            uint32_t zl = 1; while (zl < (uint32_t)cfg->max_chunk + (uint32_t)T + (uint32_t)D + 64u) zl <<= 1;
            e->zmask = zl - 1;
            bad |= dalloc(&e->d_zring, (size_t)C * zl); bad |= dalloc(&e->d_taps_f, (size_t)T, false);
            if (!bad) HIPCHK(hipMemcpy(e->d_taps_f, e->dec.taps.data(), (size_t)T * sizeof(float), hipMemcpyHostToDevice));
        }
        if (bad) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
    }
    if (cfg->opt_inv) {                                       // -i: every channel starts with inverted polarity
        std::vector<SyncState> st0(C); for (auto &q : st0) { memset(&q, 0, sizeof q); q.inv = 1; }
        HIPCHK(hipMemcpy(e->d_state, st0.data(), st0.size() * sizeof(SyncState), hipMemcpyHostToDevice));
    }
    e->opt_iq = audio ? 0 : ifiq ? (cfg->input == SONDE_IN_IFIQ0 ? 1 : cfg->input == SONDE_IN_IFIQ2 ? 2 : 3) : 5;
    { double sm = 0.0; for (float v : e->match) sm += (double)v; e->match_sum = (float)sm; }
    if (cfg->opt_dc) {
        bad = 0;
        bad |= dalloc(&e->d_afc, C); bad |= dalloc(&e->d_start, C); bad |= dalloc(&e->d_pending, 1);
        if (!audio) { bad |= dalloc(&e->d_yrot, (size_t)C * ring); bad |= dalloc(&e->d_fmraw, (size_t)C * ring); }
        if (e->opt_iq >= 2) bad |= dalloc(&e->d_corr2, (size_t)C * ring);
        if (!e->w_iq0.empty()) bad |= dalloc(&e->d_wiq0, e->w_iq0.size(), false);
        if (bad) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
        if (e->d_wiq0) HIPCHK(hipMemcpy(e->d_wiq0, e->w_iq0.data(), e->w_iq0.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipHostMalloc((void **)&e->h_pending, sizeof(unsigned), hipHostMallocDefault));
    }
    e->last_frame.assign((size_t)C * 518, 0);
    for (int c = 0; c < C; c++) memcpy(e->last_frame.data() + (size_t)c * 518, kRs41HeaderBytes, 8);
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // pipeline: stream A (staging, decimator) may run ONE call ahead of stream B (IF-rate kernels); the rings hold that (see
    // process_device).  The FM-audio path writes the rings B reads from on stream A, so it does not pipeline.
    if (cfg->pipeline && audio) { sonde_engine_destroy(e); return SONDE_E_ARG; }
    if (lk && lk->shared_b) { e->stream_b = lk->shared_b; e->borrowed_b = is_group; }
    else if (cfg->pipeline) {
        // the IF-rate tail of call k runs beside the decimator of call k+1: few, latency-bound workgroups against thousands of bandwidth-bound
        // ones — B gets the higher dispatch priority so that its workgroups take the next free CU slot instead of queueing behind the decimator's
        int lo = 0, hi = 0;
        const char *pr = getenv("SONDE_B_PRIO");                    // A/B aid: 0 = default priority, 1 (default) = highest
        if ((!pr || atoi(pr) != 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            HIPCHK(hipStreamCreateWithPriority(&e->stream_b, hipStreamNonBlocking, hi));
        else HIPCHK(hipStreamCreateWithFlags(&e->stream_b, hipStreamNonBlocking));
    }
    else e->stream_b = e->stream;              // one in-order stream: no cross-stream events needed
    HIPCHK(hipStreamCreateWithFlags(&e->stream_c, hipStreamNonBlocking));
    if (cfg->sonde_type == SONDE_RS41 && cfg->ecc_level >= 1 && cfg->ecc_level <= 2 && getenv("SONDE_ECC_INLINE") == nullptr) {
        // the Reed-Solomon decoder of a call's damaged frames runs on a stream of its own (highest priority: a few hundred four-wave workgroups that
        // should take the next free slot) so that it overlaps the next call's decimator; SONDE_ECC_INLINE=1 keeps it on stream B (A/B aid)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) HIPCHK(hipStreamCreateWithPriority(&e->stream_e, hipStreamNonBlocking, hi));
        else HIPCHK(hipStreamCreateWithFlags(&e->stream_e, hipStreamNonBlocking));
    }
    HIPCHK(hipEventCreateWithFlags(&e->ev_s, hipEventDisableTiming));
    if (cfg->sonde_type == SONDE_RS41) {
        HIPCHK(hipMalloc((void **)&e->d_ecc_list, 2 * (size_t)e->max_frames * sizeof(uint32_t)));
        HIPCHK(hipMalloc((void **)&e->d_ecc_cnt, 4 * sizeof(unsigned)));
        HIPCHK(hipMemset(e->d_ecc_cnt, 0, 4 * sizeof(unsigned)));
    }
    for (int i = 0; i < 4; i++) { HIPCHK(hipEventCreateWithFlags(&e->ev_a[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_b[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_if[i], hipEventDisableTiming)); }
    HIPCHK(hipEventCreateWithFlags(&e->ev_copy, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void **)&e->h_count, 4 * sizeof(unsigned), hipHostMallocMapped));
    memset(e->h_count, 0, 4 * sizeof(unsigned));
    HIPCHK(hipHostGetDevicePointer((void **)&e->h_count_dev, e->h_count, 0));
    HIPCHK(hipHostMalloc((void **)&e->h_recs, (size_t)e->max_frames * sizeof(FrameRec), hipHostMallocDefault));
    *out = e;
    return 0;
}

extern "C" {

void sonde_engine_destroy(sonde_engine_t *e) {
    if (!e) return;
    for (sonde_engine *g : e->groups) sonde_engine_destroy(g);      // (they borrow rows of this object's y ring)
    e->groups.clear();
    if (e->is_group) e->d_y = nullptr;
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->stream_b) hipStreamSynchronize(e->stream_b);
    if (e->stream_e) hipStreamSynchronize(e->stream_e);
    prof_collect(e);
    if (e->d_wfprof) {
        unsigned long long h[32] = {0};
        if (hipMemcpy(h, e->d_wfprof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[31]) {
            static const char *nm[7] = {"setup", "search", "headcmp", "slot", "slice", "syndromes+record", "exit"};
            unsigned long long tot = 0; for (int i = 0; i < 7; i++) tot += h[16 + i];
            fprintf(stderr, "framesync prof (channel 0, %llu launches, %llu frames, %.0f cycles per launch):", h[31], h[30], (double)tot / (double)h[31]);
            for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h[16 + i] / (double)std::max(1ull, tot));
            fprintf(stderr, "\n");
        }
        if (h[15]) {
            static const char *nm[6] = {"load", "fft1", "mul", "fft2", "argmax", "norm"};
            unsigned long long tot = 0; for (int i = 0; i < 6; i++) tot += h[i];
            fprintf(stderr, "window fft prof (workgroup 0, %llu windows, %.0f cycles each):", h[15], (double)tot / (double)h[15]);
            for (int i = 0; i < 6; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h[i] / (double)std::max(1ull, tot));
            fprintf(stderr, "\n");
        }
        hipFree(e->d_wfprof);
    }
    { hipStream_t sa = e->stream; if (sa) hipStreamDestroy(sa); }
    if (e->stream_b && e->stream_b != e->stream && !e->borrowed_b) hipStreamDestroy(e->stream_b);
    if (e->stream_c) hipStreamDestroy(e->stream_c);
    if (e->stream_e) hipStreamDestroy(e->stream_e);
    if (e->ev_s) hipEventDestroy(e->ev_s);
    hipFree(e->d_dfm_out); hipFree(e->d_m10_out); hipFree(e->d_blk_done); hipFree(e->d_m10_bits);
    hipFree(e->d_ecc_list); hipFree(e->d_ecc_cnt); hipFree(e->d_park);
    for (int i = 0; i < 4; i++) { if (e->ev_a[i]) hipEventDestroy(e->ev_a[i]); if (e->ev_b[i]) hipEventDestroy(e->ev_b[i]); if (e->ev_if[i]) hipEventDestroy(e->ev_if[i]); }
    if (e->ev_copy) hipEventDestroy(e->ev_copy);
    if (e->h_pending) hipHostFree(e->h_pending);
    if (e->h_count) hipHostFree(e->h_count);
    if (e->h_recs) hipHostFree(e->h_recs);
    void *ptrs[] = { e->d_Bop, e->d_chanf0, e->d_dcavg, e->d_dcsums, e->d_ptail[0], e->d_ptail[1], e->d_y, e->d_ifiq, e->d_fm,
                     e->d_bufs, e->d_corr, e->d_wiq, e->d_wfm, e->d_match, e->d_state, e->d_frames, e->d_fcount, e->d_soft, e->d_soft1,
                     e->d_epoch, e->d_work, e->d_work_count, e->d_consts, e->d_stage, e->d_shapes, e->d_symtype, e->d_symsign, e->d_bitwin, e->d_bitend, e->d_raw, e->d_wtab, e->d_conv, e->d_conv32,
                     e->d_dcsums_f, e->d_zring, e->d_taps_f, e->d_wiq0, e->d_yrot, e->d_fmraw, e->d_corr2, e->d_afc, e->d_start, e->d_pending,
                     e->d_etab, e->d_dcavg_prev, e->d_win, e->d_Fm, e->d_tws, e->d_epoch_phase, e->d_pcs_cnt, e->d_pcs_max, e->d_pcs_since, e->d_in_row, e->d_sum_map };
    for (void *p : ptrs) if (p) hipFree(p);
    delete e;
}

int sonde_engine_info(const sonde_engine_t *e, sonde_info_t *info) {
    if (!e || !info) return SONDE_E_ARG;
    *info = e->info;
    return 0;
}

static uint32_t pcs_room(const sonde_engine *e) {             // samples until the first channel reaches the end of its IQ-DC segment
    uint32_t r = 0xffffffffu;
    for (size_t c = 0; c < e->pcs_cnt.size(); c++) r = std::min(r, e->pcs_max[c] - e->pcs_cnt[c]);
    return r;
}
int64_t sonde_engine_samples_to_dc_boundary(const sonde_engine_t *e) {
    if (!e) return SONDE_E_ARG;
    return e->pcs ? (int64_t)pcs_room(e) : (int64_t)(e->dc_max - e->dc_cnt);
}

void *sonde_engine_stream(sonde_engine_t *e) { return e ? (void *)e->stream : nullptr; }

}  // extern "C"

// What a process call does before its first kernel, for every engine `t` that runs an IF-rate tail in it (the engine itself, or each group of a mixed
// engine); fs = the stream the call's decimator runs on.
static void tail_begin(sonde_engine *t, hipStream_t fs) {
    t->in_call = true; t->ecc_listed = false;
    // this call's frame syncs append to the work list call-2 used: its decoder kernel (stream E) must be through
    if (t->stream_e && t->call >= 2) hipStreamWaitEvent(t->stream_b, t->ev_b[(t->call - 2) & 3], 0);
    // ... and the list starts empty whatever that call left (a call that failed half way never ran the decoder that clears it: ADVICE round 4)
    if (t->d_ecc_cnt) { hipMemsetAsync(t->d_ecc_cnt + (t->call & 1), 0, sizeof(unsigned), t->stream_b); hipMemsetAsync(t->d_ecc_cnt + 2 + (t->call & 1), 0, sizeof(unsigned), t->stream_b); }
    // two streams: this call's decimator overwrites the part of the y ring that call-2 occupied (ring_len >= 2 * max_if + history), so it must
    // not start before the IF chain of call-2 has read it.  Only the IF chain reads y: the header search and the frame sync behind it work on
    // rings stream B writes itself, so however late they run (they wait for CU slots the decimator frees) the decimators stay back to back.
    static const bool wait_tail = getenv("SONDE_A_WAITS_TAIL") != nullptr;       // A/B aid: the round-3 dependency on the whole tail of call-2
    if (t->stream_b != fs && t->call >= 2) hipStreamWaitEvent(fs, wait_tail ? t->ev_b[(t->call - 2) & 3] : t->ev_if[(t->call - 2) & 3], 0);
}
static int tail_enqueue(sonde_engine *e, int32_t n_samples, uint32_t m_first, hipStream_t fs, hipEvent_t front_done);
static int tail_enqueue_merged(sonde_engine *e, int32_t n_samples, uint32_t m_first);
static int tail_finish(sonde_engine *e);
static int sync_rounds_of(const sonde_engine *e, int n_if);
static IfArgs fill_if(sonde_engine *e, int n_if, uint32_t m_first);

extern "C" {

int sonde_engine_process_device(sonde_engine_t *e, const void *d_iq, int64_t ch_stride, int32_t n_samples) {
    if (!e || !d_iq || e->is_group) return SONDE_E_ARG;        // (a group of a mixed engine is driven by its owner)
    const int D = e->info.decM, C = e->cfg.n_channels;
    // ch_stride == 0: one wideband stream shared by all channels (each mixes its own fq out of it)
    if (n_samples <= 0 || n_samples > e->cfg.max_chunk || n_samples % D || (ch_stride != 0 && ch_stride < n_samples)) return SONDE_E_RANGE;
    if (e->cfg.bits == 8) {                        // cu8: (u-128)/128 == ((u-128)*256)/32768 -> feed the 16-bit path
        const int epf = e->cfg.input == SONDE_IN_AUDIO ? std::max(1, e->cfg.audio_channels) : 2;    // bytes per sample / audio frame
        if (!e->d_conv) HIPCHK(hipMalloc((void **)&e->d_conv, (size_t)C * e->cfg.max_chunk * epf * 2));
        sonde_launch_u8_to_s16((const uint8_t *)d_iq, ch_stride * epf, e->d_conv, (long long)n_samples * epf, ch_stride == 0 ? 1 : C, n_samples * epf, e->stream);
        d_iq = e->d_conv; if (ch_stride != 0) ch_stride = n_samples;
    }
    const uint32_t m_first = e->m_out;
    int done = 0;
    const bool mixed = !e->groups.empty();
    if (mixed) { for (sonde_engine *g : e->groups) tail_begin(g, e->stream); }
    else tail_begin(e, e->stream);
    if (e->cfg.input == SONDE_IN_AUDIO) {
        // f32read_sample (demod_mod.c:379-405): b/128/256 of the selected channel; then FM low-pass / bufs
        AudioConvArgs c0{}; c0.pcm = (const int16_t *)d_iq; c0.ch_stride = ch_stride; c0.n_ch = C; c0.n = n_samples;
        c0.nch = std::max(1, e->cfg.audio_channels); c0.sel = std::min(std::max(0, e->cfg.audio_select), c0.nch - 1);
        c0.fm = e->d_raw; c0.ring_len = e->ring_len; c0.m0 = e->m_out; c0.f32 = (e->cfg.bits == 32);
        sonde_launch_audio_convert(&c0, e->stream);
        AudioChainArgs c1{}; c1.raw = e->d_raw; c1.fm = e->d_fm; c1.bufs = e->d_bufs; c1.w = e->d_wfm; c1.n_ch = C; c1.ring_len = e->ring_len;
        c1.n = n_samples; c1.taps = (int)e->w_fm.size(); c1.m0 = e->m_out;
        prof_begin(e, "if_chain", e->stream); sonde_launch_audio_chain(&c1, e->stream); prof_end(e, e->stream);
        e->samples_in += (uint64_t)n_samples; e->m_out += (uint32_t)n_samples; done = n_samples;
    }
    if (e->conv32) {
        sonde_launch_s16_to_f32((const int16_t *)d_iq, ch_stride, e->d_conv32, (long long)n_samples, ch_stride == 0 ? 1 : C, n_samples, e->stream);
        d_iq = e->d_conv32; if (ch_stride != 0) ch_stride = n_samples;
    }
    while (done < n_samples && e->f32_path && e->cfg.input != SONDE_IN_AUDIO) {
        // float32 IQ (cf32): plain mixer + FIR kernels (MixF32Args); the IQ-DC schedule is the same, sums in double like the reference
        const bool dc = !e->ifiq || e->cfg.opt_iqdc != 0;
        const int take = dc ? (int)std::min<uint32_t>((uint32_t)(n_samples - done), e->dc_max - e->dc_cnt) : n_samples - done;
        MixF32Args a{}; a.x = (const float2 *)d_iq + (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.n = take;
        a.chan_f0 = e->d_chanf0; a.lut_len = e->lut_len; a.lut_phase = (uint32_t)(e->samples_in % (uint64_t)e->lut_len);
        a.phase_f64 = (e->cfg.sonde_type == SONDE_FRONTEND); a.dc_avg = e->d_dcavg; a.dc_sums = e->d_dcsums_f;
        if (e->cfg.opt_nolut) { a.phase_f64 = 1; a.lut_len = 1 << 30; a.lut_phase = 0; a.nd_base = (double)e->samples_in; }
        if (e->ifiq) {
            a.mix = 0; a.z = e->d_y; a.zmask = (uint32_t)e->ring_len - 1; a.n0 = e->m_out;
            if (e->cfg.if_tune) {                    // channelizer output: rotate by -fq[c] with the exact phase of the channel's own sample count
                a.mix = 1; a.phase_f64 = 1; a.lut_len = 1 << 30; a.lut_phase = 0; a.nd_base = (double)e->samples_in; a.epoch = e->d_epoch;
            }
        }
        else { a.mix = 1; a.z = e->d_zring; a.zmask = e->zmask; a.n0 = e->samples_in; }
        prof_begin(e, "mix_decimate", e->stream);
        sonde_launch_mix_f32(&a, e->stream);
        if (!e->ifiq) {
            DecF32Args d{}; d.z = e->d_zring; d.zmask = e->zmask; d.n0 = e->samples_in; d.taps = e->d_taps_f; d.T = (int)e->dec.taps.size(); d.D = D;
            d.n_ch = C; d.nblocks = take / D; d.y = e->d_y; d.ring_len = e->ring_len; d.m0 = e->m_out;
            sonde_launch_decimate_f32(&d, e->stream);
        }
        prof_end(e, e->stream);
        e->samples_in += (uint64_t)take; e->m_out += (uint32_t)(take / D); done += take;
        if (dc) {
            e->dc_cnt += (uint32_t)take;
            if (e->dc_cnt == e->dc_max) {
                sonde_launch_dc_update_f64(C, e->d_dcsums_f, e->d_dcavg, (float)e->dc_max, e->stream);
                e->dc_cnt = 0;
                if (e->dc_max < e->dc_lim) e->dc_max *= 2;
            }
        }
    }
    while (done < n_samples && e->ifiq) {
        // --iq0/2/3 (f32read_csample, demod_mod.c:419-461): convert, optionally minus the running mean of the previous segment
        const bool dc = e->cfg.opt_iqdc != 0;
        const int take = dc ? (int)std::min<uint32_t>((uint32_t)(n_samples - done), e->dc_max - e->dc_cnt) : n_samples - done;
        IqConvArgs a{}; a.iq = (const int16_t *)d_iq + 2 * (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.n = take;
        a.dc_avg = e->d_dcavg; a.dc_sums = e->d_dcsums; a.y = e->d_y; a.ring_len = e->ring_len; a.m0 = e->m_out;
        prof_begin(e, "iq_convert", e->stream); sonde_launch_iq_convert(&a, e->stream); prof_end(e, e->stream);
        e->samples_in += (uint64_t)take; e->m_out += (uint32_t)take; done += take;
        if (dc) {
            e->dc_cnt += (uint32_t)take;
            if (e->dc_cnt == e->dc_max) {
                sonde_launch_dc_update(C, e->d_dcsums, e->d_dcavg, (float)e->dc_max, e->stream);
                e->dc_cnt = 0;
                if (e->dc_max < e->dc_lim) e->dc_max *= 2;
            }
        }
    }
    while (done < n_samples) {
        // never straddle an IQ-DC segment: the mean of segment s-1 is subtracted throughout segment s
        const int take = (int)std::min<uint32_t>((uint32_t)(n_samples - done), e->pcs ? pcs_room(e) : e->dc_max - e->dc_cnt);
        MixDecArgs a{};
        a.iq = (const int16_t *)d_iq + 2 * (size_t)done; a.ch_stride = ch_stride; a.n_ch = C; a.nblocks = take / D;
        a.D = D; a.Q = e->Q; a.G = e->G;
        memcpy(a.wtab, e->wtab.data(), sizeof a.wtab); a.wtab_g = e->d_wtab; a.DS = e->DS; a.chan_f0 = e->d_chanf0; a.lut_len = e->lut_len;
        a.lut_phase = (uint32_t)(e->samples_in % (uint64_t)e->lut_len);
        a.dc_avg = e->d_dcavg; a.dc_sums = e->d_dcsums;
        a.ptail_in = e->d_ptail[e->ptail_cur]; a.ptail_out = e->d_ptail[e->ptail_cur ^ 1];
        a.y = e->d_y; a.ring_len = e->ring_len; a.m0 = e->m_out;
        a.phase_f64 = (e->cfg.sonde_type == SONDE_FRONTEND);     // iq_dec.c:690 builds its table from a double phase
        if (e->cfg.opt_nolut) { a.phase_f64 = 1; a.lut_len = 1 << 30; a.lut_phase = 0; a.nd_base = (double)e->samples_in; }
        // enough waves to fill the chip, few enough that the one-tile halo per wave stays small
        { long long tiles = (long long)C * ((a.nblocks + 63) / 64); int G = (int)(tiles / 12288); a.G = G < 1 ? 1 : (G > 16 ? 16 : G); }
        {   // test aid: tiles per wave of a large batch (512 channels x 1 s: G = 16) on a batch small enough for a parity test (tests/test_gpu_batch.py)
            const char *g = getenv("SONDE_MD_G"); const int gv = g ? atoi(g) : 0; if (gv >= 1 && gv <= 16) a.G = gv;
        }
        a.etab = e->d_etab; a.etab_len = e->etab_len; a.in_row = e->d_in_row;
        a.dc_avg_prev = (e->d_etab && e->dc_since < e->Q - 1) ? e->d_dcavg_prev : nullptr; a.dc_since = e->dc_since;
        if (e->dc_since < (1 << 20)) e->dc_since += take / D;
        if (e->pcs) { a.epoch_phase = e->d_epoch_phase; a.dc_since_ch = e->d_pcs_since; a.dc_avg_prev = e->d_etab ? e->d_dcavg_prev : nullptr; }
        prof_begin(e, "mix_decimate", e->stream); const int lrc = sonde_launch_mix_decimate(&a, e->stream); prof_end(e, e->stream);
        if (lrc < 0) { e->in_call = false; e->ecc_listed = false; for (sonde_engine *g : e->groups) { g->in_call = false; g->ecc_listed = false; } return SONDE_E_ARG; }
        e->ptail_cur ^= 1;
        e->samples_in += (uint64_t)take; e->m_out += (uint32_t)(take / D); e->dc_cnt += (uint32_t)take; done += take;
        if (e->pcs) {                                    // per-channel segment edges: the device keeps the counters, the host mirrors them
            sonde_launch_dc_update_pcs(C, e->d_dcsums, e->d_dcavg, e->d_dcavg_prev, e->d_pcs_cnt, e->d_pcs_max, e->dc_lim, e->d_pcs_since, (uint32_t)take, take / D, e->stream);
            for (int c2 = 0; c2 < C; c2++) {
                e->pcs_cnt[c2] += (uint32_t)take;
                if (e->pcs_cnt[c2] >= e->pcs_max[c2]) { e->pcs_cnt[c2] = 0; if (e->pcs_max[c2] < e->dc_lim) e->pcs_max[c2] *= 2; }
            }
            e->dc_cnt = 0;
            continue;
        }
        if (e->dc_cnt == e->dc_max) {
            if (e->d_etab) e->dc_since = 0;
            sonde_launch_dc_update_keep(C, e->d_dcsums, e->d_dcavg, e->d_etab ? e->d_dcavg_prev : nullptr, (float)e->dc_max, e->stream);
            e->dc_cnt = 0;
            if (e->dc_max < e->dc_lim) e->dc_max *= 2;
        }
    }
    // the IF-rate work of the call: behind this call's decimator, on the stream(s) B of the engine — or of every group of a mixed engine
    const int slot = (int)(e->call & 3);
    int rc = 0;
    if (mixed) {
        hipEventRecord(e->ev_a[slot], e->stream);
        for (sonde_engine *g : e->groups) { g->samples_in = e->samples_in; g->m_out = e->m_out; }
        if (e->merged) rc = tail_enqueue_merged(e, n_samples, m_first);
        else for (sonde_engine *g : e->groups) {
            const int r2 = tail_enqueue(g, n_samples, m_first, e->stream, e->ev_a[slot]);
            if (r2 && !rc) rc = r2;
        }
        if (e->d_summary && e->d_summary_snap) {
            // the summary records of all groups as this call leaves them: behind every group's last frame sync, on the owner's stream B; a lagged fetch waits for it
            for (sonde_engine *g : e->groups) hipStreamWaitEvent(e->stream_b, g->ev_b[slot], 0);
            hipMemcpyAsync(e->d_summary_snap + (size_t)(e->call & 1) * C, e->d_summary, (size_t)C * sizeof(sonde_summary_t), hipMemcpyDeviceToDevice, e->stream_b);
            hipEventRecord(e->ev_b[slot], e->stream_b);
            e->snap_calls = e->call + 1;
        }
        e->call += 1;
    } else {
        if (e->stream_b != e->stream) hipEventRecord(e->ev_a[slot], e->stream);
        rc = tail_enqueue(e, n_samples, m_first, e->stream, e->ev_a[slot]);
    }
    return rc;
}

}  // extern "C"

// arguments of the IF chain of one call (k_if_chain): n_if IF samples per channel from IF index m_first on
static IfArgs fill_if(sonde_engine *e, int n_if, uint32_t m_first) {
    IfArgs b{};
    const bool fe = e->cfg.sonde_type == SONDE_FRONTEND;
    b.y = e->d_y; b.tap_ifiq = (e->cfg.keep_soft || fe) ? e->d_ifiq : nullptr; b.fm = e->d_fm; b.bufs = e->d_bufs; b.n_ch = e->cfg.n_channels; b.ring_len = e->ring_len;
    b.n = n_if; b.m0 = m_first;
    b.lpiq_on = !e->w_iq.empty(); b.lpiq_taps = (int)e->w_iq.size(); b.lpfm_on = !e->w_fm.empty(); b.lpfm_taps = (int)e->w_fm.size();
    b.tone_on = (e->cfg.input != SONDE_IN_IFIQ0); b.nwin = (int)e->sps;           // --iq0 slices the FM stream (opt_iq = 1)
    b.fm_on = (e->cfg.keep_soft || fe || !e->w_fm.empty() || !b.tone_on) ? 1 : 0;   // fm_buffer feeds only --dc/--lpFM and the parity taps
    b.w_iq = e->d_wiq; b.w_fm = e->d_wfm; b.rho = e->rho; b.sps = e->sps; b.epoch = e->d_epoch;
    return b;
}
// The IF-rate part of a process call: IF chain, header search rounds, frame sync (+ block codes), decoder, counter publish — everything behind the decimator.
// `e` is an engine that runs such a tail: a plain engine, or one group of a mixed engine; fs = the stream the decimator of the call ran on, front_done = the
// event recorded behind it there (stream B waits for it unless it IS that stream); n_samples / m_first: the call's input samples per channel and the IF index of its first output.
static int tail_enqueue(sonde_engine *e, int32_t n_samples, uint32_t m_first, hipStream_t fs, hipEvent_t front_done) {
    const int D = e->info.decM, C = e->cfg.n_channels;
    const int n_if = n_samples / D;
    const bool fe = e->cfg.sonde_type == SONDE_FRONTEND;
    IfArgs b = fill_if(e, n_if, m_first);
    // IF-rate work goes to stream B behind this call's decimator; the next call's decimator may overlap it
    const int slot = (int)(e->call & 3);
    if (e->stream_b != fs) hipStreamWaitEvent(e->stream_b, front_done, 0);
    CorrArgs c{};
    c.bufs = e->d_bufs; c.corr = e->d_corr; c.match = e->d_match; c.n_ch = C; c.ring_len = e->ring_len; c.n = n_if; c.L = e->info.L; c.m0 = m_first;
    c.state = e->d_state; c.delay = e->info.delay; c.frame_samples = e->frame_samples;
    c.ntypes = e->corr_types; c.isps = e->corr_isps; c.nsym = e->hdrlen / e->symhd; c.shapes = e->d_shapes; c.sym_type = e->d_symtype; c.sym_sign = e->d_symsign;
    if (e->cfg.opt_dc && e->cfg.input != SONDE_IN_AUDIO) {
        // --dc with IQ input: a header detection may change Df / the IF filter from its sample on (find_header,
        // demod_mod.c:1553-1600).  The chunk is processed optimistically; k_framesync stops a channel at such an event and
        // reports the sample, and everything IF-rate is redone from there with the new state until no channel reports one.
        hipStream_t sb = e->stream_b;
        sonde_launch_fill_u32(e->d_start, m_first, C, sb);
        b.y = e->d_yrot; b.tap_ifiq = e->d_ifiq; b.afc = e->d_afc; b.start = e->d_start; b.fmraw = e->d_fmraw; b.w_iq0 = e->d_wiq0; b.fm_on = 1;
        c.start = e->d_start;
        AfcRotArgs r{}; r.y = e->d_y; r.yrot = e->d_yrot; r.afc = e->d_afc; r.start = e->d_start; r.n_ch = C; r.ring_len = e->ring_len;
        r.sr = e->info.if_sr; r.m_end = e->m_out;
        for (int it = 0; it < n_if / std::max(1, e->info.K - 4) + 4; it++) {
            hipMemsetAsync(e->d_pending, 0, sizeof(unsigned), sb);
            sonde_launch_afc_rotate(&r, sb, n_if);
            prof_begin(e, "if_chain", sb); sonde_launch_if_chain(&b, sb); prof_end(e, sb);
            prof_begin(e, "header_corr", sb); sonde_launch_header_corr(&c, sb); prof_end(e, sb);
            if (e->opt_iq >= 2) { CorrArgs c2 = c; c2.bufs = e->d_fm; c2.corr = e->d_corr2; sonde_launch_header_corr(&c2, sb); }
            launch_framesync(e, 0);
            hipMemcpyAsync(e->h_pending, e->d_pending, sizeof(unsigned), hipMemcpyDeviceToHost, sb);
            HIPCHK(hipStreamSynchronize(sb));
            if (*e->h_pending == 0) break;
        }
        if (e->stream_b != fs) hipEventRecord(e->ev_if[slot], e->stream_b);
    } else {
        if (e->cfg.input != SONDE_IN_AUDIO) { prof_begin(e, "if_chain", e->stream_b); sonde_launch_if_chain(&b, e->stream_b); prof_end(e, e->stream_b); }
        if (e->stream_b != fs) hipEventRecord(e->ev_if[slot], e->stream_b);
        if (!fe && e->d_win) {
            // header search with the reference's own transform: rounds of plan -> evaluate -> sync; the sync stops where the planned
            // windows end and the next round plans from the state it left.  First round: the two windows a received sonde needs.
            // A round ends when the planned windows are used up or a hit's frame has been sliced (the search then resumes at a place the
            // plan could not know), so it covers at least min(W windows, one window + one frame) samples: two rounds for a call of up to
            // about a second, more for longer ones.
            const int rounds = sync_rounds_of(e, n_if);
            for (int round = 0; round < rounds; round++) sync_round(e, round == 0 ? 2 : e->win_W);
        } else if (!fe) {
            // two passes (corr_tile_unused in sonde_kernels.hip): correlate what two search windows can reach, sync up to there, then the
            // rest with the state that is known by then — nothing at all for a channel whose new frame covers the rest of the call
            static const bool one_pass = getenv("SONDE_CORR_ONEPASS") != nullptr;         // A/B aid
            const uint32_t lim = (!one_pass && n_if > 3 * e->info.K) ? (uint32_t)(2 * e->info.K + 64) : 0u;
            if (lim) {
                c.limit = lim; e->corr_limit = lim;
                prof_begin(e, "header_corr", e->stream_b); sonde_launch_header_corr(&c, e->stream_b); prof_end(e, e->stream_b);
                launch_framesync(e, 0);
                c.limit = 0; e->corr_limit = 0;
            }
            prof_begin(e, "header_corr", e->stream_b); sonde_launch_header_corr(&c, e->stream_b); prof_end(e, e->stream_b);
            launch_framesync(e, 0);
        }
    }
    if (e->d_summary && e->d_summary_snap && !e->is_group)         // (a mixed engine copies once, behind all its groups: sonde_engine_process_device)
        hipMemcpyAsync(e->d_summary_snap + (size_t)(e->call & 1) * C, e->d_summary, (size_t)C * sizeof(sonde_summary_t), hipMemcpyDeviceToDevice, e->stream_b);
    return tail_finish(e);
}

// The tail of a call of a mixed engine whose groups share launches: ONE IF chain, and per header-search round ONE plan, ONE window transform and ONE frame sync
// over all rows (k_*_multi: the row's group brings its arguments), on the one stream B the groups share; behind each frame sync the block codes of the DFM / M10
// groups, at the end every group's own finish (counter snapshot, RS41 decoder on its stream E, publish, ev_b).
static int tail_enqueue_merged(sonde_engine *e, int32_t n_samples, uint32_t m_first) {
    const int n_if = n_samples / e->info.decM, G = (int)e->groups.size();
    hipStream_t sb = e->stream_b;
    const int slot = (int)(e->call & 3);
    hipStreamWaitEvent(sb, e->ev_a[slot], 0);
    IfArgs ia[SONDE_MAX_GROUPS]; WinPlanArgs pa[SONDE_MAX_GROUPS]; WinFftArgs fa[SONDE_MAX_GROUPS]; SyncArgs sa[SONDE_MAX_GROUPS];
    for (int g = 0; g < G; g++) ia[g] = fill_if(e->groups[(size_t)g], n_if, m_first);
    prof_begin(e, "if_chain", sb); sonde_launch_if_chain_multi(ia, G, sb); prof_end(e, sb);
    for (sonde_engine *g : e->groups) hipEventRecord(g->ev_if[slot], sb);         // (what tail_begin makes the next-but-one decimator wait for)
    int rounds = 0;
    for (sonde_engine *g : e->groups) rounds = std::max(rounds, sync_rounds_of(g, n_if));
    for (int round = 0; round < rounds; round++) {
        for (int g = 0; g < G; g++) fill_round(e->groups[(size_t)g], round == 0 ? 2 : e->groups[(size_t)g]->win_W, pa[g], fa[g]);
        prof_begin(e, "header_corr", sb);
        sonde_launch_sync_plan_multi(pa, G, sb);
        sonde_launch_sync_window_fft_multi(fa, G, sb);
        prof_end(e, sb);
        for (int g = 0; g < G; g++) sa[g] = fill_sync(e->groups[(size_t)g], 0);
        prof_begin(e, "framesync", sb); sonde_launch_framesync_multi(sa, G, sb); prof_end(e, sb);
        for (sonde_engine *g : e->groups) launch_blockcodes(g);
    }
    int rc = 0;
    for (sonde_engine *g : e->groups) { const int r2 = tail_finish(g); if (r2 && !rc) rc = r2; }
    return rc;
}

// header-search rounds of a call of n_if IF samples: a round ends when the planned windows are used up or a hit's frame has been sliced
static int sync_rounds_of(const sonde_engine *e, int n_if) {
    const int kw = std::max(1, e->info.K - 4);
    return 2 + n_if / (kw + (int)e->frame_samples) + n_if / (e->win_W * kw);
}

// the end of a call's tail, behind its last frame sync on stream B: frame counter snapshot, the decoder of the damaged RS41 frames (stream E), counter publish, ev_b
static int tail_finish(sonde_engine *e) {
    const int C = e->cfg.n_channels;
    const int slot = (int)(e->call & 3);
    // the call's records are complete (ev_b) once its damaged frames are decoded and the counter is published
    hipStream_t se = e->stream_b;
    // the counter as THIS call's last frame sync leaves it, taken on the frame sync's own stream: what the call publishes.  (Published from the live counter
    // behind the decoder on stream E it could already include slots the NEXT call's frame sync — which only waits for the call before last — has counted but not
    // written yet: ADVICE round 4.)
    hipMemcpyAsync(e->d_fcount + 1 + slot, e->d_fcount, sizeof(unsigned), hipMemcpyDeviceToDevice, e->stream_b);
    if (e->ecc_listed) {
        if (e->stream_e) { hipEventRecord(e->ev_s, e->stream_b); hipStreamWaitEvent(e->stream_e, e->ev_s, 0); se = e->stream_e; }
        const int par = (int)(e->call & 1);
        prof_begin(e, "rs_ecc", se);
        sonde_launch_rs41_ecc_frames(e->d_frames, e->d_ecc_list + (size_t)par * e->max_frames, e->d_ecc_cnt + par, e->d_ecc_cnt + 2 + par, e->max_frames,
                                     e->cfg.ecc_level, e->d_consts + 136, e->d_consts + 648, std::min(512, std::max(1, C)), se);
        prof_end(e, se);
    }
    sonde_launch_publish_u32(e->d_fcount + 1 + slot, e->h_count_dev + slot, se);
    hipEventRecord(e->ev_b[slot], se);
    e->in_call = false;
    e->call += 1;
    if (hipPeekAtLastError() != hipSuccess) { fprintf(stderr, "libsonde_hip: launch failed: %s\n", hipGetErrorString(hipGetLastError())); return SONDE_E_NOGPU; }
    return 0;
}


extern "C" {

// arguments of one header-search round (k_sync_plan + k_sync_window_fft); advances the round counter
static void fill_round(sonde_engine *e, int W, WinPlanArgs &p, WinFftArgs &f) {
    // the item table has win_W slots per channel; this round plans and evaluates the first W of them (the others are cleared)
    const int C = e->cfg.n_channels;
    p = WinPlanArgs{}; p.state = e->d_state; p.items = e->d_win; p.n_ch = C; p.stride = e->win_W; p.W = W; p.K = e->info.K; p.L = e->info.L;
    p.delay = e->info.delay; p.frame_samples = e->frame_samples; p.avail = e->m_out; p.epoch = e->d_epoch;
    p.work = e->d_work; p.work_count = e->d_work_count; p.round_parity = e->sync_rounds & 1;
    f = WinFftArgs{}; f.bufs = e->d_bufs; f.items = e->d_win; f.Fm = e->d_Fm; f.tws = e->d_tws; f.n_ch = C; f.stride = e->win_W; f.W = W;
    f.K = e->info.K; f.L = e->info.L; f.ring_len = e->ring_len;
    f.work = e->d_work; f.work_count = e->d_work_count; f.round_parity = e->sync_rounds & 1; f.small_wg = e->small_tail; f.park = e->d_park;
    if (f.small_wg && !e->d_park) { if (hipMalloc((void **)&e->d_park, (size_t)SONDE_WFH_MAXGRID * 8192 * sizeof(float2)) != hipSuccess) f.small_wg = 0; f.park = e->d_park; }
    static const bool want_prof = getenv("SONDE_WF_PROF") != nullptr;          // profiling aid: cycles per phase of workgroup 0, printed when the engine is destroyed
    if (want_prof && !e->d_wfprof) { if (hipMalloc((void **)&e->d_wfprof, 32 * sizeof(unsigned long long)) == hipSuccess) hipMemset(e->d_wfprof, 0, 32 * sizeof(unsigned long long)); }
    f.prof = e->d_wfprof;
    e->sync_rounds++;
}
static void sync_round(sonde_engine *e, int W) {
    hipStream_t sb = e->stream_b;
    WinPlanArgs p; WinFftArgs f;
    fill_round(e, W, p, f);
    prof_begin(e, "header_corr", sb);
    sonde_launch_sync_plan(&p, sb);
    sonde_launch_sync_window_fft(&f, sb);
    prof_end(e, sb);
    launch_framesync(e, 0);
}

// arguments of a frame-sync launch (k_framesync); a launch inside a process call of an engine with the device decoder claims the call's work list
static SyncArgs fill_sync(sonde_engine *e, int eof) {
    const int C = e->cfg.n_channels;
    SyncArgs s{};
    s.eof = eof; s.eof_ch = e->eof_ch; s.epoch = e->d_epoch; s.rs41 = (e->cfg.sonde_type == SONDE_RS41); s.ecc_level = (s.rs41 && e->dev_ecc && e->cfg.ecc_level >= 1 && e->cfg.ecc_level <= 2) ? e->cfg.ecc_level : 0;
    // damaged frames of a process call go on that call's work list (k_rs41_ecc_frames at the end of the call); end-of-stream launches have none
    if (s.ecc_level && !eof && e->in_call && e->d_ecc_list) { s.ecc_list = e->d_ecc_list + (size_t)(e->call & 1) * e->max_frames; s.ecc_count = e->d_ecc_cnt + (e->call & 1); e->ecc_listed = true; }
    s.bufs = e->d_bufs; s.corr = e->d_corr; s.state = e->d_state; s.frames = e->d_frames; s.frame_count = e->d_fcount; s.soft = e->d_soft; s.soft1 = e->d_soft1;
    s.hdr = e->d_consts; s.hdr_bytes = e->d_consts + 64; s.mask = e->d_consts + 72; s.gf_exp = e->d_consts + 136; s.gf_log = e->d_consts + 648;
    s.bitwin = e->d_bitwin; s.bitend = e->d_bitend;
    s.small_wg = e->small_tail && !e->cfg.opt_dc;
    s.n_ch = C; s.ring_len = e->ring_len; s.max_frames = e->max_frames; s.avail = e->m_out;
    s.K = e->info.K; s.L = e->info.L; s.delay = e->info.delay; s.hdrlen = e->hdrlen; s.symhd = e->symhd; s.symlen = e->symlen;
    s.hdmax = e->hdmax; s.bitofs = e->bitofs; s.nbits = e->nbits; s.frame_samples = e->frame_samples;
    s.sps = e->sps; s.thres = e->thres; s.l_win = e->l_win;
    s.opt_auto = e->cfg.opt_auto != 0 || e->cfg.sonde_type == SONDE_M10 || e->cfg.sonde_type == SONDE_M20;      // M10: either polarity (differential coding)
    s.opt_dc = e->cfg.opt_dc != 0; s.opt_iq = e->opt_iq; s.lpiq_on = !e->w_iq.empty(); s.lpfm_taps = (int)e->w_fm.size(); s.N = e->info.N; s.sr = e->info.if_sr;
    s.match_sum = e->match_sum; s.fm = e->d_fm; s.corr2 = e->d_corr2; s.ifiq = e->d_ifiq; s.afc = e->d_afc; s.start = e->d_start; s.pending = e->d_pending;
    s.corr_limit = e->corr_limit;
    s.win = e->d_win; s.win_W = e->win_W;
    s.prof = e->d_wfprof ? e->d_wfprof + 16 : nullptr;
    s.summary = e->d_summary; s.summary_base = e->summary_base; s.summary_map = e->d_sum_map; s.summary_type = e->cfg.sonde_type; s.summary_epoch = e->samples_in / (uint64_t)std::max(1, e->info.decM);     // IF samples produced so far, 64 bit
    return s;
}
static void launch_framesync_impl(sonde_engine *e, int eof) {
    const SyncArgs s = fill_sync(e, eof);
    prof_begin(e, "framesync", e->stream_b); sonde_launch_framesync(&s, e->stream_b); prof_end(e, e->stream_b);
}

int sonde_engine_process_host(sonde_engine_t *e, const void *h_iq, int64_t ch_stride, int32_t n_samples) {
    if (!e || !h_iq) return SONDE_E_ARG;
    const int C = e->cfg.n_channels;
    if (n_samples <= 0 || n_samples > e->cfg.max_chunk || (ch_stride != 0 && ch_stride < n_samples)) return SONDE_E_RANGE;
    const size_t unit = (e->cfg.input == SONDE_IN_AUDIO ? (size_t)std::max(1, e->cfg.audio_channels) : 2) * (size_t)(e->cfg.bits / 8);
    const int rows = ch_stride == 0 ? 1 : C;                   // shared wideband stream: staged once
    const size_t need = (size_t)rows * n_samples * unit;
    if (need > e->stage_bytes) {
        if (e->d_stage) { hipStreamSynchronize(e->stream); hipFree(e->d_stage); e->d_stage = nullptr; }
        HIPCHK(hipMalloc((void **)&e->d_stage, need)); e->stage_bytes = need;
    }
    HIPCHK(hipMemcpy2DAsync(e->d_stage, (size_t)n_samples * unit, h_iq, (size_t)std::max<int64_t>(ch_stride, n_samples) * unit, (size_t)n_samples * unit, rows,
                            hipMemcpyHostToDevice, e->stream));
    // the caller may free or overwrite h_iq as soon as this returns; a pageable source can be pinned in place and read by the
    // copy engine after hipMemcpy2DAsync has returned, so wait for the copy itself (not for the kernels queued behind it)
    HIPCHK(hipEventRecord(e->ev_copy, e->stream));
    HIPCHK(hipEventSynchronize(e->ev_copy));
    return sonde_engine_process_device(e, e->d_stage, ch_stride == 0 ? 0 : n_samples, n_samples);
}

int sonde_engine_sync(sonde_engine_t *e) {
    if (!e) return SONDE_E_ARG;
    for (sonde_engine *g : e->groups) { const int rc = sonde_engine_sync(g); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->stream_b));
    if (e->stream_e) HIPCHK(hipStreamSynchronize(e->stream_e));
    prof_collect(e);
    return 0;
}

static int fetch_rs41(sonde_engine_t *e, sonde_frame_t *out, int32_t max, int lag) {
    if (!e || !out || max < 0 || e->cfg.sonde_type != SONDE_RS41) return SONDE_E_ARG;
    std::vector<FrameRec> recs;
    const int n = collect_records(e, lag, recs, e->d_soft ? &e->last_soft : nullptr, max);
    if (n < 0) return n;
    for (int i = 0; i < n; i++) {
        const FrameRec &r = recs[i];
        sonde_frame_t &f = out[i];
        memset(&f, 0, sizeof f);
        f.channel = r.channel; f.len = r.len; f.mv = r.mv; f.mv_pos = rel_pos(e, r); f.nbytes = r.nbytes;
        uint8_t *keepf = e->last_frame.data() + (size_t)r.channel * 518;
        if (r.nbytes >= 518) {
            memcpy(f.frame, r.frame, 518);
            f.ecc = 0;
            if (e->cfg.ecc_level > 0) {
                bool clean = true;
                for (int k = 0; k < 48; k++) clean &= (r.synd[k] == 0);
                if (r.ecc_done == 1) f.ecc = r.ecc;                  // rs41_ecc() is done on the device: corrected bytes, zero tail, its return value
                else if (clean) { for (int k = f.len; k < 518; k++) f.frame[k] = 0; }
                else { f.ecc = rs41_ecc(f.frame, f.len, e->cfg.ecc_level, r.synd); e->host_ecc_frames++; }
            }
        } else {
            // end-of-stream frame: bytes not read keep the previous frame's content unless fewer than
            // pos_GPS1 = 0x93 bytes exist, then they are zeroed (print_frame, rs41mod.c:2479-2490)
            memcpy(f.frame, keepf, 518);
            memcpy(f.frame, r.frame, (size_t)r.nbytes);
            if (r.nbytes < 0x93) for (int k = r.nbytes; k < 518; k++) f.frame[k] = 0;
            f.len = (rs41_frametype(f.frame) >= 0) ? 320 : 518;
            f.ecc = e->cfg.ecc_level > 0 ? rs41_ecc(f.frame, f.len, e->cfg.ecc_level, nullptr) : 0;
        }
        memcpy(keepf, f.frame, 518);
    }
    e->last_n = n;
    return n;                                  // frames dropped by a full queue: sonde_engine_overflowed()
}

int sonde_engine_set_summary(sonde_engine_t *e, void *d_summary, uint32_t channel_base) {
    if (!e) return SONDE_E_ARG;
    e->d_summary = (sonde_summary_t *)d_summary; e->summary_base = channel_base;
    for (sonde_engine *g : e->groups) { g->d_summary = e->d_summary; g->summary_base = channel_base; }     // (each group writes its channels' records: d_sum_map)
    return 0;
}

int sonde_engine_set_summary_snapshots(sonde_engine_t *e, void *d_snap) {
    if (!e) return SONDE_E_ARG;
    e->d_summary_snap = (sonde_summary_t *)d_snap;
    return (int)(e->call & 1);
}

long long sonde_engine_host_ecc_frames(sonde_engine_t *e) {
    if (!e) return SONDE_E_ARG;
    long long n = e->host_ecc_frames;
    for (sonde_engine *g : e->groups) n += g->host_ecc_frames;
    return n;
}
int sonde_engine_set_device_ecc(sonde_engine_t *e, int32_t on) {
    if (!e) return SONDE_E_ARG;
    for (sonde_engine *g : e->groups) { const int rc = sonde_engine_set_device_ecc(g, on); if (rc) return rc; }
    if ((on != 0) == e->dev_ecc) return 0;
    if (e->d_blk_done) {
        // DFM / M10 engines decide per record range where its block codes run: k_dfm_hits / k_m10_hits behind the frame sync that queued it, or the host inside
        // the fetch.  A switch therefore takes effect between records only: everything queued so far must have been fetched (SONDE_E_ARG otherwise — fetch first),
        // and the device decoder then starts at the record the next frame sync queues (a toggle between a call and its fetch read slots never decoded).
        if (hipStreamSynchronize(e->stream_b) != hipSuccess) return SONDE_E_NOGPU;
        unsigned count = 0;
        if (hipMemcpy(&count, e->d_fcount, sizeof count, hipMemcpyDeviceToHost) != hipSuccess) return SONDE_E_NOGPU;
        if (count != e->read_idx) return SONDE_E_ARG;
        const unsigned dn[2] = { count, 0u };
        if (on && hipMemcpy(e->d_blk_done, dn, sizeof dn, hipMemcpyHostToDevice) != hipSuccess) return SONDE_E_NOGPU;
    }
    e->dev_ecc = on != 0;
    return 0;
}

int sonde_engine_overflowed(sonde_engine_t *e) {
    if (!e) return SONDE_E_ARG;
    bool ovf = e->overflow; e->overflow = false;
    for (sonde_engine *g : e->groups) { ovf |= g->overflow; g->overflow = false; }
    return ovf ? 1 : 0;
}

int sonde_engine_fetch_frames(sonde_engine_t *e, sonde_frame_t *out, int32_t max) { return sonde_engine_fetch_frames_lagged(e, out, max, 0); }

int sonde_engine_fetch_frames_lagged(sonde_engine_t *e, sonde_frame_t *out, int32_t max, int32_t lag) {
    if (lag < 0) lag = 0;
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_RS41, out, max, lag, [lag](sonde_engine *g, sonde_frame_t *o, int32_t m) { return fetch_rs41(g, o, m, lag); });
    return fetch_rs41(e, out, max, lag);
}

int sonde_engine_fetch_dfm(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t finish) {
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_DFM09, out, max, 0, [finish](sonde_engine *g, sonde_dfm_frame_t *o, int32_t m) { return fetch_dfm_impl(g, o, m, finish, 0); });
    return fetch_dfm_impl(e, out, max, finish, 0);
}
int sonde_engine_fetch_dfm_lagged(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t lag) {
    if (lag < 0) lag = 0;
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_DFM09, out, max, lag, [lag](sonde_engine *g, sonde_dfm_frame_t *o, int32_t m) { return fetch_dfm_impl(g, o, m, 0, lag); });
    return fetch_dfm_impl(e, out, max, 0, lag);
}
int sonde_engine_fetch_m10(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t finish) {
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_M10, out, max, 0, [finish](sonde_engine *g, sonde_m10_frame_t *o, int32_t m) { return fetch_m10_impl(g, o, m, finish, 0); });
    return fetch_m10_impl(e, out, max, finish, 0);
}
int sonde_engine_fetch_m10_lagged(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t lag) {
    if (lag < 0) lag = 0;
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_M10, out, max, lag, [lag](sonde_engine *g, sonde_m10_frame_t *o, int32_t m) { return fetch_m10_impl(g, o, m, 0, lag); });
    return fetch_m10_impl(e, out, max, 0, lag);
}

static int fetch_dfm_impl(sonde_engine_t *e, sonde_dfm_frame_t *out, int32_t max, int32_t finish, int lag) {
    if (!e || !out || max < 0 || e->cfg.sonde_type != SONDE_DFM09) return SONDE_E_ARG;
    if (finish) { launch_framesync(e, 1); lag = 0; }
    std::vector<FrameRec> recs;
    std::vector<float> soft;
    const bool on_dev = blockcodes_on_device(e);
    const int nh = collect_records(e, lag, recs, on_dev ? nullptr : &soft, max / 8);
    if (nh < 0) return nh;
    e->last_n = nh;
    if (on_dev) {
        // the frames were sliced and decoded on the device behind the frame sync (k_dfm_hits): only they come over — the soft bits stay where they are until
        // sonde_engine_fetch_soft asks for them
        const unsigned start = e->read_idx - (unsigned)nh;
        e->soft_lazy = true; e->soft_lazy_start = start;
        if (copy_decoded(e, e->d_dfm_out, e->h_dfm, start, nh, 8, lag)) return SONDE_E_NOGPU;
        int n = 0;
        for (int h = 0; h < nh; h++) {
            const FrameRec &r = recs[h];
            for (int f = 0; f < 8 && n < max; f++) {
                const sonde_dfm_frame_t &d = e->h_dfm[(size_t)h * 8 + f];
                if (d.frame_in_hit != f) break;                            // (-1: the hit ends before this frame)
                sonde_dfm_frame_t &o = out[n++];
                o = d;
                o.mv_pos = rel_pos(e, r);
                o.frm_count = (float)(rel_pos(e, r) / (2.0 * e->sps * 280) + f);   // gpx._frmcnt (dfm09mod.c:1662)
            }
        }
        return n;
    }
    e->last_soft = soft; e->soft_lazy = false;
    int n = 0;
    for (int h = 0; h < nh; h++) {
        const FrameRec &r = recs[h];
        const float *sb = soft.data() + (size_t)h * e->nbits;
        // frame 0 of a hit holds bits 16..279 (the header was consumed by the correlator), then 7 x 280 (dfm09mod.c:1652-1717)
        for (int f = 0; f < 8 && n < max; f++) {
            const int first = f == 0 ? 0 : 264 + 280 * (f - 1), skip = f == 0 ? 16 : 0;
            if (first + (280 - skip) > r.nbytes) break;           // nbytes = valid bits of the hit; partial frame is dropped
            uint8_t hb[280]; float sf[280];
            memset(hb, 0, sizeof hb); memset(sf, 0, sizeof sf);
            for (int i = skip; i < 280; i++) {
                const int b = first + i - skip;
                hb[i] = (r.frame[b >> 3] >> (b & 7)) & 1; sf[i] = sb[b];
            }
            sonde_dfm_frame_t &o = out[n++];
            memset(&o, 0, sizeof o);
            o.channel = r.channel; o.frame_in_hit = f; o.mv = r.mv; o.mv_pos = rel_pos(e, r);
            o.frm_count = (float)(rel_pos(e, r) / (2.0 * e->sps * 280) + f);   // gpx._frmcnt (dfm09mod.c:1662)
            o.inv = r.mv < 0.f;                                            // an accepted header has the sign of the polarity in effect
            for (int i = 0; i < 280; i++) o.rawbits[i >> 3] |= (uint8_t)(hb[i] << (i & 7));
            o.ecc[0] = dfm_block(e->cfg.ecc_level, hb + 16, sf + 16, 7, o.conf);
            o.ecc[1] = dfm_block(e->cfg.ecc_level, hb + 72, sf + 72, 13, o.dat1);
            o.ecc[2] = dfm_block(e->cfg.ecc_level, hb + 176, sf + 176, 13, o.dat2);
        }
    }
    return n;                                  // frames dropped by a full queue: sonde_engine_overflowed()
}

// M10 / M20: sliced Manchester bits -> differentially decoded frame bytes.  The bit characters persist per channel like the
// reference's gpx.frame_bits: a frame cut short by the end of the stream keeps the tail of the previous one behind the terminator.
static int mxx_bytes(sonde_engine *e, const FrameRec &r, int nbytes_max, uint8_t *frame) {
    const int NB = nbytes_max * 8;
    if (e->m10_bits.empty()) e->m10_bits.assign((size_t)e->cfg.n_channels * (NB + 8), 0);
    char *fb = e->m10_bits.data() + (size_t)r.channel * (NB + 8);
    const int nv = std::min(r.nbytes, NB);                          // nbytes = valid bits of the hit
    int bit0 = '0';                                                 // differential decoding: 1 = same as the previous bit (m10mod.c:1484)
    for (int p = 0; p < nv; p++) { const int bit = (r.frame[p >> 3] >> (p & 7)) & 1; fb[p] = (char)(0x31 ^ (bit0 ^ bit)); bit0 = bit; }
    fb[nv] = 0;
    for (int i = 0; i < nbytes_max; i++) {                          // bits2bytes, big endian; anything but '1' counts as 0 (m10mod.c:141-166)
        int v = 0;
        for (int k = 0; k < 8; k++) if (fb[8 * i + 7 - k] == '1') v |= 1 << k;
        frame[i] = (uint8_t)v;
    }
    return nv;
}

int sonde_engine_fetch_m20(sonde_engine_t *e, sonde_m20_frame_t *out, int32_t max, int32_t finish) {
    if (e && !e->groups.empty()) return mixed_fetch(e, SONDE_M20, out, max, 0, [finish](sonde_engine *g, sonde_m20_frame_t *o, int32_t m) { return sonde_engine_fetch_m20(g, o, m, finish); });
    if (!e || !out || max < 0 || e->cfg.sonde_type != SONDE_M20) return SONDE_E_ARG;
    if (finish) launch_framesync(e, 1);
    std::vector<FrameRec> recs;
    std::vector<float> soft;
    const int n = collect_records(e, 0, recs, &soft, max);
    if (n < 0) return n;
    e->last_soft = soft; e->last_n = n; e->soft_lazy = false;
    for (int h = 0; h < n; h++) {
        const FrameRec &r = recs[h];
        sonde_m20_frame_t &o = out[h];
        memset(&o, 0, sizeof o);
        o.nbits = mxx_bytes(e, r, 101 + 64, o.frame);
        o.channel = r.channel; o.mv = r.mv; o.mv_pos = rel_pos(e, r);
        sonde_m20_frame_finish(&o);                                 // length, firmware byte, checksums (m20mod.c:875-907)
    }
    return n;                                  // frames dropped by a full queue: sonde_engine_overflowed()
}

static int fetch_m10_impl(sonde_engine_t *e, sonde_m10_frame_t *out, int32_t max, int32_t finish, int lag) {
    if (!e || !out || max < 0 || e->cfg.sonde_type != SONDE_M10) return SONDE_E_ARG;
    if (finish) { launch_framesync(e, 1); lag = 0; }
    std::vector<FrameRec> recs;
    std::vector<float> soft;
    const bool on_dev = blockcodes_on_device(e);
    const int n = collect_records(e, lag, recs, on_dev ? nullptr : &soft, max);
    if (n < 0) return n;
    e->last_n = n;
    if (on_dev) {
        // differential decoding, bytes and checkM10 were done on the device behind the frame sync (k_m10_hits)
        const unsigned start = e->read_idx - (unsigned)n;
        e->soft_lazy = true; e->soft_lazy_start = start;
        if (copy_decoded(e, e->d_m10_out, e->h_m10, start, n, 1, lag)) return SONDE_E_NOGPU;
        for (int h = 0; h < n; h++) { out[h] = e->h_m10[h]; out[h].mv_pos = rel_pos(e, recs[h]); }
        return n;
    }
    e->last_soft = soft; e->soft_lazy = false;
    if (e->m10_chk3 && e->d_soft1 && !soft.empty()) {
        // --chk3 (m10mod.c:1476-1479): the bit is re-decided from both soft values of read_softbit2p, (sb + 0.25 sb1) >= 0, before the differential decoding
        const unsigned start = e->read_idx - (unsigned)n;
        std::vector<float> s1((size_t)e->nbits);
        for (int h = 0; h < n; h++) {
            const unsigned idx = (start + (unsigned)h) % (unsigned)e->max_frames;
            if (hipMemcpy(s1.data(), e->d_soft1 + (size_t)idx * e->nbits, (size_t)e->nbits * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SONDE_E_NOGPU;
            FrameRec &r = recs[h];
            const int nv = std::min(r.nbytes, e->nbits);
            for (int p = 0; p < nv; p++) {
                const int bit = ((double)soft[(size_t)h * e->nbits + p] + 0.25 * (double)s1[p]) >= 0.0;
                r.frame[p >> 3] = (uint8_t)((r.frame[p >> 3] & ~(1u << (p & 7))) | ((unsigned)bit << (p & 7)));
            }
        }
    }
    for (int h = 0; h < n; h++) {
        const FrameRec &r = recs[h];
        sonde_m10_frame_t &o = out[h];
        memset(&o, 0, sizeof o);
        const int nv = mxx_bytes(e, r, 101 + 20, o.frame);
        o.channel = r.channel; o.nbits = nv; o.mv = r.mv; o.mv_pos = rel_pos(e, r);
        sonde_m10_frame_finish(&o);
    }
    return n;                                  // frames dropped by a full queue: sonde_engine_overflowed()
}

int sonde_engine_fetch_hits(sonde_engine_t *e, sonde_hit_t *out, int32_t max, int32_t finish) {
    if (!e || !e->groups.empty() || !out || max < 0 || !e->d_soft || e->cfg.sonde_type == SONDE_FRONTEND) return SONDE_E_ARG;
    if (finish) launch_framesync(e, 1);
    std::vector<FrameRec> recs;
    std::vector<float> soft;
    const int n = collect_records(e, 0, recs, &soft, max);
    if (n < 0) return n;
    e->last_soft = soft; e->last_n = n; e->soft_lazy = false;
    if (e->d_soft1) {                                          // the same ring slots of the second soft-bit array
        const unsigned start = e->read_idx - (unsigned)n;       // first ring slot of the records just taken (read_idx is already behind them)
        e->last_soft1.resize((size_t)n * e->nbits);
        for (int i = 0; i < n; i++) {
            const unsigned idx = (start + (unsigned)i) % (unsigned)e->max_frames;
            if (hipMemcpy(e->last_soft1.data() + (size_t)i * e->nbits, e->d_soft1 + (size_t)idx * e->nbits, (size_t)e->nbits * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SONDE_E_NOGPU;
        }
    }
    for (int i = 0; i < n; i++) {
        const FrameRec &r = recs[i];
        out[i].channel = r.channel; out[i].mv = r.mv; out[i].mv_pos = rel_pos(e, r);
        out[i].nbits = e->cfg.sonde_type == SONDE_RS41 ? 8 * (r.nbytes - 8) : r.nbytes;      // RS41 records count bytes incl. the 8 header bytes
    }
    return n;                                  // frames dropped by a full queue: sonde_engine_overflowed()
}

int sonde_engine_set_m10_chk3(sonde_engine_t *e, int32_t on) {
    if (!e || e->cfg.sonde_type != SONDE_M10 || (on && (!e->d_soft1 || e->cfg.input == SONDE_IN_AUDIO))) return SONDE_E_ARG;   // needs keep_soft = 2; IQ forms only
    e->m10_chk3 = on != 0;
    return 0;
}

int sonde_engine_set_sync(sonde_engine_t *e, int32_t hdmax, int32_t bitofs) {
    if (!e || !e->groups.empty() || hdmax < 0 || hdmax > 64 || bitofs < -8 || bitofs > 64 || e->call > 0) return SONDE_E_ARG;      // before the first process call only
    e->hdmax = hdmax; e->bitofs = bitofs;
    return 0;
}

int sonde_engine_set_threshold(sonde_engine_t *e, float thres) {
    if (!e || !e->groups.empty() || !(thres > 0.f) || thres >= 1.f) return SONDE_E_ARG;
    e->thres = thres;
    return 0;
}

int sonde_engine_finish(sonde_engine_t *e, sonde_frame_t *out, int32_t max) {
    if (!e || !out) return SONDE_E_ARG;
    if (!e->groups.empty())       // mixed: the RS41 groups end here; the other types with their own fetch (finish != 0)
        return mixed_fetch(e, SONDE_RS41, out, max, 0, [](sonde_engine *g, sonde_frame_t *o, int32_t m) { return sonde_engine_finish(g, o, m); });
    launch_framesync(e, 1);
    return sonde_engine_fetch_frames(e, out, max);
}

int sonde_engine_finish_channel(sonde_engine_t *e, int32_t channel) {
    if (!e || channel < 0 || channel >= e->cfg.n_channels) return SONDE_E_ARG;
    if (!e->groups.empty()) { const int gi = e->grp_of_ch[(size_t)channel]; return sonde_engine_finish_channel(e->groups[(size_t)gi], e->row_of_ch[(size_t)channel] - e->groups[(size_t)gi]->front_row); }
    e->eof_ch = channel;
    launch_framesync(e, 1);
    e->eof_ch = -1;
    return 0;
}

int sonde_engine_restart_channel(sonde_engine_t *e, int32_t channel) {
    if (!e || !e->groups.empty() || channel < 0 || channel >= e->cfg.n_channels) return SONDE_E_ARG;
    // channels of an engine share the base-rate sample clock (mixer table phase, IQ-DC segment schedule): only engines without that
    // front end can give one channel a new origin; the AFC loop of --dc and the pipelined streams are left out as well
    const bool base = e->cfg.input == SONDE_IN_IQ;            // --IQ fq: mixer + decimator in front of the IF-rate chain
    if ((!base && e->info.decM != 1) || e->cfg.opt_dc || e->cfg.opt_iqdc || e->cfg.pipeline || e->cfg.sonde_type == SONDE_FRONTEND) return SONDE_E_ARG;
    if (base && (e->f32_path || e->cfg.opt_nolut || e->lut_len <= 0 || e->lut_len % e->info.decM)) return SONDE_E_ARG;   // int16 / uint8 input through the mixer table only
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->stream_b));
    if (e->stream_e) HIPCHK(hipStreamSynchronize(e->stream_e));
    const int C = e->cfg.n_channels;
    if (base) {
        // the channel's own sample clock starts here: mixer table phase 0, IQ-DC mean 0 with the first (shortest) segment, empty decimator history
        if (!e->pcs) {
            e->pcs_cnt.assign((size_t)C, e->dc_cnt); e->pcs_max.assign((size_t)C, e->dc_max);
            std::vector<uint32_t> zero((size_t)C, 0u); std::vector<int32_t> since((size_t)C, e->dc_since);
            if (dalloc(&e->d_epoch_phase, (size_t)C) || dalloc(&e->d_pcs_cnt, (size_t)C) || dalloc(&e->d_pcs_max, (size_t)C) || dalloc(&e->d_pcs_since, (size_t)C)) return SONDE_E_NOMEM;
            HIPCHK(hipMemcpy(e->d_pcs_cnt, e->pcs_cnt.data(), (size_t)C * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_pcs_max, e->pcs_max.data(), (size_t)C * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_pcs_since, since.data(), (size_t)C * sizeof(int32_t), hipMemcpyHostToDevice));
            if (e->d_etab && !e->d_dcavg_prev) return SONDE_E_ARG;
            e->pcs = true;
        }
        const uint32_t ph = (uint32_t)(e->samples_in % (uint64_t)e->lut_len), zero = 0u, m0 = e->dc_max0; const int32_t far = 1 << 20;
        HIPCHK(hipMemcpy(e->d_epoch_phase + channel, &ph, sizeof ph, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_pcs_cnt + channel, &zero, sizeof zero, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_pcs_max + channel, &m0, sizeof m0, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_pcs_since + channel, &far, sizeof far, hipMemcpyHostToDevice));
        e->pcs_cnt[channel] = 0; e->pcs_max[channel] = m0;
        HIPCHK(hipMemset(e->d_dcavg + channel, 0, sizeof(float2)));
        if (e->d_dcavg_prev) HIPCHK(hipMemset(e->d_dcavg_prev + channel, 0, sizeof(float2)));
        HIPCHK(hipMemset(e->d_dcsums + 2 * (size_t)channel, 0, 2 * sizeof(long long)));
        for (int k = 0; k < 2; k++) if (e->d_ptail[k]) HIPCHK(hipMemset(e->d_ptail[k] + (size_t)channel * 64, 0, 64 * sizeof(float2)));
    }
    const size_t ring = (size_t)e->ring_len, row = (size_t)channel * ring;
    // history older than the new origin reads as silence, like the reference's freshly allocated buffers
    if (e->d_y)    HIPCHK(hipMemset(e->d_y + row, 0, ring * sizeof(float2)));
    if (e->d_ifiq) HIPCHK(hipMemset(e->d_ifiq + row, 0, ring * sizeof(float2)));
    if (e->d_fm)   HIPCHK(hipMemset(e->d_fm + row, 0, ring * sizeof(float)));
    if (e->d_bufs) HIPCHK(hipMemset(e->d_bufs + row, 0, ring * sizeof(float)));
    if (e->d_corr) HIPCHK(hipMemset(e->d_corr + row, 0, ring * sizeof(float)));
    if (e->d_raw)  HIPCHK(hipMemset(e->d_raw + row, 0, ring * sizeof(float)));
    if (e->epoch.empty()) {
        e->epoch.assign((size_t)C, 0u);
        if (dalloc(&e->d_epoch, (size_t)C)) return SONDE_E_NOMEM;
    }
    e->epoch[channel] = e->m_out;
    HIPCHK(hipMemcpy(e->d_epoch, e->epoch.data(), (size_t)C * sizeof(uint32_t), hipMemcpyHostToDevice));
    SyncState st; memset(&st, 0, sizeof st);
    st.s_in = e->m_out; st.mv_pos = e->m_out; st.inv = e->cfg.opt_inv ? 1u : 0u;
    HIPCHK(hipMemcpy(e->d_state + channel, &st, sizeof st, hipMemcpyHostToDevice));
    if (!e->last_frame.empty()) { memset(e->last_frame.data() + (size_t)channel * 518, 0, 518); memcpy(e->last_frame.data() + (size_t)channel * 518, kRs41HeaderBytes, 8); }
    if (!e->m10_bits.empty()) { const size_t per = e->m10_bits.size() / (size_t)C; memset(e->m10_bits.data() + (size_t)channel * per, 0, per); }
    if (e->d_m10_bits) { const size_t per = (101 + 20) * 8 + 8; HIPCHK(hipStreamSynchronize(e->stream_b)); HIPCHK(hipMemset(e->d_m10_bits + (size_t)channel * per, 0, per)); }
    return 0;
}

int sonde_engine_tune_channel(sonde_engine_t *e, int32_t channel, double fq) {
    if (!e || !e->groups.empty() || channel < 0 || channel >= e->cfg.n_channels || !(fq >= -0.5 && fq <= 0.5)) return SONDE_E_ARG;
    if (e->cfg.input == SONDE_IN_IQ && !e->ifiq && e->cfg.bits != 32 && !e->cfg.opt_nolut) {
        // base-rate engine: the channel's mixer table is that of `--IQ fq` (carrier snapped to the table's raster, demod_mod.c:1265-1288) and, in fold
        // mode, its row of the E table; meant to be followed by sonde_engine_restart_channel() — the samples the channel has seen were mixed with the old carrier
        HIPCHK(hipStreamSynchronize(e->stream));
        const Mixer m = design_mixer(-fq, e->cfg.sample_rate);
        if (m.lut_len != e->lut_len) return SONDE_E_ARG;
        HIPCHK(hipMemcpy(e->d_chanf0 + channel, &m.f0, sizeof m.f0, hipMemcpyHostToDevice));
        if (e->d_etab) {
            sonde_launch_md_etable(e->d_chanf0 + channel, e->d_wtab, e->info.decM, e->Q, e->etab_len, 1, e->d_etab + (size_t)channel * e->etab_len, e->stream);
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        return 0;
    }
    if (!e->cfg.if_tune) return SONDE_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    const double f0 = -fq;
    HIPCHK(hipMemcpy(e->d_chanf0 + channel, &f0, sizeof f0, hipMemcpyHostToDevice));
    return 0;
}

int sonde_engine_fetch_soft(sonde_engine_t *e, float *soft, int32_t max_frames) {
    if (!e || !e->groups.empty() || !soft || !e->d_soft) return SONDE_E_ARG;
    const int n = std::min(e->last_n, (int)max_frames);
    if (e->soft_lazy) {
        // the last fetch left the soft bits on the device (block codes decoded there): their ring slots now
        for (int i = 0; i < n; i++) {
            const unsigned idx = (e->soft_lazy_start + (unsigned)i) % (unsigned)e->max_frames;
            if (hipMemcpy(soft + (size_t)i * e->nbits, e->d_soft + (size_t)idx * e->nbits, (size_t)e->nbits * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SONDE_E_NOGPU;
        }
        return n;
    }
    memcpy(soft, e->last_soft.data(), (size_t)n * e->nbits * sizeof(float));
    return n;
}

int sonde_engine_fetch_soft1(sonde_engine_t *e, float *soft, int32_t max_frames) {
    if (!e || !soft || !e->d_soft1) return SONDE_E_ARG;
    const int n = std::min({e->last_n, (int)max_frames, (int)(e->last_soft1.size() / (size_t)e->nbits)});
    memcpy(soft, e->last_soft1.data(), (size_t)n * e->nbits * sizeof(float));
    return n;
}

int sonde_engine_read_tap(sonde_engine_t *e, int32_t channel, int32_t tap, int64_t first, int32_t count, float *out) {
    if (!e || !out || channel < 0 || channel >= e->cfg.n_channels || count < 0 || count > e->ring_len || first < 0) return SONDE_E_ARG;
    if (!e->groups.empty()) {
        HIPCHK(hipStreamSynchronize(e->stream));
        const int gi = e->grp_of_ch[(size_t)channel];
        return sonde_engine_read_tap(e->groups[(size_t)gi], e->row_of_ch[(size_t)channel] - e->groups[(size_t)gi]->front_row, tap, first, count, out);
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->stream_b));
    if (e->stream_e) HIPCHK(hipStreamSynchronize(e->stream_e));
    prof_collect(e);
    const void *base; size_t esz;
    switch (tap) {
        case SONDE_TAP_DECIM: base = e->d_y; esz = 8; break;
        case SONDE_TAP_IFIQ:  base = e->d_ifiq; esz = 8; break;
        case SONDE_TAP_FM:    base = e->d_fm; esz = 4; break;
        case SONDE_TAP_BUFS:  base = e->d_bufs; esz = 4; break;
        case SONDE_TAP_CORR:  base = e->d_corr; esz = 4; break;
        default: return SONDE_E_ARG;
    }
    const char *row = (const char *)base + (size_t)channel * e->ring_len * esz;
    const uint32_t mask = (uint32_t)e->ring_len - 1;
    int32_t done = 0;
    while (done < count) {
        const uint32_t idx = (uint32_t)(first + done) & mask;
        const int32_t run = std::min<int32_t>(count - done, (int32_t)(e->ring_len - idx));
        HIPCHK(hipMemcpy((char *)out + (size_t)done * esz, row + (size_t)idx * esz, (size_t)run * esz, hipMemcpyDeviceToHost));
        done += run;
    }
    return count;
}

int sonde_engine_profile(sonde_engine_t *e, int enable) {
    if (!e) return SONDE_E_ARG;
    for (sonde_engine *g : e->groups) sonde_engine_profile(g, enable);
    (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream_b); if (e->stream_e) (void)hipStreamSynchronize(e->stream_e); prof_collect(e);
    e->prof = enable != 0; e->prof_level = enable == 1 ? 1 : 2; e->stats.clear();    // 1: dominant kernel only (2 events per launch), 2: every kernel
    return 0;
}

int sonde_engine_kernel_ms(sonde_engine_t *e, const char *kernel, double *avg_ms, int64_t *launches) {
    if (!e || !kernel) return SONDE_E_ARG;
    (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream_b); if (e->stream_e) (void)hipStreamSynchronize(e->stream_e); prof_collect(e);
    if (!e->groups.empty() && strcmp(kernel, "mix_decimate") != 0) {
        // IF-rate kernels of a mixed engine: the groups' launches together (average per launch over all of them)
        double ms = 0; int64_t n = 0;
        { auto it = e->stats.find(kernel); if (it != e->stats.end()) { ms += it->second.ms; n += it->second.n; } }      // (launches the groups share are booked here)
        for (sonde_engine *g : e->groups) { double m1 = 0; int64_t n1 = 0; sonde_engine_kernel_ms(g, kernel, &m1, &n1); ms += m1 * (double)n1; n += n1; }
        if (avg_ms) *avg_ms = n ? ms / (double)n : 0; if (launches) *launches = n;
        return 0;
    }
    auto it = e->stats.find(kernel);
    if (it == e->stats.end() || it->second.n == 0) { if (avg_ms) *avg_ms = 0; if (launches) *launches = 0; return 0; }
    if (avg_ms) *avg_ms = it->second.ms / (double)it->second.n;
    if (launches) *launches = it->second.n;
    return 0;
}

// ---- mixed engines ---------------------------------------------------------------------------------------------------------------
int sonde_engine_create_mixed(const sonde_cfg_t *cfg, const double *fq, const sonde_group_t *groups, int32_t n_groups, const int32_t *group_of_channel, sonde_engine_t **out) {
    if (!cfg || !fq || !groups || !group_of_channel || !out || cfg->abi_version != SONDE_ABI_VERSION) return SONDE_E_ARG;
    if (cfg->sonde_type != SONDE_MIXED || n_groups < 1 || n_groups > 64 || cfg->n_channels < 1) return SONDE_E_ARG;
    // the shared front end is the base-rate `--IQ fq` one through the mixer table; what couples the sync back into it (--dc), other input forms and the
    // per-hit soft-bit interface stay with the single-type engines
    if (cfg->input != SONDE_IN_IQ || (cfg->bits != 16 && cfg->bits != 8) || cfg->opt_dc || cfg->opt_nolut || cfg->if_tune || cfg->keep_soft || cfg->opt_iqdc) return SONDE_E_ARG;
    const int C = cfg->n_channels;
    std::vector<std::vector<int32_t>> ch_of((size_t)n_groups);
    for (int c = 0; c < C; c++) {
        if (group_of_channel[c] < 0 || group_of_channel[c] >= n_groups) return SONDE_E_ARG;
        ch_of[(size_t)group_of_channel[c]].push_back(c);
    }
    for (int g = 0; g < n_groups; g++) {
        const int t = groups[g].sonde_type;
        if (t != SONDE_RS41 && t != SONDE_DFM09 && t != SONDE_M10 && t != SONDE_M20) return SONDE_E_ARG;
    }
    // One stream B for all groups when their IF-rate stages share launches (k_*_multi: one IF chain, one plan / window transform / frame sync per round over all rows) —
    // the default for up to SONDE_MAX_GROUPS groups; SONDE_MIXED_SPLIT=1: every group enqueues its own kernels on its own stream B (the A/B arm)
    static const bool split = getenv("SONDE_MIXED_SPLIT") != nullptr;
    int n_used = 0;
    for (int g = 0; g < n_groups; g++) n_used += ch_of[(size_t)g].empty() ? 0 : 1;
    const bool merged = !split && n_used <= SONDE_MAX_GROUPS;
    hipStream_t shared_b = nullptr;
    if (merged) {
        int lo = 0, hi = 0;
        HIPCHK(hipSetDevice(cfg->device));
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) HIPCHK(hipStreamCreateWithPriority(&shared_b, hipStreamNonBlocking, hi));
        else HIPCHK(hipStreamCreateWithFlags(&shared_b, hipStreamNonBlocking));
    }
    // groups first (their rings tell how long the shared y ring must be), then the owner of the front end
    std::vector<sonde_engine *> parts; std::vector<int> part_grp;
    auto drop = [&]() { for (sonde_engine *g : parts) sonde_engine_destroy(g); parts.clear(); part_grp.clear(); };
    int min_ring = 0;
    for (int pass = 0; pass < 2; pass++) {
        int ring = 0; bool same = true;
        for (int g = 0; g < n_groups; g++) {
            const std::vector<int32_t> &chs = ch_of[(size_t)g];
            if (chs.empty()) continue;
            sonde_cfg_t c2 = *cfg;
            c2.n_channels = (int32_t)chs.size(); c2.sonde_type = groups[g].sonde_type; c2.ecc_level = groups[g].ecc_level; c2.thres = groups[g].thres;
            c2.lpiq_bw = groups[g].lpiq_bw; c2.opt_inv = groups[g].opt_inv; c2.opt_auto = groups[g].opt_auto; c2.m10_noskip = groups[g].m10_noskip;
            c2.pipeline = 1;                                          // every group has its own stream B (+ E): the tails of the types run beside each other
            c2.max_frames = cfg->max_frames > 0 ? std::max(16, (int)(((long long)cfg->max_frames * (long long)chs.size() + C - 1) / C)) : 0;
            std::vector<double> f2(chs.size());
            for (size_t i = 0; i < chs.size(); i++) f2[i] = fq[chs[i]];
            const CreateLink lk{ true, false, min_ring, shared_b };
            sonde_engine *ge = nullptr;
            const int rc = create_impl(&c2, f2.data(), nullptr, &lk, &ge);
            if (rc) { drop(); if (shared_b) hipStreamDestroy(shared_b); return rc; }
            parts.push_back(ge); part_grp.push_back(g);
            if (ring && ge->ring_len != ring) same = false;
            ring = std::max(ring, ge->ring_len);
        }
        if (parts.empty()) { if (shared_b) hipStreamDestroy(shared_b); return SONDE_E_ARG; }
        min_ring = ring;
        if (same) break;
        if (pass == 0) drop();                                        // once more, every part at the longest ring
    }
    // the owner: channels in row order = group after group
    std::vector<int32_t> in_row; std::vector<double> f_rows;
    for (size_t k = 0; k < parts.size(); k++) for (int32_t c : ch_of[(size_t)part_grp[k]]) { in_row.push_back(c); f_rows.push_back(fq[c]); }
    sonde_cfg_t cf = *cfg;
    cf.sonde_type = SONDE_RS41; cf.ecc_level = 0; cf.pipeline = 1; cf.max_frames = 16; cf.thres = 0.f; cf.lpiq_bw = 0; cf.opt_inv = 0; cf.opt_auto = 0;      // (the preset is never used: front_only)
    const CreateLink lf{ false, true, min_ring, shared_b };
    sonde_engine *e = nullptr;
    const int rc = create_impl(&cf, f_rows.data(), nullptr, &lf, &e);
    if (rc) { drop(); if (shared_b) hipStreamDestroy(shared_b); return rc; }
    e->cfg.sonde_type = SONDE_MIXED; e->merged = merged;
    for (sonde_engine *g : parts) if (merged && (!g->d_win || g->small_tail)) e->merged = false;      // (every group must be on the window-transform search, plain forms)
    e->groups = parts; e->ch_of_grp.resize(parts.size());
    e->grp_of_ch.assign((size_t)C, 0); e->row_of_ch.assign((size_t)C, 0);
    int row = 0;
    for (size_t k = 0; k < parts.size(); k++) {
        sonde_engine *g = parts[k];
        g->front = e; g->front_row = row; g->d_y = e->d_y + (size_t)row * e->ring_len;
        e->ch_of_grp[k] = ch_of[(size_t)part_grp[k]];
        for (size_t i = 0; i < e->ch_of_grp[k].size(); i++) { e->grp_of_ch[(size_t)e->ch_of_grp[k][i]] = (int32_t)k; e->row_of_ch[(size_t)e->ch_of_grp[k][i]] = row + (int32_t)i; }
        if (dalloc(&g->d_sum_map, e->ch_of_grp[k].size(), false) ||
            hipMemcpy(g->d_sum_map, e->ch_of_grp[k].data(), e->ch_of_grp[k].size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
        row += g->cfg.n_channels;
    }
    if (e->ring_len != min_ring || dalloc(&e->d_in_row, in_row.size(), false) ||
        hipMemcpy(e->d_in_row, in_row.data(), in_row.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { sonde_engine_destroy(e); return SONDE_E_NOMEM; }
    *out = e;
    return 0;
}

int sonde_engine_group_info(const sonde_engine_t *e, int32_t channel, int32_t *sonde_type, sonde_info_t *info) {
    if (!e || channel < 0 || channel >= e->cfg.n_channels) return SONDE_E_ARG;
    const sonde_engine *g = e->groups.empty() ? e : e->groups[(size_t)e->grp_of_ch[(size_t)channel]];
    if (sonde_type) *sonde_type = g->cfg.sonde_type;
    if (info) *info = g->info;
    return 0;
}

}  // extern "C"
