// sonde_fsk.hip — gfx950 kernels of the batched 2-/4-FSK modem (the reference's utils/fsk.c fsk_demod_core).
//
// Two schedules of the same arithmetic:
//   k_fsk_stream   the workgroup's four waves as a pipeline around the serial oscillator walk (producer / consumer / two estimators,
//                  further down) — the one that runs for every sonde configuration (Ndft <= 256)
//   k_fsk_demod    frame at a time, every stage on all 256 threads with workgroup barriers between them — for longer transforms
// k_fsk_demod: one workgroup per channel walks the modem frames that fit into the samples queued for that channel — the frame loop
// is sequential in the reference too (nin, the smoothed spectrum Sf and the oscillator phases feed the next frame).
// Inside a frame everything that is data-parallel runs on 256 threads out of LDS:
//   frequency estimator   half-overlapped Hann-windowed FFTs (one wave per block), magnitude, per-bin exponential
//                         smoothing in block order, peak / mask search                              fsk.c:438-590
//   down-conversion       f_dc = in * conj(phi_c), phi_c advanced by the reference's float recurrence — kept serial on
//                         one lane per tone because its rounding drift (~2e-4 over a frame) is part of the output
//   integrate             (nsym+1) P sliding sums of Ts samples per tone                              fsk.c:659-668
//   fine timing           |f_int|^2 against the tabulated spectral-line oscillator, atan2             fsk.c:682-731
//   soft decisions        linear interpolation of the integrators, |t0| - |t1|                        fsk.c:751-805
// Floating-point contraction is off in this file: the reference is plain C on x86-64 (separately rounded mul/add),
// and the serial sums keep its order, so f_dc / f_int / soft decisions reproduce it up to libm (atan2f, log10f).
#pragma clang fp contract(off)
#include "sonde_fsk_dev.h"
#include "sonde_fsk_wave.h"
#include <limits.h>
#include <cstdlib>
#include <cstring>
extern "C" int sonde_launch_fsk_old(const FskArgs *a, hipStream_t s);

#define WAVE 64
#define FMT_S16  1
#define FMT_CS16 2
#define FMT_CU8  3
#define FMT_CF32 4

__device__ __forceinline__ float2 cmult(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// first index of the maximum of v[lo..hi) with the reference's `if (v > max)` scan from max = 0; dflt if nothing is > 0
__device__ int block_argmax(const float *v, int lo, int hi, int dflt, float *s_rf, int *s_ri) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = 0.f; int bi = INT_MAX;
    for (int i = lo + tid; i < hi; i += FSK_THREADS) { const float x = v[i]; if (x > best) { best = x; bi = i; } }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __syncthreads();                                  // s_rf / s_ri may still be read from a previous call
    if (lane == 0) { s_rf[wave] = best; s_ri[wave] = bi; }
    __syncthreads();
    best = 0.f; bi = INT_MAX;
    for (int w = 0; w < FSK_THREADS / WAVE; w++) { const float ob = s_rf[w]; const int oi = s_ri[w]; if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; } }
    return bi == INT_MAX ? dflt : bi;
}

// the same search on one wave (no workgroup barrier): the next frame's estimator runs on a single wave beside the oscillator
__device__ __forceinline__ int wave_argmax(const float *v, int lo, int hi, int dflt, int lane) {
    float best = 0.f; int bi = INT_MAX;
    for (int i = lo + lane; i < hi; i += WAVE) { const float x = v[i]; if (x > best) { best = x; bi = i; } }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    return bi == INT_MAX ? dflt : bi;
}

// x / 1000.f for an integer-valued x of 16 bits, correctly rounded, in three instructions instead of the division's dozen: q = x r with
// r = fl(1/1000), then one Newton step on the exact residual, q + (x - 1000 q) r.  Checked against the division for all 65536 inputs
// (tests/test_fsk_div1000.py); the conversion runs for every sample in the frame's input pass and again, twice, in the estimator blocks.
__device__ __forceinline__ float div1000(const float x) {
    const float r = 1.0f / 1000.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 1000.0f, x), r, q);
}
// input sample p of channel ch as the modem sees it (fsk_demod.c:283-311)
__device__ __forceinline__ float2 fsk_sample(const FskArgs &a, int ch, uint32_t p) {
    if (a.format == FMT_CS16) {
        const uint32_t raw = reinterpret_cast<const uint32_t *>(a.in)[(size_t)ch * a.ring + p];
        return make_float2(div1000((float)(short)(raw & 0xffffu)), div1000((float)(((int)raw) >> 16)));
    }
    if (a.format == FMT_CF32) return reinterpret_cast<const float2 *>(a.in)[(size_t)ch * a.ring + p];      // the fsk.h seam: COMP samples as the caller scaled them
    if (a.format == FMT_S16) return make_float2(div1000((float)reinterpret_cast<const int16_t *>(a.in)[(size_t)ch * a.ring + p]), 0.f);
    const uint16_t raw = reinterpret_cast<const uint16_t *>(a.in)[(size_t)ch * a.ring + p];
    return make_float2(((float)(raw & 0xffu) - 127.0f) / 128.0f, ((float)(raw >> 8) - 127.0f) / 128.0f);
}

// the same with the channel's ring base already formed (the consumer's inner loop)
__device__ __forceinline__ float2 fsk_sample_at(const FskArgs &a, const void *base, uint32_t p) {
    if (a.format == FMT_CS16) {
        const uint32_t raw = reinterpret_cast<const uint32_t *>(base)[p];
        return make_float2(div1000((float)(short)(raw & 0xffffu)), div1000((float)(((int)raw) >> 16)));
    }
    if (a.format == FMT_CF32) return reinterpret_cast<const float2 *>(base)[p];
    if (a.format == FMT_S16) return make_float2(div1000((float)reinterpret_cast<const int16_t *>(base)[p]), 0.f);
    const uint16_t raw = reinterpret_cast<const uint16_t *>(base)[p];
    return make_float2(((float)(raw & 0xffu) - 127.0f) / 128.0f, ((float)(raw >> 8) - 127.0f) / 128.0f);
}

// one butterfly stage of the reference's transform on buf[Ndft], butterflies lt, lt + TPF, ...  (kiss_fft.c kf_bfly4 / kf_bfly2: separately
// rounded mul / add, so that Sf and with it every estimator decision is the reference's bit for bit)
__device__ __forceinline__ void fsk_stage(float2 *buf, const float2 *tw, const int p, const int m, const int fs, const int Ndft, const int lt, const int TPF) {
    const int lm = __builtin_ctz((unsigned)m);               // m is a power of two (Ndft is)
    for (int b = lt; b < Ndft / p; b += TPF) {
        const int blk = b >> lm, u = b & (m - 1);
        float2 *F = buf + blk * p * m + u;
        if (p == 4) {
            const float2 s0 = cmult(F[m], tw[u * fs]), s1 = cmult(F[2 * m], tw[2 * u * fs]), s2 = cmult(F[3 * m], tw[3 * u * fs]);
            float2 f0 = F[0];
            const float2 s5 = make_float2(f0.x - s1.x, f0.y - s1.y);
            f0 = cadd(f0, s1);
            const float2 s3 = cadd(s0, s2), s4 = make_float2(s0.x - s2.x, s0.y - s2.y);
            F[2 * m] = make_float2(f0.x - s3.x, f0.y - s3.y);
            F[0] = cadd(f0, s3);
            F[m] = make_float2(s5.x + s4.y, s5.y - s4.x);
            F[3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
        } else {
            const float2 t = cmult(F[m], tw[u * fs]), f0 = F[0];
            F[m] = make_float2(f0.x - t.x, f0.y - t.y);
            F[0] = cadd(f0, t);
        }
    }
}

#ifndef FSK_NAP
#define FSK_NAP 1               // s_sleep argument of the wait loops (64 clocks each)
#endif
#define FSK_SPIN_MAX (1u << 21)   // iterations of a wait loop (64 clocks of sleep each, ~60 ms) after which a wave gives up
#define FSK_AE 4               // transform bins per lane the ahead-estimator holds in registers: Ndft <= 256 (sondes: 64, 128, 256)
// barrier of the `n` estimator waves only (wave 0 is inside the oscillator walk and takes no part): a counter in LDS that only grows —
// every wave adds one and waits until the count says all have arrived for this `phase` (1, 2, ...).  DS operations of a wave execute in
// order, so what a wave wrote before its add is visible to whoever sees the count; all waves of a workgroup are resident, so the wait ends.
__device__ __forceinline__ void fsk_group_barrier(unsigned *cnt, const unsigned n, unsigned &phase, const int lane, unsigned *abort_flag = nullptr) {
    phase++;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        atomicAdd(cnt, 1u);
        unsigned spins = 0;                                       // a wait that long means a bug: give up instead of hanging the device —
        while (*reinterpret_cast<volatile unsigned *>(cnt) < n * phase && ++spins < FSK_SPIN_MAX) __builtin_amdgcn_s_sleep(FSK_NAP);
        if (spins >= FSK_SPIN_MAX && abort_flag) __atomic_store_n(abort_flag, 1u, __ATOMIC_RELAXED);     // — and say so: the channel reports frames = -1, nobody goes on with half an estimate
    }
    __builtin_amdgcn_wave_barrier();
}

// The frequency estimator of the frame that starts at input sample rd (fsk_demod_freq_est, fsk.c:438-590), run AHEAD on waves 1..NG while
// wave 0 walks the oscillator of the frame before — that walk is one lane per tone for thousands of dependent steps, and the estimator of the
// next frame needs nothing from it (only the samples and the smoothed spectrum).  Rounds of NG blocks, one per wave: window, transform,
// magnitude (left in the wave's scratch); then the waves share the BINS and apply the round's magnitudes to Sf in block order — the same
// arithmetic in the same order as the workgroup-wide form in the kernel, so either may produce a frame's estimate.
template <int M>
__device__ void fsk_estimate_ahead(const FskArgs &a, const int ch, const uint32_t rd, const int numffts, float2 *s_fb, const float2 *s_tw,
                                   float *s_Sf, float *s_Sc, float *Sf_g, float *o_fest, float2 *o_dphi, const int gw, const int NG, const int lane,
                                   unsigned *s_bar, unsigned &phase, unsigned *abort_flag = nullptr) {
    const int Ndft = a.Ndft;
    const float tc = a.tc, omt = 1 - tc;
    // a wave takes BPW blocks at a time, one per group of GL lanes, FSK_AE transform elements per lane: short transforms (Ndft 64 / 128) would
    // leave most lanes of a wave without a butterfly otherwise
    // (a.est_bpw: the launcher may allow fewer blocks per wave than the lanes could take, when the scratch has to be small)
    const int BPW = min(a.est_bpw > 0 ? a.est_bpw : 64, (FSK_AE * WAVE) / Ndft), GL = WAVE / BPW, sub = lane / GL, lt = lane - sub * GL;
    const int EPL = Ndft / GL;                                     // transform elements per lane, <= FSK_AE
    const int slot = gw * BPW + sub, RB = NG * BPW;               // this lane group's block within a round; blocks per round
    float2 *buf = s_fb + slot * Ndft;
    float *mag = reinterpret_cast<float *>(buf);                // the block's magnitudes, fftshifted, over the first half of its scratch
    const int gt = gw * WAVE + lane, GT = NG * WAVE;             // thread index / count of the group
    // window, permutation and the samples of the group's NEXT block live in registers: a round then waits for LDS only
    float hn[FSK_AE]; int pm[FSK_AE]; float2 xs[FSK_AE];
#pragma unroll
    for (int e = 0; e < FSK_AE; e++) { const int i = lt + e * GL; hn[e] = 0.f; pm[e] = 0; xs[e] = make_float2(0.f, 0.f); if (e < EPL) { hn[e] = a.hann[i]; pm[e] = a.perm[i]; } }
    auto fetch = [&](const int j) {
#pragma unroll
        for (int e = 0; e < FSK_AE; e++) {
            const int i = lt + e * GL;
            if (e < EPL && j < numffts) xs[e] = fsk_sample(a, ch, (rd + (uint32_t)(i + j * (Ndft / 2))) & (a.ring - 1));
        }
    };
    fetch(slot);
#define EST_MARK(k) do { if (a.prof && ch == 0 && gw == 0 && lane == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); a.prof[k] += t_ - te_; te_ = t_; } } while (0)
    unsigned long long te_ = (a.prof && ch == 0) ? __builtin_readcyclecounter() : 0ull;
    for (int j0 = 0; j0 < numffts; j0 += RB) {
        const int j = j0 + slot;
        // (a wave whose groups have no block left in the last round still walks the stages: its lanes are masked by `act`)
        const bool act = j < numffts;
        if (act) {
#pragma unroll
            for (int e = 0; e < FSK_AE; e++) if (e < EPL) buf[pm[e]] = make_float2(hn[e] * xs[e].x, hn[e] * xs[e].y);
        }
        fetch(j + RB);                                         // in flight during the transform
        __builtin_amdgcn_wave_barrier();
        EST_MARK(10);
        for (int s = 0; s < a.n_stage; s++) { if (act) fsk_stage(buf, s_tw, a.st_p[s], a.st_m[s], a.st_fs[s], Ndft, lt, GL); __builtin_amdgcn_wave_barrier(); }
        EST_MARK(11);
        // magnitudes in place: every lane reads its bins first, then the wave writes (mag[q] overlays buf[q / 2])
        float mg[FSK_AE];
#pragma unroll
        for (int e = 0; e < FSK_AE; e++) { mg[e] = 0.f; if (act && e < EPL) { const float2 X = buf[lt + e * GL]; mg[e] = sqrtf((X.x * X.x) + (X.y * X.y)); } }
        __builtin_amdgcn_wave_barrier();
        if (act) {
#pragma unroll
            for (int e = 0; e < FSK_AE; e++) if (e < EPL) mag[(lt + e * GL + Ndft / 2) & (Ndft - 1)] = mg[e];
        }
        EST_MARK(12);
        fsk_group_barrier(s_bar, (unsigned)NG, phase, lane, abort_flag);
        EST_MARK(13);
        const int nb = min(RB, numffts - j0);
        for (int k = gt; k < Ndft; k += GT) {                  // Sf = Sf (1 - tc) + |X| tc, block after block (fsk.c:497-503)
            float sf = s_Sf[k];
            for (int g = 0; g < nb; g++) sf = (sf * omt) + (reinterpret_cast<const float *>(s_fb + g * Ndft)[k] * tc);
            s_Sf[k] = sf;
        }
        EST_MARK(12);
        fsk_group_barrier(s_bar, (unsigned)NG, phase, lane, abort_flag);
        EST_MARK(13);
    }
    if (gw != 0) return;
    // the searches are short: one wave
    for (int k = lane; k < Ndft; k += WAVE) { const float sf = s_Sf[k]; Sf_g[k] = sf; s_Sc[k] = sf; }
    __builtin_amdgcn_wave_barrier();
    float f_est[4]; float2 dphi[4];
    {
        int freqi[4];
        for (int m = 0; m < M; m++) {
            const int imax = wave_argmax(s_Sc, a.st, a.en, 0, lane);
            const int f_min = max(imax - a.f_zero, 0), f_max = min(imax + a.f_zero, Ndft);
            __builtin_amdgcn_wave_barrier();
            for (int k = f_min + lane; k < f_max; k += WAVE) s_Sc[k] = 0.f;
            __builtin_amdgcn_wave_barrier();
            freqi[m] = imax - Ndft / 2;
        }
        for (int i = 1; i < M; i++)
            for (int j = i; j > 0 && freqi[j] < freqi[j - 1]; j--) { const int t = freqi[j]; freqi[j] = freqi[j - 1]; freqi[j - 1] = t; }
        for (int m = 0; m < M; m++) { f_est[m] = (float)freqi[m] * ((float)a.Fs / (float)Ndft); dphi[m] = a.dphi_peak[freqi[m] + Ndft / 2]; }
    }
    if (a.est_type) {
        for (int b = a.st + lane; b < a.en - a.len_mask; b += WAVE) {
            float corr = 0.0f;
            for (int i = 0; i < a.n_mask; i++) corr += s_Sf[b + a.mask_idx[i]];
            s_Sc[b] = corr;
        }
        __builtin_amdgcn_wave_barrier();
        const int b_max = wave_argmax(s_Sc, a.st, a.en - a.len_mask, a.st, lane);
        for (int m = 0; m < M; m++) { f_est[m] = a.f_mask[M * b_max + m]; dphi[m] = a.dphi_mask[M * b_max + m]; }
    }
    if (lane == 0) for (int m = 0; m < M; m++) { o_fest[m] = f_est[m]; o_dphi[m] = dphi[m]; }
    EST_MARK(14);
}

// one step of the local oscillator phi *= d on the register pair %0 with temporaries %1, %2 and d = %3; the result goes to LDS at %4 + 8 k.
// A lone wave issues an instruction every ~8 cycles whatever it depends on, so the step is three packed instructions instead of six plain ones:
//   A = phi.x (d.x, d.y),  B = phi.y (d.y, d.x),  phi = (A.x - B.x, A.y + B.y)     — the four products and two sums of cmult(), each rounded once
#define FSK_OSC1(k) "v_pk_mul_f32 %1, %0, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
                    "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]\n\tds_write_b64 %4, %0 offset:" #k "*8\n\t"
#define FSK_OSC8 FSK_OSC1(0) FSK_OSC1(1) FSK_OSC1(2) FSK_OSC1(3) FSK_OSC1(4) FSK_OSC1(5) FSK_OSC1(6) FSK_OSC1(7)

// profiling aid: thread 0 of channel 0 adds the shader-clock cycles since the previous mark to phase k
#define FSK_MARK(k) do { if (a.prof && ch == 0 && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); a.prof[k] += t_ - t_prev; t_prev = t_; } } while (0)

// BIG: the regions live in global memory (a.scratch, one slice per workgroup) instead of LDS — frames of more samples than a CU's LDS holds (fsk_demod takes any
// Fs / Rs / nsym; the sondes' own configurations all fit).  Same code, same barriers: a workgroup's wavefronts share their CU's vector cache, so what one has
// written before a barrier the others read behind it.  est_waves is 0 there (no estimator running ahead: its hand-over counts on LDS ordering).
template <int M, bool BIG = false>
__global__ __launch_bounds__(FSK_THREADS)
void k_fsk_demod(const FskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];      // 16-byte base whatever the static arrays in front of it add up to (float2 / b64 accesses everywhere)
    float *lds;
    if constexpr (BIG) lds = a.scratch + (size_t)blockIdx.x * (size_t)a.scratch_stride; else lds = lds_dyn;
    __shared__ float s_rf[FSK_THREADS / WAVE]; __shared__ int s_ri[FSK_THREADS / WAVE];
    __shared__ float2 s_phi[4]; __shared__ float s_tc[2], s_eb[2];
    __shared__ float s_nfest[4]; __shared__ float2 s_ndphi[4];            // the next frame's estimate, when it was made ahead
    __shared__ unsigned s_bar;                                              // arrival count of the estimator waves (fsk_group_barrier)
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ts = a.Ts, P = a.P, nsym = a.nsym, N = a.N, Ndft = a.Ndft, Nmem = a.Nmem, NT = a.NT;
    const int W = (nsym + 1) * P;
    const int n_in = max(N + Ts / 2, W);
    // LDS regions (two workgroups per CU must fit: a 300-symbol RS41 frame needs 73 KB this way, 97 KB with one region per array):
    //   A  [nA]  the input frame s_in until the down-conversion has consumed it, then the integrators f_int[M][W] and the Eb/N0 terms
    //   B  [nB]  during the estimator: FFT scratch [4 Ndft] + block magnitudes mag[max_fft][Ndft]; then f_dc[M][Nmem]; then ft1 * phi_ft [W]
    //   Sf, Sc   smoothed spectrum and its search copy [Ndft] each
    const int nA = max(n_in, M * W + (nsym + 1));           // f_int[M][W] + the two Eb/N0 term arrays (nsym floats each)
    const int nB = max(max(M * Nmem, 4 * Ndft + (a.max_fft * Ndft + 1) / 2), W);
    float2 *s_in  = reinterpret_cast<float2 *>(lds);
    float2 *s_fint = s_in;
    float  *s_ebv = reinterpret_cast<float *>(s_fint + M * W);
    float2 *s_fdc = s_in + nA;
    float  *s_mag = reinterpret_cast<float *>(s_fdc + 4 * Ndft);
    float2 *s_ft  = s_fdc;
    float  *s_Sf = reinterpret_cast<float *>(s_fdc + nB), *s_Sc = s_Sf + Ndft;
    float2 *s_tw = reinterpret_cast<float2 *>(s_Sc + Ndft), *s_fb = s_tw + Ndft;      // twiddles; transform scratch of the estimator that runs ahead
    FskChan st = a.chan[ch];
    float *Sf_g = a.Sf + (size_t)ch * Ndft;
    float2 *tail_g = a.tail + (size_t)ch * M * NT;
    int frames = 0;
    const uint32_t wr = a.wr_ch ? a.wr_ch[ch] : a.wr;
    unsigned long long t_prev = a.prof ? __builtin_readcyclecounter() : 0ull;
    for (int k = tid; k < Ndft; k += FSK_THREADS) s_tw[k] = a.tw[k];
    bool have_est = false;                    // this frame's estimate was made while the previous frame's oscillator ran
    // frames whose length may differ by +-Ts/2 have the same estimator blocks when all three lengths give the same block count
    const bool same_blocks = a.burst || ((N - Ts / 2) / (Ndft / 2) == (N + Ts / 2) / (Ndft / 2));

    for (;;) {
        const int nin = st.nin;
        if ((int32_t)(wr - st.rd) < nin) break;
        if (frames >= a.rec_cap || (frames + 1) * nsym * (M / 2) > a.sd_cap) break;
        // ---- input conversion (fsk_demod.c:283-311)
        for (int i = tid; i < nin; i += FSK_THREADS) s_in[i] = fsk_sample(a, ch, (st.rd + (uint32_t)i) & (a.ring - 1));
        if (tid == 0) s_bar = 0;
        __syncthreads();
        FSK_MARK(0);

        float f_est[4]; float2 dphi[4];
        if (have_est) {
            for (int m = 0; m < M; m++) { f_est[m] = s_nfest[m]; dphi[m] = s_ndphi[m]; }
        } else {
        // ---- frequency estimator (fsk_demod_freq_est): numffts half-overlapped windowed FFTs, one wave each
        const int numffts = nin / (Ndft / 2) - 1;
        // threads per transform: one wave each while there are four blocks to do at a time, more when a frame has fewer (a 150-symbol M10 frame
        // has ONE block: all 256 threads work on it) — the stages are separated by workgroup barriers either way, so any split is the same arithmetic
        const int TPF = numffts >= 4 ? WAVE : numffts >= 2 ? 2 * WAVE : FSK_THREADS, par = FSK_THREADS / TPF;
        const int sub = tid / TPF, lt = tid - sub * TPF;
        for (int j0 = 0; j0 < numffts; j0 += par) {
            const int j = j0 + sub;
            const bool act = j < numffts;
            float2 *buf = s_fdc + sub * Ndft;
            if (act) for (int i = lt; i < Ndft; i += TPF) {
                const float h = a.hann[i]; const float2 x = s_in[i + j * (Ndft / 2)];
                buf[a.perm[i]] = make_float2(h * x.x, h * x.y);
            }
            __syncthreads();
            // the reference's transform, butterfly for butterfly (kiss_fft.c kf_work / kf_bfly4 / kf_bfly2: radix-4 stages, one radix-2
            // stage when log2 Ndft is odd, innermost factor first, separately rounded mul / add) so that Sf and with it every estimator
            // decision is the reference's bit for bit
            for (int s = 0; s < a.n_stage; s++) { if (act) fsk_stage(buf, s_tw, a.st_p[s], a.st_m[s], a.st_fs[s], Ndft, lt, TPF); __syncthreads(); }
            // fftshift (DC at Ndft/2) and magnitude
            if (act) for (int k = lt; k < Ndft; k += TPF) {
                const float2 X = buf[k];
                s_mag[j * Ndft + ((k + Ndft / 2) & (Ndft - 1))] = sqrtf((X.x * X.x) + (X.y * X.y));
            }
            __syncthreads();
        }
        FSK_MARK(1);
        // Sf = Sf (1 - tc) + |X| tc, block after block (fsk.c:497-503)
        for (int k = tid; k < Ndft; k += FSK_THREADS) {
            float sf = Sf_g[k];
            const float tc = a.tc, omt = 1 - tc;
            for (int j = 0; j < numffts; j++) sf = (sf * omt) + (s_mag[j * Ndft + k] * tc);
            Sf_g[k] = sf; s_Sf[k] = sf; s_Sc[k] = sf;
        }
        __syncthreads();
        // peak estimator: the M largest bins in [st, en), +-f_zero blanked after each, ascending (fsk.c:508-546)
        {
            int freqi[4];
            for (int m = 0; m < M; m++) {
                const int imax = block_argmax(s_Sc, a.st, a.en, 0, s_rf, s_ri);
                const int f_min = max(imax - a.f_zero, 0), f_max = min(imax + a.f_zero, Ndft);
                __syncthreads();
                for (int k = f_min + tid; k < f_max; k += FSK_THREADS) s_Sc[k] = 0.f;
                __syncthreads();
                freqi[m] = imax - Ndft / 2;
            }
            for (int i = 1; i < M; i++)                                     // the reference's gnome sort: ascending
                for (int j = i; j > 0 && freqi[j] < freqi[j - 1]; j--) { const int t = freqi[j]; freqi[j] = freqi[j - 1]; freqi[j - 1] = t; }
            for (int m = 0; m < M; m++) { f_est[m] = (float)freqi[m] * ((float)a.Fs / (float)Ndft); dphi[m] = a.dphi_peak[freqi[m] + Ndft / 2]; }
        }
        // mask estimator: M 3-bin groups fs_tx apart dragged over Sf (fsk.c:551-581)
        if (a.est_type) {
            for (int b = a.st + tid; b < a.en - a.len_mask; b += FSK_THREADS) {
                float corr = 0.0f;
                for (int i = 0; i < a.n_mask; i++) corr += s_Sf[b + a.mask_idx[i]];      // the non-zero mask entries, ascending
                s_Sc[b] = corr;
            }
            __syncthreads();
            const int b_max = block_argmax(s_Sc, a.st, a.en - a.len_mask, a.st, s_rf, s_ri);
            for (int m = 0; m < M; m++) { f_est[m] = a.f_mask[M * b_max + m]; dphi[m] = a.dphi_mask[M * b_max + m]; }
        }
        __syncthreads();

        }
        FSK_MARK(2);
        // ---- down-conversion with continuous phase (fsk.c:633-656); the oscillator recurrence stays serial
        const int nold = Nmem - nin;
        // the next frame's estimate is made now, on wave 1, if that frame is certain to be demodulated by this launch (samples queued for its
        // longest form, room for its outputs) and its blocks do not depend on the length this frame's timing will choose for it
        const uint32_t rd_next = st.rd + (uint32_t)nin;
        const int NG = a.est_waves;                         // waves 1..NG (0: no room in LDS for their scratch — every frame estimates for itself)
        const bool est_ahead = NG > 0 && same_blocks && (int32_t)(wr - rd_next) >= (a.burst ? N : N + Ts / 2)
                               && frames + 1 < a.rec_cap && (frames + 2) * nsym * (M / 2) <= a.sd_cap;
        if (wave == 0 && lane < M) {
            float2 phi = st.phi_c[0], d = dphi[0];
            for (int m = 1; m < M; m++) if (lane == m) { phi = st.phi_c[m]; d = dphi[m]; }
            float2 *o = s_fdc + lane * Nmem + nold;
            // one dependent complex multiply per sample — the chain cannot be shortened: every product and sum is rounded like the reference's
            // (cmult: x = a.x b.x - a.y b.y, y = a.x b.y + a.y b.x, no contraction); eight steps per statement (FSK_OSC1)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f ph = {phi.x, phi.y};
            const v2f dd = {d.x, d.y};
            uint32_t oaddr = (uint32_t)reinterpret_cast<uintptr_t>(o);
            int j = 0;
            if constexpr (!BIG) {                                   // (the statement writes through LDS addresses; the global-memory form walks in C++: the same products and sums)
                for (; j + 8 <= nin; j += 8) {
                    v2f ta, tb;
                    asm volatile(FSK_OSC8 : "+v"(ph), "=&v"(ta), "=&v"(tb) : "v"(dd), "v"(oaddr) : "memory");
                    oaddr += 64;
                }
            }
            const float pr = ph.x, pi = ph.y;
            phi = make_float2(pr, pi);
            for (; j < nin; j++) { phi = cmult(phi, d); o[j] = phi; }
            const float av = sqrtf((phi.x * phi.x) + (phi.y * phi.y));
            s_phi[lane] = make_float2(phi.x / av, phi.y / av);
        } else if (tid >= WAVE) {
            for (int m = 0; m < M; m++)
                for (int i = tid - WAVE; i < nold; i += FSK_THREADS - WAVE) s_fdc[m * Nmem + i] = tail_g[m * NT + (NT - nold) + i];
            if (est_ahead && wave <= NG) {
                unsigned phase = 0;                                 // (s_bar was cleared at the top of the frame)
                fsk_estimate_ahead<M>(a, ch, rd_next, N / (Ndft / 2) - 1, s_fb, s_tw, s_Sf, s_Sc, Sf_g, s_nfest, s_ndphi, wave - 1, NG, lane, &s_bar, phase);
            }
        }
        __syncthreads();
        have_est = est_ahead;
        FSK_MARK(3);
        for (int m = 0; m < M; m++) st.phi_c[m] = s_phi[m];
        for (int j = tid; j < nin; j += FSK_THREADS) {
            const float2 x = s_in[j];
#pragma unroll
            for (int m = 0; m < M; m++) { const float2 p = s_fdc[m * Nmem + nold + j]; s_fdc[m * Nmem + nold + j] = cmult(x, make_float2(p.x, -p.y)); }
        }
        __syncthreads();
        for (int m = 0; m < M; m++) for (int i = tid; i < NT; i += FSK_THREADS) tail_g[m * NT + i] = s_fdc[m * Nmem + (Nmem - NT) + i];

        FSK_MARK(4);
        // ---- integrate over a symbol period at (nsym+1) P offsets (fsk.c:659-668)
        {
            const int step = Ts / P;                           // P divides Ts (fsk.c:129), so i * Ts / P = i * step
            for (int m = 0; m < M; m++)
                for (int i = tid; i < W; i += FSK_THREADS) {
                    const float2 *f = s_fdc + m * Nmem + i * step;
                    float2 acc = make_float2(0.f, 0.f);
                    for (int j = 0; j < Ts; j++) acc = cadd(acc, f[j]);
                    s_fint[m * W + i] = acc;
                }
        }
        __syncthreads();
        FSK_MARK(5);
        // ---- fine timing: sum_i (sum_m |f_int[m]|^2) phi_ft[i]  (fsk.c:682-703)
        for (int i = tid; i < W; i += FSK_THREADS) {
            float ft1 = 0;
            for (int m = 0; m < M; m++) { const float2 v = s_fint[m * W + i]; ft1 += (v.x * v.x) + (v.y * v.y); }
            const float2 ph = a.phi_ft[i];
            s_ft[i] = make_float2(ft1 * ph.x, ft1 * ph.y);
        }
        __syncthreads();
        if (wave == 0 && lane < 2) {
            const float *pp = reinterpret_cast<const float *>(s_ft) + lane;
            float t = 0;                                // serial sum in the reference's order; a batch of LDS reads first, then the dependent adds
            int i = 0;
            for (; i + 16 <= W; i += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = pp[2 * (i + k)];
#pragma unroll
                for (int k = 0; k < 16; k++) t = t + v[k];
            }
            for (; i < W; i++) t = t + pp[2 * i];
            s_tc[lane] = t;
        }
        __syncthreads();
        const float norm_rx_timing = (float)((double)fsk_atan2f(s_tc[1], s_tc[0]) / (2 * 3.14159265358979323846));
        const float rx_timing = norm_rx_timing * (float)P;
        const float d_norm = norm_rx_timing - st.norm_rx_timing;
        st.norm_rx_timing = norm_rx_timing;
        if (fabsf(d_norm) < .2) {
            const float appm = (float)(1e6 * d_norm / (float)nsym);
            st.ppm = (float)(.9 * st.ppm + .1 * appm);
        }
        int nin_next = N;
        if (!a.burst) {
            if (norm_rx_timing > 0.25) nin_next = N + Ts / 2;
            else if (norm_rx_timing < -0.25) nin_next = N - Ts / 2;
        }

        FSK_MARK(6);
        // ---- soft decisions: integrators resampled by linear interpolation (fsk.c:733-805)
        const int low = (int)floorf(rx_timing), high = (int)ceilf(rx_timing);
        const float fract = rx_timing - (float)low, omf = 1 - fract;
        float *sd = a.sd + (size_t)ch * a.sd_cap + (size_t)frames * nsym * (M / 2);
        uint8_t *hb = a.hb + (size_t)ch * a.sd_cap + (size_t)frames * nsym * (M / 2);
        for (int i = tid; i < nsym; i += FSK_THREADS) {
            const int sp = (i + 1) * P;
            float tmax[4];
            for (int m = 0; m < M; m++) {
                const float2 lo = s_fint[m * W + sp + low], hi = s_fint[m * W + sp + high];
                const float2 t = cadd(make_float2(omf * lo.x, omf * lo.y), make_float2(fract * hi.x, fract * hi.y));
                tmax[m] = (t.x * t.x) + (t.y * t.y);
            }
            float mx = tmax[0]; int sym = 0;                                // first maximum wins (fsk.c:760-768)
            for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
            s_ebv[i] = mx; s_ebv[nsym + i] = sqrtf(mx);          // both Eb/N0 terms of the symbol (the roots are taken here, in parallel; the sums stay serial)
            if (M == 2) { sd[i] = sqrtf(tmax[0]) - sqrtf(tmax[1]); hb[i] = (uint8_t)(sym == 1); }
            else {
                hb[2 * i + 1] = (uint8_t)(sym & 1); hb[2 * i] = (uint8_t)((sym & 2) >> 1);                                                          // 4-FSK: two soft bits per symbol, summed in the reference's order (fsk.c:793-802)
                const float t0 = sqrtf(tmax[0]), t1 = sqrtf(tmax[1]), t2 = sqrtf(tmax[2]), t3 = sqrtf(tmax[3]);
                float lsb = -t0, msb = -t0;
                lsb += t1; msb += -t1;
                lsb += -t2; msb += t2;
                lsb += t3; msb += t3;
                sd[2 * i + 1] = lsb; sd[2 * i] = msb;
            }
        }
        // eye diagram samples (fsk.c:857-889): 4 traces of two symbols per tone, |f_int[m][2 P i + high + 1 + j dec]|; the
        // reference overwrites them every frame, normalisation happens when the stats are read
        if (a.eye) {
            const int dec = (int)ceilf(((float)P * 2) / 160.0f), nes = (P * 2) / dec;
            float *eye = a.eye + (size_t)ch * 8 * 160;
            for (int q = tid; q < 8 * nes; q += FSK_THREADS) {
                const int row = q / nes, j = q - row * nes, i = row / M, m = row - i * M;
                const int ind = 2 * P * i + high + 1 + j * dec;
                // high + 1 can be -1 (rx_timing in [-P/2, -2]): the reference then reads f_int[m][-1], i.e. the last integrator of
                // the previous tone for m > 0 and memory in front of the array for m = 0 (undefined there; 0 here)
                const float2 v = (ind < W && m * W + ind >= 0) ? s_fint[m * W + ind] : make_float2(0.f, 0.f);
                eye[row * 160 + j] = sqrtf((v.x * v.x) + (v.y * v.y));
            }
        }
        __syncthreads();
        FSK_MARK(7);
        // EbNo estimate (fsk.c:807-836): serial sums in symbol order
        if (wave == 0 && lane < 2) {
            float acc = 0;
            int i = 0;                                   // lane 0: sum of the largest |t|^2 per symbol, lane 1: of their roots — the reference's order, reads batched
            for (; i + 16 <= nsym; i += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = s_ebv[lane * nsym + i + k];
#pragma unroll
                for (int k = 0; k < 16; k++) acc += v[k];
            }
            for (; i < nsym; i++) acc += s_ebv[lane * nsym + i];
            s_eb[lane] = acc;
        }
        __syncthreads();
        {
            const float meanebno = s_eb[1] / (float)nsym;
            float stdebno = (s_eb[0] / (float)nsym) - (meanebno * meanebno);
            if (stdebno > 0.0) stdebno = (float)sqrt((double)stdebno); else stdebno = 0.0f;
            st.EbNodB = -6 + (20 * log10f((float)((1e-6 + meanebno) / (1e-6 + stdebno))));
            st.snr_est = (float)(.5 * st.snr_est + .5 * st.EbNodB);
        }
        for (int m = 0; m < M; m++) st.f_est[m] = f_est[m];
        if (tid == 0) {
            FskFrameRec r; r.nin = nin; r.nin_next = nin_next; for (int m = 0; m < 4; m++) r.f_est[m] = m < M ? f_est[m] : 0.f;
            r.norm_rx_timing = norm_rx_timing; r.ppm = st.ppm; r.EbNodB = st.EbNodB; r.snr_est = st.snr_est;
            a.recs[(size_t)ch * a.rec_cap + frames] = r;
        }
        st.rd += (uint32_t)nin; st.samples += nin; st.nin = nin_next;
        frames++;
        __syncthreads();
        FSK_MARK(8);
    }
    if (tid == 0) { st.frames = frames; a.chan[ch] = st; }
}

// ------------------------------------------------------------------------------------------------
// k_fsk_stream: the same modem as a pipeline of the workgroup's four waves
// ------------------------------------------------------------------------------------------------
// k_fsk_demod spends most of a frame in the oscillator walk — one lane per tone, a dependent complex multiply per sample — with the other
// 250 threads idle, and keeps f_dc[M][Nmem] in LDS for it (48 KB of the 76 KB a 300-symbol RS41 frame takes: two channels per CU).  Here
// the walk never stops: every other step of the modem runs beside it, on what it has produced so far.
//   producer   (one wave, a lane per tone)  phi *= d for every sample of every frame of the launch, into a RING of R samples per tone;
//              a frame's length is only needed when the walk has done the shortest form of the frame (the fine timing of the frame before
//              decides it, long before), its frequency estimate when the frame starts (made ahead, below)
//   consumer   (one wave)  pieces of up to 64 samples as they appear: f_dc = in conj(phi) in place in the ring, the integrator windows that
//              are complete by then (a lane per window, Ts sequential adds each), their fine-timing products summed serially in window order;
//              behind a frame's last window: timing -> the next frame's length (published at once), soft decisions, Eb/N0, the frame record
//   estimators (two waves)  fsk_estimate_ahead for the next frame as soon as its start is known
// They meet through counters in LDS that only grow (FskPipe): samples produced, lowest sample still needed, frames whose length / estimate
// has been published.  A wave's DS operations execute in order, so data written before a counter is bumped is visible to whoever has seen
// the bump.  Every sum and product is the same operation in the same order as in k_fsk_demod (and in the reference), only the schedule
// differs; the windows read the f_dc stream where the frame-at-a-time form reads its copy of it.
// LDS: f_int[M][W] + ring + a few KB — 37 KB for RS41 instead of 76, so FOUR channels per CU are resident.
struct FskPipe {
    unsigned prod_pos;                 // stream samples the producer has written (0 = first new sample of the launch)
    int      cons_free;                // stream position below which the consumer needs nothing any more
    unsigned nin_seq;                  // frames whose length has been published: frame k when nin_seq > k; length 0 = the stream stops in front of k
    unsigned est_seq;                  // frames whose estimate has been published
    unsigned bar;                      // fsk_group_barrier of the estimator waves
    unsigned prod_done;
    unsigned abort;                    // a wave waited too long: everybody leaves, the channel reports frames = -1
    int      nin[2];                   // by frame parity
    float    f_est[2][4]; float2 dphi[2][4];
    float2   phi_end[4];
    float    tc[2], eb[2];
};
// (every lane reads the same word: the value is handed on as a scalar, so that the loops steered by it stay on the scalar unit)
__device__ __forceinline__ unsigned pipe_ld(const unsigned *p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(p, __ATOMIC_RELAXED)); }
__device__ __forceinline__ int pipe_ldi(const int *p) { return __builtin_amdgcn_readfirstlane(__atomic_load_n(p, __ATOMIC_RELAXED)); }
// false: another wave gave up, or this wait took so long that something is wrong — the caller leaves (the launch reports an error, no hang)
__device__ __forceinline__ bool pipe_wait_ge(const unsigned *p, const unsigned v, unsigned *abort_flag) {
    unsigned spins = 0;
    while ((int)(pipe_ld(p) - v) < 0) {
        if (++spins > FSK_SPIN_MAX || pipe_ld(abort_flag)) { __atomic_store_n(abort_flag, 1u, __ATOMIC_RELAXED); return false; }
        __builtin_amdgcn_s_sleep(FSK_NAP);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}
__device__ __forceinline__ void pipe_publish(unsigned *p, const unsigned v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __atomic_store_n(p, v, __ATOMIC_RELAXED);
}

// profiling aid (SONDE_FSK_PROF): cycles channel 0's waves spend in all / waiting; slot k of a.prof
#define PIPE_T0() (a.prof && ch == 0 ? __builtin_readcyclecounter() : 0ull)
#define PIPE_ADD(k, t0) do { if (a.prof && ch == 0 && lane == 0) a.prof[k] += __builtin_readcyclecounter() - (t0); } while (0)
template <int M>
#ifndef FSK_STREAM_WPE
#define FSK_STREAM_WPE 5      // waves per SIMD = workgroups per CU the registers allow: a thousand channels are exactly four per CU, and the
#endif                        // dispatcher does not spread three kernels that evenly — without a fifth place the last workgroups wait for a whole round
__global__ __launch_bounds__(FSK_THREADS) __attribute__((amdgpu_waves_per_eu(FSK_STREAM_WPE, FSK_STREAM_WPE)))
void k_fsk_stream(const FskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ FskPipe pp;
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ts = a.Ts, P = a.P, nsym = a.nsym, N = a.N, Ndft = a.Ndft, Nmem = a.Nmem, NT = a.NT, R = a.R;
    const int W = (nsym + 1) * P, step = Ts / P;
    float2 *s_ring = reinterpret_cast<float2 *>(lds);                      // [M][R]
    float2 *s_fint = s_ring + M * R;                                       // [M][W]
    float  *s_ebv  = reinterpret_cast<float *>(s_fint + M * W);            // [2][nsym] (+ pad to an even count)
    float2 *s_ftp  = reinterpret_cast<float2 *>(s_ebv + ((2 * nsym + 1) & ~1));   // [64] fine-timing products of a batch of windows
    float  *s_Sf   = reinterpret_cast<float *>(s_ftp + WAVE), *s_Sc = s_Sf + Ndft;
    float2 *s_tw   = reinterpret_cast<float2 *>(s_Sc + Ndft), *s_fb = s_tw + Ndft;
    FskChan st = a.chan[ch];
    float *Sf_g = a.Sf + (size_t)ch * Ndft;
    float2 *tail_g = a.tail + (size_t)ch * M * NT;
    const uint32_t wr = a.wr_ch ? a.wr_ch[ch] : a.wr, rd0 = st.rd;
    const uint32_t rmask = (uint32_t)R - 1;
    // a frame k exists if its samples are queued and its outputs have room (the loop conditions of k_fsk_demod)
    auto frame_fits = [&](const int k, const uint32_t S, const int nin) -> bool {
        return (int32_t)(wr - (rd0 + S)) >= nin && k < a.rec_cap && (k + 1) * nsym * (M / 2) <= a.sd_cap;
    };

    for (int k = tid; k < Ndft; k += FSK_THREADS) { s_tw[k] = a.tw[k]; s_Sf[k] = Sf_g[k]; }
    for (int m = 0; m < M; m++) for (int i = tid; i < NT; i += FSK_THREADS) s_ring[m * R + ((uint32_t)(i - NT) & rmask)] = tail_g[m * NT + i];
    if (tid == 0) {
        pp.prod_pos = 0; pp.cons_free = -NT; pp.est_seq = 0; pp.bar = 0; pp.prod_done = 0; pp.abort = 0;
        pp.nin[0] = frame_fits(0, 0u, st.nin) ? st.nin : 0; pp.nin[1] = 0; pp.nin_seq = 1;
    }
    __syncthreads();
    // the roles rotate with the channel, so that the producers of the channels sharing a CU do not all sit on the same SIMD
    const int role = (wave + ch) & 3;

    if (role == 0) {
        // ---- producer: the oscillator walk (fsk.c:633-656), one lane per tone
        if (lane < M) {
            typedef float v2f __attribute__((ext_vector_type(2)));
            float2 phi = st.phi_c[0];
            for (int m = 1; m < M; m++) if (lane == m) phi = st.phi_c[m];
            float2 *ring = s_ring + lane * R;
            uint32_t pos = 0;
            const unsigned long long tp0 = PIPE_T0();
            for (int k = 0;; k++) {
                bool stop = false;
                const unsigned long long tw0 = PIPE_T0();
                for (unsigned spins = 0;; ) {                           // the frame's estimate — or the word that the stream ends here
                    if ((int)(pipe_ld(&pp.est_seq) - (unsigned)(k + 1)) >= 0) break;
                    if ((int)(pipe_ld(&pp.nin_seq) - (unsigned)(k + 1)) >= 0 && pipe_ldi(&pp.nin[k & 1]) == 0) { stop = true; break; }
                    if (++spins > FSK_SPIN_MAX || pipe_ld(&pp.abort)) { __atomic_store_n(&pp.abort, 1u, __ATOMIC_RELAXED); stop = true; break; }
                    __builtin_amdgcn_s_sleep(FSK_NAP);
                }
                PIPE_ADD(1, tw0);
                if (stop) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float2 d = pp.dphi[k & 1][0];
                for (int m = 1; m < M; m++) if (lane == m) d = pp.dphi[k & 1][m];
                v2f ph = {phi.x, phi.y};
                const v2f dd = {d.x, d.y};
                int remaining = a.burst ? N : N - Ts / 2;               // the shortest form of the frame; its real length is known by the time this is done
                bool have_nin = false;
                // (pos, remaining and everything that steers this loop are wave-uniform scalars; a vector compare per step would cost as much as the step)
                for (;;) {
                    while (remaining > 0) {
                        if ((pos & 63u) == 0) {                         // every 64 samples: say how far the walk is, make sure the ring has room for the next 64
                            pipe_publish(&pp.prod_pos, pos);
                            unsigned spins = 0;
                            const unsigned long long tw1 = PIPE_T0();
                            while ((int)(pos + 64u - (uint32_t)R) - pipe_ldi(&pp.cons_free) > 0) {
                                if (++spins > FSK_SPIN_MAX || pipe_ld(&pp.abort)) { __atomic_store_n(&pp.abort, 1u, __ATOMIC_RELAXED); break; }
                                __builtin_amdgcn_s_sleep(FSK_NAP);
                            }
                            PIPE_ADD(2, tw1);
                            if (pipe_ld(&pp.abort)) { remaining = 0; have_nin = true; break; }
                        }
                        if ((pos & 7u) == 0 && remaining >= 8) {
                            // groups of eight steps up to the next multiple of 64 (a ring of R samples, R a multiple of 64, does not wrap inside)
                            const int g = min((int)((64u - (pos & 63u)) >> 3), remaining >> 3);
                            uint32_t oaddr = (uint32_t)reinterpret_cast<uintptr_t>(ring + (pos & rmask));
                            for (int u = 0; u < g; u++) {
                                v2f ta, tb;
                                asm volatile(FSK_OSC8 : "+v"(ph), "=&v"(ta), "=&v"(tb) : "v"(dd), "v"(oaddr) : "memory");
                                oaddr += 64;
                            }
                            pos += 8u * (uint32_t)g; remaining -= 8 * g;
                        } else {
                            const float pr = ph.x, pi = ph.y;
                            const float2 q = cmult(make_float2(pr, pi), d);
                            ring[pos & rmask] = q;
                            ph.x = q.x; ph.y = q.y;
                            pos += 1; remaining -= 1;
                        }
                    }
                    if (have_nin) break;
                    const unsigned long long tw2 = PIPE_T0();
                    if (!pipe_wait_ge(&pp.nin_seq, (unsigned)(k + 1), &pp.abort)) break;
                    PIPE_ADD(3, tw2);
                    remaining = pipe_ldi(&pp.nin[k & 1]) - (a.burst ? N : N - Ts / 2);
                    have_nin = true;
                }
                if (pipe_ld(&pp.abort)) break;
                {   // end of the frame: phi /= |phi| (fsk.c:654-656)
                    const float pr = ph.x, pi = ph.y;
                    const float av = sqrtf((pr * pr) + (pi * pi));
                    phi = make_float2(pr / av, pi / av);
                }
                pipe_publish(&pp.prod_pos, pos);
            }
            pp.phi_end[lane] = phi;
            PIPE_ADD(0, tp0);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) pipe_publish(&pp.prod_done, 1u);
        return;
    }

    if (role >= 2) {
        // ---- estimators: the frequency estimate of every frame, as early as its start is known
        const int gw = role - 2;
        uint32_t S = 0;
        unsigned phase = 0;
        const bool same_blocks = a.burst || ((N - Ts / 2) / (Ndft / 2) == (N + Ts / 2) / (Ndft / 2));
        const unsigned long long te0 = PIPE_T0();
        for (int k = 0;; k++) {
            const unsigned long long tw0 = PIPE_T0();
            if (k > 0) {
                if (!pipe_wait_ge(&pp.nin_seq, (unsigned)k, &pp.abort)) break;
                const int prev = pipe_ldi(&pp.nin[(k - 1) & 1]);
                if (prev == 0) break;
                S += (uint32_t)prev;
            }
            // ahead of the frame's length: when it is certain to be demodulated whatever length the timing gives it, and its blocks are the same for all three
            const bool ahead = k > 0 && same_blocks && frame_fits(k, S, a.burst ? N : N + Ts / 2);
            int numffts = N / (Ndft / 2) - 1;
            if (!ahead) {
                if (!pipe_wait_ge(&pp.nin_seq, (unsigned)(k + 1), &pp.abort)) break;
                const int nin = pipe_ldi(&pp.nin[k & 1]);
                if (nin == 0) break;
                numffts = nin / (Ndft / 2) - 1;
            }
            if (gw == 0) PIPE_ADD(9, tw0);
            fsk_estimate_ahead<M>(a, ch, rd0 + S, numffts, s_fb, s_tw, s_Sf, s_Sc, Sf_g, pp.f_est[k & 1], pp.dphi[k & 1], gw, 2, lane, &pp.bar, phase, &pp.abort);
            if (pipe_ld(&pp.abort)) break;
            if (gw == 0 && lane == 0) pipe_publish(&pp.est_seq, (unsigned)(k + 1));
        }
        if (gw == 0) PIPE_ADD(8, te0);
        return;
    }

    // ---- consumer
    {
        uint32_t S = 0;
        int frames = 0;
        int nin = pipe_ldi(&pp.nin[0]);
        uint32_t E_last = 0;
        constexpr int WB = WAVE / M;                                    // integrator windows per pass
        typedef float v2f __attribute__((ext_vector_type(2)));
        const int unit = a.format == FMT_CS16 ? 4 : a.format == FMT_CF32 ? 8 : 2;
        const char *in_ch = reinterpret_cast<const char *>(a.in) + (size_t)ch * a.ring * unit;
        const int sshift = (step & (step - 1)) == 0 ? __builtin_ctz((unsigned)step) : -1;      // Ts / P is 1, 2 or 4 for the sondes
        const unsigned long long tc0_ = PIPE_T0();
        for (int k = 0; nin != 0; k++) {
            const uint32_t E = S + (uint32_t)nin;
            const int32_t wbase = (int32_t)E - Nmem;                    // stream position of f_dc[.][0] of this frame
            int i_done = 0;
            float t = 0;                                                // lanes 0 / 1: the timing sum (re / im)
            for (uint32_t c = S; c != E; ) {
                uint32_t ce = (c | 63u) + 1u; if ((int32_t)(ce - E) > 0) ce = E;      // pieces end where the producer reports (multiples of 64, frame ends)
                const int cl = (int)(ce - c);
                float2 x = make_float2(0.f, 0.f);
                if (lane < cl) x = fsk_sample_at(a, in_ch, (rd0 + c + (uint32_t)lane) & (a.ring - 1));
                // (both global reads of the piece are issued before the wait for the producer: the input sample and the timing phasor of the lane's next window)
                const float2 ph_next = (i_done + lane / M < W) ? a.phi_ft[i_done + lane / M] : make_float2(0.f, 0.f);
                const unsigned long long tw0 = PIPE_T0();
                if (!pipe_wait_ge(&pp.prod_pos, ce, &pp.abort)) { nin = 0; break; }
                PIPE_ADD(5, tw0);
                if (lane < cl) {
                    // x conj(p): the products x.x p.x, x.y p.y and x.y p.x, x.x p.y as two packed multiplies, then one add and one subtract —
                    // cmult(x, (p.x, -p.y)) value for value (negating a factor negates the rounded product)
                    const v2f xv = {x.x, x.y}, xs = {x.y, x.x};
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        v2f *q = reinterpret_cast<v2f *>(s_ring + m * R + ((c + (uint32_t)lane) & rmask));
                        const v2f p = *q, aa = xv * p, bb = xs * p;
                        *q = v2f{aa.x + aa.y, bb.x - bb.y};
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // integrator windows complete by now (fsk.c:659-668): window i reads f_dc[i step .. i step + Ts)
                const int32_t span = (int32_t)ce - wbase - Ts;
                int i_new = span < 0 ? 0 : (sshift >= 0 ? (span >> sshift) : span / step) + 1;
                if (i_new > W) i_new = W;
                // a lane per (window, tone): WB windows at a time.  The Ts samples of a window are summed in order; they are contiguous in the ring
                // unless the window straddles its end
                for (int ib = i_done; ib < i_new; ib += WB) {
                    const int wl = lane / M, m = lane - wl * M, i = ib + wl;
                    if (i < i_new) {
                        const v2f *rg = reinterpret_cast<const v2f *>(s_ring + m * R);
                        const uint32_t q0 = (uint32_t)(wbase + i * step) & rmask;
                        v2f acc = {0.f, 0.f};                           // (re, im) advance together: one packed add per sample, each component rounded as before
                        if (q0 + (uint32_t)Ts <= (uint32_t)R) {
                            const v2f *src = rg + q0;
                            int j = 0;
                            for (; j + 5 <= Ts; j += 5) {                // (Ts is 5, 10 or 20 for the sondes)
                                const v2f v0 = src[j], v1 = src[j + 1], v2 = src[j + 2], v3 = src[j + 3], v4 = src[j + 4];
                                acc += v0; acc += v1; acc += v2; acc += v3; acc += v4;
                            }
                            for (; j < Ts; j++) acc += src[j];
                        } else {
                            for (int j = 0; j < Ts; j++) acc += rg[(q0 + (uint32_t)j) & rmask];
                        }
                        s_fint[m * W + i] = make_float2(acc.x, acc.y);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i < i_new && m == 0) {                          // fine timing: sum_i (sum_m |f_int[m]|^2) phi_ft[i]  (fsk.c:682-703)
                        float ft1 = 0;
#pragma unroll
                        for (int m2 = 0; m2 < M; m2++) { const float2 v = s_fint[m2 * W + i]; ft1 += (v.x * v.x) + (v.y * v.y); }
                        const float2 ph = (ib == i_done) ? ph_next : a.phi_ft[i];
                        s_ftp[wl] = make_float2(ft1 * ph.x, ft1 * ph.y);
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int nb = min(WB, i_new - ib);
                    if (lane < 2) {
                        const float *pq = reinterpret_cast<const float *>(s_ftp) + lane;
                        int q = 0;
                        for (; q + 16 <= nb; q += 16) {
                            float v[16];
#pragma unroll
                            for (int u = 0; u < 16; u++) v[u] = pq[2 * (q + u)];
#pragma unroll
                            for (int u = 0; u < 16; u++) t = t + v[u];
                        }
                        for (; q < nb; q++) t = t + pq[2 * q];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (i_new > i_done) i_done = i_new;
                // what the producer may overwrite: everything below the first incomplete window — but never the last NT samples of the frame,
                // the history of the next one
                if (lane == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __atomic_store_n(&pp.cons_free, min(wbase + i_done * step, (int32_t)E - NT), __ATOMIC_RELAXED); }
                c = ce;
            }
            if (nin == 0) break;                                        // gave up waiting
            const unsigned long long tt0 = PIPE_T0();
            if (lane < 2) pp.tc[lane] = t;
            __builtin_amdgcn_wave_barrier();
            const float tc0 = pp.tc[0], tc1 = pp.tc[1];
            const float norm_rx_timing = (float)((double)fsk_atan2f(tc1, tc0) / (2 * 3.14159265358979323846));
            const float rx_timing = norm_rx_timing * (float)P;
            const float d_norm = norm_rx_timing - st.norm_rx_timing;
            st.norm_rx_timing = norm_rx_timing;
            if (fabsf(d_norm) < .2) {
                const float appm = (float)(1e6 * d_norm / (float)nsym);
                st.ppm = (float)(.9 * st.ppm + .1 * appm);
            }
            int nin_next = N;
            if (!a.burst) {
                if (norm_rx_timing > 0.25) nin_next = N + Ts / 2;
                else if (norm_rx_timing < -0.25) nin_next = N - Ts / 2;
            }
            float f_est[4];
            for (int m = 0; m < M; m++) f_est[m] = pp.f_est[k & 1][m];    // (published before the producer started this frame)
            // the next frame: its length, or 0 if it does not fit into this launch — the other waves go on (or stop) from here
            const bool more = frame_fits(k + 1, E, nin_next);
            if (lane == 0) {
                __atomic_store_n(&pp.nin[(k + 1) & 1], more ? nin_next : 0, __ATOMIC_RELAXED);
                __atomic_store_n(&pp.cons_free, (int32_t)E - NT, __ATOMIC_RELAXED);
                pipe_publish(&pp.nin_seq, (unsigned)(k + 2));
            }
            // ---- soft decisions: integrators resampled by linear interpolation (fsk.c:733-805)
            const int low = (int)floorf(rx_timing), high = (int)ceilf(rx_timing);
            const float fract = rx_timing - (float)low, omf = 1 - fract;
            float *sd = a.sd + (size_t)ch * a.sd_cap + (size_t)frames * nsym * (M / 2);
            uint8_t *hb = a.hb + (size_t)ch * a.sd_cap + (size_t)frames * nsym * (M / 2);
            for (int i = lane; i < nsym; i += WAVE) {
                const int sp = (i + 1) * P;
                float tmax[4];
                for (int m = 0; m < M; m++) {
                    const float2 lo = s_fint[m * W + sp + low], hi = s_fint[m * W + sp + high];
                    const float2 tt = cadd(make_float2(omf * lo.x, omf * lo.y), make_float2(fract * hi.x, fract * hi.y));
                    tmax[m] = (tt.x * tt.x) + (tt.y * tt.y);
                }
                float mx = tmax[0]; int sym = 0;                                // first maximum wins (fsk.c:760-768)
                for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
                s_ebv[i] = mx; s_ebv[nsym + i] = sqrtf(mx);
                if (M == 2) { sd[i] = sqrtf(tmax[0]) - sqrtf(tmax[1]); hb[i] = (uint8_t)(sym == 1); }
                else {
                    hb[2 * i + 1] = (uint8_t)(sym & 1); hb[2 * i] = (uint8_t)((sym & 2) >> 1);
                    const float t0 = sqrtf(tmax[0]), t1 = sqrtf(tmax[1]), t2 = sqrtf(tmax[2]), t3 = sqrtf(tmax[3]);
                    float lsb = -t0, msb = -t0;
                    lsb += t1; msb += -t1;
                    lsb += -t2; msb += t2;
                    lsb += t3; msb += t3;
                    sd[2 * i + 1] = lsb; sd[2 * i] = msb;
                }
            }
            if (a.eye) {                                                // eye diagram samples (fsk.c:857-889), see k_fsk_demod
                const int dec = (int)ceilf(((float)P * 2) / 160.0f), nes = (P * 2) / dec;
                float *eye = a.eye + (size_t)ch * 8 * 160;
                for (int q = lane; q < 8 * nes; q += WAVE) {
                    const int row = q / nes, j = q - row * nes, i = row / M, m = row - i * M;
                    const int ind = 2 * P * i + high + 1 + j * dec;
                    const float2 v = (ind < W && m * W + ind >= 0) ? s_fint[m * W + ind] : make_float2(0.f, 0.f);
                    eye[row * 160 + j] = sqrtf((v.x * v.x) + (v.y * v.y));
                }
            }
            __builtin_amdgcn_wave_barrier();
            // EbNo estimate (fsk.c:807-836): serial sums in symbol order
            if (lane < 2) {
                float acc = 0;
                int i = 0;
                for (; i + 16 <= nsym; i += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = s_ebv[lane * nsym + i + u];
#pragma unroll
                    for (int u = 0; u < 16; u++) acc += v[u];
                }
                for (; i < nsym; i++) acc += s_ebv[lane * nsym + i];
                pp.eb[lane] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            {
                const float meanebno = pp.eb[1] / (float)nsym;
                float stdebno = (pp.eb[0] / (float)nsym) - (meanebno * meanebno);
                if (stdebno > 0.0) stdebno = (float)sqrt((double)stdebno); else stdebno = 0.0f;
                st.EbNodB = -6 + (20 * log10f((float)((1e-6 + meanebno) / (1e-6 + stdebno))));
                st.snr_est = (float)(.5 * st.snr_est + .5 * st.EbNodB);
            }
            for (int m = 0; m < M; m++) st.f_est[m] = f_est[m];
            if (lane == 0) {
                FskFrameRec r; r.nin = nin; r.nin_next = nin_next; for (int m = 0; m < 4; m++) r.f_est[m] = m < M ? f_est[m] : 0.f;
                r.norm_rx_timing = norm_rx_timing; r.ppm = st.ppm; r.EbNodB = st.EbNodB; r.snr_est = st.snr_est;
                a.recs[(size_t)ch * a.rec_cap + frames] = r;
            }
            __builtin_amdgcn_wave_barrier();                            // s_fint / s_ebv / pp.tc are rewritten by the next frame
            PIPE_ADD(6, tt0);
            st.rd += (uint32_t)nin; st.samples += nin; st.nin = nin_next;
            frames++;
            S = E; E_last = E;
            nin = more ? nin_next : 0;
            if (a.test_abort_ch == ch && frames == 1 && lane == 0) __atomic_store_n(&pp.abort, 1u, __ATOMIC_RELAXED);      // (test hook: as if a wait had run out)
        }
        PIPE_ADD(4, tc0_);
        if (a.prof && ch == 0 && lane == 0) a.prof[15] = 1;             // (this kernel's slots, not k_fsk_demod's phases)
        // the launch is over for this channel: oscillator phases, the last NT f_dc samples, the channel state
        if (!pipe_wait_ge(&pp.prod_done, 1u, &pp.abort) || pipe_ld(&pp.abort)) {
            if (lane == 0) a.chan[ch].frames = -1;                      // the host turns this into an error
            return;
        }
        if (frames > 0) {
            for (int m = 0; m < M; m++) st.phi_c[m] = pp.phi_end[m];
            for (int m = 0; m < M; m++) for (int i = lane; i < NT; i += WAVE) tail_g[m * NT + i] = s_ring[m * R + ((E_last - (uint32_t)NT + (uint32_t)i) & rmask)];
        }
        if (lane == 0) { st.frames = frames; a.chan[ch] = st; }
    }
}

// ------------------------------------------------------------------------------------------------
// k_fsk_wave: walker + worker wavefronts meeting at barriers (sonde_fsk_wave.h) — the form that runs for every sonde configuration
// ------------------------------------------------------------------------------------------------
// Register budget: 96 vector registers (five wavefronts per SIMD, twenty per CU = five channels of four roles).  With the 128 the code would take if left alone a CU
// holds four channels, and a launch of more channels than that runs in rounds: 1026 channels of three configurations 2.2 ms at 128, 1.75 ms at 96; one
// configuration's 342 the same 1.0-1.2 ms at either (profiles/r5e_fsk_registers_ab.txt).  At 80 and below the worker's sums spill.  What made 96 possible without
// spills: the channel record outside the registers, the estimator's Sf backup in LDS, each role's loads behind the role dispatch, uniform bookkeeping in scalar
// registers (fw_uni) and the timing angle's arctangent as one dependent chain (fsk_atan2f).
template <int M, int LOG2N, bool SPLIT, int FMT>
__global__ __launch_bounds__(SPLIT ? 256 : 64) __attribute__((amdgpu_waves_per_eu(SPLIT ? 5 : 2, 8)))
void k_fsk_wave(const FskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ FwCtl ctl;
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x;
    fsk_wave_channel<M, LOG2N, SPLIT, FMT>(a, ch, (int)threadIdx.x, lds, ctl);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (function, device): the launchers remember what each device has been given
#define FSK_MAX_DEV 16
static int fsk_cur_dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0; return d % FSK_MAX_DEV; }

template <int M, int LOG2N, bool SPLIT, int FMT>
static int launch_wave_f(const FskArgs &b, const size_t lds_bytes, hipStream_t s) {
    static size_t attr[FSK_MAX_DEV] = {};                     // (the attribute is per device: one high-water mark each)
    size_t &hw = attr[fsk_cur_dev()];
    if (lds_bytes > hw) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fsk_wave<M, LOG2N, SPLIT, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
        hw = lds_bytes;
    }
    hipLaunchKernelGGL((k_fsk_wave<M, LOG2N, SPLIT, FMT>), dim3(b.n_ch), dim3(SPLIT ? (b.fin ? 256 : 192) : 64), lds_bytes, s, b);
    return 0;
}
// (cs16 — what auto_rx pipes in — gets a kernel of its own: with the format a constant there are no branches around the sample loads)
template <int M, int LOG2N, bool SPLIT>
static int launch_wave(const FskArgs &b, const size_t lds_bytes, hipStream_t s) {
    return b.format == FMT_CS16 ? launch_wave_f<M, LOG2N, SPLIT, FMT_CS16>(b, lds_bytes, s) : launch_wave_f<M, LOG2N, SPLIT, 0>(b, lds_bytes, s);
}
template <int M, bool SPLIT>
static int launch_wave_n(const FskArgs &b, const size_t lds_bytes, hipStream_t s) {
    return b.Ndft == 64 ? launch_wave<M, 6, SPLIT>(b, lds_bytes, s) : b.Ndft == 128 ? launch_wave<M, 7, SPLIT>(b, lds_bytes, s) : launch_wave<M, 8, SPLIT>(b, lds_bytes, s);
}

// which kernel a launch with these arguments gets: the wave form (mode 1 / 2; Ndft 64 / 128 / 256: every sonde configuration) or one of the older kernels (mode 0).
// SONDE_FSK_KERNEL = wave2 (walker + worker, the default) | wave1 (one wave per channel) | stream (round 3's pipeline of four waves) | demod (frame at a time) — an A/B aid
static int fsk_wave_plan(const FskArgs *a, int &R, int &fin, size_t &lds_w, bool &demod_env) {
    const int M = a->M, W = (a->nsym + 1) * a->P;
    static const char *k_env = getenv("SONDE_FSK_KERNEL");
    int mode = 2; demod_env = false;
    if (k_env) { mode = !strcmp(k_env, "wave1") ? 1 : !strcmp(k_env, "wave2") ? 2 : 0; demod_env = !strcmp(k_env, "demod"); }
    static const char *st_env0 = getenv("SONDE_FSK_STREAM");
    if (st_env0) mode = 0;                                                  // (the older switch between the two older kernels implies one of them)
    R = fw_ring_len(a->NT, a->Ts / a->P);
    // the finisher (a fourth wave: soft decisions, Eb/N0 and record of a frame while the worker is in the next) where f_int fits twice: the short frames of DFM / M10
    fin = 2 * (size_t)M * W * sizeof(float2) <= 16384 ? 1 : 0;
    lds_w = fw_lds_floats(M, a->nsym, a->P, R, a->Ndft, fin) * sizeof(float);
    const bool fits = (a->Ndft == 64 || a->Ndft == 128 || a->Ndft == 256) && a->P >= 1 && a->Ts % a->P == 0 && lds_w + sizeof(FwCtl) + 64 <= 160 * 1024 && a->iperm;
    return (mode && !a->force_demod && fits) ? mode : 0;
}
// 1: sonde_launch_fsk will run the wave form, which keeps the Sf / tone-tail backups a repeat needs itself; 0: an older kernel, which does not (the host copies them)
extern "C" int sonde_fsk_wave_selected(const FskArgs *a) {
    int R, fin; size_t lds_w; bool demod_env;
    return (a->M == 2 || a->M == 4) && fsk_wave_plan(a, R, fin, lds_w, demod_env) != 0;
}

extern "C" int sonde_launch_fsk(const FskArgs *a, hipStream_t s) {
    const int M = a->M;
    if (a->M != 2 && a->M != 4) return -1;
    {
        int R, fin; size_t lds_w; bool demod_env;
        const int mode = fsk_wave_plan(a, R, fin, lds_w, demod_env);
        if (mode) {
            FskArgs b = *a; b.R = R; b.wave_mode = mode; b.fin = fin;
            static const char *rot_env = getenv("SONDE_FSK_ROT");
            b.role_rot = rot_env ? atoi(rot_env) : 1;
            if (mode == 2) return M == 2 ? launch_wave_n<2, true>(b, lds_w, s) : launch_wave_n<4, true>(b, lds_w, s);
            return M == 2 ? launch_wave_n<2, false>(b, lds_w, s) : launch_wave_n<4, false>(b, lds_w, s);
        }
        if (demod_env) { FskArgs b = *a; b.force_demod = 1; return sonde_launch_fsk_old(&b, s); }
    }
    return sonde_launch_fsk_old(a, s);
}

// floats of global scratch per workgroup k_fsk_demod<M, true> needs for this configuration; 0 where a frame fits into LDS (every kernel form)
extern "C" long long sonde_fsk_scratch_floats(const FskArgs *a) {
    const int W = (a->nsym + 1) * a->P, M = a->M;
    const int n_in = (a->N + a->Ts / 2) > W ? (a->N + a->Ts / 2) : W;
    const int nA = n_in > M * W + (a->nsym + 1) ? n_in : M * W + (a->nsym + 1);
    int nB = M * a->Nmem;
    if (4 * a->Ndft + (a->max_fft * a->Ndft + 1) / 2 > nB) nB = 4 * a->Ndft + (a->max_fft * a->Ndft + 1) / 2;
    if (W > nB) nB = W;
    const size_t base = (size_t)(nA + nB) * sizeof(float2) + (size_t)2 * a->Ndft * sizeof(float) + (size_t)a->Ndft * sizeof(float2);
    if (base <= 150 * 1024) return 0;
    return (long long)((base / sizeof(float) + 63) & ~(size_t)63);
}

extern "C" int sonde_launch_fsk_old(const FskArgs *a, hipStream_t s) {
    const int W = (a->nsym + 1) * a->P, M = a->M;
    if (a->M != 2 && a->M != 4) return -1;
    {   // the pipelined kernel, where its estimator fits (Ndft <= 256: every sonde configuration)
        static const char *st_env = getenv("SONDE_FSK_STREAM");          // A/B aid: 0 = the frame-at-a-time kernel
        // LDS of a channel: f_int, the Eb/N0 terms, the timing products, Sf + search copy, twiddles — fixed — plus the ring (R samples per tone) and the
        // scratch of the two estimator waves (bpw blocks of Ndft each).  A longer ring lets the oscillator run on while the consumer is busy with a frame's
        // soft decisions (R = 256: it stood still 17-40 % of the time); it is taken as long as the channel stays in its bracket of workgroups per CU
        // (160 KB / 5 or / 4).  (Giving up estimator blocks per wave for a longer ring was tried for RS41: the estimators then became what the oscillator waits for.)
        const size_t fixed = (size_t)M * W * sizeof(float2) + (size_t)((2 * a->nsym + 1) & ~1) * sizeof(float) + 64 * sizeof(float2)
                           + (size_t)2 * a->Ndft * sizeof(float) + (size_t)a->Ndft * sizeof(float2);
        int Rmin = 256; while (Rmin < a->NT + 128) Rmin <<= 1;
        const int bpw_max = (FSK_AE * 64) / a->Ndft > 0 ? (FSK_AE * 64) / a->Ndft : 1;
        auto total = [&](int R, int bpw) { return fixed + (size_t)M * R * sizeof(float2) + (size_t)2 * bpw * a->Ndft * sizeof(float2); };
        const size_t small = total(Rmin, bpw_max);
        const size_t bracket = small <= 160 * 1024 / 5 - 512 ? 160 * 1024 / 5 - 512 : small <= 160 * 1024 / 4 - 512 ? 160 * 1024 / 4 - 512 : small;
        int R = Rmin, bpw = bpw_max;
        static const char *r_env = getenv("SONDE_FSK_RING");                 // A/B aid: ring length
        if (r_env && atoi(r_env) >= Rmin) { R = atoi(r_env); while (R & (R - 1)) R++; }
        else {
            for (int cand = 2 * Rmin; cand > Rmin; cand >>= 1) if (total(cand, bpw_max) <= bracket) { R = cand; break; }
        }
        const size_t lds_s = total(R, bpw);
        if (!(st_env && atoi(st_env) == 0) && !a->force_demod && a->Ndft <= FSK_AE * 64 && a->Ndft >= FSK_AE && a->P >= 1 && a->Ts % a->P == 0 && lds_s <= 150 * 1024) {
            static size_t attr_s[FSK_MAX_DEV][2] = {};
            const void *fn = M == 2 ? reinterpret_cast<const void *>(k_fsk_stream<2>) : reinterpret_cast<const void *>(k_fsk_stream<4>);
            size_t &hw = attr_s[fsk_cur_dev()][M == 4];
            if (lds_s > hw) {
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s) != hipSuccess) return -2;
                hw = lds_s;
            }
            FskArgs b = *a; b.R = R; b.est_bpw = bpw;
            if (M == 2) hipLaunchKernelGGL(k_fsk_stream<2>, dim3(a->n_ch), dim3(FSK_THREADS), lds_s, s, b);
            else        hipLaunchKernelGGL(k_fsk_stream<4>, dim3(a->n_ch), dim3(FSK_THREADS), lds_s, s, b);
            return 0;
        }
    }
    const int n_in = (a->N + a->Ts / 2) > W ? (a->N + a->Ts / 2) : W;
    const int nA = n_in > M * W + (a->nsym + 1) ? n_in : M * W + (a->nsym + 1);
    int nB = M * a->Nmem;
    if (4 * a->Ndft + (a->max_fft * a->Ndft + 1) / 2 > nB) nB = 4 * a->Ndft + (a->max_fft * a->Ndft + 1) / 2;
    if (W > nB) nB = W;
    // regions A and B, Sf and its search copy, the twiddles — and the scratch of the waves that estimate the next frame ahead: as many of
    // the three as fit without costing the second workgroup of a CU its place (80 KB each, 256 B of static data)
    const size_t base = (size_t)(nA + nB) * sizeof(float2) + (size_t)2 * a->Ndft * sizeof(float) + (size_t)a->Ndft * sizeof(float2);
    const size_t cap = base + 256 <= 80 * 1024 ? (size_t)80 * 1024 - 256 : (size_t)150 * 1024;
    int ng = 3;
    static const char *ng_env = getenv("SONDE_FSK_AHEAD");               // A/B aid: 0 = every frame estimates for itself
    if (ng_env) { ng = atoi(ng_env); if (ng < 0) ng = 0; if (ng > 3) ng = 3; }
    const size_t per_wave = (size_t)FSK_AE * 64 * sizeof(float2);        // (FSK_AE * 64 / Ndft) blocks of Ndft elements
    while (ng > 0 && base + (size_t)ng * per_wave > cap) ng--;
    if (a->Ndft > FSK_AE * 64 || a->Ndft < FSK_AE) ng = 0;                // a lane holds FSK_AE elements of its block
    const size_t lds = base + (size_t)ng * per_wave;
    if (a->Ndft > 1024) return -1;
    if (lds > 150 * 1024) {                                              // a frame that no CU's LDS holds: the same kernel on a slice of global memory per workgroup
        if (!a->scratch || (size_t)a->scratch_stride * sizeof(float) < base) return -1;
        FskArgs b = *a; b.est_waves = 0; b.est_bpw = 0;
        if (a->M == 2) hipLaunchKernelGGL((k_fsk_demod<2, true>), dim3(a->n_ch), dim3(FSK_THREADS), 0, s, b);
        else           hipLaunchKernelGGL((k_fsk_demod<4, true>), dim3(a->n_ch), dim3(FSK_THREADS), 0, s, b);
        return 0;
    }
    FskArgs b = *a; b.est_waves = ng; b.est_bpw = 0; a = &b;
    if (a->M != 2 && a->M != 4) return -1;
    static size_t attr[FSK_MAX_DEV][2] = {};
    const void *fn = a->M == 2 ? reinterpret_cast<const void *>(k_fsk_demod<2>) : reinterpret_cast<const void *>(k_fsk_demod<4>);
    size_t &hw = attr[fsk_cur_dev()][a->M == 4];
    if (lds > hw) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        hw = lds;
    }
    if (a->M == 2) hipLaunchKernelGGL(k_fsk_demod<2>, dim3(a->n_ch), dim3(FSK_THREADS), lds, s, *a);
    else           hipLaunchKernelGGL(k_fsk_demod<4>, dim3(a->n_ch), dim3(FSK_THREADS), lds, s, *a);
    return 0;
}
