// sonde_scan.hip — gfx950 kernels of the scanner (the reference's scan/dft_detect.c).
//
//   k_iq_convert    --iq input: cs16 -> (x/32768 - mean) into the IF ring, exact integer IQ-DC sums   dft_detect.c:539-573
//   k_audio_convert FM-audio input: s16 -> b/32768 into FM stream 0                                   dft_detect.c:505-533
//   k_scan_if       3 IF low-passes on one delay line + 4 FM discriminators -> buf_fm[4]              dft_detect.c:737-819
//   k_scan_corr     one workgroup = one correlation window x one template:
//                   window -> FFT-8192 -> (dc, FM low-pass, matched filter) in the frequency domain -> FFT of the
//                   conjugate -> peak, norm, header bit check, M10 type bits                          dft_detect.c:357-443,866-977
//
// k_scan_corr keeps the reference's structure (circular correlation by two radix-2 DIT transforms) *and its
// twiddle factors*: the 8192-point data and the stage twiddles live in 128 KB of LDS, three butterfly stages per
// pass are held in registers.
#include "sonde_scan_dev.h"

#define WAVE 64

__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// ------------------------------------------------------------------------------------------------
// input converters
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_iq_convert(const IqConvArgs a) {
    const int ch = blockIdx.y;
    const uint32_t *iq = reinterpret_cast<const uint32_t *>(a.iq) + (size_t)ch * a.ch_stride;
    const float2 avg = a.dc_avg[ch];
    float2 *y = a.y + (size_t)ch * a.ring_len;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    long long sx = 0, sy = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        const uint32_t raw = iq[i];
        const int xi = (int)(short)(raw & 0xffffu), yi = ((int)raw) >> 16;
        sx += xi; sy += yi;
        // x = b/32768.0 exact; z = (x - avg) rounded once (dft_detect.c:554-560)
        y[(a.m0 + (uint32_t)i) & mask] = make_float2(__builtin_fmaf((float)xi, 3.0517578125e-05f, -avg.x),
                                                     __builtin_fmaf((float)yi, 3.0517578125e-05f, -avg.y));
    }
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch), (unsigned long long)sx);
        atomicAdd(reinterpret_cast<unsigned long long *>(a.dc_sums + 2 * (size_t)ch + 1), (unsigned long long)sy);
    }
}

__global__ __launch_bounds__(256)
void k_audio_convert(const AudioConvArgs a) {
    const int ch = blockIdx.y;
    const int16_t *pcm = a.pcm + (size_t)ch * a.ch_stride * a.nch;
    float *fm = a.fm + (size_t)ch * a.ring_len;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    if (a.f32) {
        const float *pf = reinterpret_cast<const float *>(a.pcm) + (size_t)ch * a.ch_stride * a.nch;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
            fm[(a.m0 + (uint32_t)i) & mask] = pf[(size_t)i * a.nch + a.sel];
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
        fm[(a.m0 + (uint32_t)i) & mask] = (float)pcm[(size_t)i * a.nch + a.sel] * 3.0517578125e-05f;     // b/128.0/256.0, exact
}

// ------------------------------------------------------------------------------------------------
// k_scan_if: 1024 IF samples of one channel per workgroup, four consecutive outputs per thread
// ------------------------------------------------------------------------------------------------
// Round 4: a thread keeps a sliding window of four y samples in registers and advances it by ONE LDS read per tap (round 3: one read per tap and
// OUTPUT, and three tap reads beside it — the kernel was bound by LDS instruction issue at 4 LDS operations per 3 packed multiply-adds; now 1.75
// per 12).  Per output and filter the multiply-adds are the same fmaf chain in the same tap order: streams unchanged to the bit.
#define SI_TILE 1024
#define SI_THREADS 256
#define SI_PER 4
typedef float si_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(SI_THREADS)
void k_scan_if(const ScanIfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem2[];
    const int ch = blockIdx.y, T = a.taps, tid = threadIdx.x;
    const uint32_t t0 = a.m0 + (uint32_t)blockIdx.x * SI_TILE;
    const int nout = min(SI_TILE, (int)(a.m0 + (uint32_t)a.n - t0));
    if (nout <= 0) return;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const int T4 = (T + 3) & ~3;
    float2 *sy = smem2;                          // y[t0 - T + k], k < SI_TILE + T + 4   (one sample more than the FIR history: z[t0-1])
    float2 *sz = sy + SI_TILE + T4 + 4;          // [3][SI_TILE + 4] z_b[t0 - 1 + k]   (an even number of float2 in front: the tap rows behind stay 16-byte aligned)
    float  *sw = reinterpret_cast<float *>(sz + 3 * (SI_TILE + 4));   // [nfilt][T4], zero padded: 16-byte rows
    const float2 *yr = a.y + (size_t)ch * a.ring_len;
    const ScanFold &fo = a.fold;
    for (int k = tid; k < SI_TILE + T + 4; k += SI_THREADS) {
        const int64_t m = (int64_t)t0 - T + k;
        float2 v = make_float2(0.f, 0.f);
        if (m >= 0 && k < nout + T) {
            if (!fo.etab) v = yr[(uint32_t)m & mask];
            else {
                // y holds the raw sums of k_mix_decimate50r: the IQ-DC mean of the block's window comes off here, y -= mean * E (+ what the first Q-1 outputs of a
                // window need on top, k_scan_dc_edges).  Samples of earlier launches: the folded history the launch before left (their means are gone)
                const int32_t j = (int32_t)((uint32_t)m - fo.m0);
                if (j >= 0) {
                    v = yr[(uint32_t)m & mask];
                    const uint32_t jo = (uint32_t)j + (uint32_t)fo.dc_seg_off, B = (uint32_t)fo.dc_seg_blocks;
                    uint32_t kw = jo / B; const uint32_t rem = jo - kw * B;
                    kw = kw < (uint32_t)fo.dc_seg_n - 1u ? kw : (uint32_t)fo.dc_seg_n - 1u;
                    const uint32_t e = (fo.e0 + (uint32_t)j) % (uint32_t)fo.etab_len;
                    const float2 mean = fo.dc_seg[(size_t)ch * fo.dc_seg_n + kw], E = fo.etab[(size_t)ch * fo.etab_len + e];
                    v.x = fmaf(-mean.x, E.x, v.x); v.y = fmaf(-mean.x, E.y, v.y);
                    v.x = fmaf(mean.y, E.y, v.x);  v.y = fmaf(-mean.y, E.x, v.y);
                    if (rem < (uint32_t)fo.edge_n) { const float2 c = fo.corr[((size_t)ch * fo.dc_seg_n + kw) * 8 + rem]; v.x += c.x; v.y += c.y; }
                } else if (-j <= fo.hist_n) v = fo.hist_in[(size_t)ch * fo.hist_n + (fo.hist_n + j)];
            }
        }
        sy[k] = v;
    }
    for (int k = tid; k < a.nfilt * T4; k += SI_THREADS) { const int bq = k / T4, kk = k % T4; sw[k] = kk < T ? a.w[bq * T + kk] : 0.f; }
    __syncthreads();
    if (fo.etab) {
        // the folded outputs the next launch's first tile needs as its history: the last hist_n of this launch (a launch shorter than that: the older ones move up)
        const uint32_t m_end = fo.m0 + (uint32_t)fo.nblocks;
        for (int i = tid; i < nout; i += SI_THREADS) {
            const uint32_t back = m_end - (t0 + (uint32_t)i);                  // 1 = the launch's last output
            if (back <= (uint32_t)fo.hist_n) fo.hist_out[(size_t)ch * fo.hist_n + (fo.hist_n - (int)back)] = sy[T + i];
        }
        if (blockIdx.x == 0) for (int i = tid; i < fo.hist_n - fo.nblocks; i += SI_THREADS) fo.hist_out[(size_t)ch * fo.hist_n + i] = fo.hist_in[(size_t)ch * fo.hist_n + i + fo.nblocks];
    }
    // z_b[m] = sum_k w_b[k] * y[m-(T-1)+k] for m = t0-1 .. t0+nout-1 (oldest sample pairs with tap 0, dft_detect.c:696-705): outputs o = 4 tid .. 4 tid + 3
    // of the nout + 1; the last thread's window runs into the zero padding behind the tile
    for (int o0 = SI_PER * tid; o0 < nout + 1; o0 += SI_PER * SI_THREADS) {
        si_f2 acc[3][SI_PER];
#pragma unroll
        for (int bq = 0; bq < 3; bq++)
#pragma unroll
            for (int u = 0; u < SI_PER; u++) acc[bq][u] = (si_f2){0.f, 0.f};
        const float2 *yy = sy + o0;              // y[t0 - 1 + o0 - (T-1)] = sy[o0]
        si_f2 win[SI_PER];
#pragma unroll
        for (int u = 0; u < SI_PER - 1; u++) { const float2 v = yy[u]; win[u + 1] = (si_f2){v.x, v.y}; }
        for (int k0 = 0; k0 < T; k0 += 4) {
            float4 w4[3];
#pragma unroll
            for (int bq = 0; bq < 3; bq++) w4[bq] = bq < a.nfilt ? *reinterpret_cast<const float4 *>(sw + bq * T4 + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int k = k0 + kk;
                if (k < T) {
#pragma unroll
                    for (int u = 0; u < SI_PER - 1; u++) win[u] = win[u + 1];
                    const float2 v = yy[k + SI_PER - 1];
                    win[SI_PER - 1] = (si_f2){v.x, v.y};
#pragma unroll
                    for (int bq = 0; bq < 3; bq++) if (bq < a.nfilt) {
                        const float w = kk == 0 ? w4[bq].x : kk == 1 ? w4[bq].y : kk == 2 ? w4[bq].z : w4[bq].w;
                        const si_f2 wq = {w, w};
#pragma unroll
                        for (int u = 0; u < SI_PER; u++) acc[bq][u] = __builtin_elementwise_fma(wq, win[u], acc[bq][u]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SI_PER; u++) {
            const int o = o0 + u;
            if (o < nout + 1) {
                const bool before = (int64_t)t0 - 1 + o < 0;  // z[-1] = 0: static z0 of f32buf_sample starts at 0
#pragma unroll
                for (int bq = 0; bq < 3; bq++) if (bq < a.nfilt) sz[bq * (SI_TILE + 4) + o] = before ? make_float2(0.f, 0.f) : make_float2(acc[bq][u].x, acc[bq][u].y);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nout; i += SI_THREADS) {
        const uint32_t m = (t0 + (uint32_t)i) & mask;
        const size_t cs = (size_t)a.n_ch * a.ring_len, co = (size_t)ch * a.ring_len + m;
        // s = 0.8 * carg(z * conj(z_prev)) / pi   (dft_detect.c:776-803)
#pragma unroll
        for (int bq = 0; bq < 3; bq++) if (bq < a.nfilt) {
            const float2 z1 = sz[bq * (SI_TILE + 4) + i + 1], z0 = sz[bq * (SI_TILE + 4) + i];
            const float2 w = cmulc(z1, z0);
            a.fm[(size_t)a.filt_stream[bq] * cs + co] = (float)(0.8 * (double)atan2f(w.y, w.x) / 3.14159265358979323846);
        }
        const float2 y1 = sy[T + i], y0 = ((int64_t)t0 + i - 1 < 0) ? make_float2(0.f, 0.f) : sy[T + i - 1];
        const float2 w = cmulc(y1, y0);
        a.fm[(size_t)a.raw_stream * cs + co] = (float)(0.8 * (double)atan2f(w.y, w.x) / 3.14159265358979323846);
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_corr
// ------------------------------------------------------------------------------------------------
#include "sonde_fft_dev.h"

__device__ __forceinline__ float wsumf(float v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }
__device__ __forceinline__ double wsumd(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }

// one workgroup = one correlation window x one template (getCorrDFT, dft_detect.c:357-443)
// BIG = false: N_DFT = 8192, the data array and the first stages' twiddles in LDS (sonde_fft_dev.h).  BIG = true: N_DFT = 16384 / 32768 (IF rates above
// ~51 kHz, dft_detect.c:1196-1202): the same network stage by stage on a per-workgroup array in global memory — rare (wide --bw, wide --iq input),
// exactness over speed.
// dft_big: the reference's dft_raw on a bit-reversed array in global memory, one radix-2 stage per pass; twiddle of butterfly column j of stage s = tws[2^s - 1 + j]
__device__ __forceinline__ void dft_big(float2 *x, const float2 *tws, int log2n, int tid) {
    const int half = 1 << (log2n - 1);
    for (int s = 0; s < log2n; s++) {
        const int l2 = 1 << s;
        for (int bfly = tid; bfly < half; bfly += SC_THREADS) {
            const int j = bfly & (l2 - 1), i = ((bfly >> s) << (s + 1)) | j, k = i + l2;
            const float2 t = cmul(x[k], tws[(l2 - 1) + j]), u = x[i];
            x[k] = make_float2(u.x - t.x, u.y - t.y);
            x[i] = make_float2(u.x + t.x, u.y + t.y);
        }
        __syncthreads();
    }
}

template <bool BIG>
__global__ __launch_bounds__(SC_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))      // two workgroups per CU: 78 KB of LDS, 64 VGPRs
void k_scan_corr_t(const ScanCorrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem2[];
    float2 *x = BIG ? a.scratch + (size_t)blockIdx.x * a.N : smem2;      // [SC_XN] padded in LDS, or the [N] global array of this workgroup
    float2 *tws = smem2 + SC_XN;       // [SC_TW_LDS + 1] twiddles of stages 0..8 (LDS form only)
    const int log2n = BIG ? a.log2n : SC_LOG2N;
    auto XP = [&](int i) -> int { return BIG ? i : XI(i); };
    auto BR = [&](int k) -> int { return (int)(__brev((unsigned)k) >> (32 - log2n)); };
    auto DFT = [&](int tid_) { if (BIG) dft_big(x, a.tws, log2n, tid_); else dft_ref(x, tws, a.tws, tid_); };
    __shared__ float s_rf[SC_THREADS / WAVE];
    __shared__ int s_ri[SC_THREADS / WAVE];
    __shared__ double s_rd[SC_THREADS / WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int item = blockIdx.x, j = blockIdx.y;
    if (a.work) { const ScanWork w = a.work[blockIdx.x]; item = w.item; j = w.tpl; }      // listed pairs only (behind the prefilter)
    const ScanItem it = a.items[item];
    const int K = a.K, N = BIG ? a.N : SC_N;
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const ScanTpl tp = a.tpl[j];
    ScanRes *out = a.out + (size_t)item * SC_NTPL + j;
    if (!tp.active) {
        if (tid == 0) *out = ScanRes{ 0, 0.f, 0u, 0.f, -1, 0u };
        return;
    }
    const float *str = a.fm + ((size_t)tp.stream * a.n_ch + it.ch) * a.ring_len;
    const int L = tp.L, wl = K + L;
    const int64_t start = (int64_t)it.pos - (wl - 1);
    if (!BIG) for (int k = tid; k < SC_TW_LDS; k += SC_THREADS) tws[k] = a.tws[k];
    float dc = 0.f;
    // xn[i] = stream[pos - (K+L-1) + i], i < K+L, zero padded (dft_detect.c:378-379); stored bit-reversed for the DIT network
    auto load_window = [&](bool want_dc) {
        float dcp = 0.f;
        for (int i = tid; i < N; i += SC_THREADS) {
            const int64_t p = start + i;
            const float v = (i < wl && p >= 0) ? str[(uint32_t)p & mask] : 0.f;
            if (i >= K - L && i < wl) dcp += v;                        // last 2L samples (dft_detect.c:389)
            x[XP(BR(i))] = make_float2(v, 0.f);
        }
        if (want_dc) {
            const float sw = wsumf(dcp); if (lane == 0) s_rf[wave] = sw;
            __syncthreads();
            float sum = 0.f;
            for (int w = 0; w < SC_THREADS / WAVE; w++) sum += s_rf[w];
            dc = (float)((double)sum / (2.0 * (double)(float)L));
        }
        __syncthreads();
    };
    // Z = X * H with X[0] -= N*dc*0.98 for --dc (dft_detect.c:387-403); Nidft() transforms conj(Z): conjugate and swap
    // into bit-reversed order for the next pass of the same network.  H == nullptr: Z = X.
    auto spectrum_step = [&](const float2 *H) {
        const float dcsub = a.opt_dc ? (float)((double)((float)N * dc) * 0.98) : 0.f;
        for (int i = tid; i < N; i += SC_THREADS) {
            const int r = BR(i);
            if (r < i) continue;
            float2 xi = x[XP(i)], xr = x[XP(r)];
            if (i == 0) { xi.x -= dcsub; xr.x -= dcsub; }             // i = r = 0
            const float2 zi = H ? cmul(xi, H[i]) : xi, zr = H ? cmul(xr, H[r]) : xr;
            x[XP(r)] = make_float2(zi.x, -zi.y);
            x[XP(i)] = make_float2(zr.x, -zr.y);
        }
        __syncthreads();
    };
    // The reference filters the window first (X * WS[lpFM] -> Nidft -> xn, dft_detect.c:394-402: it only feeds the norm) and correlates then.  The
    // two computations are independent, so the correlation runs first here and the filtered window is left in `x` when the norm is taken: no
    // second copy of the window in LDS, which is what lets two workgroups share a CU.
    const bool filt = a.opt_dc || a.opt_iq;
    load_window(a.opt_dc != 0);
    DFT(tid);                                         // X = dft(xn)
    spectrum_step(a.G + (size_t)j * N);                                  // G = WS * Fm (Fm alone for FM-audio input)
    DFT(tid);                                         // cx = Nidft(Z), real part used

    // arg-max of cx^2 over i in [L-1, K+L), first maximum wins (dft_detect.c:415-423)
    float best = 0.f; int bidx = -1;
    for (int i = tid; i < N; i += SC_THREADS) {
        if (i >= L - 1 && i < wl) { const float c = x[XP(i)].x, c2 = c * c; if (c2 > best) { best = c2; bidx = i; } }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bidx, off);
        if (ob > best || (ob == best && oi >= 0 && (bidx < 0 || oi < bidx))) { best = ob; bidx = oi; }
    }
    if (lane == 0) { s_rf[wave] = best; s_ri[wave] = bidx; }
    __syncthreads();
    int mp = -1;
    {
        float b = 0.f;
        for (int w = 0; w < SC_THREADS / WAVE; w++) {
            const float ob = s_rf[w]; const int oi = s_ri[w];
            if (ob > b || (ob == b && oi >= 0 && (mp < 0 || oi < mp))) { b = ob; mp = oi; }
        }
    }
    const float mx = (mp >= 0) ? x[XP(mp)].x : 0.f;
    __syncthreads();                                                     // everybody has mx before x is reused
    // norm over the L (filtered) window samples under the peak (dft_detect.c:431-433)
    double e2 = 0.0;
    if (filt) {
        load_window(false);
        DFT(tid);
        spectrum_step(a.opt_iq ? a.WS + (size_t)tp.lpfm * N : nullptr);
        DFT(tid);                                     // x[i].x / N = filtered xn[i]
        if (mp >= 0) for (int k = tid; k < L; k += SC_THREADS) { const float v = x[XP(mp - k)].x / (float)N; e2 += (double)(v * v); }
    } else if (mp >= 0) {
        for (int k = tid; k < L; k += SC_THREADS) {                      // raw window, read again from the stream
            const int i = mp - k; const int64_t p = start + i;
            const float v = (i < wl && p >= 0) ? str[(uint32_t)p & mask] : 0.f;
            e2 += (double)(v * v);
        }
    }
    { const double sw = wsumd(e2); if (lane == 0) s_rd[wave] = sw; }
    __syncthreads();

    if (wave == 0) {
        ScanRes r{ 0, 0.f, 0u, dc, -1, 0u };
        if (mp < 0) {
            // a window without a single non-zero correlation value (a stream that begins with digital silence): the reference's loop leaves mp = -1, which is
            // no edge value, so getCorrDFT runs on and hands back mpos = pos - (K + L - 1) - 1 (- the low-pass delay), wrapped — and `mv_pos > mv0_pos`
            // (dft_detect.c:1521) then fails for whatever the NEXT window finds.  Mirrored: the score there is 0 / (a norm read in front of the array)
            r.mp = -1;
            r.mpos = it.pos - (uint32_t)(K + L - 1) - 1u;
            if (a.opt_iq) r.mpos -= (uint32_t)(a.lpfm_taps / 2);
        }
        else if (mp == L - 1 || mp == K + L - 1) r.mp = -4;                          // edge value
        else {
            double e = 0.0;
            for (int w = 0; w < SC_THREADS / WAVE; w++) e += s_rd[w];
            const double xnorm = sqrt(e);
            r.mp = mp;
            r.mv = (float)((double)mx / (xnorm * (double)N));
            r.mpos = it.pos - (uint32_t)(K + L - 1) + (uint32_t)mp;
            if (a.opt_iq) r.mpos -= (uint32_t)(a.lpfm_taps / 2);                       // low-pass delay
            if (r.mv > tp.thres || r.mv < -tp.thres) {
                const float dcv = a.opt_dc ? dc : 0.f;
                const int sign = r.mv < 0.f ? 1 : 0;
                // headcmp (dft_detect.c:866-905): hard bits of the header from the raw stream
                const uint32_t mvp = r.mpos + 1u - (uint32_t)(int)((float)tp.hLen * tp.spb);
                const int *bnd = a.bnd + tp.bnd_off;
                int errs = 0;
                for (int b0 = 0; b0 < tp.hLen; b0 += WAVE) {
                    const int b = b0 + lane;
                    int bad = 0;
                    if (b < tp.hLen) {
                        const int q0 = b ? bnd[b - 1] : 0, q1 = bnd[b];
                        double sum = 0.0;
                        for (int q = q0; q < q1; q++) {
                            const int64_t p = (int64_t)(int32_t)(mvp + (uint32_t)q);
                            const float v = (p >= 0) ? str[(uint32_t)p & mask] : 0.f;
                            sum += (double)(v - dcv);
                        }
                        const int bit = sum >= 0.0 ? 1 : 0;
                        bad = ((bit ^ sign) != (a.hdrbits[tp.hdr_off + b] & 1));
                    }
                    errs += __popcll(__ballot(bad));
                }
                r.herrs = errs;
                if (tp.is_m10) {
                    // frm_M10 (dft_detect.c:932-977): 14 Manchester symbols behind the header, first half minus second
                    const int *bm = bnd + tp.hLen;
                    int one = 0;
                    if (lane < 14) {
                        const int q0 = lane ? bm[2 * lane - 1] : 0, q1 = bm[2 * lane], q2 = bm[2 * lane + 1];
                        double sum = 0.0;
                        for (int q = q0; q < q1; q++) sum += (double)(str[(r.mpos + (uint32_t)q) & mask] - dcv);
                        for (int q = q1; q < q2; q++) sum -= (double)(str[(r.mpos + (uint32_t)q) & mask] - dcv);
                        one = sum >= 0.0;
                    }
                    r.m10 = (uint32_t)(__ballot(one) & 0x3fffULL);
                }
            }
        }
        if (lane == 0) *out = r;
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" void sonde_launch_scan_if(const ScanIfArgs *a, hipStream_t s) {
    const size_t lds = (size_t)(SI_TILE + ((a->taps + 3) & ~3) + 4 + 3 * (SI_TILE + 4)) * sizeof(float2) + (size_t)a->nfilt * ((a->taps + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(k_scan_if, dim3((a->n + SI_TILE - 1) / SI_TILE, a->n_ch), dim3(SI_THREADS), lds, s, *a);
}
extern "C" int sonde_launch_scan_corr(const ScanCorrArgs *a, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)(SC_XN + SC_TW_LDS + 1) * sizeof(float2);
    if (a->N > SC_N) {                                     // the global-memory form: the caller lists the pairs and owns the scratch arrays
        if (!a->work || !a->scratch || (1 << a->log2n) != a->N) return -1;
        if (a->n_work > 0) hipLaunchKernelGGL(k_scan_corr_t<true>, dim3(a->n_work, 1), dim3(SC_THREADS), 0, s, *a);
        return 0;
    }
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_corr_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_set = true;
    }
    if (a->work) { if (a->n_work > 0) hipLaunchKernelGGL(k_scan_corr_t<false>, dim3(a->n_work, 1), dim3(SC_THREADS), lds, s, *a); return 0; }      // listed pairs only (behind the prefilter)
    if (a->n_items <= 0) return 0;
    hipLaunchKernelGGL(k_scan_corr_t<false>, dim3(a->n_items, SC_NTPL), dim3(SC_THREADS), lds, s, *a);
    return 0;
}
// cs16 -> cf32 (x / 32768, exact) for front ends that only exist in float32 form (decimators longer than 8 blocks: wide IF)
__global__ __launch_bounds__(256)
void k_s16_to_f32(const int16_t *in, long long in_stride, float2 *out, long long out_stride, int n) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(in) + (size_t)blockIdx.y * in_stride;
    float2 *q = out + (size_t)blockIdx.y * out_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t w = p[i];
        q[i] = make_float2((float)(int16_t)(w & 0xffffu) / 32768.0f, (float)(int16_t)(w >> 16) / 32768.0f);
    }
}
extern "C" void sonde_launch_s16_to_f32(const int16_t *in, long long in_stride, float2 *out, long long out_stride, int n_ch, int n, hipStream_t s) {
    int gx = (n + 255) / 256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_s16_to_f32, dim3(gx, n_ch), dim3(256), 0, s, in, in_stride, out, out_stride, n);
}
extern "C" void sonde_launch_iq_convert(const IqConvArgs *a, hipStream_t s) {
    int gx = (a->n + 255) / 256; if (gx > 64) gx = 64; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_iq_convert, dim3(gx, a->n_ch), dim3(256), 0, s, *a);
}
extern "C" void sonde_launch_audio_convert(const AudioConvArgs *a, hipStream_t s) {
    int gx = (a->n + 255) / 256; if (gx > 64) gx = 64; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_audio_convert, dim3(gx, a->n_ch), dim3(256), 0, s, *a);
}
