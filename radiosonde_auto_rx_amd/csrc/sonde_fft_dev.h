// sonde_fft_dev.h — the reference's own 8192-point transform on the device (dft_raw: scan/dft_detect.c:285-322, demod/mod/demod_mod.c:26-63):
// radix-2 decimation in time whose stage twiddles advance by a float recurrence.  Shared by the scanner (k_scan_corr) and the
// demodulators' header search (k_sync_window_fft); included once per .hip file, under that file's floating-point contraction mode.
#ifndef SONDE_FFT_DEV_H
#define SONDE_FFT_DEV_H
#include "sonde_scan_dev.h"
#ifndef FFT_THREADS
#define FFT_THREADS SC_THREADS      // threads of the workgroup that runs a transform (the including file may choose)
#endif

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// LDS index of element i: position i + (i >> 4) + (i >> 8).  With it every access pattern of the transform spreads evenly over the banks
// (tools/probes/lds_banks.py: each aligned group of 32 lanes touches every 8-byte bank slot exactly twice): the element strides 1, 8, 64, 512
// and 4096 of the register passes AND the bit-reversed order — lanes i..i+63 go to brev13(i) = base + 128 * brev6(lane) — in which the
// input is stored and the spectra are conjugated and swapped.  (One pad per 8 elements served the passes but put all 64 lanes of a
// bit-reversed access into one slot.)
#define XI(i) ((i) + ((i) >> 4) + ((i) >> 8))
#define SC_XN (SC_N + SC_N / 16 + SC_N / 256)     // elements of the padded array

__device__ __forceinline__ int brev13(int k) { return (int)(__brev((unsigned)k) >> (32 - SC_LOG2N)); }

// R merged radix-2 decimation-in-time stages starting at stage t0 (bit-reversed in -> natural out), groups of 2^R
// elements held in registers.  The twiddles are the reference's own: stage t uses w_t[j] = w_t[j-1] * cexp(-i pi/2^t)
// accumulated in float (dft_raw, dft_detect.c:306-319), tabulated by the host at tws[2^t - 1 + j].  The recurrence
// drifts by up to ~2e-4 in the last stages, and the reference's scores carry that drift; using its table reproduces
// them instead of the exact DFT.
template <int R>
__device__ __forceinline__ void dit_pass(float2 *x, const float2 *tws, const int t0, const int tid) {
    constexpr int E = 1 << R;
    const int p_lo = t0;
#pragma unroll 1
    for (int g = tid; g < (SC_N >> R); g += FFT_THREADS) {
        const int low = g & ((1 << p_lo) - 1), high = g >> p_lo;
        const int base = (high << (p_lo + R)) | low;
        float2 v[E];
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = x[XI(base + (e << p_lo))];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int t = t0 + s, bit = 1 << s;
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (e & bit) continue;
                const int idx = base + (e << p_lo);
                const float2 w = tws[((1 << t) - 1) + (idx & ((1 << t) - 1))];
                const float2 p = v[e], r = cmul(v[e | bit], w);
                v[e] = make_float2(p.x + r.x, p.y + r.y);
                v[e | bit] = make_float2(p.x - r.x, p.y - r.y);
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) x[XI(base + (e << p_lo))] = v[e];
    }
    __syncthreads();
}

// input already bit-reversed; stages 0..8 take their twiddles from an LDS copy (SC_TW_LDS = 2^9 - 1 entries), the last four from the global table
// (L2-resident, 32 KB, shared by every workgroup): the data array and 4 KB of twiddles are all the LDS a transform needs, so two workgroups of
// 1024 threads fit on a CU
#define SC_TW_LDS 511
// stages 0..11; the caller runs the last stage (12) itself when it wants the outputs in registers
__device__ __forceinline__ void dft_ref_head(float2 *x, const float2 *tws, const float2 *tws_g, const int tid) {
    dit_pass<3>(x, tws, 0, tid);
    dit_pass<3>(x, tws, 3, tid);
    dit_pass<3>(x, tws, 6, tid);
    dit_pass<3>(x, tws_g, 9, tid);
}
__device__ __forceinline__ void dft_ref(float2 *x, const float2 *tws, const float2 *tws_g, const int tid) {
    dft_ref_head(x, tws, tws_g, tid);
    dit_pass<1>(x, tws_g, 12, tid);
}

#endif
