// sonde_scan_pre.hip — k_scan_pre: the scanner's prefilter on the matrix cores (gfx950, v_mfma_f32_16x16x32_f16).
//
// getCorrDFT (scan/dft_detect.c:357-443) scores a header template against a window as
//     mv = c[mp] / sqrt(e[mp]),  c[p] = sum_k match[k] xf[p-L+1+k],  e[p] = sum_{i<L} xf[p-i]^2,  mp = argmax_p c[p]^2,
// where xf is the window (K+L samples ending at sample_out, zeros before and behind it) after `X[0] -= N dc 0.98` (a constant 0.98 dc taken off
// every sample) and the FM low-pass — all evaluated through its own drifting radix-2 transform, which is what k_scan_corr mirrors at 3-4
// transforms per (window, template).  Almost every pair is far below its threshold (noise: |mv| < 0.35 against 0.6 ... 0.8), so this kernel
// evaluates the SAME quantities in the time domain with f16 operands and f32 accumulation — within ~1e-4 of the reference's values — and
// reports an upper bound smax = max_p |c[p]| / sqrt(e[p]) >= |mv|; only pairs with smax > thres - margin (margin 0.03 = 300 times the rounding)
// are handed to the exact kernel, together with the same template's window before them (its exact peak position feeds the reference's
// `mv_pos > mv0_pos` test).  Decisions, printed scores and positions therefore still come from the reference's own transform network.
//
// Both the low-pass and the correlation are Toeplitz products out[i] = sum_u h[u] x[i+u]; with i = 16 a + b, u = 32 c + e - b:
//     out[16 a + b] = sum_c sum_{e<32} A_c[b][e] x[16 a + 32 c + e],   A_c[b][e] = h[32 c + e - b]
// i.e. per step c (32 taps, no zero half in the fragment) one 16x16x32 MFMA: A_c is a constant fragment (tabulated by the host, 1 KB per step), the B
// fragment of lane (n = lane & 15, g = lane >> 4) is the 8 consecutive halves x[16 (a0+n) + 32 c + 8 g ...] — one 16-byte LDS read.  The D fragment of a lane is out[16 (a0+n) + 4 g + r],
// r < 4.  (A and B use the same k order within a lane, so the products pair up whatever the hardware's internal k numbering is.)
// One workgroup (8 waves) = one (window, template); LDS: window f16 (17 KB) + filtered window f16 (19 KB) + every fourth prefix sum of its squares (8 KB)
// + the A fragments of the low-pass and of the template (1 KB per 32 taps, at most 32 KB at a time) <= 77 KB, two per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sonde_scan_dev.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SP_THREADS 512
#define SP_WAVES 8
#define SP_MAXT 4                      // 256-sample tiles per wave (8192 samples / 256 / 8 waves)
#define SP_CK 2                        // Toeplitz steps per block of prefetched A fragments (the host pads every table to a multiple)
#define SP_CH 16                       // samples per thread in the load and prefix phases (8192 / 512)
#ifndef SP_ACAP
#define SP_ACAP 8                      // A-fragment steps (1 KB each) the LDS holds at a time: SP_AQ x 16 bytes per thread; longer templates reload
#endif
#define SP_AQ (SP_ACAP * 64 / SP_THREADS)
#define SP_CP 1.9073486e-6f            // 32 x 2^-24: the float32 rounding of a prefix sum that is built in three short levels (16 per thread, the wave's scan, the waves), DESIGN.md §4.6b

// S(i) = sum_{q<i} xf[q]^2 from the sums kept for every fourth i and the squares between
__device__ __forceinline__ float sp_S(const float *P, const _Float16 *xf, int i) {
    float s = P[i >> 2];
    for (int q = i & ~3; q < i; q++) { const float x = (float)xf[q]; s += x * x; }
    return s;
}

// acc[t] = sum_c A_c x B(tile t, step c) for the NT tiles of this wave (tile = wave + SP_WAVES t); x: f16 array in LDS whose element 0 pairs with
// h[0] of output 0.  NT is a template parameter so that the step loop has no branches: the B fragments of a step are NT independent 16-byte
// LDS reads at a fixed distance, followed by NT MFMAs.  The A fragments of ALL steps are in LDS (afrag, 1 KB per step, staged once per workgroup
// while the window is on its way): round 3 streamed them from global memory inside this loop, one block of steps ahead — every block then waited
// a global-memory round trip of ~1 us for ~0.1 us of MFMAs, and all eight waves fetched the same bytes.
template <int NT>
__device__ __forceinline__ void sp_toeplitz_nt(const _Float16 *x, const _Float16 *afrag, int nc, int wave, int lane, f32x4 *acc) {
    const int n = lane & 15, g = lane >> 4;
    f32x4 r[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) r[t] = acc[t];                       // continues what earlier chunks of steps added up (the caller zeroes acc)
    const half8 *af = reinterpret_cast<const half8 *>(afrag) + lane;
    const _Float16 *xb = x + 16 * (16 * wave + n) + 8 * g;           // tile t adds 16 * 16 * SP_WAVES * t halves, step c adds 32
    for (int c0 = 0; c0 < nc; c0 += SP_CK) {                          // nc is a multiple of SP_CK (the host pads the table with zero steps)
#pragma unroll
        for (int k = 0; k < SP_CK; k++) {
            const half8 A = af[(size_t)64 * (c0 + k)];
            half8 B[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) B[t] = *reinterpret_cast<const half8 *>(xb + 256 * SP_WAVES * t + 32 * k);
#pragma unroll
            for (int t = 0; t < NT; t++) r[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B[t], r[t], 0, 0, 0);
        }
        xb += 32 * SP_CK;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = r[t];
}
// number of tiles of this wave (wave-uniform) and the dispatch on it
__device__ __forceinline__ int sp_toeplitz(const _Float16 *x, const _Float16 *afrag, int nc, int wave, int ntiles, int lane, f32x4 (&acc)[SP_MAXT]) {
    const int cnt = ntiles > wave ? (ntiles - wave + SP_WAVES - 1) / SP_WAVES : 0;
    switch (cnt) {
        case 1: sp_toeplitz_nt<1>(x, afrag, nc, wave, lane, acc); break;
        case 2: sp_toeplitz_nt<2>(x, afrag, nc, wave, lane, acc); break;
        case 3: sp_toeplitz_nt<3>(x, afrag, nc, wave, lane, acc); break;
        case 4: sp_toeplitz_nt<4>(x, afrag, nc, wave, lane, acc); break;
        default: break;
    }
    return cnt;
}

// Wave-wide maximum of a 32-bit unsigned value (non-negative floats order like their bit patterns), every lane's copy in an SGPR: the row-shift /
// row-broadcast DPP scan of gfx9 (shifts by 1, 2, 4, 8 inside a row of 16, then the last lane of rows 0 / 2 into rows 1 / 3 and of row 1 into rows 2-3) —
// six vector instructions and no LDS, where six __shfl_xor steps are six ds_bpermute round trips.  Lanes without a source keep 0, the identity.
__device__ __forceinline__ uint32_t sp_wave_umax(uint32_t v) {
    int x = (int)v;
#define SP_DPP_MAX(ctrl, rmask) { const int y = __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false); x = (int)max((uint32_t)x, (uint32_t)y); }
    SP_DPP_MAX(0x111, 0xf) SP_DPP_MAX(0x112, 0xf) SP_DPP_MAX(0x114, 0xf) SP_DPP_MAX(0x118, 0xf)     // row_shr:1, 2, 4, 8
    SP_DPP_MAX(0x142, 0xa) SP_DPP_MAX(0x143, 0xc)                                                      // row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3)
#undef SP_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ float sp_hwmax(float a, float b) {      // v_max_f32 as the hardware does it: a NaN operand loses (no canonicalisation in front)
    float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ float sp_hwmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the same scan for floats: inclusive prefix of OP over the wave's lanes (lane 63: the whole wave); IDENT (as bits) is what a lane without a source combines with
#define SP_DPP_SCAN(x, OP, IDENT) do { \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x111, 0xf, 0xf, false))); \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x112, 0xf, 0xf, false))); \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x114, 0xf, 0xf, false))); \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x118, 0xf, 0xf, false))); \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x142, 0xa, 0xf, false))); \
    x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(IDENT, __float_as_int(x), 0x143, 0xc, 0xf, false))); } while (0)
__device__ __forceinline__ float sp_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float sp_last(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)); }

// Scores of one correlation tile: e[p] = S(p + L) - S(p) from the kept sums and the squares between, |c[p]| / sqrt(e[p]) and the arg-max of |c|.
// HO = (p0 + L) & 3 = L & 3 for every lane (p0 is a multiple of 4), so which of the eight squares read at p0 + L joins the upper sum at step r is known
// at compile time.  No branches: a position outside the arg-max range (EDGE tiles only) gets |c| = -1, which neither the score (negative) nor the
// arg-max (never above what is held) takes; e < 0 (rounding of the difference where the window is empty) makes the score a NaN that v_max_f32 drops,
// e = 0 with c != 0 makes it +inf: the pair goes to the exact kernel.  Within a lane p only grows (r, then the tiles in order), so a later equal |c| never
// replaces an earlier one — first maximum wins — and what is kept is bidx = 4 t + r: position and c are looked up once, behind the loop.
// Round 6: what is maximised is not |c| / sqrt(e) but that value plus the bound of its own rounding error at the position:
//     (|c[p]| + kx) / sqrt(e[p]) + be / e[p],   kx = kappa_j X (f16 roundings of window, taps and filtered window; X = the window's maximum), be = C_P E_win (prefix sums)
// so that "the maximum is below thres - margin" rules a pair out whatever the signal looks like — a quiet span beside a loud one gets a wide bound and goes to the
// exact kernel instead of being trusted to a flat margin.
template <int HO, bool EDGE>
__device__ __forceinline__ void sp_tile_scores(const _Float16 *xfh, const float *P, const f32x4 c4, const int p0, const int L, const int K, const int code0,
                                               const float kx, const float be, float &bs, float &bc, int &bidx) {
    const int hb = p0 + L - HO;                                          // a multiple of 4 — S(p0 + r) and S(p0 + L + r) from three 8-byte reads of xf
    const half4 lo = *reinterpret_cast<const half4 *>(xfh + p0), h0 = *reinterpret_cast<const half4 *>(xfh + hb), h1 = *reinterpret_cast<const half4 *>(xfh + hb + 4);
    const _Float16 hh[8] = { h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3] };
    float slo = P[p0 >> 2], shi = P[hb >> 2];
#pragma unroll
    for (int q = 0; q < HO; q++) shi += (float)hh[q] * (float)hh[q];   // S(p0 + L)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float e = shi - slo;
        slo += (float)lo[r] * (float)lo[r];
        shi += (float)hh[HO + r] * (float)hh[HO + r];
        float ac = fabsf(c4[r]);
        const float rs = __builtin_amdgcn_rsqf(e);
        float sc = __builtin_fmaf(be * rs, rs, (ac + kx) * rs);
        if (EDGE) { const bool in = p0 + r <= K; ac = in ? ac : -1.f; sc = in ? sc : 0.f; }
        bs = sp_hwmax(bs, sc);
        const bool up = ac > bc;
        bc = up ? ac : bc; bidx = up ? code0 + r : bidx;
    }
}
template <int HO>
__device__ __forceinline__ void sp_scores(const _Float16 *xfh, const float *P, const f32x4 (&acc)[SP_MAXT], int wave, int nT2, int n, int g, int L, int K,
                                          const float kx, const float be, float &bs, float &bc, int &bidx) {
#pragma unroll
    for (int t = 0; t < SP_MAXT; t++) {
        const int tile = wave + SP_WAVES * t;
        if (tile < nT2) {
            const int p0 = 256 * tile + 16 * n + 4 * g;
            if (256 * tile + 255 > K) sp_tile_scores<HO, true>(xfh, P, acc[t], p0, L, K, 4 * t, kx, be, bs, bc, bidx);       // (uniform) the tile the arg-max range ends in
            else sp_tile_scores<HO, false>(xfh, P, acc[t], p0, L, K, 4 * t, kx, be, bs, bc, bidx);
        }
    }
}

#ifndef SP_MINW
#define SP_MINW 8                      // waves per SIMD the register allocation leaves room for
#endif
__global__ __launch_bounds__(SP_THREADS, SP_MINW)
void k_scan_pre(const ScanPreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int item = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ScanTpl tp = a.tpl[j];
    ScanPre *out = a.out + (size_t)item * SC_NTPL + j;
    // profiling aid (SONDE_SP_PROF): thread 0 of template 1's workgroups adds the cycles since the previous mark to phase k
#define SP_MARK(k) do { if (a.prof && j == 1 && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(a.prof + (k), t_ - t_prev); t_prev = t_; } } while (0)
    unsigned long long t_prev = a.prof ? __builtin_readcyclecounter() : 0ull;
    if (!tp.active) { if (tid == 0) *out = ScanPre{0.f, 0.f, 0, 0u, 0.f, 0}; return; }
    const ScanItem it = a.items[item];
    const int K = a.K, L = tp.L, wl = K + L;
    const int nT1 = (wl + 255) >> 8, nT2 = (K + 1 + 255) >> 8, nc2 = a.nc2[j];
    const int padL = a.opt_iq ? a.ws_pad : 0;
    // LDS carve-up (halves / floats); every array starts on a 16-byte boundary
    const int NXH = (256 * nT1 + 32 * a.nc1 + 48 + 7) & ~7;
    int NXF = 256 * nT2 + 32 * nc2 + 48; if (NXF < 256 * nT1 + 8) NXF = 256 * nT1 + 8; NXF = (NXF + 7) & ~7;
    const int NB = (a.opt_iq && NXH > NXF) ? NXH : NXF;
    _Float16 *xh = reinterpret_cast<_Float16 *>(sp_smem);                 // padL zeros, the window minus 0.98 dc, zeros
    _Float16 *xfh = xh;                                                    // the filtered window, zeros behind it — IN PLACE of the unfiltered one: every output of the
                                                                           // low-pass is in an accumulator register before the first is stored (a barrier between)
    float *P = reinterpret_cast<float *>(xfh + NB);                        // P[i / 4] = sum_{q<i} xf[q]^2 for i = 0, 4, 8, .. 256 nT1 (the up to three squares between come from xfh)
    const int NP4 = ((64 * nT1 + 1) + 3) & ~3;
    _Float16 *sA = reinterpret_cast<_Float16 *>(P + NP4);                  // A fragments: the FM low-pass's nc1 steps, then the template's nc2 steps
    __shared__ float s_f[2 * SP_WAVES], s_mx[SP_WAVES], s_mn[SP_WAVES], s_rc[SP_WAVES], s_rs[SP_WAVES], s_cv[SP_WAVES];
    __shared__ int s_i[SP_WAVES];

    // ---- the window: xn[i] = stream[pos - (K+L-1) + i], i < K+L (dft_detect.c:378-379); dc over its last 2L samples (:389-391).
    // A thread owns the pairs i = 2 tid + 1024 r + {0, 1}: ring index and byte offset in 32 bits (the ring length is a power of two, so the window's start
    // may wrap modulo 2^32), and everything that depends on where the window or the dc range ends is decided per r on wave-uniform values — the
    // per-sample compares of round 3's form were a third of this kernel's vector instructions
    const uint32_t mask = (uint32_t)a.ring_len - 1;
    const char *str = reinterpret_cast<const char *>(a.fm + ((size_t)tp.stream * a.n_ch + it.ch) * a.ring_len);
    const uint32_t s32 = it.pos - (uint32_t)(wl - 1);
    const int first = it.pos >= (uint32_t)(wl - 1) ? 0 : (wl - 1) - (int)it.pos;      // samples in front of the stream's first one are zeros
    float v[SP_CH];
#pragma unroll
    for (int r = 0; r < SP_CH / 2; r++) {
        const int i0 = 2 * tid + 2 * SP_THREADS * r;
        const uint32_t x0 = (s32 + (uint32_t)i0) & mask, x1 = (x0 + 1u) & mask;
        if (first == 0 && 2 * SP_THREADS * (r + 1) <= wl) {              // (uniform) the whole row is inside the window
            v[2 * r] = *reinterpret_cast<const float *>(str + (x0 << 2)); v[2 * r + 1] = *reinterpret_cast<const float *>(str + (x1 << 2));
        } else {
            v[2 * r]     = (i0 >= first && i0 < wl) ? *reinterpret_cast<const float *>(str + (x0 << 2)) : 0.f;
            v[2 * r + 1] = (i0 + 1 >= first && i0 + 1 < wl) ? *reinterpret_cast<const float *>(str + (x1 << 2)) : 0.f;
        }
    }
    // A fragments -> registers now (their latency overlaps the window's), -> LDS behind the window
    const int ncl = a.opt_iq ? a.nc1 : 0;                                 // steps of the FM low-pass, in front of the template's
    const int nA = min(ncl + nc2, SP_ACAP);                               // steps of 1 KB = 64 uint4 staged now; a longer template reloads (below)
    uint4 areg[SP_AQ];
#pragma unroll
    for (int q = 0; q < SP_AQ; q++) {
        const int idx = tid + SP_THREADS * q;                             // uint4 index into the concatenated fragment list
        const int stp = idx >> 6;
        areg[q] = make_uint4(0u, 0u, 0u, 0u);
        if (stp < nA) {
            const bool lp = a.opt_iq && stp < a.nc1;
            const uint16_t *src = lp ? a.a_ws + (size_t)tp.lpfm * a.nc1 * 512 + (size_t)stp * 512 : a.a_match + a.a_off[j] + (size_t)(stp - (a.opt_iq ? a.nc1 : 0)) * 512;
            areg[q] = reinterpret_cast<const uint4 *>(src)[idx & 63];
        }
    }
    // what the -0.98 dc constant loses in the filter's first taps-1 outputs (tile 0): requested now, used behind the low-pass
    float tl[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.opt_iq && wave == 0) {
        const float *tail = a.ws_tail + tp.lpfm * a.taps;
#pragma unroll
        for (int r = 0; r < 4; r++) { const int i = 16 * (lane & 15) + 4 * (lane >> 4) + r; if (i < a.taps - 1) tl[r] = tail[i]; }
    }
    // one reduction for everything the conversion needs: the dc sum and the window's extremes — max_i |v[i] - c| = max(vmax - c, c - vmin) to the bit
    // (the rounded difference is monotonic in v), so the level of the window is known without a second pass behind the dc
    float dcp = 0.f, vmx = -3.0e38f, vmn = 3.0e38f;
#pragma unroll
    for (int r = 0; r < SP_CH / 2; r++) {
        const int lo_r = 2 * SP_THREADS * r, hi_r = lo_r + 2 * SP_THREADS, i0 = 2 * tid + lo_r;
        if (hi_r <= wl) { vmx = fmaxf(vmx, fmaxf(v[2 * r], v[2 * r + 1])); vmn = fminf(vmn, fminf(v[2 * r], v[2 * r + 1])); }
        else if (lo_r < wl) {
            if (i0 < wl) { vmx = fmaxf(vmx, v[2 * r]); vmn = fminf(vmn, v[2 * r]); }
            if (i0 + 1 < wl) { vmx = fmaxf(vmx, v[2 * r + 1]); vmn = fminf(vmn, v[2 * r + 1]); }
        }
        if (lo_r >= K - L && hi_r <= wl) dcp += v[2 * r] + v[2 * r + 1];
        else if (hi_r > K - L && lo_r < wl) {                            // (samples behind the window are zeros: only the range's start needs a test)
            if (i0 >= K - L) dcp += v[2 * r];
            if (i0 + 1 >= K - L) dcp += v[2 * r + 1];
        }
    }
    SP_DPP_SCAN(dcp, sp_addf, 0);
    SP_DPP_SCAN(vmx, sp_hwmax, (int)0xff7fffffu);                        // -FLT_MAX
    SP_DPP_SCAN(vmn, sp_hwmin, 0x7f7fffff);                              // +FLT_MAX
    if (lane == 63) { s_f[wave] = dcp; s_mx[wave] = vmx; s_mn[wave] = vmn; }
    __syncthreads();
    float dc = 0.f;
    if (a.opt_dc) { float sm = 0.f; for (int w = 0; w < SP_WAVES; w++) sm += s_f[w]; dc = sm * __builtin_amdgcn_rcpf(2.0f * (float)L); }      // (a bound's dc: 1 ulp of the mean is 1e-7 of the window)
    for (int w = 0; w < SP_WAVES; w++) { vmx = fmaxf(vmx, s_mx[w]); vmn = fminf(vmn, s_mn[w]); }
    SP_MARK(0);                                                           // window + A fragments requested, dc and level known
    const float dcs = 0.98f * dc;
    // The score c / sqrt(e) does not depend on the scale of the window, the f16 operands do: a window of a few LSB of FM audio (1 / 32768 = 3e-5) would
    // sit in f16's subnormals.  The window is therefore brought to [0.5, 1) by a power of two (exact) before the conversion, so that the rounding — and
    // with it the margin of the bound — is the same whatever the input level.
    const float amax = fmaxf(fmaxf(vmx - dcs, dcs - vmn), 0.f);
    int ex = 0;
    if (amax > 0.f) { (void)frexpf(amax, &ex); ex = ex > 120 ? 120 : (ex < -120 ? -120 : ex); }
    const float wscale = ldexpf(1.0f, -ex);
    _Float16 *dst = a.opt_iq ? xh + padL : xfh;                          // (padL is a multiple of 8: pairs are 4-byte aligned)
    const int ndst = NB - (a.opt_iq ? padL : 0);
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < SP_CH / 2; r++) {
        const int i0 = 2 * tid + 2 * SP_THREADS * r;
        half2v h = { (_Float16)((v[2 * r] - dcs) * wscale), (_Float16)((v[2 * r + 1] - dcs) * wscale) };
        if (2 * SP_THREADS * (r + 1) > wl) {                              // (uniform) the row the window ends in, and the rows behind it
            if (i0 >= wl) h[0] = (_Float16)0.f;
            if (i0 + 1 >= wl) h[1] = (_Float16)0.f;
        }
        if (i0 < ndst) *reinterpret_cast<half2v *>(dst + i0) = h;        // (ndst is even)
    }
    for (int i = SP_CH * SP_THREADS + 2 * tid; i < ndst; i += 2 * SP_THREADS) *reinterpret_cast<half2v *>(dst + i) = half2v{ (_Float16)0.f, (_Float16)0.f };
    if (a.opt_iq) for (int i = tid; i < padL; i += SP_THREADS) xh[i] = (_Float16)0.f;
#pragma unroll
    for (int q = 0; q < SP_AQ; q++) { const int idx = tid + SP_THREADS * q; if ((idx >> 6) < nA) reinterpret_cast<uint4 *>(sA)[idx] = areg[q]; }
    __syncthreads();

    SP_MARK(1);                                                           // window scaled, converted, stored; A fragments stored
    const int n = lane & 15, g = lane >> 4;
    // ---- FM low-pass (X *= WS[lpFM], dft_detect.c:396-399): xf[i] = sum_t ws[t] xn'[i-t]; the constant -0.98 dc reaches every sample of the
    // reference's (circular, zero padded) array, so the first taps-1 outputs get the part of it the filter has not seen yet (ws_tail)
    if (a.opt_iq) {
        f32x4 acc[SP_MAXT];
#pragma unroll
        for (int t = 0; t < SP_MAXT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sp_toeplitz(xh, sA, a.nc1, wave, nT1, lane, acc);
        __syncthreads();                                                  // every wave has read what it needs of the unfiltered window: the outputs take its place
#pragma unroll
        for (int t = 0; t < SP_MAXT; t++) {
            const int tile = wave + SP_WAVES * t;
            if (tile < nT1) {
                const int i0 = 256 * tile + 16 * n + 4 * g;
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float x = acc[t][r];
                    if (t == 0 && wave == 0) x -= dcs * wscale * tl[r];      // (taps - 1 <= 256: tile 0 only; tl is 0 from taps - 1 on)
                    h[r] = (_Float16)x;
                }
                if (256 * (tile + 1) > wl) {                                  // (uniform) the tile the window ends in
#pragma unroll
                    for (int r = 0; r < 4; r++) if (i0 + r >= wl) h[r] = (_Float16)0.f;
                }
                *reinterpret_cast<half4 *>(xfh + i0) = h;
            }
        }
        for (int i = 256 * nT1 + tid; i < NB; i += SP_THREADS) xfh[i] = (_Float16)0.f;
        __syncthreads();
    }

    SP_MARK(2);                                                           // FM low-pass
    // ---- prefix sums of xf^2 (the 2-norm under the template, dft_detect.c:431-433): thread t owns samples [16 t, 16 t + 16)
    {
        float run[SP_CH];
        float s = 0.f;
        const half8 *src = reinterpret_cast<const half8 *>(xfh + SP_CH * tid);
#pragma unroll
        for (int q = 0; q < SP_CH / 8; q++) {
            const half8 h = (SP_CH * tid + 8 * q < 256 * nT1) ? src[q] : (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 8; r++) { run[8 * q + r] = s; const float x = (float)h[r]; s += x * x; }
        }
        float inc = s;                                            // inclusive scan of the per-thread totals
        SP_DPP_SCAN(inc, sp_addf, 0);
        if (lane == 63) s_f[SP_WAVES + wave] = inc;
        __syncthreads();
        float base = inc - s;
        for (int w = 0; w < wave; w++) base += s_f[SP_WAVES + w];
#pragma unroll
        for (int q = 0; q < SP_CH; q += 4) if (SP_CH * tid + q <= 256 * nT1) P[(SP_CH * tid + q) >> 2] = base + run[q];     // P holds 64 nT1 + 1 sums
        if (tid == SP_THREADS - 1 && 256 * nT1 == SP_CH * SP_THREADS) P[(SP_CH * SP_THREADS) >> 2] = base + s;
        __syncthreads();
    }

    SP_MARK(3);                                                           // prefix sums
    // ---- header correlation c'[p'] = sum_k match[k] xf[p'+k], p' = p - (L-1) in [0, K] (Z = X Fm, Nidft; arg-max range dft_detect.c:415)
    float bc = -1.f, bs = 0.f; int bidx = -1;
    int bp = 0x7fffffff; float bcv = 0.f;
    {
        f32x4 acc[SP_MAXT];
#pragma unroll
        for (int t = 0; t < SP_MAXT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int done = min(nc2, SP_ACAP - ncl);                             // steps staged at the start (SP_ACAP - ncl is even: both are)
        sp_toeplitz(xfh, sA + (size_t)ncl * 512, done, wave, nT2, lane, acc);
        while (done < nc2) {                                            // a template of more than ~900 taps: the next steps, one more round trip each time
            const int take = min(nc2 - done, SP_ACAP);
            __syncthreads();
            for (int idx = tid; idx < 64 * take; idx += SP_THREADS)
                reinterpret_cast<uint4 *>(sA)[idx] = reinterpret_cast<const uint4 *>(a.a_match + a.a_off[j] + (size_t)done * 512)[idx];
            __syncthreads();
            sp_toeplitz(xfh + 32 * done, sA, take, wave, nT2, lane, acc);
            done += take;
        }
        const float kx = a.kap[j] * (amax * wscale);                     // kappa_j X in the window's scaled units (X in [0.5, 1))
        const float be = SP_CP * P[64 * nT1];                            // C_P E_win: P's last entry is the sum over the whole array
        switch (L & 3) {                                                 // (p0 is a multiple of 4: where S(p0 + L) sits between the kept sums is the template's own)
            case 0: sp_scores<0>(xfh, P, acc, wave, nT2, n, g, L, K, kx, be, bs, bc, bidx); break;
            case 1: sp_scores<1>(xfh, P, acc, wave, nT2, n, g, L, K, kx, be, bs, bc, bidx); break;
            case 2: sp_scores<2>(xfh, P, acc, wave, nT2, n, g, L, K, kx, be, bs, bc, bidx); break;
            default: sp_scores<3>(xfh, P, acc, wave, nT2, n, g, L, K, kx, be, bs, bc, bidx); break;
        }
        // the lane's best position and its c, from bidx = 4 t + r
        f32x4 sel = acc[0];
#pragma unroll
        for (int t = 1; t < SP_MAXT; t++) if ((bidx >> 2) == t) sel = acc[t];
        bcv = sel[0];
#pragma unroll
        for (int r = 1; r < 4; r++) if ((bidx & 3) == r) bcv = sel[r];
        if (bidx >= 0) bp = 256 * (wave + SP_WAVES * (bidx >> 2)) + 16 * n + 4 * g + (bidx & 3);
    }
    SP_MARK(4);                                                           // correlation, scores
    {   // the wave's largest |c| (first position wins), the c there, and the largest score
        const uint32_t kc = __float_as_uint(fmaxf(bc, 0.f));
        const uint32_t wc = sp_wave_umax(kc);
        const bool mine = bidx >= 0 && kc == wc;
        const uint32_t wp = ~sp_wave_umax(mine ? ~(uint32_t)bp : 0u);        // the smallest position among the lanes that hold it (0xffffffff: none)
        const unsigned long long own = __ballot(mine && (uint32_t)bp == wp);
        const float wv = own ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(bcv), (int)__builtin_ctzll(own))) : 0.f;
        bs = __uint_as_float(sp_wave_umax(__float_as_uint(bs)));
        bc = own ? __uint_as_float(wc) : -1.f; bp = own ? (int)wp : 0x7fffffff; bcv = wv;
    }
    if (lane == 0) { s_rc[wave] = bc; s_rs[wave] = bs; s_i[wave] = bp; s_cv[wave] = bcv; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SP_WAVES; w++) {
            if (s_rc[w] > bc || (s_rc[w] == bc && s_i[w] < bp)) { bc = s_rc[w]; bp = s_i[w]; bcv = s_cv[w]; }
            if (s_rs[w] > bs) bs = s_rs[w];
        }
        ScanPre r{bs, 0.f, -1, 0u, dc, 0};
        if (bp <= K && bc >= 0.f) {
            const float e = sp_S(P, xfh, bp + L) - sp_S(P, xfh, bp);
            r.mv = e > 0.f ? bcv / sqrtf(e) : 0.f;
            const int mp = bp + L - 1;
            r.mp = (bp == 0 || bp == K) ? -4 : mp;                                 // edge value (dft_detect.c:424)
            r.mpos = it.pos - (uint32_t)(K + L - 1) + (uint32_t)mp;
            if (a.opt_iq) r.mpos -= (uint32_t)(a.lpfm_taps / 2);
        }
        *out = r;
    }
    SP_MARK(5);
    if (a.prof && j == 1 && tid == 0) atomicAdd(a.prof + 7, 1ull);
}

extern "C" int sonde_launch_scan_pre(const ScanPreArgs *a, hipStream_t s) {
    if (a->n_items <= 0) return 0;
    int maxL = 0, maxc2 = 0;
    for (int j = 0; j < SC_NTPL; j++) if (a->tpl[j].active) { if (a->tpl[j].L > maxL) maxL = a->tpl[j].L; if (a->nc2[j] > maxc2) maxc2 = a->nc2[j]; }
    const int wl = a->K + maxL, nT1 = (wl + 255) >> 8, nT2 = (a->K + 1 + 255) >> 8;
    if (nT1 > SP_WAVES * SP_MAXT || 256 * nT1 > SP_CH * SP_THREADS) return -1;       // window longer than 8192 samples
    const size_t nxh = a->opt_iq ? (size_t)((256 * nT1 + 32 * a->nc1 + 48 + 7) & ~7) : 0;
    size_t nxf = (size_t)256 * nT2 + 32 * (size_t)maxc2 + 48; if (nxf < (size_t)256 * nT1 + 8) nxf = (size_t)256 * nT1 + 8; nxf = (nxf + 7) & ~(size_t)7;
    const size_t np4 = (size_t)(((64 * nT1 + 1) + 3) & ~3);
    const size_t lds = 2 * (nxh > nxf ? nxh : nxf) + 4 * np4 + 1024 * (size_t)SP_ACAP;
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_pre), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr = lds;
    }
    hipLaunchKernelGGL(k_scan_pre, dim3(a->n_items, SC_NTPL), dim3(SP_THREADS), lds, s, *a);
    return 0;
}
