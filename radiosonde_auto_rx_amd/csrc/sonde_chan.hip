// sonde_chan.hip — polyphase FFT channelizer (include/sonde_chan.h): kernel and host side.
//
// y_k[m] = sum_n h[n] x[mD-n] e^{-2 pi i k (mD-n)/M}.  With n = r + pM (r < M):
//   y_k[m] = e^{-2 pi i k (mD mod M)/M} * sum_r e^{+2 pi i k r/M} u_m[r],   u_m[r] = sum_p h[r+pM] x[mD - r - pM]
// i.e. per output sample index m: P multiply-adds per branch r (the polyphase filter), one M-point inverse DFT, one phase factor.
// One workgroup = CH_F consecutive m: the span of input samples they need (M P + (CH_F-1) D) is staged once in LDS as raw cs16,
// the branch sums and the transform stay in LDS, every channel gets CH_F consecutive output samples (coalesced rows of 8 CH_F bytes).
// Arithmetic per input sample: (M P + 2.5 M log2 M) / D flops ~ 30 at M = 256, P = 16, D = 200 — against 256 x (mixer + FIR) for
// per-channel front ends; the kernel is bound by its 4 B/sample input stream and M/D x 8 B/sample output stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/sonde_chan.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

#define CH_F 16               // output samples per workgroup
#define CH_THREADS 256

struct ChanArgs {
    const uint32_t *x;        // staged stream: x[i] = sample (i - hist) of this call, hist = T - 1 samples of history in front
    const float *h;           // [T] prototype
    const float2 *tw;         // [M/2] e^{+2 pi i j / M}
    float2 *out; long long out_stride;
    int M, log2M, D, P, T;
    long long m0;             // absolute index of this call's first output sample
    long long n0;             // absolute stream index of x[hist] (first new sample of the call)
    int hist, n_frames;
};

__global__ __launch_bounds__(CH_THREADS)
void k_channelize(const ChanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_c[];
    const int M = a.M, D = a.D, T = a.T, tid = threadIdx.x;
    const int f0 = blockIdx.x * CH_F, nf = min(CH_F, a.n_frames - f0);
    if (nf <= 0) return;
    const int span = T + (CH_F - 1) * D;
    uint32_t *sx = smem_c;                                      // [span] raw cs16
    float2 *su = reinterpret_cast<float2 *>(smem_c + ((span + 1) & ~1));   // [CH_F][M]
    // output sample m (absolute) ends at stream sample m D: in staged coordinates its newest sample is e_m = m D - n0 + hist
    const long long mabs0 = a.m0 + f0;
    const long long e0 = mabs0 * D - a.n0 + a.hist;            // newest staged sample of the block's first output
    const long long lo = e0 - (T - 1);                          // oldest staged sample needed (>= 0 by construction of hist)
    for (int i = tid; i < span; i += CH_THREADS) {
        const long long p = lo + i;
        sx[i] = (p >= 0 && p <= e0 + (long long)(nf - 1) * D) ? a.x[p] : 0u;
    }
    __syncthreads();
    // polyphase branch sums: thread r (and r + 256, ...) keeps its P taps in registers for all CH_F outputs
    for (int r = tid; r < M; r += CH_THREADS) {
        float hp[32];
#pragma unroll
        for (int p = 0; p < 32; p++) hp[p] = p < a.P ? a.h[r + p * M] : 0.f;
        for (int f = 0; f < nf; f++) {
            const int e = (T - 1) + f * D;                     // newest sample of output f in span coordinates
            float ur = 0.f, ui = 0.f;
#pragma unroll
            for (int p = 0; p < 32; p++) {
                if (p < a.P) {
                    const uint32_t w = sx[e - r - p * M];
                    ur = fmaf(hp[p], (float)(int)(short)(w & 0xffffu), ur);
                    ui = fmaf(hp[p], (float)(((int)w) >> 16), ui);
                }
            }
            // bit-reversed store for the in-place radix-2 network below
            su[f * M + (int)(__brev((unsigned)r) >> (32 - a.log2M))] = make_float2(ur * 3.0517578125e-05f, ui * 3.0517578125e-05f);
        }
    }
    __syncthreads();
    // inverse DFT (positive exponent), radix-2 decimation in time, all CH_F transforms of the block side by side
    for (int s = 0; s < a.log2M; s++) {
        const int half = 1 << s;
        for (int b = tid; b < nf * (M / 2); b += CH_THREADS) {
            const int f = b / (M / 2), j = b - f * (M / 2);
            const int pos = j & (half - 1), grp = j >> s;
            const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
            const float2 w = a.tw[pos << (a.log2M - 1 - s)];
            float2 *u = su + f * M;
            const float2 p = u[i0], q = u[i1];
            const float2 t = make_float2(q.x * w.x - q.y * w.y, q.x * w.y + q.y * w.x);
            u[i0] = make_float2(p.x + t.x, p.y + t.y);
            u[i1] = make_float2(p.x - t.x, p.y - t.y);
        }
        __syncthreads();
    }
    // phase factor e^{-2 pi i k (mD mod M)/M} = conj(tw-table entry), then CH_F consecutive samples per channel row
    for (int k = tid; k < M; k += CH_THREADS) {
        float2 *row = a.out + (size_t)k * a.out_stride + f0;
        for (int f = 0; f < nf; f++) {
            const long long md = ((mabs0 + f) * D) % M;
            const int idx = (int)(((long long)k * md) % M);     // e^{-2 pi i idx / M}
            const int ih = idx & (M / 2 - 1);
            float2 w = a.tw[ih];                                // e^{+2 pi i ih / M}
            if (idx >= M / 2) { w.x = -w.x; w.y = -w.y; }
            const float2 v = su[f * M + k];
            row[f] = make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);       // v * conj(w)
        }
    }
}

struct sonde_chan {
    sonde_chan_cfg_t cfg{};
    sonde_chan_info_t info{};
    hipStream_t stream = nullptr;
    uint32_t *d_x = nullptr; float *d_h = nullptr; float2 *d_tw = nullptr; void *d_stage = nullptr;
    float2 *d_own = nullptr;            // sonde_chan_output(): [M][max_frames] owned by the channelizer (callers without a device allocator of their own)
    std::vector<void *> rows;           // sonde_chan_rows_alloc()
    int T = 0, log2M = 0, hist = 0;
    long long n_in = 0, m_out = 0;      // stream samples consumed, output samples produced (per channel)
    double ms = 0; int64_t launches = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
};

extern "C" {

int sonde_chan_create(const sonde_chan_cfg_t *cfg, sonde_chan_t **out) {
    if (!cfg || !out || cfg->abi_version != SONDE_ABI_VERSION) return SONDE_E_ARG;
    if (cfg->M < 16 || cfg->M > 1024 || (cfg->M & (cfg->M - 1)) || cfg->D < 1 || cfg->D > cfg->M || cfg->P < 4 || cfg->P > 32) return SONDE_E_ARG;
    if (cfg->sample_rate < 1 || cfg->max_chunk < 1) return SONDE_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device >= ndev) {
        fprintf(stderr, "libsonde_hip: no usable HIP device (the channelizer has no CPU fallback)\n");
        return SONDE_E_NOGPU;
    }
    HIPCHK(hipSetDevice(cfg->device));
    sonde_chan *c = new sonde_chan();
    c->cfg = *cfg;
    const int M = cfg->M, T = M * cfg->P;
    c->T = T; c->hist = T - 1;
    while ((1 << c->log2M) < M) c->log2M++;
    // prototype: Blackman-windowed sinc, -6 dB at half the channel spacing, unit DC gain
    std::vector<float> h(T);
    {
        const double fc = 0.5 / (double)M;                     // cycles per input sample
        std::vector<double> hd(T);
        double sum = 0.0;
        for (int n = 0; n < T; n++) {
            const double t = n - 0.5 * (T - 1), x = 2.0 * M_PI * fc * t;
            const double sinc = fabs(t) < 1e-12 ? 1.0 : sin(x) / x;
            const double w = 0.42 - 0.5 * cos(2.0 * M_PI * n / (T - 1)) + 0.08 * cos(4.0 * M_PI * n / (T - 1));
            hd[n] = 2.0 * fc * sinc * w; sum += hd[n];
        }
        for (int n = 0; n < T; n++) h[n] = (float)(hd[n] / sum);
    }
    std::vector<float2> tw(M / 2);
    for (int j = 0; j < M / 2; j++) tw[j] = make_float2((float)cos(2.0 * M_PI * j / M), (float)sin(2.0 * M_PI * j / M));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->ev_a)); HIPCHK(hipEventCreate(&c->ev_b));
    HIPCHK(hipMalloc((void **)&c->d_x, ((size_t)c->hist + cfg->max_chunk) * sizeof(uint32_t)));
    HIPCHK(hipMemset(c->d_x, 0, ((size_t)c->hist + cfg->max_chunk) * sizeof(uint32_t)));      // the filter starts from silence
    HIPCHK(hipMalloc((void **)&c->d_h, T * sizeof(float)));
    HIPCHK(hipMalloc((void **)&c->d_tw, (M / 2) * sizeof(float2)));
    HIPCHK(hipMemcpy(c->d_h, h.data(), T * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_tw, tw.data(), (M / 2) * sizeof(float2), hipMemcpyHostToDevice));
    c->info.out_rate_num = cfg->sample_rate; c->info.out_rate_den = cfg->D; c->info.taps = T;
    c->info.max_frames = cfg->max_chunk / cfg->D + 2; c->info.spacing_hz = (float)cfg->sample_rate / (float)M;
    *out = c;
    return 0;
}

void sonde_chan_destroy(sonde_chan_t *c) {
    if (!c) return;
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->ev_a) hipEventDestroy(c->ev_a);
    if (c->ev_b) hipEventDestroy(c->ev_b);
    for (void *p : { (void *)c->d_x, (void *)c->d_h, (void *)c->d_tw, c->d_stage, (void *)c->d_own }) if (p) hipFree(p);
    for (void *p : c->rows) if (p) hipFree(p);
    delete c;
}

int sonde_chan_info(const sonde_chan_t *c, sonde_chan_info_t *info) {
    if (!c || !info) return SONDE_E_ARG;
    *info = c->info;
    return 0;
}

int sonde_chan_process_device(sonde_chan_t *c, const void *d_iq, int32_t n_samples, void *d_out, int64_t out_stride) {
    if (!c || !d_iq || !d_out) return SONDE_E_ARG;
    if (n_samples < 0 || n_samples > c->cfg.max_chunk) return SONDE_E_RANGE;
    const int D = c->cfg.D;
    // output sample m needs the stream up to index m D: those with m D <= n_in + n - 1 are complete after this call
    const long long last = c->n_in + n_samples - 1;
    const long long m_end = last >= 0 ? last / D + 1 : 0;     // exclusive
    const int n_frames = (int)(m_end - c->m_out);
    if (out_stride < n_frames) return SONDE_E_RANGE;
    if (d_iq != (const void *)(c->d_x + c->hist))
        HIPCHK(hipMemcpyAsync(c->d_x + c->hist, d_iq, (size_t)n_samples * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (n_frames > 0) {
        ChanArgs a{};
        a.x = c->d_x; a.h = c->d_h; a.tw = c->d_tw; a.out = (float2 *)d_out; a.out_stride = out_stride;
        a.M = c->cfg.M; a.log2M = c->log2M; a.D = D; a.P = c->cfg.P; a.T = c->T; a.m0 = c->m_out; a.n0 = c->n_in; a.hist = c->hist; a.n_frames = n_frames;
        const int span = c->T + (CH_F - 1) * D;
        const size_t lds = (size_t)((span + 1) & ~1) * sizeof(uint32_t) + (size_t)CH_F * c->cfg.M * sizeof(float2);
        hipEventRecord(c->ev_a, c->stream);
        hipLaunchKernelGGL(k_channelize, dim3((n_frames + CH_F - 1) / CH_F), dim3(CH_THREADS), lds, c->stream, a);
        hipEventRecord(c->ev_b, c->stream);
        c->launches++;
    }
    // the last T-1 samples become the history in front of the next call's samples
    if (n_samples >= c->hist) HIPCHK(hipMemcpyAsync(c->d_x, c->d_x + n_samples, (size_t)c->hist * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    else if (n_samples > 0) {
        // short call: shift the window by n_samples (overlapping ranges: through the staging buffer)
        if (!c->d_stage) HIPCHK(hipMalloc(&c->d_stage, (size_t)c->hist * sizeof(uint32_t)));
        HIPCHK(hipMemcpyAsync(c->d_stage, c->d_x + n_samples, (size_t)c->hist * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_x, c->d_stage, (size_t)c->hist * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    }
    c->n_in += n_samples; c->m_out = m_end;
    if (hipPeekAtLastError() != hipSuccess) { fprintf(stderr, "libsonde_hip: channelizer launch failed: %s\n", hipGetErrorString(hipGetLastError())); return SONDE_E_NOGPU; }
    return n_frames;
}

int sonde_chan_process_host(sonde_chan_t *c, const void *h_iq, int32_t n_samples, void *d_out, int64_t out_stride) {
    if (!c || !h_iq) return SONDE_E_ARG;
    if (n_samples < 0 || n_samples > c->cfg.max_chunk) return SONDE_E_RANGE;
    // the stream's samples go straight behind the history; process_device's own copy is then a no-op onto itself
    HIPCHK(hipMemcpyAsync(c->d_x + c->hist, h_iq, (size_t)n_samples * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                   // the caller may reuse h_iq
    return sonde_chan_process_device(c, c->d_x + c->hist, n_samples, d_out, out_stride);
}

int sonde_chan_sync(sonde_chan_t *c) {
    if (!c) return SONDE_E_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->launches > 0) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->ms = ms; }
    return 0;
}

void *sonde_chan_stream(sonde_chan_t *c) { return c ? (void *)c->stream : nullptr; }

// For callers that have no device allocator (the C receiver host/sonde_wideband.c): an output array owned by the channelizer, row buffers for
// the decoder engines, and the copy of chosen channel rows into such a buffer (on the channelizer's stream, behind the call that produced them).
int sonde_chan_output(sonde_chan_t *c, void **d_out, int64_t *out_stride) {
    if (!c || !d_out || !out_stride) return SONDE_E_ARG;
    if (!c->d_own) {
        HIPCHK(hipMalloc((void **)&c->d_own, (size_t)c->cfg.M * c->info.max_frames * sizeof(float2)));
        HIPCHK(hipMemset(c->d_own, 0, (size_t)c->cfg.M * c->info.max_frames * sizeof(float2)));
    }
    *d_out = c->d_own; *out_stride = c->info.max_frames;
    return 0;
}
int sonde_chan_rows_alloc(sonde_chan_t *c, int32_t n_rows, void **d_rows) {
    if (!c || !d_rows || n_rows < 1) return SONDE_E_ARG;
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)n_rows * c->info.max_frames * sizeof(float2)) != hipSuccess) return SONDE_E_NOMEM;
    HIPCHK(hipMemset(p, 0, (size_t)n_rows * c->info.max_frames * sizeof(float2)));
    c->rows.push_back(p);
    *d_rows = p;
    return 0;
}
int sonde_chan_gather(sonde_chan_t *c, const void *d_out, int64_t out_stride, const int32_t *channels, int32_t n_rows, int32_t n_frames, void *d_rows) {
    if (!c || !d_out || !channels || !d_rows || n_rows < 0 || n_frames < 0 || n_frames > c->info.max_frames) return SONDE_E_ARG;
    for (int r = 0; r < n_rows; r++) {
        if (channels[r] < 0) continue;                            // row not in use
        if (channels[r] >= c->cfg.M) return SONDE_E_RANGE;
        HIPCHK(hipMemcpyAsync((float2 *)d_rows + (size_t)r * c->info.max_frames, (const float2 *)d_out + (size_t)channels[r] * out_stride,
                              (size_t)n_frames * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

int sonde_chan_kernel_ms(sonde_chan_t *c, double *avg_ms, int64_t *launches) {
    if (!c) return SONDE_E_ARG;
    sonde_chan_sync(c);
    if (avg_ms) *avg_ms = c->ms;                               // the last launch
    if (launches) *launches = c->launches;
    return 0;
}

}  // extern "C"
