// sonde_probe.hip — what this box's HBM delivers to a plain read stream, measured in the bench's own process next to the roofline's nominal peak
// ("measured device HBM peak (stream/copy test in the same run)"): boxes of the pool differ by several per cent,
// and a fraction of a nominal 8 TB/s cannot tell a slow box from slow code.  The access pattern is the decimator's — every byte once, 16 bytes per
// lane, consecutive lanes consecutive addresses, non-temporal — without any arithmetic behind it.
#include "../../include/sonde_hip.h"
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_read(const v4u *p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // four loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const v4u a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride), c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const v4u a = __builtin_nontemporal_load(p + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[0] = acc;                     // (keeps the loads)
}

extern "C" int sonde_probe_read_gbps(const void *d_buf, size_t bytes, int32_t reps, double *gbps) {
    if (!d_buf || !gbps || bytes < (1u << 20) || reps < 1) return SONDE_E_ARG;
    unsigned *sink = nullptr;
    if (hipMalloc((void **)&sink, 4) != hipSuccess) return SONDE_E_NOGPU;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) hipEventDestroy(e0); hipFree(sink); return SONDE_E_NOGPU; }
    const size_t n16 = bytes / 16;
    const int grid = 256 * 16;                                 // 16 workgroups per CU
    hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(256), 0, 0, (const v4u *)d_buf, n16, sink);       // warm-up (clocks, TLB)
    double best = 0.0;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(256), 0, 0, (const v4u *)d_buf, n16, sink);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink); return SONDE_E_NOGPU; }
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        if (ms > 0.f) { const double g = (double)n16 * 16.0 / (ms * 1e-3) / 1e9; if (g > best) best = g; }
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink);
    *gbps = best;
    return hipGetLastError() == hipSuccess ? 0 : SONDE_E_NOGPU;
}
