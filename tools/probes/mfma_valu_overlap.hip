// Does v_mfma_f32_16x16x4_f32 overlap with ordinary f32 VALU work on the same SIMD (gfx950)?
// mode 0: MFMA only, 1: VALU FMA only, 2: both in one wave (independent chains), 3: even waves MFMA / odd waves VALU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float x = out[threadIdx.x & 63], y = 1.0001f;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int i = 0; i < iters; i++) {
        if (do_m) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
        if (do_v) {   // 16 independent FMAs
            v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
            v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
            v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
            v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
template <int MODE> float run(float *d, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *d; hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
    const int iters = 20000;
    for (int wpc : {1, 2}) {      // 256-thread blocks per CU: 1 -> one wave per SIMD, 2 -> two
        const int blocks = 256 * wpc;
        printf("blocks/CU=%d: mfma %.3f ms, valu %.3f ms, both-in-wave %.3f ms, split-waves %.3f ms\n", wpc,
               run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters));
    }
    return 0;
}
