// Issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the decimator uses, at 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    f2 p0 = {a0, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7}, p4 = {8, 9}, p5 = {1, 2}, p6 = {3, 4}, p7 = {5, 6};
    double d0 = a0, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    const float x = out[threadIdx.x & 63], y = 1.0001f; const f2 px = {x, y}; const double dx = x + 1.0;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a0 = fmaf(a0, y, x); a1 = fmaf(a1, y, x); a2 = fmaf(a2, y, x); a3 = fmaf(a3, y, x); a4 = fmaf(a4, y, x); a5 = fmaf(a5, y, x); a6 = fmaf(a6, y, x); a7 = fmaf(a7, y, x); }
        if (OP == 1) { p0 = __builtin_elementwise_fma(p0, px, px); p1 = __builtin_elementwise_fma(p1, px, px); p2 = __builtin_elementwise_fma(p2, px, px); p3 = __builtin_elementwise_fma(p3, px, px);
                       p4 = __builtin_elementwise_fma(p4, px, px); p5 = __builtin_elementwise_fma(p5, px, px); p6 = __builtin_elementwise_fma(p6, px, px); p7 = __builtin_elementwise_fma(p7, px, px); }
        if (OP == 2) { d0 = d0 * dx; d1 = d1 * dx; d2 = d2 * dx; d3 = d3 * dx; d4 = d4 * dx; d5 = d5 * dx; d6 = d6 * dx; d7 = d7 * dx; }
        if (OP == 3) { d0 = d0 + dx; d1 = d1 + dx; d2 = d2 + dx; d3 = d3 + dx; d4 = d4 + dx; d5 = d5 + dx; d6 = d6 + dx; d7 = d7 + dx; }
        if (OP == 4) { a0 = (float)d0 + a0; a1 = (float)d1 + a1; a2 = (float)d2 + a2; a3 = (float)d3 + a3; a4 = (float)d4 + a4; a5 = (float)d5 + a5; a6 = (float)d6 + a6; a7 = (float)d7 + a7;
                       d0 = a0; d1 = a1; d2 = a2; d3 = a3; d4 = a4; d5 = a5; d6 = a6; d7 = a7; }   // cvt_f32_f64 + add + cvt_f64_f32
        if (OP == 5) { a0 = __builtin_amdgcn_sinf(a0); a1 = __builtin_amdgcn_sinf(a1); a2 = __builtin_amdgcn_sinf(a2); a3 = __builtin_amdgcn_sinf(a3); a4 = __builtin_amdgcn_sinf(a4); a5 = __builtin_amdgcn_sinf(a5); a6 = __builtin_amdgcn_sinf(a6); a7 = __builtin_amdgcn_sinf(a7); }
        if (OP == 6) { a0 = __builtin_amdgcn_fractf(a0 + y); a1 = __builtin_amdgcn_fractf(a1 + y); a2 = __builtin_amdgcn_fractf(a2 + y); a3 = __builtin_amdgcn_fractf(a3 + y); a4 = __builtin_amdgcn_fractf(a4 + y); a5 = __builtin_amdgcn_fractf(a5 + y); a6 = __builtin_amdgcn_fractf(a6 + y); a7 = __builtin_amdgcn_fractf(a7 + y); }  // add + fract
        if (OP == 7) { i0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i0, false); i3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i3, false);
                       i4 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i4, false); i5 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i5, false);
                       i6 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i6, false); i7 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, i1), __builtin_bit_cast(s2, i2), i7, false);
                       i1 += i0; i2 += i3; }
        if (OP == 8) { d0 = (double)(unsigned)i0 * dx; d1 = (double)(unsigned)i1 * dx; i0 += (int)d2; i1 += (int)d3; d2 = d0 + d1; d3 = d1 + dx; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}
template <int OP> float run(float *d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, iters);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *d; hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
    const int it = 20000; const double clk = 2.3e9;   // nominal; compare ratios
    // 1024 blocks of 256 = 4 blocks per CU = 4 waves per SIMD; per SIMD: 4 waves * it * N instr
    auto cyc = [&](float ms, int n) { return ms * 1e-3 * clk / (4.0 * it * n); };
    printf("v_fma_f32        %.2f cyc/instr\n", cyc(run<0>(d, it), 8));
    printf("v_pk_fma_f32     %.2f cyc/instr\n", cyc(run<1>(d, it), 8));
    printf("v_mul_f64        %.2f cyc/instr\n", cyc(run<2>(d, it), 8));
    printf("v_add_f64        %.2f cyc/instr\n", cyc(run<3>(d, it), 8));
    printf("cvt32<-64,add,cvt64<-32 (3 instr) %.2f cyc/instr\n", cyc(run<4>(d, it), 24));
    printf("v_sin_f32        %.2f cyc/instr\n", cyc(run<5>(d, it), 8));
    printf("v_add+v_fract    %.2f cyc/instr\n", cyc(run<6>(d, it), 16));
    printf("v_dot2c_i32_i16  %.2f cyc/instr (8 instr/iter)\n", cyc(run<7>(d, it), 8));
    return 0;
}
