// accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) vs double sin/cos, for the mixer-table phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double f0, int n, float *cs, float *sn, float *tt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = (float)(f0 * (double)i);
    float fr = t - floorf(t);
    cs[i] = __builtin_amdgcn_cosf(fr);
    sn[i] = __builtin_amdgcn_sinf(fr);
    tt[i] = t;
}
int main() {
    const int n = 150000;
    float *cs, *sn, *tt; hipMalloc(&cs, n * 4); hipMalloc(&sn, n * 4); hipMalloc(&tt, n * 4);
    for (double fhz : {240000.0, -569072.0, 960000.0, 16.0, -1199984.0}) {
        double f0 = fhz / 2400000.0;
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, f0, n, cs, sn, tt);
        std::vector<float> c(n), s(n), t(n);
        hipMemcpy(c.data(), cs, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), sn, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), tt, n * 4, hipMemcpyDeviceToHost);
        double emax = 0, e2 = 0; int tbad = 0;
        for (int i = 0; i < n; i++) {
            float th = (float)(f0 * (double)i);
            if (th != t[i]) tbad++;
            double ph = th * 6.2831853071795864769;
            double ec = c[i] - (double)(float)cos(ph), es = s[i] - (double)(float)sin(ph);
            emax = fmax(emax, fmax(fabs(ec), fabs(es))); e2 += ec * ec + es * es;
        }
        printf("f=%.0f Hz: t mismatches %d, max abs err %.3e, rms %.3e\n", fhz, tbad, emax, sqrt(e2 / (2.0 * n)));
    }
    return 0;
}
