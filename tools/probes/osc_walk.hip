// How fast can one wavefront run the modem's oscillator walk (phi *= d, every product and sum rounded once, each phi written to LDS)?
// The walk is a dependent chain on 2 (or 4) lanes; the question is its period alone on a SIMD and beside 1..7 other walking waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/osc_walk osc_walk.hip && /tmp/osc_walk
// Forms:  0  three packed instructions + ds_write_b64 per step (sonde_fsk.hip FSK_OSC1)
//         1  the same without the LDS write
//         2  packed, one ds_write_b128 per two steps
//         3  a lane per component (re / im), the partner's value through DPP quad_perm: two v_mul_f32 + v_add_f32 + ds_write_b32
//         4  form 3 without the LDS write
//         5  serial float sum from LDS (16 reads, then 16 dependent adds): the fine-timing / Eb/N0 sums
//         6  plain C cmult (what the compiler makes of it), ds write through a pointer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang fp contract(off)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define OSC1(k) "v_pk_mul_f32 %1, %0, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
                "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]\n\tds_write_b64 %4, %0 offset:" #k "*8\n\t"
#define OSC8 OSC1(0) OSC1(1) OSC1(2) OSC1(3) OSC1(4) OSC1(5) OSC1(6) OSC1(7)
#define OSCN(k) "v_pk_mul_f32 %1, %0, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
                "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]\n\t"
#define OSCN8 OSCN(0) OSCN(1) OSCN(2) OSCN(3) OSCN(4) OSCN(5) OSCN(6) OSCN(7)
// two steps into a register quad, one 16-byte write
#define OSC2(k) "v_pk_mul_f32 %1, %0, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %0, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
                "v_pk_add_f32 %3, %1, %2 neg_lo:[0,1]\n\t" \
                "v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
                "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]\n\t"
// component per lane: x = own c1 + other c2 (lane re: c2 = -d.y, lane im: c2 = +d.y); quad_perm [1,0,3,2] swaps the pair
#define OSCD(k) "v_mul_f32 %1, %0, %3\n\ts_nop 0\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                "v_add_f32 %0, %1, %2\n\tds_write_b32 %5, %0 offset:" #k "*8\n\t"
#define OSCD8 OSCD(0) OSCD(1) OSCD(2) OSCD(3) OSCD(4) OSCD(5) OSCD(6) OSCD(7)
#define OSCE(k) "v_mul_f32 %1, %0, %3\n\ts_nop 0\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                "v_add_f32 %0, %1, %2\n\t"
#define OSCE8 OSCE(0) OSCE(1) OSCE(2) OSCE(3) OSCE(4) OSCE(5) OSCE(6) OSCE(7)

template <int FORM>
__global__ __launch_bounds__(64) void k_walk(float *out, unsigned long long *cyc, int steps, int lanes) {
    __shared__ __attribute__((aligned(16))) float2 ring[512];
    const int lane = threadIdx.x;
    const float ang = 0.05f + 0.001f * (blockIdx.x & 7) + 0.01f * lane;
    v2f ph = {1.f, 0.f};
    const v2f dd = {cosf(ang), sinf(ang)};
    for (int i = lane; i < 512; i += 64) ring[i] = make_float2(1.f / (1 + i), 0.5f);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = 0.f;
    if (FORM == 5) {
        if (lane < lanes) {
            const float *pp = reinterpret_cast<const float *>(ring) + lane;
            for (int i = 0; i + 16 <= steps; i += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = pp[2 * ((i + k) & 511)];
#pragma unroll
                for (int k = 0; k < 16; k++) acc = acc + v[k];
            }
        }
    } else if (FORM == 3 || FORM == 4) {
        if (lane < 2 * lanes) {
            float x = (lane & 1) ? 0.f : 1.f, ta, tb;
            const float c1 = dd.x, c2 = (lane & 1) ? dd.y : -dd.y;
            for (int j = 0; j + 8 <= steps; j += 8) {
                uint32_t oaddr = (uint32_t)reinterpret_cast<uintptr_t>(reinterpret_cast<float *>(ring + (lane >> 1) * 256 + (j & 255)) + (lane & 1));
                if (FORM == 3) asm volatile(OSCD8 : "+v"(x), "=&v"(ta), "=&v"(tb) : "v"(c1), "v"(c2), "v"(oaddr) : "memory");
                else           asm volatile(OSCE8 : "+v"(x), "=&v"(ta), "=&v"(tb) : "v"(c1), "v"(c2));
            }
            acc = x;
        }
    } else if (lane < lanes) {
        uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>(ring + lane * 256);
        for (int j = 0; j + 8 <= steps; j += 8) {
            v2f ta, tb;
            const uint32_t oaddr = base + (uint32_t)(j & 255) * 8u;
            if (FORM == 0) asm volatile(OSC8 : "+v"(ph), "=&v"(ta), "=&v"(tb) : "v"(dd), "v"(oaddr) : "memory");
            if (FORM == 1) asm volatile(OSCN8 : "+v"(ph), "=&v"(ta), "=&v"(tb) : "v"(dd));
            if (FORM == 2) {
                v2f p1;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    asm volatile(OSC2(0) : "+v"(ph), "=&v"(ta), "=&v"(tb), "=&v"(p1) : "v"(dd));
                    const v4f q = {p1.x, p1.y, ph.x, ph.y};
                    *reinterpret_cast<v4f *>(ring + lane * 256 + ((j + 2 * u) & 255)) = q;
                }
            }
            if (FORM == 6) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float pr = ph.x, pi = ph.y;
                    ph.x = pr * dd.x - pi * dd.y; ph.y = pr * dd.y + pi * dd.x;
                    ring[lane * 256 + ((j + u) & 255)] = make_float2(ph.x, ph.y);
                }
            }
        }
        acc = ph.x + ph.y;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) { cyc[blockIdx.x] = t1 - t0; out[blockIdx.x] = acc + ring[(blockIdx.x * 7) & 511].x; }
}

template <int FORM> void run(const char *name, int grid, int steps, int lanes, float *d_out, unsigned long long *d_cyc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_walk<FORM>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, steps, lanes);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_walk<FORM>, dim3(grid), dim3(64), 0, 0, d_out, d_cyc, steps, lanes);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[1 << 16];
    (void)hipMemcpy(h, d_cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0; unsigned long long mx = 0; for (int i = 0; i < grid; i++) { s += (double)h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%-28s grid %5d lanes %2d: %7.3f ms  %6.1f ns/step   wave cycles/step mean %6.2f max %6.2f   (clock ~%.2f GHz)\n", name, grid, lanes, ms, ms * 1e6 / steps,
           s / grid / steps, (double)mx / steps, (double)mx / (ms * 1e6));
}

int main() {
    float *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 1 << 20);
    const int steps = 48000;
    const int grids[] = {256, 1024, 2048, 4096, 8192};
    for (int g : grids) {
        run<0>("pk3 + ds_write_b64", g, steps, 2, d_out, d_cyc);
        run<1>("pk3, no write", g, steps, 2, d_out, d_cyc);
        run<2>("pk3 x2 + ds_write_b128", g, steps, 2, d_out, d_cyc);
        run<3>("dpp pair + ds_write_b32", g, steps, 2, d_out, d_cyc);
        run<4>("dpp pair, no write", g, steps, 2, d_out, d_cyc);
        run<6>("plain C cmult + store", g, steps, 2, d_out, d_cyc);
        run<5>("serial sum from LDS", g, steps, 2, d_out, d_cyc);
    }
    run<0>("pk3 + ds_write_b64, 64 lanes", 1024, steps, 64 > 2 ? 2 : 2, d_out, d_cyc);
    return 0;
}
