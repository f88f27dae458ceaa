// Achievable HBM read bandwidth for the decimator's access pattern (4.9 GB read once, 16 B per lane, per-wave contiguous segments).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw read_bw.hip && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct __attribute__((packed, aligned(4))) u4u { uint32_t x, y, z, w; };

// (a) plain grid-stride 16-byte reads
__global__ __launch_bounds__(256) void k_stride(const uint4 *p, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// (b) the decimator's pattern: channel -> XCD, 4 waves per WG, each wave walks `tiles` tiles of 12800 bytes (13 x 16 B per lane,
// the last load only in the low lanes), next tile in flight while the current one is "used"
template <bool NT, int DEPTH, bool PARK = false, bool HALO = false, int STORE = 0, int YSTRIDE = 131072>
__global__ __launch_bounds__(256) void k_tiles(const uint32_t *base, long long ch_stride_dw, int n_ch, int wgs_per_ch, int tiles, int n_waves, uint32_t *out, float2 *y = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int ch = (slot / wgs_per_ch) * 8 + xcd, wg = slot % wgs_per_ch;
    if (ch >= n_ch || wg * 4 + wave >= n_waves) return;
    const int seg = wg * 4 + wave;
    // HALO: segments as the decimator cuts them — 64*tiles - 6 rows each, later ones start 6 rows early (tile bases not 128-byte aligned)
    const size_t row0 = HALO ? (seg == 0 ? 0 : (size_t)seg * (64 * tiles - 6) - 6) : (size_t)seg * 64 * tiles;
    const uint32_t *p = base + (size_t)ch * ch_stride_dw + row0 * 50 + 4 * lane;
    uint32_t acc = 0;
    extern __shared__ uint32_t lds[];
    uint32_t *sRaw = lds + wave * 3204;
    u4u buf[DEPTH][13];
    auto fetch = [&](int t, int s) {
        const uint32_t *q = p + (size_t)t * 3200;
#pragma unroll
        for (int v = 0; v < 13; v++) {
            const uint32_t *a = (v < 12 || lane < 32) ? q + 256 * v : q;
            if (NT) { buf[s][v].x = __builtin_nontemporal_load(a); buf[s][v].y = __builtin_nontemporal_load(a + 1); buf[s][v].z = __builtin_nontemporal_load(a + 2); buf[s][v].w = __builtin_nontemporal_load(a + 3); }
            else buf[s][v] = *reinterpret_cast<const u4u *>(a);
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; s++) fetch(s, s);
    for (int t = 0; t < tiles; t += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) {
            if (t + s + DEPTH - 1 < tiles) fetch(t + s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
            if (STORE == 1) y[(size_t)ch * YSTRIDE + ((row0 + (size_t)(t + s) * 64 + lane) & 131071)] = make_float2((float)acc, 1.f);
            if (STORE >= 10) {
                float2 *q = &y[(size_t)ch * 131072 + ((row0 + (size_t)(t + s) * 64 + lane) & 131071)];
                const float2 v = make_float2((float)acc, 1.f);
                if (STORE == 10) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(q), "v"(v) : "memory");
                if (STORE == 11) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
                if (STORE == 12) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
                if (STORE == 13) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(v) : "memory");
                if (STORE == 14) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(q), "v"(v) : "memory");
                if (STORE == 15) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" :: "v"(q), "v"(v) : "memory");
            }
            if (STORE == 2) { float2 *q = &y[(size_t)ch * 131072 + ((row0 + (size_t)(t + s) * 64 + lane) & 131071)]; __builtin_nontemporal_store((float)acc, &q->x); __builtin_nontemporal_store(1.f, &q->y); }
            if (STORE == 3 && ((t + s) & 3) == 3) {          // four tiles' outputs at once: 2 KB per wave
                float4 *q = reinterpret_cast<float4 *>(&y[(size_t)ch * 131072 + ((row0 + (size_t)(t + s - 3) * 64) & 131071)]);
                q[lane] = make_float4((float)acc, 1.f, 2.f, 3.f); q[64 + lane] = make_float4((float)acc, 1.f, 2.f, 3.f);
            }
            if (PARK) {
#pragma unroll
                for (int v = 0; v < 13; v++) { if (v < 12 || lane < 32) *reinterpret_cast<uint4 *>(sRaw + 256 * v + 4 * lane) = make_uint4(buf[s][v].x, buf[s][v].y, buf[s][v].z, buf[s][v].w); }
                acc ^= sRaw[lane * 50 + ((t + s) & 31)];
            } else {
#pragma unroll
                for (int v = 0; v < 13; v++) acc ^= buf[s][v].x ^ buf[s][v].y ^ buf[s][v].z ^ buf[s][v].w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const int C = 512; const long long SR = 2400000;           // dwords per channel
    const size_t bytes = (size_t)C * SR * 4;
    uint32_t *d, *out; (void)hipMalloc(&d, bytes + (1 << 20)); (void)hipMalloc(&out, 64); (void)hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 5; i++) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms; }
        printf("%-44s best %.3f ms  avg %.3f ms  %.2f TB/s (best)\n", name, best, sum / 5, bytes / best / 1e9);
    };
    for (int g : {2048, 8192, 32768})
        timeit(g == 2048 ? "grid-stride uint4, 2048 WGs" : g == 8192 ? "grid-stride uint4, 8192 WGs" : "grid-stride uint4, 32768 WGs",
               [&] { hipLaunchKernelGGL(k_stride, dim3(g), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, out); });
    // 750 tiles per channel: 16 tiles per wave -> 47 waves -> 12 WGs per channel (the last partly idle), as the engine launches it
    for (int tiles : {16, 8, 4}) {
        const int waves = (750 + tiles - 1) / tiles, wgs = (waves + 3) / 4;
        char nm[96];
        snprintf(nm, sizeof nm, "decimator tiles, %d per wave, depth 2", tiles);
        timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<false, 2>), dim3(C * wgs), dim3(256), 0, 0, d, SR, C, wgs, tiles, waves, out); });
        snprintf(nm, sizeof nm, "decimator tiles, %d per wave, depth 2, nt", tiles);
        timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<true, 2>), dim3(C * wgs), dim3(256), 0, 0, d, SR, C, wgs, tiles, waves, out); });
    }
    // the same with the decimator's LDS footprint (51264 B per workgroup -> 3 workgroups = 12 waves per CU), with and without parking
    for (int depth : {2, 3}) {
        const int tiles = 16, waves = 47, wgs = 12;
        char nm[96];
        snprintf(nm, sizeof nm, "tiles 16/wave, nt, 12 waves/CU, depth %d", depth);
        if (depth == 2) timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<true, 2>), dim3(C * wgs), dim3(256), 51264, 0, d, SR, C, wgs, tiles, waves, out); });
        else timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<true, 3>), dim3(C * wgs), dim3(256), 51264, 0, d, SR, C, wgs, tiles, waves, out); });
        snprintf(nm, sizeof nm, "tiles 16/wave, nt, 12 waves/CU, depth %d, park", depth);
        if (depth == 2) timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<true, 2, true>), dim3(C * wgs), dim3(256), 51264, 0, d, SR, C, wgs, tiles, waves, out); });
        else timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<true, 3, true>), dim3(C * wgs), dim3(256), 51264, 0, d, SR, C, wgs, tiles, waves, out); });
    }
    float2 *y; (void)hipMalloc(&y, (size_t)C * 131072 * 8);
    timeit("12 waves/CU, nt, park, halo segments", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, true>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, stores", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 1>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, halo + stores", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, true, 1>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, plain loads, park, halo + stores", [&] { hipLaunchKernelGGL((k_tiles<false, 2, true, true, 1>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    (void)hipFree(y); (void)hipMalloc(&y, (size_t)C * (131072 + 4096) * 8);
    timeit("12 waves/CU, nt, park, stores, ring stride +32", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 1, 131072 + 32>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, stores, ring stride +544", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 1, 131072 + 544>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, stores, ring stride +2080", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 1, 131072 + 2080>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc0", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 10>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc1", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 11>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc0 sc1", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 12>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc0 sc1 nt", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 13>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc1 nt", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 14>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("stores sc0 nt", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 15>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, nt stores", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 2>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    timeit("12 waves/CU, nt, park, 2 KB stores", [&] { hipLaunchKernelGGL((k_tiles<true, 2, true, false, 3>), dim3(C * 12), dim3(256), 51264, 0, d, SR, C, 12, 16, 47, out, y); });
    return 0;
}
