// Second oscillator-walk probe: (1) where do the waves of small workgroups land (XCC / SE / CU / SIMD histogram by grid, block and LDS size);
// (2) walk forms whose LDS writes are deferred (a batch of steps in registers, the writes issued behind the next batch's arithmetic).
//   hipcc --offload-arch=gfx950 -O3 -o build/osc_walk2 tools/probes/osc_walk2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#pragma clang fp contract(off)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- forms
// 10: six plain ops per step, a lane per tone, no write
// 11: six plain ops, 8 steps kept in registers, 8 ds_write_b64 behind them (written while the NEXT eight steps compute)
// 12: dpp pair (lane per component), 8 steps in registers, 4 ds_write2_b32 behind them
// 13: packed, 8 steps in registers, 4 ds_write_b128 behind them
// 14: dpp pair, no s_nop (v_mul first, then an independent instruction), no write
#define MUL6(pr, pi, nr, ni) "v_mul_f32 %[t0], " pr ", %[dr]\n\tv_mul_f32 %[t1], " pi ", %[di]\n\tv_mul_f32 %[t2], " pr ", %[di]\n\tv_mul_f32 %[t3], " pi ", %[dr]\n\t" \
                             "v_sub_f32 " nr ", %[t0], %[t1]\n\tv_add_f32 " ni ", %[t2], %[t3]\n\t"

template <int FORM>
__global__ __launch_bounds__(256) void k_walk(float *out, unsigned long long *cyc, unsigned *hw, int steps, int lanes) {
    extern __shared__ __attribute__((aligned(16))) float2 ring_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *ring = ring_all + wave * 512;
    const float ang = 0.05f + 0.001f * (blockIdx.x & 7) + 0.01f * lane;
    const float dr = cosf(ang), di = sinf(ang);
    for (int i = lane; i < 512; i += 64) ring[i] = make_float2(1.f / (1 + i), 0.5f);
    __syncthreads();
    if (lane == 0) {
        const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        hw[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = id; hw[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = xcc;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = 0.f;
    if (FORM == 10 || FORM == 11) {
        if (lane < lanes) {
            float pr = 1.f, pi = 0.f, t0_, t1_, t2_, t3_;
            float r[8], q[8];
            for (int k = 0; k < 8; k++) { r[k] = 0; q[k] = 0; }
            for (int j = 0; j + 8 <= steps; j += 8) {
                if (FORM == 11 && j) {
                    float2 *o = ring + lane * 256 + ((j - 8) & 255);
#pragma unroll
                    for (int k = 0; k < 8; k++) o[k] = make_float2(r[k], q[k]);
                }
                asm volatile(MUL6("%[pr]", "%[pi]", "%[r0]", "%[q0]") MUL6("%[r0]", "%[q0]", "%[r1]", "%[q1]") MUL6("%[r1]", "%[q1]", "%[r2]", "%[q2]") MUL6("%[r2]", "%[q2]", "%[r3]", "%[q3]")
                             MUL6("%[r3]", "%[q3]", "%[r4]", "%[q4]") MUL6("%[r4]", "%[q4]", "%[r5]", "%[q5]") MUL6("%[r5]", "%[q5]", "%[r6]", "%[q6]") MUL6("%[r6]", "%[q6]", "%[r7]", "%[q7]")
                             : [r0] "=&v"(r[0]), [q0] "=&v"(q[0]), [r1] "=&v"(r[1]), [q1] "=&v"(q[1]), [r2] "=&v"(r[2]), [q2] "=&v"(q[2]), [r3] "=&v"(r[3]), [q3] "=&v"(q[3]),
                               [r4] "=&v"(r[4]), [q4] "=&v"(q[4]), [r5] "=&v"(r[5]), [q5] "=&v"(q[5]), [r6] "=&v"(r[6]), [q6] "=&v"(q[6]), [r7] "=&v"(r[7]), [q7] "=&v"(q[7]),
                               [t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_)
                             : [pr] "v"(pr), [pi] "v"(pi), [dr] "v"(dr), [di] "v"(di));
                pr = r[7]; pi = q[7];
            }
            acc = pr + pi;
        }
    } else if (FORM == 12 || FORM == 14) {
        if (lane < 2 * lanes) {
            float x = (lane & 1) ? 0.f : 1.f, ta, tb;
            const float c1 = dr, c2 = (lane & 1) ? di : -di;
            float r[8];
            for (int k = 0; k < 8; k++) r[k] = 0;
            for (int j = 0; j + 8 <= steps; j += 8) {
                if (FORM == 12 && j) {
                    float *o = reinterpret_cast<float *>(ring + (lane >> 1) * 256 + ((j - 8) & 255)) + (lane & 1);
#pragma unroll
                    for (int k = 0; k < 8; k++) o[2 * k] = r[k];
                }
#define DS(xi, xo) "v_mul_f32 %[ta], " xi ", %[c1]\n\ts_nop 0\n\tv_mul_f32_dpp %[tb], " xi ", %[c2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 " xo ", %[ta], %[tb]\n\t"
                asm volatile(DS("%[x]", "%[r0]") DS("%[r0]", "%[r1]") DS("%[r1]", "%[r2]") DS("%[r2]", "%[r3]") DS("%[r3]", "%[r4]") DS("%[r4]", "%[r5]") DS("%[r5]", "%[r6]") DS("%[r6]", "%[r7]")
                             : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]), [r7] "=&v"(r[7]),
                               [ta] "=&v"(ta), [tb] "=&v"(tb)
                             : [x] "v"(x), [c1] "v"(c1), [c2] "v"(c2));
                x = r[7];
            }
            acc = x;
        }
    } else if (FORM == 13) {
        if (lane < lanes) {
            v2f ph = {1.f, 0.f}, ta, tb; const v2f dd = {dr, di};
            v2f r[8];
            for (int k = 0; k < 8; k++) r[k] = v2f{0, 0};
            for (int j = 0; j + 8 <= steps; j += 8) {
                if (j) {
                    v4f *o = reinterpret_cast<v4f *>(ring + lane * 256 + ((j - 8) & 255));
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = v4f{r[2 * k].x, r[2 * k].y, r[2 * k + 1].x, r[2 * k + 1].y};
                }
#define PK(pi_, po_) "v_pk_mul_f32 %[ta], " pi_ ", %[dd] op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %[tb], " pi_ ", %[dd] op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 " po_ ", %[ta], %[tb] neg_lo:[0,1]\n\t"
                asm volatile(PK("%[p]", "%[r0]") PK("%[r0]", "%[r1]") PK("%[r1]", "%[r2]") PK("%[r2]", "%[r3]") PK("%[r3]", "%[r4]") PK("%[r4]", "%[r5]") PK("%[r5]", "%[r6]") PK("%[r6]", "%[r7]")
                             : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]), [r7] "=&v"(r[7]),
                               [ta] "=&v"(ta), [tb] "=&v"(tb)
                             : [p] "v"(ph), [dd] "v"(dd));
                ph = r[7];
            }
            acc = ph.x + ph.y;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) { cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0; out[blockIdx.x * (blockDim.x >> 6) + wave] = acc + ring[(blockIdx.x * 7) & 511].x; }
}

static unsigned h_hw[1 << 18];
static unsigned long long h_cyc[1 << 17];
template <int FORM> void run(const char *name, int grid, int block, int lds_extra, int steps, float *d_out, unsigned long long *d_cyc, unsigned *d_hw, bool placement) {
    const int waves = grid * (block / 64);
    const size_t lds = (size_t)(block / 64) * 512 * sizeof(float2) + lds_extra;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_walk<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_walk<FORM>, dim3(grid), dim3(block), lds, 0, d_out, d_cyc, d_hw, steps, 2);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_walk<FORM>, dim3(grid), dim3(block), lds, 0, d_out, d_cyc, d_hw, steps, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h_cyc, d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h_hw, d_hw, waves * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    double s = 0; unsigned long long mx = 0; for (int i = 0; i < waves; i++) { s += (double)h_cyc[i]; if (h_cyc[i] > mx) mx = h_cyc[i]; }
    printf("%-26s grid %5d x %3d lds %6zu: %7.3f ms %6.1f ns/step  cyc/step mean %6.2f max %6.2f", name, grid, block, lds, ms, ms * 1e6 / steps, s / waves / steps, (double)mx / steps);
    if (placement) {
        // waves per SIMD: key = xcc, se, sh, cu, simd
        std::map<unsigned, int> simd, cu;
        for (int i = 0; i < waves; i++) {
            const unsigned id = h_hw[2 * i], xcc = h_hw[2 * i + 1] & 0xf;
            const unsigned k_cu = (xcc << 16) | (id & 0xff00);                    // cu_id [11:8], sh_id [12], se_id [15:13]
            cu[k_cu]++; simd[(k_cu << 2) | ((id >> 4) & 3)]++;
        }
        int hist[40] = {0}; for (auto &kv : simd) hist[kv.second < 39 ? kv.second : 39]++;
        printf("  | CUs used %zu, SIMDs used %zu, waves/SIMD histogram:", cu.size(), simd.size());
        for (int i = 1; i < 40; i++) if (hist[i]) printf(" %d:%d", i, hist[i]);
    }
    printf("\n");
}

int main() {
    float *d_out; unsigned long long *d_cyc; unsigned *d_hw;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 1 << 20); hipMalloc(&d_hw, 1 << 21);
    const int steps = 48000;
    printf("-- placement (form 14: dpp pair, no write)\n");
    for (int lds : {0, 16 * 1024, 28 * 1024, 36 * 1024}) for (int g : {256, 512, 1024, 2048, 4096}) run<14>("dpp pair", g, 64, lds, steps, d_out, d_cyc, d_hw, true);
    for (int lds : {0, 36 * 1024, 72 * 1024, 140 * 1024}) for (int g : {64, 128, 256, 512, 1024}) run<14>("dpp pair", g, 256, lds, steps, d_out, d_cyc, d_hw, true);
    for (int lds : {0, 36 * 1024, 72 * 1024}) for (int g : {128, 256, 512, 1024, 2048}) run<14>("dpp pair", g, 128, lds, steps, d_out, d_cyc, d_hw, true);
    printf("-- forms, block 256\n");
    for (int g : {64, 256, 512, 1024}) {
        run<10>("plain6, no write", g, 256, 0, steps, d_out, d_cyc, d_hw, false);
        run<11>("plain6 + deferred b64", g, 256, 0, steps, d_out, d_cyc, d_hw, false);
        run<12>("dpp + deferred b32", g, 256, 0, steps, d_out, d_cyc, d_hw, false);
        run<13>("pk3 + deferred b128", g, 256, 0, steps, d_out, d_cyc, d_hw, false);
        run<14>("dpp, no write", g, 256, 0, steps, d_out, d_cyc, d_hw, false);
    }
    return 0;
}
