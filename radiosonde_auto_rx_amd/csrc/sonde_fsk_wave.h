// sonde_fsk_wave.h — the 2-/4-FSK modem (utils/fsk.c fsk_demod_core) with the oscillator walk and everything else of a
// channel on wavefronts of their own that meet at workgroup barriers — no wait loops.
//
// Why this form (profiles/r5a_osc_walk_probe.txt, r5b_osc_walk_placement_probe.txt): a wavefront issues one instruction every ~5 cycles
// whatever it depends on (10 when it is alone on its SIMD and the CU's other three SIMDs are busy too), a SIMD takes one every 2 — so a
// channel's serial chains (48 000 dependent oscillator steps per second of signal, the fine-timing and Eb/N0 sums) run at the same speed
// beside other waves as alone, and what a launch needs is (a) few instructions on each chain and (b) at least two waves per SIMD.
// Round 3's pipeline had four waves per channel handing 64-sample pieces over through LDS counters they polled: 54 % of its wave-cycles
// were waits (profiles/r4m_fsk_sq.txt).  Here a channel is
//   SPLIT   two waves: the WALKER runs the oscillator (fsk.c:643-656) — a lane per (tone, component), the partner's component through
//           DPP, three VALU instructions and one LDS write per sample — one piece of FW_L samples ahead of the WORKER, which does all
//           the rest of the modem on its 64 lanes: f_dc = in conj(phi) in place, the integrator windows that are complete by then (a
//           lane per window), the fine-timing products and their serial sum, and, spread over the pieces of a frame, the frequency
//           estimate of the NEXT frame (whose start is known as soon as this frame's length is); behind a frame's last piece the timing
//           estimate -> the next frame's length, soft decisions, Eb/N0, the record.  The two meet at two s_barriers per piece: what one
//           wave publishes in a slot the other reads in a snapshot taken between the barriers, so the schedule is deterministic.
//   !SPLIT  one wave takes both roles in turn (same code, no barriers): for launches with so many channels that every SIMD holds
//           several of them anyway.
// The estimator's transform is kiss_fft's own butterfly network (kiss_fft.c kf_work / kf_bfly4 / kf_bfly2) with FOUR elements per lane
// in registers: the first stage (m = 1: trivial twiddles) needs no LDS at all, every later radix-4 stage one LDS exchange, twiddles
// live in registers, magnitudes are formed in registers and the smoothed spectrum Sf stays in registers for the whole launch.
// Every product and sum is the same operation in the same order as in k_fsk_demod / k_fsk_stream (and the reference) — only the
// schedule differs; tests/test_gpu_fsk.py applies unchanged, and tests/test_fsk_wave_emu.py runs this file on the CPU (tests/emu).
//
// The file compiles for the device (included by sonde_fsk.hip) and, with SONDE_FSK_EMU, for the host under tests/emu/wave_emu.h
// (test infrastructure: every thread a fiber, every cross-lane operation a rendezvous).
#ifndef SONDE_FSK_WAVE_H
#define SONDE_FSK_WAVE_H
#include "sonde_fsk_dev.h"
#include <limits.h>
#include <math.h>
// (tests/test_fsk_wave_emu.py builds the emulator once with this set: every guessed frame start of the estimator wrong, see est_slot)
#ifndef SPEC_TEST_WRONG
#define SPEC_TEST_WRONG 0
#endif
#ifndef FW_EST_SPEC
#define FW_EST_SPEC 1          // (0: A/B builds without the estimator's guessed starts)
#endif

#ifdef SONDE_FSK_EMU
#define FW_DEV static inline
static inline void fw_sync() { emu::wave_rendezvous(); }
static inline void fw_barrier() { emu::Group *g = emu::g_grp; emu::rendezvous(g->wg_gen, g->wg_count, g->n); }
static inline float fw_shfl_xor_f(float v, int mask) {
    emu::Wave &w = emu::my_wave(); const int l = emu::tid() & 63; float *b = reinterpret_cast<float *>(w.buf);
    b[l] = v; emu::wave_rendezvous(); const float r = b[l ^ mask]; emu::wave_rendezvous(); return r;
}
static inline int fw_shfl_xor_i(int v, int mask) {
    emu::Wave &w = emu::my_wave(); const int l = emu::tid() & 63; int *b = reinterpret_cast<int *>(w.buf);
    b[l] = v; emu::wave_rendezvous(); const int r = b[l ^ mask]; emu::wave_rendezvous(); return r;
}
static inline unsigned fw_wave_umax(unsigned v) {
    emu::Wave &w = emu::my_wave(); const int l = emu::tid() & 63; unsigned *b = reinterpret_cast<unsigned *>(w.buf);
    b[l] = v; emu::wave_rendezvous(); unsigned r = 0; for (int i = 0; i < 64; i++) r = b[i] > r ? b[i] : r; emu::wave_rendezvous(); return r;
}
static inline unsigned fw_wave_umin(unsigned v) {
    emu::Wave &w = emu::my_wave(); const int l = emu::tid() & 63; unsigned *b = reinterpret_cast<unsigned *>(w.buf);
    b[l] = v; emu::wave_rendezvous(); unsigned r = 0xFFFFFFFFu; for (int i = 0; i < 64; i++) r = b[i] < r ? b[i] : r; emu::wave_rendezvous(); return r;
}
static inline unsigned fw_float_bits(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
static inline unsigned long long fw_clock() { return 0ull; }
static inline int fw_uni(int v) { return v; }
#else
#define FW_DEV __device__ __forceinline__
// lanes of ONE wave exchanging data through LDS: its DS operations execute in order, so all that is needed is that the compiler keeps them in order
FW_DEV void fw_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
FW_DEV void fw_barrier() { __syncthreads(); }
FW_DEV float fw_shfl_xor_f(float v, int mask) { return __shfl_xor(v, mask); }
FW_DEV int fw_shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask); }
FW_DEV unsigned long long fw_clock() { return __builtin_readcyclecounter(); }
// a value every lane of the wave holds alike: into a scalar register (the channel's bookkeeping is all of this kind, and vector registers decide how many channels a CU holds)
FW_DEV int fw_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// wave-wide unsigned max / min in seven DPP steps (row shifts, then the row broadcasts; lanes a step does not reach keep their value), the result from lane 63 —
// a butterfly of `ds_bpermute` exchanges costs an LDS round trip per step, and the estimator's searches were a third of its time for short frames
#define FW_DPP_STEP(OP, ID, ctrl, rm, bm) do { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp((int)(ID), (int)v, ctrl, rm, bm, false); v = OP(v, o_); } while (0)
FW_DEV unsigned fw_umax2(unsigned a, unsigned b) { return a > b ? a : b; }
FW_DEV unsigned fw_umin2(unsigned a, unsigned b) { return a < b ? a : b; }
FW_DEV unsigned fw_wave_umax(unsigned v) {
    FW_DPP_STEP(fw_umax2, 0u, 0x111, 0xf, 0xf); FW_DPP_STEP(fw_umax2, 0u, 0x112, 0xf, 0xf); FW_DPP_STEP(fw_umax2, 0u, 0x113, 0xf, 0xf);
    FW_DPP_STEP(fw_umax2, 0u, 0x114, 0xf, 0xe); FW_DPP_STEP(fw_umax2, 0u, 0x118, 0xf, 0xc);
    FW_DPP_STEP(fw_umax2, 0u, 0x142, 0xa, 0xf); FW_DPP_STEP(fw_umax2, 0u, 0x143, 0xc, 0xf);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
FW_DEV unsigned fw_wave_umin(unsigned v) {
    FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x111, 0xf, 0xf); FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x112, 0xf, 0xf); FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x113, 0xf, 0xf);
    FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x114, 0xf, 0xe); FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x118, 0xf, 0xc);
    FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x142, 0xa, 0xf); FW_DPP_STEP(fw_umin2, 0xFFFFFFFFu, 0x143, 0xc, 0xf);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
FW_DEV unsigned fw_float_bits(float f) { return __float_as_uint(f); }
#endif

#define FW_FMT_S16  1
#define FW_FMT_CS16 2
#define FW_FMT_CU8  3
#define FW_FMT_CF32 4

// what the two roles of a channel tell each other (LDS).  Counters only grow; a value is written before the counter that announces it.
struct FwCtl {
    // the counters, by slot parity: a role writes ALL of its own into the copy the running slot will end with, the barrier at the end of the slot makes that copy
    // everybody's snapshot, and the other copy is not written before the next barrier — one barrier per slot, and what a role sees does not depend on timing
    int      c[2][12];
    int      E[4];                     // end of frame k (stream position, 0 = first new sample of the launch), valid once FW_C_NIN > k
    float2   dphi[2][4];               // oscillator steps of frame k, valid once FW_C_EST > k
    float    f_est[4][4];              // the estimates, for the frame records
    float2   phi_end[2][4];            // the oscillators behind frame k, normalised (fsk.c:654-656)
    struct { float rx_timing, norm_rx_timing, ppm; int nin, nin_next; } fe[2];   // a frame's timing results, for the finisher
    float2   pend_dphi[4]; float pend_fest[4];   // the estimator's result of a frame between its searches and its publication (est_slot)
    float    fin_state[6];             // EbNodB, snr_est, f_est[4] behind the finisher's last frame, for the channel record
    unsigned pacc[32];                 // profiling aid (SONDE_FSK_PROF): cycles per phase of channel 0's waves (each phase belongs to one role)
};
enum { FW_C_WALK = 0,                  // stream position up to which the oscillator's phi is in the ring
       FW_C_WORK = 1,                  // ... up to which the worker has turned phi into f_dc and integrated: the walker stays within 2 FW_L of it (ring)
       FW_C_EST = 2,                   // frames whose frequency estimate is published
       FW_C_NIN = 3,                   // frames whose length is published
       FW_C_STOP = 4,                  // first frame that does not exist in this launch (INT_MAX while unknown)
       FW_C_WDONE = 5, FW_C_KDONE = 6, FW_C_EDONE = 7,
       FW_C_FE = 8,                    // frames whose integrators and timing are done and handed to the finisher (FwCtl::fe[k & 1])
       FW_C_FEDONE = 9,                // frames the finisher is through with (their half of f_int may be rewritten)
       FW_C_FDONE = 10 };

FW_DEV float2 fw_cmul(const float2 a, const float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
FW_DEV float2 fw_cadd(const float2 a, const float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
FW_DEV float2 fw_csub(const float2 a, const float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// x / 1000.f for an integer-valued x of 16 bits, correctly rounded (tests/test_fsk_div1000.py): q = x r, one Newton step on the exact residual
FW_DEV float fw_div1000(const float x) {
    const float r = 1.0f / 1000.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 1000.0f, x), r, q);
}
// a raw input sample (fetched early, converted when it is used) and the value the modem sees (fsk_demod.c:283-311)
struct FwRaw { uint32_t lo, hi; };
FW_DEV FwRaw fw_load_raw(const int format, const void *base, const uint32_t p) {
    FwRaw r; r.hi = 0;
    if (format == FW_FMT_CS16) r.lo = reinterpret_cast<const uint32_t *>(base)[p];
    else if (format == FW_FMT_CF32) { const uint32_t *q = reinterpret_cast<const uint32_t *>(base) + 2 * (size_t)p; r.lo = q[0]; r.hi = q[1]; }
    else r.lo = reinterpret_cast<const uint16_t *>(base)[p];
    return r;
}
FW_DEV float2 fw_convert(const int format, const FwRaw r) {
    if (format == FW_FMT_CS16) return make_float2(fw_div1000((float)(short)(r.lo & 0xffffu)), fw_div1000((float)(((int)r.lo) >> 16)));
    if (format == FW_FMT_CF32) { float2 v; __builtin_memcpy(&v.x, &r.lo, 4); __builtin_memcpy(&v.y, &r.hi, 4); return v; }
    if (format == FW_FMT_S16) return make_float2(fw_div1000((float)(short)(r.lo & 0xffffu)), 0.f);
    return make_float2(((float)(r.lo & 0xffu) - 127.0f) / 128.0f, ((float)((r.lo >> 8) & 0xffu) - 127.0f) / 128.0f);
}

// first index of the maximum of v[lo..hi) with the reference's `if (v > max)` scan from max = 0 (fsk.c:511-518); dflt if nothing is > 0
FW_DEV int fw_argmax(const float *v, const int lo, const int hi, const int dflt, const int lane) {
    float best = 0.f; int bi = INT_MAX;
    for (int i = lo + lane; i < hi; i += 64) { const float x = v[i]; if (x > best) { best = x; bi = i; } }
    // the largest value (the bit patterns of non-negative floats order like the floats), then the smallest index that holds it
    const unsigned bits = fw_float_bits(best), mx = fw_wave_umax(bits);
    const unsigned mi = fw_wave_umin((bits == mx && bi != INT_MAX) ? (unsigned)bi : 0xFFFFFFFFu);
    return mi == 0xFFFFFFFFu ? dflt : (int)mi;
}

// kiss_fft's radix-4 butterfly (kf_bfly4, forward): separately rounded products and sums in its order
FW_DEV void fw_bfly4(float2 &F0, float2 &F1, float2 &F2, float2 &F3, const float2 s0, const float2 s1, const float2 s2) {
    const float2 s5 = fw_csub(F0, s1);
    const float2 f0 = fw_cadd(F0, s1);
    const float2 s3 = fw_cadd(s0, s2), s4 = fw_csub(s0, s2);
    F2 = fw_csub(f0, s3);
    F0 = fw_cadd(f0, s3);
    F1 = make_float2(s5.x + s4.y, s5.y - s4.x);
    F3 = make_float2(s5.x - s4.y, s5.y + s4.x);
}

template <int LOG2N> struct FwN {
    static constexpr int N = 1 << LOG2N;
    static constexpr int BPW = 256 / N;                                   // transform blocks a wave takes at a time: four elements per lane
    static constexpr int GL = 64 / BPW;                                   // lanes per block = N / 4
    static constexpr int P0 = (LOG2N & 1) ? 2 : 4;                        // radix of the first (innermost) stage, sub-transform length 1
    static constexpr int NS = (LOG2N & 1) ? (LOG2N - 1) / 2 : LOG2N / 2 - 1;   // radix-4 stages behind it: lengths P0 4^r, twiddle strides N / (4 m)
    static constexpr int SPL = N / 64;                                    // Sf bins per lane
    static_assert(LOG2N >= 6 && LOG2N <= 8, "Ndft 64, 128 or 256");
};

#ifndef FW_L
#define FW_L 128                       // samples per piece (a power of two, a multiple of 64); the ring holds 2 FW_L + NT of them per tone
#endif

// LDS of a channel, in floats (the launcher and the emulator harness size the allocation with this)
static inline size_t fw_lds_floats(const int M, const int nsym, const int P, const int R, const int Ndft, const int fin) {
    const size_t W = (size_t)(nsym + 1) * P;
    return 2 * (size_t)M * W * (fin ? 2 : 1) + 2 * (size_t)M * R + (size_t)((2 * nsym + 1) & ~1) + 2 * 2 * 64 + 3 * (size_t)Ndft + 2 * 256 + 256;
}
// the ring: two pieces (the walker's lead), the history a frame starts with, the samples of up to 63 windows left over for the next piece
static inline int fw_ring_len(const int NT, const int step) { int R = 256; while (R < 2 * FW_L + NT + 64 * step + 8) R <<= 1; return R; }

// ---- the oscillator walk: `steps` samples from ring position `o` on (no wrap inside), lanes 0 .. 2M-1 = (tone, component)
//      x = own component; c1 = d.x; c2 = -d.y (real lanes) / +d.y (imaginary lanes):  x' = x c1 + partner c2
//      real: pr dr - pi di;  imaginary: pi dr + pr di = cmult(phi, d) value for value (a + (-b) rounds as a - b, the sum commutes)
#ifndef SONDE_FSK_EMU
// One step: x' = x c1 + partner(x) c2.  Between the sum that makes a value and the DPP read of it sit the first product and a filler (two wait
// states).  A wave pays ~20 cycles for every LDS write it issues beside this chain (profiles/r5a_osc_walk_probe.txt), so eight steps stay in
// registers and go out as four ds_write2_b32 — the fillers of the NEXT eight steps, each placed before its registers are overwritten.
#define FW_ST(src, dst, fill) "v_mul_f32 %[ta], " src ", %[c1]\n\t" fill "\n\tv_mul_f32_dpp %[tb], " src ", %[c2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 " dst ", %[ta], %[tb]\n\t"
#define FW_W2(ra, rb, o0, o1) "ds_write2_b32 %[o], " ra ", " rb " offset0:" #o0 " offset1:" #o1
#define FW_NOP "s_nop 0"
#define FW_BLOCK8_W FW_ST("%[r7]", "%[r0]", FW_W2("%[r0]", "%[r1]", 0, 2)) FW_ST("%[r0]", "%[r1]", FW_NOP) FW_ST("%[r1]", "%[r2]", FW_W2("%[r2]", "%[r3]", 4, 6)) FW_ST("%[r2]", "%[r3]", FW_NOP) \
                    FW_ST("%[r3]", "%[r4]", FW_W2("%[r4]", "%[r5]", 8, 10)) FW_ST("%[r4]", "%[r5]", FW_NOP) FW_ST("%[r5]", "%[r6]", FW_W2("%[r6]", "%[r7]", 12, 14)) FW_ST("%[r6]", "%[r7]", FW_NOP)
#define FW_BLOCK8_N FW_ST("%[r7]", "%[r0]", FW_NOP) FW_ST("%[r0]", "%[r1]", FW_NOP) FW_ST("%[r1]", "%[r2]", FW_NOP) FW_ST("%[r2]", "%[r3]", FW_NOP) \
                    FW_ST("%[r3]", "%[r4]", FW_NOP) FW_ST("%[r4]", "%[r5]", FW_NOP) FW_ST("%[r5]", "%[r6]", FW_NOP) FW_ST("%[r6]", "%[r7]", FW_NOP)
#define FW_REGS8 [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [r4] "+v"(r4), [r5] "+v"(r5), [r6] "+v"(r6), [r7] "+v"(r7), [ta] "=&v"(ta), [tb] "=&v"(tb)
#endif
template <int M>
FW_DEV void fw_walk(float &x, const float c1, const float c2, float2 *ring_m_base, const int R, const uint32_t pos0, const int steps, const int lane) {
    // the ring holds phi[n] at position n; x enters as phi[pos0 - 1] and leaves as phi[pos0 + steps - 1]
    const uint32_t rmask = (uint32_t)R - 1;
#ifdef SONDE_FSK_EMU
    // (the emulator: the even lane of a pair runs both components, the same products and sums; one exchange in front, one behind)
    const float other = fw_shfl_xor_f(x, 1);
    float pr = x, pi = other;
    if (lane < 2 * M && !(lane & 1)) {
        const float dr = c1, di = -c2;
        for (int j = 0; j < steps; j++) {
            const float nr = pr * dr + pi * (-di), ni = pi * dr + pr * di;
            pr = nr; pi = ni;
            ring_m_base[(lane >> 1) * R + ((pos0 + (uint32_t)j) & rmask)] = make_float2(pr, pi);
        }
    }
    const float got = fw_shfl_xor_f(pi, 1);
    x = (lane & 1) ? got : pr;
#else
    if (lane < 2 * M) {
        float *o = reinterpret_cast<float *>(ring_m_base + (lane >> 1) * R + (pos0 & rmask)) + (lane & 1);
        uint32_t oaddr = (uint32_t)reinterpret_cast<uintptr_t>(o);
        int j = 0;
        float ta, tb;
        if (steps >= 8) {
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f, r5 = 0.f, r6 = 0.f, r7 = x;
            asm volatile(FW_BLOCK8_N : FW_REGS8 : [c1] "v"(c1), [c2] "v"(c2));
            for (j = 8; j + 8 <= steps; j += 8) {
                asm volatile(FW_BLOCK8_W : FW_REGS8 : [c1] "v"(c1), [c2] "v"(c2), [o] "v"(oaddr) : "memory");
                oaddr += 64;
            }
            asm volatile(FW_W2("%[r0]", "%[r1]", 0, 2) "\n\t" FW_W2("%[r2]", "%[r3]", 4, 6) "\n\t" FW_W2("%[r4]", "%[r5]", 8, 10) "\n\t" FW_W2("%[r6]", "%[r7]", 12, 14)
                         :: [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [r4] "v"(r4), [r5] "v"(r5), [r6] "v"(r6), [r7] "v"(r7), [o] "v"(oaddr) : "memory");
            oaddr += 64;
            x = r7;
        }
        for (; j < steps; j++) {                                 // what is left of a piece cut by a frame's end
            asm volatile("v_mul_f32 %[ta], %[x], %[c1]\n\ts_nop 0\n\tv_mul_f32_dpp %[tb], %[x], %[c2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %[x], %[ta], %[tb]\n\t"
                         "s_nop 0\n\tds_write_b32 %[o], %[x]"
                         : [x] "+v"(x), [ta] "=&v"(ta), [tb] "=&v"(tb) : [c1] "v"(c1), [c2] "v"(c2), [o] "v"(oaddr) : "memory");
            oaddr += 8;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // all of it in LDS before anybody is told
    }
#endif
}

// ---- a channel.  tid: thread of the workgroup (SPLIT: wave 0 = worker, wave 1 = walker, wave 2 = estimator, wave 3 = finisher); lds: fw_lds_floats() floats, 16-byte aligned
template <int M, int LOG2N, bool SPLIT, int FMT>
FW_DEV void fsk_wave_channel(const FskArgs &a, const int ch, const int tid, float *lds, FwCtl &ctl) {
    typedef FwN<LOG2N> FN;
    constexpr int NDFT = FN::N, BPW = FN::BPW, GL = FN::GL, NS = FN::NS, SPL = FN::SPL;
    const int lane = tid & 63;
    // (FMT: the input format when the kernel is compiled for one — no branches around the sample loads, which then stay in flight across a slot — or 0: a.format)
    const int format = FMT ? FMT : a.format;
    // which wavefront plays which role: the hardware puts wavefront w of every workgroup of a CU on the same SIMD, so with the roles in the same order everywhere the
    // four oscillator walks of a CU's four channels share one SIMD's issue slots while the SIMD of the estimators idles — role_rot turns the order by the channel number
    const int nwv = a.fin ? 4 : 3;
    const int role = SPLIT ? (((tid >> 6) + (a.role_rot ? (ch + a.role_rot - 1) % nwv : 0)) % nwv) : 0;
    const bool is_worker = !SPLIT || role == 0, is_walker = !SPLIT || role == 1, is_est = !SPLIT || role == 2, is_fin = !SPLIT || role == 3;
    const int Ts = a.Ts, P = a.P, nsym = a.nsym, N = a.N, Nmem = a.Nmem, NT = a.NT, R = a.R;
    const int W = (nsym + 1) * P, step = Ts / P;
    const uint32_t rmask = (uint32_t)R - 1;
    // (a.fin: a frame's soft decisions, Eb/N0 and record are made by the finisher while the worker is in the next frame — f_int twice, by frame parity)
    const int fin_on = a.fin ? 1 : 0;
    float2 *s_fint0 = reinterpret_cast<float2 *>(lds);                             // [1 + fin][M][W]
    float2 *s_ring = s_fint0 + (1 + fin_on) * M * W;                               // [M][R]
    float  *s_ebv  = reinterpret_cast<float *>(s_ring + M * R);                    // [2][nsym] (+ pad to an even count)
    float2 *s_ftp  = reinterpret_cast<float2 *>(s_ebv + ((2 * nsym + 1) & ~1));    // [2][64] fine-timing products of a batch of windows, double-buffered
    float  *s_Sf   = reinterpret_cast<float *>(s_ftp + 2 * 64), *s_Sc = s_Sf + NDFT;
    float2 *s_fb   = reinterpret_cast<float2 *>(s_Sc + NDFT);                      // [256] transform scratch
    float  *s_mag  = reinterpret_cast<float *>(s_fb + 256);                        // [256] the round's magnitudes, fftshifted, block after block
    float  *s_sfbak = s_mag + 256;                                                 // [NDFT] Sf as it was before a guessed estimate (est_slot)
    // (of the channel's record only what changes from frame to frame stays in registers; the rest is put together when the launch ends)
    struct FwSt { int nin; float norm_rx_timing, ppm, EbNodB, snr_est; uint32_t rd; } st;
    { const FskChan &g = a.chan[ch]; st.nin = g.nin; st.norm_rx_timing = g.norm_rx_timing; st.ppm = g.ppm; st.EbNodB = g.EbNodB; st.snr_est = g.snr_est; st.rd = g.rd; }
    float *Sf_g = a.Sf + (size_t)ch * NDFT;
    float2 *tail_g = a.tail + (size_t)ch * M * NT;
    const uint32_t wr = a.wr_ch ? a.wr_ch[ch] : a.wr, rd0 = st.rd;
    const int unit = format == FW_FMT_CS16 ? 4 : format == FW_FMT_CF32 ? 8 : 2;
    const char *in_ch = reinterpret_cast<const char *>(a.in) + (size_t)ch * a.ring * unit;
    const int Nmin = a.burst ? N : N - Ts / 2, Nmax = a.burst ? N : N + Ts / 2;
    auto frame_fits = [&](const int k, const uint32_t S, const int nin) -> bool {
        return (int32_t)(wr - (rd0 + S)) >= nin && k < a.rec_cap && (k + 1) * nsym * (M / 2) <= a.sd_cap;
    };
    unsigned long long t_prev = (a.prof && ch == 0) ? fw_clock() : 0ull;
    // (cycles per phase of channel 0's waves, kept in registers and added to a.prof when the launch ends: a mark that wrote to memory cost more than what it measured)
    const bool prof_on = a.prof && ch == 0;
#define FW_MARK(k) do { if (prof_on && (k) >= 0 && lane == 0) { const unsigned long long t_ = fw_clock(); ctl.pacc[(k) >= 0 ? (k) : 0] += (unsigned)(t_ - t_prev); t_prev = t_; } } while (0)

    // ---- set-up: the last NT f_dc samples of the previous launch in front of stream position 0, the control block
    if (is_worker) {
        // (and aside, for the host's repeat of a channel whose launch gives up: a.tail_bak / a.Sf_bak — two copies the host would otherwise enqueue in front of every launch)
        for (int m = 0; m < M; m++) for (int i = lane; i < NT; i += 64) { const float2 v = tail_g[m * NT + i]; s_ring[m * R + ((uint32_t)(i - NT) & rmask)] = v; if (a.tail_bak) a.tail_bak[(size_t)ch * M * NT + m * NT + i] = v; }
        if (lane == 0) {
            const bool f0 = frame_fits(0, 0u, st.nin);
            for (int q = 0; q < 2; q++) {
                ctl.c[q][FW_C_WALK] = 0; ctl.c[q][FW_C_WORK] = 0; ctl.c[q][FW_C_EST] = 0; ctl.c[q][FW_C_WDONE] = 0; ctl.c[q][FW_C_KDONE] = 0; ctl.c[q][FW_C_EDONE] = 0;
                ctl.c[q][FW_C_STOP] = f0 ? INT_MAX : 0; ctl.c[q][FW_C_NIN] = f0 ? 1 : 0;
                ctl.c[q][FW_C_FE] = 0; ctl.c[q][FW_C_FEDONE] = 0; ctl.c[q][FW_C_FDONE] = a.fin ? 0 : 1;      // (without the finisher its wave is not launched)
            }
            ctl.E[0] = st.nin;
        }
        if (lane < 32) ctl.pacc[lane] = 0u;
    }
    if (SPLIT) fw_barrier(); else fw_sync();

    // ---- walker state
    int kw = 0; uint32_t cw = 0, Sw = 0; bool w_done = false, w_have_d = false;
    float wx = 0.f, wc1 = 0.f, wc2 = 0.f;
    // (each role's own loads are made where its loop starts, not here: what is loaded here stays in registers through every other role's set-up)
    auto walker_setup = [&]() { if (is_walker && lane < 2 * M) { const float2 p = a.chan[ch].phi_c[lane >> 1]; wx = (lane & 1) ? p.y : p.x; } };

    // ---- worker state
    const bool frame0 = frame_fits(0, 0u, st.nin);
    int my_nin_seq = frame0 ? 1 : 0, my_stop = frame0 ? INT_MAX : 0, my_fe = 0, par = 0;
    float2 *s_fint = s_fint0;                                   // the worker's half of f_int (frame parity when the finisher is on)
    int kk = 0; uint32_t cc = 0, Sk = 0; bool k_done = false; int frames = 0; uint32_t E_last = 0; bool gave_up = false;
    int nin = st.nin; uint32_t Ek = (uint32_t)st.nin; int32_t wbase = (int32_t)Ek - Nmem; int i_done = 0;
    float t_sum = 0.f;                                          // the timing sum, component lane & 1 (every lane carries one of the two: no divergence around the adds)

    // ---- estimator state: frame being estimated, its start, rounds done / to do; registers of the transform
    int ke = 0; bool e_done = false;
    uint32_t est_S = 0, spec_S = 0; int est_round = 0, est_rounds = 0, est_numffts = 0; bool est_active = false, est_spec = false, est_hold = false;
    const bool same_blocks = a.burst || ((N - Ts / 2) / (NDFT / 2) == (N + Ts / 2) / (NDFT / 2));
    // rounds per slot: a frame's rounds (and the searches behind them) spread over the slots its pieces take
    // (the estimator may be a frame ahead, so the average is what has to fit: rounded, not rounded up)
    const int est_steps = (Nmax / (NDFT / 2) - 1 + BPW - 1) / BPW + 1, est_slots = (Nmin + FW_L - 1) / FW_L;
    const int est_per_slot = (2 * est_steps + est_slots) / (2 * est_slots) > 0 ? (2 * est_steps + est_slots) / (2 * est_slots) : 1;
    const int sub = lane / GL, lt = lane - sub * GL;
    float hn[4]; int ip[4]; float2 tw1[NS > 0 ? NS : 1], tw2[NS > 0 ? NS : 1], tw3[NS > 0 ? NS : 1]; float sf[SPL] = {0.f};
    FwRaw xs[4];
    const float tc = a.tc, omt = 1 - tc;
    auto est_setup = [&]() {
      if (is_est) {
#pragma unroll
        for (int q = 0; q < 4; q++) { ip[q] = a.iperm[4 * lt + q]; hn[q] = a.hann[ip[q]]; xs[q].lo = 0; xs[q].hi = 0; }
#pragma unroll
        for (int r = 0; r < NS; r++) {
            const int m_r = FN::P0 << (2 * r), fs = NDFT / (4 * m_r), u = lt & (m_r - 1);
            tw1[r] = a.tw[u * fs]; tw2[r] = a.tw[2 * u * fs]; tw3[r] = a.tw[3 * u * fs];
        }
#pragma unroll
        for (int r = 0; r < SPL; r++) { sf[r] = Sf_g[lane + 64 * r]; if (a.Sf_bak) a.Sf_bak[(size_t)ch * NDFT + lane + 64 * r] = sf[r]; }
      }
    };
    auto est_fetch = [&](const int j0) {                        // the raw samples of round j0's blocks (one block per group of GL lanes)
        const int j = j0 + sub;
        if (j < est_numffts) {
#pragma unroll
            for (int q = 0; q < 4; q++) xs[q] = fw_load_raw(format, in_ch, (rd0 + est_S + (uint32_t)(ip[q] + j * (NDFT / 2))) & (a.ring - 1));
        }
    };
    auto est_begin = [&](const uint32_t S, const int numffts) {
        est_S = S; est_numffts = numffts; est_round = 0; est_rounds = (numffts + BPW - 1) / BPW; est_active = true;
        est_fetch(0);
    };
    // one round: BPW blocks windowed, transformed, their magnitudes applied to Sf in block order (fsk.c:472-504)
    auto est_round_do = [&]() {
        const int j0 = est_round * BPW, j = j0 + sub;
        const bool act = j < est_numffts;
        float2 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float2 x = act ? fw_convert(format, xs[q]) : make_float2(0.f, 0.f);
            v[q] = make_float2(hn[q] * x.x, hn[q] * x.y);
        }
        FW_MARK(SPLIT ? 18 : -1);
        est_fetch(j0 + BPW);                                    // in flight during the transform
        FW_MARK(SPLIT ? 19 : -1);
        // first stage, sub-transform length 1: the lane's own four elements (positions 4 lt .. 4 lt + 3), twiddle (1, -0): a product with it
        // returns the factor unchanged (up to the sign of a zero, which no later sum or magnitude can see)
        if (FN::P0 == 4) fw_bfly4(v[0], v[1], v[2], v[3], v[1], v[2], v[3]);
        else {
            const float2 a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3];
            v[1] = fw_csub(a0, a1); v[0] = fw_cadd(a0, a1);
            v[3] = fw_csub(b0, b1); v[2] = fw_cadd(b0, b1);
        }
        FW_MARK(SPLIT ? 20 : -1);
        float2 *buf = s_fb + sub * NDFT;
        int pos = 4 * lt, stride = 1;
#pragma unroll
        for (int r = 0; r < NS; r++) {
            const int m_r = FN::P0 << (2 * r);
#pragma unroll
            for (int q = 0; q < 4; q++) buf[pos + q * stride] = v[q];
            fw_sync();
            pos = (lt / m_r) * 4 * m_r + (lt & (m_r - 1)); stride = m_r;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = buf[pos + q * stride];
            fw_bfly4(v[0], v[1], v[2], v[3], fw_cmul(v[1], tw1[r]), fw_cmul(v[2], tw2[r]), fw_cmul(v[3], tw3[r]));
        }
        FW_MARK(SPLIT ? 21 : -1);
        // the lane now holds bins lt + q N/4; fftshift (DC at N/2, fsk.c:484-490) and magnitude
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; q++) s_mag[sub * NDFT + ((lt + q * (NDFT / 4) + NDFT / 2) & (NDFT - 1))] = sqrtf((v[q].x * v[q].x) + (v[q].y * v[q].y));
        }
        fw_sync();
        FW_MARK(SPLIT ? 22 : -1);
        const int nb = est_numffts - j0 < BPW ? est_numffts - j0 : BPW;
#pragma unroll
        for (int r = 0; r < SPL; r++) {                        // Sf = Sf (1 - tc) + |X| tc, block after block (fsk.c:497-503)
            float s = sf[r];
            for (int g = 0; g < nb; g++) s = (s * omt) + (s_mag[g * NDFT + lane + 64 * r] * tc);
            sf[r] = s;
        }
        FW_MARK(SPLIT ? 23 : -1);
        est_round++;
    };
    // behind the last round: the searches (fsk.c:508-581) -> pend_dphi / pend_fest (registers), published as ctl.f_est[ke & 3], ctl.dphi[ke & 1] by est_publish
    auto est_finish = [&]() {
#pragma unroll
        for (int r = 0; r < SPL; r++) { s_Sf[lane + 64 * r] = sf[r]; s_Sc[lane + 64 * r] = sf[r]; }
        fw_sync();
        float2 dphi[4]; float f_est[4];
        {
            int freqi[4];
            for (int m = 0; m < M; m++) {
                const int imax = fw_argmax(s_Sc, a.st, a.en, 0, lane);
                const int f_min = imax - a.f_zero > 0 ? imax - a.f_zero : 0, f_max = imax + a.f_zero < NDFT ? imax + a.f_zero : NDFT;
                fw_sync();
                for (int k = f_min + lane; k < f_max; k += 64) s_Sc[k] = 0.f;
                fw_sync();
                freqi[m] = imax - NDFT / 2;
            }
            for (int i = 1; i < M; i++)                                      // the reference's gnome sort: ascending
                for (int j = i; j > 0 && freqi[j] < freqi[j - 1]; j--) { const int t = freqi[j]; freqi[j] = freqi[j - 1]; freqi[j - 1] = t; }
            for (int m = 0; m < M; m++) { f_est[m] = (float)freqi[m] * ((float)a.Fs / (float)NDFT); dphi[m] = a.dphi_peak[freqi[m] + NDFT / 2]; }
        }
        if (a.est_type) {                                                     // mask estimator (fsk.c:551-581)
            for (int b = a.st + lane; b < a.en - a.len_mask; b += 64) {
                float corr = 0.0f;
                for (int i = 0; i < a.n_mask; i++) corr += s_Sf[b + a.mask_idx[i]];
                s_Sc[b] = corr;
            }
            fw_sync();
            const int b_max = fw_argmax(s_Sc, a.st, a.en - a.len_mask, a.st, lane);
            for (int m = 0; m < M; m++) { f_est[m] = a.f_mask[M * b_max + m]; dphi[m] = a.dphi_mask[M * b_max + m]; }
        }
        if (lane == 0) {
            for (int m = 0; m < M; m++) { ctl.pend_dphi[m] = dphi[m]; ctl.pend_fest[m] = f_est[m]; }
        }
        est_active = false;
    };
    auto est_publish = [&]() {
        if (lane == 0) {
            for (int m = 0; m < M; m++) { ctl.dphi[ke & 1][m] = ctl.pend_dphi[m]; ctl.f_est[ke & 3][m] = ctl.pend_fest[m]; }
        }
    };

    // prefetched raw samples of the worker's next piece (positions cc + lane, cc + 64 + lane, ...) and the timing phasor of its next batch of windows
    FwRaw xr[FW_L / 64];
    auto piece_fetch = [&](const uint32_t c0) {
#pragma unroll
        for (int u = 0; u < FW_L / 64; u++) xr[u] = fw_load_raw(format, in_ch, (rd0 + c0 + (uint32_t)(64 * u + lane)) & (a.ring - 1));
    };
    int ftp_par = 0;
    float2 ph_next = make_float2(1.f, 0.f);
    auto worker_setup = [&]() {
        if (is_worker) {
            if (frame0) { piece_fetch(0u); if (lane < W) ph_next = a.phi_ft[lane]; }
            else k_done = true;
        }
    };
    // snapshot of the control block (taken between the barriers)
    int sn_walk = 0, sn_work = 0, sn_fe = 0, sn_fedone = 0; unsigned sn_est = 0, sn_nin = 0; int sn_stop = INT_MAX;
    int all_done = 0;
    auto snapshot = [&](const int q) {
        const int *c = ctl.c[q];
        sn_walk = fw_uni(c[FW_C_WALK]); sn_work = fw_uni(c[FW_C_WORK]); sn_est = (unsigned)fw_uni(c[FW_C_EST]); sn_nin = (unsigned)fw_uni(c[FW_C_NIN]); sn_stop = fw_uni(c[FW_C_STOP]);
        sn_fe = fw_uni(c[FW_C_FE]); sn_fedone = fw_uni(c[FW_C_FEDONE]);
        all_done = fw_uni((c[FW_C_WDONE] && c[FW_C_KDONE] && c[FW_C_EDONE] && c[FW_C_FDONE]) ? 1 : 0);
    };
    if (SPLIT) fw_barrier(); else fw_sync();
    snapshot(0);
    all_done = 0;

    // (every slot walks or works a piece, finishes a frame or part of an estimate, or is the one slot a role waits for another's publication: the bound is generous)
    const int max_slots = 8 * ((int)((wr - rd0) / FW_L) + 4 * a.rec_cap + 16);
    // ---- the second half of a frame's end: soft decisions, eye, Eb/N0, the record — by the worker itself, or by the finisher while the worker is in the next frame
    auto frame_finish = [&](const int k, const float2 *fint, const float rx_timing, const float norm_rx_timing, const float ppm, const int nin, const int nin_next, FwSt &cs) {
        // ---- soft decisions: integrators resampled by linear interpolation (fsk.c:733-805)
        const int low = (int)floorf(rx_timing), high = (int)ceilf(rx_timing);
        const float fract = rx_timing - (float)low, omf = 1 - fract;
        float *sd = a.sd + (size_t)ch * a.sd_cap + (size_t)k * nsym * (M / 2);
        uint8_t *hb = a.hb + (size_t)ch * a.sd_cap + (size_t)k * nsym * (M / 2);
        for (int i = lane; i < nsym; i += 64) {
            const int sp = (i + 1) * P;
            float tmax[4];
            for (int m = 0; m < M; m++) {
                const float2 lo = fint[m * W + sp + low], hi = fint[m * W + sp + high];
                const float2 tt = fw_cadd(make_float2(omf * lo.x, omf * lo.y), make_float2(fract * hi.x, fract * hi.y));
                tmax[m] = (tt.x * tt.x) + (tt.y * tt.y);
            }
            float mx = tmax[0]; int sym = 0;                                // first maximum wins (fsk.c:760-768)
            for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
            s_ebv[i] = mx; s_ebv[nsym + i] = sqrtf(mx);
            if (M == 2) { sd[i] = sqrtf(tmax[0]) - sqrtf(tmax[1]); hb[i] = (uint8_t)(sym == 1); }
            else {
                hb[2 * i + 1] = (uint8_t)(sym & 1); hb[2 * i] = (uint8_t)((sym & 2) >> 1);
                const float t0 = sqrtf(tmax[0]), t1 = sqrtf(tmax[1]), t2 = sqrtf(tmax[2]), t3 = sqrtf(tmax[3]);
                float lsb = -t0, msb = -t0;
                lsb += t1; msb += -t1;
                lsb += -t2; msb += t2;
                lsb += t3; msb += t3;
                sd[2 * i + 1] = lsb; sd[2 * i] = msb;
            }
        }
        if (a.eye) {                                                // eye diagram samples (fsk.c:857-889), see k_fsk_demod
            const int dec = (int)ceilf(((float)P * 2) / 160.0f), nes = (P * 2) / dec;
            float *eye = a.eye + (size_t)ch * 8 * 160;
            for (int q = lane; q < 8 * nes; q += 64) {
                const int row = q / nes, j = q - row * nes, i = row / M, m = row - i * M;
                const int ind = 2 * P * i + high + 1 + j * dec;
                const float2 vv = (ind < W && m * W + ind >= 0) ? fint[m * W + ind] : make_float2(0.f, 0.f);
                eye[row * 160 + j] = sqrtf((vv.x * vv.x) + (vv.y * vv.y));
            }
        }
        fw_sync();
        FW_MARK(fin_on && SPLIT ? 26 : 11);
        // EbNo estimate (fsk.c:807-836): serial sums in symbol order; lanes with bit 0 clear sum the largest |t|^2 per symbol, the others their roots
        float eacc = 0;
        {
            const float *pe = s_ebv + (lane & 1) * nsym;
            int i = 0;
            for (; i + 16 <= nsym; i += 16) {
                float vv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) vv[u] = pe[i + u];
#pragma unroll
                for (int u = 0; u < 16; u++) eacc += vv[u];
            }
            for (; i < nsym; i++) eacc += pe[i];
        }
        float f_est_cur[4];
        for (int m = 0; m < M; m++) f_est_cur[m] = ctl.f_est[k & 3][m];      // (published before the walker started this frame)
        {
            const float eo = fw_shfl_xor_f(eacc, 1);
            const float e0 = (lane & 1) ? eo : eacc, e1 = (lane & 1) ? eacc : eo;
            const float meanebno = e1 / (float)nsym;
            float stdebno = (e0 / (float)nsym) - (meanebno * meanebno);
            if (stdebno > 0.0) stdebno = (float)sqrt((double)stdebno); else stdebno = 0.0f;
            cs.EbNodB = -6 + (20 * log10f((float)((1e-6 + meanebno) / (1e-6 + stdebno))));
            cs.snr_est = (float)(.5 * cs.snr_est + .5 * cs.EbNodB);
        }
        if (lane == 0) {
            FskFrameRec r; r.nin = nin; r.nin_next = nin_next; for (int m = 0; m < 4; m++) r.f_est[m] = m < M ? f_est_cur[m] : 0.f;
            r.norm_rx_timing = norm_rx_timing; r.ppm = ppm; r.EbNodB = cs.EbNodB; r.snr_est = cs.snr_est;
            a.recs[(size_t)ch * a.rec_cap + k] = r;
        }
    };
    auto est_slot = [&]() {
        // =========================================================== estimator: the frequency estimate of every frame, as early as its start is known
        if (is_est && !e_done) {
            // a speculative estimate (below) whose guess has been settled meanwhile
            if (est_spec && sn_nin > (unsigned)(ke - 1) && (uint32_t)ctl.E[(ke - 1) & 3] != spec_S) {
                // the frame before this one did not come out at its nominal length (one in twenty): Sf back to where it was, the estimate again from the real start
#pragma unroll
                for (int r = 0; r < SPL; r++) sf[r] = s_sfbak[lane + 64 * r];
                est_active = false; est_spec = false; est_hold = false;
            }
            if (sn_stop <= ke && (est_active || est_hold) && est_spec) {                       // (the guessed frame does not exist after all: nothing of it may stay in Sf)
#pragma unroll
                for (int r = 0; r < SPL; r++) sf[r] = s_sfbak[lane + 64 * r];
                est_active = false; est_spec = false; est_hold = false;
            }
            if (!est_active && !est_hold) {
                if (sn_stop <= ke) e_done = true;
                else if (ke == 0 || sn_nin > (unsigned)(ke - 1)) {
                    const uint32_t S = ke == 0 ? 0u : (uint32_t)ctl.E[(ke - 1) & 3];
                    est_spec = false;
                    if (sn_nin > (unsigned)ke) est_begin(S, (int)((uint32_t)ctl.E[ke & 3] - S) / (NDFT / 2) - 1);
                    // ahead of the frame's length: when it is certain to be demodulated by this launch whatever length the timing gives it, and its blocks are the same for all three
                    else if (same_blocks && frame_fits(ke, S, Nmax)) est_begin(S, N / (NDFT / 2) - 1);
                } else if (FW_EST_SPEC && same_blocks && !a.burst && (ke == 1 || sn_nin > (unsigned)(ke - 2))) {
                    // ahead of the frame's START too: the frame before it is known to start at S1 and nineteen frames in twenty are N samples long — the estimate of
                    // frame ke from S1 + N now, kept if that frame's length comes out as N (checked before anything is published), redone otherwise.  This takes the
                    // estimate off the chain timing(k-1) -> start(k+1) -> estimate(k+1) -> walk(k+1) that bounds short frames.  Only where both
                    // frames are certain to be demodulated by this launch whatever their lengths.
                    const uint32_t S1 = ke == 1 ? 0u : (uint32_t)ctl.E[(ke - 2) & 3];
                    if (frame_fits(ke, S1 + (uint32_t)Nmax, Nmax)) {
#pragma unroll
                        for (int r = 0; r < SPL; r++) s_sfbak[lane + 64 * r] = sf[r];
                        spec_S = S1 + (uint32_t)N + (SPEC_TEST_WRONG ? Ts / 2 : 0); est_spec = true;
                        est_begin(spec_S, N / (NDFT / 2) - 1);
                    }
                }
            }
            if (est_active) {
                // (the launch's first frame: all of it now — the others wait for it)
                const int cnt = ke == 0 ? est_rounds + 1 : est_per_slot;
                int r = 0;
                for (; r < cnt && est_round < est_rounds; r++) est_round_do();
                FW_MARK(SPLIT ? 8 : 1);
                // behind the last round the searches, into registers
                if (r < cnt && est_round >= est_rounds) { est_finish(); est_hold = true; FW_MARK(SPLIT ? 13 : 1); }
            }
            if (est_hold && (!est_spec || sn_nin > (unsigned)(ke - 1))) {
                // the estimate is published — a guessed one only once the guess is known to hold (a wrong one was thrown away above)
                est_publish(); est_hold = false; est_spec = false; ke++;
            }
        }
        if (is_est && lane == 0) {
            ctl.c[par ^ 1][FW_C_EST] = ke; ctl.c[par ^ 1][FW_C_EDONE] = e_done ? 1 : 0;
            if (!SPLIT) { ctl.c[par][FW_C_EST] = ke; ctl.c[par][FW_C_EDONE] = e_done ? 1 : 0; }
        }
    };
    auto walker_slot = [&]() {
        // =========================================================== walker
        if (is_walker && !w_done) {
            bool go = true;
            if (sn_nin > (unsigned)kw && cw == (uint32_t)ctl.E[kw & 3] && cw != Sw) {
                // the frame is walked: phi /= |phi| (fsk.c:654-656), on to the next
                const float other = fw_shfl_xor_f(wx, 1);
                const float pr = (lane & 1) ? other : wx, pi = (lane & 1) ? wx : other;
                const float av = sqrtf((pr * pr) + (pi * pi));
                wx = wx / av;
                if (lane < 2 * M) reinterpret_cast<float *>(&ctl.phi_end[kw & 1][lane >> 1])[lane & 1] = wx;
                kw++; Sw = cw; w_have_d = false;
            }
            if (sn_stop <= kw) { w_done = true; go = false; }
            if (go && !w_have_d) {
                if (sn_est > (unsigned)kw) {
                    const float2 d = ctl.dphi[kw & 1][(lane >> 1) & 3];
                    wc1 = d.x; wc2 = (lane & 1) ? d.y : -d.y; w_have_d = true;
                } else go = false;
            }
            if (go) {
                // a piece: up to FW_L samples, cut at the frame's end — or, while the frame's length is not published yet, at the end of its shortest form
                uint32_t ce = cw + (uint32_t)FW_L;
                const uint32_t lim = sn_nin > (unsigned)kw ? (uint32_t)ctl.E[kw & 3] : Sw + (uint32_t)Nmin;
                if ((int32_t)(ce - lim) > 0) ce = lim;
                if (ce == cw) go = false;                                                    // (waiting for the length)
                if ((int32_t)(ce - (uint32_t)sn_work) > 2 * FW_L) go = false;                // the ring: not more than two pieces ahead of the worker
                if (go && ce != cw) {
                    FW_MARK(SPLIT ? 6 : -1);
                    {   // (the ring may end inside the piece)
                        const int n = (int)(ce - cw), room = R - (int)(cw & rmask);
                        if (n <= room) fw_walk<M>(wx, wc1, wc2, s_ring, R, cw, n, lane);
                        else { fw_walk<M>(wx, wc1, wc2, s_ring, R, cw, room, lane); fw_walk<M>(wx, wc1, wc2, s_ring, R, cw + (uint32_t)room, n - room, lane); }
                    }
                    FW_MARK(5);
                    cw = ce;
                }
            }
        }
        if (is_walker && lane == 0) {
            ctl.c[par ^ 1][FW_C_WALK] = (int)cw; ctl.c[par ^ 1][FW_C_WDONE] = w_done ? 1 : 0;
            if (!SPLIT) { ctl.c[par][FW_C_WALK] = (int)cw; ctl.c[par][FW_C_WDONE] = w_done ? 1 : 0; }
        }
    };
    auto worker_slot = [&]() {
        // =========================================================== worker
        if (is_worker && !k_done) {
            const uint32_t whole = cc + (uint32_t)FW_L;
            const uint32_t ce = (int32_t)(whole - Ek) < 0 ? whole : Ek;
            // (with the finisher: this frame's half of f_int is free once the finisher is through with the frame before the last)
            if ((int32_t)((uint32_t)sn_walk - ce) >= 0 && (!fin_on || sn_fedone >= kk - 1)) {
                const int cl = (int)(ce - cc);
                // ---- f_dc = in conj(phi), in place in the ring (fsk.c:645-647)
#pragma unroll
                for (int u = 0; u < FW_L / 64; u++) {
                    const int j = 64 * u + lane;
                    if (j < cl) {
                        const float2 x = fw_convert(format, xr[u]);
#pragma unroll
                        for (int m = 0; m < M; m++) {
                            float2 *q = s_ring + m * R + ((cc + (uint32_t)j) & rmask);
                            const float2 p = *q;
                            *q = fw_cmul(x, make_float2(p.x, -p.y));
                        }
                    }
                }
                // the next piece's samples: in flight while the windows are summed
                if (ce != Ek) piece_fetch(ce);
                fw_sync();
                FW_MARK(2);
                // ---- integrator windows complete by now (fsk.c:659-668): window i reads f_dc[i step .. i step + Ts); a lane per window
                const int32_t span = (int32_t)ce - wbase - Ts;
                int i_new = span < 0 ? 0 : span / step + 1;
                if (i_new > W) i_new = W;
                // whole batches of 64 windows; what is left over waits for the next piece (the ring keeps its samples), the frame's last piece takes all
                if (ce != Ek) i_new = i_done + ((i_new - i_done) & ~63);
                for (int ib = i_done; ib < i_new; ib += 64) {
                    const int i = ib + lane;
                    const float2 ph = ph_next;
                    ph_next = (i + 64 < W) ? a.phi_ft[i + 64] : make_float2(0.f, 0.f);      // (the next batch's, a batch ahead)
                    if (i < i_new) {
                        float ft1 = 0;
                        const uint32_t q0 = (uint32_t)(wbase + i * step) & rmask;
                        float2 acc[M];
#pragma unroll
                        for (int m = 0; m < M; m++) acc[m] = make_float2(0.f, 0.f);
                        if (q0 + (uint32_t)Ts <= (uint32_t)R) {
                            // the Ts samples of a window are contiguous in the ring unless it straddles its end; five at a time from every tone, then the sums in order
                            int j = 0;
                            for (; j + 5 <= Ts; j += 5) {                 // (Ts is 5, 10 or 20 for the sondes)
                                float2 vv[M][5];
#pragma unroll
                                for (int m = 0; m < M; m++)
#pragma unroll
                                    for (int u = 0; u < 5; u++) vv[m][u] = s_ring[m * R + q0 + j + u];
#pragma unroll
                                for (int m = 0; m < M; m++)
#pragma unroll
                                    for (int u = 0; u < 5; u++) acc[m] = fw_cadd(acc[m], vv[m][u]);
                            }
                            for (; j < Ts; j++)
#pragma unroll
                                for (int m = 0; m < M; m++) acc[m] = fw_cadd(acc[m], s_ring[m * R + q0 + j]);
                        } else {
                            for (int j = 0; j < Ts; j++)
#pragma unroll
                                for (int m = 0; m < M; m++) acc[m] = fw_cadd(acc[m], s_ring[m * R + ((q0 + (uint32_t)j) & rmask)]);
                        }
#pragma unroll
                        for (int m = 0; m < M; m++) { s_fint[m * W + i] = acc[m]; ft1 += (acc[m].x * acc[m].x) + (acc[m].y * acc[m].y); }
                        FW_MARK(SPLIT ? 24 : -1);
                        s_ftp[ftp_par * 64 + lane] = make_float2(ft1 * ph.x, ft1 * ph.y);      // fine timing: sum_i (sum_m |f_int[m]|^2) phi_ft[i]  (fsk.c:682-703)
                    }
                    fw_sync();
                    const int nb = i_new - ib < 64 ? i_new - ib : 64;
                    {   // the serial sum in window order; every lane carries component (lane & 1).  A quarter of a full batch's terms are requested before
                        // the first add (sixteen registers: the kernel's register budget is what decides how many channels a CU holds)
                        const float *pq = reinterpret_cast<const float *>(s_ftp + ftp_par * 64) + (lane & 1);
                        if (nb == 64) {
#pragma unroll
                            for (int q = 0; q < 64; q += 16) {
                                float vv[16];
#pragma unroll
                                for (int u = 0; u < 16; u++) vv[u] = pq[2 * (q + u)];
#pragma unroll
                                for (int u = 0; u < 16; u++) t_sum = t_sum + vv[u];
                            }
                        } else {
                            // a batch cut short by the frame's end: groups of 16, 8, 4, 2, 1 terms, each requested as a whole before its adds
                            int q = 0;
#define FW_SUM_GROUP(n) if (nb - q >= (n)) { float vv[n]; _Pragma("unroll") for (int u = 0; u < (n); u++) vv[u] = pq[2 * (q + u)]; _Pragma("unroll") for (int u = 0; u < (n); u++) t_sum = t_sum + vv[u]; q += (n); }
                            FW_SUM_GROUP(16) FW_SUM_GROUP(16) FW_SUM_GROUP(16) FW_SUM_GROUP(8) FW_SUM_GROUP(4) FW_SUM_GROUP(2) FW_SUM_GROUP(1)
#undef FW_SUM_GROUP
                        }
                    }
                    FW_MARK(SPLIT ? 25 : -1);
                    ftp_par ^= 1;
                    if (nb < 64 && i_new < W) {
                        // a batch cut short by the piece's end: the phasors fetched ahead belong to window ib + 64 + lane, the next batch starts at i_new
                        ph_next = (i_new + lane < W) ? a.phi_ft[i_new + lane] : make_float2(0.f, 0.f);
                    }
                }
                if (i_new > i_done) i_done = i_new;
                FW_MARK(3);
                cc = ce;
                if (cc == Ek) {
                    // ================================================ the frame's last piece is in: timing, the next frame's length
                    const float tc0 = fw_shfl_xor_f(t_sum, 1);
                    const float tre = (lane & 1) ? tc0 : t_sum, tim = (lane & 1) ? t_sum : tc0;
                    const float norm_rx_timing = (float)((double)fsk_atan2f(tim, tre) / (2 * 3.14159265358979323846));
                    const float rx_timing = norm_rx_timing * (float)P;
                    const float d_norm = norm_rx_timing - st.norm_rx_timing;
                    st.norm_rx_timing = norm_rx_timing;
                    if (fabsf(d_norm) < .2) {
                        const float appm = (float)(1e6 * d_norm / (float)nsym);
                        st.ppm = (float)(.9 * st.ppm + .1 * appm);
                    }
                    int nin_next = N;
                    if (!a.burst) {
                        if (norm_rx_timing > 0.25) nin_next = N + Ts / 2;
                        else if (norm_rx_timing < -0.25) nin_next = N - Ts / 2;
                    }
                    nin_next = fw_uni(nin_next);
                    const bool more = frame_fits(kk + 1, Ek, nin_next);
                    if (more) { if (lane == 0) ctl.E[(kk + 1) & 3] = (int)(Ek + (uint32_t)nin_next); my_nin_seq = kk + 2; }
                    else my_stop = kk + 1;
                    if (more) { piece_fetch(Ek); ph_next = lane < W ? a.phi_ft[lane] : make_float2(0.f, 0.f); }
                    FW_MARK(10);
                    if (!fin_on) frame_finish(kk, s_fint, rx_timing, norm_rx_timing, st.ppm, nin, nin_next, st);
                    else {
                        if (lane == 0) { ctl.fe[kk & 1].rx_timing = rx_timing; ctl.fe[kk & 1].norm_rx_timing = norm_rx_timing; ctl.fe[kk & 1].ppm = st.ppm; ctl.fe[kk & 1].nin = nin; ctl.fe[kk & 1].nin_next = nin_next; }
                        my_fe = kk + 1;
                    }
                    fw_sync();                                                  // s_fint / s_ebv are rewritten by the next frame
                    st.nin = nin_next;
                    frames++;
                    if (fin_on) s_fint = s_fint0 + (frames & 1) * M * W;
                    E_last = Ek;
                    FW_MARK(4);
                    if (a.test_abort_ch == ch && frames == 1) gave_up = true;     // (test hook: as if the launch had run out of slots — the host repeats the channel)
                    if (!more) k_done = true;
                    else { kk++; Sk = Ek; nin = nin_next; Ek = Sk + (uint32_t)nin; wbase = (int32_t)Ek - Nmem; i_done = 0; t_sum = 0.f; }
                }
            }
        }
        if (is_worker && lane == 0) {
            int *c = ctl.c[par ^ 1];
            c[FW_C_WORK] = (int)cc; c[FW_C_NIN] = my_nin_seq; c[FW_C_STOP] = my_stop; c[FW_C_KDONE] = k_done ? 1 : 0; c[FW_C_FE] = my_fe;
            if (!SPLIT) { c = ctl.c[par]; c[FW_C_WORK] = (int)cc; c[FW_C_NIN] = my_nin_seq; c[FW_C_STOP] = my_stop; c[FW_C_KDONE] = k_done ? 1 : 0; c[FW_C_FE] = my_fe; }
        }
    };
    // =========================================================== finisher (a.fin): soft decisions, Eb/N0 and record of the frames the worker hands over
    int kf = 0; bool f_done = !fin_on;
    FwSt stf = st;
    auto fin_slot = [&]() {
        if (is_fin && !f_done) {
            if (sn_fe > kf) {
                const float rxt = ctl.fe[kf & 1].rx_timing, nrm = ctl.fe[kf & 1].norm_rx_timing, ppm = ctl.fe[kf & 1].ppm;
                const int nn = ctl.fe[kf & 1].nin, nx = ctl.fe[kf & 1].nin_next;
                frame_finish(kf, s_fint0 + (kf & 1) * M * W, rxt, nrm, ppm, nn, nx, stf);
                fw_sync();
                kf++;
                FW_MARK(SPLIT ? 27 : -1);
            } else if (sn_stop <= kf) {
                f_done = true;
                if (lane == 0) { ctl.fin_state[0] = stf.EbNodB; ctl.fin_state[1] = stf.snr_est; }
            }
        }
        if (is_fin && fin_on && lane == 0) {
            ctl.c[par ^ 1][FW_C_FEDONE] = kf; ctl.c[par ^ 1][FW_C_FDONE] = f_done ? 1 : 0;
            if (!SPLIT) { ctl.c[par][FW_C_FEDONE] = kf; ctl.c[par][FW_C_FDONE] = f_done ? 1 : 0; }
        }
    };
    // the end of a slot: what was published in it becomes everybody's snapshot
    auto slot_end = [&](const int mark) {
        if (SPLIT) fw_barrier(); else fw_sync();
        par ^= 1;
        snapshot(par);
        FW_MARK(mark);
    };
    int slot = 0;
    if (SPLIT) {
        // a loop per role, so that a wave holds only its own role's registers (the barriers pair up across the three loops)
        if (role == 3)      for (; !all_done && slot < max_slots; slot++) { fin_slot(); slot_end(28); }
        else if (role == 2) { est_setup(); for (; !all_done && slot < max_slots; slot++) { est_slot(); slot_end(9); } }
        else if (role == 1) { walker_setup(); for (; !all_done && slot < max_slots; slot++) { walker_slot(); slot_end(7); } }
        else                { worker_setup(); for (; !all_done && slot < max_slots; slot++) { worker_slot(); slot_end(0); } }
    } else {
        est_setup(); walker_setup(); worker_setup();
        for (; !all_done && slot < max_slots; slot++) {
            est_slot(); fw_sync(); snapshot(par);
            walker_slot(); fw_sync(); snapshot(par);
            worker_slot(); fw_sync(); snapshot(par);
            fin_slot(); slot_end(0);
        }
    }
    if (SPLIT) fw_barrier(); else fw_sync();
    if (prof_on && tid < 32 && tid != 15 && tid != 16 && tid != 17) a.prof[tid] += (unsigned long long)ctl.pacc[tid];
    if (prof_on && lane == 0) {
        if (tid == 0) { a.prof[15] = SPLIT ? 3 : 2; a.prof[16] += (unsigned long long)slot; a.prof[17] += (unsigned long long)frames; }
    }

    // ---- the launch is over for this channel: Sf, the oscillator phases, the last NT f_dc samples, the channel state
    if (is_est && ke > 0) {
#pragma unroll
        for (int r = 0; r < SPL; r++) Sf_g[lane + 64 * r] = sf[r];
    }
    if (is_worker) {
        if (frames > 0) {
            for (int m = 0; m < M; m++) for (int i = lane; i < NT; i += 64) tail_g[m * NT + i] = s_ring[m * R + ((E_last - (uint32_t)NT + (uint32_t)i) & rmask)];
        }
        if (lane == 0) {
            FskChan &g = a.chan[ch];
            if (frames > 0) {
                // the oscillators where the last frame left them, the estimates it used (still in place: the estimator publishes only frames this launch demodulates)
                for (int m = 0; m < M; m++) { g.phi_c[m] = ctl.phi_end[(frames - 1) & 1][m]; g.f_est[m] = ctl.f_est[(frames - 1) & 3][m]; }
            }
            const bool from_fin = fin_on && frames > 0;
            g.nin = st.nin; g.norm_rx_timing = st.norm_rx_timing; g.ppm = st.ppm;
            g.EbNodB = from_fin ? ctl.fin_state[0] : st.EbNodB; g.snr_est = from_fin ? ctl.fin_state[1] : st.snr_est;
            g.rd = rd0 + E_last; g.samples += (long long)E_last;              // (E_last: the samples the launch's frames took)
            g.frames = (all_done && !gave_up) ? frames : -1;
        }
    }
    (void)Sk;
#undef FW_MARK
}
#endif
