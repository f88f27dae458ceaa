// sonde_fsk_dev.h — structs shared by the 2-/4-FSK modem kernel (sonde_fsk.hip) and its host engine (sonde_fsk.cpp).
#ifndef SONDE_FSK_DEV_H
#define SONDE_FSK_DEV_H
#ifdef SONDE_FSK_EMU                  // tests/emu: the wave form of the modem compiled for the host (test infrastructure) — no HIP headers
#include <stdint.h>
#include <stddef.h>
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
typedef void *hipStream_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#define FSK_THREADS 256

struct FskChan {                      // per-channel struct FSK state (utils/fsk.h:47-95) that survives a modem frame
    float2 phi_c[4];                  // demod local oscillators (M of them)
    int    nin;                       // samples the next frame wants
    float  norm_rx_timing, ppm, EbNodB, snr_est;
    float  f_est[4];                  // estimates used by the last frame
    uint32_t rd;                      // absolute index of the next unread input sample
    long long samples;                // sample_count of fsk_demod.c
    int    frames;                    // frames produced by the current launch
    int    pad;
};

struct FskFrameRec { int nin, nin_next; float f_est[4]; float norm_rx_timing, ppm, EbNodB, snr_est; };

struct FskArgs {
    const void *in;                   // [n_ch][ring] raw samples (int16 / int16x2 / uint8x2)
    int format;                       // SONDE_FSK_*
    int M;                            // tones: 2 or 4 (fsk.h:40, MODE_2FSK / MODE_4FSK)
    int burst;                        // fsk_enable_burst_mode: nin stays N (fsk.c:724)
    int n_ch; uint32_t ring; uint32_t wr;          // absolute write index: samples [rd, wr) are available
    const uint32_t *wr_ch;            // per-channel write index instead of wr (channels fed different counts, sonde_fsk_process_host_var); may be nullptr
    int Fs, Rs, Ts, P, nsym, N, Ndft, log2Ndft, Nmem, NT;   // NT = 2 Ts + Ts/2 tail samples kept per tone
    int st, en, f_zero, len_mask, est_type, fs_tx;
    int n_mask, mask_idx[12];         // positions of the ones in the mask estimator's mask (fsk.c:553-560): 3 per tone
    float tc;
    const float *hann;                // [Ndft]
    const float2 *tw;                 // [Ndft] kiss_fft's twiddles: (cosf, sinf) of (float)(-2 pi k / Ndft) (kiss_fft.c:356-362)
    const uint16_t *perm;             // [Ndft] where input sample i sits before the first butterfly stage (kf_work's decimation)
    const uint16_t *iperm;            // [Ndft] its inverse: the input sample that sits at position o
    int n_stage, st_p[8], st_m[8], st_fs[8];   // butterfly stages in execution order: radix, sub-transform length, twiddle stride
    const float2 *dphi_peak;          // [Ndft]    comp_exp_j(2 pi f/Fs) for f = (k - Ndft/2) Fs/Ndft
    const float2 *dphi_mask;          // [Ndft][M] same for the mask estimator's f2_est
    const float *f_mask;              // [Ndft][M] f2_est values
    const float2 *phi_ft;             // [(nsym+1) P] timing oscillator sequence
    FskChan *chan;                    // [n_ch]
    float *Sf;                        // [n_ch][Ndft]
    float2 *tail;                     // [n_ch][M][NT]
    float *sd; int sd_cap;            // [n_ch][sd_cap] soft decisions of this launch (nsym per frame, 2 nsym for 4-FSK)
    uint8_t *hb;                      // [n_ch][sd_cap] hard bits (rx_bits of fsk_demod, fsk.c:770-778)
    FskFrameRec *recs; int rec_cap;   // [n_ch][rec_cap]
    int max_fft;                      // most FFT blocks a frame can have
    float *eye;                       // [n_ch][8][160] |f_int| samples of the last frame for the eye diagram (fsk.c:857-889; row = trace * M + tone), may be nullptr
    int R;                            // set by the launcher: ring length (samples per tone) of the pipelined kernel, a power of two
    int est_bpw;                      // set by the launcher: transform blocks a wave of the ahead-estimator takes at a time (0 = as many as its lanes hold)
    int est_waves;                    // set by the launcher: waves that estimate the next frame while the oscillator of the current one runs (0..3)
    unsigned long long *prof;         // profiling aid (SONDE_FSK_PROF): [16] shader-clock cycles per phase of channel 0's frames, summed; nullptr = off    // a launch over a LIST of channels with the frame-at-a-time kernel: how the host repeats the channels whose pipeline gave up (sonde_fsk.cpp launch_and_collect)
    const int *ch_list;               // [n_ch] channel of workgroup b (nullptr: b)
    int force_demod;                  // 1: k_fsk_demod even where the pipelined kernel applies
    float *Sf_bak; float2 *tail_bak;  // where the wave form keeps Sf / the tone tails as they were before the launch (for the host's repeat of a channel that gave up); nullptr: the host copies
    int fin;                          // set by the launcher: the wave form's finisher is on (f_int twice in LDS)
    int role_rot;                     // set by the launcher: 0: roles in wavefront order; k > 0: the channel's roles start at wavefront (channel + k - 1) mod waves (sonde_fsk_wave.h)
    int wave_mode;                    // set by the launcher: 0 = k_fsk_stream / k_fsk_demod, 1 = k_fsk_wave one wave per channel, 2 = k_fsk_wave walker + worker (sonde_fsk_wave.h)
    int test_abort_ch;                // test hook (SONDE_FSK_TEST_ABORT=<channel>): that channel's pipeline gives up behind its first frame; -1 = off
    float *scratch; long long scratch_stride;   // frames too long for a CU's LDS (N * (M + 1) beyond ~19000 samples): k_fsk_demod's regions in global memory, scratch_stride floats per workgroup; nullptr: none
};

// atan2 of two floats for the fine-timing angle (fsk.c:705): worked out in double and rounded to float once.  One short dependent chain — the odd series of atan
// on |t| <= tan(pi/8) by Horner's rule, two divisions — instead of the math library's wide evaluation: that one needs some forty vector registers at the spot
// where a channel's whole state is live, which costs the modem kernel a fifth of the channels a CU can hold (sonde_fsk_wave.h).  Error ~2e-16, far below the
// float it is rounded to; the same code on the device and in the emulator (fma and division are IEEE operations on both).
#ifdef SONDE_FSK_EMU
static inline
#else
__host__ __device__ __forceinline__
#endif
float fsk_atan2f(const float yf, const float xf) {
    const double y = (double)yf, x = (double)xf;
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    double r = 0.0;
    if (mx != 0.0) {
        double t = mn / mx;                                     // [0, 1]
        const bool upper = t > 0.41421356237309503;
        if (upper) t = (t - 1.0) / (t + 1.0);                   // atan t = pi/4 + atan((t - 1) / (t + 1)), the argument in [-tan(pi/8), 0]
        const double z = t * t;                                 // <= 0.1716: the term behind the last one kept is below 1e-19
        // (each coefficient is made where it is used: left to itself the compiler keeps all 22 in vector registers across the channel's whole loop — and spills them)
#if defined(SONDE_FSK_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define FSK_AT_STEP(c) p = __builtin_fma(p, z, (c));
#else
#define FSK_AT_STEP(c) { double c_ = (c); asm volatile("" : "+s"(c_)); p = __builtin_fma(p, z, c_); }
#endif
        double p = 1.0 / 45.0;
        FSK_AT_STEP(-1.0 / 43.0)
        FSK_AT_STEP(1.0 / 41.0)
        FSK_AT_STEP(-1.0 / 39.0)
        FSK_AT_STEP(1.0 / 37.0)
        FSK_AT_STEP(-1.0 / 35.0)
        FSK_AT_STEP(1.0 / 33.0)
        FSK_AT_STEP(-1.0 / 31.0)
        FSK_AT_STEP(1.0 / 29.0)
        FSK_AT_STEP(-1.0 / 27.0)
        FSK_AT_STEP(1.0 / 25.0)
        FSK_AT_STEP(-1.0 / 23.0)
        FSK_AT_STEP(1.0 / 21.0)
        FSK_AT_STEP(-1.0 / 19.0)
        FSK_AT_STEP(1.0 / 17.0)
        FSK_AT_STEP(-1.0 / 15.0)
        FSK_AT_STEP(1.0 / 13.0)
        FSK_AT_STEP(-1.0 / 11.0)
        FSK_AT_STEP(1.0 / 9.0)
        FSK_AT_STEP(-1.0 / 7.0)
        FSK_AT_STEP(1.0 / 5.0)
        FSK_AT_STEP(-1.0 / 3.0)
        FSK_AT_STEP(1.0 / 1.0)
#undef FSK_AT_STEP
        r = t * p;
        if (upper) r += 0.78539816339744830962;
        if (ay > ax) r = 1.57079632679489661923 - r;
    }
    if (__builtin_signbit(x)) r = 3.14159265358979323846 - r;
    return __builtin_copysignf((float)r, yf);
}

extern "C" int sonde_launch_fsk(const FskArgs *a, hipStream_t s);
extern "C" int sonde_fsk_wave_selected(const FskArgs *a);      // 1: that launch runs the wave form (which keeps the Sf / tail backups itself)
extern "C" long long sonde_fsk_scratch_floats(const FskArgs *a);      // floats of global scratch per workgroup where a frame does not fit into LDS (k_fsk_demod<M, true>); 0: fits
#endif
