/*
 * sonde_chan.h — C ABI of the polyphase channelizer in libsonde_hip.so.
 *
 * One wideband cs16 stream -> M uniformly spaced channels at the stream rate / D, in ONE pass over the stream.  The reference has
 * no channelizer: for a wideband source auto_rx starts one detector / decoder process per frequency, and each of them mixes its
 * own carrier out of the stream and low-pass filters it (demod/mod/demod_mod.c:1224-1249 decimator design, :737-754 mixer + FIR;
 * scan/dft_detect.c:737-760) — M passes over the same samples.  This replaces those M front ends for BASELINE.json configs[2]
 * ("256-channel polyphase channelize of 10 MHz wideband IQ"); what leaves it is IF-rate IQ in the form `dft_detect --iq` and the
 * decoders' `--iq2 / --iq3` input forms read (float32 IQ, one stream per channel), so the reference's own tools can be put behind
 * it unchanged (tests/test_gpu_chan.py does that).
 *
 *   y_k[m] = sum_n h[n] x[m D - n] exp(-2 pi i k (m D - n) / M),  k = 0 .. M-1 (k >= M/2: negative frequencies), centre k Fs / M
 *
 * h: Blackman-windowed sinc of M * P taps, -6 dB at half the channel spacing.  Conventions as sonde_hip.h.
 */
#ifndef SONDE_CHAN_H
#define SONDE_CHAN_H

#include "sonde_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sonde_chan sonde_chan_t;

typedef struct {
    int32_t abi_version;     /* SONDE_ABI_VERSION                                                       */
    int32_t device;
    int32_t sample_rate;     /* of the wideband stream                                                  */
    int32_t M;               /* channels: power of two, 16 .. 1024                                      */
    int32_t D;               /* decimation: output rate = sample_rate / D, D <= M                       */
    int32_t P;               /* prototype taps per branch (filter length M * P), 4 .. 32                */
    int32_t max_chunk;       /* largest n_samples per process call                                      */
    int32_t reserved[5];
} sonde_chan_cfg_t;

typedef struct {
    int32_t out_rate_num, out_rate_den;   /* output sample rate = num / den                             */
    int32_t taps;
    int32_t max_frames;      /* most output samples per channel one call can produce                    */
    float   spacing_hz;      /* channel k is centred at k * spacing_hz (k >= M/2: (k - M) * spacing_hz)  */
    int32_t reserved[3];
} sonde_chan_info_t;

int  sonde_chan_create(const sonde_chan_cfg_t *cfg, sonde_chan_t **out);
void sonde_chan_destroy(sonde_chan_t *c);
int  sonde_chan_info(const sonde_chan_t *c, sonde_chan_info_t *info);
/* n_samples complex int16 samples of the stream (device / host memory) -> d_out[k * out_stride + j] (complex float32, device),
 * j = 0 .. returned count - 1: the output samples this call completes (the filter state carries over between calls).
 * Returns the count (>= 0) or SONDE_E_*.  Work is queued on the channelizer's stream; sonde_chan_sync() waits for it.  The stream does not
 * wait for any other stream: whatever prepared d_iq / d_out elsewhere (an allocator's fill, a copy) must have completed before the call. */
int  sonde_chan_process_device(sonde_chan_t *c, const void *d_iq, int32_t n_samples, void *d_out, int64_t out_stride);
int  sonde_chan_process_host(sonde_chan_t *c, const void *h_iq, int32_t n_samples, void *d_out, int64_t out_stride);
int  sonde_chan_sync(sonde_chan_t *c);
void *sonde_chan_stream(sonde_chan_t *c);
int  sonde_chan_kernel_ms(sonde_chan_t *c, double *avg_ms, int64_t *launches);
/* For callers without a device allocator of their own (host/sonde_wideband.c --channelize; the Python receiver uses torch tensors):
 *   sonde_chan_output      an output array [M][max_frames] complex float32 owned by the channelizer (freed with it)
 *   sonde_chan_rows_alloc  a zeroed array [n_rows][max_frames] complex float32 for a decoder engine's channels (freed with the channelizer)
 *   sonde_chan_gather      rows r = 0 .. n_rows-1 of d_rows <- the first n_frames samples of channel channels[r] of d_out (channels[r] < 0: row left
 *                          alone); queued on the channelizer's stream, i.e. behind the process call that produced them — sonde_chan_sync() before an
 *                          engine on another stream reads d_rows.  This is what auto_rx does per sonde with one rtl_fm / ss_iq process each
 *                          (sdr_wrappers.py:270-371): here it is a row copy. */
int  sonde_chan_output(sonde_chan_t *c, void **d_out, int64_t *out_stride);
int  sonde_chan_rows_alloc(sonde_chan_t *c, int32_t n_rows, void **d_rows);
int  sonde_chan_gather(sonde_chan_t *c, const void *d_out, int64_t out_stride, const int32_t *channels, int32_t n_rows, int32_t n_frames, void *d_rows);

#ifdef __cplusplus
}
#endif
#endif
