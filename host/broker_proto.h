/*
 * host/broker_proto.h — wire format between the per-sonde shim processes and the resident broker (host/sonde_broker.c).
 *
 * auto_rx starts one decoder pipeline per sonde (auto_rx/autorx/decode.py:1489-1529: `... | fsk_demod ... | rs41mod --softin`); with
 * SONDE_BROKER=<socket path> in the environment the shims of this repo do not open the GPU themselves but hand their samples to one
 * broker process, which owns one HIP context and one batched engine per modem configuration and runs every client's frame in the same
 * launch.  UNIX stream socket, host byte order (same machine), one request -> one reply:
 *
 *   client: HELLO  { brk_hello_t }                          broker: INFO   { sonde_fsk_info_t }        or ERROR { text }
 *   client: DATA   { brk_data_t, n_samples * unit bytes }   broker: RESULT { brk_result_t, Nbits float soft decisions, Nbits hard bits,
 *                                                                            [stats: Ndft float Sf, brk_eye_t] }   or ERROR
 *   close = the slot is free again.
 * DATA carries exactly fsk_nin() samples — the frame loop of utils/fsk_demod.c:281 stays in the client.
 *
 * Decoder shims (BRK_KIND_DEMOD: rs41mod / dfm09mod / m10mod / m20mod on FM audio or IF-rate IQ — the forms auto_rx pipes into them,
 * decode.py:375-417; not the base-rate --IQ form, whose channels share one sample clock inside an engine):
 *   client: HELLO  { brk_hello_demod_t }                    broker: INFO   { sonde_info_t }            or ERROR { text }
 *   client: DATA   { brk_data_t, n_samples * unit bytes }   broker: RESULT { brk_dresult_t, count records of rec_size bytes }
 * A DATA message is one read of the client's input loop (any length up to one second); want_stats bit 1 (BRK_FINISH) marks the end of the
 * stream: the broker flushes the frame in progress on that channel.  Records are what the client's own fetch call would return
 * (sonde_frame_t, sonde_dfm_frame_t, sonde_m10_frame_t, sonde_m20_frame_t — chosen by cfg.sonde_type), with channel = 0.
 */
#ifndef BROKER_PROTO_H
#define BROKER_PROTO_H
#include <stdint.h>
#include "sonde_fsk.h"

#define BRK_MAGIC 0x42444e53u            /* "SNDB" */
enum { BRK_HELLO = 1, BRK_INFO = 2, BRK_DATA = 3, BRK_RESULT = 4, BRK_ERROR = 5 };
enum { BRK_KIND_FSK = 1, BRK_KIND_DEMOD = 2 };
#define BRK_WANT_STATS 1u
#define BRK_FINISH     2u

typedef struct { uint32_t magic, type, length; } brk_hdr_t;      /* length = payload bytes that follow */

typedef struct {
    uint32_t kind;                       /* BRK_KIND_FSK */
    uint32_t reserved;
    sonde_fsk_cfg_t fsk;                 /* n_channels / device / max_chunk are the broker's business and ignored */
} brk_hello_t;

typedef struct { uint32_t n_samples; uint32_t want_stats; } brk_data_t;

typedef struct {
    sonde_fsk_frame_t frame;             /* the frame's record: nin, nin_next, tone estimates, timing, ppm, Eb/N0 */
    int64_t samples;                     /* samples consumed by this stream so far */
    uint32_t nbits, has_stats;
} brk_result_t;

typedef struct { int32_t neyetr, neyesamp; float eye[8 * 160]; } brk_eye_t;

typedef struct {
    uint32_t kind;                       /* BRK_KIND_DEMOD */
    uint32_t reserved;
    sonde_cfg_t cfg;                     /* n_channels / device / max_chunk / max_frames / pipeline are the broker's business and ignored */
    int32_t set_sync, hdmax, bitofs;     /* set_sync != 0: sonde_engine_set_sync(hdmax, bitofs) (the decoders' -d <shift>) */
    int32_t reserved2;
} brk_hello_demod_t;

typedef struct { uint32_t count, rec_size; } brk_dresult_t;

#endif
