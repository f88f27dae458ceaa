// sonde_softin.cpp — soft-bit input framing behind sonde_softin_* (include/sonde_hip.h).
//
// The reference's decoders also accept a stream of float32 soft bits instead of samples (`rs41mod --softin [-i]`, fed by
// `fsk_demod -s`; auto_rx/autorx/decode.py:901-909): find_softbinhead() slides the last 64 soft bits against the +-1
// header, normalised, threshold 0.7 (demod_mod.c:1692-1762); the bit loop of rs41mod.c:2893-2968 then takes 510 x 8 hard
// decisions, de-whitens and hands the frame to print_frame()/rs41_ecc().  This is bit-rate work (4800 b/s per channel) with
// no samples involved: it runs on the host and completes the fsk_demod chain; arithmetic follows the reference exactly
// (double sums of float products), so the frames are bit-identical for identical soft bits.
#include "../../include/sonde_hip.h"
#include "sonde_host.h"
#include <cmath>
#include <cstring>
#include <vector>

using namespace sonde;

struct sonde_softin {
    int ecc_level = 1, inv_in = 0;            // inv_in: --softinv / -i applied to the stream (f32soft_read inv)
    int opt_inv = 0, opt_auto = 0;            // gpx.option.inv / .aut (rs41mod.c:2888-2891)
    float ths = 0.7f;
    float sbuf[64]; int bufpos = -1;
    int state = 0;                            // 0 searching, 1 in frame
    int byte_count = 8, b8pos = 0; uint8_t bitbuf[8];
    uint8_t frame[518];                       // gpx.frame persists across frames like the reference's
    float mv = 0.f; uint64_t bits_in = 0, hdr_bit = 0;
    std::vector<sonde_frame_t> queue;
};

static void emit(sonde_softin *s, int nbytes) {           // print_frame(gpx, byte_count) (rs41mod.c:2472-2490)
    sonde_frame_t f; memset(&f, 0, sizeof f);
    if (nbytes < 518 && nbytes < 0x93) for (int k = nbytes; k < 518; k++) s->frame[k] = 0;
    else if (nbytes < 518) { /* tail keeps the previous frame */ }
    f.channel = 0; f.nbytes = nbytes; f.mv = s->mv; f.mv_pos = (uint32_t)s->hdr_bit;
    f.len = (rs41_frametype(s->frame) >= 0) ? 320 : 518;
    memcpy(f.frame, s->frame, 518);
    f.ecc = s->ecc_level > 0 ? rs41_ecc(f.frame, f.len, s->ecc_level, nullptr) : 0;
    if (s->ecc_level == 0) for (int k = 0; k < 0; k++) {}
    memcpy(s->frame, f.frame, 518);
    s->queue.push_back(f);
}

extern "C" {

int sonde_softin_create(int32_t sonde_type, int32_t ecc_level, int32_t invert_stream, int32_t opt_inv, int32_t opt_auto, sonde_softin_t **out) {
    if (!out || sonde_type != SONDE_RS41) return SONDE_E_ARG;
    sonde_softin *s = new sonde_softin();
    s->ecc_level = ecc_level; s->inv_in = invert_stream ? 1 : 0; s->opt_inv = opt_inv ? 1 : 0; s->opt_auto = opt_auto ? 1 : 0;
    memset(s->sbuf, 0, sizeof s->sbuf); memset(s->frame, 0, sizeof s->frame);
    memcpy(s->frame, kRs41HeaderBytes, 8);
    *out = s;
    return 0;
}

void sonde_softin_destroy(sonde_softin_t *s) { delete s; }

int sonde_softin_push(sonde_softin_t *s, const float *soft, int32_t n) {
    if (!s || (!soft && n > 0) || n < 0) return SONDE_E_ARG;
    for (int32_t i = 0; i < n; i++) {
        float sb = soft[i];
        if (s->inv_in) sb = -sb;
        s->bits_in++;
        if (s->state == 0) {
            // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            s->bufpos = (s->bufpos + 1) % 64;
            s->sbuf[s->bufpos] = sb;
            double sum = 0.0, normx = 0.0, normy = 0.0;
            int j = s->bufpos + 1;
            for (int k = 0; k < 64; k++) {
                if (j >= 64) j = 0;
                const float x = s->sbuf[j];
                const float y = (float)(2.0 * (kRs41Header[k] & 0x1) - 1.0);
                sum += (double)y * (double)s->sbuf[j];
                normx += x * x;
                normy += y * y;
                j++;
            }
            sum /= std::sqrt(normx * normy);
            const float mv = (float)sum;
            if (std::fabs(mv) > s->ths) {
                int found = 1;
                if (mv * (0.5 - s->opt_inv) < 0) { if (!s->opt_auto) found = 0; else s->opt_inv ^= 1; }
                if (found) { s->state = 1; s->byte_count = 8; s->b8pos = 0; s->mv = mv; s->hdr_bit = s->bits_in; }
            }
        } else {
            int bit = sb >= 0.0f;
            if (s->opt_inv) bit ^= 1;
            s->bitbuf[s->b8pos++] = (uint8_t)bit;
            if (s->b8pos == 8) {
                uint8_t byte = 0;
                for (int k = 0; k < 8; k++) byte |= (uint8_t)(s->bitbuf[k] << k);          // bits2byte, LSB first (rs41mod.c:224)
                s->frame[s->byte_count] = byte ^ kRs41Mask[s->byte_count % 64];
                s->b8pos = 0;
                s->byte_count++;
                if (s->byte_count == 518) { emit(s, 518); s->state = 0; }
            }
        }
    }
    return 0;
}

int sonde_softin_finish(sonde_softin_t *s) {               // EOF inside a frame: print_frame with the bytes that exist
    if (!s) return SONDE_E_ARG;
    if (s->state == 1) { emit(s, s->byte_count); s->state = 0; }
    return 0;
}

int sonde_softin_fetch(sonde_softin_t *s, sonde_frame_t *out, int32_t max) {
    if (!s || (!out && max > 0) || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->queue.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->queue[i];
    s->queue.erase(s->queue.begin(), s->queue.begin() + n);
    return n;
}

}  // extern "C"
