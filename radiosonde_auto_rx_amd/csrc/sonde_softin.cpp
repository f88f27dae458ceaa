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
#include <cstdio>
#include <cstring>
#include <vector>

using namespace sonde;

struct sonde_softin {
    int type = SONDE_RS41;
    // DFM (dfm09mod.c:1604-1720): 32 raw header symbols, then two soft symbols per bit (s = s2 - s1), 8 frames of 280 bits per hit
    float dsb[32]; int dpos = 16, dfrm = 0, dhalf = 0; float ds1 = 0.f;
    uint8_t dhb[280]; float dsf[280];
    std::vector<sonde_dfm_frame_t> dqueue;
    int ecc_level = 1, inv_in = 0;            // inv_in: --softinv / -i applied to the stream (f32soft_read inv)
    int opt_inv = 0, opt_auto = 0;            // gpx.option.inv / .aut (rs41mod.c:2888-2891)
    unsigned char hexbyte = 0;                // frmbyte of the --rawhex reader (:2980): keeps its value over pairs that are not hex
    float ths = 0.7f;
    float sbuf[64]; int bufpos = -1;
    char hbuf[64];                            // --bin: last header-length hard bits as '0'/'1' (hdb.buf, demod_mod.c:1668-1690)
    int state = 0;                            // 0 searching, 1 in frame
    int byte_count = 8, b8pos = 0; uint8_t bitbuf[8];
    uint8_t frame[518];                       // gpx.frame persists across frames like the reference's
    float mv = 0.f; uint64_t bits_in = 0, hdr_bit = 0;
    char rawbuf[280 / 4 + 12 + 1] = {}; int rawpos = 0; float rawcnt = -1.0f;      // --rawhex reader of dfm09mod (:1732-1740)
    uint32_t hdrcnt = 0;                      // DFM: 8 per header seen (dfm09mod.c:1628,1632), base of the frame time stamp
    std::vector<sonde_frame_t> queue;
    // --ecc3 / --ecc4 behind soft input: the soft value of every frame bit travels with the frame (rs41mod.c:2910-2916,2941)
    float fsoft[4080]; int nsoft = 0;
    std::vector<std::vector<float>> qsoft, last_soft;
    // M10 / M20 (m10mod.c:1405-1510): 32-symbol header at threshold 0.8, two soft symbols per bit (s2 - s1), differential decoding,
    // then ONE symbol per counted bit is dropped until 5 x 808 (the reference's skip loop reads a single float per step)
    int mpos = 0, mhalf = 0, mbit0 = '0', mskip = 0, mdoskip = 1; float ms1 = 0.f;
    char mbits[(101 + 64) * 8 + 8];
    std::vector<sonde_m10_frame_t> q10; std::vector<sonde_m20_frame_t> q20;
};

static void mxx_emit(sonde_softin *s, int pos) {
    const bool m20 = s->type == SONDE_M20;
    const int nb = m20 ? 101 + 64 : 101 + 20;
    uint8_t fr[172]; memset(fr, 0, sizeof fr);
    s->mbits[pos] = 0;
    for (int i = 0; i < nb; i++) { int v = 0; for (int k = 0; k < 8; k++) if (s->mbits[8 * i + 7 - k] == '1') v |= 1 << k; fr[i] = (uint8_t)v; }
    if (!m20) {
        sonde_m10_frame_t o; memset(&o, 0, sizeof o);
        memcpy(o.frame, fr, 121);
        o.nbits = pos; o.mv = s->mv; o.mv_pos = (uint32_t)s->hdr_bit;
        sonde_m10_frame_finish(&o);
        s->q10.push_back(o);
    } else {
        sonde_m20_frame_t o; memset(&o, 0, sizeof o);
        memcpy(o.frame, fr, 165);
        o.nbits = pos; o.mv = s->mv; o.mv_pos = (uint32_t)s->hdr_bit;
        sonde_m20_frame_finish(&o);
        s->q20.push_back(o);
    }
}

static void emit(sonde_softin *s, int nbytes) {           // print_frame(gpx, byte_count) (rs41mod.c:2472-2490)
    sonde_frame_t f; memset(&f, 0, sizeof f);
    if (nbytes < 518 && nbytes < 0x93) for (int k = nbytes; k < 518; k++) s->frame[k] = 0;
    else if (nbytes < 518) { /* tail keeps the previous frame */ }
    f.channel = 0; f.nbytes = nbytes; f.mv = s->mv; f.mv_pos = (uint32_t)s->hdr_bit;
    f.len = (rs41_frametype(s->frame) >= 0) ? 320 : 518;
    memcpy(f.frame, s->frame, 518);
    if (s->ecc_level >= 3) {                               // list decoding needs the decoder's state: sonde_rs41_dec_ecc() by the caller
        f.ecc = 0;
        s->qsoft.emplace_back(s->fsoft, s->fsoft + s->nsoft);
        s->qsoft.back().push_back((float)s->opt_inv);      // last element: polarity in effect for this frame
    } else {
        f.ecc = s->ecc_level > 0 ? rs41_ecc(f.frame, f.len, s->ecc_level, nullptr) : 0;
        memcpy(s->frame, f.frame, 518);
    }
    s->queue.push_back(f);
}

extern "C" {

int sonde_softin_create(int32_t sonde_type, int32_t ecc_level, int32_t invert_stream, int32_t opt_inv, int32_t opt_auto, sonde_softin_t **out) {
    if (!out || (sonde_type != SONDE_RS41 && sonde_type != SONDE_DFM09 && sonde_type != SONDE_M10 && sonde_type != SONDE_M20)) return SONDE_E_ARG;
    sonde_softin *s = new sonde_softin();
    s->type = sonde_type;
    memset(s->dsb, 0, sizeof s->dsb); memset(s->dhb, 0, sizeof s->dhb); memset(s->dsf, 0, sizeof s->dsf);
    { static const char kDfmHdr[] = "0100010111001111"; for (int i = 0; i < 16; i++) s->dhb[i] = (uint8_t)(kDfmHdr[i] & 1); }      // dfm09mod.c:1503-1506
    s->ecc_level = ecc_level; s->inv_in = invert_stream ? 1 : 0; s->opt_inv = opt_inv ? 1 : 0; s->opt_auto = opt_auto ? 1 : 0;
    memset(s->sbuf, 0, sizeof s->sbuf); memset(s->frame, 0, sizeof s->frame); memset(s->hbuf, 0, sizeof s->hbuf);
    memcpy(s->frame, kRs41HeaderBytes, 8);
    memset(s->mbits, 0, sizeof s->mbits);
    if (sonde_type == SONDE_M10 || sonde_type == SONDE_M20) s->ths = 0.8f;
    *out = s;
    return 0;
}

void sonde_softin_destroy(sonde_softin_t *s) { delete s; }

int sonde_softin_push(sonde_softin_t *s, const float *soft, int32_t n) {
    if (!s || (!soft && n > 0) || n < 0) return SONDE_E_ARG;
    if (s->type == SONDE_M10 || s->type == SONDE_M20) {
        const int nbits = (s->type == SONDE_M20 ? 101 + 64 : 101 + 20) * 8;
        for (int32_t i = 0; i < n; i++) {
            float sb = soft[i];
            if (s->inv_in) sb = -sb;
            s->bits_in++;
            if (s->state == 0) {                                   // header search on the symbol stream
                s->bufpos = (s->bufpos + 1) % 32;
                s->dsb[s->bufpos] = sb;
                double sum = 0.0, normx = 0.0, normy = 0.0;
                int j = s->bufpos + 1;
                for (int k = 0; k < 32; k++) {
                    if (j >= 32) j = 0;
                    const float x = s->dsb[j];
                    const float y = (float)(2.0 * (kM10RawHeader[k] & 0x1) - 1.0);
                    sum += (double)y * (double)s->dsb[j];
                    normx += x * x; normy += y * y;
                    j++;
                }
                sum /= std::sqrt(normx * normy);
                const float mv = (float)sum;
                if (std::fabs(mv) > s->ths) {
                    if (mv * (0.5 - s->opt_inv) < 0) s->opt_inv ^= 1;          // irrelevant for the differential code (m10mod.c:1447)
                    s->state = 1; s->mpos = 0; s->mhalf = 0; s->mbit0 = '0'; s->mv = mv; s->hdr_bit = s->bits_in;
                }
            } else if (s->state == 1) {                            // frame bits: two symbols each
                if (!s->mhalf) { s->ms1 = sb; s->mhalf = 1; continue; }
                s->mhalf = 0;
                const int bit = (sb - s->ms1) >= 0.0f;
                s->mbits[s->mpos++] = (char)(0x31 ^ (s->mbit0 ^ bit));
                s->mbit0 = bit;
                if (s->mpos == nbits) { mxx_emit(s, s->mpos); s->state = s->mdoskip ? 2 : 0; s->mskip = nbits; }
            } else {                                               // rest of the second: one symbol per counted bit up to 5 x 808
                if (++s->mskip >= 5 * 808) s->state = 0;
            }
        }
        return 0;
    }
    if (s->type == SONDE_DFM09) {
        for (int32_t i = 0; i < n; i++) {
            float sb = soft[i];
            if (s->inv_in) sb = -sb;
            s->bits_in++;
            if (s->state == 0) {
                s->bufpos = (s->bufpos + 1) % 32;
                s->dsb[s->bufpos] = sb;
                double sum = 0.0, normx = 0.0, normy = 0.0;
                int j = s->bufpos + 1;
                for (int k = 0; k < 32; k++) {
                    if (j >= 32) j = 0;
                    const float x = s->dsb[j];
                    const float y = (float)(2.0 * (kDfmRawHeader[k] & 0x1) - 1.0);
                    sum += (double)y * (double)s->dsb[j];
                    normx += x * x;
                    normy += y * y;
                    j++;
                }
                sum /= std::sqrt(normx * normy);
                const float mv = (float)sum;
                if (std::fabs(mv) > s->ths) {
                    int found = 1;
                    s->hdrcnt += 8;
                    if (mv * (0.5 - s->opt_inv) < 0) { if (!s->opt_auto) found = 0; else s->opt_inv ^= 1; }
                    if (found) { s->state = 1; s->dpos = 16; s->dfrm = 0; s->dhalf = 0; s->mv = mv; s->hdr_bit = s->bits_in; }
                }
            } else {
                if (!s->dhalf) { s->ds1 = sb; s->dhalf = 1; continue; }
                s->dhalf = 0;
                float v = sb - s->ds1;                          // integrate both Manchester symbols (dfm09mod.c:1684)
                int hb = v >= 0.0f;
                if (s->opt_inv) { hb ^= 1; v = -v; }
                s->dhb[s->dpos] = (uint8_t)hb; s->dsf[s->dpos] = v;
                if (++s->dpos == 280) {
                    sonde_dfm_frame_t o; memset(&o, 0, sizeof o);
                    o.channel = 0; o.frame_in_hit = s->dfrm; o.mv = s->mv; o.mv_pos = (uint32_t)s->hdr_bit;
                    o.frm_count = (float)(s->hdrcnt + (uint32_t)s->dfrm); o.inv = s->opt_inv;
                    for (int i = 0; i < 280; i++) o.rawbits[i >> 3] |= (uint8_t)((s->dhb[i] & 1) << (i & 7));
                    o.ecc[0] = dfm_block(s->ecc_level, s->dhb + 16, s->dsf + 16, 7, o.conf);
                    o.ecc[1] = dfm_block(s->ecc_level, s->dhb + 72, s->dsf + 72, 13, o.dat1);
                    o.ecc[2] = dfm_block(s->ecc_level, s->dhb + 176, s->dsf + 176, 13, o.dat2);
                    s->dqueue.push_back(o);
                    s->dpos = 0;
                    if (++s->dfrm == 8) s->state = 0;           // nfrms frames per header hit, then search again (:1656,1718)
                }
            }
        }
        return 0;
    }
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
                if (found) { s->state = 1; s->byte_count = 8; s->b8pos = 0; s->mv = mv; s->hdr_bit = s->bits_in; s->nsoft = 0; }
            }
        } else {
            int bit = sb >= 0.0f;
            if (s->nsoft < 4080) s->fsoft[s->nsoft++] = sb;
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

// cmp_hdb (demod_mod.c:1639-1666): header bit errors of the circular buffer in both polarities, as a +-score
static float hdr_bit_score(const char *buf, int bufpos, const char *hdr, int len) {
    int e1 = 0, e2 = 0;
    for (int i = 0, j = bufpos; i < len; i++, j--) {
        if (j < 0) j = len - 1;
        if (buf[j] != hdr[len - 1 - i]) e1++;
        if ((buf[j] ^ 0x01) != hdr[len - 1 - i]) e2++;
    }
    return e2 < e1 ? (float)(-len + e2) / (float)len : (float)(len - e1) / (float)len;
}

// --bin: one byte per hard bit (`fsk_demod` without -s), find_binhead + the bit loops (rs41mod.c:2875,2899-2906; dfm09mod.c:1665-1676)
int sonde_softin_push_bits(sonde_softin_t *s, const uint8_t *bits, int32_t n) {
    if (!s || (!bits && n > 0) || n < 0) return SONDE_E_ARG;
    const bool dfm = s->type == SONDE_DFM09;
    const int hl = dfm ? 32 : 64;
    const char *hdr = dfm ? kDfmRawHeader : kRs41Header;
    const float thb = (float)(1.0 - (dfm ? 2.1 : 3.1) / (float)hl);
    for (int32_t i = 0; i < n; i++) {
        const int b = bits[i] & 1;
        s->bits_in++;
        if (s->state == 0) {
            s->bufpos = (s->bufpos + 1) % hl;
            s->hbuf[s->bufpos] = (char)(0x30 | b);
            const float mv = hdr_bit_score(s->hbuf, s->bufpos, hdr, hl);
            if (std::fabs(mv) > thb) {
                int found = 1;
                s->hdrcnt += 8;
                if (mv * (0.5 - s->opt_inv) < 0) { if (!s->opt_auto) found = 0; else s->opt_inv ^= 1; }
                if (found) {
                    s->state = 1; s->mv = mv; s->hdr_bit = s->bits_in;
                    s->byte_count = 8; s->b8pos = 0;
                    s->dpos = 16; s->dfrm = 0; s->dhalf = 0;
                }
            }
        } else if (dfm) {
            if (!s->dhalf) { s->dhalf = 1; continue; }              // first Manchester symbol is read and dropped
            s->dhalf = 0;
            int hb = b; float v = (float)(2 * hb - 1);
            if (s->opt_inv) { hb ^= 1; v = -v; }
            s->dhb[s->dpos] = (uint8_t)hb; s->dsf[s->dpos] = v;
            if (++s->dpos == 280) {
                sonde_dfm_frame_t o; memset(&o, 0, sizeof o);
                o.channel = 0; o.frame_in_hit = s->dfrm; o.mv = s->mv; o.mv_pos = (uint32_t)s->hdr_bit;
                o.frm_count = (float)(s->hdrcnt + (uint32_t)s->dfrm); o.inv = s->opt_inv;
                for (int i = 0; i < 280; i++) o.rawbits[i >> 3] |= (uint8_t)((s->dhb[i] & 1) << (i & 7));
                o.ecc[0] = dfm_block(s->ecc_level, s->dhb + 16, s->dsf + 16, 7, o.conf);
                o.ecc[1] = dfm_block(s->ecc_level, s->dhb + 72, s->dsf + 72, 13, o.dat1);
                o.ecc[2] = dfm_block(s->ecc_level, s->dhb + 176, s->dsf + 176, 13, o.dat2);
                s->dqueue.push_back(o);
                s->dpos = 0;
                if (++s->dfrm == 8) s->state = 0;
            }
        } else {
            int bit = b;
            if (s->opt_inv) bit ^= 1;
            s->bitbuf[s->b8pos++] = (uint8_t)bit;
            if (s->b8pos == 8) {
                uint8_t byte = 0;
                for (int k = 0; k < 8; k++) byte |= (uint8_t)(s->bitbuf[k] << k);
                s->frame[s->byte_count] = byte ^ kRs41Mask[s->byte_count % 64];
                s->b8pos = 0;
                s->byte_count++;
                if (s->byte_count == 518) { emit(s, 518); s->state = 0; }
            }
        }
    }
    return 0;
}

// --rawhex / --xorhex (rs41mod.c:2976-3002): frame bytes given directly; gpx.frame persists between lines
int sonde_softin_push_frame(sonde_softin_t *s, const uint8_t *bytes, int32_t len, int32_t xorhex) {
    if (!s || !bytes || len < 0 || len > 518 || s->type != SONDE_RS41) return SONDE_E_ARG;
    for (int i = 0; i < len; i++) s->frame[i] = xorhex ? (uint8_t)(bytes[i] ^ kRs41Mask[i % 64]) : bytes[i];
    s->mv = 0.f; s->hdr_bit = 0;
    emit(s, len);
    return 0;
}

int sonde_softin_push_hexline(sonde_softin_t *s, const char *line, int32_t xorhex) {
    if (!s || !line || s->type != SONDE_RS41) return SONDE_E_ARG;
    static thread_local char buf[2 * 518 + 12];
    strncpy(buf, line, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    buf[2 * 518] = '\0';
    char *sp = strchr(buf, ' ');
    if (sp) *sp = '\0';
    const int len = (int)strlen(buf) / 2;
    if (len <= 0x3D + 10) return 0;
    for (int i = 0; i < len; i++) {
        sscanf(buf + 2 * i, "%2hhx", &s->hexbyte);
        if (xorhex) s->hexbyte ^= kRs41Mask[i % 64];
        s->frame[i] = s->hexbyte;
    }
    s->mv = 0.f; s->hdr_bit = 0;
    emit(s, len);
    return 1;
}

int sonde_softin_finish(sonde_softin_t *s) {               // EOF inside a frame: print_frame with the bytes that exist
    if (!s) return SONDE_E_ARG;
    if (s->type == SONDE_DFM09) { s->state = 0; return 0; }          // a partial DFM frame is dropped (dfm09mod.c:1702,1713)
    if (s->type == SONDE_M10 || s->type == SONDE_M20) {               // EOF inside a frame: printed with the bits that exist (m10mod.c:1486-1490)
        if (s->state == 1) mxx_emit(s, s->mpos);
        s->state = 0;
        return 0;
    }
    if (s->state == 1) { emit(s, s->byte_count); s->state = 0; }
    return 0;
}

int sonde_softin_set_m10_skip(sonde_softin_t *s, int32_t skip) { if (!s) return SONDE_E_ARG; s->mdoskip = skip != 0; return 0; }

int sonde_softin_fetch_m10(sonde_softin_t *s, sonde_m10_frame_t *out, int32_t max) {
    if (!s || (!out && max > 0) || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->q10.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->q10[i];
    s->q10.erase(s->q10.begin(), s->q10.begin() + n);
    return n;
}
int sonde_softin_fetch_m20(sonde_softin_t *s, sonde_m20_frame_t *out, int32_t max) {
    if (!s || (!out && max > 0) || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->q20.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->q20[i];
    s->q20.erase(s->q20.begin(), s->q20.begin() + n);
    return n;
}

int sonde_softin_push_dfm_rawhex(sonde_softin_t *s, const char *text, int32_t n) {
    if (!s || n < 0 || (n > 0 && !text) || s->type != SONDE_DFM09) return SONDE_E_ARG;
    const int BUFLEN = 280 / 4 + 12;
    for (int t = 0; t < n; t++) {
        const int ch = (unsigned char)text[t];
        if (ch == ' ') continue;
        if (ch != '\n') {
            const bool keep = (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'f') || (ch >= 'A' && ch <= 'F') || ch == '+' || ch == '-' || ch == '<' || ch == '.' || ch == '>';
            if (keep && s->rawpos < BUFLEN) s->rawbuf[s->rawpos++] = (char)ch;
            continue;
        }
        s->rawbuf[s->rawpos] = '\0';
        s->opt_inv = s->rawbuf[0] == '-' ? 1 : 0;
        sscanf(s->rawbuf + 1, "<%f>", &s->rawcnt);
        const char *p = strchr(s->rawbuf, '>');
        if (p) {
            p++;
            const int len = (int)strlen(p);
            if (len * 4 == 280 - 16) {
                for (int i = 0; i < len; i++) {
                    unsigned char nib = 0xFF;
                    sscanf(p + i, "%1hhx", &nib);
                    for (int j = 0; j < 4; j++) { s->dhb[16 + 4 * i + j] = (nib >> j) & 1; s->dsf[16 + 4 * i + j] = (float)(2 * s->dhb[16 + 4 * i + j] - 1); }
                }
                sonde_dfm_frame_t o; memset(&o, 0, sizeof o);
                o.frame_in_hit = 1; o.frm_count = s->rawcnt; o.inv = s->opt_inv;
                for (int i = 0; i < 280; i++) o.rawbits[i >> 3] |= (uint8_t)((s->dhb[i] & 1) << (i & 7));
                o.ecc[0] = dfm_block(s->ecc_level, s->dhb + 16, s->dsf + 16, 7, o.conf);
                o.ecc[1] = dfm_block(s->ecc_level, s->dhb + 72, s->dsf + 72, 13, o.dat1);
                o.ecc[2] = dfm_block(s->ecc_level, s->dhb + 176, s->dsf + 176, 13, o.dat2);
                s->dqueue.push_back(o);
            }
        }
        s->rawpos = 0;
    }
    return 0;
}

int sonde_softin_fetch_dfm(sonde_softin_t *s, sonde_dfm_frame_t *out, int32_t max) {
    if (!s || (!out && max > 0) || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->dqueue.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->dqueue[i];
    s->dqueue.erase(s->dqueue.begin(), s->dqueue.begin() + n);
    return n;
}

int sonde_softin_fetch(sonde_softin_t *s, sonde_frame_t *out, int32_t max) {
    if (!s || (!out && max > 0) || max < 0) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->queue.size(), (size_t)max);
    for (int i = 0; i < n; i++) out[i] = s->queue[i];
    s->queue.erase(s->queue.begin(), s->queue.begin() + n);
    if (s->ecc_level >= 3) {
        s->last_soft.assign(s->qsoft.begin(), s->qsoft.begin() + std::min<size_t>((size_t)n, s->qsoft.size()));
        s->qsoft.erase(s->qsoft.begin(), s->qsoft.begin() + (long)s->last_soft.size());
    }
    return n;
}

int sonde_softin_fetch_soft(sonde_softin_t *s, float *soft, int32_t *nbits, int32_t *inv, int32_t max) {
    if (!s || max < 0 || (max > 0 && (!soft || !nbits || !inv))) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->last_soft.size(), (size_t)max);
    for (int i = 0; i < n; i++) {
        const std::vector<float> &v = s->last_soft[i];
        nbits[i] = (int32_t)v.size() - 1; inv[i] = v.back() != 0.f;
        memcpy(soft + (size_t)i * 4080, v.data(), (v.size() - 1) * sizeof(float));
    }
    return n;
}

}  // extern "C"
