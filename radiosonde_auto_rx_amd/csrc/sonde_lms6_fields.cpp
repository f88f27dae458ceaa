// sonde_lms6_fields.cpp — LMS6-403 / LMS-X blocks -> frames -> the reference's text / JSON (include/sonde_lms6.h).  Host code, bit rate.
//
// One object = what demod/mod/lms6Xmod.c keeps in gpx_t plus the little state of its main loop.  Everything below a header hit is here:
//   raw bits of a block (sign alternation of the (c0, inv(c1)) pairs, --ecc3 merge of the two soft values)        lms6Xmod.c:1376-1424
//   Viterbi over the rate-1/2 K = 7 code (hard or soft metric) or the algebraic inverse                             :232-374
//   bytes (LSB first), RS(255,223) through sonde_ecc.h, frame sync 24 54 00 00|05 (LMS6) / 24 46 05 00 (LMS-X)     :415-441,:800-989
//   CRC-16 (poly 0x1021, init 0), fields, text line, JSON                                                           :376-412,:464-798
//   LMS6 <-> LMS-X auto detection                                                                                   :893-903,:931-963,:1434-1462
//   soft-bit input: header search by normalised correlation over the last 64 soft bits                              demod_mod.c:1692-1762
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_ecc.h"
#include "../../include/sonde_lms6.h"

namespace {

constexpr int BITS = 8, SYNC_LEN = 5, FRM_LEN = 223, BLOCKSTART = SYNC_LEN * BITS * 2, BLOCK_LEN = FRM_LEN + 32 + SYNC_LEN;
constexpr int RAWBLK = 300 * BITS * 2, RAWBLK6 = (BLOCK_LEN + 1) * BITS * 2, FRAME_LEN = 300, BITFRAME_LEN = FRAME_LEN * BITS, OVERLAP = 64;
constexpr int RAWBITFRAME_LEN = BITFRAME_LEN * 2;
constexpr int KL = 7, NST = 1 << KL, MST = 1 << (KL - 1);          // constraint length, code words, trellis states
constexpr int OFS = 4, P_SN = OFS, P_FRNB = OFS + 4, P_TOW = OFS + 6, P_LAT = OFS + 0x0E, P_LON = OFS + 0x12, P_ALT = OFS + 0x16;
constexpr int P_VE = OFS + 0x1A, P_VN = OFS + 0x1D, P_VU = OFS + 0x20, P_VH = OFS + 0x1A, P_VD = OFS + 0x1C, P_VV = OFS + 0x1E;

const char kRawHeader[] = "0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110";     // (c0, inv(c1)) of 58 f3 3f b8
const char kBlkSync[] = "0000000000000000" "0000001101011101" "0100100111000010" "0100111111110010" "0110100001101011";
const uint8_t kSync6[4] = { 0x24, 0x54, 0x00, 0x00 }, kSyncX[4] = { 0x24, 0x46, 0x05, 0x00 };
const char kPolyA[] = "1001111", kPolyB[] = "1101101";
const char kDay[7][4] = { "Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat" };

struct HS { uint8_t hb; float sb; };
struct St { uint8_t bIn, codeIn, prevState; float w; };

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[512]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

int crc16_0(const uint8_t *p, int n) {
    int rem = 0;
    for (int i = 0; i < n; i++) {
        rem ^= p[i] << 8;
        for (int j = 0; j < 8; j++) rem = (rem & 0x8000) ? ((rem << 1) ^ 0x1021) & 0xFFFF : (rem << 1) & 0xFFFF;
    }
    return rem;
}

}  // namespace

struct sonde_lms6_dec {
    sonde_lms6_opts_t o{};
    sonde_ecc_t *rs = nullptr;
    // gpx_t
    int frnr = 0, sn = 0, week = 0, gpstow = 0, gpssec = 0, jahr = 0, monat = 0, tag = 0, wday = 0, std_ = 0, min_ = 0;
    double gpstowX = 0; float sek = 0;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    std::vector<HS> blk;                       // blk_rawbits
    uint8_t frame[FRM_LEN];
    int frm_pos = 0, sf6 = 0, sfX = 0, typ = 6, auto_detect = 1, reset_dsp = 0;
    float frm_rate = 0.f;
    int gpstow_start = -1; double time_elapsed = 0.0;
    // Viterbi
    uint8_t code[NST];
    std::vector<HS> vraw; std::vector<St> vstate; St vd[NST];
    // main loop
    int rawblk_len = RAWBLK6, rate_changed = 0;
    // soft input framer
    float sbuf[64]; int bufpos = -1; int in_block = 0, pos = 0; unsigned bc = 0;

    St &S(int t, int j) { return vstate[(size_t)t * MST + j]; }

    // ---- Viterbi (:232-341) ------------------------------------------------------------------------------------------------
    static float dist2(int c, const HS *rc) {
        const int c0 = 2 * ((c >> 1) & 1) - 1, c1 = 2 * (c & 1) - 1;
        return (c0 - rc[0].sb) * (c0 - rc[0].sb) + (c1 - rc[1].sb) * (c1 - rc[1].sb);
    }
    static int hbstr_len(const HS *h) { int n = 0; while (h[n].hb) n++; return n; }
    void viterbi(const HS *rc) {
        int t = KL - 1, m = MST;
        while (t > 0) { for (int j = 0; j < m; j++) S(t, j).prevState = (uint8_t)(j / 2); t--; m /= 2; }
        m = 2;
        for (t = 1; t < KL; t++) {
            for (int j = 0; j < m; j++) {
                const int c = code[j];
                S(t, j).bIn = (uint8_t)(j % 2); S(t, j).codeIn = (uint8_t)c;
                S(t, j).w = S(t - 1, S(t, j).prevState).w + dist2(c, rc + 2 * (t - 1));
            }
            m *= 2;
        }
        const int tmax = hbstr_len(rc) / 2;
        for (t = KL - 1; t < tmax; t++) {
            for (int j = 0; j < MST; j++)
                for (int b = 0; b < 2; b++) {
                    const int ns = j * 2 + b;
                    vd[ns].bIn = (uint8_t)b; vd[ns].codeIn = code[ns]; vd[ns].prevState = (uint8_t)j;
                    vd[ns].w = S(t, j).w + dist2(vd[ns].codeIn, rc + 2 * t);
                }
            for (int j = 0; j < MST; j++) S(t + 1, j) = (vd[j].w <= vd[j + MST].w) ? vd[j] : vd[j + MST];
        }
        float w_min = -1; int j_min = 0;
        for (int j = 0; j < MST; j++) {
            if (w_min < 0) { w_min = S(tmax, j).w; j_min = j; }
            if (S(tmax, j).w < w_min) { w_min = S(tmax, j).w; j_min = j; }
        }
        int j = j_min; t = tmax;
        vraw[2 * t].hb = 0;
        while (t > 0) {
            const int c = S(t, j).codeIn;
            vraw[2 * t - 2].hb = (uint8_t)(0x30 + ((c >> 1) & 1));
            vraw[2 * t - 1].hb = (uint8_t)(0x30 + (c & 1));
            j = S(t, j).prevState;
            t--;
        }
    }

    // ---- algebraic inverse of the code on hard bits (:343-374) ------------------------------------------------------------------
    static int deconv(const HS *raw, char *bits) {
        const int len = hbstr_len(raw), m = KL - 1;
        int errors = 0, n = 0;
        for (int j = 0; j < m; j++) bits[j] = '0';
        while (2 * (m + n) < len) {
            const HS *p = raw + 2 * (m + n);
            int bitA = 0, bitB = 0;
            for (int j = 0; j < m; j++) { bitA ^= (bits[n + j] & 1) & (kPolyA[j] & 1); bitB ^= (bits[n + j] & 1) & (kPolyB[j] & 1); }
            const int a = bitA ^ (p[0].hb & 1), b = bitB ^ (p[1].hb & 1);
            if (a == (kPolyA[m] & 1) && b == (kPolyB[m] & 1)) bits[n + m] = '1';
            else if (a == 0 && b == 0) bits[n + m] = '0';
            else { bits[n + m] = (a != (kPolyA[m] & 1) && b == (kPolyB[m] & 1)) ? 0x39 : 0x38; errors = n; break; }
            n += 1;
        }
        bits[n + m] = 0;
        return errors;
    }

    // ---- fields (:464-697) -----------------------------------------------------------------------------------------------------
    static int be(const uint8_t *p, int n) { unsigned v = 0; for (int i = 0; i < n; i++) v |= (unsigned)p[i] << (8 * (n - 1 - i)); return (int)v; }
    void gps2date() {
        const long GpsDays = (long)week * 7 + (gpssec / 86400), Mjd = 44244 + GpsDays;
        long J = Mjd + 2468570; const long Cc = 4 * J / 146097;
        J = J - (146097 * Cc + 3) / 4;
        const long Y = 4000 * (J + 1) / 1461001;
        J = J - 1461 * Y / 4 + 31;
        const long Mo = 80 * J / 2447;
        tag = (int)(J - 2447 * Mo / 80);
        J = Mo / 11;
        monat = (int)(Mo + 2 - (12 * J));
        jahr = (int)(100 * (Cc - 49) + Y + J);
    }
    int gps_time6(int crc_err) {
        int t = be(frame + P_TOW, 4);
        if (gpstow_start < 0 && !crc_err) {
            gpstow_start = t;
            if (week > 0 && t / 1000.0 < time_elapsed) week += 1;
        }
        gpstow = t;
        const float ms = (float)(t % 1000);
        t /= 1000; gpssec = t;
        const int day = t / (24 * 3600);
        t %= (24 * 3600);
        if (day < 0 || day > 6) return -1;
        wday = day; std_ = t / 3600; min_ = (t % 3600) / 60; sek = (float)(t % 60 + ms / 1000.0);
        return 0;
    }
    int gps_timeX() {
        uint32_t w[2] = { (uint32_t)be(frame + P_TOW, 4), (uint32_t)be(frame + P_TOW + 4, 4) };
        double f64; memcpy(&f64, w, 8);
        gpstowX = f64;
        gpstow = (int)(uint32_t)(int64_t)(gpstowX * 1e3);
        const uint32_t tow_u4 = (uint32_t)(int64_t)gpstowX;
        uint32_t t = tow_u4; gpssec = (int)tow_u4;
        const int day = (int)(t / (24 * 3600));
        t %= (24 * 3600);
        if (day < 0 || day > 6) return -1;
        wday = day; std_ = (int)(t / 3600); min_ = (int)((t % 3600) / 60); sek = (float)((t % 60) + f64 - tow_u4);
        return 0;
    }
    void gps_pos(int &err_alt) {
        const double B60B60 = (1 << 30) / 90.0;
        const int la = be(frame + P_LAT, 4), lo = be(frame + P_LON, 4), h = be(frame + P_ALT, 4);
        const bool six = (typ & 0xFF) == 6;
        lat = six ? la / B60B60 : la / 1e7; lon = six ? lo / B60B60 : lo / 1e7; alt = six ? h / 1000.0 : h / 100.0;
        err_alt = (alt < -200 || alt > 60000) ? -1 : 0;
    }
    static int s24(const uint8_t *p) { int v = p[0] << 16 | p[1] << 8 | p[2]; if (v > 0x7FFFFF) v -= 0x1000000; return v; }
    void gps_vel24() {
        const double vx = s24(frame + P_VE) / 1e3, vy = s24(frame + P_VN) / 1e3, vz = s24(frame + P_VU) / 1e3;
        vH = sqrt(vx * vx + vy * vy);
        double dir = atan2(vx, vy) * 180 / M_PI;
        if (dir < 0) dir += 360;
        vD = dir; vV = vz;
    }
    void gps_vel16X() {
        vH = (short)(frame[P_VH] << 8 | frame[P_VH + 1]) / 1e2;
        vD = (short)(frame[P_VD] << 8 | frame[P_VD + 1]) / 1e2;
        vV = (short)(frame[P_VV] << 8 | frame[P_VV + 1]) / 1e2;
    }

    // ---- print_frame (:713-798) ---------------------------------------------------------------------------------------------------
    void print_frame(Out &w, int crc_err) {
        if (frame[0] == 0 || !frame[P_SN + 1]) return;
        int err1 = 0, err2 = 0;
        sn = be(frame + P_SN, 4) & 0xFFFFFF;
        frnr = (frame[P_FRNB] << 8) + frame[P_FRNB + 1];
        w.f(" (%7d) ", sn); w.f(" [%5d] ", frnr);
        gps_pos(err2);
        if ((typ & 0xFF) == 6) { err1 = gps_time6(crc_err); gps_vel24(); }
        else { err1 = gps_timeX(); gps_vel16X(); }
        if (!err1) w.f("%s ", kDay[wday]);
        if (week > 0) {
            if (gpstow < gpstow_start && !crc_err) { week += 1; gpstow_start = gpstow; }
            gps2date();
            w.f("%04d-%02d-%02d ", jahr, monat, tag);
        }
        w.f("%02d:%02d:%06.3f ", std_, min_, sek);
        if (!err2) {
            w.f(" lat: %.5f ", lat); w.f(" lon: %.5f ", lon); w.f(" alt: %.2fm ", alt);
            w.f("  vH: %.1fm/s  D: %.1f  vV: %.1fm/s ", vH, vD, vV);
        }
        w.f(crc_err == 0 ? " [OK]" : " [NO]");
        w.f("\n");
        if (o.json && crc_err == 0) {
            char sntyp[] = "LMS6-", subtyp[12] = "LMS6-403";
            if (typ == 10) { sntyp[3] = 'X'; subtyp[3] = 'X'; }
            else if (typ == 0x0206) strcpy(subtyp, "LMS6-403-2");
            w.f("{ \"type\": \"%s\"", "LMS");
            w.f(", \"frame\": %d, \"id\": \"%s%d\", \"datetime\": \"", frnr, sntyp, sn);
            w.f("%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, \"heading\": %.5f, \"vel_v\": %.5f", std_, min_, sek, lat, lon, alt, vH, vD, vV);
            w.f(", \"gpstow\": %d", gpstow);
            w.f(", \"subtype\": \"%s\"", subtyp);
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            w.f(", \"ref_datetime\": \"%s\"", "GPS");
            w.f(", \"ref_position\": \"%s\"", "GPS");
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
            w.f("\n");
        }
    }
    void emit(Out &w, int crc_err) {
        if (o.raw == 1) {
            for (int i = 0; i < FRM_LEN; i++) w.f("%02x ", frame[i]);
            w.f(crc_err == 0 ? " [OK]" : " [NO]");
            w.f("\n");
        }
        if (o.raw == 0) print_frame(w, crc_err);
    }
    static int check_crc(const uint8_t *f) { return (((f[221] << 8) | f[222]) != crc16_0(f, 221)) ? 1 : 0; }

    int sync6_at(const uint8_t *bb, int p, int &is05) const {
        int s = 0;
        for (int j = 0; j < 3; j++) s += (bb[p + j] == kSync6[j]);
        is05 = (s + (bb[p + 3] == 0x05)) == 4;
        return (s + (bb[p + 3] == 0x00)) == 4 || is05;
    }
    int frmsync_X(const uint8_t *bb) {
        int p = SYNC_LEN;
        auto cnt = [&](int at) { int s = 0; for (int j = 0; j < 4; j++) s += (bb[at + j] == kSyncX[j]); return s; };
        sfX = cnt(SYNC_LEN);
        if (sfX < 4) {
            sfX = cnt(SYNC_LEN + 35);
            if (sfX == 4) p = SYNC_LEN + 35;
            else { sfX = cnt(SYNC_LEN + 40); if (sfX == 4) p = SYNC_LEN + 40; }
        }
        return p;
    }
    void rs_block(uint8_t *bb, int at) {                        // codeword reversed in the block (:885-889)
        uint8_t cw[255], ep[32], ev[32];
        for (int j = 0; j < 255; j++) cw[254 - j] = bb[at + j];
        sonde_ecc_decode(rs, cw, ep, ev);
        for (int j = 0; j < 255; j++) bb[at + j] = cw[254 - j];
    }

    // ---- proc_frame (:829-989) --------------------------------------------------------------------------------------------------------
    void proc_frame(Out &w, int len) {
        uint8_t bb[FRAME_LEN + 8];
        static thread_local char fbits[BITFRAME_LEN + OVERLAP * BITS + 8];
        if ((len % 8) > 4) while (len % 8) { blk[len].hb = '0'; blk[len].sb = -1; len++; }
        blk[len].hb = 0;
        const HS *raw = blk.data();
        if (o.vit) { viterbi(blk.data()); raw = vraw.data(); }
        const int err = deconv(raw, fbits);
        if (err) for (int i = err; i < RAWBLK / 2; i++) fbits[i] = 0;
        int blen;
        {   // bits2bytes: 8 characters LSB first, '1' and '9' count (:415-441)
            const int n = (int)strlen(fbits) / 8;
            for (int b = 0; b < n; b++) { int v = 0; for (int i = 0; i < 8; i++) { const char c = fbits[8 * b + i]; if (c == '1' || c == '9') v += 1 << i; } bb[b] = (uint8_t)v; }
            blen = n;
        }
        for (int j = blen; j < FRAME_LEN + 8; j++) bb[j] = 0;
        int p = SYNC_LEN;
        if ((typ & 0xFF) == 6) {
            if (o.ecc) rs_block(bb, SYNC_LEN);
            while (p - SYNC_LEN < FRM_LEN) {
                if (sf6 == 0) {
                    while (p - SYNC_LEN < FRM_LEN) {                       // frmsync_6
                        int is05; sf6 = 0;
                        for (int j = 0; j < 3; j++) sf6 += (bb[p + j] == kSync6[j]);
                        if (sync6_at(bb, p, is05)) { sf6 = 4; frm_pos = 0; typ = 6; if (is05) typ |= 0x0200; break; }
                        p++;
                    }
                    if (sf6 < 4) {
                        frmsync_X(bb);
                        if (sfX == 4) { if (auto_detect) { typ = 10; reset_dsp = 1; } break; }
                    }
                }
                if (sf6 && frm_pos < FRM_LEN) { frame[frm_pos] = bb[p]; frm_pos++; p++; }
                if (frm_pos == FRM_LEN) { emit(w, check_crc(frame)); frm_pos = 0; sf6 = 0; }
            }
        }
        if (typ == 10) {
            p = frmsync_X(bb);
            if (sfX < 4) {
                while (p - SYNC_LEN < FRM_LEN) {
                    int is05; sf6 = 0;
                    for (int j = 0; j < 3; j++) sf6 += (bb[p + j] == kSync6[j]);
                    if (sync6_at(bb, p, is05)) {
                        sf6 = 4; frm_pos = 0;
                        if (auto_detect) { reset_dsp = 1; typ = 6; if (is05) typ |= 0x0200; }
                        break;
                    }
                    p++;
                }
                if (frm_rate > 5000.0 || frm_rate < 4000.0) { if (auto_detect) { reset_dsp = 1; typ = 6; } }
            } else {
                if (blen > 100 && o.ecc) rs_block(bb, p);
                for (int j = 0; j < FRM_LEN; j++) frame[j] = bb[p + j];
                emit(w, check_crc(frame));
            }
        }
    }

    // what main does after a block (:1434-1462): the type the auto detection arrived at decides the next block's length / symbol rate
    void after_block() {
        rate_changed = 0;
        if (auto_detect && reset_dsp) {
            if (typ == 10) { rawblk_len = RAWBLK; rate_changed = 1; }
            if ((typ & 0xFF) == 6) { rawblk_len = RAWBLK6; rate_changed = 1; }
            reset_dsp = 0;
        }
    }
    void put_bit(float s0, const float *s1, int ecc3) {          // (:1393-1421)
        float sb = s0; int hb = s0 >= 0.0f;
        if (ecc3 && s1 && s0 * *s1 < 0) { sb += *s1; hb = sb >= 0.0f; }
        HS h; h.hb = (uint8_t)(hb ^ (int)(bc % 2));
        const int sgn = -2 * (int)(bc % 2) + 1;
        h.sb = sgn * sb;
        if (o.vit == 1) h.sb = (float)(2 * h.hb - 1);
        h.hb += 0x30;
        blk[pos] = h;
        bc++; pos++;
    }
};

extern "C" {

int sonde_lms6_dec_create(const sonde_lms6_opts_t *opts, sonde_lms6_dec_t **out) {
    if (!opts || !out || (opts->typ != 0 && opts->typ != 6 && opts->typ != 10) || (opts->ecc != 0 && opts->ecc != 1 && opts->ecc != 3) ||
        opts->vit < 0 || opts->vit > 2 || opts->raw < 0 || opts->raw > 1) return SONDE_E_ARG;
    sonde_lms6_dec *d = new sonde_lms6_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.json) { if (!d->o.ecc) d->o.ecc = 1; if (!d->o.vit) d->o.vit = 1; }
    d->rs = sonde_ecc_create(SONDE_ECC_RS255CCSDS);
    d->blk.assign(RAWBLK + BLOCKSTART + 9 + 16, HS{ 0, 0.f });
    for (int k = 0; k < BLOCKSTART; k++) { const int hb = kBlkSync[k] & 1; d->blk[k].hb = (uint8_t)(hb + 0x30); d->blk[k].sb = (float)(2 * hb - 1); }
    memset(d->frame, 0, sizeof d->frame); memcpy(d->frame, kSync6, 4);
    d->week = (opts->gpsweek >= 1024 && opts->gpsweek <= 3072) ? opts->gpsweek : 0;
    d->auto_detect = opts->typ == 0; d->typ = opts->typ == 10 ? 10 : 6;
    d->rawblk_len = d->typ == 10 ? RAWBLK : RAWBLK6;
    for (int bits = 0; bits < NST; bits++) {
        int cA = 0, cB = 0;
        for (int i = 0; i < KL; i++) { cA ^= (kPolyA[KL - 1 - i] & 1) & ((bits >> i) & 1); cB ^= (kPolyB[KL - 1 - i] & 1) & ((bits >> i) & 1); }
        d->code[bits] = (uint8_t)((cA << 1) | cB);
    }
    if (d->o.vit) {
        d->vraw.assign(RAWBITFRAME_LEN + OVERLAP * BITS * 2 + 8, HS{ 0, 0.f });
        St z; memset(&z, 0, sizeof z);
        d->vstate.assign((size_t)(RAWBITFRAME_LEN + OVERLAP + 8) * MST, z);
    }
    memset(d->sbuf, 0, sizeof d->sbuf);
    *out = d;
    return 0;
}

void sonde_lms6_dec_destroy(sonde_lms6_dec_t *d) { if (d) { sonde_ecc_destroy(d->rs); delete d; } }

int sonde_lms6_dec_block_bits(const sonde_lms6_dec_t *d) { return d ? d->rawblk_len - BLOCKSTART : SONDE_E_ARG; }

int sonde_lms6_dec_type(const sonde_lms6_dec_t *d, int32_t *changed) {
    if (!d) return SONDE_E_ARG;
    if (changed) *changed = d->rate_changed;
    return d->typ;
}

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

int sonde_lms6_dec_block(sonde_lms6_dec_t *d, const float *soft0, const float *soft1, int32_t nbits, float mv, float frm_rate, double t_elapsed,
                         char *out, size_t outlen) {
    if (!d || !out || nbits < 0 || (nbits > 0 && !soft0) || nbits > d->rawblk_len - BLOCKSTART) return SONDE_E_ARG;
    d->frm_rate = frm_rate;
    d->pos = BLOCKSTART; d->bc = mv > 0 ? 0u : 1u;
    for (int j = 0; j < nbits; j++) d->put_bit(soft0[j], soft1 ? soft1 + j : nullptr, d->o.ecc == 3);
    d->time_elapsed = t_elapsed;
    Out w;
    d->proc_frame(w, d->pos);
    d->after_block();
    return finish_out(w, out, outlen);
}

int sonde_lms6_dec_push_soft(sonde_lms6_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    const double nan = std::nan("");                            // dsp_t stays zeroed with soft input: frm_rate and the elapsed time are 0 / 0
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_block) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % 64;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < 64; k++) {
                if (j >= 64) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kRawHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            const float mv = (float)sum;
            if (fabs(mv) > 0.7f) { d->in_block = 1; d->pos = BLOCKSTART; d->bc = mv > 0 ? 0u : 1u; d->frm_rate = (float)nan; }
        } else {
            d->put_bit(s, nullptr, 0);
            if (d->pos >= d->rawblk_len) {
                d->time_elapsed = nan;
                d->proc_frame(w, d->pos);
                d->after_block();
                d->in_block = 0;
            }
        }
    }
    if (finish && d->in_block) {                                 // end of input inside a block: decoded with the bits that exist (:1399,1427-1430)
        d->time_elapsed = nan;
        d->proc_frame(w, d->pos);
        d->in_block = 0;
    }
    return finish_out(w, out, outlen);
}

}  // extern "C"
