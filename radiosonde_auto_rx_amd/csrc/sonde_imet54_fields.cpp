// sonde_imet54_fields.cpp — iMet-54 / iMet-50 frames -> the reference's text / JSON (include/sonde_imet54.h).  Host code, bit rate.
//
// One object = the gpx_t of demod/mod/imet54mod.c (frame bytes and bits persist between frames as the reference's do) plus the polarity
// state of its main loop.  print_frame :618-707, print_position :494-616, the two check sums :229-303,:350-360, getters :362-475.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_imet54.h"

namespace {

constexpr int BITS = 10, FRAME_LEN = 220, BITFRAME_LEN = FRAME_LEN * BITS;
constexpr int P_SN = 0x00, P_TIME = 0x04, P_LAT = 0x08, P_LON = 0x0C, P_ALT = 0x10, P_T = 0x1C, P_RH = 0x20, P_TRH = 0x24, P_STATUS = 0x2A, P_F8 = 0x52, P_CRC32CONT = 0x34;
const char kHeader[] = "0000000001" "0101010101" "0001001001" "0001001001";
const uint8_t kH[4][8] = { { 1, 0, 1, 0, 1, 0, 1, 0 }, { 0, 1, 1, 0, 0, 1, 1, 0 }, { 0, 0, 0, 1, 1, 1, 1, 0 }, { 1, 1, 1, 1, 1, 1, 1, 1 } };
const uint8_t kHe[8] = { 0x9, 0xA, 0xB, 0xC, 0xD, 0xE, 0xF, 0x8 };
const uint8_t kHamLut[16] = { 0x00, 0x87, 0x99, 0x1E, 0xAA, 0x2D, 0x33, 0xB4, 0x4B, 0xCC, 0xD2, 0x55, 0xE1, 0x66, 0x78, 0xFF };

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[640]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

int check(uint8_t code[8]) {
    uint32_t synval = 0;
    for (int i = 0; i < 4; i++) { uint8_t s = 0; for (int j = 0; j < 8; j++) s ^= kH[i][j] & code[j]; synval |= (uint32_t)s << i; }
    int ret = 0;
    if (synval) { ret = -1; for (int j = 0; j < 8; j++) if (synval == kHe[j]) { ret = j + 1; break; } }
    if (ret > 0) code[ret - 1] ^= 0x1;
    return ret;
}

uint8_t hamming(int opt_ecc, uint8_t *cwb, uint8_t *sym) {
    const int ecc = opt_ecc ? check(cwb) : 0;
    uint8_t byt = 0, nib;
    for (int j = 0; j < 8; j++) byt |= (cwb[j] & 1) << j;
    for (nib = 0; nib < 16; nib++) if (byt == kHamLut[nib]) break;
    *sym = nib;
    return (ecc < 0 || nib >= 16) ? 0xF0 : ecc > 0 ? 1 : 0;
}

int crc32ok(const uint8_t *bytes, int len) {            // two 16-bit registers stepping a 32-bit polynomial over the frame, words taken backwards (:229-284)
    const uint32_t poly0 = 0x0EDB, poly1 = 0x8260;
    int n = 104, b = 0;
    uint32_t c0 = 0x48EB, c1 = 0x1ACA, nx_c0 = c0, nx_c1 = c1, crc0 = 0, crc1 = 0;
    const uint32_t data_c0 = (bytes[100] << 8) | bytes[101], data_c1 = (bytes[106] << 8) | bytes[107];
    if (len < 108) return 0;
    while (n >= 0) {
        if (n < 100 || (n > 101 && n < 106)) if ((bytes[n] >> b) & 1) { crc0 ^= c0; crc1 ^= c1; }
        if (c1 & 0x8000) { nx_c0 ^= poly0; nx_c1 ^= poly1; }
        nx_c0 <<= 1; nx_c1 <<= 1;
        if (c1 & 0x8000) nx_c0 |= 1;
        if ((c1 ^ c0) & 0x8000) nx_c1 |= 1;
        nx_c0 &= 0xFFFF;
        c0 = nx_c0; c1 = nx_c1;
        if (b < 7) b += 1;
        else { b = 0; if (n % 4 == 3) n -= 7; else n += 1; }
    }
    crc0 ^= data_c0 ^ 0x5000; crc1 ^= data_c1 ^ 0x1DAD;
    return crc1 == 0 && (crc0 & 0xF000) == 0;
}

uint32_t crc32_802(const uint8_t *msg, int len) {
    uint32_t rem = 0;
    for (int i = 0; i < len; i++) {
        rem ^= (uint32_t)msg[i] << 24;
        for (int j = 0; j < 8; j++) rem = (rem & 0x80000000u) ? (rem << 1) ^ 0x04C11DB7u : rem << 1;
    }
    return rem ^ 0x63D60875u;
}

uint32_t u4be(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int crc32ok_cont(const uint8_t *bytes) {
    uint8_t m4[P_CRC32CONT] = { 0 };
    for (int i = 0; i < P_CRC32CONT / 4; i++) for (int j = 0; j < 4; j++) m4[4 * i + j] = bytes[4 * i + 3 - j];
    return crc32_802(m4, P_CRC32CONT) == u4be(bytes + P_CRC32CONT);
}

float vaporSatP(float Tc) {                             // Hyland and Wexler
    const double T = Tc + 273.15f;
    const double p = expf((float)(-5800.2206 / T + 1.3914993 + 6.5459673 * log(T) - 4.8640239e-2 * T + 4.1764768e-5 * T * T - 1.4452093e-8 * T * T * T));
    return (float)p;
}

}  // namespace

struct sonde_imet54_dec {
    unsigned char hexbyte = 0;                               // --rawhex: the byte a pair that is not hex leaves in place
    sonde_imet54_opts_t o{};
    uint32_t SNu32 = 0;
    int timems = 0, std_ = 0, min_ = 0;
    float sek = 0.f;
    double lat = 0, lon = 0, alt = 0;
    float T = 0, _RH = 0, Trh = 0, RH = 0;
    uint16_t status = 0;
    uint8_t frame[FRAME_LEN + 4];
    uint8_t frame_bits[BITFRAME_LEN + 8];
    // print_frame's nib[] / ec[] (:623-624) are locals the reference never clears; a frame cut short sums entries it did not compute (:651).  In the
    // compiled reference they keep what the previous frame left there (print_frame is inlined into main's frame), which is what members do.
    uint8_t nib[FRAME_LEN] = { 0 }, ec[FRAME_LEN] = { 0 };
    int inv = 0;
    float sbuf[40]; int bufpos = -1, in_frame = 0, pos = 0;

    int get_GPS() {
        int val = (int)u4be(frame + P_TIME);
        timems = val;
        sek = (float)((val % 100000) / 1e3);
        val /= 1000; val /= 100; min_ = val % 100; val /= 100; std_ = val % 100;
        val = (int)u4be(frame + P_LAT);
        int deg = (int)(val / 1e6); float mn = (float)((val / 1e6 - deg) * 100.0 / 60.0);
        lat = (float)deg + mn;
        val = (int)u4be(frame + P_LON);
        deg = (int)(val / 1e6); mn = (float)((val / 1e6 - deg) * 100.0 / 60.0);
        lon = (float)deg + mn;
        val = (int)u4be(frame + P_ALT);
        alt = val / 1e1;
        if (timems < 0.0 || timems > 235959999) return -1;
        if (lat < -90.0 || lat > 90.0) return -2;
        if (lon < -180.0 || lon > 180.0) return -2;
        if (alt < -400.0 || alt > 60000.0) return -2;
        return 0;
    }
    int get_PTU() {
        int count_1e9 = 0;
        float rh = -1.0f, f;
        uint32_t val = u4be(frame + P_T); memcpy(&f, &val, 4);
        T = (f > -120.0f && f < 80.0f) ? f : -273.15f;
        if (val == 0x4E6E6B28) { T = -273.15f; count_1e9 += 1; }
        val = u4be(frame + P_RH); memcpy(&f, &val, 4);
        _RH = f < 0.0f ? 0.0f : f > 100.0f ? 100.0f : f;
        if (val == 0x4E6E6B28) { _RH = -1.0f; count_1e9 += 1; }
        val = u4be(frame + P_TRH); memcpy(&f, &val, 4);
        Trh = (f > -120.0f && f < 80.0f) ? f : -273.15f;
        if (val == 0x4E6E6B28) { Trh = -273.15f; count_1e9 += 1; }
        if (T > -273.0f && Trh > -273.0f) {
            rh = _RH * vaporSatP(Trh) / vaporSatP(T);
            if (rh < 0.0f) rh = 0.0f;
            if (rh > 100.0f) rh = 100.0f;
        }
        RH = rh;
        return count_1e9;
    }
    void crc_tag(Out &w, int &crc_ok, int ecc_std, int *std_ok) {
        if (crc_ok) { w.f(" [OK]"); return; }
        crc_ok = crc32ok_cont(frame);
        if (crc_ok) w.f(" [ok]");
        else if (ecc_std == 0) { w.f(" [oo]"); if (std_ok) *std_ok = 1; }
        else if (frame[P_F8] == 0xF8) w.f(" [NO]");
        else w.f(" [no]");
    }

    void print_position(Out &w, int len, int ecc_frm, int ecc_tlm, int ecc_std) {
        int prnGPS = 0, prnPTU = 0, prnSTS = 0, ptu1e9 = 0, std_ok = 0, rs_type = 54;
        int crc_ok = crc32ok(frame, len);
        int frm_ok = (ecc_frm >= 0 && len > P_F8);
        SNu32 = 0; timems = 0; std_ = 0; min_ = 0; sek = 0.0f; lat = lon = alt = 0.0; T = -273.15f; Trh = -273.15f; _RH = -1.0f; RH = -1.0f; status = 0;
        if (len > P_ALT + 4) {
            SNu32 = u4be(frame + P_SN);
            if (get_GPS() == 0) prnGPS = 1; else frm_ok = 0;
        }
        if (len > P_TRH + 4) { ptu1e9 = get_PTU(); prnPTU = 1; }
        if (len > P_STATUS + 2) { status = (uint16_t)((frame[P_STATUS] << 8) | frame[P_STATUS + 1]); prnSTS = 1; }
        if (frm_ok) {
            int sum = 0;
            for (int p = P_STATUS + 2; p < P_F8; p++) sum += frame[p];
            if (sum == 0 && (status & 0xF0F) == 0 && ptu1e9 == 3) rs_type = 50;
        }
        if (prnGPS && !o.silent) {
            w.f(" (%d) ", (int)SNu32);
            w.f(" %02d:%02d:%06.3f ", std_, min_, sek);
            w.f(" lat: %.5f ", lat); w.f(" lon: %.5f ", lon); w.f(" alt: %.1f ", alt);
            if (o.ptu && prnPTU) {
                w.f(" ");
                if (T > -273.0f) w.f(" T=%.1fC ", T);
                if (o.verbose) {
                    if (_RH > -0.5f) w.f(" _RH=%.0f%% ", _RH);
                    if (Trh > -273.0f) w.f(" _Trh=%.1fC ", Trh);
                }
                if (RH > -0.5f) w.f(" RH=%.0f%% ", RH);
            }
            crc_tag(w, crc_ok, ecc_std, &std_ok);
            if (o.verbose && prnSTS) w.f("  [%04X] ", status);
            if (o.ecc && ecc_frm != 0) { w.f(" #  (%d)", ecc_frm); if (o.verbose) w.f(" [%d]", ecc_tlm); }
            w.f("\n");
        }
        if (o.json && frm_ok && (crc_ok || std_ok) && (status & 0x30) == 0x30) {
            const unsigned long count_day = (unsigned long)((float)(std_ * 3600 + min_ * 60) + sek + 0.5);
            w.f("{ \"type\": \"%s\"", "IMET5");
            w.f(", \"frame\": %lu", count_day);
            w.f(", \"id\": \"IMET5-%u\", \"datetime\": \"%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f", SNu32, std_, min_, sek, lat, lon, alt);
            if (o.ptu) {
                if (T > -273.0f) w.f(", \"temp\": %.1f", T);
                if (RH > -0.5f) w.f(", \"humidity\": %.1f", RH);
            }
            w.f(", \"subtype\": \"%s\"", rs_type == 54 ? "iMet-54" : "iMet-50");
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            w.f(", \"ref_datetime\": \"%s\"", "UTC");
            w.f(", \"ref_position\": \"%s\"", "MSL");
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
            w.f("\n");
        }
    }

    void print_frame(Out &w, int len, int b2B) {
        int ecc_frm = 0, ecc_std = 0, ecc_tlm = 0;
        if (b2B) {
            static thread_local uint8_t bits8n1[BITFRAME_LEN + 10], bits[BITFRAME_LEN];
            for (int i = len; i < BITFRAME_LEN; i++) frame_bits[i] = 0;
            memset(bits8n1, 0, sizeof bits8n1); memset(bits, 0, sizeof bits);      // nib[] / ec[] are not cleared: see the members' comment
            uint8_t *q = bits8n1;
            for (int n = 0; n < len; n++) if (n % 10 > 0 && n % 10 < 9) *q++ = frame_bits[n];          // de8n1
            len = (8 * len) / 10;
            len -= 24;                                                                                // (0x24 0x24) 0x24 0x24 0x42
            int n = 0;
            while (n + 64 <= len) {                                                                   // deinter64: 8 x 8 transpose
                for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) bits[n + 8 * j + i] = bits8n1[24 + n + 8 * i + j];
                n += 64;
            }
            len -= len - n;
            for (int j = 0; j < len / 8; j++) ec[j] = hamming(o.ecc, bits + 8 * j, nib + j);
            for (int j = 0; j < len / 16; j++) frame[j] = (uint8_t)((nib[2 * j] << 4) | (nib[2 * j + 1] & 0xF));
            int j;
            for (j = 0; j < 2 * P_CRC32CONT; j++) {
                ecc_frm += ec[j];
                if (ec[j] > 0x10) ecc_frm = -1;
                if (j < 2 * (P_STATUS + 2)) ecc_tlm = ecc_frm;
                ecc_std = ecc_frm;
                if (ecc_frm < 0) break;
            }
            if (j < 2 * P_CRC32CONT) ecc_std = -1;
        } else ecc_frm = -2;
        if (o.raw) {
            int crc_ok = crc32ok(frame, len / 16);
            for (int i = 0; i < len / 16; i++) {
                w.f("%02X", frame[i]);
                if (o.raw > 1) { w.f(" "); if (o.raw == 4 && i % 4 == 3) w.f(" "); }
            }
            crc_tag(w, crc_ok, ecc_std, nullptr);
            if (o.ecc && ecc_frm != 0) { w.f(" # (%d)", ecc_frm); w.f(" [%d]", ecc_tlm); }
            w.f("\n");
            if (o.silent) print_position(w, len / 16, ecc_frm, ecc_tlm, ecc_std);
        } else print_position(w, len / 16, ecc_frm, ecc_tlm, ecc_std);
    }
};

extern "C" {

int sonde_imet54_dec_create(const sonde_imet54_opts_t *opts, sonde_imet54_dec_t **out) {
    if (!opts || !out || (opts->raw != 0 && opts->raw != 1 && opts->raw != 4)) return SONDE_E_ARG;
    sonde_imet54_dec *d = new sonde_imet54_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.json) d->o.ecc = 1;
    if (d->o.raw && d->o.json) d->o.silent = 1;
    d->inv = opts->inv != 0;
    memset(d->frame, 0, sizeof d->frame); memset(d->frame_bits, 0, sizeof d->frame_bits); memset(d->sbuf, 0, sizeof d->sbuf);
    *out = d;
    return 0;
}

void sonde_imet54_dec_destroy(sonde_imet54_dec_t *d) { delete d; }

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

int sonde_imet54_dec_frame(sonde_imet54_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen) {
    if (!d || !out || n < 0 || n > BITFRAME_LEN || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int j = 0; j < n; j++) d->frame_bits[j] = (uint8_t)(soft[j] >= 0.0f);
    d->print_frame(w, n, 1);
    return finish_out(w, out, outlen);
}

int sonde_imet54_dec_rawhex(sonde_imet54_dec_t *d, const char *line, char *out, size_t outlen) {
    if (!d || !line || !out) return SONDE_E_ARG;
    Out w;
    char buf[2 * FRAME_LEN + 12];
    strncpy(buf, line, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    buf[2 * FRAME_LEN] = '\0';
    char *sp = strchr(buf, ' ');
    if (sp) *sp = '\0';
    const int len = (int)strlen(buf) / 2;
    if (len > 20) {
        unsigned char &b = d->hexbyte;                           // keeps its value from line to line, as the reference's variable does
        for (int i = 0; i < len; i++) { sscanf(buf + 2 * i, "%2hhx", &b); d->frame[i] = b; }          // a pair that is not hex keeps the previous byte (:1104-1107)
        d->print_frame(w, len * 16, 0);
    }
    return finish_out(w, out, outlen);
}

int sonde_imet54_dec_push_soft(sonde_imet54_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_frame) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % 40;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < 40; k++) {
                if (j >= 40) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            const float mv = (float)sum;
            if (fabs(mv) > 0.8f) {
                int found = 1;
                if (mv * (0.5 - d->inv) < 0) { if (!d->o.aut) found = 0; else d->inv ^= 1; }      // :1018-1021
                if (found) { d->in_frame = 1; d->pos = 0; }
            }
        } else {
            d->frame_bits[d->pos++] = (uint8_t)((s >= 0.0f) ^ d->inv);
            if (d->pos >= BITFRAME_LEN) { d->print_frame(w, d->pos, 1); d->in_frame = 0; }
        }
    }
    if (finish && d->in_frame) { d->print_frame(w, d->pos, 1); d->in_frame = 0; }
    return finish_out(w, out, outlen);
}

}  // extern "C"
