// sonde_mrz_fields.cpp — MRZ (MP3-H1) frames -> the reference's text / JSON (include/sonde_mrz.h).  Host code, bit rate.
//
// One object = the gpx_t of demod/mod/mp3h1mod.c plus the file statics of print_gpx (alt0, t0) and the polarity state of its main loop.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_mrz.h"

namespace {

constexpr int CRCLEN_ECEF = 45, CRCLEN_LATLON = 42, BITFRAME_LEN = (CRCLEN_ECEF + 6) * 8, FRAME_LEN = BITFRAME_LEN / 8, HEADLEN = 44;
constexpr int P_CNT1 = 3, P_TIME = 4, P_ECEFX = 8, P_ECEFV = 20, P_NSATS = 26, P_T16 = 29, P_H16 = 31, P_ADCT = 35, P_ADCH = 39, P_CNT2 = 43, P_CFG = 44;
constexpr int P_LAT = 7, P_LON = 11, P_ALT = 15, P_VH = 19, P_VD = 21;
const char kHeader[] = "100110011001100110011001100110011001" "10101010";
const double EA = 6378137.0, EB = 6356752.31424518, EA2B2 = EA * EA - EB * EB, E2 = EA2B2 / (EA * EA), EE2 = EA2B2 / (EB * EB);

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[640]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

uint32_t u4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint16_t u2(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
int16_t i2(const uint8_t *p) { int v = p[0] | (p[1] << 8); if (v & 0x8000) v -= 0x10000; return (int16_t)v; }
float f32(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }

}  // namespace

struct sonde_mrz_dec {
    unsigned char hexbyte = 0;                               // --rawhex: the byte a pair that is not hex leaves in place
    sonde_mrz_opts_t o{};
    int bits_ofs = 8;
    uint8_t subcnt1 = 0, subcnt2 = 0, numSats = 0, cfg_ntc = 0, cfg_T = 0, cfg_H = 0, crcOK = 0;
    int yr = 0, mth = 0, day = 0, hrs = 0, min_ = 0, sec = 0;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    float calA = 0, calB = 0, calC = 0, A_adcT = 0, B_adcT = 0, C_adcT = 0, A_adcH = 0, B_adcH = 0, C_adcH = 0, Tadc = 0, RHadc = 0, T = 0, RH = 0;
    uint8_t frame[FRAME_LEN + 16];
    char frame_bits[BITFRAME_LEN + 16];
    uint32_t cfg[16];
    uint32_t snC = 0, snD = 0;
    int crclen = CRCLEN_ECEF, bitfrm_len = (CRCLEN_ECEF + 6) * 8;
    int sec_day = 0, sec_day_prev = 0, gps_cnt = 0, gps_cnt_prev = 0, week = 0;
    float alt0 = 0; int t0 = 0;               // statics of print_gpx (:651-652)
    int inv = 0;
    float sbuf[HEADLEN]; int bufpos = -1, in_frame = 0, pos = 0, have_s1 = 0; float s1 = 0;

    int crc16rev(int start, int len) const {
        int rem = 0xFFFF;
        if (start + len + 2 > FRAME_LEN) return -1;
        for (int i = 0; i < len; i++) {
            rem ^= frame[start + i];
            for (int j = 0; j < 8; j++) { rem = (rem & 1) ? (rem >> 1) ^ 0xA001 : rem >> 1; rem &= 0xFFFF; }
        }
        return rem;
    }
    int check_CRC(uint32_t len) const { return (uint32_t)u2(frame + len + 3) != (uint32_t)crc16rev(P_CNT1, (int)len); }
    void reset_time() { gps_cnt = 0; yr = 0; week = 0; }

    void bits2bytes(const char *bitstr, int len) {
        int bytepos = 0, bitpos = 0;
        while (bytepos < len) {
            int v = 0, d = 1, i;
            for (i = 0; i < 8; i++) {
                const char bit = bitstr[bitpos + 7 - i];
                if (bit == '\0') goto frame_end;
                if (bit == '1') v += d;
                d <<= 1;
            }
            bitpos += 8;
            frame[bytepos++] = (uint8_t)v;
        }
    frame_end:
        for (int i = bytepos; i < FRAME_LEN; i++) frame[i] = 0;
    }

    void get_cfg(int ofs) {
        subcnt1 = frame[P_CNT1] & 0xF;
        subcnt2 = frame[P_CNT2 + ofs];
        if (!crcOK) return;
        uint32_t c = u4(frame + P_CFG + ofs);
        cfg[subcnt1] = c;
        switch (subcnt1) {
            case 0x0: calA = f32(c); cfg_ntc |= 0x1; break;
            case 0x1: calB = f32(c); cfg_ntc |= 0x2; break;
            case 0x2: calC = f32(c); cfg_ntc |= 0x4; break;
            case 0x3: A_adcT = f32(c); cfg_T |= 0x1; break;
            case 0x4: B_adcT = f32(c); cfg_T |= 0x2; break;
            case 0x5: C_adcT = f32(c); cfg_T |= 0x4; break;
            case 0x6: A_adcH = f32(c); cfg_H |= 0x1; break;
            case 0x7: B_adcH = f32(c); cfg_H |= 0x2; break;
            case 0x8: C_adcH = f32(c); cfg_H |= 0x4; break;
            case 0xC: if (c != snC && snC > 0) { snD = 0; reset_time(); } snC = c; break;
            case 0xD: if (c != snD && snD > 0) { snC = 0; reset_time(); } snD = c; break;
            case 0xF: yr = (int)(c % 100); yr += 2000; c /= 100; mth = (int)(c % 100); c /= 100; day = (int)(c % 100); break;
            default: break;
        }
    }
    void get_time() {
        hrs = frame[P_TIME]; min_ = frame[P_TIME + 1]; sec = frame[P_TIME + 2];
        if (!crcOK) return;
        int wk = 0, tow = 0;
        gps_cnt_prev = gps_cnt; sec_day_prev = sec_day;
        sec_day = hrs * 60 * 60 + min_ * 60 + sec;
        if (yr == 0) { wk = 0; tow = sec_day; }
        else {                                              // datetime2GPSweek (:187-205)
            int yy = yr, mm = mth;
            if (mm < 3) { yy -= 1; mm += 12; }
            const int gpsDays = (int)(365.25 * yy) + (int)(30.6001 * (mm + 1.0)) + day - 723263;
            wk = gpsDays / 7;
            tow = (gpsDays % 7) * 86400 + hrs * 3600 + min_ * 60 + (int)(sec + 0.5);
        }
        const int sec_gps = wk * 604800 + tow;
        week = wk;
        if (sec_gps > gps_cnt_prev) gps_cnt = sec_gps;
    }
    void get_ecef() {
        double X[3], V[3];
        for (int k = 0; k < 3; k++) {
            int xyz; memcpy(&xyz, frame + P_ECEFX + 4 * k, 4);
            X[k] = xyz / 100.0;
            const uint8_t *g = frame + P_ECEFV + 2 * k;
            const short v16 = (short)(g[0] | g[1] << 8);
            V[k] = v16 / 100.0;
        }
        const double lam = atan2(X[1], X[0]), p = sqrt(X[0] * X[0] + X[1] * X[1]), t = atan2(X[2] * EA, p * EB);
        const double phi = atan2(X[2] + EE2 * EB * sin(t) * sin(t) * sin(t), p - E2 * EA * cos(t) * cos(t) * cos(t));
        const double R = EA / sqrt(1 - E2 * sin(phi) * sin(phi));
        alt = p / cos(phi) - R; lat = phi * 180 / M_PI; lon = lam * 180 / M_PI;
        if (alt < -1000.0 || alt > 80000.0) return;
        const double ph = lat * M_PI / 180.0, la = lon * M_PI / 180.0;
        const double vN = -V[0] * sin(ph) * cos(la) - V[1] * sin(ph) * sin(la) + V[2] * cos(ph);
        const double vE = -V[0] * sin(la) + V[1] * cos(la);
        const double vU = V[0] * cos(ph) * cos(la) + V[1] * cos(ph) * sin(la) + V[2] * sin(ph);
        vH = sqrt(vN * vN + vE * vE);
        double dir = atan2(vE, vN) * 180.0 / M_PI;
        if (dir < 0) dir += 360.0;
        vD = dir; vV = vU;
        numSats = frame[P_NSATS];
    }
    void get_latlon() {
        int v;
        memcpy(&v, frame + P_LAT, 4); lat = v * 1e-6;
        memcpy(&v, frame + P_LON, 4); lon = v * 1e-6;
        memcpy(&v, frame + P_ALT, 4); alt = v * 1e-2;
        if (alt < -1000.0 || alt > 80000.0) return;
        const short h = (short)(frame[P_VH] | (frame[P_VH + 1] << 8));
        const unsigned short dd = (unsigned short)(frame[P_VD] | (frame[P_VD + 1] << 8));
        vH = h / 100.0; vD = dd / 100.0; vV = 0;
        numSats = frame[P_NSATS - 3];
    }
    void get_ptu(int ofs) {
        float t = -273.15f, rh = -1.0f;
        const float ADC_MAX = 32767.0;
        const int ADCT = (int)u4(frame + P_ADCT + ofs); const float adc_t = (float)(ADCT / 100.0);
        const int ADCH = (int)u4(frame + P_ADCH + ofs); const float adc_h = (float)(ADCH / 100.0);
        if (cfg_ntc == 0x7 && cfg_T == 0x7) {
            const float poly1 = adc_t * adc_t * A_adcT + adc_t * B_adcT + C_adcT;
            const float Rt = (float)(100000.0 * poly1 / (ADC_MAX - poly1));
            if (Rt > 0.0) {
                t = (float)(calB / log(Rt / calA) - calC - 273.15f);
                if (t < -120.0f || t > 120.0f) t = -273.15f;
            }
        }
        Tadc = t;
        if (Tadc > -273.0f && cfg_H == 0x7) {
            const float poly2 = adc_h * adc_h * A_adcH + adc_h * B_adcH + A_adcH;      // (sic: A twice, :516)
            const float K = poly2 / ADC_MAX;
            rh = (float)((K - 0.1515) / (0.00636 * (1.05460 - 0.00216 * Tadc)));
            if (rh < -10.0f || rh > 120.0f) rh = -1.0f;
            else { if (rh < 0.0f) rh = 0.0f; if (rh > 100.0f) rh = 100.0f; }
        }
        RHadc = rh;
        T = (float)(i2(frame + P_T16 + ofs) / 100.0);
        RH = (float)(i2(frame + P_H16 + ofs) / 100.0);
    }

    void print_gpx(Out &w, int ok) {
        const int ofs = (crclen == CRCLEN_ECEF) ? 0 : -3;
        crcOK = (uint8_t)ok;
        get_cfg(ofs);
        get_time();
        if (ofs) get_latlon(); else get_ecef();
        get_ptu(ofs);
        if (sec_day != sec_day_prev || !o.uniq) {
            w.f(" [%2d] ", subcnt1);
            w.f(" (%02d:%02d:%02d) ", hrs, min_, sec);
            w.f(" lat: %.5f ", lat); w.f(" lon: %.5f ", lon); w.f(" alt: %.2f ", alt);
            w.f("  vH: %4.1f  D: %5.1f ", vH, vD);
            if (!ofs) w.f(" vV: %3.1f ", vV);
            if (o.verbose > 1) w.f("  sats: %d ", numSats);
            if (o.verbose > 1 && ofs < 0) {
                if (crcOK && sec_day > t0) {
                    if (t0 > 0 && sec_day < t0 + 10) w.f(" (d_alt: %+4.1f) ", (alt - alt0) / (float)(sec_day - t0));
                    alt0 = (float)alt; t0 = sec_day;
                }
            }
            if (o.ptu) {
                if (T > -273.0f || RH > -0.5f) w.f(" ");
                if (T > -273.0f) w.f(" T=%.2fC", T);
                if (RH > -0.5f) w.f(" RH=%.2f%%", RH);
                if (T > -273.0f || RH > -0.5f) w.f(" ");
                if (o.verbose > 1) {
                    if (Tadc > -273.0f || RHadc > -0.5f) w.f("  (");
                    if (Tadc > -273.0f) w.f(" T0=%.1fC", Tadc);
                    if (RHadc > -0.5f) w.f(" RH0=%.0f%%", RHadc);
                    if (Tadc > -273.0f || RHadc > -0.5f) w.f(" ) ");
                }
            }
            if (o.color) w.f(crcOK ? "  \x1b[38;5;2m[OK]\x1b[0m" : "  \x1b[38;5;1m[NO]\x1b[0m");
            else w.f("  %s", crcOK ? "[OK]" : "[NO]");
            if (crcOK) {
                if (o.verbose) {
                    switch (subcnt1) {
                        case 0x0: if (o.verbose > 1) w.f("  <%d> A: %.5f", subcnt2, calA); break;
                        case 0x1: if (o.verbose > 1) w.f("  <%d> B: %.2f", subcnt2, calB); break;
                        case 0x2: if (o.verbose > 1) w.f("  <%d> C: %.3f", subcnt2, calC); break;
                        case 0xC: w.f("  <%d> snC: %d", subcnt2, (int)snC); break;
                        case 0xD: w.f("  <%d> snD: %d", subcnt2, (int)snD); break;
                        case 0xE: w.f("  <%d> calDate: %06d", subcnt2, (int)cfg[subcnt1]); break;
                        case 0xF: w.f("  <%d> %04d-%02d-%02d", subcnt2, yr, mth, day); break;
                        default: if (o.verbose > 1) w.f("  <%d>", subcnt2); break;
                    }
                }
                if (o.dbg) {
                    w.f("    : ");
                    w.f("  0x%08X =", cfg[subcnt1]);
                    if (subcnt1 > 0x8) w.f(" %u ", cfg[subcnt1]);
                    else w.f(" %g ", f32(cfg[subcnt1]));
                }
            }
            w.f("\n");
        }
        if (o.json && crcOK && week > 0 && gps_cnt > gps_cnt_prev && snC > 0 && snD > 0) {
            if (gps_cnt - gps_cnt_prev > 60 && gps_cnt_prev > sec_day_prev) { snC = 0; snD = 0; reset_time(); }      // TIMEOUT_JSN
            else {
                w.f("{ \"type\": \"%s\"", "MRZ");
                w.f(", \"frame\": %lu, ", (unsigned long)gps_cnt);
                w.f("\"id\": \"MRZ-%d-%d\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%02dZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f",
                    (int)snC, (int)snD, yr, mth, day, hrs, min_, sec, lat, lon, alt);
                w.f(", \"vel_h\": %.5f, \"heading\": %.5f", vH, vD);
                if (!ofs) w.f(", \"vel_v\": %.5f", vV);
                w.f(", \"sats\": %d", numSats);
                if (o.ptu) {
                    if (T > -273.0f) w.f(", \"temp\": %.1f", T);
                    if (RH > -0.5f) w.f(", \"humidity\": %.1f", RH);
                }
                if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
                w.f(", \"ref_datetime\": \"%s\"", "UTC");
                w.f(", \"ref_position\": \"%s\"", !ofs ? "GPS" : "MSL");
                if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
                w.f(" }\n");
            }
        }
    }

    void classify(int &ok) {
        crclen = (u2(frame + 30) == 0xFFFF) ? CRCLEN_LATLON : CRCLEN_ECEF;
        ok = check_CRC((uint32_t)crclen) == 0;
        if (ok) bitfrm_len = (crclen + 6) * 8;
    }
    void print_frame(Out &w, int npos, int b2B) {
        int ok = 0;
        if (b2B) {
            if (o.raw == 2) { for (int j = 0; j < npos; j++) w.s += frame_bits[j]; w.f("\n"); return; }
            const int frmlen = (npos - bits_ofs) / 8;
            bits2bytes(frame_bits + bits_ofs, frmlen);
            classify(ok);
            if (o.raw == 1) {
                for (int j = 0; j < frmlen; j++) w.f("%02X ", frame[j]);
                w.f(" %s", ok ? "[OK]" : "[NO]"); w.f("\n");
            } else if (npos / 8 > P_ECEFV + 6) print_gpx(w, ok);
        } else {
            classify(ok);
            if (o.raw) {
                for (int j = 0; j < npos; j++) w.f("%02X ", frame[j]);
                w.f(" %s", ok ? "[OK]" : "[NO]"); w.f("\n");
            } else if (npos > P_ECEFV + 6) print_gpx(w, ok);
        }
    }
    void put(float s) { frame_bits[pos++] = (char)(0x30 + (((s >= 0.0f) ^ (inv ? 0 : 1)) & 1)); }       // Manchester1 unless inverted (:1227-1233)
};

extern "C" {

int sonde_mrz_dec_create(const sonde_mrz_opts_t *opts, sonde_mrz_dec_t **out) {
    if (!opts || !out || opts->raw < 0 || opts->raw > 2 || opts->bits_ofs < 0 || opts->bits_ofs > 64) return SONDE_E_ARG;
    sonde_mrz_dec *d = new sonde_mrz_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    d->bits_ofs = opts->bits_ofs_given ? opts->bits_ofs : 8;
    d->inv = opts->inv != 0;
    memset(d->frame, 0, sizeof d->frame); memset(d->frame_bits, 0, sizeof d->frame_bits); memset(d->cfg, 0, sizeof d->cfg); memset(d->sbuf, 0, sizeof d->sbuf);
    for (int i = 0; i < HEADLEN / 2; i++) d->frame_bits[i] = (kHeader[2 * i] == '1' && kHeader[2 * i + 1] == '0') ? '1' : '0';      // manchester1(mrz_header) (:1173)
    *out = d;
    return 0;
}

void sonde_mrz_dec_destroy(sonde_mrz_dec_t *d) { delete d; }

int sonde_mrz_dec_frame_bits(const sonde_mrz_dec_t *d) { return d ? d->bitfrm_len - HEADLEN / 2 : SONDE_E_ARG; }

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.data(), w.s.size()); out[w.s.size()] = 0;
    return (int)w.s.size();
}

int sonde_mrz_dec_frame(sonde_mrz_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen) {
    if (!d || !out || n < 0 || n > d->bitfrm_len - HEADLEN / 2 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    d->pos = HEADLEN / 2;
    for (int j = 0; j < n; j++) d->frame_bits[d->pos++] = (char)(0x30 + !(soft[j] >= 0.0f));        // the engine's bits carry the polarity; Manchester1 = the other sense
    d->frame_bits[d->pos] = '\0';
    d->print_frame(w, d->pos, 1);
    return finish_out(w, out, outlen);
}

int sonde_mrz_dec_rawhex(sonde_mrz_dec_t *d, const char *line, char *out, size_t outlen) {
    if (!d || !line || !out) return SONDE_E_ARG;
    Out w;
    char buf[3 * FRAME_LEN + 12];
    strncpy(buf, line, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    buf[3 * FRAME_LEN] = '\0';
    char *sp = strchr(buf, '[');
    if (sp) *sp = '\0';
    const int len = (int)strlen(buf) / 3;
    if (len > 20) {
        unsigned char &b = d->hexbyte;                           // keeps its value from line to line, as the reference's variable does
        for (int i = 0; i < len; i++) { sscanf(buf + 3 * i, "%2hhx", &b); d->frame[i] = b; }
        d->print_frame(w, len, 0);
    }
    return finish_out(w, out, outlen);
}

int sonde_mrz_dec_push_soft(sonde_mrz_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_frame) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % HEADLEN;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < HEADLEN; k++) {
                if (j >= HEADLEN) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            const float mv = (float)sum;
            if (fabs(mv) > 0.82f) {
                int found = 1;
                if (mv * (0.5 - d->inv) < 0) { if (!d->o.aut) found = 0; else d->inv ^= 1; }
                if (found) { d->in_frame = 1; d->pos = HEADLEN / 2; d->have_s1 = 0; }
            }
        } else {
            if (!d->have_s1) { d->s1 = s; d->have_s1 = 1; continue; }
            d->have_s1 = 0;
            d->put(s - d->s1);                                   // both half symbols of a bit (:1205-1211)
            if (d->pos >= d->bitfrm_len) { d->frame_bits[d->pos] = '\0'; d->print_frame(w, d->pos, 1); d->in_frame = 0; }
        }
    }
    if (finish && d->in_frame) { d->frame_bits[d->pos] = '\0'; d->print_frame(w, d->pos, 1); d->in_frame = 0; }
    return finish_out(w, out, outlen);
}

}  // extern "C"
