// sonde_mts01_fields.cpp — Meteosis MTS01 frames -> the reference's text / JSON (include/sonde_mts01.h).  Host code, bit rate.
//
// One object = the gpx_t of demod/mod/mts01mod.c: the bit and byte buffers persist from frame to frame exactly as the reference's do (a
// short last frame is completed by what the previous one left behind).  print_frame :151-286, crc16_re :76-99, fn :129-137, get_Temp :139-148.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_mts01.h"

namespace {

constexpr int OFS = 1, FRAMELEN = 130 + OFS, BITFRAMELEN = 8 * FRAMELEN, DATLEN = 128;
const char kRawHeader[] = "10101010" "10101010" "10110100" "00101011";

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[640]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

uint32_t crc16_re(const uint8_t *p, int len) {
    uint32_t rem = 0xFFFF, re = 0;
    for (int i = 0; i < len; i++) {
        rem ^= (uint32_t)p[i] << 8;
        for (int j = 0; j < 8; j++) { rem = (rem & 0x8000) ? (rem << 1) ^ 0x8005 : (rem << 1); rem &= 0xFFFF; }
    }
    for (int j = 0; j < 16; j++) if (rem & (1u << (15 - j))) re |= 1u << j;
    return re;
}

float get_temp(float R) {                                // thermistor: 1/T = 1/T0 + log(R/R0)/B
    const float B0 = 3000.0, T0 = 0.0 + 273.15, R0 = 15.0;
    float T = 0;
    if (R > 0) T = (float)(1.0 / (1.0 / T0 + 1.0 / B0 * log(R / R0)));
    return (float)(T - 273.15);
}

}  // namespace

struct sonde_mts01_dec {
    sonde_mts01_opts_t o{};
    int frnr = 0, year = 0, month = 0, day = 0, hrs = 0, min_ = 0, sec = 0, batt = 0;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0;
    float T = 0;
    char ID[12];
    uint8_t frame_bytes[FRAMELEN + 4];
    char frame_bits[BITFRAMELEN + 8];
    char frm_str[FRAMELEN + 4];
    float sbuf[32]; int bufpos = -1, in_frame = 0, pos = 0;

    int fn(int n) const {
        int p = 0;
        if (n <= 0) return 0;
        while (n > 0 && p < DATLEN) { if (frm_str[p] == '\0') n -= 1; p += 1; }
        return p;
    }

    void print_frame(Out &w, int npos) {
        if (npos / 8 < OFS + DATLEN) return;
        for (int b = 0; b < FRAMELEN; b++) {                 // MSB first; anything but '1' is 0
            int v = 0;
            for (int i = 0; i < 8; i++) if (frame_bits[8 * b + 7 - i] == '1') v += 1 << i;
            frame_bytes[b] = (uint8_t)v;
        }
        const int crcdat = (frame_bytes[OFS + DATLEN + 1] << 8) | frame_bytes[OFS + DATLEN];
        const int crcval = (int)crc16_re(frame_bytes + OFS, DATLEN);
        const bool crc_ok = crcdat == crcval;
        if (o.raw) {
            if (o.raw == 1) {
                for (int j = 0; j < FRAMELEN; j++) w.f("%02X ", frame_bytes[j]);
                w.f(" # [%04X:%04X]", crcdat, crcval);
                w.f(" # [%s]", crc_ok ? "OK" : "NO");
            } else {
                for (int j = 0; j < BITFRAMELEN; j++) { w.s += frame_bits[j]; if (j % 8 == 7) w.s += ' '; }
            }
            w.f("\n");
            return;
        }
        w.s += (const char *)(frame_bytes + OFS);            // "%s": up to the first NUL (the buffer ends in zeros)
        w.f("  [%s]", crc_ok ? "OK" : "NO");
        w.f("\n");
        memset(frm_str, 0, FRAMELEN);
        strncpy(frm_str, (const char *)frame_bytes + OFS, DATLEN);
        for (int j = 0; j < DATLEN; j++) if (frm_str[j] == ',') frm_str[j] = '\0';
        strncpy(ID, frm_str + fn(0), 8);
        frnr = atoi(frm_str + fn(2));
        char dt[13];
        strncpy(dt, frm_str + fn(3), 12); dt[12] = '\0';
        sec = atoi(dt + 10); dt[10] = '\0';
        min_ = atoi(dt + 8); dt[8] = '\0';
        hrs = atoi(dt + 6); dt[6] = '\0';
        day = atoi(dt + 4); dt[4] = '\0';
        month = atoi(dt + 2); dt[2] = '\0';
        year = atoi(dt) + 2000;
        batt = (int)atof(frm_str + fn(4));
        lat = atof(frm_str + fn(5));
        lon = atof(frm_str + fn(6));
        alt = atof(frm_str + fn(7));
        vD = atof(frm_str + fn(8));
        vH = atof(frm_str + fn(9));
        T = get_temp((float)atof(frm_str + fn(11)));
        if (o.verbose) {
            w.f(" [%4d] ", frnr);
            w.f(" (%s) ", ID);
            w.f(" %4d-%02d-%02d ", year, month, day);
            w.f("%02d:%02d:%02d ", hrs, min_, sec);
            w.f(" lat: %.6f  lon: %.6f  alt: %.0f ", lat, lon, alt);
            w.f("  vH: %4.1f  D: %5.1f ", vH, vD);
            w.f(" Vbat:%.1fV ", batt / 1000.0);
            if (T > -270.0f) w.f("  T=%.1fC ", T);
            w.f("\n");
        }
        if (o.json && crc_ok) {
            w.f("{ \"type\": \"%s\"", "MTS01");
            w.f(", \"frame\": %d, \"id\": \"MTS01-%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, \"heading\": %.5f",
                frnr, ID, year, month, day, hrs, min_, (float)sec, lat, lon, alt, vH, vD);
            w.f(", \"batt\": %.2f", batt / 1000.0);
            if (T > -270.0f) w.f(", \"temp\": %.1f", T);
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            w.f(", \"ref_datetime\": \"%s\"", "UTC");
            w.f(", \"ref_position\": \"%s\"", "MSL");
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
        }
        if (o.verbose || (o.json && crc_ok)) w.f("\n");
    }
};

extern "C" {

int sonde_mts01_dec_create(const sonde_mts01_opts_t *opts, sonde_mts01_dec_t **out) {
    if (!opts || !out || opts->raw < 0 || opts->raw > 2) return SONDE_E_ARG;
    sonde_mts01_dec *d = new sonde_mts01_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    memset(d->ID, 0, sizeof d->ID); memset(d->frame_bytes, 0, sizeof d->frame_bytes); memset(d->frame_bits, 0, sizeof d->frame_bits);
    memset(d->frm_str, 0, sizeof d->frm_str); memset(d->sbuf, 0, sizeof d->sbuf);
    *out = d;
    return 0;
}

void sonde_mts01_dec_destroy(sonde_mts01_dec_t *d) { delete d; }

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.data(), w.s.size()); out[w.s.size()] = 0;
    return (int)w.s.size();
}

int sonde_mts01_dec_frame(sonde_mts01_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen) {
    if (!d || !out || n < 0 || n > BITFRAMELEN || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int j = 0; j < n; j++) d->frame_bits[j] = (char)(0x30 + (soft[j] >= 0.0f));
    d->frame_bits[n] = '\0';
    d->print_frame(w, n);
    return finish_out(w, out, outlen);
}

int sonde_mts01_dec_push_soft(sonde_mts01_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_frame) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % 32;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < 32; k++) {
                if (j >= 32) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kRawHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            if (fabs((float)sum) > 0.8f) { d->in_frame = 1; d->pos = 0; }
        } else {
            d->frame_bits[d->pos++] = (char)(0x30 + (s >= 0.0f));
            if (d->pos >= BITFRAMELEN) { d->frame_bits[d->pos] = '\0'; d->print_frame(w, d->pos); d->in_frame = 0; }
        }
    }
    if (finish && d->in_frame) { d->frame_bits[d->pos] = '\0'; d->print_frame(w, d->pos); d->in_frame = 0; }
    return finish_out(w, out, outlen);
}

}  // extern "C"
