// sonde_m10_fields.cpp — M10 / M10+ telemetry and the text / JSON lines of the reference's m10mod print_pos()
// (include/sonde_m10.h).  Host side, one 101-byte frame per second per sonde.
//
//   Trimble block (type 0x9F): velocities, time of week, lat / lon as 2^32/360 fractions, altitude, satellites, GPS-UTC offset,
//        week with the 1024-week rollover repair ............................................ Decoder::trimble()   m10mod.c:287-470
//   Gtop block (type 0xAF): decimal date / time, 1e-6 degrees, signed 24-bit altitude ........ Decoder::gtop()      :488-575
//   serial number text ....................................................................... Decoder::serial()    :472-486
//   thermistor (3 ranges, Steinhart-Hart fit), second NTC, humidity counter ratio with the
//        empirical temperature compensation, battery ......................................... Decoder::temp() ...  :635-860
//   text line and JSON ....................................................................... Decoder::print()     :862-1047
#include "../../include/sonde_m10.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace {
struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[1024];
        va_list ap; va_start(ap, fmt);
        const int n = vsnprintf(b, sizeof b, fmt, ap);
        va_end(ap);
        if (n > 0) s.append(b, (size_t)(n < (int)sizeof b ? n : (int)sizeof b - 1));
    }
};
const char *const kDay[7] = { "Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat" };
inline int be32(const uint8_t *p) { return (int)((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]); }
inline short be16(const uint8_t *p) { return (short)(p[0] << 8 | p[1]); }
}  // namespace

struct sonde_m10_dec {
    sonde_m10_opts_t o;
    const uint8_t *fb = nullptr;
    int type = 0;
    // gpx_t members (they persist between frames in the reference)
    int week = 0, tow_ms = 0, gpssec = 0, year = 0, month = 0, day = 0, wday = 0, hour = 0, minute = 0; float sec = 0.f;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    float T = 0, RH = 0, Ti = 0, batV = 0;
    uint8_t numSV = 0, utc_ofs = 0; char SN[12] = {0};

    int trimble() {
        int err = 0;
        numSV = fb[0x1E]; utc_ofs = fb[0x1F];
        {
            int w = (fb[0x20] << 8) + fb[0x21];
            if (w > 4000) err = -1;
            else { if (w < 1304) w += 1024; week = w; }
        }
        {
            int t = be32(fb + 0x0A);
            tow_ms = t;
            const int ms = t % 1000;
            t /= 1000;
            gpssec = t;
            const int d = t / 86400;
            if (d < 0 || d > 6) err = -1;
            else {
                t %= 86400;
                wday = d; hour = t / 3600; minute = (t % 3600) / 60; sec = (float)(t % 60 + ms / 1000.0);
            }
        }
        const double unit = (1 << 30) / 90.0;                  // 2^32 / 360
        lat = be32(fb + 0x0E) / unit;
        lon = be32(fb + 0x12) / unit;
        alt = be32(fb + 0x16) / 1000.0;
        const double vx = be16(fb + 0x04) / 2e2, vy = be16(fb + 0x06) / 2e2;
        vH = sqrt(vx * vx + vy * vy);
        double dir = atan2(vx, vy) * 180 / M_PI;
        if (dir < 0) dir += 360;
        vD = dir;
        vV = be16(fb + 0x08) / 2e2;
        return err;
    }
    void gtop() {
        int t = fb[0x15] << 16 | fb[0x16] << 8 | fb[0x17];
        hour = t / 10000; minute = (t % 10000) / 100; sec = (float)((t % 100) / 1.0);
        const int d = fb[0x18] << 16 | fb[0x19] << 8 | fb[0x1A];
        year = 2000 + d % 100; month = (d % 10000) / 100; day = d / 10000;
        lat = be32(fb + 0x04) / 1e6;
        lon = be32(fb + 0x08) / 1e6;
        int a = fb[0x0C] << 16 | fb[0x0D] << 8 | fb[0x0E];
        if (a & 0x800000) a -= 0x1000000;
        alt = a / 1e2;
        const double vx = be16(fb + 0x0F) / 1e2, vy = be16(fb + 0x11) / 1e2;
        vH = sqrt(vx * vx + vy * vy);
        double dir = atan2(vx, vy) * 180 / M_PI;
        if (dir < 0) dir += 360;
        vD = dir;
        vV = be16(fb + 0x13) / 1e2;
    }
    void gps_date(long wk, long secs, int *yy, int *mm, int *dd) const {
        const long days = wk * 7 + secs / 86400, mjd = 44244 + days;
        long J = mjd + 2468570;
        const long C = 4 * J / 146097;
        J = J - (146097 * C + 3) / 4;
        const long Y = 4000 * (J + 1) / 1461001;
        J = J - 1461 * Y / 4 + 31;
        const long M = 80 * J / 2447;
        *dd = (int)(J - 2447 * M / 80);
        J = M / 11;
        *mm = (int)(M + 2 - 12 * J);
        *yy = (int)(100 * (C - 49) + Y + J);
    }
    void serial() {
        for (int i = 0; i < 11; i++) SN[i] = ' ';
        SN[11] = 0;
        const uint8_t *r = fb + 0x5D;
        unsigned b = r[2];
        sprintf(SN, "%1X%02u", (b >> 4) & 0xF, b & 0xF);
        b = r[3] | (r[4] << 8);
        sprintf(SN + 3, " %1X %1u%04u", r[0] & 0xF, (b >> 13) & 0x7, b & 0x1FFF);
    }
    // NTC thermistor behind a three-range voltage divider, cubic fit in ln R
    float temp() const {
        const float p0 = 1.07303516e-03f, p1 = 2.41296733e-04f, p2 = 2.26744154e-06f, p3 = 6.52855181e-08f;
        const float Rs[3] = { 12.1e3f, 36.5e3f, 475.0e3f }, Rp[3] = { 1e20f, 330.0e3f, 2000.0e3f };
        const uint8_t sc = fb[0x3E];
        uint16_t adc = (uint16_t)((fb[0x40] << 8) | fb[0x3F]);
        adc = (uint16_t)(adc - 0xA000);
        const float adc_max = 4095.0f;
        const float x = (adc_max - adc) / adc;
        float R, Tk = 0;
        if (sc < 3) R = Rs[sc] / (x - Rs[sc] / Rp[sc]); else R = -1;
        if (R > 0) Tk = (float)(1 / (p0 + p1 * log(R) + p2 * log(R) * log(R) + p3 * log(R) * log(R) * log(R)));
        return (float)(Tk - 273.15);
    }
    float temp_ntc2() const {
        const float Rs = 22.1e3f, p0 = 4.42606809e-03f, p1 = -6.58184309e-04f, p2 = 8.95735557e-05f, p3 = -2.84347503e-06f;
        float Tk = 0.0f;
        const uint16_t adc = (uint16_t)((fb[0x5A] << 8) | fb[0x59]);
        const float x = (float)((4095.0 - adc) / adc);
        const float R = Rs / x;
        if (R > 0) Tk = (float)(1 / (p0 + p1 * log(R) + p2 * log(R) * log(R) + p3 * log(R) * log(R) * log(R)));
        return (float)(Tk - 273.15);
    }
    float count55() const { const uint32_t v = fb[0x32] | (fb[0x33] << 8) | (fb[0x34] << 16); return (float)(v / 1000.0); }
    float countRH() const { const uint32_t v = fb[0x35] | (fb[0x36] << 8) | (fb[0x37] << 16); return (float)(v / 1000.0); }
    float humidity() const {
        const float ratio = countRH() / count55();
        const float Tc = temp();
        float rh = (float)((ratio - 0.8955) / 0.002);
        const float T0 = 0.0f, T1 = -30.0f;
        if (Tc < T0) rh = (float)(rh + (T0 - Tc / 5.5));
        if (Tc < T1) rh = (float)(rh * (1.0 + (T1 - Tc) / 75.0));
        if (rh < 0.0) rh = 0.0f;
        if (rh > 100.0) rh = 100.0f;
        return rh;
    }
    // On-chip temperature diode.  The reference's get_intTemp() (m10mod.c:693-705) computes (raw/4095*1.5 - 0.986)/0.00355 but has no
    // return statement, and its caller stores the (undefined) return value over the computed one: the compiled reference prints
    // "(Ti:0.0C)" at -vvv whatever the diode reads.  Mirrored as 0 so that the -vvv line stays byte-identical; the physical value is
    // available through int_temp_physical().
    float int_temp() const { return 0.0f; }
    float int_temp_physical() const {
        const uint16_t raw = (uint16_t)((fb[0x49] << 8) | fb[0x48]);
        const float v = (float)(raw / 4095.0 * 1.5);
        return (float)((v - 0.986) / 0.00355);
    }
    float battery() const {
        const uint32_t adc = (uint32_t)((fb[0x46] << 8) | fb[0x45]);
        const float v = (float)(2.709 * adc * 2.5 / 1023.0);
        return (float)(double)v;
    }

    void print(Out &w, int csOK) {
        int err = 0, err2 = 0;
        if (type == 0x9F) err = trimble();
        else if (type == 0xAF) gtop();
        else err = 0xFF;
        if (err) return;
        if (type == 0x9F) gps_date(week, gpssec, &year, &month, &day);
        T = temp(); RH = humidity(); Ti = int_temp(); batV = battery();
        serial();
        if (!o.silent) {
            // -c: the same line with ANSI colours around the fields (m10mod.c:253-277,886-923); empty strings without it
            const bool c = o.color != 0;
            const char *TXT = c ? "\x1b[38;5;244m" : "", *WK = c ? "\x1b[38;5;20m" : "", *TOW = c ? "\x1b[38;5;27m" : "", *DAT = c ? "\x1b[38;5;94m" : "";
            const char *LAT = c ? "\x1b[38;5;34m" : "", *LON = c ? "\x1b[38;5;70m" : "", *ALT = c ? "\x1b[38;5;82m" : "", *VEL = c ? "\x1b[38;5;36m" : "", *SNC = c ? "\x1b[38;5;58m" : "";
            const char *OKC = c ? "\x1b[38;5;2m" : "", *NOC = c ? "\x1b[38;5;1m" : "";
            w.f("%s", TXT);
            if (type == 0x9F) {
                if (o.verbose >= 3) w.f(" (W %s%d%s) ", WK, week, TXT);
                w.f("%s%s%s ", TOW, kDay[wday], TXT);
            }
            w.f("%s%04d-%02d-%02d%s %s%02d:%02d:%06.3f%s ", DAT, year, month, day, TXT, TOW, hour, minute, sec, TXT);
            w.f(" lat: %s%.5f%s ", LAT, lat, TXT); w.f(" lon: %s%.5f%s ", LON, lon, TXT); w.f(" alt: %s%.2f%s ", ALT, alt, TXT);
            if (!err2) w.f("  vH: %s%.1f%s  D: %s%.1f%s  vV: %s%.1f%s ", VEL, vH, TXT, VEL, vD, TXT, VEL, vV, TXT);
            if (o.verbose >= 2) w.f("  SN: %s%s%s", SNC, SN, TXT);
            if (o.verbose >= 2) { w.f("  # "); if (csOK) w.f(" %s[OK]%s", OKC, TXT); else w.f(" %s[NO]%s", NOC, TXT); }
            if (o.ptu && csOK) {
                if (T > -270.0) w.f("  T=%.1fC", T);
                if (o.verbose >= 2 && RH > -0.5) w.f(" _RH=%.0f%%", RH);
                if (o.verbose >= 3) {
                    const float t2 = temp_ntc2();
                    const float fq = (float)((8e6 / 2) / countRH());
                    w.f("  (Ti:%.1fC)", Ti);
                    if (t2 > -270.0) w.f(" (T2:%.1fC) (%.3fkHz)", t2, fq / 1e3);
                }
            }
            if (o.verbose >= 3 && csOK) w.f(" (bat:%.2fV)", batV);
            if (c) w.f("\x1b[0m");
            w.f("\n");
        }
        if (o.json && csOK) {
            const double sec_gps0 = (double)week * 604800.0 + tow_ms / 1e3;
            int utc_s = gpssec - utc_ofs, utc_week = week, uy, um, ud, uh, umi; float us;
            if (utc_s < 0) { utc_week -= 1; utc_s += 604800; }
            if (type == 0x9F) {
                gps_date(utc_week, utc_s, &uy, &um, &ud);
                utc_s %= 86400;
                uh = utc_s / 3600; umi = (utc_s % 3600) / 60; us = (float)(utc_s % 60 + (tow_ms % 1000) / 1000.0);
            } else { uy = year; um = month; ud = day; uh = hour; umi = minute; us = sec; }
            char id[16] = "M10-";
            strncpy(id + 4, SN, 12); id[15] = 0;
            for (int j = 0; id[j]; j++) if (id[j] == ' ') id[j] = '-';
            w.f("{ \"type\": \"%s\"", "M10");
            w.f(", \"frame\": %lu, ", (unsigned long)(sec_gps0 + 0.5));
            w.f("\"id\": \"%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, "
                "\"heading\": %.5f, \"vel_v\": %.5f", id, uy, um, ud, uh, umi, us, lat, lon, alt, vH, vD, vV);
            if (type == 0x9F) w.f(", \"sats\": %d", numSV);
            const uint8_t *r = fb + 0x5D;
            w.f(", \"aprsid\": \"ME%02X%1X%02X%02X\"", r[2], r[0] & 0xF, r[4], r[3]);
            w.f(", \"batt\": %.2f", batV);
            if (o.ptu) {
                if (T > -273.0) w.f(", \"temp\": %.1f", T);
                if (o.verbose >= 2 && RH > -0.5) w.f(", \"humidity\": %.1f", RH);
            }
            w.f(", \"rawid\": \"M10_%02X%02X%02X%02X%02X\"", r[0], r[1], r[2], r[3], r[4]);
            w.f(", \"subtype\": \"0x%02X\"", type);
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            w.f(", \"ref_datetime\": \"%s\"", "UTC");
            w.f(", \"ref_position\": \"%s\"", "GPS");
            w.f(", \"gpsutc_leapsec\": %d", utc_ofs);
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
            w.f("\n");
        }
    }
};

extern "C" {

int sonde_m10_dec_create(const sonde_m10_opts_t *opts, sonde_m10_dec_t **out) {
    if (!opts || !out || opts->verbose < 0 || opts->verbose > 3) return SONDE_E_ARG;
    sonde_m10_dec *d = new sonde_m10_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.raw && d->o.json) d->o.silent = 1;
    *out = d;
    return 0;
}
void sonde_m10_dec_destroy(sonde_m10_dec_t *d) { delete d; }

int sonde_m10_dec_frame(sonde_m10_dec_t *d, const sonde_m10_frame_t *f, char *out, size_t outlen) {
    if (!d || !f || !out || outlen < 1) return SONDE_E_ARG;
    d->fb = f->frame;
    switch (f->frame[1]) {                                   // m10mod.c:1068-1074
        case 0x8F: d->type = 0x8F; break;
        case 0xAF: d->type = 0xAF; break;
        case 0x20: d->type = 0x20; break;
        default: d->type = 0x9F; break;
    }
    Out w;
    if (d->o.raw) { if (f->frame[1] != 0x49 && d->o.silent) d->print(w, f->cs_ok); }
    else if (f->frame[1] == 0x49) {                          // satellite signal-level frame: hex dump at -vvv only
        if (d->o.verbose == 3) { for (int i = 0; i < f->len; i++) w.f("%02x", f->frame[i]); w.f(f->cs_ok ? " [OK]" : " [NO]"); w.f("\n"); }
    } else d->print(w, f->cs_ok);
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

}  // extern "C"
