// sonde_m20_fields.cpp — M20 telemetry and the text / JSON lines of the reference's m20mod print_pos() (include/sonde_m20.h).
//
//   GPS: time of week in seconds (3 bytes) + week with the rollover repair, frame counter difference, lat / lon in 1e-6 degrees,
//        unsigned 24-bit altitude in cm, E / N / U velocities in cm/s ........................... Decoder::gps()        m20mod.c:268-452
//   serial number text (year, month, line, number; an all-zero serial shows the counter difference) . Decoder::serial()  :454-484
//   thermistor with the range taken from the ADC word, humidity-sensor NTC (beta model),
//        humidity (capacitance word against its calibration word, temperature-compensated cubic),
//        optional pressure word, battery ...................................................... Decoder::temp() ...   :564-725
//   text line and JSON ........................................................................ Decoder::print()       :729-868
#include "../../include/sonde_m20.h"
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

struct sonde_m20_dec {
    sonde_m20_opts_t o;
    const uint8_t *fb = nullptr;
    int type = 0x20, fw = 0;
    uint32_t gps_cnt = 0; uint8_t cnt = 0, diffcnt = 0;
    int week = 0, tow_ms = 0, gpssec = 0, year = 0, month = 0, day = 0, wday = 0, hour = 0, minute = 0; float sec = 0.f;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    float T = 0, RH = 0, TH = 0, P = 0, batV = 0;
    char SN[16] = {0};

    int gps() {
        int err = 0;
        int t = fb[0x0F] << 16 | fb[0x10] << 8 | fb[0x11];
        tow_ms = (int)((uint32_t)t * 1000u);             // 24-bit seconds * 1000 wraps in the reference's int for garbage frames; same bits, defined
        gpssec = t;
        const int d = t / 86400;
        if (d < 0 || d > 6) err = -1;
        else {
            t %= 86400;
            wday = d; hour = t / 3600; minute = (t % 3600) / 60; sec = (float)(t % 60 + 0 / 1000.0);
            int w = (fb[0x1A] << 8) + fb[0x1B];
            if (w > 4000) err = -1;
            else {
                if (w < 1304) w += 1024;
                week = w;
                const double s0 = (double)week * 604800.0 + tow_ms / 1e3;
                gps_cnt = (uint32_t)(s0 + 0.5);
                cnt = fb[0x15];
                diffcnt = (uint8_t)(gps_cnt - cnt);
            }
        }
        lat = be32(fb + 0x1C) / 1e6;
        lon = be32(fb + 0x20) / 1e6;
        alt = (fb[0x08] << 16 | fb[0x09] << 8 | fb[0x0A]) / 100.0;
        const double vx = be16(fb + 0x0B) / 1e2, vy = be16(fb + 0x0D) / 1e2;
        vH = sqrt(vx * vx + vy * vy);
        double dir = atan2(vx, vy) * 180 / M_PI;
        if (dir < 0) dir += 360;
        vD = dir;
        vV = be16(fb + 0x18) / 1e2;
        return err;
    }
    void gps_date() {
        const long days = (long)week * 7 + gpssec / 86400, mjd = 44244 + days;
        long J = mjd + 2468570;
        const long C = 4 * J / 146097;
        J = J - (146097 * C + 3) / 4;
        const long Y = 4000 * (J + 1) / 1461001;
        J = J - 1461 * Y / 4 + 31;
        const long M = 80 * J / 2447;
        day = (int)(J - 2447 * M / 80);
        J = M / 11;
        month = (int)(M + 2 - 12 * J);
        year = (int)(100 * (C - 49) + Y + J);
    }
    uint8_t snraw[3] = { 0, 0, 0 };            // gpx_t.SNraw: sits directly in front of frame_bytes in the reference (m20mod.c:112-113)
    void serial() {
        snraw[0] = fb[0x12]; snraw[1] = fb[0x13]; snraw[2] = fb[0x14];
        const uint32_t sn = (uint32_t)(fb[0x14] << 16 | fb[0x13] << 8 | fb[0x12]);
        const unsigned ym = sn & 0x7F, y = (ym / 12) & 0xFF, m = ((ym % 12) + 1) & 0xFF;
        for (int i = 0; i < 11; i++) SN[i] = ' ';
        SN[11] = 0;
        for (int i = 12; i < 16; i++) SN[i] = 0;
        sprintf(SN, "%u%02u", y, m);
        sprintf(SN + 3, "-%u-", ((sn >> 7) & 0x7) + 1);
        sprintf(SN + 6, "%u", (sn >> 23) & 0x1);
        sprintf(SN + 7, "%04u", (sn >> 10) & 0x1FFF);
        if (sn == 0) { sprintf(SN, "%s", "000-0-00000"); sprintf(SN + 11, "-%03u", diffcnt & 0xFF); }
    }
    float temp() const {
        const float p0 = 1.07303516e-03f, p1 = 2.41296733e-04f, p2 = 2.26744154e-06f, p3 = 6.52855181e-08f;
        const float Rs[3] = { 12.1e3f, 36.5e3f, 475.0e3f }, Rp[3] = { 1e20f, 330.0e3f, 2000.0e3f };
        uint16_t adc = (uint16_t)((fb[0x5] << 8) | fb[0x4]);
        int sc = 0;
        if (adc > 8191) { sc = 2; adc = (uint16_t)(adc - 8192); }
        else if (adc > 4095) { sc = 1; adc = (uint16_t)(adc - 4096); }
        const float x = (float)((4095.0 - adc) / adc);
        const float R = Rs[sc] / (x - Rs[sc] / Rp[sc]);
        float Tk = 0;
        if (R > 0) Tk = (float)(1.0 / (p0 + p1 * log(R) + p2 * log(R) * log(R) + p3 * log(R) * log(R) * log(R)));
        if (Tk - 273.15 < -120.0 || Tk - 273.15 > 60.0) Tk = 0;
        return (float)(Tk - 273.15);
    }
    float temp_rh_sensor() const {
        const float Rs = 22.1e3f, R25 = 2.2e3f, b = 3650.0f, T25 = (float)(25.0 + 273.15);
        float Tk = 0.0f;
        const uint16_t adc = (uint16_t)((fb[0x07] << 8) | fb[0x06]);
        const float x = (float)((4095.0 - adc) / adc);
        const float R = Rs / x;
        if (R > 0) Tk = (float)(1.0 / (1.0 / T25 + 1.0 / b * log(R / R25)));
        return (float)(Tk - 273.15);
    }
    float humidity() const {
        const float TU = temp_rh_sensor();
        const uint16_t hum = (uint16_t)((fb[0x03] << 8) | fb[0x02]), cal = (uint16_t)((fb[0x30] << 8) | fb[0x2F]);
        const float k = 6.4e8f / (cal + 80000.0f);
        float x = (hum + 80000.0f) * k * (1.0f - 5.8e-4f * (TU - 25.0f));
        x = 4.16e9f / x;
        x = 10.087f * x * x * x - 211.62f * x * x + 1388.2f * x - 2797.0f;
        float rh = -1.0f;
        if (hum < 48000 && x > -20.0f && x < 120.f) { rh = x; if (rh < 0.0f) rh = 0.0f; if (rh > 100.0f) rh = 100.0f; }
        return rh;
    }
    float pressure() const {
        float hPa = 0.0f;
        uint32_t v = (uint32_t)((fb[0x25] << 8) | fb[0x24]);
        const uint8_t p0 = fw >= 0x07 ? fb[0x16] : 0;
        v = (v << 8) | p0;
        if (v > 0) hPa = v / (float)(16 * 256);
        if (hPa > 2560.0f) hPa = -1.0f;
        return hPa;
    }

    void print(Out &w, int bcOK, int csOK) {
        const int err = gps();
        if (err) return;
        gps_date();
        serial();
        if (o.ptu && csOK) { T = temp(); TH = temp_rh_sensor(); RH = humidity(); P = pressure(); }
        batV = fb[0x26] * (3.3f / 255);
        if (!o.silent) {
            // -c: ANSI colours around the fields (COLOPT, m20mod.c:239-267); empty strings without it
            const bool c = o.color != 0;
            const char *TXT = c ? "\x1b[38;5;244m" : "", *WK = c ? "\x1b[38;5;20m" : "", *TOW = c ? "\x1b[38;5;27m" : "", *DAT = c ? "\x1b[38;5;94m" : "";
            const char *LAT = c ? "\x1b[38;5;34m" : "", *LON = c ? "\x1b[38;5;70m" : "", *ALT = c ? "\x1b[38;5;82m" : "", *VEL = c ? "\x1b[38;5;36m" : "", *SNC = c ? "\x1b[38;5;58m" : "";
            const char *OKC = c ? "\x1b[38;5;2m" : "", *OOC = c ? "\x1b[38;5;220m" : "", *NOC = c ? "\x1b[38;5;1m" : "", *RST = c ? "\x1b[0m" : "";
            w.f("%s", TXT);
            if (o.verbose >= 3) { w.f("[%3d]", fb[0x15]); w.f(" (W %s%d%s) ", WK, week, TXT); }
            w.f("%s%s%s ", TOW, kDay[wday], TXT);
            w.f("%s%04d-%02d-%02d%s %s%02d:%02d:%06.3f%s ", DAT, year, month, day, TXT, TOW, hour, minute, sec, TXT);
            w.f(" lat: %s%.5f%s ", LAT, lat, TXT); w.f(" lon: %s%.5f%s ", LON, lon, TXT); w.f(" alt: %s%.2f%s ", ALT, alt, TXT);
            w.f("  vH: %s%4.1f%s  D: %s%5.1f%s  vV: %s%3.1f%s ", VEL, vH, TXT, VEL, vD, TXT, VEL, vV, TXT);
            if (o.verbose >= 1 && (bcOK || csOK)) w.f("  SN: %s%s%s", SNC, SN, TXT);
            if (o.verbose >= 1) {
                w.f("  # ");
                if (fw < 0x07) { if (bcOK > 0) w.f(" %s(ok)%s", OKC, TXT); else if (bcOK < 0) w.f(" %s(oo)%s", OOC, TXT); else w.f(" %s(no)%s", NOC, TXT); }
                if (csOK) w.f(" %s[OK]%s", OKC, TXT); else w.f(" %s[NO]%s", NOC, TXT);
            }
            if (o.ptu && csOK) {
                w.f(" ");
                if (T > -273.0f) w.f(" T:%.1fC", T);
                if (RH > -0.5f) w.f(" RH=%.0f%%", RH);
                if (o.verbose >= 2 && TH > -273.0f) w.f(" TH:%.1fC", TH);
                if (P > 0.0f) { if (P < 10.0f) w.f(" P=%.3fhPa ", P); else if (P < 100.0f) w.f(" P=%.2fhPa ", P); else w.f(" P=%.1fhPa ", P); }
            }
            if (o.verbose >= 3 && csOK) w.f(" (bat:%.2fV)", batV);
            w.f("%s", RST);
            w.f("\n");
        }
        if (o.json && csOK) {
            char id[20] = "M20-";
            strncpy(id + 4, SN, 16); id[19] = 0;
            w.f("{ \"type\": \"%s\"", "M20");
            w.f(", \"frame\": %lu, ", (unsigned long)gps_cnt);
            w.f("\"id\": \"%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, "
                "\"heading\": %.5f, \"vel_v\": %.5f", id, year, month, day, hour, minute, sec, lat, lon, alt, vH, vD, vV);
            if (o.ptu) {
                if (T > -273.0f) w.f(", \"temp\": %.1f", T);
                if (RH > -0.5f) w.f(", \"humidity\": %.1f", RH);
                if (P > 0.0f) w.f(", \"pressure\": %.2f", P);
            }
            w.f(", \"batt\": %.2f", batV);
            w.f(", \"rawid\": \"M20_%02X%02X%02X\"", fb[0x12], fb[0x13], fb[0x14]);
            w.f(", \"subtype\": \"0x%02X\"", type);
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            w.f(", \"ref_datetime\": \"%s\"", "GPS");
            w.f(", \"ref_position\": \"%s\"", "GPS");
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
            w.f("\n");
        }
    }
};

extern "C" {

int sonde_m20_dec_create(const sonde_m20_opts_t *opts, sonde_m20_dec_t **out) {
    if (!opts || !out || opts->verbose < 0 || opts->verbose > 3) return SONDE_E_ARG;
    sonde_m20_dec *d = new sonde_m20_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.raw && d->o.json) d->o.silent = 1;
    d->T = d->TH = -273.15f; d->RH = -1.0f; d->P = -1.0f;
    *out = d;
    return 0;
}
void sonde_m20_dec_destroy(sonde_m20_dec_t *d) { delete d; }

int sonde_m20_dec_frame(sonde_m20_dec_t *d, const sonde_m20_frame_t *f, char *out, size_t outlen) {
    if (!d || !f || !out || outlen < 1) return SONDE_E_ARG;
    d->fb = f->frame;
    d->fw = f->fw;
    int cs_ok = f->cs_ok;
    if (f->frame[0] == 0) {                    // length byte 0: check word and firmware byte are read from in front of the frame buffer (:897-901) =
        cs_ok = d->snraw[2] == 0;              // the serial number bytes of the last frame that was printed
        d->fw = d->snraw[1] > 0x20 ? 0 : d->snraw[1];
    }
    switch (f->frame[1]) { case 0x8F: d->type = 0x8F; break; case 0x9F: d->type = 0x9F; break; case 0xAF: d->type = 0xAF; break; case 0x20: d->type = 0x20; break; default: d->type = 0x9F; }
    Out w;
    if (!d->o.raw || d->o.silent) d->print(w, f->blk_ok, cs_ok);
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

}  // extern "C"
