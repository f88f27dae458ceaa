// sonde_rs41_fields.cpp — RS41 telemetry fields and the text / JSON lines of the reference's print_position()
// (include/sonde_rs41.h).  Host side, bit-rate work: one 320 / 518 byte frame per second per sonde.
//
// What the reference does per frame (rs41mod.c:2126-2470) and where it is restated here:
//   block walk with per-block CRC16 ............................. Decoder::run_good()        :2163-2300
//   status block: frame number, ID, battery, cal subframe ....... Decoder::status_block()    :417-552
//   GPS week / time of week -> calendar date .................... Decoder::gps_time()        :204-222,917-990
//   ECEF position / velocity -> lat, lon, alt, vH, heading, vV .. Decoder::ecef()            :1014-1096
//   combined position + UTC date/time block (0x8226) ............ Decoder::pos_datetime()    :1120-1158
//   GNSS satellite block (0x8329): satellite count .............. Decoder::gnss_sats()       :1161-1218
//   PTU: calibration coefficients out of the subframe table,
//        T / TH (platinum resistor), RH (capacitor; empirical and
//        calibrated model), P (sensor or barometric estimate) ... Decoder::ptu()             :553-916
//   configuration subframes: frequency, firmware, sub-type,
//        kill / burst timers .................................... Decoder::conf_subframe()   :1551-1667
//   xdata (0x7E) blocks -> "aux" string ......................... Decoder::xdata()           :1466-1550
//   frames whose ECC failed: blocks with good CRC, CRC flags .... Decoder::run_failed()      :2401-2466
//   JSON object ................................................. Decoder::json()            :2302-2394
// Arithmetic keeps the reference's types step by step (float where it computes in float, double where C promotes), because
// the printed digits are compared with the reference's.
#include "../../include/sonde_rs41.h"
#include "../../include/sonde_ecc.h"
#include "sonde_host.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

using namespace sonde;

namespace {

// block positions of a standard frame (rs41mod.c:336-400)
constexpr int P_STATUS = 0x039, P_PTU = 0x065, P_GPS1 = 0x093, P_GPS2 = 0x0B5, P_GPS3 = 0x112, P_ZERO = 0x12B;
constexpr int P_FRNR = 0x03B, P_ID = 0x03D, P_BATT = 0x045, P_CAL = 0x052, P_WEEK = 0x095, P_ITOW = 0x097, P_NSAT = 0x126;
constexpr int K_STATUS = 0x7928, K_PTU = 0x7A2A, K_GPS1 = 0x7C1E, K_GPS2 = 0x7D59, K_GPS3 = 0x7B15, K_XTU = 0x7F1B,
              K_CRYPT = 0x80A7, K_960A = 0x960A, K_POSDT = 0x8226, K_SATS = 0x8329;
constexpr int FL = 518, NDATA = 320;
enum { F_STATUS = 1, F_PTU = 2, F_GPS1 = 4, F_GPS2 = 8, F_GPS3 = 16, F_AUX = 32, F_ZERO = 64 };

struct Out {                       // stdout of the reference, collected
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[2048];
        va_list ap; va_start(ap, fmt);
        const int n = vsnprintf(b, sizeof b, fmt, ap);
        va_end(ap);
        if (n > 0) s.append(b, (size_t)std::min<int>(n, (int)sizeof b - 1));
    }
};

inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
inline int s16(const uint8_t *p) { int v = (int)le16(p); return (v & 0x8000) ? v - 0x10000 : v; }
inline float f32at(const uint8_t *p) { float v; memcpy(&v, p, 4); return v; }

const char *const kDay[7] = { "Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat" };

}  // namespace

struct sonde_rs41_dec {
    sonde_rs41_opts_t o;
    uint8_t fr[FL];
    // gpx_t members that survive a frame
    int frnr = 0; char id[9] = {0};
    int numSV = 0, isUTC = 0, week = 0, gpssec = 0, year = 0, month = 0, day = 0, wday = 0, hour = 0, minute = 0; float sec = 0.f;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    float T = 0, RH = 0, TH = 0, P = 0, RH2 = 0;
    unsigned crc = 0;
    uint8_t cal[51 * 16]; uint8_t have[51]; int cal_complete = 0, cal_sent = 0; int subfrm_pos = P_CAL;
    uint32_t freq = 0; float batt = 0; uint16_t fw = 0, kt = 0, bt = 0, cd = 0; uint8_t bk = 0;
    char rstyp[10], rstmp[10], rsm[10];
    int aux = 0; char xd[198 + 16];
    // PTU coefficients (get_CalData)
    float Rf1, Rf2, co1[3], calT1[3], co2[3], calT2[3], calH[2], mtxH[42], corHp[3], corHt[12], Cf1, Cf2, calP[25];
    int have_id = 0, have_time = 0, have_pos = 0;
    // --ecc3 / --ecc4 (ecdat_t, rs41mod.c:91-100): time stamps of the last good frame number / calibration subframe, and the frame
    // image that persists from frame to frame (gpx->frame: bytes a short read does not reach keep their old content)
    float ec_ts = 0.f, ec_last_frnb_ts = 0.f, ec_last_calfrm_ts = 0.f; uint16_t ec_last_frnb = 0; uint8_t ec_last_calfrm = 0;
    uint8_t ec_frame[FL]; bool ec_frame_init = false;
    float ec_score[FL + 8]; uint8_t ec_bitscore[FL];        // gpx.ecdat.frm_bytescore / gpx.dfrm_bitscore: bytes not read keep the last frame's

    int block_crc(int pos, int kind) const {                 // 0 ok, 1 mismatch, -1 not that block / does not fit
        if (((kind >> 8) & 0xFF) != fr[pos]) return -1;
        const int n = fr[pos + 1];
        if (pos + n + 4 > FL) return -1;
        return (int)le16(fr + pos + 2 + n) != crc16(fr + pos + 2, n) ? 1 : 0;
    }
    int frametype() const { int t = 0; const uint8_t b = fr[P_STATUS - 1]; for (int i = 0; i < 4; i++) t += ((b >> i) & 1) - ((b >> (i + 4)) & 1); return t; }

    // ---- status block ------------------------------------------------------------------------------------------------
    int status_block(int ofs) {
        const int bad = block_crc(P_STATUS + ofs, K_STATUS);
        if (bad) crc |= F_STATUS;
        if (bad == 0) {
            char nid[9]; memcpy(nid, fr + P_ID + ofs, 8); nid[8] = 0;
            if (strncmp(id, nid, 8) != 0) {                   // another sonde: forget everything derived from the old one
                memset(have, 0, sizeof have);
                memset(rstyp, 0, sizeof rstyp); memset(rstmp, 0, sizeof rstmp); memset(rsm, 0, sizeof rsm);
                cal_complete = 0; cal_sent = 0; freq = 0; fw = 0; bt = 0; bk = 0; cd = (uint16_t)-1; kt = (uint16_t)-1;
                year = month = day = hour = minute = 0; sec = 0.f; week = 0;
                lat = lon = alt = vH = vD = vV = 0.0; numSV = 0; isUTC = 0;
                T = -273.15f; RH = -1.0f; P = -1.0f; RH2 = -1.0f;
                memcpy(id, nid, 9);
                ec_last_frnb = 0;                             // (get_SondeID :503)
            }
        }
        frnr = (int)le16(fr + P_FRNR + ofs);
        if (bad == 0) { ec_last_frnb = (uint16_t)frnr; ec_last_frnb_ts = ec_ts; }           // (get_FrameNb :432-435)
        batt = (float)((uint16_t)fr[P_BATT + ofs] / 10.0);
        if (bad == 0) {
            const int k = fr[P_CAL + ofs];
            ec_last_calfrm = (uint8_t)k; ec_last_calfrm_ts = ec_ts;                         // (get_FrameConf :533-534)
            if (k < 51 && !have[k]) { memcpy(cal + 16 * k, fr + P_CAL + ofs + 1, 16); have[k] = 1; }
            if (!cal_complete) {
                int n = 0; for (int i = 0; i < 51; i++) n += have[i];
                if (n == 51 && (int)le16(cal) == crc16(cal + 2, 50 * 16 - 2)) cal_complete = 1;
            }
        }
        return bad;
    }

    // ---- time --------------------------------------------------------------------------------------------------------
    void gps_date() {                                          // GPS week + seconds -> calendar date via the MJD (:204-222)
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
    int gps_time(int ofs) {
        const int bad = block_crc(P_GPS1 + ofs, K_GPS1);
        if (bad) { crc |= F_GPS1; year = month = day = hour = minute = 0; sec = 0.f; isUTC = 0; return -1; }
        week = (int)le16(fr + P_WEEK + ofs);
        int32_t t; memcpy(&t, fr + P_ITOW + ofs, 4);
        const int ms = t % 1000;
        t /= 1000;
        gpssec = t;
        wday = (t / 86400) % 7;
        t %= 86400;
        hour = t / 3600; minute = (t % 3600) / 60; sec = (float)(t % 60 + ms / 1000.0);
        isUTC = 0;
        return 0;
    }

    // ---- position ----------------------------------------------------------------------------------------------------
    int ecef(int pos) {                                        // -3: altitude outside -1 .. 80 km (e.g. all-zero ECEF)
        double X[3], V[3];
        for (int k = 0; k < 3; k++) {
            int32_t c; memcpy(&c, fr + pos + 4 * k, 4);
            X[k] = c / 100.0;
            V[k] = (short)(fr[pos + 12 + 2 * k] | fr[pos + 13 + 2 * k] << 8) / 100.0;
        }
        const double a = 6378137.0, b = 6356752.31424518, a2b2 = a * a - b * b, e2 = a2b2 / (a * a), ee2 = a2b2 / (b * b);
        const double lam = atan2(X[1], X[0]);
        const double p = sqrt(X[0] * X[0] + X[1] * X[1]);
        const double t = atan2(X[2] * a, p * b);
        const double phi = atan2(X[2] + ee2 * b * sin(t) * sin(t) * sin(t), p - e2 * a * cos(t) * cos(t) * cos(t));
        const double R = a / sqrt(1 - e2 * sin(phi) * sin(phi));
        alt = p / cos(phi) - R;
        lat = phi * 180 / M_PI;
        lon = lam * 180 / M_PI;
        if (alt < -1000 || alt > 80000) return -3;
        const double ph = lat * M_PI / 180.0, la = lon * M_PI / 180.0;
        const double vN = -V[0] * sin(ph) * cos(la) - V[1] * sin(ph) * sin(la) + V[2] * cos(ph);
        const double vE = -V[0] * sin(la) + V[1] * cos(la);
        const double vU = V[0] * cos(ph) * cos(la) + V[1] * cos(ph) * sin(la) + V[2] * sin(ph);
        vH = sqrt(vN * vN + vE * vE);
        double dir = atan2(vE, vN) * 180 / M_PI;
        if (dir < 0) dir += 360;
        vD = dir;
        vV = vU;
        return 0;
    }
    int gps_pos(int ofs) {
        if (block_crc(P_GPS3 + ofs, K_GPS3)) { crc |= F_GPS3; lat = lon = alt = vH = vD = vV = 0.0; numSV = 0; return -1; }
        const int e = ecef(P_GPS3 + ofs + 2);
        numSV = fr[P_NSAT + ofs];
        return e;
    }
    int pos_datetime(int pos) {
        if (block_crc(pos, K_POSDT)) {
            crc |= F_GPS1 | F_GPS3;
            year = month = day = hour = minute = 0; sec = 0.f; isUTC = 0;
            lat = lon = alt = vH = vD = vV = 0.0; numSV = 0;
            return -1;
        }
        const int e = ecef(pos + 2);
        year = fr[pos + 20] | fr[pos + 21] << 8; month = fr[pos + 22]; day = fr[pos + 23];
        hour = fr[pos + 24]; minute = fr[pos + 25]; sec = fr[pos + 26];
        if (fr[pos + 27] < 100) sec = (float)(sec + fr[pos + 27] / 100.0);
        isUTC = 1;
        return e;
    }
    uint8_t sv_id[32], sv_status[32]; int n_sv168 = 0, n_svstatus = 0;      // gnss_sv[], gnss_numSVb168, gnss_nSVstatus (for --sat)
    int gnss_sats(int pos) {
        memset(sv_id, 0, sizeof sv_id); memset(sv_status, 0, sizeof sv_status); n_sv168 = 0; n_svstatus = 0;
        if (block_crc(pos, K_SATS)) { crc |= F_GPS2; return 1; }
        int c = 0;
        for (int j = 0; j < 21; j++) for (int k = 0; k < 8; k++) if ((fr[pos + 2 + 4 + j] >> k) & 1) { if (c < 32) sv_id[c] = (uint8_t)(j * 8 + k + 1); c++; }
        n_sv168 = c;
        int n = 0;
        for (int j = 0; j < 16; j++) {
            const uint8_t b = fr[pos + 2 + 4 + 21 + j];
            sv_status[2 * j] = b & 0xF; sv_status[2 * j + 1] = (b >> 4) & 0xF;
            if (b & 0xF) n++; if ((b >> 4) & 0xF) n++;
        }
        n_svstatus = n;
        numSV = n;
        return 0;
    }
    void t_gnss_sat2(Out &w) {                                  // prn_gnss_sat2 (:1221-1260)
        w.f("\n"); w.f("  numSV168 : %2d", n_sv168); w.f("  nSVstatus: %2d", n_svstatus); w.f("\n"); w.f("  SVids: ");
        for (int n = 0; n < 32; n++) { if (n < n_sv168) w.f(" %3d", sv_id[n]); if (n < n_svstatus) w.f(":%X", sv_status[n]); }
        w.f("\n");
        for (int n = 0; n < 32; n++) {
            if (!(n < n_sv168 || n < n_svstatus)) continue;
            if (sv_id[n] < 33) { if (n == 0) w.f("  GPS: "); w.f(" PRN%02d", sv_id[n]); }
            else if (sv_id[n] < 33 + 36) {
                if (n == 0 || sv_id[n - 1] < 33) { if (n > 0) w.f("\n"); w.f("  GAL: "); }
                w.f(" E%02d", sv_id[n] - 32);
            }
        }
        w.f("\n");
    }

    // ---- PTU ---------------------------------------------------------------------------------------------------------
    void load_coefficients() {                                 // byte offsets inside the 51 x 16 table (get_CalData :553-617)
        Rf1 = f32at(cal + 61); Rf2 = f32at(cal + 65);
        for (int j = 0; j < 3; j++) { co1[j] = f32at(cal + 77 + 4 * j); calT1[j] = f32at(cal + 89 + 4 * j); }
        calH[0] = f32at(cal + 117); calH[1] = f32at(cal + 121);
        for (int j = 0; j < 3; j++) { co2[j] = f32at(cal + 293 + 4 * j); calT2[j] = f32at(cal + 305 + 4 * j); }
        Cf1 = f32at(cal + 69); Cf2 = f32at(cal + 73);
        for (int j = 0; j < 42; j++) mtxH[j] = f32at(cal + 125 + 4 * j);
        for (int j = 0; j < 3; j++) corHp[j] = f32at(cal + 678 + 4 * j);
        for (int j = 0; j < 12; j++) corHt[j] = f32at(cal + 698 + 4 * j);
        // pressure polynomial: 18 floats from 606 on land in a 6 x 4 grid column by column (+ the scale at [24])
        static const int slot[18] = { 0, 4, 8, 12, 16, 20, 24, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11 };
        for (int j = 0; j < 18; j++) calP[slot[j]] = f32at(cal + 606 + 4 * j);
    }
    float temperature(uint32_t f, uint32_t f1, uint32_t f2, const float *p, const float *c) const {
        const float g = (float)(f2 - f1) / (Rf2 - Rf1);
        const float Rb = (f1 * Rf2 - f2 * Rf1) / (float)(f2 - f1);
        const float Rc = f / g - Rb;
        const float R = Rc * c[0];
        return (float)((p[0] + p[1] * R + p[2] * R * R + c[1]) * (1.0 + c[2]));
    }
    float rh_empirical(uint32_t f, uint32_t f1, uint32_t f2, float Tc) const {
        const float a0 = 7.5f;
        const float a1 = (float)(350.0 / calH[0]);
        const float fh = (f - f1) / (float)(f2 - f1);
        float rh = (float)(100.0 * (a1 * fh - a0));
        const float T0 = 0.0f, T1 = -20.0f, T2 = -40.0f;
        rh = (float)(rh + (T0 - Tc / 5.5));
        if (Tc < T1) rh = (float)(rh * (1.0 + (T1 - Tc) / 100.0));
        if (Tc < T2) rh = (float)(rh * (1.0 + (T2 - Tc) / 120.0));
        if (rh < 0.0) rh = 0.0f;
        if (rh > 100.0) rh = 100.0f;
        if (Tc < -273.0) rh = -1.0f;
        return rh;
    }
    static float sat_vapour(float Tc) {                        // Hyland & Wexler, argument in double, exponential in float
        const double K = Tc + 273.15;
        const double p = expf((float)(-5800.2206 / K + 1.3914993 + 6.5459673 * log(K) - 4.8640239e-2 * K + 4.1764768e-5 * K * K
                                      - 1.4452093e-8 * K * K * K));
        return (float)p;
    }
    float rh_calibrated(uint32_t f, uint32_t f1, uint32_t f2, float Tc, float Th, float Pp) const {
        const float cfh = (f - f1) / (float)(f2 - f1);
        const float cap = Cf1 + (Cf2 - Cf1) * cfh;
        double Cp = (cap / calH[0] - 1.0) * calH[1];
        const double x = (Th - 20.0) / 180.0;
        double b[6], bk = 1.0;
        for (int k = 0; k < 6; k++) { b[k] = bk; bk *= x; }
        if (Pp > 0.0) {
            const double pb = Pp / 1000.0;
            double cpj = 1.0, bp[3], corr = 0.0;
            for (int j = 0; j < 3; j++) { bp[j] = corHp[j] * (pb / (1.0 + corHp[j] * pb) - cpj / (1.0 + corHp[j])); cpj *= Cp; }
            for (int j = 0; j < 3; j++) {
                double bt_ = 0.0;
                for (int k = 0; k < 4; k++) bt_ += corHt[4 * j + k] * b[k];
                corr += bp[j] * bt_;
            }
            Cp -= corr;
        }
        double r = 0.0, aj = 1.0;
        for (int j = 0; j < 7; j++) { for (int k = 0; k < 6; k++) r += aj * b[k] * mtxH[6 * j + k]; aj *= Cp; }
        if (Pp <= 0.0) { const float T2 = -40; if (Tc < T2) r += (Tc - T2) / 12.0; }
        float rh = (float)(r * sat_vapour(Th) / sat_vapour(Tc));
        if (rh < 0.0) rh = 0.0f;
        if (rh > 100.0) rh = 100.0f;
        return rh;
    }
    float pressure(uint32_t f, uint32_t f1, uint32_t f2, int fx) const {
        if (f1 == f2 || f1 == f) return 0.0f;
        const double a0 = calP[24] / ((float)(f - f1) / (float)(f2 - f1));
        const double a1 = fx * 0.01;
        double p = 0.0, a0j = 1.0;
        for (int j = 0; j < 6; j++) { double a1k = 1.0; for (int k = 0; k < 4; k++) { p += a0j * a1k * calP[j * 4 + k]; a1k *= a1; } a0j *= a0; }
        return (float)p;
    }
    static float barometric(float h) {                         // standard atmosphere, four layers
        double Pb, Tb, Lb, hb;
        const double gMR = 9.80665 * 0.0289644 / 8.31446;
        if (h > 32000.0)      { Pb = 8.6802;  Tb = 228.65; Lb = 0.0028;  hb = 32000.0; }
        else if (h > 20000.0) { Pb = 54.7489; Tb = 216.65; Lb = 0.001;   hb = 20000.0; }
        else if (h > 11000.0) { Pb = 226.321; Tb = 216.65; Lb = 0.0;     hb = 11000.0; }
        else                  { Pb = 1013.25; Tb = 288.15; Lb = -0.0065; hb = 0.0; }
        if (Lb == 0.0) return (float)(Pb * exp(-gMR * (h - hb) / Tb));
        return (float)(Pb * pow(1.0 + Lb * (h - hb) / Tb, -gMR / Lb));
    }
    int ptu(int ofs, int kind, int valid_alt) {
        float Tc = -273.15f, Th = -273.15f, rh = -1.0f, rh2 = -1.0f, Pp = -1.0f;
        load_coefficients();
        const int bad = block_crc(P_PTU + ofs, kind);
        if (bad) { crc |= F_PTU; return bad; }
        uint32_t m[12];
        for (int i = 0; i < 12; i++) m[i] = le24(fr + P_PTU + ofs + 2 + 3 * i);
        const bool bR = have[3] && have[4], bc1 = have[4] && have[5], bT1 = have[5] && have[6], bc2 = have[0x12] && have[0x13], bT2 = have[0x13],
                   bH = have[7];
        bool bH2 = true;
        for (int k = 0x07; k <= 0x12; k++) bH2 = bH2 && have[k];
        for (int k = 0x2A; k <= 0x2E; k++) bH2 = bH2 && have[k];
        bool bP = have[0x21] && cal[0x21F] == 'P';
        for (int k = 0x25; k <= 0x2A; k++) bP = bP && have[k];
        if (bR && bc1 && bT1) Tc = temperature(m[0], m[1], m[2], co1, calT1);
        T = Tc;
        if (bR && bc2 && bT2) Th = temperature(m[6], m[7], m[8], co2, calT2);
        TH = Th;
        if (bH && Tc > -273.0) rh = rh_empirical(m[3], m[4], m[5], Tc);
        RH = rh;
        if (bP) Pp = pressure(m[9], m[10], m[11], s16(fr + P_PTU + ofs + 2 + 38));
        P = Pp;
        if (o.ptu == 2) {
            float pe = -1.0f;
            if (bP) pe = Pp;
            else if (valid_alt > 0) pe = barometric((float)alt);
            if (bH && bH2 && Tc > -273.0 && Th > -273.0) rh2 = rh_calibrated(m[3], m[4], m[5], Tc, Th, pe);
        }
        RH2 = rh2;
        return 0;
    }

    // ---- configuration subframes -------------------------------------------------------------------------------------
    void conf_subframe(Out &w, int out, int ofs) {
        subfrm_pos = P_CAL + ofs;
        const int k = fr[P_CAL + ofs];
        const uint8_t *c = fr + P_CAL + ofs;                   // c[0] = subframe number, c[1..16] = its bytes
        if (out && o.verbose == 3) {                           // -vv: the subframe bytes, before the CRC is looked at (:1565-1578)
            w.f("\n"); w.f("[%5d] ", frnr); w.f(" 0x%02x: ", k);
            for (int i = 0; i < 16; i++) w.f("%02x ", c[1 + i]);
            w.f(" ");
        }
        if (block_crc(P_STATUS + ofs, K_STATUS)) return;
        if (k == 0x00) {
            const int f0 = ((c[3] & 0xC0) * 10) / 64, f1 = 40 * c[4];
            freq = 400000 + f1 + f0;
            if (out && o.verbose) w.f(": fq %d ", (int)freq);
        }
        if (k == 0x01) { fw = (uint16_t)(c[6] | c[7] << 8); if (out && o.verbose) w.f(": fw 0x%04x ", fw); }
        if (k == 0x02) {
            bk = c[12]; kt = (uint16_t)(c[8] + (c[9] << 8));
            if (out && o.verbose) w.f(": BK %02X ", bk);
            if (out && o.verbose && kt != 0xFFFF) w.f(": kt %.1fmin ", kt / 60.0);
        }
        if (k == 0x31) {
            bt = (uint16_t)(c[7] + (c[8] << 8));
            if (out && bt != 0 && (o.verbose == 3 || (o.verbose && bk))) w.f(": bt %.1fmin ", bt / 60.0);
        }
        if (k == 0x32) {
            cd = (uint16_t)(c[1] + (c[2] << 8));
            if (out && cd != 0xFFFF && (o.verbose == 3 || (o.verbose && (bk || kt != 0xFFFF)))) w.f(": cd %.1fmin ", cd / 60.0);
        }
        if (k == 0x21) {
            memset(rstmp, 0, sizeof rstmp);
            for (int i = 0; i < 8; i++) { const uint8_t b = c[9 + i]; if (b >= 0x20 && b < 0x7F) rstmp[i] = (char)b; else if (b == 0) rstmp[i] = 0; }
            if (out && o.verbose == 3) {                       // station pressure; read at the standard position, without the block offset (:1634-1635)
                float q1, q2; memcpy(&q1, fr + P_CAL + 1, 4); memcpy(&q2, fr + P_CAL + 5, 4);
                if (q1 > 0.0 || q2 > 0.0) { w.f(" "); if (q1 > 0.0) w.f("QFE1:%.1fhPa ", q1); if (q2 > 0.0) w.f("QFE2:%.1fhPa ", q2); }
            }
        }
        if (k == 0x22) {
            const uint8_t b = c[1];
            if (b >= 0x20 && b < 0x7F) rstmp[8] = (char)b; else if (b == 0) rstmp[8] = 0;
            strcpy(rstyp, rstmp);
            memset(rstmp, 0, sizeof rstmp);
            if (out && o.verbose && *rstyp) w.f(": %s ", rstyp);
            memset(rsm, 0, sizeof rsm);
            for (int i = 0; i < 8; i++) { const uint8_t q = c[3 + i]; rsm[i] = (q >= 0x20 && q < 0x7F) ? (char)q : 0; }
            if (out && o.verbose) w.f(": %s ", rsm);
        }
    }

    // ---- xdata -------------------------------------------------------------------------------------------------------
    // ---- --aux: the ozone / frost-point instruments in the xdata text (hex2uint :1263-1278, prn_aux_IDx01 / 05 / 08 :1280-1452) -------------------
    static int hex2uint(const char *str, int nibs) {
        int erg = 0;
        if (nibs > 7) return -2;
        for (int i = 0; i < nibs; i++) {
            int h;
            if (str[i] >= '0' && str[i] <= '9') h = str[i] - '0';
            else if (str[i] >= 'a' && str[i] <= 'f') h = str[i] - 'a' + 0xA;
            else if (str[i] >= 'A' && str[i] <= 'F') h = str[i] - 'A' + 0xA;
            else return -1;
            erg = (erg << 4) | (h & 0xF);
        }
        return erg;
    }
    static const char *aux_at(const char *x, const char *id, const char *hash_id, size_t need) {
        const char *px = x;
        if (!*px) return nullptr;
        if (strncmp(px, id, 2) != 0) { px = strstr(x, hash_id); if (!px) return nullptr; px += 1; }
        return strlen(px) < need ? nullptr : px;
    }
    static void aux_ecc(Out &w, const char *x) {                // ID 0x01: ECC ozone sonde
        const char *px = aux_at(x, "01", "#01", 16);
        if (!px) return;
        w.f(" ID=0x01 ECC ");
        int v = hex2uint(px + 2, 2); if (v < 0) return; const uint8_t num = (uint8_t)v;
        v = hex2uint(px + 4, 4); if (v < 0) return; const uint16_t icell = (uint16_t)v;
        v = hex2uint(px + 8, 4); if (v < 0) return; const int16_t tpump = (int16_t)v;
        v = hex2uint(px + 12, 2); if (v < 0) return; const uint8_t ipump = (uint8_t)v;
        v = hex2uint(px + 14, 2); if (v < 0) return; const uint8_t vbat = (uint8_t)v;
        w.f(" No.%d ", num); w.f(" Icell:%.3fuA ", icell / 1000.0); w.f(" Tpump:%.2fC ", tpump / 100.0); w.f(" Ipump:%dmA ", ipump); w.f(" Vbat:%.1fV ", vbat / 10.0);
    }
    static void aux_oif411(Out &w, const char *x) {             // ID 0x05: OIF411 ozone interface
        const char *px = aux_at(x, "05", "#05", 20);
        if (!px) return;
        w.f(" ID=0x05 OIF411 ");
        int v = hex2uint(px + 2, 2); if (v < 0) return;
        w.f(" No.%d ", (uint8_t)v);
        if (px[20] == 'I') {
            char sn[9]; strncpy(sn, px + 4, 8); sn[8] = 0;
            v = hex2uint(px + 12, 4); if (v < 0) return; const uint16_t dw = (uint16_t)v;
            v = hex2uint(px + 16, 4); if (v < 0) return; const uint16_t sw = (uint16_t)v;
            w.f(" SN:%s ", sn); w.f(" DW:%04X ", dw); w.f(" SW:%.2f ", sw / 100.0);
        } else {
            v = hex2uint(px + 4, 4); if (v < 0) return; const int16_t tpump = (int16_t)v;
            v = hex2uint(px + 8, 5); if (v < 0) return; const uint32_t icell = (uint32_t)v & 0xFFFFF;
            v = hex2uint(px + 13, 2); if (v < 0) return; const uint8_t vbat = (uint8_t)v;
            v = hex2uint(px + 15, 3); if (v < 0) return; const uint16_t ipump = (uint16_t)(v & 0xFFF);
            v = hex2uint(px + 18, 2); if (v < 0) return; const uint8_t vext = (uint8_t)v;
            w.f(" Tpump:%.2fC ", tpump / 100.0); w.f(" Icell:%.4fuA ", icell / 10000.0); w.f(" Vbat:%.1fV ", vbat / 10.0); w.f(" Ipump:%dmA ", ipump); w.f(" Vext:%.1fV ", vext / 10.0);
        }
    }
    static void aux_cfh(Out &w, const char *x) {                // ID 0x08: CFH frost-point hygrometer
        const char *px = aux_at(x, "08", "#08", 24);
        if (!px) return;
        w.f(" ID=0x08 CFH ");
        const int v = hex2uint(px + 2, 2); if (v < 0) return;
        w.f(" No.%d ", (uint8_t)v);
        w.f(" Tmir:0x%.6s ", px + 4); w.f(" Vopt:0x%.6s ", px + 10); w.f(" Topt:0x%.4s ", px + 16); w.f(" Vbat:0x%.4s ", px + 20);
    }

    int xdata(int pos, Out *w = nullptr) {                     // w: -vx / -vv print the text as it is collected (:1492-1506)
        int n = 0, last = 0, cnt = 0;
        xd[0] = 0;
        if (frametype() <= 0) {
            // (the reference appends without a bound, rs41mod.c get_Aux; frames come off the air, so the text stops where xd[] ends)
            const int cap = (int)sizeof xd - 2;
            while (pos + 1 < FL && fr[pos] == 0x7E) {
                const int len = fr[pos + 1];
                if (pos + len + 4 <= FL && (int)(fr[pos + 2 + len] | fr[pos + 3 + len] << 8) == crc16(fr + pos + 2, len)) {
                    if (!cnt) { if (w) w->f("\n # xdata = "); }
                    else { if (w) w->f(" # "); if (n < cap) xd[n++] = '#'; }
                    for (int i = 1; i < len; i++) { const uint8_t ch = fr[pos + 2 + i]; if (ch > 0x1E && ch < 0x7F) { if (w) w->f("%c", ch); if (n < cap) xd[n++] = (char)ch; } }
                    cnt++; last = pos; pos += 2 + len + 2;
                } else { pos = FL; crc |= F_AUX; }
            }
        }
        xd[n] = 0;
        if (w && o.aux && xd[0]) {                             // get_Aux :1512-1540
            const char *paux = xd;
            for (int i = 0; i < cnt; i++) {
                if (paux > xd) { while (*paux && *paux != '#') paux++; paux++; }
                if (strlen(paux) > 2) {
                    const int v = hex2uint(paux, 2);
                    if (v < 0) { paux += 2; continue; }
                    switch (v & 0xFF) {
                        case 0x01: w->f("\n"); aux_ecc(*w, paux); break;
                        case 0x05: w->f("\n"); aux_oif411(*w, paux); break;
                        case 0x08: w->f("\n"); aux_cfh(*w, paux); break;
                    }
                    paux++;
                } else break;
            }
            if (!o.json) w->f("\n");
        }
        if (pos < FL - 3 ? block_crc(pos, 0x7600) : -1) crc |= F_ZERO;
        return last;
    }

    // ---- text pieces (prn_frm / prn_gpstime / prn_gpspos / prn_posdatetime / prn_ptu, :1978-2050) ----------------------
    void t_frame(Out &w) const { w.f("[%5d] ", frnr); w.f("(%s) ", id); if (o.verbose == 3) w.f("(%.1f V) ", batt); w.f(" "); }
    void t_time(Out &w) const {
        w.f("%s ", kDay[((wday % 7) + 7) % 7]);
        w.f("%04d-%02d-%02d %02d:%02d:%06.3f", year, month, day, hour, minute, sec);
        if (o.verbose == 3) w.f(" (W %d)", week);
        w.f(" ");
    }
    void t_pos(Out &w) const {
        w.f(" lat: %.5f ", lat); w.f(" lon: %.5f ", lon); w.f(" alt: %.2f ", alt);
        w.f("  vH: %4.1f  D: %5.1f  vV: %3.1f ", vH, vD, vV);
        if (o.verbose == 3) w.f(" sats: %02d ", numSV);
    }
    void t_posdt(Out &w) const {
        w.f("%04d-%02d-%02d %02d:%02d:%05.2f", year, month, day, hour, minute, sec);
        w.f(" "); w.f(" ");
        t_pos(w);
    }
    void t_ptu(Out &w) const {
        w.f(" ");
        if (T > -273.0) w.f(" T=%.1fC ", T);
        if (RH > -0.5 && o.ptu != 2) w.f(" _RH=%.0f%% ", RH);
        if (P > 0.0) { if (P < 100.0) w.f(" P=%.2fhPa ", P); else w.f(" P=%.1fhPa ", P); }
        if (o.ptu == 2 && RH2 > -0.5) w.f(" RH2=%.0f%% ", RH2);
        if (o.dewp) {
            const float rh = o.ptu == 2 ? RH2 : RH;
            if (rh > 0.0f && T > -273.0f) {
                const float g = logf(rh / 100.0f) + (17.625f * T / (243.04f + T));
                w.f(" Td=%.1fC ", 243.04f * g / (17.625f - g));
            }
        }
    }

    void json(Out &w, int err0, int encrypted) {
        w.f("{ \"type\": \"%s\"", "RS41");
        w.f(", \"frame\": %d, \"id\": \"%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, "
            "\"vel_h\": %.5f, \"heading\": %.5f, \"vel_v\": %.5f, \"sats\": %d, \"bt\": %d, \"batt\": %.2f",
            frnr, id, year, month, day, hour, minute, sec, lat, lon, alt, vH, vD, vV, numSV, (int)cd, batt);
        if (o.ptu && !err0) {
            const float rh = o.ptu == 2 ? RH2 : RH;
            if (T > -273.0) w.f(", \"temp\": %.1f", T);
            if (rh > -0.5) w.f(", \"humidity\": %.1f", rh);
            if (P > 0.0) w.f(", \"pressure\": %.2f", P);
        }
        if (aux) w.f(", \"aux\": \"%s\"", xd);
        if (encrypted) w.f(", \"subtype\": \"RS41-SGM\", \"encrypted\": true");
        else {
            w.f(", \"subtype\": \"%s\"", *rstyp ? rstyp : "RS41");
            if (strncmp(rstyp, "RS41-SGM", 8) == 0) w.f(", \"encrypted\": false");
        }
        if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", freq > 0 ? (int)freq : o.jsn_freq_khz);
        if (*rsm) w.f(", \"rs41_mainboard\": \"%s\"", rsm);
        if (fw) w.f(", \"rs41_mainboard_fw\": %d", fw);
        const uint8_t *sf = fr + subfrm_pos;
        if (o.jsn_subfrm == 1) {
            if (!cal_sent && cal_complete) {
                w.f(", \"rs41_calconf51x16\": \"");
                for (int j = 0; j < 51 * 16; j++) w.f("%02X", cal[j]);
                w.f("\"");
                cal_sent = 1;
            }
            if (sf[0] == 0x32) { w.f(", \"rs41_conf0x32\": \""); for (int j = 0; j < 16; j++) w.f("%02X", sf[1 + j]); w.f("\""); }
        }
        if (o.jsn_subfrm == 2) { w.f(", \"rs41_subfrm\": \"0x%02X:", sf[0]); for (int j = 0; j < 16; j++) w.f("%02X", sf[1 + j]); w.f("\""); }
        if (freq > 0) w.f(", \"tx_frequency\": %d", (int)freq);
        w.f(", \"ref_datetime\": \"%s\"", isUTC ? "UTC" : "GPS");
        w.f(", \"ref_position\": \"%s\"", "GPS");
        if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
        w.f(" }\n");
        w.f("\n");
    }

    // ---- --sat: raw contents of the three GPS blocks (prn_sat1/2/3, :2052-2111) ------------------------------------------------------
    static uint32_t le4(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
    static int le2(const uint8_t *p) { return p[0] | (p[1] << 8); }
    void t_sat1(Out &w, int ofs) { w.f("\n"); w.f("iTOW: 0x%08X", le4(fr + 0x097 + ofs)); w.f("  week: 0x%04X", le2(fr + 0x095 + ofs)); }
    void t_sat2(Out &w, int ofs) {
        const double c = 299.792458e6, L1 = 1575.42e6;
        w.f("\n");
        const uint32_t minPR = le4(fr + 0x0B7 + ofs);
        w.f("minPR: %d", (int)minPR); w.f("\n");
        for (int i = 0; i < 12; i++) {
            const int sv = fr[0x09B + ofs + 2 * i];
            if (sv == 0xFF) break;
            const uint8_t *p = fr + 0x0BC + ofs + 7 * i;
            int d24 = p[4] | (p[5] << 8) | (p[6] << 16);
            if (d24 & 0x800000) d24 -= 0x1000000;
            w.f("    SV: %2d ", sv); w.f("#  ");
            w.f("prMes: %.1f", le4(p) / 100.0 + minPR); w.f("  ");
            w.f("doMes: %.1f", -d24 / 100.0 * L1 / c); w.f("\n");
        }
    }
    void t_sat3(Out &w, int ofs) {
        w.f("\n");
        w.f("ECEF-POS: (%d,%d,%d)\n", (int32_t)le4(fr + 0x114 + ofs), (int32_t)le4(fr + 0x118 + ofs), (int32_t)le4(fr + 0x11C + ofs));
        w.f("ECEF-VEL: (%d,%d,%d)\n", (int16_t)le2(fr + 0x120 + ofs), (int16_t)le2(fr + 0x122 + ofs), (int16_t)le2(fr + 0x124 + ofs));
        double sAcc = fr[0x127 + ofs] / 10.0, pDOP = fr[0x128 + ofs] / 10.0;
        if (fr[0x127 + ofs] == 0xFF) sAcc = -1.0;
        if (fr[0x128 + ofs] == 0xFF) pDOP = -1.0;
        w.f("numSatsFix: %2d  sAcc: %.1f  pDOP: %.1f\n", fr[0x126 + ofs], sAcc, pDOP);
    }

    // ---- frames whose ECC passed ---------------------------------------------------------------------------------------
    void run_good(Out &w, int ec) {
        const int out = !o.silent;
        int err = 1, err0 = 1, err1 = 1, err3 = 1, err13 = 1, encrypted = 0, pos_aux = 0, ofs_ptu = 0, kind_ptu = 0, ofs_cal = 0, err2g = 1, is_gnss2 = 0;
        int flen = NDATA;
        if (frametype() < 0) flen += 198;
        int frm_end = NDATA - 2;
        switch (fr[P_PTU]) {
            case 0x7A: frm_end = flen - 2; break;
            case 0x7F: frm_end = P_ZERO + 0x1B - 0x2A - 2; break;
            case 0x80: frm_end = P_PTU + 2 + 0xA7; break;
        }
        int pos = P_STATUS;
        crc = 0;
        while (pos < flen - 1) {
            const int blk = fr[pos], len = fr[pos + 1], kind = (blk << 8) | len;
            if (block_crc(pos, blk << 8) != 0) { w.f(" [ERROR]\n"); break; }
            switch (kind) {
                case K_STATUS: ofs_cal = pos - P_STATUS; err = status_block(ofs_cal); have_id = !err; if (!err && (out || o.sat)) t_frame(w); break;
                case K_PTU: ofs_ptu = pos - P_PTU; kind_ptu = K_PTU; break;
                case K_GPS1: err1 = gps_time(pos - P_GPS1); if (!err1) { gps_date(); if (out) t_time(w); if (o.sat) t_sat1(w, pos - P_GPS1); } break;
                case K_GPS2: if (block_crc(P_GPS2 + (pos - P_GPS2), K_GPS2)) crc |= F_GPS2; else if (o.sat) t_sat2(w, pos - P_GPS2); break;
                case K_GPS3: err3 = gps_pos(pos - P_GPS3); if (!err3) { if (out) t_pos(w); if (o.sat) t_sat3(w, pos - P_GPS3); } break;
                case K_XTU: ofs_ptu = pos - P_PTU; kind_ptu = kind; break;
                case K_CRYPT: encrypted = 1; if (out) w.f(" [%04X] (RS41-SGM) ", K_CRYPT); break;
                case K_960A: break;
                case K_POSDT: err13 = pos_datetime(pos); if (!err13 && out) t_posdt(w); break;
                case K_SATS: err2g = gnss_sats(pos); is_gnss2 = 1; break;
                default:
                    if (blk == 0x7E) { if (!pos_aux) pos_aux = pos; }
                    if (blk != 0x76 && blk != 0x7E) { if (out) w.f(" [%04X] ", kind); }
            }
            pos += 2 + len + 2;
            if (pos > frm_end) {                               // end of the (sub)frame: PTU, configuration, trailer, JSON
                if (o.ptu && !o.sat && !encrypted && kind_ptu > 0) { err0 = ptu(ofs_ptu, kind_ptu, !err3); if (!err0 && out) t_ptu(w); }
                kind_ptu = 0;
                conf_subframe(w, out, ofs_cal);
                if (out && ec > 0 && pos > flen - 1) w.f(" (%d)", ec);
                if (pos_aux) aux = xdata(pos_aux, (out && o.verbose > 1) ? &w : nullptr);
                crc = 0;
                frm_end = FL - 2;
                if (is_gnss2 && o.sat && !err2g) t_gnss_sat2(w);
                if (out || o.sat) w.f("\n");
                if (o.json && !err && ((!err1 && !err3) || !err13 || encrypted)) json(w, err0, encrypted);
            }
        }
        have_time = !err1 || !err13; have_pos = !err3 || !err13;
    }

    // ---- --ecc3 / --ecc4: the frame from both soft bits of every bit, byte scores, and the list decoding of rs41_ecc -----------
    // (bit loop rs41mod.c:2918-2962, score sorting in print_frame :2490-2522, rs41_ecc :1703-1974)
    int ecc34(int level, int inv, const float *s0, const float *s1, int nbits, float ts, uint8_t *frame_out, int *len_out, int *nbytes_out) {
        static sonde_ecc_t *rs = sonde_ecc_create(SONDE_ECC_RS255);
        enum { FS = 8, PAR = 8, MSG = 56, RR = 24, KK = 231, NDATA = 320, P_ZSTD = 0x12B };
        if (!ec_frame_init) {
            memset(ec_frame, 0, FL); memcpy(ec_frame, kRs41HeaderBytes, 8); ec_frame_init = true;
            memset(ec_bitscore, 0, sizeof ec_bitscore);
            for (int i = 0; i < FL + 8; i++) ec_score[i] = 0.f;
        }
        uint8_t *F = ec_frame;
        float *score = ec_score; uint8_t *bitscore = ec_bitscore;
        int bc = FS;
        for (; bc < FL && 8 * (bc - FS + 1) <= nbits; bc++) {
            const float *a = s0 + 8 * (bc - FS), *b = s1 + 8 * (bc - FS);
            unsigned byte = 0; float sb[8];
            for (int j = 0; j < 8; j++) {
                int bit = level >= 3 ? ((a[j] + b[j]) >= 0) : (a[j] >= 0);
                sb[j] = a[j];
                if (inv) { bit ^= 1; sb[j] = -sb[j]; }
                byte |= (unsigned)bit << j;                   // bits2byte: LSB first
            }
            int j0 = 0; float m = sb[0];
            for (int j = 1; j < 8; j++) if (fabsf(sb[j]) < fabsf(m)) { m = sb[j]; j0 = j; }
            score[bc] = m; bitscore[bc] = (uint8_t)(1 << j0);
            F[bc] = (uint8_t)(byte ^ kRs41Mask[bc % 64]);
        }
        ec_ts = ts;
        int len = bc;
        if (len < 0x93) for (int i = len; i < FL; i++) F[i] = 0;              // print_frame :2479-2482
        { int t = 0; const uint8_t b = F[0x38]; for (int i = 0; i < 4; i++) t += ((b >> i) & 1) - ((b >> (i + 4)) & 1); len = t >= 0 ? NDATA : FL; }
        float mm = 0.f;
        for (int i = FS; i < len; i++) if (fabsf(score[i]) > mm) mm = fabsf(score[i]);
        mm = floorf(mm + 1.5f);
        if (level > 2) {
            for (int i = 0; i < FS; i++) score[i] = mm * 2.0f;
            for (int i = len; i < FL; i++) score[i] = mm;
        }
        int order[FL], idx1[FL], idx2[FL];
        for (int i = 0; i < FL; i++) { order[i] = i; idx1[i] = i; idx2[i] = i; }
        if (level > 2) {
            for (int i = 0; i < FL; i++)                                       // the reference's bubble sort: stable, ascending |score|
                for (int j = 0; j < FL - 1; j++)
                    if (fabsf(score[order[j + 1]]) < fabsf(score[order[j]])) { const int t = order[j + 1]; order[j + 1] = order[j]; order[j] = t; }
            int j1 = 0, j2 = 0;
            for (int i = 0; i < FL; i++) {
                const int k = order[i];
                if (k >= PAR && k < PAR + RR) idx1[j1++] = k;
                else if (k >= PAR + RR && k < PAR + 2 * RR) idx2[j2++] = k;
                else if (k >= MSG && k % 2 == 0) idx1[j1++] = k;
                else if (k >= MSG && k % 2 == 1) idx2[j2++] = k;
            }
        }
        // ---- rs41_ecc
        int frmlen = len;
        for (int i = frmlen; i < FL; i++) F[i] = 0;
        uint8_t cw1[255], cw2[255], ep1[RR], ev1[RR], ep2[RR], ev2[RR], era[RR];
        memset(cw1, 0, 255); memset(cw2, 0, 255);
        for (int i = 0; i < RR; i++) { cw1[i] = F[PAR + i]; cw2[i] = F[PAR + RR + i]; }
        auto msg1 = [&] { for (int i = 0; i < KK; i++) cw1[RR + i] = F[MSG + 2 * i]; };
        auto msg2 = [&] { for (int i = 0; i < KK; i++) cw2[RR + i] = F[MSG + 2 * i + 1]; };
        msg1(); msg2();
        int e1 = sonde_ecc_decode(rs, cw1, ep1, ev1), e2 = sonde_ecc_decode(rs, cw2, ep2, ev2);
        auto ftype = [&] { int t = 0; const uint8_t b = F[0x38]; for (int i = 0; i < 4; i++) t += ((b >> i) & 1) - ((b >> (i + 4)) & 1); return t; };
        if (level >= 2 && (e1 < 0 || e2 < 0)) {                                // 2nd pass: block ids, zero tail (:1736-1761)
            static const int pos[5] = { P_STATUS, P_PTU, P_GPS1, P_GPS2, P_GPS3 };
            static const int pck[5] = { K_STATUS, K_PTU, K_GPS1, K_GPS2, K_GPS3 };
            for (int k = 0; k < 5; k++) { F[pos[k]] = (uint8_t)(pck[k] >> 8); F[pos[k] + 1] = (uint8_t)(pck[k] & 0xFF); }
            if (ftype() < -2) { for (int i = NDATA + 7; i < FL - 2; i++) F[i] = 0; }
            else {
                for (int i = NDATA; i < FL; i++) F[i] = 0;
                F[P_ZSTD] = 0x76; F[P_ZSTD + 1] = 0x11;
                for (int i = P_ZSTD + 2; i < NDATA - 2; i++) F[i] = 0;
                F[NDATA - 2] = 0xEC; F[NDATA - 1] = 0xC7;
            }
            msg1(); msg2();
            e1 = sonde_ecc_decode(rs, cw1, ep1, ev1); e2 = sonde_ecc_decode(rs, cw2, ep2, ev2);
        }
        int frmset[FL], setcnt = 0;
        auto chk = [&](int pos_, int kind) {                                    // check_CRC on the working frame (:307-319)
            if (((kind >> 8) & 0xFF) != F[pos_]) return -1;
            const int n = F[pos_ + 1];
            if (pos_ + n + 4 > FL) return -1;
            return (int)le16(F + pos_ + 2 + n) != crc16(F + pos_ + 2, n) ? 1 : 0;
        };
        auto set_bytes = [&](int pos_, const uint8_t *src, int n, int subcw) {  // (:1667-1679): only the bytes of that codeword
            const int rem = subcw == 2 ? 1 : 0;
            int *pset = frmset + setcnt;
            for (int i = 0; i < n; i++) if ((pos_ + i) % 2 == rem) { F[pos_ + i] = src[i]; *pset++ = pos_ + i; }
        };
        if (level == 4) {                                                      // known bytes of the same sonde (:1764-1849)
            const float frnb_ts = ec_ts - ec_last_frnb_ts + 0.5f;
            const int frnb = ec_last_frnb + (int)(unsigned)frnb_ts;
            const float calfr_ts = ec_ts - ec_last_calfrm_ts + 0.5f;
            const int calfr = (ec_last_calfrm + (int)(unsigned)calfr_ts) % 51;
            for (int cwn = 1; cwn <= 2; cwn++) {
                int &e = cwn == 1 ? e1 : e2;
                if (e >= 0) continue;
                int c = chk(P_STATUS, K_STATUS);
                if (c) {
                    if (id[0] && strncmp((const char *)F + P_ID, id, 8) != 0) { set_bytes(P_ID, (const uint8_t *)id, 8, cwn); setcnt += 8 / 2; }
                    c = chk(P_STATUS, K_STATUS);
                    if (c && have[calfr]) {
                        if (F[P_CAL] == calfr) { set_bytes(P_CAL + 1, cal + calfr * 16, 16, cwn); setcnt += 16 / 2; }
                    }
                    c = chk(P_STATUS, K_STATUS);
                    if (cwn == 1) { if (c && ((frnb >> 8) & 0xFF) != F[P_FRNR + 1]) { if (ec_last_frnb > 0) { F[P_FRNR + 1] = (uint8_t)((frnb >> 8) & 0xFF); frmset[setcnt++] = P_FRNR + 1; } } }
                    else          { if (c && (frnb & 0xFF) != F[P_FRNR]) { if (ec_last_frnb > 0) { F[P_FRNR] = (uint8_t)(frnb & 0xFF); frmset[setcnt++] = P_FRNR; } } }
                }
                if (cwn == 1) { msg1(); e1 = sonde_ecc_decode(rs, cw1, ep1, ev1); } else { msg2(); e2 = sonde_ecc_decode(rs, cw2, ep2, ev2); }
            }
        }
        auto in_fixed = [&](int idx) {                                          // (:1684-1697)
            static const int fx[5] = { P_STATUS, P_PTU, P_GPS1, P_GPS2, P_GPS3 };
            for (int j = 0; j < 5; j++) if (idx == fx[j] || idx == fx[j] + 1) return true;
            if (ftype() >= -2) { if (idx >= P_ZSTD && idx < NDATA) return true; }
            for (int j = 0; j < setcnt; j++) if (idx == frmset[j]) return true;
            return false;
        };
        if (level > 2) {                                                       // 3rd pass: 2 erasures + toggled low-score bits (:1861-1941)
            const int Era_max = 12;
            for (int cwn = 1; cwn <= 2; cwn++) {
                int &e = cwn == 1 ? e1 : e2;
                if (e >= 0) continue;
                const int *sidx = cwn == 1 ? idx1 : idx2;
                uint8_t *cw = cwn == 1 ? cw1 : cw2; uint8_t *ep = cwn == 1 ? ep1 : ep2, *ev = cwn == 1 ? ev1 : ev2;
                auto cwpos = [&](int pf) { return pf < MSG ? pf - PAR - (cwn == 2 ? RR : 0) : RR + (pf - MSG) / 2; };
                for (int i = 1; i < Era_max; i++) {
                    int pf = sidx[i];
                    if (in_fixed(pf)) continue;
                    int pc = cwpos(pf);
                    if (pc < 0 || pc > 254) continue;
                    era[0] = (uint8_t)pc;
                    for (int j = 0; j < i; j++) {
                        pf = sidx[j];
                        if (in_fixed(pf)) continue;
                        pc = cwpos(pf);
                        if (pc < 0 || pc > 254) continue;
                        era[1] = (uint8_t)pc;
                        for (int k = -1; k < j; k++) {
                            if (k >= 0) {
                                pf = sidx[k];
                                if (in_fixed(pf)) continue;
                                pc = cwpos(pf);
                                if (pc < 0 || pc > 254) continue;
                                cw[pc] ^= bitscore[pf];                       // toggled bits stay toggled (the reference does not undo them)
                            }
                            e = sonde_ecc_decode_errera(rs, cw, 2, era, ep, ev);
                            if (e >= 0) { j = 256; i = 256; k = 256; }
                        }
                    }
                }
            }
        }
        for (int i = 0; i < RR; i++) { F[PAR + i] = cw1[i]; F[PAR + RR + i] = cw2[i]; }
        for (int i = 0; i < KK; i++) { F[MSG + 2 * i] = cw1[RR + i]; F[MSG + 1 + 2 * i] = cw2[RR + i]; }
        int ret = e1 + e2;
        if (e1 < 0 || e2 < 0) ret = -((e1 < 0 ? 1 : 0) | (e2 < 0 ? 2 : 0));
        memcpy(frame_out, F, FL);
        *len_out = len; *nbytes_out = bc;
        return ret;
    }

    // ---- frames whose ECC failed: what still has a good block CRC --------------------------------------------------------
    void run_failed(Out &w, int ec) {
        if (o.silent) return;
        int output = 0;
        crc = 0;
        unsigned mask = F_STATUS | F_GPS1 | F_GPS3;
        if (o.ptu) mask |= F_PTU;
        const int err = status_block(0);
        have_id = !err;
        if (!err) { t_frame(w); output = 1; }
        const int kind = (fr[P_PTU] << 8) | fr[P_PTU + 1];
        if (kind < 0x8000) {
            int ofs = 0;
            if (kind == K_XTU) ofs = 0x1B - 0x2A;
            const int err1 = gps_time(ofs);
            if (block_crc(P_GPS2 + ofs, K_GPS2)) crc |= F_GPS2;
            const int err3 = gps_pos(ofs);
            if (!err1) gps_date();
            const int err0 = ptu(0, kind, !err3);
            if (!err1) t_time(w);
            if (!err3) t_pos(w);
            if (!err0 && o.ptu) t_ptu(w);
            output = ((crc & mask) != mask);
            if (output) { w.f(" "); w.f("["); for (int i = 0; i < 5; i++) w.f("%d", (crc >> i) & 1); w.f("]"); }
            have_time = !err1; have_pos = !err3;
        } else if (kind == K_CRYPT) {
            if (!err) { w.f(" [%04X] (RS41-SGM) ", K_CRYPT); output = 1; }
        }
        if (output) { w.f(ec == -1 ? " (-+)" : ec == -2 ? " (+-)" : " (--)"); w.f("\n"); }
    }
};

extern "C" {

int sonde_rs41_dec_create(const sonde_rs41_opts_t *opts, sonde_rs41_dec_t **out) {
    if (!opts || !out || opts->verbose < 0 || opts->verbose > 3 || opts->ptu < 0 || opts->ptu > 2 || opts->jsn_subfrm < 0 || opts->jsn_subfrm > 2)
        return SONDE_E_ARG;
    sonde_rs41_dec *d = new sonde_rs41_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.jsn_subfrm) d->o.json = 1;
    if (d->o.aux) d->o.verbose = 2;                            // rs41mod.c:2763
    memset(d->fr, 0, sizeof d->fr); memset(d->cal, 0, sizeof d->cal); memset(d->have, 0, sizeof d->have);
    memset(d->rstyp, 0, sizeof d->rstyp); memset(d->rstmp, 0, sizeof d->rstmp); memset(d->rsm, 0, sizeof d->rsm); memset(d->xd, 0, sizeof d->xd);
    d->Rf1 = d->Rf2 = d->Cf1 = d->Cf2 = 0.f;
    memset(d->co1, 0, sizeof d->co1); memset(d->calT1, 0, sizeof d->calT1); memset(d->co2, 0, sizeof d->co2); memset(d->calT2, 0, sizeof d->calT2);
    memset(d->calH, 0, sizeof d->calH); memset(d->mtxH, 0, sizeof d->mtxH); memset(d->corHp, 0, sizeof d->corHp); memset(d->corHt, 0, sizeof d->corHt);
    memset(d->calP, 0, sizeof d->calP);
    *out = d;
    return 0;
}

void sonde_rs41_dec_destroy(sonde_rs41_dec_t *d) { delete d; }

int sonde_rs41_dec_frame(sonde_rs41_dec_t *d, const sonde_frame_t *f, char *out, size_t outlen) {
    if (!d || !f || !out || outlen < 1) return SONDE_E_ARG;
    memcpy(d->fr, f->frame, FL);
    d->aux = 0;
    Out w;
    if (f->ecc >= 0) d->run_good(w, f->ecc); else d->run_failed(w, f->ecc);
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

int sonde_rs41_dec_ecc(sonde_rs41_dec_t *d, int level, int inv, const float *soft0, const float *soft1, int nbits, float ts, sonde_frame_t *f) {
    if (!d || !f || !soft0 || level < 1 || level > 4 || nbits < 0) return SONDE_E_ARG;
    int len = 0, nb = 0;
    f->ecc = d->ecc34(level, inv, soft0, soft1 ? soft1 : soft0, nbits, ts, f->frame, &len, &nb);
    f->len = len; f->nbytes = nb;
    return 0;
}

int sonde_rs41_dec_fields(const sonde_rs41_dec_t *d, sonde_rs41_fields_t *o) {
    if (!d || !o) return SONDE_E_ARG;
    memset(o, 0, sizeof *o);
    o->frame_nr = d->frnr; memcpy(o->id, d->id, 9);
    o->year = d->year; o->month = d->month; o->day = d->day; o->hour = d->hour; o->minute = d->minute; o->second = d->sec; o->is_utc = d->isUTC;
    o->lat = d->lat; o->lon = d->lon; o->alt = d->alt; o->vel_h = d->vH; o->heading = d->vD; o->vel_v = d->vV;
    o->sats = d->numSV; o->batt = d->batt; o->temp = d->T; o->humidity = d->o.ptu == 2 ? d->RH2 : d->RH; o->pressure = d->P;
    strncpy(o->subtype, *d->rstyp ? d->rstyp : "RS41", sizeof o->subtype - 1);
    o->tx_freq_khz = (int)d->freq; o->crc_fail_mask = (int)d->crc;
    o->have_id = d->have_id; o->have_time = d->have_time; o->have_pos = d->have_pos;
    return 0;
}

}  // extern "C"
