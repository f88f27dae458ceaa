// sonde_rs92_fields.cpp — Vaisala RS92 frames -> the reference's text / JSON (include/sonde_rs92.h).  Host code, bit rate.
//
// One object = the gpx_t of demod/mod/rs92mod.c: frame bytes, calibration rows, satellite table and the last solution persist from frame to
// frame as the reference's do.  The position is solved here from the raw ranges (sonde_gpsnav.h).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_rs92.h"
#include "sonde_gpsnav.h"
#include "sonde_host.h"

namespace {

using sonde::gpsnav::Eph;
using sonde::gpsnav::Sat;

constexpr int FRAME_LEN = SONDE_RS92_FRAME_LEN, FRAMESTART = 6, BITS = 10, HDRLEN = 60;
constexpr int RS_R = 24, MSGPOS = 6, MSGLEN = FRAME_LEN - 6 - RS_R, PARPOS = FRAME_LEN - RS_R;      // cfg_rs92 :77
const char kRawHeader[] = "10100110011001101001" "1010011001100110100110101010100110101001";           // 2A 2A 10 (:88-92)
const uint8_t kHeaderBytes[6] = { 0x2A, 0x2A, 0x2A, 0x2A, 0x2A, 0x10 };

// block positions (:243-271): id byte, length in words, payload, CRC-16
constexpr int POS_FRAMENB = 0x08, POS_SONDEID = 0x0C, POS_CALDATA = 0x17, POS_CALFREQ = 0x1A, LEN_CFG = 2 * 0x10;
constexpr int POS_PTU = 0x2C, LEN_PTU = 2 * 0x0C;
constexpr int POS_GPS_TOW = 0x48, POS_GPS_PRN = 0x4E, POS_GPS_STATUS = 0x56, POS_GPS_DATA = 0x62, LEN_GPS = 2 * 0x3D;
constexpr int POS_AUX = 0xC6, POS_AUXDATA = 0xC8, LEN_AUX = 2 * 0x05;
constexpr uint32_t CRC_FRAME = 1, CRC_PTU = 2, CRC_GPS = 4, CRC_AUX = 8;
constexpr int WEEKSEC = 604800;
enum { RS92SGP = 0, RS92NGP = 2 };

// calibration bytes 0x170..0x17F of every RS92-SGP: the end of coefficient 0x97 and coefficients 0x98..0x9A (:339-340); an RS92-NGP sends the
// same values under its 16-byte key, which is how the key is found
const uint8_t kCal170[16] = { 0x36, 0x98, 0x92, 0x25, 0x6b, 0xb3, 0x99, 0xe1, 0x57, 0x05, 0x30, 0x9a, 0xfe, 0x51, 0xf4, 0xab };

const double kChip = 299792.458 / 1023.0 / 1024.0;      // range [m] = -chips * kChip: c / (1.023e6 chips/s) / 1024 (:971)
const double kL1 = 1575.42 / 1.023 / 4.0;               // L1 cycles per chip / 4 = 385: delta chips -> range rate (:973)

const char kWeekday[7][4] = { "Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat" };

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[640]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

float poly5(float x, const float *a) { return ((((a[5] * x + a[4]) * x + a[3]) * x + a[2]) * x + a[1]) * x + a[0]; }
float nu(float t, float t0, float y0) { const float y = t / t0; return 1.0f / (y0 - y); }        // 1/f differences against the references

struct Range { uint32_t tow; uint8_t status; int chips, deltachips; };

}  // namespace

struct sonde_rs92_dec {
    sonde_rs92_opts_t o{};
    // --- gpx_t :119-154
    int frnr = 0;
    char id[11] = {0};
    int week = 0, gpssec = 0, jahr = 0, monat = 0, tag = 0, wday = 0, std_ = 0, min_ = 0;
    float sek = 0;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vU = 0, dop = 0, diter = 0;
    int sats[4] = {0, 0, 0, 0};
    uint16_t conf_kt = 0;
    int freq = 0;
    uint32_t crc = 0;
    uint8_t frame[FRAME_LEN];
    uint8_t cal_state[2] = {0, 0}, calfrms = 0, calibytes[32 * 16], calfrchk[32];
    float cal_f32[256];
    float T = 0, _RH = 0, RH = 0, _P = 0, P = 0;
    uint8_t xcal16[16], xptu16[16];
    int rs_type = RS92SGP, ngp = 0;
    unsigned short aux[4] = {0, 0, 0, 0};
    // --- GPS_t :99-117
    int vergps = 0, iter = 0, vel = 0, exsat = -1, almanac = 0, ephem = 0;
    float dop_limit = 9.9f, d_err = 10000;
    bool der_given = false;
    uint8_t week1024epoch = 1, sat_status[12], prn[12], prn32toggle = 1, prn32next = 0;
    Eph alm[33];
    std::vector<Eph> ephs;
    Sat sat[33], sat1s[33];
    // --- soft input
    float sbuf[HDRLEN]; int bufpos = -1, in_frame = 0, nsym = 0, byte_count = FRAMESTART, b8pos = 0; float s1 = 0; char bitbuf[BITS];

    sonde_rs92_dec() {
        memset(frame, 0, sizeof frame); memcpy(frame, kHeaderBytes, 6);
        memset(calibytes, 0, sizeof calibytes); memset(calfrchk, 0, sizeof calfrchk); memset(cal_f32, 0, sizeof cal_f32);
        memset(xcal16, 0, sizeof xcal16); memset(xptu16, 0, sizeof xptu16);
        memset(sat_status, 0, sizeof sat_status); memset(prn, 0, sizeof prn);
        memset(sbuf, 0, sizeof sbuf); memset(bitbuf, 0, sizeof bitbuf);
    }

    int crc16(int start, int len) const {                       // :274-295
        if (start + len >= FRAME_LEN) return -1;
        return sonde::crc16(frame + start, len);
    }
    int block_crc(int pos, int len, uint32_t flag) {            // stored low byte first behind the payload; -2 when it does not hold (--crc is always on, :1865)
        const int in_frame = frame[pos + len] | (frame[pos + len + 1] << 8);
        const int c = crc16(pos, len);
        if (in_frame != c) { crc |= flag; return -2; }
        return 0;
    }

    // ---- date of the GPS week / second (:217-234) --------------------------------------------------------------------------------------
    void gps2date() {
        const long GpsDays = week * 7 + (gpssec / 86400);
        const long Mjd = 44244 + GpsDays;
        long J = Mjd + 2468570;
        const long C = 4 * J / 146097;
        J = J - (146097 * C + 3) / 4;
        const long Y = 4000 * (J + 1) / 1461001;
        J = J - 1461 * Y / 4 + 31;
        const long M = 80 * J / 2447;
        tag = (int)(J - 2447 * M / 80);
        J = M / 11;
        monat = (int)(M + 2 - (12 * J));
        jahr = (int)(100 * (C - 49) + Y + J);
    }

    // ---- config block: frame number, id, one calibration row per frame (:297-545) -------------------------------------------------------
    void xor_ptu() {                                            // hash over calibration bytes 0x24.. -> the RS92-NGP's key for the measurement block (:367-419)
        const uint8_t *pcal = calibytes + 0x24;
        for (int j = 0; j < 8; j++) {
            uint32_t a = 0x1d89;
            for (int k = 0; k < 4; k++) {
                a += pcal[j + k];
                a += a << 10;
                a ^= a >> 6;
            }
            a += a << 3;
            a ^= a >> 11;
            a += a << 15;
            xptu16[2 * j] = a & 0xFF;
            xptu16[2 * j + 1] = (a >> 8) & 0xFF;
        }
    }
    int config_block(Out &w) {
        const int bad = block_crc(POS_FRAMENB, LEN_CFG, CRC_FRAME);
        if (bad) return bad;
        uint8_t sid[10];
        for (int i = 0; i < 8; i++) {
            const uint8_t b = frame[POS_SONDEID + i];
            if (b < 0x20 || b > 0x7E) return -1;
            sid[i] = b;
        }
        sid[8] = 0;
        if (strncmp(id, (const char *)sid, 8) != 0) {           // another sonde: forget the calibration
            memset(calibytes, 0, sizeof calibytes); memset(calfrchk, 0, sizeof calfrchk); memset(cal_f32, 0, sizeof cal_f32);
            calfrms = 0;
            T = -275.15f; _RH = -1.0f; _P = -1.0f; RH = -1.0f; P = -1.0f;
            memcpy(id, sid, 8);
        }
        memcpy(cal_state, frame + POS_FRAMENB + 12, 2);
        const uint8_t calfr = frame[POS_CALDATA];
        if (calfr < 32 && calfrchk[calfr] == 0) {
            memcpy(calibytes + calfr * 16, frame + POS_CALDATA + 1, 16);
            calfrchk[calfr] = 1;
        }
        if (calfrms < 32) {
            calfrms = 0;
            for (int i = 0; i < 32; i++) calfrms += (calfrchk[i] > 0);
        }
        if (calfrms != 32) return 0;
        // all 32 rows are in for the first time (the count moves on to 33 and this runs once per sonde): keys, type, the 66 coefficients behind 0x40
        calfrms += 1;
        xor_ptu();
        if (o.dbg) { w.f("XPTU:"); for (int j = 0; j < 16; j++) w.f(" %02X", xptu16[j]); w.f("\n"); }
        const uint8_t *p = calibytes + 0x170, *q = kCal170;
        for (int k = 0; k < 3; k++) {                           // index byte in place, the float's bytes are stored (2, 0, 1, 3) on an NGP
            xcal16[5 * k] = p[5 * k] ^ q[5 * k];
            xcal16[5 * k + 1] = p[5 * k + 1] ^ q[5 * k + 1];
            xcal16[5 * k + 3] = p[5 * k + 3] ^ q[5 * k + 2];
            xcal16[5 * k + 4] = p[5 * k + 4] ^ q[5 * k + 3];
            xcal16[5 * k + 2] = p[5 * k + 2] ^ q[5 * k + 4];
        }
        xcal16[15] = p[15] ^ q[15];
        if (o.dbg) { w.f("XCAL:"); for (int j = 0; j < 16; j++) w.f(" %02X", xcal16[j]); w.f("\n"); }
        rs_type = memcmp(calibytes + 0x170, kCal170, 16) == 0 ? RS92SGP : RS92NGP;       // chk_toggle_type :342-365
        if ((rs_type == RS92SGP && ngp) || (rs_type == RS92NGP && !ngp)) ngp ^= 1;
        uint8_t xcal[66 * 5];
        for (int j = 0; j < 66 * 5; j++) {
            xcal[j] = calibytes[0x40 + j];
            if (ngp) xcal[j] ^= xcal16[j % 16];
        }
        for (int j = 0; j < 66; j++) {
            const uint8_t idx = xcal[5 * j], *dat = xcal + (5 * j + 1);
            const uint32_t le = dat[0] | (dat[1] << 8) | (dat[2] << 16) | ((uint32_t)dat[3] << 24);
            const uint32_t xx = dat[1] | (dat[2] << 8) | (dat[0] << 16) | ((uint32_t)dat[3] << 24);
            const uint32_t bits = ngp ? xx : le;
            float v;
            memcpy(&v, &bits, 4);
            cal_f32[idx] = v;
            if (o.dbg && (idx / 10 == 3 || idx / 10 == 4 || idx / 10 == 5)) {
                w.f(" %3d :", idx);
                for (int i = 1; i < 5; i++) w.f(" %02x", xcal[5 * j + i]);
                w.f(" : %f", v);
                w.f("\n");
            }
        }
        return 0;
    }

    // ---- PTU block: eight 24-bit counts -> T, RH, p through the calibration polynomials (:565-647) -----------------------------------------
    int measurements() {
        uint8_t *m = frame + POS_PTU;
        if (ngp && (crc & CRC_FRAME)) return -2;
        for (int j = 0; j < 24; j++) {
            uint8_t b = m[j];
            if (ngp) { b ^= frame[POS_FRAMENB + (j & 1)]; b ^= xptu16[j % 16]; }
            m[j] = b;
        }
        uint32_t ch[8];
        for (int k = 0; k < 8; k++) ch[k] = m[3 * k] | (m[3 * k + 1] << 8) | (m[3 * k + 2] << 16);
        const uint32_t temp = ch[0], hum1 = ch[1], hum2 = ch[2], ref1 = ch[3], pres = ch[5], ref3 = ch[6], ref4 = ch[7];
        if (calfrms > 0x20) {
            float x = nu((float)(ref1 - temp), (float)(ref1 - ref4), cal_f32[37]);
            const float t = poly5(x, cal_f32 + 30);
            if (t > -120.0f && t < 80.0f) T = t; else T = -273.15f;
            x = nu((float)(ref1 - hum1), (float)(ref1 - ref3), cal_f32[47]);
            const float U1 = poly5(x, cal_f32 + 40);
            x = nu((float)(ref1 - hum2), (float)(ref1 - ref3), cal_f32[57]);
            const float U2 = poly5(x, cal_f32 + 50);
            _RH = U1 > U2 ? U1 : U2;
            if (_RH < 0.0f) _RH = 0.0f;
            if (_RH > 100.0f) _RH = 100.0f;
            x = nu((float)(ref1 - pres), (float)(ref1 - ref4), cal_f32[17]);
            _P = poly5(x, cal_f32 + 10);
        }
        return 0;
    }
    int ptu_block() {
        int ret = block_crc(POS_PTU, LEN_PTU, CRC_PTU);
        if (ret == 0 && calfrms > 0x20) ret = measurements();
        return ret;
    }

    // ---- GPS block (:652-693, :777-843, :845-959, :975-1105) ---------------------------------------------------------------------------
    uint32_t tow_ms() const { uint32_t t; memcpy(&t, frame + POS_GPS_TOW, 4); return t; }
    int gps_time() {
        const int ret = block_crc(POS_GPS_TOW, LEN_GPS, CRC_GPS);
        uint32_t t = tow_ms();
        const int ms = t % 1000;
        t /= 1000;
        gpssec = (int)t;
        wday = (t / (24 * 3600)) % 7;
        t %= (24 * 3600);
        std_ = t / 3600;
        min_ = (t % 3600) / 60;
        sek = (float)(t % 60 + ms / 1000.0);
        return ret;
    }
    int aux_block() {
        const int ret = block_crc(POS_AUX, LEN_AUX, CRC_AUX);
        for (int i = 0; i < 4; i++) aux[i] = (unsigned short)(frame[POS_AUXDATA + 2 * i] + (frame[POS_AUXDATA + 2 * i + 1] << 8));
        return ret;
    }

    // twelve 5-bit numbers in four 16-bit words (3 x 5 bits + one spare bit); PRN 32 does not fit: it is sent as 0 with the bit above it set,
    // which is the lowest bit of the next number, or the word's spare bit in the third column (:777-843)
    void prn_numbers(uint8_t prns[12]) {
        uint8_t le[64];
        memset(le, 0, sizeof le);
        for (int blk = 0; blk < 4; blk++) {
            uint16_t wd = (uint16_t)(frame[POS_GPS_PRN + 2 * blk] | (frame[POS_GPS_PRN + 2 * blk + 1] << 8));
            for (int i = 0; i < 15; i++) { le[15 * blk + i] = wd & 1; wd >>= 1; }
            le[60 + blk] = wd & 1;
        }
        uint8_t ind32 = 32;
        for (int i = 0; i < 12; i++) {
            prns[i] = 0;
            for (int j = 0, d = 1; j < 5; j++, d <<= 1) if (le[5 * i + j]) prns[i] += d;
        }
        for (int i = 0; i < 12; i++) {
            if (prns[i] == 0 && (sat_status[i] & 0x0F)) {
                if ((i % 3 == 2 && (le[60 + i / 3] & 1)) || (i % 3 != 2 && (le[5 * (i + 1)] & 1))) { prns[i] = 32; ind32 = i; }
            }
            else if ((sat_status[i] & 0x0F) == 0) prns[i] = 0;
        }
        prn32next = 0;
        if (ind32 < 12 && ind32 % 3 != 2) {
            // the number behind PRN 32 lost its lowest bit to the overflow: it is either 1 or even; the toggle remembers which reading did not
            // collide with another satellite of the list
            const int nx = ind32 + 1;
            if ((sat_status[nx] & 0x0F) && prns[nx] > 1) {
                int j;
                for (j = 0; j < ind32; j++) if (prns[j] == (prns[nx] ^ prn32toggle) && (sat_status[j] & 0x0F)) break;
                if (j < ind32) prn32toggle ^= 0x1;
                else {
                    for (j = ind32 + 2; j < 12; j++) if (prns[j] == (prns[nx] ^ prn32toggle) && (sat_status[j] & 0x0F)) break;
                    if (j < 12) prn32toggle ^= 0x1;
                }
                prns[nx] ^= prn32toggle;
            }
            prn32next = prns[nx];
        }
    }

    void sat_positions_alm(double t, Sat *satp) {
        for (int j = 1; j < 33; j++) {
            if (!(alm[j].prn > 0 && alm[j].health == 0)) continue;
            int rollover = 0;
            if (t - alm[j].toa > WEEKSEC / 2) rollover = +1;
            else if (t - alm[j].toa < -WEEKSEC / 2) rollover = -1;
            const int wk = alm[j].week - rollover;
            week = wk + week1024epoch * 1024;
            if (alm[j].prn > 32) continue;
            sonde::gpsnav::sat_state((unsigned short)wk, t, alm[j], vel >= 2, satp[alm[j].prn]);
        }
    }
    void sat_positions_eph(double t, Sat *satp) {
        for (int j = 1; j < 33; j++) {
            int found = 0, pick = 0, wk = 0;
            double tdiff = WEEKSEC;
            for (size_t c = 0; ephs[c].prn > 0; c++) {
                const Eph &e = ephs[c];
                if (e.prn != j || e.health != 0) continue;
                found += 1;
                int rollover = 0;
                if (t - e.toe > WEEKSEC / 2) rollover = +1;
                else if (t - e.toe < -WEEKSEC / 2) rollover = -1;
                const double td = fabs(t - e.toe - rollover * WEEKSEC);
                if (td < tdiff) {                               // the entry nearest in time
                    tdiff = td;
                    wk = e.week - rollover;
                    week = e.gpsweek - rollover;
                    pick = (int)c;
                }
            }
            if (!found) continue;
            sonde::gpsnav::sat_state((unsigned short)wk, t, ephs[pick], vel >= 2, satp[j]);
            satp[j].ephtime = ephs[pick].toe;
        }
    }

    int pseudoranges() {
        uint8_t prns[12];
        Range range[33];
        memset(prns, 0, sizeof prns);
        memset(range, 0, sizeof range);
        const uint32_t gpstime = tow_ms();
        for (int i = 0; i < 12; i++) sat_status[i] = frame[POS_GPS_STATUS + i];
        prn_numbers(prns);

        if (almanac) sat_positions_alm(gpstime / 1000.0, sat);
        if (ephem) sat_positions_eph(gpstime / 1000.0, sat);
        if (vel == 1) {                                         // --vel1: the same a second earlier, velocity = difference of two positions
            if (almanac) sat_positions_alm(gpstime / 1000.0 - 1, sat1s);
            if (ephem) sat_positions_eph(gpstime / 1000.0 - 1, sat1s);
        }

        int k = 0;
        for (int j = 0; j < 12; j++) {
            const uint8_t *d = frame + POS_GPS_DATA + 8 * j;
            uint32_t chipbytes, deltabytes = 0;
            memcpy(&chipbytes, d, 4);
            memcpy(&deltabytes, d + 4, 3);
            Range &r = range[prns[j]];
            r.tow = gpstime;
            r.status = sat_status[j];
            if (chipbytes == 0x7FFFFFFF || chipbytes == 0x55555555) { r.chips = 0; continue; }
            if (vergps != 8 && chipbytes > 0x10000000 && chipbytes < 0xF0000000) { r.chips = 0; continue; }
            r.chips = (int)chipbytes;
            r.deltachips = (int)deltabytes;
            const Sat &s = sat[prns[j]];
            if (prns[j] > 0 && (sat_status[j] & 0x0F) == 0xF && sonde::gpsnav::dist3(s.X, s.Y, s.Z, 0, 0, 0) > 6700000) {
                int i;
                for (i = 0; i < k; i++) if (prn[i] == prns[j]) break;
                if (i == k && prns[j] != exsat) prn[k++] = prns[j];
            }
        }
        for (int j = 0; j < 12; j++) {
            const Range &r = range[prns[j]];
            sat[prns[j]].pseudorange = -r.chips * kChip;
            sat1s[prns[j]].pseudorange = -(r.chips - r.deltachips / kL1) * kChip;
            sat[prns[j]].pseudorate = -r.deltachips * kChip / kL1;
            sat[prns[j]].prn = prns[j];
            sat1s[prns[j]].prn = prns[j];
        }
        double pr0 = (double)0x01400000;
        for (int j = 0; j < k; j++) {
            const double prj = sat[prn[j]].pseudorange + sat[prn[j]].clock_corr;
            if (prj < pr0) pr0 = prj;
        }
        for (int j = 0; j < k; j++) sat[prn[j]].PR = sat[prn[j]].pseudorange + sat[prn[j]].clock_corr - pr0 + 20e6;
        for (int j = 0; j < k; j++) sat1s[prn[j]].PR = sat1s[prn[j]].pseudorange + sat[prn[j]].clock_corr - pr0 + 20e6;
        return k;
    }

    static void ecef_velocity(double la, double lo, const double v[3], double *vH_, double *vD_, double *vU_) {      // :1107-1122
        const double phi = la * M_PI / 180.0, lam = lo * M_PI / 180.0;
        const double vN = -v[0] * sin(phi) * cos(lam) - v[1] * sin(phi) * sin(lam) + v[2] * cos(phi);
        const double vE = -v[0] * sin(lam) + v[1] * cos(lam);
        *vU_ = v[0] * cos(phi) * cos(lam) + v[1] * cos(phi) * sin(lam) + v[2] * sin(phi);
        *vH_ = sqrt(vN * vN + vE * vE);
        *vD_ = atan2(vE, vN) * 180 / M_PI;
        if (*vD_ < 0) *vD_ += 360;
    }

    // position from N satellites (:1124-1351): returns the number of 4-satellite solutions found, or N for -g2
    int solve(Out &w, int N) {
        using namespace sonde::gpsnav;
        double la = 0, lo = 0, al = 0, bias = 0, h = 0, d = 0, u = 0;
        double pos[3] = {0, 0, 0}, pos1s[3] = {0, 0, 0}, dpos[3] = {0, 0, 0}, velo[3] = {0, 0, 0}, dvel[3] = {0, 0, 0};
        double gdop = 0, gdop0 = 1000.0, DOP[4] = {0, 0, 0, 0}, dit = 0;
        int num = 0, exN = -1;

        if (vergps == 8) {
            w.f("  sats: ");
            for (int j = 0; j < N; j++) w.f("%02d ", prn[j]);
            w.f("\n");
        }
        lat = lon = alt = 0;

        if (vergps != 2) {
            int ix[4];
            for (ix[0] = 0; ix[0] < N; ix[0]++) for (ix[1] = ix[0] + 1; ix[1] < N; ix[1]++) for (ix[2] = ix[1] + 1; ix[2] < N; ix[2]++) for (ix[3] = ix[2] + 1; ix[3] < N; ix[3]++) {
                Sat A[4];
                for (int q = 0; q < 4; q++) A[q] = sat[prn[ix[q]]];
                if (closed_form4(A, &la, &lo, &al, &bias, pos) != 0) continue;
                num += 1;
                if (sonde::gpsnav::dop(4, A, pos, DOP) == 0) {
                    gdop = sqrt(DOP[0] + DOP[1] + DOP[2] + DOP[3]);
                    lin_pos(4, A, pos, bias, dpos, &bias);
                    dit = dist3(0, 0, 0, dpos[0], dpos[1], dpos[2]);
                    for (int j = 0; j < 3; j++) pos[j] += dpos[j];
                    ecef2elli(pos[0], pos[1], pos[2], &la, &lo, &al);
                    if (vel == 4) {
                        velo[0] = velo[1] = velo[2] = 0;
                        lin_vel(4, A, pos, velo, 0.0, dvel, &bias);
                        for (int j = 0; j < 3; j++) velo[j] += dvel[j];
                        lin_vel(4, A, pos, velo, bias, dvel, &bias);
                        for (int j = 0; j < 3; j++) velo[j] += dvel[j];
                        ecef_velocity(la, lo, velo, &h, &d, &u);
                    }
                    if (vergps == 8 && gdop < dop_limit) {
                        w.f("       ");
                        w.f("lat: %.5f , lon: %.5f , alt: %.1f ", la, lo, al);
                        w.f(" (d:%.1f)", dit);
                        if (vel == 4) w.f("  vH: %4.1f  D: %5.1f  vV: %3.1f ", h, d, u);
                        w.f("  sats: ");
                        w.f("%02d %02d %02d %02d  ", prn[ix[0]], prn[ix[1]], prn[ix[2]], prn[ix[3]]);
                        w.f(" GDOP : %.1f  ", gdop);
                        w.f("\n");
                    }
                }
                else gdop = -1;
                if (gdop > 0 && gdop < gdop0) {                 // best geometry wins
                    lat = la; lon = lo; alt = al;
                    dop = gdop; diter = dit;
                    for (int q = 0; q < 4; q++) sats[q] = prn[ix[q]];
                    gdop0 = gdop;
                    if (vel == 4) { vH = h; vD = d; vU = u; }
                }
            }
        }

        if (vergps == 8 || vergps == 2) {
            Sat B[12], B1s[12], C[12];
            for (int j = 0; j < N; j++) B[j] = sat[prn[j]];
            for (int j = 0; j < N; j++) B1s[j] = sat1s[prn[j]];
            bancroft(N, B, pos, &bias);
            ecef2elli(pos[0], pos[1], pos[2], &la, &lo, &al);
            gdop = -1;
            if (sonde::gpsnav::dop(N, B, pos, DOP) == 0) gdop = sqrt(DOP[0] + DOP[1] + DOP[2] + DOP[3]);
            lin_pos(N, B, pos, bias, dpos, &bias);
            if (iter) {
                for (int j = 0; j < 3; j++) pos[j] += dpos[j];
                ecef2elli(pos[0], pos[1], pos[2], &la, &lo, &al);
            }
            diter = dist3(0, 0, 0, dpos[0], dpos[1], dpos[2]);

            if (diter > d_err && N > 5) {                       // one satellite with bad data? leave each out in turn
                for (int n = 0; n < N; n++) {
                    int k = 0;
                    for (int j = 0; j < N; j++) if (j != n) C[k++] = B[j];
                    double pos0[3] = {0, 0, 0}, la0, lo0, al0;
                    bancroft(N - 1, C, pos0, &bias);
                    lin_pos(N - 1, C, pos0, bias, dpos, &bias);
                    dit = dist3(0, 0, 0, dpos[0], dpos[1], dpos[2]);
                    ecef2elli(pos0[0], pos0[1], pos0[2], &la0, &lo0, &al0);
                    if (dit < diter) {
                        diter = dit;
                        for (int j = 0; j < 3; j++) pos[j] = pos0[j];
                        la = la0; lo = lo0; al = al0;
                        exN = n;
                    }
                }
                if (exN >= 0) {
                    if (prn[exN] == prn32next) prn32toggle ^= 0x1;
                    for (int k = exN; k < N - 1; k++) {
                        B[k] = B[k + 1];
                        prn[k] = prn[k + 1];
                        if (vel == 1) B1s[k] = B1s[k + 1];
                    }
                    N = N - 1;
                    if (sonde::gpsnav::dop(N, B, pos, DOP) == 0) gdop = sqrt(DOP[0] + DOP[1] + DOP[2] + DOP[3]);
                }
            }

            if (vel == 1) {
                bancroft(N, B1s, pos1s, &bias);
                if (iter) {
                    lin_pos(N, B1s, pos1s, bias, dpos, &bias);
                    for (int j = 0; j < 3; j++) pos1s[j] += dpos[j];
                }
                for (int j = 0; j < 3; j++) velo[j] = pos[j] - pos1s[j];
                ecef_velocity(la, lo, velo, &h, &d, &u);
                double la1, lo1, al1;
                ecef2elli(pos1s[0], pos1s[1], pos1s[2], &la1, &lo1, &al1);
                if (vergps == 8) {
                    w.f("\ndeltachips1s lat: %.6f , lon: %.6f , alt: %.2f ", la1, lo1, al1);
                    w.f(" vH: %4.1f  D: %5.1f  vV: %3.1f ", h, d, u);
                    w.f("\n");
                }
            }
            if (vel >= 2) {
                velo[0] = velo[1] = velo[2] = 0;
                lin_vel(N, B, pos, velo, 0.0, dvel, &bias);
                for (int j = 0; j < 3; j++) velo[j] += dvel[j];
                ecef_velocity(la, lo, velo, &h, &d, &u);
            }
            if (vergps == 8) {
                w.f("bancroft[%2d] lat: %.6f , lon: %.6f , alt: %.2f ", N, la, lo, al);
                w.f(" (d:%.1f)", diter);
                if (vel) w.f("  vH: %4.1f  D: %5.1f  vV: %3.1f ", h, d, u);
                w.f("  DOP[");
                for (int j = 0; j < N; j++) {
                    w.f("%d", prn[j]);
                    if (j < N - 1) w.f(","); else w.f("] %.1f ", gdop);
                }
                w.f("\n");
            }
            if (vergps == 2) {
                lat = la; lon = lo; alt = al;
                dop = gdop;
                num = N;
                if (vel) { vH = h; vD = d; vU = u; }
            }
        }
        return num;
    }

    // ---- calibration row of the frame (:717-771) -----------------------------------------------------------------------------------------
    void cal_row(Out &w) {
        const uint8_t calfr = frame[POS_CALDATA];
        if (o.verbose == 4) {
            w.f("\n");
            w.f("[%5d] ", frnr);
            w.f("  0x%02x:", calfr);
            for (int i = 0; i < 16; i++) w.f(" %02x", frame[POS_CALDATA + 1 + i]);
            w.f((crc & CRC_FRAME) == 0 ? " [OK]" : " [NO]");
        }
        if (o.aux && o.verbose == 4) {
            w.f("  #  ");
            for (int i = 0; i < 8; i++) w.f("%02x ", frame[POS_AUXDATA + i]);
        }
        if (calfr == 0x00) {
            const unsigned f = frame[POS_CALFREQ] + (frame[POS_CALFREQ + 1] << 8);
            freq = (ngp ? 1600000 : 400000) + 10 * (int)f;     // kHz
            w.f(": fq %d", freq);
            const uint16_t kt = (uint16_t)(frame[POS_CALFREQ + 2] + (frame[POS_CALFREQ + 3] << 8));
            if (kt < 0xFFFF && o.verbose == 4) w.f("; KT:%ds", kt);
            conf_kt = kt;
        }
    }

    // ---- one frame (:1389-1575) ---------------------------------------------------------------------------------------------------------
    int rs_correct(int msglen) {
        uint8_t cw[255];
        memset(cw, 0, sizeof cw);
        if (msglen > FRAME_LEN) msglen = FRAME_LEN;
        for (int i = msglen; i < FRAME_LEN; i++) frame[i] = 0;
        memcpy(cw, frame + PARPOS, RS_R);
        memcpy(cw + RS_R, frame + MSGPOS, MSGLEN);
        const int errors = sonde::rs255_decode(cw);
        memcpy(frame + PARPOS, cw, RS_R);
        memcpy(frame + MSGPOS, cw + RS_R, MSGLEN);
        return errors;
    }

    void print_position(Out &w, int ec) {
        int n = 0;
        frnr = frame[POS_FRAMENB] + (frame[POS_FRAMENB + 1] << 8);
        const int err1 = config_block(w);
        const int err2 = ptu_block();
        const int err3 = gps_time();
        aux_block();
        if (!err3 && (almanac || ephem)) {
            const int k = pseudoranges();
            if (k >= 4) n = solve(w, k);
        }
        if (err1) return;

        w.f("[%5d] ", frnr);
        w.f("(%s) ", id);
        if (!err3) {
            if (almanac || ephem) {
                gps2date();
                w.f("(%04d-%02d-%02d) ", jahr, monat, tag);
            }
            w.f("%s ", kWeekday[wday]);
            w.f("%02d:%02d:%06.3f", std_, min_, sek);
            if (n > 0) {
                w.f(" ");
                if (almanac) w.f(" lat: %.4f  lon: %.4f  alt: %.1f ", lat, lon, alt);
                else         w.f(" lat: %.5f  lon: %.5f  alt: %.1f ", lat, lon, alt);
                if (o.verbose && vergps != 8) w.f(" (d:%.1f)", diter);
                if (vel) w.f("  vH: %4.1f  D: %5.1f  vV: %3.1f ", vH, vD, vU);
                if (o.verbose) {
                    if (vergps != 2) w.f(" DOP[%02d,%02d,%02d,%02d] %.1f", sats[0], sats[1], sats[2], sats[3], dop);
                    else {
                        w.f(" DOP[");
                        for (int j = 0; j < n; j++) {
                            w.f("%d", prn[j]);
                            if (j < n - 1) w.f(","); else w.f("] %.1f ", dop);
                        }
                    }
                }
            }
        }
        if (!err2 && o.ptu) {
            w.f(" ");
            if (T > -273.0f) w.f(" T=%.1fC ", T);
            if (_RH > -0.5f) w.f(" _RH=%.0f%% ", _RH);
            if (_P > 0.0f) w.f(" _P=%.1fhPa ", _P);
        }
        if (o.aux && (o.verbose != 4 && (crc & CRC_AUX) == 0)) {         // (--crc is always on)
            if (aux[0] != 0 || aux[1] != 0 || aux[2] != 0 || aux[3] != 0) w.f(" # %04x %04x %04x %04x", aux[0], aux[1], aux[2], aux[3]);
        }
        w.f("  # ");
        w.f("[");
        for (int j = 0; j < 4; j++) w.f("%d", (crc >> j) & 1);
        w.f("]");
        if (o.ecc == 2) {
            if (ec > 0) w.f(" (%d)", ec);
            if (ec < 0) w.f(" (-)");
        }
        cal_row(w);

        if (o.json && (crc & (CRC_FRAME | CRC_GPS)) == 0 && (almanac || ephem)) {
            w.f("\n");
            w.f("{ \"type\": \"%s\"", "RS92");
            w.f(", \"frame\": %d, \"id\": \"%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, \"heading\": %.5f, \"vel_v\": %.5f",
                frnr, id, jahr, monat, tag, std_, min_, sek, lat, lon, alt, vH, vD, vU);
            if (o.ptu && !err2) {
                if (T > -273.0f) w.f(", \"temp\": %.1f", T);
                if (_RH > -0.5f) w.f(", \"humidity\": %.1f", _RH);
                if (_P > 0.0f) w.f(", \"pressure\": %.2f", _P);
            }
            if ((crc & CRC_AUX) == 0 && (aux[0] != 0 || aux[1] != 0 || aux[2] != 0 || aux[3] != 0))
                w.f(", \"aux\": \"%04x%04x%04x%04x\"", aux[0], aux[1], aux[2], aux[3]);
            w.f(", \"subtype\": \"RS92-%s\"", rs_type == RS92SGP ? "SGP" : "NGP");
            if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
            if (freq > 0) w.f(", \"tx_frequency\": %d", freq);
            w.f(", \"ref_datetime\": \"%s\"", "GPS");
            w.f(", \"ref_position\": \"%s\"", "GPS");
            if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
            w.f(" }\n");
        }
        w.f("\n");
    }

    void print_frame(Out &w, int len) {
        crc = 0;
        const int ec = rs_correct(len);                         // (ecc is at least 1, :1866)
        for (int i = len; i < FRAME_LEN; i++) frame[i] = 0;
        if (o.raw) {
            for (int i = 0; i < len; i++) w.f("%02x", frame[i]);
            if (o.verbose) {
                w.f(" ");
                w.f(ec >= 0 ? " [OK]" : " [NO]");
                if (ec > 0) w.f(" (%d)", ec);
                if (ec < 0) w.f(" (-)");
            }
            w.f("\n");
        }
        else print_position(w, ec);
    }
};

extern "C" {

int sonde_rs92_dec_create(const sonde_rs92_opts_t *opts, sonde_rs92_dec_t **out) {
    if (!opts || !out) return SONDE_E_ARG;
    sonde_rs92_dec *d = new sonde_rs92_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.ecc < 2) d->o.ecc = 1;
    d->ngp = opts->ngp != 0;
    d->rs_type = d->ngp ? RS92NGP : RS92SGP;
    d->vergps = opts->gps_verbose; d->iter = opts->gps_iter; d->vel = opts->gps_vel;
    d->exsat = (opts->exsat >= 1 && opts->exsat <= 32) ? opts->exsat : -1;
    d->week1024epoch = (opts->gpsepoch >= 0 && opts->gpsepoch <= 4) ? (uint8_t)opts->gpsepoch : 1;
    d->dop_limit = (opts->dop_limit > 0 && opts->dop_limit < 100) ? opts->dop_limit : 9.9f;
    if (opts->d_err > 0 && opts->d_err < 100000) { d->d_err = opts->d_err; d->der_given = true; }
    *out = d;
    return 0;
}

void sonde_rs92_dec_destroy(sonde_rs92_dec_t *d) { delete d; }

int sonde_rs92_dec_load_almanac(sonde_rs92_dec_t *d, const char *path) {
    if (!d || !path) return SONDE_E_ARG;
    FILE *fp = fopen(path, "r");
    if (!fp) return SONDE_E_ARG;
    const int rc = sonde::gpsnav::read_sem_almanac(fp, d->alm);
    fclose(fp);
    if (rc == 0 && !d->ephem) d->almanac = 1;
    if (!d->der_given && !d->ephem) d->d_err = 4000;
    return rc == 0 ? 0 : SONDE_E_ARG;
}

int sonde_rs92_dec_load_ephemeris(sonde_rs92_dec_t *d, const char *path) {
    if (!d || !path) return SONDE_E_ARG;
    FILE *fp = fopen(path, "rb");
    if (!fp) return SONDE_E_ARG;
    const bool ok = sonde::gpsnav::read_rinex_nav(fp, d->ephs);
    fclose(fp);
    if (ok) { d->ephem = 1; d->almanac = 0; }
    if (!d->der_given) d->d_err = 1000;
    return ok ? 0 : SONDE_E_ARG;
}

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.data(), w.s.size()); out[w.s.size()] = 0;
    return (int)w.s.size();
}

static int byte_of(const char bits[BITS]) {                     // 8N1, LSB first; start and stop bit are not looked at (:183-196)
    int v = 0;
    for (int i = 1; i <= 8; i++) if (bits[i] == 1) v += 1 << (i - 1);
    return v;
}

int sonde_rs92_dec_frame(sonde_rs92_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen) {
    if (!d || !out || n < 0 || n > SONDE_RS92_FRAME_BITS || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    int count = FRAMESTART;
    char bits[BITS];
    for (int i = 0; i + BITS <= n; i += BITS) {
        for (int j = 0; j < BITS; j++) bits[j] = soft[i + j] >= 0.0f;
        d->frame[count++] = (uint8_t)byte_of(bits);
    }
    d->print_frame(w, count);
    return finish_out(w, out, outlen);
}

int sonde_rs92_dec_bytes(sonde_rs92_dec_t *d, const uint8_t *frame, int32_t len, char *out, size_t outlen) {
    if (!d || !out || len < 0 || len > FRAME_LEN || (len > 0 && !frame)) return SONDE_E_ARG;
    Out w;
    memcpy(d->frame, frame, (size_t)len);
    d->print_frame(w, len);
    return finish_out(w, out, outlen);
}

int sonde_rs92_dec_push_soft(sonde_rs92_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_frame) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % HDRLEN;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < HDRLEN; k++) {
                if (j >= HDRLEN) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kRawHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            const float mv = (float)sum;
            if (fabs(mv) > 0.8f) {
                memset(d->sbuf, 0, sizeof d->sbuf);              // the next search starts on an empty buffer (:1988), the write position stays
                if (mv * (0.5 - d->o.inv) < 0) continue;         // header of the other polarity: not this decoder's (:1999-2002)
                d->in_frame = 1; d->nsym = 0; d->byte_count = FRAMESTART; d->b8pos = 0;
            }
        } else {
            if ((d->nsym++ & 1) == 0) { d->s1 = s; continue; }
            int bit = (s - d->s1) >= 0.0;                       // both Manchester symbols (:2020-2021)
            if (d->o.inv) bit ^= 1;
            d->bitbuf[d->b8pos++] = (char)bit;
            if (d->b8pos >= BITS) {
                d->b8pos = 0;
                d->frame[d->byte_count++] = (uint8_t)byte_of(d->bitbuf);
                if (d->byte_count >= FRAME_LEN) { d->print_frame(w, d->byte_count); d->in_frame = 0; }
            }
        }
    }
    if (finish && d->in_frame) { d->print_frame(w, d->byte_count); d->in_frame = 0; }
    return finish_out(w, out, outlen);
}

}  // extern "C"
