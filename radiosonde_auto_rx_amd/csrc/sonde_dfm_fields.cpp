// sonde_dfm_fields.cpp — DFM06 / DFM09 / DFM17 / PS-15 telemetry and the text / JSON lines of the reference's dfm09mod
// (include/sonde_dfm.h).  Host side, bit-rate work: 2500 bit/s per sonde.
//
// A DFM frame carries one configuration packet (7 nibbles: channel number + 24-bit value) and two data packets (13 nibbles:
// 48 data bits + packet number 0..8).  What the reference does with them and where it is restated here:
//   data packets: time / position / velocity pieces per positioning mode, date in packet 8 ... Decoder::data_packet()  :347-505
//   configuration channels: measurement floats, serial number found as the value repeated in the
//        last / highest channel, sensor set and DFM type guessed from it ..................... Decoder::conf_packet()  :694-895
//   thermistor temperature ................................................................... Decoder::temperature()  :538-575
//   output once per packet 8: packets of the last 6 frames, --dist / --json gating, frame
//        counter consistency check, text line, satellites line, JSON ......................... Decoder::report()       :897-1150
//   per frame sequencing (which packets are looked at under which ECC verdict) ............... sonde_dfm_dec_frame()   :1238-1262
#include "../../include/sonde_dfm.h"
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

inline uint32_t field(const uint8_t *bits, int len) {            // big-endian bit field (bits2val)
    uint32_t v = 0;
    for (int j = 0; j < len; j++) v |= (uint32_t)bits[j] << (len - 1 - j);
    return v;
}
inline int popcount15(int v) { int n = 0; for (int i = 0; i < 15; i++) n += (v >> i) & 1; return n; }
inline float float24(int d) { return (d & 0xFFFFF) / (float)(1 << ((d >> 20) & 0xF)); }      // 4-bit exponent, 20-bit mantissa

const char *const kTypes[] = { "", "DFMxX", "DFM06", "DFM06P", "PS15", "DFM09", "DFM09P", "DFM17", "DFM17P" };
enum { T_UNDEF, T_UNKNOWN, T_06, T_06P, T_PS15, T_09, T_09P, T_17, T_17P };
constexpr int SNBIT = 0x100;

}  // namespace

struct sonde_dfm_dec {
    sonde_dfm_opts_t o;
    int inv = 0;                       // polarity of the frame being decoded
    // gpx_t
    int frnr = 0, sonde_typ = 0; uint32_t SN6 = 0, SN = 0; char SN_out[10] = {0};
    int week = 0, tow = 0; uint32_t sec_gps = 0;
    int year = 0, month = 0, day = 0, hour = 0, minute = 0; float sec = 0.f;
    double lat = 0, lon = 0, alt = 0, dir = 0, horiV = 0, vertV = 0, lat2 = 0, lon2 = 0, alt2 = 0, dir2 = 0, horiV2 = 0, vertV2 = 0;
    float T = 0.f, Rf = 0.f, frmcnt = 0.f, meas24[9] = {0}, status[3] = {0};
    uint32_t val24[9] = {0}; uint8_t have24[9] = {0};
    int posmode = 0; uint8_t xdata[26] = {0};
    int cfgchk = 0; char sonde_id[16] = {0};
    struct { uint8_t max_ch, nul_ch, sn_ch, chXbit; uint32_t SN_X, chX[2]; } snc = {0, 0, 0, 0, 0, {0, 0}};
    struct { int ec; float ts; } pck[9];
    int ptu_out = 0; char sensortyp = 0; int dfmtyp = T_UNDEF;
    struct { uint32_t prn; float dMSL; uint8_t nSV, nPRN; } gps = {0, 0.f, 0, 0};
    int prev_cntsec_diff = 0, prev_manpol = 0;

    char dat_str[9][14] = {};
    uint8_t hdr_bits[16] = { 0, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1 };      // gpx.frame[0..15]: the header to begin with (:1503-1506); not read for frame 0 of a hit, so there they are the previous frame's (:1649-1650,1710)

    void reset_cfg() { memset(have24, 0, sizeof have24); cfgchk = 0; ptu_out = 0; SN_out[0] = 0; T = -273.15f; }

    // ---- data packets ------------------------------------------------------------------------------------------------------
    int data_packet(const uint8_t *b, int ec) {
        const int id = (int)field(b + 48, 4);
        if (id >= 0 && id <= 8) {
            for (int i = 0; i < 13; i++) { const unsigned nb = field(b + 4 * i, 4); dat_str[id][i] = (char)(nb < 0xA ? 0x30 + nb : 0x41 + nb - 0xA); }      // -R (:360-365)
            dat_str[id][13] = 0;
            pck[id].ts = frmcnt;
            if (o.ecc) {
                pck[id].ec = ec;
                if (ec > 0) {
                    const int n = popcount15(ec);
                    pck[id].ec = n;
                    if ((o.dist || o.json) && n > 4) pck[id].ec = -2;
                }
            }
        }
        if (id == 0) {
            const int mode = (int)field(b + 16, 8);
            posmode = (mode > 1 && mode < 5) ? mode : -1;
            frnr = (int)field(b + 24, 8);
        }
        const auto s16v = [&](int off) { return (int)(short)field(b + off, 16); };
        const auto u16v = [&](int off) { return (int)(field(b + off, 16) & 0xFFFF); };
        const auto i32v = [&](int off) { return (int)field(b + off, 32); };
        if (posmode <= 2) {
            if (id == 1) {
                gps.prn = field(b, 32);
                gps.nPRN = 0; for (int j = 0; j < 32; j++) if ((gps.prn >> j) & 1) gps.nPRN += 1;
                sec = (float)((int)field(b + 32, 16) / 1000.0);
            }
            if (id == 2) { lat = i32v(0) / 1e7; horiV = s16v(32) / 1e2; }
            if (id == 3) { lon = i32v(0) / 1e7; dir = u16v(32) / 1e2; }
            if (id == 4) { alt = i32v(0) / 1e2; vertV = s16v(32) / 1e2; }
            if (id == 5) gps.dMSL = (float)((short)field(b, 16) / 1e2);
        } else {                                       // modes 3 and 4: time and horizontal speed in packet 0, position one packet earlier
            if (id == 0) { sec = (float)((int)field(b, 16) / 1000.0); horiV = s16v(32) / 1e2; }
            if (id == 1) { lat = i32v(0) / 1e7; dir = u16v(32) / 1e2; }
            if (id == 2) { lon = i32v(0) / 1e7; vertV = s16v(32) / 1e2; }
            if (posmode == 3) {
                if (id == 3) alt = i32v(0) / 1e2;
                if (id == 5) { lat2 = i32v(0) / 1e7; horiV2 = s16v(32) / 1e2; }
                if (id == 6) { lon2 = i32v(0) / 1e7; dir2 = u16v(32) / 1e2; }
                if (id == 7) { alt2 = i32v(0) / 1e2; vertV2 = s16v(32) / 1e2; }
            } else {                                   // mode 4: xdata bytes ride in packets 3..7
                if (id == 3) { alt = i32v(0) / 1e2; for (int j = 0; j < 2; j++) xdata[j] = (uint8_t)field(b + 32 + 8 * j, 8); }
                if (id > 3 && id < 8) for (int j = 0; j < 6; j++) xdata[2 + 6 * (id - 4) + j] = (uint8_t)field(b + 8 * j, 8);
            }
        }
        if (id == 8) {
            year = (int)field(b, 12); month = (int)field(b + 12, 4); day = (int)field(b + 16, 5);
            hour = (int)field(b + 21, 5); minute = (int)field(b + 26, 6);
            gps.nSV = (uint8_t)field(b + 32, 8);
        }
        return id;
    }

    // ---- configuration channels --------------------------------------------------------------------------------------------
    void conf_packet(const uint8_t *b, int ec) {
        const int ch = (int)field(b, 4);
        if (ch > 4 && field(b + 8, 20) == 0) snc.nul_ch = (uint8_t)field(b, 8);
        const bool dfm6 = ((snc.nul_ch & 0xF0) == 0x50) && (snc.nul_ch & 0x0F);
        if (dfm6) ptu_out = 6;
        if (dfm6 && (sonde_typ & 0xF) > 6) { sonde_typ = 0; snc.max_ch = (uint8_t)ch; reset_cfg(); }
        if (ch > 5 && ch > snc.max_ch && ec == 0 && field(b + 4, 4) == 0xC) snc.max_ch = (uint8_t)ch;

        if (ch > 5 && (ch == (snc.nul_ch >> 4) + 1 || ch == snc.max_ch)) {           // the serial number lives in this channel
            const int two = (int)field(b, 8), sn_ch = (two >> 4) & 0xF;
            if ((snc.nul_ch & 0x58) == 0x58) {                                         // DFM-06 family: 6 BCD-like nibbles, sent twice
                const uint32_t s6 = field(b + 4, 24);
                if (s6 == SN6 && s6 != 0) {
                    sonde_typ = SNBIT | sn_ch; ptu_out = 6;
                    snprintf(sonde_id, sizeof sonde_id, "IDx%1X:%6X", sn_ch & 0xF, SN6);
                    snprintf(SN_out, sizeof SN_out, "%6X", SN6);
                } else { sonde_typ = 0; reset_cfg(); }
                SN6 = s6;
            } else if ((two & 0xF) == 0xC || (two & 0xF) == 0x0) {                     // DFM-09 and later: two 16-bit halves, sent twice
                const int val = (int)field(b + 8, 20), hl = val & 0xF;
                if (hl < 2) {
                    if (snc.sn_ch != sn_ch) { snc.chXbit = 0; snc.chX[0] = snc.chX[1] = 0; reset_cfg(); }
                    snc.sn_ch = (uint8_t)sn_ch;
                    snc.chX[hl] = (uint32_t)(val >> 4) & 0xFFFF;
                    snc.chXbit |= (uint8_t)(1 << hl);
                    if (snc.chXbit == 3) {
                        const uint32_t sn = (snc.chX[0] << 16) | snc.chX[1];
                        if (sn == snc.SN_X || snc.SN_X == 0) {
                            sonde_typ = SNBIT | sn_ch; SN = sn;
                            ptu_out = (sn_ch >= 0xA && sn_ch <= 0xD) ? sn_ch : 0;
                            if (SN6 == 0 || (sonde_typ & 0xF) >= 0xA) {
                                snprintf(sonde_id, sizeof sonde_id, "IDx%1X:%6u", sonde_typ & 0xF, SN);
                                snprintf(SN_out, sizeof SN_out, "%6u", SN);
                            }
                        } else { sonde_typ = 0; reset_cfg(); }
                        snc.SN_X = sn;
                        snc.chXbit = 0;
                    }
                }
            }
        }
        const bool dfm17_A = (SN >= 23000000 && inv);       // newer serial numbers with the inverted Manchester convention

        if (ch >= 0 && ch <= 8 && ec == 0) {
            have24[ch] = 1;
            const int val = (int)field(b + 4, 24);
            val24[ch] = (uint32_t)val;
            meas24[ch] = float24(val);
            cfgchk = 0;
            if (ptu_out >= 0x5) cfgchk = have24[0] * have24[1] * have24[2] * have24[3] * have24[4] * have24[5];
            if (ptu_out >= 0x7) cfgchk *= have24[6] * have24[7];
            if (ptu_out >= 0x8) cfgchk *= have24[8];
        }
        sensortyp = 'T';
        Rf = 220e3f;
        if (cfgchk) {
            if (ptu_out >= 0xD || (ptu_out >= 0xC && meas24[6] < 220e3)) sensortyp = 'P';
            if (((ptu_out == 0xB || ptu_out == 0xC) && sensortyp == 'T') || ptu_out >= 0xD) Rf = 332e3f;
            if (ptu_out == 0xA && sensortyp == 'T' && dfm17_A) Rf = 332e3f;
            if (ptu_out == 6 && (sonde_typ & 0xF) == 8) sensortyp = 'P';
            if (ptu_out >= 0xA) {                            // STM32 generations: battery, MCU temperature, seconds counter
                const int ofs = sensortyp == 'P' ? 2 : 0;
                if (ch == 0x5 + ofs) status[0] = (float)((int)field(b + 8, 16) / 1000.0);
                if (ch == 0x6 + ofs) status[1] = (float)((int)field(b + 8, 16) / 100.0);
                if (ch == 0x7 + ofs && Rf > 300e3) status[2] = (float)((int)field(b + 8, 16) / 1.0);
            } else status[0] = status[1] = status[2] = 0;
        }
        dfmtyp = T_UNDEF;
        switch (sonde_typ & 0xF) {
            case 0x6: dfmtyp = T_06; break;
            case 0x7: case 0x8: dfmtyp = SN6 ? T_06P : T_PS15; break;
            case 0xA: dfmtyp = dfm17_A ? T_17 : T_09; break;
            case 0xB: dfmtyp = T_17; break;
            case 0xC: dfmtyp = sensortyp == 'P' ? T_09P : T_17; break;
            case 0xD: dfmtyp = T_17P; break;
            default: dfmtyp = T_UNKNOWN; break;
        }
    }

    float temperature() const {                              // NTC thermistor against the two reference resistors
        float Tk = 0;
        float f = meas24[0], f1 = meas24[3], f2 = meas24[4];
        if (sensortyp == 'P') { f = meas24[1]; f1 = meas24[5]; f2 = meas24[6]; }
        if (cfgchk) {
            const float B0 = 3260.0f, T0 = (float)(25 + 273.15), R0 = 5.0e3f;
            const float g = f2 / Rf;
            float R = (f - f1) / g;
            if (f * f1 * f2 == 0) R = 0;
            if (R > 0) Tk = (float)(1 / (1 / T0 + 1 / B0 * log(R / R0)));
        }
        return (float)(Tk - 273.15);
    }

    // --dbg: the two alternative thermistor evaluations (get_Temp2 :593-636, get_Temp4 :638-688); w: -vvv --ptu adds the resistor estimates
    bool alt_channels() const { return (ptu_out >= 0xC && meas24[6] < 220e3) || sonde_id[3] == '8'; }
    float temperature2(Out *w) const {
        float f = meas24[0], f1 = meas24[3], f2 = meas24[4];
        if (alt_channels()) { f = meas24[1]; f1 = meas24[5]; f2 = meas24[6]; }
        const float B0 = 3260.0, T0 = 25 + 273.15, R0 = 5.0e3, Rf2 = 220e3;
        const float g_o = f2 / Rf2, Rs_o = f1 / g_o;
        float Rf1 = Rs_o, g, Rb, R, Tk = 0;
        if (8e3 < Rs_o && Rs_o < 12e3) Rf1 = 10e3;
        else if (18e3 < Rs_o && Rs_o < 22e3) Rf1 = 20e3;
        g = (f2 - f1) / (Rf2 - Rf1);
        Rb = (f1 * Rf2 - f2 * Rf1) / (f2 - f1);
        R = (f - f1) / g;
        if (R > 0) Tk = (float)(1 / (1 / T0 + 1 / B0 * log(R / R0)));
        if (w) w->f("  (Rso: %.1f , Rb: %.1f)", Rs_o / 1e3, Rb / 1e3);
        return (float)(Tk - 273.15);
    }
    float temperature4() const {
        const float p0 = 1.09698417e-03, p1 = 2.39564629e-04, p2 = 2.48821437e-06, p3 = 5.84354921e-08;
        float f = meas24[0], f1 = meas24[3], f2 = meas24[4];
        if (alt_channels()) { f = meas24[1]; f1 = meas24[5]; f2 = meas24[6]; }
        const float Rf4 = 220e3, g = f2 / Rf4, R = (f - f1) / g;
        float Tk = 0;
        if (R > 0) Tk = (float)(1 / (p0 + p1 * log(R) + p2 * log(R) * log(R) + p3 * log(R) * log(R) * log(R)));
        return (float)(Tk - 273.15);
    }

    static void to_gps_week(int yy, int mm, int dd, int hr, int mi, int se, int *wk, int *tw) {
        if (mm < 3) { yy -= 1; mm += 12; }
        const int days = (int)(365.25 * yy) + (int)(30.6001 * (mm + 1.0)) + dd - 723263;
        *wk = days / 7;
        *tw = (days % 7) * 86400 + hr * 3600 + mi * 60 + se;
    }

    // ---- output, once per packet 8 -------------------------------------------------------------------------------------------
    void report(Out &w) {
        int output = frnr > 0 ? 0x1000 : 0;
        for (int i = 0; i < 9; i++)
            if (!((o.dist || o.json) && pck[i].ec < 0) && pck[8].ts - pck[i].ts < 6.0) output |= 1 << i;
        int jsonout = output;
        const bool contgps = (output & 0x11F) == 0x11F;
        const bool contaux = posmode == 4 && (output & 0xF8) == 0xF8;
        if (o.dist && !contgps) output = 0;
        if (o.json && !contgps) jsonout = 0;

        if (!o.raw || o.json) {
            to_gps_week(year, month, day, hour, minute, (int)(sec + 0.5), &week, &tow);
            sec_gps = (uint32_t)week * 604800u + (uint32_t)tow;      // the reference's int product wraps for garbage dates; the same bits without the overflow
            if (contgps) {
                int diff = (int)(uint8_t)sec_gps - frnr;
                if (diff < 0) diff += 256;
                if (o.json && (diff != prev_cntsec_diff || inv != prev_manpol)) { jsonout = 0; sonde_typ = 0; reset_cfg(); }
                prev_cntsec_diff = diff;
                prev_manpol = inv;
            }
            T = -273.15f;
            if (cfgchk && ptu_out) {
                T = temperature();
                if (T < -270.0f && dfmtyp != T_UNDEF && ((sonde_typ & 0xF) == 0x8 || (sonde_typ & 0xF) == 0xC)) dfmtyp = T_UNKNOWN;
            }
        }
        if (output & 0xF000) {
            if (o.raw == 2) {                                  // -R: the nine data packets as hex (:972-981)
                for (int i = 0; i < 9; i++) { w.f(" %s", dat_str[i]); if (o.ecc) w.f(" (%1X) ", pck[i].ec & 0xF); }
                for (int i = 0; i < 9; i++) for (int j = 0; j < 13; j++) dat_str[i][j] = ' ';
                w.f("\n");
            }
            if (!o.raw) {
                if (o.opt_auto && o.verbose >= 2) w.f("<%c> ", inv ? '-' : '+');
                w.f("[%3d] ", frnr);
                w.f("%4d-%02d-%02d ", year, month, day);
                w.f("%02d:%02d:%04.1f ", hour, minute, sec);
                const bool vv = o.verbose >= 2 && o.ecc;
                if (vv) w.f("(%1X,%1X,%1X) ", pck[0].ec & 0xF, pck[8].ec & 0xF, pck[1].ec & 0xF);
                w.f(" ");
                w.f(" lat: %.5f ", lat); if (vv) w.f("(%1X)  ", pck[2].ec & 0xF);
                w.f(" lon: %.5f ", lon); if (vv) w.f("(%1X)  ", pck[3].ec & 0xF);
                w.f(" alt: %.1f ", alt); if (vv) w.f("(%1X)  ", pck[4].ec & 0xF);
                w.f(" vH: %5.2f ", horiV);
                w.f(" D: %5.1f ", dir);
                w.f(" vV: %5.2f ", vertV);
                if (cfgchk) {
                    if (o.ptu && ptu_out) {
                        if (T > -270.0f) {
                            w.f("  T=%.1fC ", T);
                            if (o.verbose == 3) w.f(" (0x%X:%c%c) ", sonde_typ & 0xF, sensortyp, inv ? '-' : '+');
                        }
                        if (o.dbg) {
                            const float t2 = temperature2(o.verbose == 3 ? &w : nullptr), t4 = temperature4();
                            if (t2 > -270.0f) w.f("  T2=%.1fC ", t2);
                            if (t4 > -270.0f) w.f(" T4=%.1fC  ", t4);
                        }
                    }
                    if (o.verbose == 3 && ptu_out >= 0xA) {
                        if (status[0] > 0.0) w.f("  U: %.2fV ", status[0]);
                        if (status[1] > 0.0) w.f("  Ti: %.1fK ", status[1]);
                        if (status[2] > 0.0) w.f("  sec: %.0f ", status[2]);
                    }
                }
                if (o.dbg) {
                    for (int j = 0; j < 5; j++) w.f(" f%d:%.1f", j, meas24[j]);
                    if (ptu_out >= 0xA || sonde_id[3] == '8') { w.f(" f5:%.1f", meas24[5]); w.f(" f6:%.1f", meas24[6]); }
                    w.f(" ");
                }
                if (o.verbose && (sonde_typ & SNBIT)) {
                    w.f(" (%s", sonde_id);
                    if (o.verbose > 1 && *kTypes[dfmtyp]) w.f(":%s", kTypes[dfmtyp]);
                    w.f(") ");
                    sonde_typ ^= SNBIT;
                }
                w.f("\n");
                if (o.sat && posmode <= 2) {
                    w.f("  ");
                    w.f("  dMSL: %+.2f", gps.dMSL);
                    w.f("  sats: %d", gps.nSV);
                    w.f("  (");
                    for (int j = 0; j < 32; j++) if ((gps.prn >> j) & 1) w.f(" %02d", j + 1);
                    w.f("  nPRN: %d", gps.nPRN);
                    w.f(" )");
                    w.f("\n");
                }
            }
            if (o.json && jsonout && sec < 60.0) {
                char jid[] = "DFM-xxxxxxxx\0\0";
                const int xtyp = sonde_typ & 0xF;
                if (*SN_out) strncpy(jid + 4, SN_out, 9);
                int sats = gps.nSV;
                if (sats == 0) sats = gps.nPRN;
                w.f("{ \"type\": \"%s\"", "DFM");
                w.f(", \"frame\": %u, ", sec_gps);
                w.f("\"id\": \"%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, "
                    "\"heading\": %.5f, \"vel_v\": %.5f, \"sats\": %d", jid, year, month, day, hour, minute, sec, lat, lon, alt, horiV, dir, vertV, sats);
                if (ptu_out >= 0xA && status[0] > 0) w.f(", \"batt\": %.2f", status[0]);
                if (ptu_out && T > -270.0f) w.f(", \"temp\": %.1f", T);
                if (posmode == 4 && contaux && xdata[0]) {
                    w.f(", \"aux\": \"");
                    for (int j = 0; j < 26; j++) w.f("%02X", xdata[j]);
                    w.f("\"");
                }
                if (xtyp > 0) { w.f(", \"subtype\": \"0x%1X", xtyp); if (*kTypes[dfmtyp]) w.f(":%s", kTypes[dfmtyp]); w.f("\""); }
                if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
                w.f(", \"ref_datetime\": \"%s\"", "UTC");
                if (posmode <= 2) { w.f(", \"ref_position\": \"%s\"", "GPS"); w.f(", \"diff_GPS_MSL\": %.2f", -gps.dMSL); }
                else w.f(", \"ref_position\": \"%s\"", "MSL");
                if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
                w.f(" }\n");
                w.f("\n");
            }
        }
        for (int i = 0; i < 9; i++) pck[i].ec = -1;
    }
};

extern "C" {

int sonde_dfm_dec_create(const sonde_dfm_opts_t *opts, sonde_dfm_dec_t **out) {
    if (!opts || !out || opts->verbose < 0 || opts->verbose > 3 || opts->ecc < 0 || opts->ecc > 2) return SONDE_E_ARG;
    sonde_dfm_dec *d = new sonde_dfm_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.dist || d->o.json) d->o.ecc = 1;                // dfm09mod.c:1487
    for (int i = 0; i < 9; i++) { d->pck[i].ec = -1; d->pck[i].ts = 0.f; }
    *out = d;
    return 0;
}

void sonde_dfm_dec_destroy(sonde_dfm_dec_t *d) { delete d; }

int sonde_dfm_dec_frame(sonde_dfm_dec_t *d, const sonde_dfm_frame_t *f, char *out, size_t outlen) {
    if (!d || !f || !out || outlen < 1) return SONDE_E_ARG;
    uint8_t conf[28], dat[2][52];
    for (int i = 0; i < 7; i++) for (int j = 0; j < 4; j++) conf[4 * i + j] = (f->conf[i] >> (3 - j)) & 1;
    for (int i = 0; i < 13; i++) for (int j = 0; j < 4; j++) { dat[0][4 * i + j] = (f->dat1[i] >> (3 - j)) & 1; dat[1][4 * i + j] = (f->dat2[i] >> (3 - j)) & 1; }
    d->frmcnt = f->frm_count;
    d->inv = f->inv ? 1 : 0;
    Out w;
    {   // --rawecc (:1177-1196): the frame's bits as sliced, as hex nibbles (LSB first), if its 16 header bits are intact
        static const char kHdr[] = "0100010111001111";
        uint8_t fb[280];
        for (int i = 0; i < 280; i++) fb[i] = (f->rawbits[i >> 3] >> (i & 7)) & 1;
        if (f->frame_in_hit == 0) memcpy(fb, d->hdr_bits, 16); else memcpy(d->hdr_bits, fb, 16);
        if (d->o.raw == 9) {
            int diff = 0;
            for (int i = 0; i < 16; i++) diff += (fb[i] != (kHdr[i] & 1));
            if (diff == 0) {
                unsigned nb = 0;
                w.f("%c", d->inv ? '-' : '+'); w.f("<%7.1f>  ", d->frmcnt);
                for (int i = 16; i < 280; i++) {
                    if (i == 72 || i == 176) w.f(" ");
                    nb |= (unsigned)fb[i] << (i % 4);
                    if (i % 4 == 3) { w.f("%1X", nb & 0xF); nb = 0; }
                }
                w.f("\n");
            }
        }
    }
    if ((d->o.raw & 1) == 1 && !d->o.json) {                    // -r / --rawecc without --json: no telemetry tier (:1239)
        if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
        memcpy(out, w.s.c_str(), w.s.size() + 1);
        return (int)w.s.size();
    }
    const auto take = [&](int ret) { return !d->o.ecc || ret >= 0 || d->o.ecc == 2; };      // uncorrectable packets are skipped unless --ecc2
    if (take(f->ecc[0])) d->conf_packet(conf, f->ecc[0]);
    for (int k = 0; k < 2; k++)
        if (take(f->ecc[1 + k]) && d->data_packet(dat[k], f->ecc[1 + k]) == 8) d->report(w);
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

}  // extern "C"
