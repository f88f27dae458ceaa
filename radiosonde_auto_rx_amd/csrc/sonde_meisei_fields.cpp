// sonde_meisei_fields.cpp — Meisei iMS-100 / RS-11G frames -> the reference's text / JSON (include/sonde_meisei.h).  Host code, bit rate.
//
// One object = the gpx_t of demod/mod/meisei100mod.c plus the locals of its main() that live across frames (counter, sn, freq, the
// variant in effect, the pending reset).  Per header hit: biphase-S half symbols -> bits (:213-229), per subframe 6 BCH(63,51) blocks with
// the two odd-parity bits (:735-776), then the RS-11G (:779-1017), iMS-100 (:1018-1283) or raw (:1284-1310) printer; the two telemetry
// printers hand over to each other when the type word says so (the reference's goto jmpIMS / jmpRS11).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/sonde_hip.h"
#include "../../include/sonde_ecc.h"
#include "../../include/sonde_meisei.h"

namespace {

constexpr int HEADLEN = 24, BITFRAME_LEN = 1200, NSYM = SONDE_MEISEI_FRAME_SYMBOLS;
const char kHdrBits[] = "000001001001110111001110";                                     // 0x049DCE
const char kRawHeader[] = "101010101011010100101011001101001100101011001101";           // the same as biphase-S half symbols

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char b[640]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); s += b;
    }
};

uint32_t bits2val(const uint8_t *bits, int len) {
    uint32_t v = 0;
    for (int j = 0; j < len; j++) v |= (uint32_t)bits[j] << (len - 1 - j);
    return v;
}

float f32e2(uint32_t num) {                             // sign | 8-bit exponent | 23-bit mantissa with the sign behind the exponent (:163-191)
    uint32_t val = (num & 0x800000u) << 8;
    val |= (num >> 1) & 0x7F800000u;
    val |= num & 0x7FFFFFu;
    float f; memcpy(&f, &val, 4);
    f /= 4.0;
    return f;
}

int est_year_ims100(int y, int yr) {
    int rollover = 20, offset = 20;
    if (yr > 2003 && yr < 2100) { rollover = yr - 2004; offset = (rollover / 10) * 10; }
    y %= 10; y += offset;
    if (y < rollover) y += 10;
    return 2000 + y;
}

}  // namespace

struct sonde_meisei_dec {
    sonde_meisei_opts_t o{};
    sonde_ecc_t *bch = nullptr;
    // gpx_t
    int frnr = 0, frnr1 = 0, ref_yr = 2024, jahr = 0, monat = 0, tag = 0, std_ = 0, min_ = 0;
    float sek = 0.f;
    double lat = 0, lon = 0, alt = 0, vH = 0, vD = 0, vV = 0;
    uint16_t f_ref = 0;
    float T = NAN, RH = NAN;
    uint8_t frame_bits[BITFRAME_LEN + 10];
    float cfg[64];
    uint64_t cfg_valid = 0;
    uint32_t _sn = 0;
    float gsn = -1.f, fq = 0.f;
    int frm0_count = 0, frm0_valid = 0, frm1_count = 0, frm1_valid = 0, vV_valid = 0;
    // locals of main()
    int option_ims100 = 0, rst_gpx = 0, counter = 0;
    float sn = -1.f, freq = -1.f;
    uint8_t block_err[6] = { 0, 0, 0, 0, 0, 0 };
    // soft input framer
    float sbuf[48]; int bufpos = -1, in_frame = 0, pos = 0;
    float fsoft[NSYM];

    void reset_gpx() {
        for (int j = 0; j < 64; j++) cfg[j] = 0.0f;
        gsn = -1; frnr = frnr1 = 0; jahr = monat = tag = 0; std_ = min_ = 0; sek = 0.0f;
        lat = lon = alt = 0.0; vH = vD = vV = 0.0; vV_valid = 0; f_ref = 0; RH = NAN; T = NAN; cfg_valid = 0; _sn = 0; fq = 0.0f;
        frm0_count = 0; frm0_valid = 0; frm1_count = 0; frm1_valid = 0;
    }
    bool sane(int r0, int n) const {                     // resistances cfg[r0 ..] rising from > 0, temperatures cfg[17 ..] falling (:254-300)
        bool ok = true; float R_old = 0, T_old = INFINITY;
        for (int i = 0; i < n; i++) { if (cfg[r0 + i] <= R_old) ok = false; R_old = cfg[r0 + i]; }
        for (int i = 0; i < n; i++) { if (cfg[17 + i] >= T_old) ok = false; T_old = cfg[17 + i]; }
        return ok;
    }
    static int w16(const uint8_t *sf, int j) { return (int)bits2val(sf + HEADLEN + 46 * (j / 2) + 17 * (j % 2), 16); }

    void json_tail(Out &w, const char *subtype) {
        if (o.ptu) {
            if (!std::isnan(T)) w.f(", \"temp\": %.1f", T);
            if (!std::isnan(RH)) w.f(", \"humidity\": %.1f", RH);
        }
        w.f(", \"subtype\": \"%s\"", subtype);
        if (o.jsn_freq_khz > 0) w.f(", \"freq\": %d", o.jsn_freq_khz);
        if (fq > 0) w.f(", \"tx_frequency\": %.0f", fq);
        w.f(", \"ref_datetime\": \"%s\"", "UTC");
        w.f(", \"ref_position\": \"%s\"", "MSL");
        if (o.version[0]) w.f(", \"version\": \"%s\"", o.version);
        w.f(" }\n");
        w.f("\n");
    }
    void id_of(char *id) const { strcpy(id, "xxxxxx"); if (gsn > 0 && gsn < 1e9) sprintf(id, "%.0f", gsn); }

    // temperature from the thermistor frequency ratio: polynomial in 1/(f-1) -> resistance, log-linear interpolation in the sonde's table
    void thermistor(Out &w, const uint8_t *sf, int c0, int r0, int n, int &T_cfg) {
        const uint16_t t_raw = (uint16_t)bits2val(sf + HEADLEN + 2 * 46 + 17, 16);
        float f = ((float)t_raw / (float)f_ref) * 4.0f;
        if (f > 1.0f) {
            f = 1.0f / (f - 1.0f);
            const float R = cfg[c0] + cfg[c0 + 1] * f + cfg[c0 + 2] * f * f - cfg[c0 + 3];
            if (R <= cfg[r0]) T = cfg[17];
            else if (R >= cfg[r0 + n - 1]) T = cfg[17 + n - 1];
            else {
                for (int j = 0; j < n - 1; j++) {
                    if (R < cfg[r0 + 1 + j]) {
                        f = (logf(R) - logf(cfg[r0 + j])) / (logf(cfg[r0 + 1 + j]) - logf(cfg[r0 + j]));
                        T = cfg[17 + j] - f * (cfg[17 + j] - cfg[18 + j]);
                        break;
                    }
                }
            }
        }
        if (!std::isnan(T)) w.f("T=%.1fC ", T); else T_cfg = 0;
    }
    void humidity(Out &w, const uint8_t *sf) {
        const uint16_t u_raw = (uint16_t)bits2val(sf + HEADLEN + 3 * 46, 16);
        const float f = ((float)u_raw / (float)f_ref) * 4.0f;
        RH = cfg[49] + cfg[50] * f + cfg[51] * f * f + cfg[52] * f * f * f;
        RH = fmaxf(RH, 0.0f); RH = fminf(RH, 100.0f);
        w.f("RH=%.0f%% ", RH);
    }

    void frame(Out &w, const float *soft) {
        for (int j = 0; j < HEADLEN; j++) frame_bits[j] = (uint8_t)(kHdrBits[j] - 0x30);
        for (int j = 0; j < NSYM / 2; j++) frame_bits[HEADLEN + j] = ((soft[2 * j] >= 0.0f) == (soft[2 * j + 1] >= 0.0f)) ? 1 : 0;   // biphase-S
        frame_bits[HEADLEN + NSYM / 2] = 0;
        int gps_chk_sum = 0, gps_err = 0, err_frm = 0, err_blks = 0, header_found = 1;
        for (int subframe = 0; subframe < 2; subframe++) {
            uint8_t *sf = frame_bits + (subframe ? BITFRAME_LEN / 4 : 0);
            if (o.ecc) {
                for (int block = 0; block < 6; block++) {
                    uint8_t cw[64], ep[4], ev[4];
                    for (int j = 0; j < 46; j++) cw[45 - j] = sf[HEADLEN + block * 46 + j];
                    for (int j = 46; j < 63; j++) cw[j] = 0;
                    int errors = sonde_ecc_decode_bch_gf2t2(bch, cw, ep, ev);
                    if (errors >= 0) {
                        int chk = 0, par;
                        for (int j = 46; j < 63; j++) if (cw[j] != 0) chk = 0x1;
                        par = 1; for (int j = 13; j < 13 + 16; j++) par ^= cw[j];
                        if (cw[12] != par) chk |= 0x100;
                        par = 1; for (int j = 30; j < 30 + 16; j++) par ^= cw[j];
                        if (cw[29] != par) chk |= 0x10;
                        if (chk) errors = -3;
                    }
                    if (errors >= 0) for (int j = 0; j < 46; j++) sf[HEADLEN + block * 46 + j] = cw[45 - j];
                    if (errors < 0) { block_err[block] = errors == -3 ? 0xF : 0xE; err_frm += 1; }
                    else block_err[block] = (uint8_t)errors;
                    err_blks += (errors != 0);
                }
            }
            if (o.raw) {
                w.f("%06X ", bits2val(sf, HEADLEN) & 0xFFFFFF);
                for (int j = 0; j < 6; j++) {
                    w.f("%04X ", bits2val(sf + HEADLEN + 46 * j, 16) & 0xFFFF);
                    w.f("%04X ", bits2val(sf + HEADLEN + 46 * j + 17, 16) & 0xFFFF);
                }
                if (o.ecc && o.verbose) { w.f("#"); for (int b = 0; b < 6; b++) w.f("%X", block_err[b]); w.f("#  "); }
                if (subframe > 0) w.f("\n");
            } else {
                int ims = option_ims100;
                for (;;) {                                         // at most one hand-over: the two conditions exclude each other
                    if (rst_gpx) { reset_gpx(); sn = -1; freq = -1; rst_gpx = 0; }
                    if (!ims) { if (rs11g(w, sf, header_found, err_frm, err_blks)) { ims = 1; continue; } }
                    else      { if (ims100(w, sf, header_found, err_frm, gps_chk_sum, gps_err)) { ims = 0; continue; } }
                    break;
                }
            }
            header_found += 1;
        }
    }

    // RS-11G (:779-1017).  Returns 1 when the type word says iMS-100: the caller goes on in ims100().
    int rs11g(Out &w, const uint8_t *sf, int header_found, int err_frm, int err_blks) {
        if (header_found % 2 == 1) {
            counter = (int)(bits2val(sf + HEADLEN, 16) & 0xFFFF);
            w.f("[%d] ", counter);
            uint32_t val = bits2val(sf + HEADLEN + 46 * 3 + 17, 16);
            if ((val & 0xFF) >= 0xC0 && err_frm == 0) { option_ims100 = 1; w.f("\n"); rst_gpx = 1; return 1; }
            const uint16_t a = (uint16_t)bits2val(sf + HEADLEN + 46, 16), b = (uint16_t)bits2val(sf + HEADLEN + 46 + 17, 16);
            const uint32_t w32 = (uint32_t)((b & 0xFF00) >> 8 | (b & 0xFF) << 8) << 16 | (uint32_t)((a & 0xFF00) >> 8 | (a & 0xFF) << 8);
            const float fw32 = f32e2(w32);
            if (o.dbg) w.f(" # [%02d] %08x : %.1f # ", counter % 64, w32, fw32);
            if (err_blks == 0) {
                cfg[counter % 64] = fw32;
                cfg_valid |= 1uLL << (counter % 64);
                if (counter % 16 == 0) { sn = fw32; gsn = fw32; _sn = w32; }
                if (counter % 64 == 15) { freq = (float)(403700 + fw32 * 100.0); fq = freq; }
                if (counter % 4 == 0) f_ref = (uint16_t)bits2val(sf + HEADLEN + 17, 16);
                if (counter % 2 == 0 && o.ptu) {
                    T = NAN; RH = NAN;
                    if (f_ref != 0) {
                        int T_cfg = (cfg_valid & 0x0000FFFE0FFE0000ULL) == 0x0000FFFE0FFE0000ULL;
                        const int U_cfg = (cfg_valid & 0x001E000000000000ULL) == 0x001E000000000000ULL;
                        if (T_cfg && sane(37, 11)) thermistor(w, sf, 33, 37, 11, T_cfg);
                        if (U_cfg) humidity(w, sf);
                        if (T_cfg || U_cfg) w.f(" ");
                    }
                }
            }
            if (counter % 2 == 1) {
                const uint32_t t2 = bits2val(sf + HEADLEN + 5 * 46, 8), t1 = bits2val(sf + HEADLEN + 5 * 46 + 8, 8), ms = (t1 << 8) | t2;
                const uint32_t hh = bits2val(sf + HEADLEN + 5 * 46 + 17, 8), mi = bits2val(sf + HEADLEN + 5 * 46 + 25, 8);
                if (hh < 24 && mi < 60 && ms < 60000) { w.f("  "); w.f("%02d:%02d:%06.3f ", hh, mi, (double)ms / 1000.0); }
                w.f("\n");
                if (err_blks == 0) {
                    frnr1 = counter; std_ = (int)hh; min_ = (int)mi; sek = (float)((double)ms / 1000.0);
                    if (o.json && frnr1 - frnr == 1) {
                        char id[16]; id_of(id);
                        w.f("{ \"type\": \"%s\"", "MEISEI");
                        w.f(", \"frame\": %d, \"id\": \"RS11G-%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, \"heading\": %.5f, \"vel_v\": %.5f",
                            frnr, id, jahr, monat, tag, std_, min_, sek, lat, lon, alt, vH, vD, vV);
                        json_tail(w, "RS11G");
                    }
                }
            }
        }
        if (header_found % 2 == 0) {
            if (counter % 2 == 0) {
                const int lat1 = w16(sf, 1), lat2 = w16(sf, 2), lon1 = w16(sf, 3), lon2 = w16(sf, 4), alt1 = w16(sf, 5), alt2 = w16(sf, 6);
                const int la = (int)((uint32_t)lat1 << 16) | lat2, lo = (int)((uint32_t)lon1 << 16) | lon2, al = (int)((uint32_t)alt1 << 16) | alt2;
                w.f("  ");
                w.f("lat: %.5f  lon: %.5f  alt: %.2f", (double)la / 1e7, (double)lo / 1e7, (double)al / 1e2);
                w.f("  ");
                const uint16_t h = (uint16_t)w16(sf, 7), dd = (uint16_t)w16(sf, 8); const int16_t u = (int16_t)w16(sf, 9);
                const double velH = (double)h / 1e2, velD = (double)dd / 1e2, velU = (double)u / 1e2;
                w.f(" vH: %.2fm/s  D: %.1f  vV: %.2fm/s", velH, velD, velU);
                w.f("  ");
                const uint32_t jj = bits2val(sf + HEADLEN + 5 * 46 + 8, 8) + 0x0700, mm = bits2val(sf + HEADLEN + 5 * 46 + 17, 8), tt = bits2val(sf + HEADLEN + 5 * 46 + 25, 8);
                if (jj > 1980 && mm > 0 && mm < 13 && tt > 0 && tt < 32) w.f(" %4d-%02d-%02d ", jj, mm, tt);
                if (err_blks == 0) {
                    frnr = counter; tag = (int)tt; monat = (int)mm; jahr = (int)jj;
                    lat = (double)la / 1e7; lon = (double)lo / 1e7; alt = (double)al / 1e2; vH = velH; vD = velD; vV = velU;
                }
                if (o.verbose && err_blks == 0) {
                    if (sn > 0) { w.f(" : sn %.0f (0x%08x)", sn, _sn); sn = -1; }
                    if (freq > 0) { w.f(" : fq %.0f", freq); freq = -1; }
                }
                w.f("\n");
            }
        }
        return 0;
    }

    // iMS-100 (:1018-1283).  Returns 1 when the type word says RS-11G.
    int ims100(Out &w, const uint8_t *sf, int header_found, int err_frm, int &gps_chk_sum, int &gps_err) {
        if (header_found % 2 == 1) {
            for (int j = 10; j < 12; j++) gps_chk_sum += w16(sf, j);
            uint32_t val = bits2val(sf + HEADLEN + 46 * 3 + 17, 16);
            if ((val & 0xFF) < 0xC0 && err_frm == 0) { option_ims100 = 0; w.f("\n"); rst_gpx = 1; return 1; }
            counter = (int)(bits2val(sf + HEADLEN, 16) & 0xFFFF);
            w.f("[%d] ", counter);
            const uint16_t a = (uint16_t)bits2val(sf + HEADLEN + 46, 16), b = (uint16_t)bits2val(sf + HEADLEN + 46 + 17, 16);
            const uint32_t w32 = ((uint32_t)b << 16) | a;
            float fcfg; memcpy(&fcfg, &w32, 4);
            if (o.dbg) w.f(" # [%02d] %08x : %.1f # ", counter % 64, w32, fcfg);
            if (err_frm == 0 && block_err[0] < 2 && block_err[1] < 2) {
                cfg[counter % 64] = fcfg;
                cfg_valid |= 1uLL << (counter % 64);
                if (counter % 0x10 == 0) { sn = fcfg; gsn = sn; _sn = w32; }
                if (counter % 64 == 15) { freq = (float)(400e3 + fcfg * 100.0); fq = freq; }
                if (counter % 4 == 0) f_ref = (uint16_t)bits2val(sf + HEADLEN + 17, 16);
                if (counter % 4 == 3) f_ref = (uint16_t)bits2val(sf + HEADLEN + 3 * 46, 16);
            }
            if (counter % 2 == 0) {
                frnr = counter;
                const uint32_t t1 = bits2val(sf + HEADLEN + 5 * 46, 8), t2 = bits2val(sf + HEADLEN + 5 * 46 + 8, 8), ms = (t1 << 8) | t2;
                sek = (float)((float)ms / 1000.0);
                std_ = (int)bits2val(sf + HEADLEN + 5 * 46 + 17, 8); min_ = (int)bits2val(sf + HEADLEN + 5 * 46 + 25, 8);
                w.f("  "); w.f("%02d:%02d:%06.3f ", std_, min_, sek); w.f("  ");
                if (o.ptu) {
                    T = NAN; RH = NAN;
                    if (f_ref != 0) {
                        int T_cfg = (cfg_valid & 0x01E01FFE1FFE0000ULL) == 0x01E01FFE1FFE0000ULL;
                        const int U_cfg = (cfg_valid & 0x001E000000000000ULL) == 0x001E000000000000ULL;
                        if (T_cfg && sane(33, 12)) thermistor(w, sf, 53, 33, 12, T_cfg);
                        if (U_cfg) humidity(w, sf);
                        if (T_cfg || U_cfg) w.f(" ");
                    }
                }
            }
        }
        if (header_found % 2 == 0) {
            for (int j = 0; j < 11; j++) gps_chk_sum += w16(sf, j);
            gps_err = (gps_chk_sum & 0xFFFF) != w16(sf, 11);
            if (counter % 2 == 0) {
                const uint32_t dat2 = bits2val(sf + HEADLEN, 16);
                tag = (int)(dat2 / 1000); monat = (int)((dat2 / 10) % 100);
                jahr = est_year_ims100((int)(dat2 % 10), ref_yr);
                w.f("(%04d-%02d-%02d) ", jahr, monat, tag);
                const int lat1 = w16(sf, 1), lat2 = w16(sf, 2), lon1 = w16(sf, 3), lon2 = w16(sf, 4), alt1 = w16(sf, 5);
                const int alt2 = (int)bits2val(sf + HEADLEN + 46 * 3, 8);
                const int la = (int)((uint32_t)lat1 << 16) | lat2, lo = (int)((uint32_t)lon1 << 16) | lon2, al = (alt1 << 8) | alt2;
                const int latdeg = (int)((int)la / 1e6); const double latmin = (double)(la / 1e6 - latdeg) * 100 / 60.0;
                const int londeg = (int)((int)lo / 1e6); const double lonmin = (double)(lo / 1e6 - londeg) * 100 / 60.0;
                lat = (double)latdeg + latmin; lon = (double)londeg + lonmin; alt = (double)al / 1e2;
                w.f("  "); w.f("lat: %.5f  lon: %.5f  alt: %.2f", lat, lon, alt); w.f("  ");
                const uint16_t dd = (uint16_t)w16(sf, 9), h = (uint16_t)w16(sf, 10);
                vD = (double)dd / 1e2; vH = (double)h / 1.94384e2;
                w.f(" (vH: %.1fm/s  D: %.2f)", vH, vD); w.f("  ");
            }
            if (counter % 2 == 1) {
                const int16_t u = (int16_t)w16(sf, 1);
                vV = (double)u / 1.94384e1; vV_valid = (u != 0);
                if (vV_valid) w.f("  (vV: %.1fm/s)", vV); else w.f("  (vV: --- m/s)");
                w.f("  ");
            }
            if (counter % 2 == 0) {
                frm0_count = counter;
                if (o.ecc) { w.f(gps_err ? "(no)" : "(ok)"); w.f(err_frm ? "[NO]" : "[OK]"); frm0_valid = (err_frm == 0 && gps_err == 0); }
                if (o.verbose && sn > 0) { w.f(" : sn %.0f", sn); sn = -1; }
                w.f("\n");
            }
            if (counter % 2 == 1) {
                frm1_count = counter;
                if (o.ecc) { w.f(gps_err ? "(no)" : "(ok)"); w.f(err_frm ? "[NO]" : "[OK]"); frm1_valid = (err_frm == 0 && gps_err == 0); }
                if (o.verbose && freq > 0) { w.f(" : fq %.0f", freq); freq = -1; }
                w.f("\n");
                if (o.json && frm0_valid) {
                    char id[16]; id_of(id);
                    w.f("{ \"type\": \"%s\"", "MEISEI");
                    w.f(", \"frame\": %d, \"id\": \"IMS100-%s\", \"datetime\": \"%04d-%02d-%02dT%02d:%02d:%06.3fZ\", \"lat\": %.5f, \"lon\": %.5f, \"alt\": %.5f, \"vel_h\": %.5f, \"heading\": %.5f",
                        frnr, id, jahr, monat, tag, std_, min_, sek, lat, lon, alt, vH, vD);
                    if (frm1_valid && frm1_count == frm0_count + 1 && vV_valid) w.f(", \"vel_v\": %.5f", vV);
                    json_tail(w, "IMS100");
                    frm0_valid = 0;
                }
            }
        }
        return 0;
    }
};

extern "C" {

int sonde_meisei_dec_create(const sonde_meisei_opts_t *opts, sonde_meisei_dec_t **out) {
    if (!opts || !out) return SONDE_E_ARG;
    sonde_meisei_dec *d = new sonde_meisei_dec();
    d->o = *opts;
    d->o.version[sizeof d->o.version - 1] = 0;
    if (d->o.json) d->o.ecc = 1;
    d->bch = sonde_ecc_create(SONDE_ECC_BCH64);
    d->option_ims100 = opts->ims100 != 0;
    d->ref_yr = (opts->ref_year > 2003 && opts->ref_year < 2100) ? opts->ref_year : 2024;
    memset(d->frame_bits, 0, sizeof d->frame_bits);
    memset(d->cfg, 0, sizeof d->cfg);
    memset(d->sbuf, 0, sizeof d->sbuf);
    *out = d;
    return 0;
}

void sonde_meisei_dec_destroy(sonde_meisei_dec_t *d) { if (d) { sonde_ecc_destroy(d->bch); delete d; } }

static int finish_out(const Out &w, char *out, size_t outlen) {
    if (w.s.size() + 1 > outlen) return SONDE_E_ARG;
    memcpy(out, w.s.c_str(), w.s.size() + 1);
    return (int)w.s.size();
}

int sonde_meisei_dec_frame(sonde_meisei_dec_t *d, const float *soft, int32_t n, char *out, size_t outlen) {
    if (!d || !out || n < 0 || n > NSYM || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    if (n == NSYM) d->frame(w, soft);
    return finish_out(w, out, outlen);
}

int sonde_meisei_dec_push_soft(sonde_meisei_dec_t *d, const float *soft, int32_t n, int32_t invert, int32_t finish, char *out, size_t outlen) {
    if (!d || !out || n < 0 || (n > 0 && !soft)) return SONDE_E_ARG;
    Out w;
    for (int i = 0; i < n; i++) {
        const float s = invert ? -soft[i] : soft[i];
        if (!d->in_frame) {                                      // find_softbinhead / corr_softhdb (demod_mod.c:1692-1762)
            d->bufpos = (d->bufpos + 1) % 48;
            d->sbuf[d->bufpos] = s;
            double sum = 0.0, nx = 0.0, ny = 0.0;
            int j = d->bufpos + 1;
            for (int k = 0; k < 48; k++) {
                if (j >= 48) j = 0;
                const float x = d->sbuf[j], y = (float)(2.0 * (kRawHeader[k] & 1) - 1.0);
                sum += y * d->sbuf[j]; nx += x * x; ny += y * y;          // float products, double sums
                j++;
            }
            sum /= sqrt(nx * ny);
            if (fabs((float)sum) > 0.8f) { d->in_frame = 1; d->pos = 0; }
        } else {
            d->fsoft[d->pos++] = s;
            if (d->pos >= NSYM) { d->frame(w, d->fsoft); d->in_frame = 0; }
        }
    }
    if (finish) w.f("\n");
    return finish_out(w, out, outlen);
}

}  // extern "C"
