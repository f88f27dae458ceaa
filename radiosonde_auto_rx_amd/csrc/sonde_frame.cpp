// sonde_frame.cpp — RS41 frame-level host code of the engine: GF(2^8) tables, RS(255,231) Euclid decoder,
// CRC-16, the two-codeword ECC passes and the raw text line.
//
// Reference behaviour reproduced (not its code): bch_ecc_mod.c (GF_genTab :136, syndromes :638,
// polyGF_lfsr :547, rs_decode_ErrEra :877, forney :596, rs_encode :860) and rs41mod.c
// (crc16 :284, frametype :407, rs41_ecc :1703-1769/:1955-1974, print_frame raw line :2530-2545).
// Decoder failure semantics are kept identical (same key-equation solver and acceptance tests) so that
// [NO] frames and miscorrections come out byte-for-byte like the reference.
#include "sonde_host.h"
#include "../../include/sonde_hip.h"
#include <cstdio>
#include <cstring>

namespace sonde {

static uint8_t g_exp[512], g_log[256];
static bool g_ready = false;
static void gf_init() {
    if (g_ready) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { g_exp[i] = (uint8_t)x; g_log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11D; }
    for (int i = 255; i < 512; i++) g_exp[i] = g_exp[i - 255];
    g_log[0] = 0;
    g_ready = true;
}
const uint8_t *gf_exp_table() { gf_init(); return g_exp; }
const uint8_t *gf_log_table() { gf_init(); return g_log; }
static inline uint8_t mul(uint8_t a, uint8_t b) { return (a && b) ? g_exp[g_log[a] + g_log[b]] : 0; }
static inline uint8_t inv(uint8_t a) { return a ? g_exp[255 - g_log[a]] : 0; }

// small dense polynomials, degree < PD
enum { PD = 64, T = 12, R = 24 };
struct Poly { uint8_t c[PD]; Poly() { memset(c, 0, PD); } int deg() const { int n = PD - 1; while (n >= 0 && !c[n]) n--; return n; } };
static uint8_t eval(const Poly &p, uint8_t x) { uint8_t y = 0; for (int n = p.deg(); n >= 0; n--) y = mul(y, x) ^ p.c[n]; return y; }
static Poly pmul(const Poly &a, const Poly &b) {
    Poly r; const int da = a.deg(), db = b.deg();
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) r.c[i + j] ^= mul(a.c[i], b.c[j]);
    return r;
}
static void divmod(const Poly &p, const Poly &q, Poly &d, Poly &r) {
    d = Poly(); r = Poly();
    int dp = p.deg(); const int dq = q.deg();
    if (dq < 0) return;
    if (dq == 0) { const uint8_t c = inv(q.c[0]); for (int i = 0; i <= dp; i++) d.c[i] = mul(p.c[i], c); return; }
    r = p;
    if (dp < dq) return;
    const uint8_t qi = inv(q.c[dq]);
    while (dp >= dq) {
        const uint8_t c = mul(r.c[dp], qi);
        d.c[dp - dq] = c;
        for (int i = 0; i <= dq; i++) r.c[dp - i] ^= mul(q.c[dq - i], c);
        dp = r.deg();
    }
}

int rs255_syndromes(const uint8_t cw[255], uint8_t S[24]) {
    gf_init();
    int any = 0;
    for (int j = 0; j < R; j++) {
        const uint8_t x = g_exp[j];
        uint8_t y = 0;
        for (int n = 254; n >= 0; n--) y = mul(y, x) ^ cw[n];
        S[j] = y; any |= (y != 0);
    }
    return any;
}

int rs255_decode_syn(uint8_t cw[255], const uint8_t S[24]) {
    gf_init();
    int any = 0;
    Poly r0, r1, s0, s1;
    for (int j = 0; j < R; j++) { r0.c[j] = S[j]; any |= S[j]; }
    if (!any) return 0;
    r1.c[R] = 1; s0.c[0] = 1;
    while (r1.deg() >= T) {                        // Euclid on (S, x^2t) until deg(remainder) < t
        Poly quo, rem; divmod(r0, r1, quo, rem);
        r0 = r1; r1 = rem;
        Poly s2 = pmul(quo, s1);
        for (int i = 0; i < PD; i++) s2.c[i] ^= s0.c[i];
        s0 = s1; s1 = s2;
    }
    Poly &Om = r1, &La = s1;
    const int dL = La.deg(), dO = Om.deg();
    if (dO >= dL) return -3;
    if (La.c[0] == 0) return -2;
    const uint8_t gi = inv(La.c[0]);
    for (int i = 0; i <= dL; i++) La.c[i] = mul(La.c[i], gi);
    for (int i = 0; i <= dO; i++) Om.c[i] = mul(Om.c[i], gi);
    Poly dLa;
    for (int i = 1; i <= dL; i += 2) dLa.c[i - 1] = La.c[i];
    uint8_t pos[R], val[R]; int nerr = 0;
    for (int x = 1; x < 256 && nerr < dL; x++) {
        if (eval(La, (uint8_t)x)) continue;
        const uint8_t z = eval(dLa, (uint8_t)x);
        val[nerr] = z ? mul(mul(eval(Om, (uint8_t)x), inv(z)), inv((uint8_t)x)) : 0;   // Forney, b = 0
        pos[nerr] = g_log[inv((uint8_t)x)];
        nerr++;
    }
    if (nerr < dL) return -1;
    for (int i = 0; i < nerr; i++) cw[pos[i]] ^= val[i];
    return nerr;
}

int rs255_decode(uint8_t cw[255]) { uint8_t S[24]; rs255_syndromes(cw, S); return rs255_decode_syn(cw, S); }

void rs255_encode(uint8_t cw[255]) {
    gf_init();
    static uint8_t gen[R + 1]; static bool have = false;
    if (!have) {
        memset(gen, 0, sizeof gen); gen[0] = 1;
        for (int i = 0; i < R; i++) {               // g(X) *= (X - alpha^i)
            for (int j = i + 1; j >= 1; j--) gen[j] = gen[j - 1] ^ mul(gen[j], g_exp[i]);
            gen[0] = mul(gen[0], g_exp[i]);
        }
        have = true;
    }
    uint8_t rem[255]; memset(rem, 0, R); memcpy(rem + R, cw + R, 255 - R);
    for (int d = 254; d >= R; d--) {
        const uint8_t c = rem[d];
        if (c) for (int j = 0; j <= R; j++) rem[d - R + j] ^= mul(gen[j], c);
    }
    memcpy(cw, rem, R);
}

int crc16(const uint8_t *p, int len) {
    int rem = 0xFFFF;
    for (int i = 0; i < len; i++) {
        rem ^= p[i] << 8;
        for (int j = 0; j < 8; j++) rem = (rem & 0x8000) ? ((rem << 1) ^ 0x1021) & 0xFFFF : (rem << 1) & 0xFFFF;
    }
    return rem;
}

const char kRs41Header[65] = "0000100001101101010100111000100001000100011010010100100000011111";
const uint8_t kRs41HeaderBytes[8] = { 0x86, 0x35, 0xf4, 0x40, 0x93, 0xdf, 0x1a, 0x60 };
const uint8_t kRs41Mask[64] = {
    0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
    0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
    0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
    0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1 };

int rs41_frametype(const uint8_t *f) {
    int ft = 0; const uint8_t b = f[0x38];
    for (int i = 0; i < 4; i++) ft += ((b >> i) & 1) - ((b >> (i + 4)) & 1);
    return ft;
}

static void gather(const uint8_t *frame, uint8_t *cw1, uint8_t *cw2, bool parity) {
    if (parity) for (int i = 0; i < 24; i++) { cw1[i] = frame[8 + i]; cw2[i] = frame[32 + i]; }
    for (int i = 0; i < 231; i++) { cw1[24 + i] = frame[56 + 2 * i]; cw2[24 + i] = frame[57 + 2 * i]; }
}

int rs41_ecc(uint8_t frame[518], int frmlen, int level, const uint8_t *synd) {
    uint8_t cw1[255], cw2[255];
    if (frmlen > 518) frmlen = 518;
    for (int i = frmlen; i < 518; i++) frame[i] = 0;
    gather(frame, cw1, cw2, true);
    int e1, e2;
    if (synd) { e1 = rs255_decode_syn(cw1, synd); e2 = rs255_decode_syn(cw2, synd + 24); }
    else      { e1 = rs255_decode(cw1);           e2 = rs255_decode(cw2); }
    if (level >= 2 && (e1 < 0 || e2 < 0)) {          // 2nd pass: known block ids, zero tail
        static const int pos[5] = { 0x039, 0x065, 0x093, 0x0B5, 0x112 };
        static const int pck[5] = { 0x7928, 0x7A2A, 0x7C1E, 0x7D59, 0x7B15 };
        for (int k = 0; k < 5; k++) { frame[pos[k]] = (uint8_t)(pck[k] >> 8); frame[pos[k] + 1] = (uint8_t)(pck[k] & 0xFF); }
        if (rs41_frametype(frame) < -2) {
            for (int i = 320 + 7; i < 518 - 2; i++) frame[i] = 0;
        } else {
            for (int i = 320; i < 518; i++) frame[i] = 0;
            frame[0x12B] = 0x76; frame[0x12C] = 0x11;
            for (int i = 0x12D; i < 318; i++) frame[i] = 0;
            frame[318] = 0xEC; frame[319] = 0xC7;
        }
        gather(frame, cw1, cw2, false);
        e1 = rs255_decode(cw1); e2 = rs255_decode(cw2);
    }
    for (int i = 0; i < 24; i++) { frame[8 + i] = cw1[i]; frame[32 + i] = cw2[i]; }
    for (int i = 0; i < 231; i++) { frame[56 + 2 * i] = cw1[24 + i]; frame[57 + 2 * i] = cw2[24 + i]; }
    if (e1 < 0 || e2 < 0) return -((e1 < 0 ? 1 : 0) | (e2 < 0 ? 2 : 0));
    return e1 + e2;
}


// ---- DFM: Hamming(8,4), systematic generator / parity check of dfm09mod.c:181-195; de-interleave :231
const char kDfmRawHeader[33] = "10011010100110010101101001010101";
const char kM10RawHeader[33] = "10011001100110010100110010011001";       // m10mod.c:76

// M10 frame checksum: a 16-bit register stepped once per byte; the low byte mixes the rotated-and-folded input byte with two
// parity folds of the old register, the old low byte moves up (m10mod.c:594-628).
int m10_checksum(const uint8_t *msg, int len) {
    int c = 0;
    for (int i = 0; i < len; i++) {
        uint8_t b = msg[i];
        b = (uint8_t)((b >> 1) | ((b & 1) << 7));
        b ^= (b >> 2) & 0xFF;
        const int t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1), t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1);
        const int t = (c & 0x3F) | (t6 << 6) | (t7 << 7);
        int sreg = (c >> 7) & 0xFF;
        sreg ^= (sreg >> 2) & 0xFF;
        c = (((c & 0xFF) << 8) | ((b ^ t ^ sreg) & 0xFF)) & 0xFFFF;
    }
    return c & 0xFFFF;
}

static void dfm_codeword(int n, uint8_t *c) {
    const uint8_t d[4] = { (uint8_t)((n >> 3) & 1), (uint8_t)((n >> 2) & 1), (uint8_t)((n >> 1) & 1), (uint8_t)(n & 1) };
    c[0] = d[0]; c[1] = d[1]; c[2] = d[2]; c[3] = d[3];
    c[4] = d[1] ^ d[2] ^ d[3]; c[5] = d[0] ^ d[2] ^ d[3]; c[6] = d[0] ^ d[1] ^ d[3]; c[7] = d[0] ^ d[1] ^ d[2];
}

// one 8-bit codeword: returns 0 clean, j+1 = bit j fixed, -1 uncorrectable (then ecc level 2 picks, among the
// codewords at distance 2, the one best correlated with the soft bits — dfm09mod.c:262-307)
static int dfm_check(int level, uint8_t hb[8], const float sb[8]) {
    static const uint8_t Hm[4][8] = { {0,1,1,1,1,0,0,0}, {1,0,1,1,0,1,0,0}, {1,1,0,1,0,0,1,0}, {1,1,1,0,0,0,0,1} };
    static const uint8_t He[8] = { 0x7, 0xB, 0xD, 0xE, 0x8, 0x4, 0x2, 0x1 };
    unsigned syn = 0;
    for (int i = 0; i < 4; i++) { uint8_t s = 0; for (int j = 0; j < 8; j++) s ^= Hm[i][j] & hb[j]; syn = (syn << 1) | s; }
    if (!syn) return 0;
    for (int j = 0; j < 8; j++) if (syn == He[j]) { hb[j] ^= 1; return j + 1; }
    if (level == 2) {
        int best = -1; float bestsum = 0.0f;
        for (int n = 0; n < 16; n++) {
            uint8_t c[8]; int d = 0;
            dfm_codeword(n, c);
            for (int i = 0; i < 8; i++) d += (hb[i] != c[i]);
            if (d != 2) continue;
            float sum = 0.0f;
            for (int i = 0; i < 8; i++) sum += (2 * c[i] - 1) * sb[i];
            if (sum >= bestsum) { bestsum = sum; best = n; }
        }
        if (best >= 0) dfm_codeword(best, hb);
    }
    return -1;
}

// str: L*8 interleaved bits of one block (hard + soft) -> L nibbles; return value as hamming() of the reference
int dfm_block(int level, const uint8_t *hb, const float *sb, int L, uint8_t *nib) {
    int ret = 0;
    for (int i = 0; i < L; i++) {
        uint8_t c[8]; float s[8];
        for (int j = 0; j < 8; j++) { c[j] = hb[L * j + i]; s[j] = sb[L * j + i]; }
        if (level) {
            const int e = dfm_check(level, c, s);
            if (e > 0) ret |= (1 << i);
            if (e < 0) ret |= e;
        }
        nib[i] = (uint8_t)((c[0] << 3) | (c[1] << 2) | (c[2] << 1) | c[3]);
    }
    return ret;
}

}  // namespace sonde

extern "C" {
int sonde_dfm_rawline(const sonde_dfm_frame_t *f, int ecc_level, char *buf, size_t buflen) {
    if (!f || !buf || buflen < 96) return SONDE_E_ARG;
    const uint8_t *blk[3] = { f->conf, f->dat1, f->dat2 };
    const int len[3] = { 7, 13, 13 };
    int n = 0;
    for (int b = 0; b < 3; b++) {
        if (b) n += snprintf(buf + n, buflen - n, "  ");
        for (int i = 0; i < len[b]; i++) n += snprintf(buf + n, buflen - n, "%01X", blk[b][i]);
        if (ecc_level) n += snprintf(buf + n, buflen - n, f->ecc[b] == 0 ? " [OK] " : f->ecc[b] > 0 ? " [KO] " : " [NO] ");
    }
    return n;
}
int sonde_m10_rawline(const sonde_m10_frame_t *f, int verbose, char *buf, size_t buflen) {
    bool col = (verbose & SONDE_M10_COLOR) != 0;
    verbose &= 0xFF;
    if (!f || !buf || f->len < 101 || f->len > 121 || buflen < (size_t)((col ? 26 : 2) * f->len + 96)) return SONDE_E_ARG;
    const int typ = f->frame[1];
    if (typ == 0x49 || !(typ == 0x9F || typ == 0xAF || !(typ == 0x8F || typ == 0x20))) col = false;      // coloured for t_M10 / t_M10plus (unknown types count as M10, :1066-1072)
    int n = 0;
    if (!col) {
        for (int i = 0; i < f->len; i++) n += snprintf(buf + n, buflen - n, "%02x", f->frame[i]);
        if (verbose) n += snprintf(buf + n, buflen - n, " # %04x%s", f->cs_calc, f->cs_ok ? " [OK]" : " [NO]");
        return n;
    }
    const char *FR = "\x1b[38;5;244m";
    const bool m10 = typ != 0xAF;
    const int aux = f->len - 101, pc = 0x63 + aux;
    n += snprintf(buf + n, buflen - n, "%s", FR);
    for (int i = 0; i < f->len; i++) {
        const auto put = [&](const char *c) { n += snprintf(buf + n, buflen - n, "%s", c); };
        if (i == 1) put("\x1b[38;5;250m");
        if (m10) {
            if (i >= 0x0A && i < 0x0A + 4) put("\x1b[38;5;27m");
            if (i >= 0x0E && i < 0x0E + 4) put("\x1b[38;5;34m");
            if (i >= 0x12 && i < 0x12 + 4) put("\x1b[38;5;70m");
            if (i >= 0x16 && i < 0x16 + 4) put("\x1b[38;5;82m");
            if (i >= 0x20 && i < 0x20 + 2) put("\x1b[38;5;20m");
            if (i >= 0x04 && i < 0x04 + 6) put("\x1b[38;5;36m");
        } else {
            if (i >= 0x04 && i < 0x04 + 4) put("\x1b[38;5;34m");
            if (i >= 0x08 && i < 0x08 + 4) put("\x1b[38;5;70m");
            if (i >= 0x0C && i < 0x0C + 3) put("\x1b[38;5;82m");
            if (i >= 0x0F && i < 0x0F + 6) put("\x1b[38;5;36m");
            if (i >= 0x15 && i < 0x15 + 3) put("\x1b[38;5;27m");
            if (i >= 0x18 && i < 0x18 + 3) put("\x1b[38;5;20m");
        }
        if (i >= 0x5D && i < 0x5D + 5) put("\x1b[38;5;58m");
        if (i == 0x62) put("\x1b[38;5;172m");
        if (i >= pc && i < pc + 2) put("\x1b[38;5;11m");
        n += snprintf(buf + n, buflen - n, "%02x", f->frame[i]);
        put(FR);
    }
    if (verbose) {
        n += snprintf(buf + n, buflen - n, " # %s%04x%s", "\x1b[38;5;11m", f->cs_calc, FR);
        n += snprintf(buf + n, buflen - n, " %s%s%s", f->cs_ok ? "\x1b[38;5;2m" : "\x1b[38;5;1m", f->cs_ok ? "[OK]" : "[NO]", FR);
    }
    n += snprintf(buf + n, buflen - n, "%s", "\x1b[0m");
    return n;
}
int sonde_m20_rawline(const sonde_m20_frame_t *f, int verbose, char *buf, size_t buflen) {
    const bool col = (verbose & SONDE_M20_COLOR) != 0;
    verbose &= 0xFF;
    if (!f || !buf || f->len < 1 || f->len > 0x45 + 64 + 1 || buflen < (size_t)((col ? 26 : 2) * f->len + 96)) return SONDE_E_ARG;
    int n = 0;
    if (!col) {
        for (int i = 0; i < f->len; i++) n += snprintf(buf + n, buflen - n, "%02x", f->frame[i]);
        if (verbose) {
            n += snprintf(buf + n, buflen - n, " # %04x", f->cs_calc);
            if (f->fw < 0x07) n += snprintf(buf + n, buflen - n, f->blk_ok > 0 ? " (ok)" : f->blk_ok < 0 ? " (oo)" : " (no)");
            n += snprintf(buf + n, buflen - n, f->cs_ok ? " [OK]" : " [NO]");
        }
        return n;
    }
    // -c: a colour in front of every byte of a field, the text colour behind every byte (m20mod.c:918-958)
    const char *FR = "\x1b[38;5;244m", *TXT = FR;
    const int pc = f->len - 2;                                       // pos_check = flen - 1, len = flen + 1
    n += snprintf(buf + n, buflen - n, "%s", FR);
    for (int i = 0; i < f->len; i++) {
        const auto put = [&](const char *c) { n += snprintf(buf + n, buflen - n, "%s", c); };
        if (i == 1) put("\x1b[38;5;250m");
        if (i >= 0x0F && i < 0x0F + 3) put("\x1b[38;5;27m");
        if (i >= 0x1C && i < 0x1C + 4) put("\x1b[38;5;34m");
        if (i >= 0x20 && i < 0x20 + 4) put("\x1b[38;5;70m");
        if (i >= 0x08 && i < 0x08 + 3) put("\x1b[38;5;82m");
        if (i >= 0x1A && i < 0x1A + 2) put("\x1b[38;5;20m");
        if (i >= 0x0B && i < 0x0B + 2) put("\x1b[38;5;36m");
        if (i >= 0x0D && i < 0x0D + 2) put("\x1b[38;5;36m");
        if (i >= 0x18 && i < 0x18 + 2) put("\x1b[38;5;36m");
        if (i >= 0x12 && i < 0x12 + 3) put("\x1b[38;5;58m");
        if (i == 0x15) put("\x1b[38;5;172m");
        if (f->fw < 0x07) { if (i >= 0x16 && i < 0x16 + 2) put("\x1b[38;5;11m"); }
        else { if (i >= 0x16 + 1 && i < 0x16 + 2) put("\x1b[38;5;11m"); }
        if (i >= 0x02 && i <= 0x03) put("\x1b[38;5;120m");
        if (i >= 0x04 && i <= 0x05) put("\x1b[38;5;110m");
        if (i >= 0x06 && i <= 0x07) put("\x1b[38;5;115m");
        if ((i == 0x16 && f->fw >= 0x07) || (i >= 0x24 && i <= 0x25)) put("\x1b[38;5;180m");
        if (i >= pc && i < pc + 2) put("\x1b[38;5;11m");
        n += snprintf(buf + n, buflen - n, "%02x", f->frame[i]);
        put(FR);
    }
    if (verbose) {
        n += snprintf(buf + n, buflen - n, " # %s%04x%s", "\x1b[38;5;11m", f->cs_calc, FR);
        if (f->fw < 0x07) n += snprintf(buf + n, buflen - n, " %s%s%s", f->blk_ok > 0 ? "\x1b[38;5;2m" : f->blk_ok < 0 ? "\x1b[38;5;220m" : "\x1b[38;5;1m", f->blk_ok > 0 ? "(ok)" : f->blk_ok < 0 ? "(oo)" : "(no)", TXT);
        n += snprintf(buf + n, buflen - n, " %s%s%s", f->cs_ok ? "\x1b[38;5;2m" : "\x1b[38;5;1m", f->cs_ok ? "[OK]" : "[NO]", TXT);
    }
    n += snprintf(buf + n, buflen - n, "%s", "\x1b[0m");
    return n;
}
// what print_frame() derives from the frame bytes before printing (m10mod.c:1049-1070; m20mod.c:875-907)
int sonde_m10_frame_finish(sonde_m10_frame_t *f) {
    if (!f) return SONDE_E_ARG;
    int aux = f->frame[0] - 0x64;
    if (aux < 0 || aux > 20) aux = 0;
    f->len = 101 + aux;
    f->cs_calc = (uint32_t)sonde::m10_checksum(f->frame, 99 + aux);
    f->cs_ok = ((uint32_t)((f->frame[99 + aux] << 8) | f->frame[100 + aux]) == f->cs_calc);
    return 0;
}
int sonde_m20_frame_finish(sonde_m20_frame_t *f) {
    if (!f) return SONDE_E_ARG;
    int flen = f->frame[0], pos_fw = 0x43;
    if (flen < 0x45) pos_fw = flen - 2;
    else if (flen - 0x45 > 64) flen = 0x45 + 64;
    const int pc = flen - 1;
    f->fw = pos_fw >= 0 ? f->frame[pos_fw] : 0;
    if (f->fw > 0x20) f->fw = 0;
    f->len = flen + 1;
    // length byte 0: the reference then reads the two bytes in FRONT of its frame buffer (m20mod.c:897-901: frame_bytes[-2], [-1] = two bytes of the
    // previous frame's serial number, gpx_t :112-113); they are taken as 0 here, which is what they are until a frame has been printed
    f->cs_calc = pc >= 0 ? (uint32_t)sonde::m10_checksum(f->frame, pc) : 0;
    f->cs_ok = pc >= 0 ? ((uint32_t)((f->frame[pc] << 8) | f->frame[pc + 1]) == f->cs_calc) : (f->frame[0] == 0);
    uint8_t blk[0x16]; blk[0] = 0x16; memcpy(blk + 1, f->frame + 2, 0x14);       // blk_checkM10 (m20mod.c:548-560): length byte, then the block
    const int bc2 = sonde::m10_checksum(blk, 0x15), bc1 = (f->frame[0x16] << 8) | f->frame[0x17];
    f->blk_ok = bc1 == bc2 ? 1 : bc1 == 0 ? -1 : 0;
    return 0;
}

int sonde_rs41_rawline(const sonde_frame_t *f, char *buf, size_t buflen) {
    if (!f || !buf || buflen < (size_t)(2 * f->len + 16)) return SONDE_E_ARG;
    int n = 0;
    for (int i = 0; i < f->len; i++) n += snprintf(buf + n, buflen - n, "%02x", f->frame[i]);
    n += snprintf(buf + n, buflen - n, f->ecc >= 0 ? " [OK]" : " [NO]");
    if (f->ecc > 0) n += snprintf(buf + n, buflen - n, " (%d)", f->ecc);
    if (f->ecc < 0) n += snprintf(buf + n, buflen - n, f->ecc == -1 ? " (-+)" : f->ecc == -2 ? " (+-)" : " (--)");
    return n;
}
int sonde_rs255_encode(uint8_t cw[255]) { sonde::rs255_encode(cw); return 0; }
int sonde_rs255_decode(uint8_t cw[255]) { return sonde::rs255_decode(cw); }
int sonde_crc16(const uint8_t *data, int len) { return sonde::crc16(data, len); }
}
