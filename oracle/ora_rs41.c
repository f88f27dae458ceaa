/*
 * oracle/ora_rs41.c — TEST INFRASTRUCTURE: CPU restatement of the RS41 framer, RS(255,231)
 * decoder and CRC (reference: demod/mod/rs41mod.c, demod/mod/bch_ecc_mod.c).
 *
 *  - GF(2^8)/0x11D tables                     bch_ecc_mod.c:136-166, bch_ecc_mod.h:52-56
 *  - syndromes, Sugiyama/Euclid key equation,  bch_ecc_mod.c:547-586, 596-660, 877-959
 *    Chien search by evaluation, Forney (b=0)
 *  - two interleaved codewords + 2nd pass      rs41mod.c:1703-1769, 1955-1974
 *  - frame bytes: LSB-first, XOR mask          rs41mod.c:224-234, 2900-2962
 *  - raw output line                           rs41mod.c:2530-2545
 */
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "ora_dsp.h"

/* ------------------------------------------------------------------ GF(256) */
static uint8_t gexp[512], glog[256];
static int gf_ready = 0;
static void gf_setup(void) {
    if (gf_ready) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        gexp[i] = (uint8_t)x; glog[x] = (uint8_t)i;
        x <<= 1; if (x & 0x100) x ^= 0x11D;
    }
    for (int i = 255; i < 512; i++) gexp[i] = gexp[i - 255];
    gf_ready = 1;
}
static inline uint8_t gmul(uint8_t a, uint8_t b) { return (a && b) ? gexp[glog[a] + glog[b]] : 0; }
static inline uint8_t ginv(uint8_t a) { return a ? gexp[255 - glog[a]] : 0; }

#define PN 256                       /* polynomial storage (degree <= 254) */
static int pdeg(const uint8_t *p) { int n = 254; while (n >= 0 && p[n] == 0) n--; return n; }
static uint8_t peval(const uint8_t *p, uint8_t x) {   /* p(x), plain Horner: same value as the reference's power form */
    uint8_t y = 0;
    for (int n = 254; n >= 0; n--) y = gmul(y, x) ^ p[n];
    return y;
}
static void pmul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    uint8_t c[PN]; memset(c, 0, PN);
    int da = pdeg(a), db = pdeg(b);
    if (da + db > 254) return;        /* reference refuses silently (poly_mul returns -1) */
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) c[i + j] ^= gmul(a[i], b[j]);
    memcpy(out, c, PN);
}
/* p = q*d + r  (bch_ecc_mod.c:407-449) */
static void pdivmod(const uint8_t *p, const uint8_t *q, uint8_t *d, uint8_t *r) {
    int dp = pdeg(p), dq = pdeg(q);
    memset(d, 0, PN); memset(r, 0, PN);
    if (dq < 0) return;
    if (dq == 0) { uint8_t c = ginv(q[0]); for (int i = 0; i <= dp; i++) d[i] = gmul(p[i], c); return; }
    if (dp < dq) { for (int i = 0; i <= dp; i++) r[i] = p[i]; return; }
    for (int i = 0; i <= dp; i++) r[i] = p[i];
    uint8_t qi = ginv(q[dq]);
    while (dp >= dq) {
        uint8_t c = gmul(r[dp], qi);
        d[dp - dq] = c;
        for (int i = 0; i <= dq; i++) r[dp - i] ^= gmul(q[dq - i], c);
        dp = pdeg(r);
    }
}

static uint8_t rs_gen[PN];
static int rs_ready = 0;
static void rs_setup(void) {
    if (rs_ready) return;
    gf_setup();
    memset(rs_gen, 0, PN); rs_gen[0] = 1;
    for (int i = 0; i < 24; i++) {            /* g(X) = prod (X - alpha^i), b = 0 (:742-763) */
        uint8_t f[PN]; memset(f, 0, PN); f[0] = gexp[i]; f[1] = 1;
        pmul(rs_gen, f, rs_gen);
    }
    rs_ready = 1;
}

int ora_rs255_encode(uint8_t cw[255]) {       /* parity into cw[0..23] (:860-874) */
    uint8_t m[PN], q[PN], r[PN];
    rs_setup();
    memset(m, 0, PN);
    for (int j = 24; j < 255; j++) m[j] = cw[j];
    pdivmod(m, rs_gen, q, r);
    for (int j = 0; j < 24; j++) cw[j] = r[j];
    return 0;
}

/* errors-only decode; returns number of corrected symbols, or <0 (same codes as the reference) */
int ora_rs255_decode(uint8_t cw[255], uint8_t *err_pos, uint8_t *err_val) {
    const int t = 12;
    uint8_t S[PN], r0[PN], r1[PN], r2[PN], s0[PN], s1[PN], s2[PN], quo[PN], c[PN];
    uint8_t epos[24], eval[24];
    int any = 0;
    rs_setup();
    memset(epos, 0, 24); memset(eval, 0, 24);
    memset(S, 0, PN); memset(c, 0, PN); memcpy(c, cw, 255);
    for (int i = 0; i < 2 * t; i++) { S[i] = peval(c, gexp[i]); if (S[i]) any = 1; }
    if (err_pos) memset(err_pos, 0, 24);
    if (err_val) memset(err_val, 0, 24);
    if (!any) return 0;

    /* S*Lambda = Omega mod x^2t by Euclid on (S, x^2t), stop at deg(remainder) < t (:547-586) */
    memcpy(r0, S, PN); memset(r1, 0, PN); r1[2 * t] = 1;
    memset(s0, 0, PN); s0[0] = 1; memset(s1, 0, PN);
    while (pdeg(r1) >= t) {
        pdivmod(r0, r1, quo, r2);
        memcpy(r0, r1, PN); memcpy(r1, r2, PN);
        memset(s2, 0, PN);
        pmul(quo, s1, s2);
        for (int i = 0; i < PN; i++) s2[i] ^= s0[i];
        memcpy(s0, s1, PN); memcpy(s1, s2, PN);
    }
    uint8_t *Om = r1, *La = s1;
    int dL = pdeg(La), dO = pdeg(Om);
    if (dO >= dL) return -3;
    if (La[0] == 0) return -2;
    uint8_t gi = ginv(La[0]);
    for (int i = 0; i <= dL; i++) La[i] = gmul(La[i], gi);
    for (int i = 0; i <= dO; i++) Om[i] = gmul(Om[i], gi);

    uint8_t dLa[PN]; memset(dLa, 0, PN);
    for (int i = 1; i <= dL; i += 2) dLa[i - 1] = La[i];
    int nerr = 0;
    for (int x = 1; x < 256 && nerr < dL; x++) {
        if (peval(La, (uint8_t)x) != 0) continue;
        uint8_t z = peval(dLa, (uint8_t)x);
        uint8_t Y = z ? gmul(gmul(peval(Om, (uint8_t)x), ginv(z)), ginv((uint8_t)x)) : 0;
        epos[nerr] = glog[ginv((uint8_t)x)];
        eval[nerr] = Y;
        nerr++;
    }
    if (nerr < dL) return -1;
    for (int i = 0; i < nerr; i++) cw[epos[i]] ^= eval[i];
    if (err_pos) memcpy(err_pos, epos, 24);
    if (err_val) memcpy(err_val, eval, 24);
    return nerr;
}

/* ------------------------------------------------------------------ CRC-16/CCITT-FALSE (:284-304) */
int ora_crc16(const uint8_t *p, int len) {
    int rem = 0xFFFF;
    for (int i = 0; i < len; i++) {
        rem ^= p[i] << 8;
        for (int j = 0; j < 8; j++) rem = (rem & 0x8000) ? ((rem << 1) ^ 0x1021) & 0xFFFF : (rem << 1) & 0xFFFF;
    }
    return rem;
}

/* ------------------------------------------------------------------ RS41 frame */
#define RS41_FRAME_MAX 518
#define RS41_NDATA     320
static const char rs41_hdr[] = "0000100001101101010100111000100001000100011010010100100000011111";
static const uint8_t rs41_hdr_bytes[8] = { 0x86, 0x35, 0xf4, 0x40, 0x93, 0xdf, 0x1a, 0x60 };
static const uint8_t rs41_mask[64] = {
    0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
    0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
    0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
    0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1 };

/* +4 .. -4 vote on byte 0x38: 0x0F -> std frame, 0xF0 -> extended (:407-415) */
static int rs41_frametype(const uint8_t *f) {
    int ft = 0; uint8_t b = f[0x38];
    for (int i = 0; i < 4; i++) ft += ((b >> i) & 1) - ((b >> (i + 4)) & 1);
    return ft;
}

/* ecc level 1: one pass; level 2: re-impose block ids / zero tail and retry (:1703-1769) */
int ora_rs41_ecc(uint8_t frame[RS41_FRAME_MAX], int frmlen, int level) {
    uint8_t cw1[255], cw2[255];
    int e1, e2, leak;
    if (frmlen > RS41_FRAME_MAX) frmlen = RS41_FRAME_MAX;
    leak = frmlen % 2;
    for (int i = frmlen; i < RS41_FRAME_MAX; i++) frame[i] = 0;
    for (int i = 0; i < 24; i++) { cw1[i] = frame[8 + i]; cw2[i] = frame[32 + i]; }
    for (int i = 0; i < 231; i++) { cw1[24 + i] = frame[56 + 2 * i]; cw2[24 + i] = frame[57 + 2 * i]; }
    e1 = ora_rs255_decode(cw1, NULL, NULL);
    e2 = ora_rs255_decode(cw2, NULL, NULL);
    if (level >= 2 && (e1 < 0 || e2 < 0)) {
        static const int pos[5] = { 0x039, 0x065, 0x093, 0x0B5, 0x112 };
        static const int pck[5] = { 0x7928, 0x7A2A, 0x7C1E, 0x7D59, 0x7B15 };
        for (int k = 0; k < 5; k++) { frame[pos[k]] = pck[k] >> 8; frame[pos[k] + 1] = pck[k] & 0xFF; }
        if (rs41_frametype(frame) < -2) {
            for (int i = RS41_NDATA + 7; i < RS41_FRAME_MAX - 2; i++) frame[i] = 0;
        } else {
            for (int i = RS41_NDATA; i < RS41_FRAME_MAX; i++) frame[i] = 0;
            frame[0x12B] = 0x76; frame[0x12C] = 0x11;
            for (int i = 0x12D; i < RS41_NDATA - 2; i++) frame[i] = 0;
            frame[RS41_NDATA - 2] = 0xEC; frame[RS41_NDATA - 1] = 0xC7;
        }
        for (int i = 0; i < 231; i++) { cw1[24 + i] = frame[56 + 2 * i]; cw2[24 + i] = frame[57 + 2 * i]; }
        e1 = ora_rs255_decode(cw1, NULL, NULL);
        e2 = ora_rs255_decode(cw2, NULL, NULL);
    }
    for (int i = 0; i < 24; i++) { frame[8 + i] = cw1[i]; frame[32 + i] = cw2[i]; }
    for (int i = 0; i < 231; i++) { frame[56 + 2 * i] = cw1[24 + i]; frame[57 + 2 * i] = cw2[24 + i]; }
    (void)leak;   /* 518 is even: the reference's odd-length tail write (:1963-1965) indexes past cw */
    if (e1 < 0 || e2 < 0) return -((e1 < 0 ? 1 : 0) | (e2 < 0 ? 2 : 0));
    return e1 + e2;
}

/* raw text line: hex bytes + ECC verdict (:2530-2545); returns length written */
int ora_rs41_rawline(const uint8_t *frame, int len, int ec, char *out) {
    int n = 0;
    for (int i = 0; i < len; i++) n += sprintf(out + n, "%02x", frame[i]);
    n += sprintf(out + n, ec >= 0 ? " [OK]" : " [NO]");
    if (ec > 0) n += sprintf(out + n, " (%d)", ec);
    if (ec < 0) n += sprintf(out + n, ec == -1 ? " (-+)" : ec == -2 ? " (+-)" : " (--)");
    return n;
}

/*
 * Whole-capture decode of one RS41 channel in the way `rs41mod -r --ecc<level> [--IQ fq] [--lpIQ] [--dc] - sr bps`
 * drives the seam (rs41mod.c:2873-2968, print_frame :2472-2553).
 * Outputs per frame f: frames[f*518 ..] (post-ECC bytes), flen[f], ecc[f], meta[4f..] = {mv, mv_pos, s_in_after, 0},
 * softbits[f*4080 ..] (optional).  Returns the number of frames.
 */
int ora_rs41_decode(const void *data, size_t nbytes, int sr, int bps, int iq_mode, double fq,
                    int lp_mask, int afc, int ecc_level, float thres, int max_frames,
                    uint8_t *frames, int *flen, int *ecc, double *meta, float *softbits, float *raw_pre_ecc) {
    ora_dsp d; memset(&d, 0, sizeof(d));
    d.src = (const uint8_t *)data; d.src_len = nbytes; d.src_pos = 0;
    d.sr_in = sr; d.bps = bps; d.iq_mode = iq_mode; d.lp_mask = lp_mask; d.afc = afc;
    if (iq_mode == 5 && afc) d.lp_mask |= ORA_LP_FM;            /* rs41mod.c:2747 */
    d.xlt_fq = -fq; d.baud = 4800.0f; d.symlen = 1; d.symhd = 1; d.bt = 0.5f; d.h = 0.6f;
    d.lpiq_bw = (int)7.4e3; d.lpfm_bw = (int)6e3; d.hdr = rs41_hdr; d.hdrlen = 64;
    if (ora_init(&d) < 0) return -1;
    const int bitofs = 2;
    const float bl = (iq_mode > 2) ? 2.0f : -1.0f;
    int nf = 0;
    uint8_t fr[RS41_FRAME_MAX];                                  /* gpx.frame persists across frames */
    memset(fr, 0, RS41_FRAME_MAX);
    memcpy(fr, rs41_hdr_bytes, 8);
    while (nf < max_frames) {
        if (ora_find_header(&d, thres, 4) < 0) break;
        if (d.mv * 0.5f < 0) continue;                           /* no -i, no --auto */
        int nbytes_got = 8, eof = 0;
        for (int bp = 0; nbytes_got < RS41_FRAME_MAX; ) {
            uint8_t byte = 0;
            int k;
            for (k = 0; k < 8; k++, bp++) {
                ora_bit b, b1;
                if (ora_softbit2p(&d, &b, 0, bitofs, bp, bl, 0, &b1) < 0) { eof = 1; break; }
                if (softbits) softbits[(size_t)nf * 4080 + bp] = b.sb;
                byte |= (uint8_t)(b.hb << k);                    /* LSB first */
            }
            if (eof) break;
            fr[nbytes_got] = byte ^ rs41_mask[nbytes_got % 64];
            nbytes_got++;
        }
        meta[4 * nf] = d.mv; meta[4 * nf + 1] = d.mv_pos; meta[4 * nf + 2] = d.s_in; meta[4 * nf + 3] = nbytes_got;
        /* print_frame: short read -> zero tail; length from frame type byte */
        int len = nbytes_got;
        if (len < 0x093) for (int i = len; i < RS41_FRAME_MAX; i++) fr[i] = 0;
        len = (rs41_frametype(fr) >= 0) ? RS41_NDATA : RS41_FRAME_MAX;
        if (raw_pre_ecc) for (int i = 0; i < RS41_FRAME_MAX; i++) raw_pre_ecc[(size_t)nf * RS41_FRAME_MAX + i] = fr[i];
        ecc[nf] = ecc_level ? ora_rs41_ecc(fr, len, ecc_level) : 0;
        flen[nf] = len;
        memcpy(frames + (size_t)nf * RS41_FRAME_MAX, fr, RS41_FRAME_MAX);
        nf++;
        if (eof) break;
    }
    ora_free(&d);
    return nf;
}

/* front-end only: per-IF-sample taps for stream parity (mirrors oracle/ref_harness.c:ref_streams) */
int ora_streams(const void *data, size_t nbytes, int sr, int bps, int iq_mode, double xlt_fq,
                int lp_mask, int afc, float baud, float bt, float h, int lpiq_bw, int lpfm_bw,
                const char *hdr, int symlen, int symhd, int max_if,
                float *iq_out, float *fm_out, float *bufs_out, int *consts) {
    ora_dsp d; memset(&d, 0, sizeof(d));
    d.src = (const uint8_t *)data; d.src_len = nbytes;
    d.sr_in = sr; d.bps = bps; d.iq_mode = iq_mode; d.lp_mask = lp_mask; d.afc = afc;
    d.xlt_fq = xlt_fq; d.baud = baud; d.symlen = symlen; d.symhd = symhd; d.bt = bt; d.h = h;
    d.lpiq_bw = lpiq_bw; d.lpfm_bw = lpfm_bw; d.hdr = hdr; d.hdrlen = (int)strlen(hdr);
    if (ora_init(&d) < 0) return -1;
    if (consts) {
        consts[0] = d.N; consts[1] = d.M; consts[2] = d.L; consts[3] = d.K; consts[4] = d.delay;
        consts[5] = d.dectaps; consts[6] = d.decM; consts[7] = d.lut_len; consts[8] = d.lpiq_taps;
        consts[9] = d.lpfm_taps; consts[10] = d.sr;
    }
    int n = 0;
    while (n < max_if && ora_sample(&d, 0) == 0) {
        uint32_t s = d.s_in - 1;
        if (iq_out && d.iq_mode) { iq_out[2 * n] = d.ziq[s % (uint32_t)d.N].re; iq_out[2 * n + 1] = d.ziq[s % (uint32_t)d.N].im; }
        if (fm_out) fm_out[n] = d.fmb[s % (uint32_t)d.M];
        if (bufs_out) bufs_out[n] = d.bufs[s % (uint32_t)d.M];
        n++;
    }
    ora_free(&d);
    return n;
}

/* Any sonde of the family through the restatement: header hits (score, position) and their soft bits — the same contract as
 * ref_softframes() of oracle/ref_harness.c (which drives the reference's own demod_mod.o), so that tests can hand both the same configuration.
 * Polarity as rs41mod without -i / --auto: hits with a negative score are skipped (rs41mod.c:2888-2891). */
typedef struct {
    int sr, bps, opt_iq, opt_lp, opt_dc, opt_iqdc, opt_min, opt_nolut;
    double xlt_fq; float baud; int symlen, symhd; float BT, h; int lpIQ_bw, lpFM_bw; const char *hdr;
} ora_cfg_t;

int ora_softframes(const ora_cfg_t *c, const void *data, size_t nbytes, float thres, int hdmax, int bitofs, float l, int nbits, int max_hits,
                   double *hits, float *sb, float *sb1) {
    ora_dsp d; memset(&d, 0, sizeof(d));
    d.src = (const uint8_t *)data; d.src_len = nbytes; d.src_pos = 0;
    d.sr_in = c->sr; d.bps = c->bps; d.iq_mode = c->opt_iq; d.lp_mask = c->opt_lp; d.afc = c->opt_dc; d.iqdc = c->opt_iqdc; d.if_min = c->opt_min;
    d.xlt_fq = c->xlt_fq; d.baud = c->baud; d.symlen = c->symlen; d.symhd = c->symhd; d.bt = c->BT; d.h = c->h;
    d.lpiq_bw = c->lpIQ_bw; d.lpfm_bw = c->lpFM_bw; d.hdr = c->hdr; d.hdrlen = (int)strlen(c->hdr);
    if (ora_init(&d) < 0) return -1;
    int nh = 0;
    while (nh < max_hits) {
        if (ora_find_header(&d, thres, hdmax) < 0) break;
        if (d.mv * 0.5f < 0) continue;
        hits[4 * nh] = d.mv; hits[4 * nh + 1] = d.mv_pos;
        int i, q = 0;
        for (i = 0; i < nbits; i++) {
            ora_bit a, b;
            q = ora_softbit2p(&d, &a, 0, bitofs, i, l, 0, &b);
            if (q < 0) break;
            sb[(size_t)nh * nbits + i] = a.sb; sb1[(size_t)nh * nbits + i] = b.sb;
        }
        hits[4 * nh + 2] = i; hits[4 * nh + 3] = d.s_in;
        nh++;
        if (q < 0) break;
    }
    ora_free(&d);
    return nh;
}
