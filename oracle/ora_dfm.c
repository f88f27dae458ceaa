/*
 * oracle/ora_dfm.c — TEST INFRASTRUCTURE: CPU restatement of the DFM06/09 framer of the reference
 * (demod/mod/dfm09mod.c): header search on the raw Manchester header (:141, main loop :1625-1722: thres 0.65,
 * hdmax 2, bitofs 2, symlen 2, centre window l = 4 in IQ mode, nfrms = 8 frames sliced per header hit without
 * re-correlating), block de-interleave (:231), Hamming(8,4) with 1-bit fix and the soft 2-bit pass (:240-310),
 * raw text line (:1198-1236).
 */
#include <stdio.h>
#include <string.h>
#include "ora_dsp.h"

#define DFM_BITS 280
static const char dfm_rawhdr[] = "10011010100110010101101001010101";
static const uint8_t Hm[4][8] = { {0,1,1,1,1,0,0,0}, {1,0,1,1,0,1,0,0}, {1,1,0,1,0,0,1,0}, {1,1,1,0,0,0,0,1} };
static const uint8_t He[8] = { 0x7, 0xB, 0xD, 0xE, 0x8, 0x4, 0x2, 0x1 };

static void codeword_of(int n, uint8_t *c) {          /* systematic generator, dfm09mod.c:181-188 */
    uint8_t d[4] = { (uint8_t)((n >> 3) & 1), (uint8_t)((n >> 2) & 1), (uint8_t)((n >> 1) & 1), (uint8_t)(n & 1) };
    c[0] = d[0]; c[1] = d[1]; c[2] = d[2]; c[3] = d[3];
    c[4] = d[1] ^ d[2] ^ d[3]; c[5] = d[0] ^ d[2] ^ d[3]; c[6] = d[0] ^ d[1] ^ d[3]; c[7] = d[0] ^ d[1] ^ d[2];
}

static int check8(int ecc, ora_bit code[8]) {
    unsigned syn = 0;
    int ret = 0;
    for (int i = 0; i < 4; i++) {
        uint8_t s = 0;
        for (int j = 0; j < 8; j++) s ^= Hm[i][j] & code[j].hb;
        syn = (syn << 1) | s;
    }
    if (syn) {
        ret = -1;
        for (int j = 0; j < 8; j++) if (syn == He[j]) { ret = j + 1; break; }
    }
    if (ret > 0) code[ret - 1].hb ^= 1;
    else if (ret < 0 && ecc == 2) {
        int best = -1; float bestsum = 0.0f;
        for (int n = 0; n < 16; n++) {
            uint8_t c[8]; int d = 0;
            codeword_of(n, c);
            for (int i = 0; i < 8; i++) d += (code[i].hb != c[i]);
            if (d == 2) {
                float sum = 0.0f;
                for (int i = 0; i < 8; i++) sum += (2 * c[i] - 1) * code[i].sb;
                if (sum >= bestsum) { bestsum = sum; best = n; }
            }
        }
        if (best >= 0) { uint8_t c[8]; codeword_of(best, c); for (int i = 0; i < 8; i++) code[i].hb = c[i]; }
    }
    return ret;
}

static int block_decode(int ecc, const ora_bit *str, int L, uint8_t *nib) {
    ora_bit blk[13 * 8];
    int ret = 0;
    for (int j = 0; j < 8; j++) for (int i = 0; i < L; i++) blk[8 * i + j] = str[L * j + i];
    for (int i = 0; i < L; i++) {
        if (ecc) {
            int e = check8(ecc, blk + 8 * i);
            if (e > 0) ret |= (1 << i);
            if (e < 0) ret |= e;
        }
        nib[i] = (uint8_t)((blk[8 * i].hb << 3) | (blk[8 * i + 1].hb << 2) | (blk[8 * i + 2].hb << 1) | blk[8 * i + 3].hb);
    }
    return ret;
}

/* one `-r --ecc` text line from 280 frame bits; returns strlen */
int ora_dfm_rawline(const ora_bit *frame, int ecc, char *out) {
    static const int off[3] = { 16, 72, 176 }, len[3] = { 7, 13, 13 };
    int n = 0;
    for (int b = 0; b < 3; b++) {
        uint8_t nib[13];
        int r = block_decode(ecc, frame + off[b], len[b], nib);
        if (b) n += sprintf(out + n, "  ");
        for (int i = 0; i < len[b]; i++) n += sprintf(out + n, "%01X", nib[i]);
        if (ecc) n += sprintf(out + n, r == 0 ? " [OK] " : r > 0 ? " [KO] " : " [NO] ");
    }
    return n;
}

/*
 * Whole-capture decode in the way `dfm09mod -r --ecc<level> --IQ fq [--lpIQ] - sr bps` drives the seam.
 * lines: max_lines x 96 chars; meta[4h..] = {mv, mv_pos, bits_read, 0} per header hit; soft[h*2224 + i] optional.
 */
int ora_dfm_decode(const void *data, size_t nbytes, int sr, int bps, int iq_mode, double fq, int lp_mask, int ecc,
                   float thres, int max_lines, char *lines, int max_hits, double *meta, float *soft, int *nhits_out) {
    ora_dsp d; memset(&d, 0, sizeof(d));
    d.src = (const uint8_t *)data; d.src_len = nbytes;
    d.sr_in = sr; d.bps = bps; d.iq_mode = iq_mode; d.lp_mask = lp_mask; d.afc = 0;
    d.xlt_fq = -fq; d.baud = 2500.0f; d.symlen = 2; d.symhd = 2; d.bt = 0.5f; d.h = 1.8f;
    d.lpiq_bw = (int)12e3; d.lpfm_bw = (int)4e3; d.hdr = dfm_rawhdr; d.hdrlen = 32;
    if (ora_init(&d) < 0) return -1;
    const int bitofs = 2, nfrms = 8;
    const float bl = (iq_mode > 2) ? 4.0f : -1.0f;
    ora_bit frame[DFM_BITS];
    memset(frame, 0, sizeof frame);
    int nl = 0, nh = 0;
    while (nl < max_lines) {
        if (ora_find_header(&d, thres, 2) < 0) break;
        if (d.mv * 0.5f < 0) continue;
        int bitpos = 0, pos = 16, eof = 0;
        if (nh < max_hits) { meta[4 * nh] = d.mv; meta[4 * nh + 1] = d.mv_pos; }
        for (int frm = 0; frm < nfrms && nl < max_lines; frm++) {
            while (pos < DFM_BITS) {
                ora_bit b, b1;
                if (ora_softbit2p(&d, &b, 0, bitofs, bitpos, bl, 0, &b1) < 0) { eof = 1; break; }
                if (soft && nh < max_hits) soft[(size_t)nh * 2224 + bitpos] = b.sb;
                frame[pos++] = b; bitpos++;
            }
            if (pos < DFM_BITS) break;
            int n = ora_dfm_rawline(frame, ecc, lines + (size_t)nl * 96);
            lines[(size_t)nl * 96 + n] = 0;
            nl++; pos = 0;
        }
        if (nh < max_hits) { meta[4 * nh + 2] = bitpos; meta[4 * nh + 3] = d.s_in; nh++; }
        if (eof) break;
    }
    if (nhits_out) *nhits_out = nh;
    ora_free(&d);
    return nl;
}
