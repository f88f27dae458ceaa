/*
 * host/seam/demod_mod_hip.c — the reference's function-level seam (demod/mod/demod_mod.h:179-192) on top of libsonde_hip.
 *
 * Drop this file into the reference's demod/mod/ in place of demod_mod.c (it includes the reference's own demod_mod.h, which is not
 * shipped here) and link the decoders against libsonde_hip:
 *
 *     gcc -O2 -I<repo>/include -I<repo>/host rs41mod.c bch_ecc_mod.c demod_mod_hip.c -L<repo>/radiosonde_auto_rx_amd -lsonde_hip -lm -o rs41mod
 *
 * The reference's own main() then drives the GPU engine: every option, every print_frame() / print_position() of rs41mod.c,
 * dfm09mod.c, m10mod.c, m20mod.c stays the reference's code; only the sample-rate work behind find_header() / read_softbit*()
 * runs on the MI355X.  oracle/Makefile builds exactly that (the *_seam binaries under oracle/_ref) and tests/test_gpu_seam.py compares it with the
 * all-CPU reference binaries.
 *
 *   init_buffers()   demod_mod.c:1208   -> sonde_engine_create() from the dsp_t fields the caller filled in (rs41mod.c:2816-2836);
 *                                          writes back sr / sps / _spb / decM / dectaps / L / M / K / delay, prints IF: / dec:
 *   find_header()    :1533              -> pull: read a block from dsp->fp, sonde_engine_process_host(), sonde_engine_fetch_hits();
 *                                          sets dsp->mv / mv_pos, returns 1 per header (either polarity: the caller decides), EOF at the end
 *   read_softbit2p() :1087, read_softbit() :1012, read_slbit() :942  -> soft bit `pos` of the current hit (raw polarity)
 *   free_buffers()   :1476              -> sonde_engine_destroy()
 *   read_wav_header() :313, f32soft_read() :1718, find_softbinhead() :1740, find_binhead() :1668  -> host code, no GPU
 *
 * Decoders: rs41mod, dfm09mod, m10mod, m20mod (the engine's own presets) and, through the engine's generic sonde description filled from the
 * caller's dsp_t, rs92mod, imet54mod, mp3h1mod, mts01mod, meisei100mod, lms6Xmod (kFamily below).  lms6Xmod rewrites dsp.br / dsp.sps after
 * init_buffers() (lms6Xmod.c:1336-1348 for --lms6 / --lmsX): the seam picks the value up at the first find_header() call and re-creates the engine with that
 * symbol rate for the bit clock and slicers (sonde_generic_t.slice_baud), 4096 raw bits per block for LMS6, 4720 for LMS-X.  A change of dsp.sps later in the
 * stream (auto detection switching between LMS6 and LMS-X, :1436-1462) sets the engine up again the same way, fed from the input history (3 s) starting
 * 64 bits before the end of the block handed out last; hits before that block's end are dropped, mv_pos stays a position in the whole stream.
 *
 * Differences to demod_mod.c a caller can see: one dsp_t at a time (the reference keeps file-static state too); thres / hdmax / bitofs are
 * taken from the first find_header call; a header of the wrong polarity that the
 * caller skips still has its frame consumed; f32buf_sample() is not part of the seam (a call ends the program with a message).
 * The bit readers take per-call arguments the batched engine fixed when the hit was sliced: `ofs` must be the bitofs given to find_header() and
 * `l` the window this decoder always passes (-1, or its centre window for opt_iq > 2) — every decoder of the family does exactly that; any other
 * value, and spike != 0, ends the program with a message instead of returning bits that were sliced differently.  (spike: the reference's
 * clipping compares against a local `avg` that is read before it is ever written, demod_mod.c:945,975 / :1016,1046 / :1092,1121 — its output
 * depends on what the compiler left in that register, so there is nothing well defined to mirror.)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "demod_mod.h"          /* the reference's header */
#include "sonde_hip.h"
#include "wav_header.h"

#define SEAM_MAXHITS 16

static struct {
    sonde_engine_t *eng;
    sonde_info_t info;
    int nbits;                  /* soft bits per hit of this sonde type */
    size_t unit;                /* bytes per input sample */
    int chunk;
    /* input history: the samples since hist0 (an input sample index, multiple of decM), ~3 s — enough to set the engine up again from the end of a
     * block when the caller changes the bit clock in mid-stream (lms6Xmod.c:1436-1462: LMS6 <-> LMS-X) */
    char *hist; size_t cap; int64_t hist0, hist_n, fed, eng0, skip_before, keep;
    int64_t last_gpos;          /* IF-rate position (whole stream) of the hit handed out last */
    int eof, started;
    sonde_hit_t hit[SEAM_MAXHITS];
    float *soft, *soft1;
    int qn, qi;
    const float *cur, *cur1; int cur_nbits, cur_inv;
    int bitofs; float l_win;      /* what the hits were sliced with: find_header()'s bitofs, the decoder's window */
    float sps_cur; int fam;       /* dsp->sps the engine slices with; index into kFamily or -1 */
    sonde_cfg_t cfg; sonde_generic_t gen; double fq; int generic;
} S;

/* How the bit loop of each other decoder of the family consumes a header hit (what demod_mod.c learns one call at a time, the batched
 * engine has to know up front): bits read per hit, the centre window it passes for opt_iq > 2, and the hdmax / bitofs defaults of its
 * find_header() call (replaced by the real arguments at the first call). */
static const struct { const char *name; int br, hdrlen, symlen, nbits, hdmax, bitofs; float l; const char *hdr16; } kFamily[] = {
    { "rs92mod",      4800, 60, 2, (240 - 6) * 10, 3, 2, 4.0f },      /* rs92mod.c:1992,2010-2040: 234 bytes of 10 bits, read_slbit            */
    { "imet54mod",    4798, 40, 1, 220 * 10,       4, 1, 2.0f },      /* imet54mod.c:1013,1029-1060                                            */
    { "mp3h1mod",     2399, 44, 2, 51 * 8 - 22,    2, 2, 2.0f },      /* mp3h1mod.c:1181,1196-1235: bitfrm_len (45+6)*8 from pos 22            */
    { "mts01mod",     1200, 32, 1, 8 * 131,        2, 0, 2.0f },      /* mts01mod.c:572,588-612                                                */
    { "meisei100mod", 2400, 48, 1, 1200 - 48,      1, 0, -1.0f },     /* meisei100mod.c:691,704-718: 2*600-48 raw bits, read_slbit             */
    { "lms6Xmod",     4800, 64, 1, 261 * 16 - 80, 10, 0, -1.0f, "0101011000001000" },   /* lms6Xmod.c:89-92,101,1358,1394: RAWBITBLOCK_LEN_6 - BLOCKSTART bits */
};

static int seam_type(const dsp_t *dsp, int *fam) {
    const int br = (int)(dsp->br + 0.5f);
    *fam = -1;
    for (int i = 0; i < (int)(sizeof kFamily / sizeof kFamily[0]); i++)          /* same rates as another type: told apart by the header itself */
        if (kFamily[i].hdr16 && br == kFamily[i].br && dsp->hdrlen == kFamily[i].hdrlen && dsp->symlen == kFamily[i].symlen &&
            dsp->hdr && strncmp(dsp->hdr, kFamily[i].hdr16, 16) == 0) { *fam = i; return SONDE_GENERIC; }
    if (dsp->hdrlen == 64 && br == 4800 && dsp->symlen == 1) return SONDE_RS41;
    if (dsp->hdrlen == 32 && br == 2500) return SONDE_DFM09;
    if (dsp->hdrlen == 32 && (br == 9615 || br == 9616)) return SONDE_M10;
    if (dsp->hdrlen == 32 && br == 9600) return SONDE_M20;
    for (int i = 0; i < (int)(sizeof kFamily / sizeof kFamily[0]); i++)
        if (!kFamily[i].hdr16 && br == kFamily[i].br && dsp->hdrlen == kFamily[i].hdrlen && dsp->symlen == kFamily[i].symlen) { *fam = i; return SONDE_GENERIC; }
    return -1;
}

int init_buffers(dsp_t *dsp) {
    sonde_cfg_t cfg;
    double fq = -dsp->xlt_fq;
    int fam;
    const int type = seam_type(dsp, &fam);
    if (S.eng) { fprintf(stderr, "demod_mod_hip: one dsp_t at a time\n"); return -1; }
    if (type < 0) { fprintf(stderr, "demod_mod_hip: sonde type (baud %.0f, header %d) not supported\n", dsp->br, dsp->hdrlen); return -1; }
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SONDE_ABI_VERSION;
    cfg.n_channels = 1;
    cfg.sample_rate = dsp->sr;
    cfg.bits = dsp->bps;
    cfg.sonde_type = type;
    cfg.opt_lp = dsp->opt_lp & (SONDE_LP_IQ | SONDE_LP_FM);
    cfg.opt_dc = dsp->opt_dc != 0;
    cfg.opt_min = dsp->opt_IFmin != 0;
    cfg.lpiq_bw = dsp->lpIQ_bw;
    cfg.opt_iqdc = dsp->opt_iqdc != 0;
    cfg.opt_nolut = dsp->opt_nolut != 0;
    cfg.opt_auto = 1;                               /* headers of both polarities are reported; the caller skips or flips */
    cfg.keep_soft = 2;                              /* both soft bits of read_softbit2p */
    cfg.max_frames = SEAM_MAXHITS;
    cfg.max_chunk = dsp->sr;
    switch (dsp->opt_iq) {
        case 0: cfg.input = SONDE_IN_AUDIO; cfg.audio_channels = dsp->nch < 1 ? 1 : dsp->nch;
                cfg.audio_select = (dsp->ch >= 0 && dsp->ch < cfg.audio_channels) ? dsp->ch : 0; break;
        case 1: cfg.input = SONDE_IN_IFIQ0; break;
        case 2: cfg.input = SONDE_IN_IFIQ2; break;
        case 3: cfg.input = SONDE_IN_IFIQ3; break;
        case 5: cfg.input = SONDE_IN_IQ; break;
        default: return -1;
    }
    if (dsp->opt_iq && dsp->nch != 2) return -1;
    memset(&S.gen, 0, sizeof S.gen);
    if (type == SONDE_GENERIC) {
        if (dsp->hdrlen > 64) return -1;
        memcpy(S.gen.header, dsp->hdr, (size_t)dsp->hdrlen);
        S.gen.baud = dsp->br; S.gen.bt = dsp->BT; S.gen.h = dsp->h; S.gen.symlen = dsp->symlen; S.gen.symhd = dsp->symhd;
        S.gen.hdmax = kFamily[fam].hdmax; S.gen.bitofs = kFamily[fam].bitofs; S.gen.nbits = kFamily[fam].nbits; S.gen.l_win = kFamily[fam].l;
        S.gen.lpiq_bw = dsp->lpIQ_bw; S.gen.lpfm_bw = dsp->lpFM_bw;
        cfg.lpiq_bw = 0;
    }
    S.generic = type == SONDE_GENERIC; S.cfg = cfg; S.fq = fq;
    /* centre window only for opt_iq > 2 (rs41mod.c:2920-2921, dfm09mod.c:1692-1694, m10mod.c:1470-1472, ...), whole symbols otherwise */
    S.l_win = dsp->opt_iq > 2 ? (type == SONDE_GENERIC ? kFamily[fam].l : type == SONDE_RS41 ? 2.0f : 4.0f) : -1.0f;
    int rc = S.generic ? sonde_engine_create_generic(&cfg, &fq, &S.gen, &S.eng) : sonde_engine_create(&cfg, &fq, &S.eng);
    if (rc < 0) { fprintf(stderr, "demod_mod_hip: %s\n", sonde_strerror(rc)); S.eng = NULL; return -1; }
    sonde_engine_info(S.eng, &S.info);
    if (dsp->opt_iq == 5) {
        dsp->sr_base = (ui32_t)dsp->sr;
        dsp->sr = S.info.if_sr;
        dsp->sps /= (float)S.info.decM;
        dsp->_spb /= (float)S.info.decM;
        dsp->decM = S.info.decM;
        dsp->dectaps = (ui32_t)S.info.dectaps;
        fprintf(stderr, "IF: %d\n", S.info.if_sr);
        fprintf(stderr, "dec: %d\n", S.info.decM);
    }
    dsp->L = S.info.L; dsp->M = S.info.M; dsp->K = S.info.K; dsp->delay = (ui32_t)S.info.delay;
    S.nbits = type == SONDE_GENERIC ? S.gen.nbits : type == SONDE_RS41 ? 510 * 8 : type == SONDE_DFM09 ? 264 + 7 * 280 : type == SONDE_M10 ? (101 + 20) * 8 : (101 + 64) * 8;
    S.unit = (size_t)(dsp->opt_iq ? 2 : cfg.audio_channels) * (size_t)(dsp->bps / 8);
    S.chunk = cfg.sample_rate / 10;
    S.chunk -= S.chunk % S.info.decM;
    if (S.chunk < S.info.decM) S.chunk = S.info.decM;
    S.keep = (int64_t)cfg.sample_rate * 3;
    S.cap = (size_t)(S.keep + 2 * (int64_t)S.chunk) * S.unit;
    S.hist = (char *)malloc(S.cap);
    S.soft = (float *)malloc((size_t)SEAM_MAXHITS * S.nbits * sizeof(float));
    S.soft1 = (float *)malloc((size_t)SEAM_MAXHITS * S.nbits * sizeof(float));
    S.hist0 = S.hist_n = S.fed = S.eng0 = S.skip_before = 0; S.last_gpos = 0;
    S.eof = 0; S.started = 0; S.qn = S.qi = 0; S.cur = NULL; S.cur_nbits = 0;
    S.sps_cur = dsp->sps; S.fam = fam;
    if (!S.hist || !S.soft || !S.soft1) return -1;
    return S.info.K;
}

int free_buffers(dsp_t *dsp) {
    (void)dsp;
    if (S.eng) sonde_engine_destroy(S.eng);
    free(S.hist); free(S.soft); free(S.soft1);
    memset(&S, 0, sizeof S);
    return 0;
}

int find_header(dsp_t *dsp, float thres, int hdmax, int bitofs, int opt_dc) {
    (void)opt_dc;
    if (!S.eng) return EOF;
    if (S.generic && dsp->sps != S.sps_cur) {
        /* the caller changed dsp.br / dsp.sps: after init_buffers() and before the first search (lms6Xmod.c:1336-1348), or in mid-stream behind the block
         * that showed the other type (:1436-1462).  Same filters and header template, new bit clock: the engine is set up again — in mid-stream from 64
         * bits before the end of that block, out of the input history (the reference carries its filter state over that point; the same frames follow) */
        const int midstream = S.started;
        const double sps_old = S.sps_cur;
        const int64_t block_end = S.last_gpos + (int64_t)((double)S.cur_nbits * sps_old);
        sonde_engine_destroy(S.eng); S.eng = NULL;
        S.gen.slice_baud = dsp->br;
        if (S.fam >= 0 && kFamily[S.fam].hdr16) S.gen.nbits = (dsp->br > 4799.9f && dsp->br < 4800.1f) ? kFamily[S.fam].nbits : 300 * 16 - 80;      /* LMS-X: RAWBITBLOCK_LEN - BLOCKSTART */
        double fq = S.fq;
        const int rc = sonde_engine_create_generic(&S.cfg, &fq, &S.gen, &S.eng);
        if (rc < 0) { fprintf(stderr, "demod_mod_hip: %s\n", sonde_strerror(rc)); S.eng = NULL; return EOF; }
        S.nbits = S.gen.nbits; S.sps_cur = dsp->sps;
        free(S.soft); free(S.soft1);
        S.soft = (float *)malloc((size_t)SEAM_MAXHITS * S.nbits * sizeof(float));
        S.soft1 = (float *)malloc((size_t)SEAM_MAXHITS * S.nbits * sizeof(float));
        if (!S.soft || !S.soft1) return EOF;
        if (midstream) {
            int64_t from = (block_end - (int64_t)(64 * sps_old)) * S.info.decM;
            from -= from % S.info.decM;
            if (from < S.hist0) from = S.hist0;
            if (from > S.fed) from = S.fed;
            S.eng0 = from; S.fed = from; S.skip_before = block_end;
            S.qn = S.qi = 0; S.cur = S.cur1 = NULL; S.cur_nbits = 0;
            if (S.eof) S.eof = 1;
            S.started = 0;                          /* threshold / sync arguments are applied to the new engine below */
        }
    }
    else if (dsp->sps != S.sps_cur) {
        fprintf(stderr, "demod_mod_hip: dsp.sps changed from %g to %g in mid-stream; only the generic family follows that\n", S.sps_cur, dsp->sps);
        exit(2);
    }
    if (!S.started) {                                  /* the caller's threshold, accepted header errors and bit offset (e.g. -d <shift>) */
        sonde_engine_set_threshold(S.eng, thres);
        if (sonde_engine_set_sync(S.eng, hdmax, bitofs) < 0) { fprintf(stderr, "demod_mod_hip: hdmax %d / bitofs %d out of range\n", hdmax, bitofs); return EOF; }
        S.bitofs = bitofs;
        S.started = 1;
    }
    for (;;) {
        while (S.qi < S.qn) {
            const int idx = S.qi++;
            const sonde_hit_t *h = &S.hit[idx];
            const int64_t gpos = S.eng0 / S.info.decM + (int64_t)h->mv_pos;       /* IF-rate position in the whole stream */
            if (gpos < S.skip_before) continue;                                      /* belongs to a block already handed out before a restart */
            S.cur = S.soft + (size_t)idx * S.nbits; S.cur1 = S.soft1 + (size_t)idx * S.nbits;
            S.cur_nbits = h->nbits; S.cur_inv = h->mv < 0.f;
            S.last_gpos = gpos;
            dsp->mv = h->mv; dsp->mv_pos = (ui32_t)gpos;
            return 1;
        }
        S.cur = S.cur1 = NULL; S.cur_nbits = 0;
        if (S.eof == 2) return EOF;                                                 /* the stream's last hits have been handed out */
        if (!S.eof && S.hist0 + S.hist_n - S.fed < S.chunk) {          /* (after a restart: first catch up with what is held) */
            if ((size_t)(S.hist_n + S.chunk) * S.unit > S.cap) {                        /* drop what is older than `keep` */
                int64_t drop = S.hist_n - S.keep;
                drop -= drop % S.info.decM;
                if (drop > S.fed - S.hist0) drop = S.fed - S.hist0;
                if (drop > 0) { memmove(S.hist, S.hist + (size_t)drop * S.unit, (size_t)(S.hist_n - drop) * S.unit); S.hist0 += drop; S.hist_n -= drop; }
            }
            const size_t got = fread(S.hist + (size_t)S.hist_n * S.unit, S.unit, (size_t)S.chunk, dsp->fp);
            S.hist_n += (int64_t)got;
            if (got == 0) S.eof = 1;
        }
        int64_t n = S.hist0 + S.hist_n - S.fed;
        if (n > S.chunk) n = S.chunk;
        n -= n % S.info.decM;
        if (n > 0) {
            const int rc = sonde_engine_process_host(S.eng, S.hist + (size_t)(S.fed - S.hist0) * S.unit, n, (int32_t)n);
            if (rc < 0) { fprintf(stderr, "demod_mod_hip: %s\n", sonde_strerror(rc)); return EOF; }
            S.fed += n;
        }
        const int at_end = S.eof && S.hist0 + S.hist_n - S.fed < S.info.decM;
        if (n <= 0 && !at_end) continue;
        S.qn = sonde_engine_fetch_hits(S.eng, S.hit, SEAM_MAXHITS, at_end);
        if (S.qn < 0) { fprintf(stderr, "demod_mod_hip: %s\n", sonde_strerror(S.qn)); S.qn = 0; return EOF; }
        if (S.qn > 0) { sonde_engine_fetch_soft(S.eng, S.soft, S.qn); sonde_engine_fetch_soft1(S.eng, S.soft1, S.qn); }
        S.qi = 0;
        if (at_end) { if (S.qn == 0) return EOF; S.eof = 2; }             /* the stream's last hits are handed out, then EOF */
    }
}

static void seam_reject(const char *what, double got, double want) {
    fprintf(stderr, "demod_mod_hip: %s = %g is not what the hit was sliced with (%g); the GPU engine cannot re-slice per call\n", what, got, want);
    exit(2);
}

static int seam_bit(int inv, int ofs, float l, int spike, int pos, float *sb, float *sb1) {
    if (!S.cur || pos < 0 || pos >= S.cur_nbits) return EOF;
    if (spike) { fprintf(stderr, "demod_mod_hip: spike clipping is not supported (undefined in the reference: demod_mod.c:945,975)\n"); exit(2); }
    if (ofs != S.bitofs) seam_reject("ofs", ofs, S.bitofs);
    if ((l < 0) != (S.l_win < 0) || (l >= 0 && l != S.l_win)) seam_reject("l", l, S.l_win);
    float s = S.cur[pos], s1 = S.cur1[pos];
    if (S.cur_inv) { s = -s; s1 = -s1; }            /* the engine stores the bits in the polarity in effect; the reference returns them raw */
    if (inv) { s = -s; s1 = -s1; }
    *sb = s; *sb1 = s1;
    return 0;
}

int read_softbit2p(dsp_t *dsp, hsbit_t *shb, int inv, int ofs, int pos, float l, int spike, hsbit_t *shb1) {
    float s, s1;
    (void)dsp;
    if (seam_bit(inv, ofs, l, spike, pos, &s, &s1) == EOF) return EOF;
    shb->sb = s; shb->hb = (s >= 0.f);
    if (shb1) { shb1->sb = s1; shb1->hb = (s1 >= 0.f); }
    return 0;
}

int read_softbit(dsp_t *dsp, hsbit_t *shb, int inv, int ofs, int pos, float l, int spike) {
    return read_softbit2p(dsp, shb, inv, ofs, pos, l, spike, NULL);
}

/* hard bit; behind the end of a hit (the M10 / M20 "rest of the second") the engine has already skipped: 0 until the stream is over */
int read_slbit(dsp_t *dsp, int *bit, int inv, int ofs, int pos, float l, int spike) {
    float s, s1;
    (void)dsp;
    if (seam_bit(inv, ofs, l, spike, pos, &s, &s1) == 0) { *bit = (s >= 0.f); return 0; }
    if (S.eof && S.qi >= S.qn) return EOF;
    *bit = 0;
    return 0;
}

/* The per-sample pull of the reference (demod_mod.c:722): none of its decoders calls it directly — they go through find_header() / read_softbit2p() — and a
 * batched engine has no "next sample" to hand out.  A caller that does call it is told so instead of reading a silent EOF. */
int f32buf_sample(dsp_t *dsp, int inv) {
    (void)dsp; (void)inv;
    fprintf(stderr, "demod_mod_hip: f32buf_sample() is not part of this seam (samples are consumed inside find_header / read_softbit2p); the decoders of the reference do not call it\n");
    exit(70);
}

/* ---------------------------------------------------------------- host-only helpers of demod_mod.c the decoders link against */

int read_wav_header(pcm_t *pcm, FILE *fp) {
    int sr = 0, bits = 0, nch = 0;
    if (wav_read_header(fp, &sr, &bits, &nch) < 0) return -1;
    pcm->sr = sr; pcm->bps = bits; pcm->nch = nch;
    if (pcm->sel_ch < 0 || pcm->sel_ch >= nch) pcm->sel_ch = 0;
    return 0;
}

int f32soft_read(FILE *fp, float *s, int inv) {
    float v;
    if (fread(&v, 4, 1, fp) != 1) return EOF;
    *s = inv ? -v : v;
    return 0;
}

/* share of header bits the last hdb->len bits agree with, signed by polarity (cmp_hdb, demod_mod.c:1639-1666) */
int find_binhead(FILE *fp, hdb_t *hdb, float *score) {
    const int n = hdb->len;
    int c;
    while ((c = fgetc(fp)) != EOF) {
        int e1 = 0;
        hdb->bufpos = (hdb->bufpos + 1) % n;
        hdb->buf[hdb->bufpos] = (char)(0x30 | (c & 1));
        for (int i = 0, j = hdb->bufpos; i < n; i++, j--) {
            if (j < 0) j = n - 1;
            if (hdb->buf[j] != hdb->hdr[n - 1 - i]) e1++;
        }
        const int e2 = n - e1;                       /* errors against the inverted header */
        const float mv = e2 < e1 ? (float)(-n + e2) / (float)n : (float)(n - e1) / (float)n;
        if (mv > hdb->thb || -mv > hdb->thb) { *score = mv; return 1; }
    }
    return EOF;
}

/* normalised correlation of the last hdb->len soft bits with the +-1 header (corr_softhdb, demod_mod.c:1692-1716) */
int find_softbinhead(FILE *fp, hdb_t *hdb, float *score, int inv) {
    const int n = hdb->len;
    float sbit;
    while (f32soft_read(fp, &sbit, inv) != EOF) {
        double sum = 0.0, nx = 0.0, ny = 0.0;
        hdb->bufpos = (hdb->bufpos + 1) % n;
        hdb->sbuf[hdb->bufpos] = sbit;
        for (int i = 0, j = hdb->bufpos + 1; i < n; i++, j++) {
            if (j >= n) j = 0;
            const float x = hdb->sbuf[j], y = (float)(2.0 * (hdb->hdr[i] & 1) - 1.0);
            sum += y * hdb->sbuf[j]; nx += x * x; ny += y * y;
        }
        sum /= sqrt(nx * ny);
        const float mv = (float)sum;
        if (mv > hdb->ths || -mv > hdb->ths) { *score = mv; return 1; }
    }
    return EOF;
}
