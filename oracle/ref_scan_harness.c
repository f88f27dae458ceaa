/*
 * oracle/ref_scan_harness.c — TEST INFRASTRUCTURE.  Our code around the *reference's own* scan/dft_detect.c,
 * which is compiled where it lies (oracle/Makefile: -I$(REF)/scan, nothing copied): the file is included with
 * its main() renamed so that the static functions of the scanner (init_buffers, f32buf_sample, getCorrDFT,
 * headcmp, frm_M10; dft_detect.c:357,737,866,932,995) can be driven over an in-memory capture and the values the
 * CLI never prints — score / position / dc of *every* template in *every* correlation window, and the four FM
 * streams — can be copied out for the parity tests of the GPU scanner.
 *
 * Flags as scan/Makefile:1,10 (-O3 -DNOC34C50 -DNOIMET1AB, -Ofast).
 */
#define _GNU_SOURCE
#include <stdio.h>
#define main dft_detect_main
#include "dft_detect.c"
#undef main

typedef struct {
    int    sr;        /* input sample rate                               */
    int    bps;       /* 8 / 16 / 32                                     */
    int    opt_iq;    /* 0 FM audio, 1 IF-rate IQ (--iq), 5 --IQ fq      */
    int    opt_dc;    /* --dc                                            */
    int    opt_min;   /* --min                                           */
    double fq;        /* --IQ argument                                   */
    double bw_khz;    /* --bw argument (0: three fixed bandwidths)       */
    int    nch;       /* channels of the FM-audio input (1 or 2)         */
} ref_scan_cfg_t;

/* consts[]: K, N_DFT, delay, M, sr_if, decM, lpFMtaps, lpIQtaps, Nrs(active), L[16..] */
int ref_scan_windows(const ref_scan_cfg_t *c, const void *raw, size_t nbytes, int max_win,
                     float *mv_out /*[w][16]*/, unsigned *mpos_out, int *mp_out, float *dc_out, int *herrs_out,
                     unsigned *m10_out, unsigned *pos_out /*[w]*/, int *consts /*[32]*/,
                     float *fm_tap /*[4][max_fm] or NULL*/, int max_fm)
{
    FILE *fp = fmemopen((void *)raw, nbytes, "rb");
    int j, k = 0, K, nw = 0;
    float mv[Nrs]; unsigned mpos[Nrs]; int mp[Nrs];
    if (!fp) return -1;

    option_iq = c->opt_iq; option_dc = c->opt_dc; option_min = c->opt_min; option_inv = 0;
    option_singleLpIQ = 0; set_lpIQ = (float)(c->bw_khz < 1.0 ? 0.0 : c->bw_khz * 1e3);
    dsp__xlt_fq = -c->fq;
    sample_rate = c->sr; bits_sample = c->bps; channels = c->opt_iq ? 2 : (c->nch > 0 ? c->nch : 1);
    wav_ch = 0; option_pcmraw = 1;
    lpIQ_bw[0] = 6e3; lpIQ_bw[1] = 12e3; lpIQ_bw[2] = 22e3; lpIQ_bw[3] = 200e3;
    dsp__decM = 1; dsp__sample_decX = 0; dsp__sample_decM = 0;

    K = init_buffers();
    if (K < 0) { fclose(fp); return -2; }
    consts[0] = K; consts[1] = N_DFT; consts[2] = (int)delay; consts[3] = M; consts[4] = sr_if; consts[5] = dsp__decM;
    consts[6] = dsp__lpFMtaps; consts[7] = dsp__lpIQtaps; consts[8] = idxIMETafsk + 1;
    for (j = 0; j <= idxIMETafsk; j++) consts[9 + j] = rs_hdr[j].L;
    for (j = 0; j < Nrs; j++) { mv[j] = 0; mpos[j] = 0; mp[j] = 0; }

    while (f32buf_sample(fp, 0) != EOF) {
        if (fm_tap && (int)(sample_in - 1) < max_fm)
            for (j = 0; j < N_bwIQ; j++) fm_tap[(size_t)j * max_fm + (sample_in - 1)] = buf_fm[j][(sample_in - 1) % M];
        k += 1;
        if (k < K - 4) continue;
        k = 0;
        if (nw >= max_win) break;
        pos_out[nw] = sample_out;
        for (j = 0; j <= idxIMETafsk; j++) {
            float *o_mv = mv_out + (size_t)nw * 16; unsigned *o_mp = mpos_out + (size_t)nw * 16;
            o_mv[j] = 0; o_mp[j] = mpos[j]; mp_out[(size_t)nw * 16 + j] = 0; dc_out[(size_t)nw * 16 + j] = 0;
            herrs_out[(size_t)nw * 16 + j] = -1; m10_out[(size_t)nw * 16 + j] = 0;
            if (j == idx_MTS01 || j == idx_C34C50 || j == idx_WXR301 || j == idx_WXRPN9 || j == idx_IMET1AB) continue;
            mv[j] = 0;
            mp[j] = getCorrDFT(K, 0, mv + j, mpos + j, rs_hdr + j);
            o_mv[j] = mv[j]; o_mp[j] = mpos[j]; mp_out[(size_t)nw * 16 + j] = mp[j]; dc_out[(size_t)nw * 16 + j] = rs_hdr[j].dc;
            if (mp[j] > 0 && (mv[j] > rs_hdr[j].thres || mv[j] < -rs_hdr[j].thres)) {
                herrs_out[(size_t)nw * 16 + j] = headcmp(1, mpos[j], mv[j] < 0, rs_hdr + j);
                if (strncmp(rs_hdr[j].type, "M10", 3) == 0 || strncmp(rs_hdr[j].type, "M20", 3) == 0)
                    m10_out[(size_t)nw * 16 + j] = frm_M10(mpos[j], mv[j] < 0, rs_hdr + j);
            }
        }
        nw++;
    }
    free_buffers();
    fclose(fp);
    return nw;
}
