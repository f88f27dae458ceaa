// sonde_softin_dev.hip — `rs41mod --softin [-i]` for MANY channels on the device, fed from the modem's soft decisions where they lie
// (include/sonde_fsk.h sonde_softin_dev_*): the consumer half of auto_rx's production pipe `fsk_demod ... | rs41mod --softin -i`
// (auto_rx/autorx/decode.py:901-909), so that BASELINE configs[3] ends at decoded, repaired frames without a device-to-host copy of the
// soft-decision stream.
//   find_softbinhead / corr_softhdb (demod/mod/demod_mod.c:1692-1762): the last 64 soft bits against the +-1 header, normalised, in double,
//     |mv| > 0.7 — one lane per bit position, 64 positions at a time; the polarity rule of rs41mod.c:2886-2890 (-i / --auto) on the first hit
//   the bit loop (rs41mod.c:2893-2962): 510 x 8 hard decisions, LSB first, de-whitened — one lane per byte
//   rs41_ecc (rs41mod.c:1703-1769): k_rs41_ecc_batch (sonde_kernels.hip), one workgroup per completed frame
// A wavefront per channel; the state that survives a call (search ring, frame in progress, pending bits of a byte, polarity) lives in
// device memory.  The ring is only advanced while SEARCHING — the reference's frame loop does not touch hdb.sbuf — so a search behind a
// frame starts from the ring as the header left it.  Arithmetic as the reference's (float products, double sums, in order): frames are
// bit-identical for identical soft bits (tests/test_gpu_softin_dev.py against oracle/_ref/fsk_demod | oracle/_ref/rs41mod --softin).
// (no contraction: the reference is plain C on x86-64 — every product and sum rounded on its own)
#pragma clang fp contract(off)
#include "../../include/sonde_fsk.h"
#include "sonde_fsk_dev.h"
#include "sonde_host.h"
#include "sonde_pinned.h"
#include <cstdio>
#include <cstring>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libsonde_hip: %s failed: %s\n", #x, hipGetErrorString(e_)); return SONDE_E_NOGPU; } } while (0)

struct SoftinChan {
    int   mode;                    // 0 searching, 1 inside a frame
    int   inv;                     // gpx.option.inv as it stands (--auto may flip it)
    int   body_done;               // frame bits consumed (0 .. 4080)
    int   carry_n;                 // soft bits of an unfinished byte
    float carry[8];
    float hist[64];                // the last 64 soft bits seen while searching, oldest first (hdb.sbuf)
    float mv;                      // score of the header in front of the frame in progress
    unsigned long long bits_in, hdr_bit;
    unsigned char frame[520];
};
struct SoftinMeta { int channel, len, nbytes; float mv; unsigned long long hdr_bit; };

struct SoftinArgs {
    const float *sd; long long ch_stride;          // soft decisions of channel c at sd + c * ch_stride
    const int *nbits_ch; int nbits;                // per channel (device) or one count for all
    const FskChan *fsk_chan; int bits_per_frame;   // or: frames of the modem's last launch x bits per frame
    const int *ch_list;                            // the channels of this launch (one workgroup each); null: all of them
    int n_ch, inv_in, opt_auto; float ths;
    SoftinChan *chan;
    unsigned char *frames; int *flen; SoftinMeta *meta; unsigned *count; int cap;
    const unsigned char *hdr;                      // 64 header bits ('0'/'1'), 8 header bytes, 64 mask bytes
};

__global__ __launch_bounds__(64)
void k_softin_rs41(const SoftinArgs a) {
    __shared__ float s_hist[64];
    __shared__ unsigned char s_frame[520];
    __shared__ float s_carry[8];
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    if (ch >= a.n_ch) return;
    SoftinChan *st = a.chan + ch;
    int nb = a.nbits;
    if (a.fsk_chan) { const int fr = a.fsk_chan[ch].frames; nb = fr > 0 ? fr * a.bits_per_frame : 0; }
    else if (a.nbits_ch) nb = a.nbits_ch[ch];
    const float *x = a.sd + (size_t)ch * a.ch_stride;
    int mode = st->mode, inv = st->inv, body_done = st->body_done, carry_n = st->carry_n;
    float mv_hdr = st->mv; unsigned long long hdr_bit = st->hdr_bit; const unsigned long long bits0 = st->bits_in;
    s_hist[lane] = st->hist[lane];
    for (int i = lane; i < 520; i += 64) s_frame[i] = st->frame[i];
    if (lane < 8) s_carry[lane] = st->carry[lane];
    // the header as +-1, one element per lane (for the dot products every lane needs all 64: from LDS-free registers via the constant array)
    __builtin_amdgcn_wave_barrier();
    const float sgn = a.inv_in ? -1.f : 1.f;
    int cur = 0;
    while (cur < nb) {
        if (mode == 0) {
            // ---- find_softbinhead: position q of the call = stream element 64 + (q - cur) of hist ++ x[cur ..]
            bool found = false;
            int base = cur;
            for (; base < nb && !found; base += 64) {
                const int q = base + lane;
                float mv = 0.f;
                if (q < nb) {
                    double sum = 0.0, normx = 0.0;
                    const int e = 64 + (q - cur);                       // window = elements e-63 .. e
                    for (int i = 0; i < 64; i++) {
                        const int k = e - 63 + i;
                        const float v = k < 64 ? s_hist[k] : sgn * x[cur + (k - 64)];
                        const float y = (a.hdr[i] & 1) ? 1.f : -1.f;
                        sum += (double)(y * v);
                        normx += (double)(v * v);
                    }
                    sum /= sqrt(normx * 64.0);
                    mv = (float)sum;
                }
                unsigned long long hits = __ballot(q < nb && fabsf(mv) > a.ths);
                while (hits) {
                    const int l = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const float mvl = __shfl(mv, l);
                    // the polarity rule (rs41mod.c:2886-2890): a header of the other sign is skipped, or with --auto flips the option
                    if ((double)mvl * (0.5 - inv) < 0) { if (!a.opt_auto) continue; inv ^= 1; }
                    found = true;
                    const int qs = base + l;
                    // the ring as the header leaves it: the 64 elements up to the hit
                    const int e = 64 + (qs - cur), k = e - 63 + lane;
                    const float v = k < 64 ? s_hist[k] : sgn * x[cur + (k - 64)];
                    __builtin_amdgcn_wave_barrier();
                    s_hist[lane] = v;
                    __builtin_amdgcn_wave_barrier();
                    mode = 1; body_done = 0; carry_n = 0; mv_hdr = mvl; hdr_bit = bits0 + (unsigned long long)qs + 1ull;
                    cur = qs + 1;
                    break;
                }
            }
            if (!found) {
                const int e = 64 + (nb - 1 - cur), k = e - 63 + lane;
                const float v = k < 64 ? s_hist[k] : sgn * x[cur + (k - 64)];
                __builtin_amdgcn_wave_barrier();
                s_hist[lane] = v;
                __builtin_amdgcn_wave_barrier();
                cur = nb;
            }
        } else {
            // ---- the bit loop: bytes from the pending bits of the last call and the new ones
            const int take = min(nb - cur, 4080 - body_done);
            const int tot = carry_n + take, nbytes = tot / 8, byte0 = 8 + (body_done - carry_n) / 8;
            for (int j = lane; j < nbytes; j += 64) {
                unsigned byte = 0;
                for (int b = 0; b < 8; b++) {
                    const int k = 8 * j + b;
                    const float v = k < carry_n ? s_carry[k] : sgn * x[cur + (k - carry_n)];
                    int bit = v >= 0.0f;
                    if (inv) bit ^= 1;
                    byte |= (unsigned)bit << b;                         // bits2byte: LSB first (rs41mod.c:224)
                }
                const int bc = byte0 + j;
                s_frame[bc] = (unsigned char)(byte ^ a.hdr[72 + (bc & 63)]);
            }
            __builtin_amdgcn_wave_barrier();
            const int rest = tot - 8 * nbytes;
            float cv = 0.f;
            if (lane < rest) { const int k = 8 * nbytes + lane; cv = k < carry_n ? s_carry[k] : sgn * x[cur + (k - carry_n)]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < 8) s_carry[lane] = cv;
            __builtin_amdgcn_wave_barrier();
            carry_n = rest; body_done += take; cur += take;
            if (body_done == 4080) {
                // print_frame(): the frame goes out (header bytes in front), rs41_ecc() follows in k_rs41_ecc_batch
                unsigned slot = 0;
                if (lane == 0) slot = atomicAdd(a.count, 1u);
                slot = __shfl(slot, 0);
                if ((int)slot < a.cap) {
                    unsigned char *o = a.frames + (size_t)slot * 518;
                    for (int i = lane; i < 518; i += 64) o[i] = i < 8 ? a.hdr[64 + i] : s_frame[i];
                    if (lane == 0) {
                        int ft = 0; const unsigned char b = s_frame[0x38];
                        for (int i = 0; i < 4; i++) ft += ((b >> i) & 1) - ((b >> (i + 4)) & 1);        // frametype (rs41mod.c:407-415)
                        a.flen[slot] = ft >= 0 ? 320 : 518;
                        SoftinMeta m; m.channel = ch; m.len = ft >= 0 ? 320 : 518; m.nbytes = 518; m.mv = mv_hdr; m.hdr_bit = hdr_bit;
                        a.meta[slot] = m;
                    }
                }
                mode = 0; body_done = 0; carry_n = 0;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    st->hist[lane] = s_hist[lane];
    for (int i = lane; i < 520; i += 64) st->frame[i] = s_frame[i];
    if (lane < 8) st->carry[lane] = s_carry[lane];
    if (lane == 0) { st->mode = mode; st->inv = inv; st->body_done = body_done; st->carry_n = carry_n; st->mv = mv_hdr; st->hdr_bit = hdr_bit; st->bits_in = bits0 + (unsigned long long)nb; }
}


// ------------------------------------------------------------------------------------------------
// DFM09: dfm09mod --softin (dfm09mod.c:1604-1720) — 32 raw header symbols at 0.7, then two soft symbols per bit (s2 - s1), 8 frames of 280 bits per hit
// (the first one starts behind its 16 header bits), de-interleave + Hamming(8,4) with the soft 2-bit pass of --ecc2 (:231-345) on a lane per codeword
// ------------------------------------------------------------------------------------------------
struct SoftinDfmChan {
    int   mode, inv, dpos, dfrm, dhalf;
    float ds1;
    unsigned hdrcnt;
    float hist[32];
    float mv;
    unsigned long long bits_in, hdr_bit;
    unsigned char dhb[280];
    float dsf[280];
};
// one 8-bit codeword (dfm09mod.c:240-307): 0 clean, j + 1 = bit j fixed, -1 uncorrectable (level 2: the codeword at distance 2 that correlates best with the soft bits)
__device__ __forceinline__ void dfm_dev_codeword(int n, unsigned char *c) {
    const unsigned char d0 = (n >> 3) & 1, d1 = (n >> 2) & 1, d2 = (n >> 1) & 1, d3 = n & 1;
    c[0] = d0; c[1] = d1; c[2] = d2; c[3] = d3; c[4] = d1 ^ d2 ^ d3; c[5] = d0 ^ d2 ^ d3; c[6] = d0 ^ d1 ^ d3; c[7] = d0 ^ d1 ^ d2;
}
__device__ int dfm_dev_check(int level, unsigned char hb[8], const float sb[8]) {
    const unsigned char Hm[4][8] = { {0,1,1,1,1,0,0,0}, {1,0,1,1,0,1,0,0}, {1,1,0,1,0,0,1,0}, {1,1,1,0,0,0,0,1} };
    const unsigned char He[8] = { 0x7, 0xB, 0xD, 0xE, 0x8, 0x4, 0x2, 0x1 };
    unsigned syn = 0;
    for (int i = 0; i < 4; i++) { unsigned char s = 0; for (int j = 0; j < 8; j++) s ^= Hm[i][j] & hb[j]; syn = (syn << 1) | s; }
    if (!syn) return 0;
    for (int j = 0; j < 8; j++) if (syn == He[j]) { hb[j] ^= 1; return j + 1; }
    if (level == 2) {
        int best = -1; float bestsum = 0.0f;
        for (int n = 0; n < 16; n++) {
            unsigned char c[8]; int d = 0;
            dfm_dev_codeword(n, c);
            for (int i = 0; i < 8; i++) d += (hb[i] != c[i]);
            if (d != 2) continue;
            float sum = 0.0f;
            for (int i = 0; i < 8; i++) sum += (2 * c[i] - 1) * sb[i];
            if (sum >= bestsum) { bestsum = sum; best = n; }
        }
        if (best >= 0) dfm_dev_codeword(best, hb);
    }
    return -1;
}

struct SoftinDfmArgs {
    SoftinArgs base;                       // sd / counts / n_ch / inv_in / opt_auto / ths / count / cap / hdr (32 raw header symbols)
    SoftinDfmChan *chan;
    sonde_dfm_frame_t *out;
    int ecc_level;
};

__global__ __launch_bounds__(64)
void k_softin_dfm(const SoftinDfmArgs A) {
    const SoftinArgs &a = A.base;
    __shared__ float s_hist[32];
    __shared__ unsigned char s_hb[280];
    __shared__ float s_sf[280];
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    if (ch >= a.n_ch) return;
    SoftinDfmChan *st = A.chan + ch;
    int nb = a.nbits;
    if (a.fsk_chan) { const int fr = a.fsk_chan[ch].frames; nb = fr > 0 ? fr * a.bits_per_frame : 0; }
    else if (a.nbits_ch) nb = a.nbits_ch[ch];
    const float *x = a.sd + (size_t)ch * a.ch_stride;
    int mode = st->mode, inv = st->inv, dpos = st->dpos, dfrm = st->dfrm, dhalf = st->dhalf;
    float ds1 = st->ds1, mv_hdr = st->mv; unsigned hdrcnt = st->hdrcnt; unsigned long long hdr_bit = st->hdr_bit; const unsigned long long bits0 = st->bits_in;
    if (lane < 32) s_hist[lane] = st->hist[lane];
    for (int i = lane; i < 280; i += 64) { s_hb[i] = st->dhb[i]; s_sf[i] = st->dsf[i]; }
    __builtin_amdgcn_wave_barrier();
    const float sgn = a.inv_in ? -1.f : 1.f;
    int cur = 0;
    while (cur < nb) {
        if (mode == 0) {
            bool found = false;
            for (int base = cur; base < nb && !found; base += 64) {
                const int q = base + lane;
                float mv = 0.f;
                if (q < nb) {
                    double sum = 0.0, normx = 0.0;
                    const int e = 32 + (q - cur);
                    for (int i = 0; i < 32; i++) {
                        const int k = e - 31 + i;
                        const float v = k < 32 ? s_hist[k] : sgn * x[cur + (k - 32)];
                        const float y = (a.hdr[i] & 1) ? 1.f : -1.f;
                        sum += (double)(y * v);
                        normx += (double)(v * v);
                    }
                    sum /= sqrt(normx * 32.0);
                    mv = (float)sum;
                }
                unsigned long long hits = __ballot(q < nb && fabsf(mv) > a.ths);
                while (hits) {
                    const int l = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const float mvl = __shfl(mv, l);
                    hdrcnt += 8;                                        // every header seen counts (dfm09mod.c:1628,1632), accepted or not
                    if ((double)mvl * (0.5 - inv) < 0) { if (!a.opt_auto) continue; inv ^= 1; }
                    found = true;
                    const int qs = base + l, e = 32 + (qs - cur), k = e - 31 + (lane & 31);
                    const float v = k < 32 ? s_hist[k] : sgn * x[cur + (k - 32)];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 32) s_hist[lane] = v;
                    __builtin_amdgcn_wave_barrier();
                    mode = 1; dpos = 16; dfrm = 0; dhalf = 0; mv_hdr = mvl; hdr_bit = bits0 + (unsigned long long)qs + 1ull;
                    cur = qs + 1;
                    break;
                }
            }
            if (!found) {
                const int e = 32 + (nb - 1 - cur), k = e - 31 + (lane & 31);
                const float v = k < 32 ? s_hist[k] : sgn * x[cur + (k - 32)];
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) s_hist[lane] = v;
                __builtin_amdgcn_wave_barrier();
                cur = nb;
            }
        } else {
            // ---- frame bits: two symbols each (the first of a pair may be left over from the last call)
            const int need = 280 - dpos, take = min(nb - cur, 2 * need - dhalf), nbits = (dhalf + take) / 2;
            for (int j = lane; j < nbits; j += 64) {
                const float s1 = (j == 0 && dhalf) ? ds1 : sgn * x[cur + 2 * j - dhalf], s2 = sgn * x[cur + 2 * j + 1 - dhalf];
                float v = s2 - s1;                                      // integrate both Manchester symbols (dfm09mod.c:1684)
                int hb = v >= 0.0f;
                if (inv) { hb ^= 1; v = -v; }
                s_hb[dpos + j] = (unsigned char)hb; s_sf[dpos + j] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if ((dhalf + take) & 1) { ds1 = sgn * x[cur + take - 1]; dhalf = 1; } else dhalf = 0;
            dpos += nbits; cur += take;
            if (dpos == 280) {
                unsigned slot = 0;
                if (lane == 0) slot = atomicAdd(a.count, 1u);
                slot = __shfl(slot, 0);
                // de-interleave + Hamming: lanes 0..6 conf (block at bit 16, 7 codewords), 7..19 dat1 (72, 13), 20..32 dat2 (176, 13)
                int e = 0; unsigned char nib = 0;
                const int blk = lane < 7 ? 0 : lane < 20 ? 1 : 2, i = lane < 7 ? lane : lane < 20 ? lane - 7 : lane - 20;
                const int L = blk == 0 ? 7 : 13, off = blk == 0 ? 16 : blk == 1 ? 72 : 176;
                if (lane < 33) {
                    unsigned char c[8]; float sb[8];
                    for (int j = 0; j < 8; j++) { c[j] = s_hb[off + L * j + i]; sb[j] = s_sf[off + L * j + i]; }
                    if (A.ecc_level) e = dfm_dev_check(A.ecc_level, c, sb);
                    nib = (unsigned char)((c[0] << 3) | (c[1] << 2) | (c[2] << 1) | c[3]);
                }
                const unsigned long long fixed = __ballot(lane < 33 && e > 0), bad = __ballot(lane < 33 && e < 0);
                if ((int)slot < a.cap) {
                    sonde_dfm_frame_t *o = A.out + slot;
                    if (lane < 7) o->conf[i] = nib; else if (lane < 20) o->dat1[i] = nib; else if (lane < 33) o->dat2[i] = nib;
                    if (lane < 35) { unsigned char rb = 0; for (int b = 0; b < 8; b++) rb |= (unsigned char)((s_hb[8 * lane + b] & 1) << b); o->rawbits[lane] = rb; }
                    if (lane == 0) {
                        o->channel = ch; o->frame_in_hit = dfrm; o->mv = mv_hdr; o->mv_pos = (uint32_t)hdr_bit;
                        o->frm_count = (float)(hdrcnt + (unsigned)dfrm); o->inv = inv; o->pad[0] = o->pad[1] = o->pad[2] = 0; o->pad2 = 0;
                        // hamming()'s return per block: a bit per fixed codeword, or'ed with -1 for an uncorrectable one
                        o->ecc[0] = ((bad & 0x7Full) ? -1 : 0) | (int)(fixed & 0x7Full);
                        o->ecc[1] = (((bad >> 7) & 0x1FFFull) ? -1 : 0) | (int)((fixed >> 7) & 0x1FFFull);
                        o->ecc[2] = (((bad >> 20) & 0x1FFFull) ? -1 : 0) | (int)((fixed >> 20) & 0x1FFFull);
                    }
                }
                dpos = 0;
                if (++dfrm == 8) mode = 0;                              // nfrms frames per header hit, then search again (:1656,1718)
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) st->hist[lane] = s_hist[lane];
    for (int i = lane; i < 280; i += 64) { st->dhb[i] = s_hb[i]; st->dsf[i] = s_sf[i]; }
    if (lane == 0) { st->mode = mode; st->inv = inv; st->dpos = dpos; st->dfrm = dfrm; st->dhalf = dhalf; st->ds1 = ds1; st->hdrcnt = hdrcnt; st->mv = mv_hdr; st->hdr_bit = hdr_bit; st->bits_in = bits0 + (unsigned long long)nb; }
}

// ------------------------------------------------------------------------------------------------
// M10: m10mod --softin (m10mod.c:1405-1510) — 32 raw header symbols at 0.8 in either polarity, two soft symbols per bit (s2 - s1), differential decoding,
// 968 bits, then one symbol per counted bit dropped up to 5 x 808; the frame checksum (m10mod.c:594-628) on the device
// ------------------------------------------------------------------------------------------------
struct SoftinM10Chan {
    int   mode, inv, mpos, mhalf, mbit0, mskip;
    float ms1;
    float hist[32];
    float mv;
    unsigned long long bits_in, hdr_bit;
    char  mbits[976];
};
struct SoftinM10Args { SoftinArgs base; SoftinM10Chan *chan; sonde_m10_frame_t *out; int stage_cap; };
#define M10_STAGE_MAX 12288            // soft decisions of a call the kernel keeps in LDS (48 KB); longer calls read them from global memory

// The header search (find_softbinhead / corr_softhdb, demod_mod.c:1692-1762) evaluates at EVERY symbol the normalised correlation of the last 32 with the header, in
// double.  Round 5 did exactly that on a lane per position — 64 dependent double adds and 32 global loads each, 0.37 ms for 341 channels.  Now the call's soft decisions
// are staged in LDS once and every position gets the same quantity in float first (error < 1e-5); only a position whose float value comes within 1e-3 of the
// threshold — a header, a few per second — is evaluated again the reference's way (same operands, same order, double), and that value decides and is recorded.
__global__ __launch_bounds__(64)
void k_softin_m10(const SoftinM10Args A) {
    const SoftinArgs &a = A.base;
    extern __shared__ float s_x[];                     // [stage_cap] sgn * x of this call (staged: nb <= stage_cap)
    __shared__ float s_hist[32];
    __shared__ char s_mb[976];
    __shared__ unsigned char s_fr[124];
    const int ch = a.ch_list ? a.ch_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    if (ch >= a.n_ch) return;
    SoftinM10Chan *st = A.chan + ch;
    int nb = a.nbits;
    if (a.fsk_chan) { const int fr = a.fsk_chan[ch].frames; nb = fr > 0 ? fr * a.bits_per_frame : 0; }
    else if (a.nbits_ch) nb = a.nbits_ch[ch];
    const float *x = a.sd + (size_t)ch * a.ch_stride;
    constexpr int NBITS = 121 * 8;
    int mode = st->mode, inv = st->inv, mpos = st->mpos, mhalf = st->mhalf, mbit0 = st->mbit0, mskip = st->mskip;
    float ms1 = st->ms1, mv_hdr = st->mv; unsigned long long hdr_bit = st->hdr_bit; const unsigned long long bits0 = st->bits_in;
    if (lane < 32) s_hist[lane] = st->hist[lane];
    for (int i = lane; i < 976; i += 64) s_mb[i] = st->mbits[i];
    const float sgn = a.inv_in ? -1.f : 1.f;
    const bool staged = nb <= A.stage_cap;
    if (staged) for (int i = lane; i < nb; i += 64) s_x[i] = sgn * x[i];
    __builtin_amdgcn_wave_barrier();
    auto X = [&](int p) -> float { return staged ? s_x[p] : sgn * x[p]; };       // soft decision p of this call, in the polarity in effect
    unsigned hbits = 0;                                                          // the 32 header symbols as a bit mask
    for (int i = 0; i < 32; i++) hbits |= (unsigned)(a.hdr[i] & 1) << i;
    int cur = 0;
    while (cur < nb) {
        if (mode == 0) {
            bool found = false;
            for (int base = cur; base < nb && !found; base += 64) {
                const int q = base + lane;
                float mv = 0.f;
                if (q < nb) {
                    const int e = 32 + (q - cur);
                    float fs = 0.f, fn = 0.f;
                    for (int i = 0; i < 32; i++) {
                        const int k = e - 31 + i;
                        const float v = k < 32 ? s_hist[k] : X(cur + (k - 32));
                        fs += ((hbits >> i) & 1u) ? v : -v;
                        fn = fmaf(v, v, fn);
                    }
                    mv = fs * __builtin_amdgcn_rsqf(fn * 32.0f);
                    if (!(fabsf(mv) < a.ths - 1e-3f)) {                   // (also NaN: an all-zero window is the reference's 0 / 0)
                        double sum = 0.0, normx = 0.0;
                        for (int i = 0; i < 32; i++) {
                            const int k = e - 31 + i;
                            const float v = k < 32 ? s_hist[k] : X(cur + (k - 32));
                            const float y = ((hbits >> i) & 1u) ? 1.f : -1.f;
                            sum += (double)(y * v);
                            normx += (double)(v * v);
                        }
                        sum /= sqrt(normx * 32.0);
                        mv = (float)sum;
                    }
                }
                const unsigned long long hits = __ballot(q < nb && fabsf(mv) > a.ths);
                if (hits) {
                    const int l = __builtin_ctzll(hits);
                    const float mvl = __shfl(mv, l);
                    if ((double)mvl * (0.5 - inv) < 0) inv ^= 1;            // irrelevant for the differential code (m10mod.c:1447)
                    found = true;
                    const int qs = base + l, e = 32 + (qs - cur), k = e - 31 + (lane & 31);
                    const float v = k < 32 ? s_hist[k] : X(cur + (k - 32));
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 32) s_hist[lane] = v;
                    __builtin_amdgcn_wave_barrier();
                    mode = 1; mpos = 0; mhalf = 0; mbit0 = '0'; mv_hdr = mvl; hdr_bit = bits0 + (unsigned long long)qs + 1ull;
                    cur = qs + 1;
                }
            }
            if (!found) {
                const int e = 32 + (nb - 1 - cur), k = e - 31 + (lane & 31);
                const float v = k < 32 ? s_hist[k] : X(cur + (k - 32));
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) s_hist[lane] = v;
                __builtin_amdgcn_wave_barrier();
                cur = nb;
            }
        } else if (mode == 1) {
            const int need = NBITS - mpos, take = min(nb - cur, 2 * need - mhalf), nbits = (mhalf + take) / 2;
            // bit j of this call and the one before it (the differential code): out = 0x31 ^ (previous ^ bit); the previous of the frame's first bit is '0' (0x30),
            // which leaves that character outside '0' / '1' — the reference's own quirk, kept
            int last_bit = mbit0;
            for (int j0 = 0; j0 < nbits; j0 += 64) {
                const int j = j0 + lane;
                int bit = 0, prev = 0;
                if (j < nbits) {
                    const float s1 = (j == 0 && mhalf) ? ms1 : X(cur + 2 * j - mhalf), s2 = X(cur + 2 * j + 1 - mhalf);
                    bit = (s2 - s1) >= 0.0f;
                }
                prev = __shfl_up(bit, 1);
                if (lane == 0) prev = last_bit;
                if (j < nbits) s_mb[mpos + j] = (char)(0x31 ^ (prev ^ bit));
                const int cnt = min(64, nbits - j0);
                last_bit = __shfl(bit, cnt - 1);
            }
            __builtin_amdgcn_wave_barrier();
            if (nbits > 0) mbit0 = last_bit;
            if ((mhalf + take) & 1) { ms1 = X(cur + take - 1); mhalf = 1; } else mhalf = 0;
            mpos += nbits; cur += take;
            if (mpos == NBITS) {
                unsigned slot = 0;
                if (lane == 0) slot = atomicAdd(a.count, 1u);
                slot = __shfl(slot, 0);
                for (int i = lane; i < 124; i += 64) {
                    unsigned v = 0;
                    if (i < 121) for (int k = 0; k < 8; k++) if (s_mb[8 * i + 7 - k] == '1') v |= 1u << k;
                    s_fr[i] = (unsigned char)v;
                }
                __builtin_amdgcn_wave_barrier();
                if ((int)slot < a.cap) {
                    sonde_m10_frame_t *o = A.out + slot;
                    for (int i = lane; i < 124; i += 64) o->frame[i] = s_fr[i];
                    if (lane == 0) {
                        int aux = s_fr[0] - 0x64;
                        if (aux < 0 || aux > 20) aux = 0;
                        int c = 0;
                        for (int i = 0; i < 99 + aux; i++) {                 // checkM10 (m10mod.c:594-628)
                            unsigned char b = s_fr[i];
                            b = (unsigned char)((b >> 1) | ((b & 1) << 7));
                            b ^= (b >> 2) & 0xFF;
                            const int t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1), t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1);
                            const int t = (c & 0x3F) | (t6 << 6) | (t7 << 7);
                            int sreg = (c >> 7) & 0xFF;
                            sreg ^= (sreg >> 2) & 0xFF;
                            c = (((c & 0xFF) << 8) | ((b ^ t ^ sreg) & 0xFF)) & 0xFFFF;
                        }
                        o->channel = ch; o->nbits = NBITS; o->len = 101 + aux; o->cs_calc = (uint32_t)c;
                        o->cs_ok = ((uint32_t)((s_fr[99 + aux] << 8) | s_fr[100 + aux]) == (uint32_t)c);
                        o->mv = mv_hdr; o->mv_pos = (uint32_t)hdr_bit;
                    }
                }
                mode = 2; mskip = NBITS;
            }
        } else {
            // the rest of the second: one symbol per counted bit up to 5 x 808 (m10mod.c:1494-1506)
            const int take = min(nb - cur, 5 * 808 - mskip);
            mskip += take; cur += take;
            if (mskip >= 5 * 808) mode = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) st->hist[lane] = s_hist[lane];
    for (int i = lane; i < 976; i += 64) st->mbits[i] = s_mb[i];
    if (lane == 0) { st->mode = mode; st->inv = inv; st->mpos = mpos; st->mhalf = mhalf; st->mbit0 = mbit0; st->mskip = mskip; st->ms1 = ms1; st->mv = mv_hdr; st->hdr_bit = hdr_bit; st->bits_in = bits0 + (unsigned long long)nb; }
}

// ------------------------------------------------------------------------------------------------
// The same block codes for the BASE-RATE engine's hits (sonde_engine.cpp): what sonde_engine_fetch_dfm / _m10 used to do per frame on one host thread
// now runs over the records a process call has queued, behind its frame-sync kernel on the same stream.  Only decoded frames come to the host (DFM: 8 x 104 bytes
// a hit instead of 2224 soft bits).
//   DFM (dfm09mod.c:1652-1717, :231-345): a hit = frame 0 (bits 16..279, the header went to the correlator) + 7 x 280 bits; de-interleave, Hamming(8,4) per
//        codeword on a lane (hard, or --ecc2's soft 2-bit pass), block status by ballot
//   M10 (m10mod.c:1484, :141-166, :594-628): differential decoding into the channel's persistent bit characters, bits2bytes, checkM10
// Records [*done, *fcount) of the ring; the last workgroup to finish moves *done up (done[1] is its ticket counter).
// ------------------------------------------------------------------------------------------------
#include "sonde_dev.h"
__global__ __launch_bounds__(64)
void k_dfm_hits(const FrameRec *frames, const float *soft, const int nbits, const int max_frames, const unsigned *fcount, unsigned *done, sonde_dfm_frame_t *out, const int ecc_level) {
    __shared__ unsigned char s_hb[280];
    __shared__ float s_sf[280];
    const int lane = threadIdx.x;
    const unsigned end = *fcount; unsigned start = done[0];
    if ((int)(end - start) > max_frames) start = end - (unsigned)max_frames;       // (what the ring no longer holds)
    for (unsigned i = start + blockIdx.x; (int)(end - i) > 0; i += gridDim.x) {
        const unsigned idx = i % (unsigned)max_frames;
        const FrameRec &r = frames[idx];
        const float *sb = soft + (size_t)idx * nbits;
        for (int f = 0; f < 8; f++) {
            const int first = f == 0 ? 0 : 264 + 280 * (f - 1), skip = f == 0 ? 16 : 0;
            sonde_dfm_frame_t *o = out + (size_t)idx * 8 + f;
            if (first + (280 - skip) > r.nbytes) { if (lane == 0) o->frame_in_hit = -1; continue; }       // nbytes = valid bits of the hit; a partial frame is dropped
            for (int k = lane; k < 280; k += 64) {
                const int b = first + k - skip;
                s_hb[k] = k < skip ? 0 : (unsigned char)((r.frame[b >> 3] >> (b & 7)) & 1);
                s_sf[k] = k < skip ? 0.f : sb[b];
            }
            __builtin_amdgcn_wave_barrier();
            int e = 0; unsigned char nib = 0;
            const int blk = lane < 7 ? 0 : lane < 20 ? 1 : 2, ci = lane < 7 ? lane : lane < 20 ? lane - 7 : lane - 20;
            const int L = blk == 0 ? 7 : 13, off = blk == 0 ? 16 : blk == 1 ? 72 : 176;
            if (lane < 33) {
                unsigned char c[8]; float sv[8];
                for (int j = 0; j < 8; j++) { c[j] = s_hb[off + L * j + ci]; sv[j] = s_sf[off + L * j + ci]; }
                if (ecc_level) e = dfm_dev_check(ecc_level, c, sv);
                nib = (unsigned char)((c[0] << 3) | (c[1] << 2) | (c[2] << 1) | c[3]);
            }
            const unsigned long long fixed = __ballot(lane < 33 && e > 0), bad = __ballot(lane < 33 && e < 0);
            if (lane < 7) o->conf[ci] = nib; else if (lane < 20) o->dat1[ci] = nib; else if (lane < 33) o->dat2[ci] = nib;
            if (lane < 35) { unsigned char rb = 0; for (int b = 0; b < 8; b++) rb |= (unsigned char)((s_hb[8 * lane + b] & 1) << b); o->rawbits[lane] = rb; }
            if (lane == 0) {
                o->channel = r.channel; o->frame_in_hit = f; o->mv = r.mv; o->mv_pos = r.mv_pos; o->frm_count = 0.f;     // (positions relative to the channel's start: the host, which knows it)
                o->inv = r.mv < 0.f; o->pad[0] = o->pad[1] = o->pad[2] = 0; o->pad2 = 0;
                o->ecc[0] = ((bad & 0x7Full) ? -1 : 0) | (int)(fixed & 0x7Full);
                o->ecc[1] = (((bad >> 7) & 0x1FFFull) ? -1 : 0) | (int)((fixed >> 7) & 0x1FFFull);
                o->ecc[2] = (((bad >> 20) & 0x1FFFull) ? -1 : 0) | (int)((fixed >> 20) & 0x1FFFull);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __threadfence();
    if (lane == 0 && atomicAdd(&done[1], 1u) == gridDim.x - 1) { done[1] = 0; done[0] = end; }
}

__global__ __launch_bounds__(64)
void k_m10_hits(const FrameRec *frames, const float *soft, const float *soft1, const int chk3, const int nbits, const int max_frames, const unsigned *fcount, unsigned *done,
                char *chan_bits, sonde_m10_frame_t *out) {
    constexpr int NBYTES = 101 + 20, NB = NBYTES * 8;
    __shared__ unsigned char s_fr[124];
    const int lane = threadIdx.x;
    const unsigned end = *fcount; unsigned start = done[0];
    if ((int)(end - start) > max_frames) start = end - (unsigned)max_frames;
    for (unsigned i = start; (int)(end - i) > 0; i++) {
        const unsigned idx = i % (unsigned)max_frames;
        const FrameRec &r = frames[idx];
        if ((unsigned)r.channel % gridDim.x != blockIdx.x) continue;          // a channel's records in order, by one workgroup: its bit characters persist from frame to frame
        char *fb = chan_bits + (size_t)r.channel * (NB + 8);
        const int nv = r.nbytes < NB ? r.nbytes : NB;                          // nbytes = valid bits of the hit
        const float *sb = soft + (size_t)idx * nbits, *sb1 = soft1 ? soft1 + (size_t)idx * nbits : nullptr;
        auto bit_at = [&](const int p) -> int {
            // --chk3 (m10mod.c:1476-1479): the bit from both soft values of read_softbit2p, (sb + 0.25 sb1) >= 0
            if (chk3 && sb1) return ((double)sb[p] + 0.25 * (double)sb1[p]) >= 0.0;
            return (r.frame[p >> 3] >> (p & 7)) & 1;
        };
        for (int p = lane; p < nv; p += 64) {
            const int bit = bit_at(p), prev = p == 0 ? 0x30 : bit_at(p - 1);  // differential decoding: 1 = same as the previous bit; the first against '0' (m10mod.c:1484)
            fb[p] = (char)(0x31 ^ (prev ^ bit));
        }
        if (lane == 0) fb[nv] = 0;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int k = lane; k < 124; k += 64) {
            unsigned v = 0;
            if (k < NBYTES) for (int q = 0; q < 8; q++) if (fb[8 * k + 7 - q] == '1') v |= 1u << q;       // bits2bytes, big endian; anything but '1' counts as 0
            s_fr[k] = (unsigned char)v;
        }
        __builtin_amdgcn_wave_barrier();
        sonde_m10_frame_t *o = out + idx;
        for (int k = lane; k < 124; k += 64) o->frame[k] = s_fr[k];
        if (lane == 0) {
            int aux = s_fr[0] - 0x64;
            if (aux < 0 || aux > 20) aux = 0;
            int c = 0;
            for (int k = 0; k < 99 + aux; k++) {                            // checkM10 (m10mod.c:594-628)
                unsigned char b = s_fr[k];
                b = (unsigned char)((b >> 1) | ((b & 1) << 7));
                b ^= (b >> 2) & 0xFF;
                const int t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1), t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1);
                const int t = (c & 0x3F) | (t6 << 6) | (t7 << 7);
                int sreg = (c >> 7) & 0xFF;
                sreg ^= (sreg >> 2) & 0xFF;
                c = (((c & 0xFF) << 8) | ((b ^ t ^ sreg) & 0xFF)) & 0xFFFF;
            }
            o->channel = r.channel; o->nbits = nv; o->len = 101 + aux; o->cs_calc = (uint32_t)c;
            o->cs_ok = ((uint32_t)((s_fr[99 + aux] << 8) | s_fr[100 + aux]) == (uint32_t)c);
            o->mv = r.mv; o->mv_pos = r.mv_pos;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence();
    if (lane == 0 && atomicAdd(&done[1], 1u) == gridDim.x - 1) { done[1] = 0; done[0] = end; }
}

extern "C" void sonde_launch_dfm_hits(const FrameRec *frames, const float *soft, int nbits, int max_frames, const unsigned *fcount, unsigned *done, sonde_dfm_frame_t *out,
                                      int ecc_level, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_dfm_hits, dim3(grid), dim3(64), 0, s, frames, soft, nbits, max_frames, fcount, done, out, ecc_level);
}
extern "C" void sonde_launch_m10_hits(const FrameRec *frames, const float *soft, const float *soft1, int chk3, int nbits, int max_frames, const unsigned *fcount, unsigned *done,
                                      char *chan_bits, sonde_m10_frame_t *out, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_m10_hits, dim3(grid), dim3(64), 0, s, frames, soft, soft1, chk3, nbits, max_frames, fcount, done, chan_bits, out);
}

extern "C" void sonde_launch_rs41_ecc_batch_n(uint8_t *frames, const int32_t *flen, const unsigned *count, int cap, int level, int32_t *ecc, int32_t *codes, uint8_t *synd,
                                              const uint8_t *gf_exp, const uint8_t *gf_log, hipStream_t s);
extern "C" int sonde_fsk_wait(sonde_fsk_t *f);
extern "C" int sonde_fsk_host_frames(sonde_fsk_t *f, int32_t *out);
extern "C" int sonde_fsk_dev_view(sonde_fsk_t *f, const float **d_sd, long long *sd_cap, const FskChan **d_chan, int *bits_per_frame, int *n_ch, hipStream_t *stream);
extern "C" int sonde_fsk_dev_reader_done(sonde_fsk_t *f, hipStream_t consumer_stream);
extern "C" int sonde_fsk_dev_view_prev(sonde_fsk_t *f, const float **d_sd, long long *sd_cap, int *bits_per_frame, int *n_ch, int32_t *frames_out);
extern "C" int sonde_fsk_dev_reader_done_prev(sonde_fsk_t *f, hipStream_t consumer_stream);

struct sonde_softin_dev {
    int C = 0, ecc_level = 0, cap = 0, type = SONDE_RS41;
    SoftinDfmChan *d_dfm_chan = nullptr; sonde_dfm_frame_t *d_dfm_out = nullptr; std::vector<sonde_dfm_frame_t> qdfm; Pinned<sonde_dfm_frame_t> h_dfm;
    SoftinM10Chan *d_m10_chan = nullptr; sonde_m10_frame_t *d_m10_out = nullptr; std::vector<sonde_m10_frame_t> qm10; Pinned<sonde_m10_frame_t> h_m10;
    SoftinArgs args{};
    hipStream_t stream = nullptr; bool own_stream = false;
    SoftinChan *d_chan = nullptr; unsigned char *d_frames = nullptr, *d_hdr = nullptr, *d_gf = nullptr, *d_synd = nullptr;
    int *d_flen = nullptr, *d_ecc = nullptr, *d_codes = nullptr; SoftinMeta *d_meta = nullptr; unsigned *d_count = nullptr;
    std::vector<sonde_frame_t> queue;
    long long frames_total = 0, ecc_ok_total = 0, repaired_total = 0, symbols_total = 0, dropped = 0;
    Pinned<int> h_ecc; Pinned<SoftinMeta> h_meta; Pinned<unsigned char> h_frames;       // page-locked landing buffers of a call's records (sonde_pinned.h)
    unsigned *h_count = nullptr;                   // pinned: the counters of a call's two passes
    int head = 0;                                  // records copied to the host without asking how many there are (what a call of a second normally completes)
    bool pending = false; hipStream_t pend_stream = nullptr;
    int *h_nbits = nullptr, *d_nbits = nullptr;    // bits per channel of the modem launch a call consumes (pinned host copy, device copy)
};

extern "C" {

int sonde_softin_dev_create(int32_t n_channels, int32_t sonde_type, int32_t ecc_level, int32_t invert_stream, int32_t opt_inv, int32_t opt_auto, sonde_softin_dev_t **out) {
    if (!out || n_channels < 1 || (sonde_type != SONDE_RS41 && sonde_type != SONDE_DFM09 && sonde_type != SONDE_M10) || ecc_level < 0 || ecc_level > 2) return SONDE_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "libsonde_hip: no usable HIP device (the batched soft-bit framer has no CPU fallback)\n"); return SONDE_E_NOGPU; }
    sonde_softin_dev *s = new sonde_softin_dev();
    s->C = n_channels; s->ecc_level = ecc_level; s->type = sonde_type;
    // frames a call of about a second can complete per channel: RS41 two, DFM09 six (280 bits at 1250 b/s), M10 two (one per second; the rest of the second is skipped)
    s->cap = (sonde_type == SONDE_DFM09 ? 8 : 2) * n_channels + 16;
    const size_t C = (size_t)n_channels, cap = (size_t)s->cap;
    std::vector<SoftinChan> init(C);
    memset(init.data(), 0, C * sizeof(SoftinChan));
    for (auto &c : init) { c.inv = opt_inv ? 1 : 0; memcpy(c.frame, sonde::kRs41HeaderBytes, 8); }
    unsigned char hdr[136];
    memset(hdr, 0, sizeof hdr);
    if (sonde_type == SONDE_RS41) { memcpy(hdr, sonde::kRs41Header, 64); memcpy(hdr + 64, sonde::kRs41HeaderBytes, 8); memcpy(hdr + 72, sonde::kRs41Mask, 64); }
    else memcpy(hdr, sonde_type == SONDE_DFM09 ? sonde::kDfmRawHeader : sonde::kM10RawHeader, 32);
    bool ok = hipMalloc((void **)&s->d_chan, C * sizeof(SoftinChan)) == hipSuccess && hipMalloc((void **)&s->d_frames, cap * 518) == hipSuccess
           && hipMalloc((void **)&s->d_hdr, sizeof hdr) == hipSuccess && hipMalloc((void **)&s->d_gf, 768) == hipSuccess && hipMalloc((void **)&s->d_synd, cap * 48) == hipSuccess
           && hipMalloc((void **)&s->d_flen, cap * 4) == hipSuccess && hipMalloc((void **)&s->d_ecc, cap * 4) == hipSuccess && hipMalloc((void **)&s->d_codes, cap * 8) == hipSuccess
           && hipMalloc((void **)&s->d_meta, cap * sizeof(SoftinMeta)) == hipSuccess && hipMalloc((void **)&s->d_count, 8) == hipSuccess && hipHostMalloc((void **)&s->h_count, 8) == hipSuccess;
    ok = ok && hipMemcpy(s->d_chan, init.data(), C * sizeof(SoftinChan), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice) == hipSuccess
            && hipMemcpy(s->d_gf, sonde::gf_exp_table(), 512, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->d_gf + 512, sonde::gf_log_table(), 256, hipMemcpyHostToDevice) == hipSuccess
            && hipMemset(s->d_ecc, 0, cap * 4) == hipSuccess;
    if (ok && sonde_type == SONDE_DFM09) {
        std::vector<SoftinDfmChan> di(C);
        memset(di.data(), 0, C * sizeof(SoftinDfmChan));
        static const char kDfmHdr[] = "0100010111001111";                                 // dfm09mod.c:1503-1506: the header's 16 bits in front of the first frame of a hit
        for (auto &c : di) { c.inv = opt_inv ? 1 : 0; c.dpos = 16; for (int i = 0; i < 16; i++) c.dhb[i] = (unsigned char)(kDfmHdr[i] & 1); }
        ok = hipMalloc((void **)&s->d_dfm_chan, C * sizeof(SoftinDfmChan)) == hipSuccess && hipMalloc((void **)&s->d_dfm_out, cap * sizeof(sonde_dfm_frame_t)) == hipSuccess
          && hipMemcpy(s->d_dfm_chan, di.data(), C * sizeof(SoftinDfmChan), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && s->h_dfm.alloc(cap);
    }
    if (ok && sonde_type == SONDE_M10) {
        std::vector<SoftinM10Chan> mi(C);
        memset(mi.data(), 0, C * sizeof(SoftinM10Chan));
        for (auto &c : mi) { c.inv = opt_inv ? 1 : 0; c.mbit0 = '0'; }
        ok = hipMalloc((void **)&s->d_m10_chan, C * sizeof(SoftinM10Chan)) == hipSuccess && hipMalloc((void **)&s->d_m10_out, cap * sizeof(sonde_m10_frame_t)) == hipSuccess
          && hipMemcpy(s->d_m10_chan, mi.data(), C * sizeof(SoftinM10Chan), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && s->h_m10.alloc(cap);
    }
    if (!ok) { sonde_softin_dev_destroy(s); return SONDE_E_NOMEM; }
    SoftinArgs &a = s->args;
    a.n_ch = n_channels; a.inv_in = invert_stream ? 1 : 0; a.opt_auto = opt_auto ? 1 : 0; a.ths = sonde_type == SONDE_M10 ? 0.8f : 0.7f;
    a.chan = s->d_chan; a.frames = s->d_frames; a.flen = s->d_flen; a.meta = s->d_meta; a.count = s->d_count; a.cap = s->cap; a.hdr = s->d_hdr;
    if (!s->h_ecc.alloc(cap) || !s->h_meta.alloc(cap) || !s->h_frames.alloc(cap * 518)) { sonde_softin_dev_destroy(s); return SONDE_E_NOMEM; }
    s->head = std::min(s->cap, (sonde_type == SONDE_DFM09 ? 5 : 1) * n_channels + 16);
    (void)hipGetLastError();
    *out = s;
    return 0;
}

void sonde_softin_dev_destroy(sonde_softin_dev_t *s) {
    if (!s) return;
    if (s->pending && s->pend_stream) (void)hipStreamSynchronize(s->pend_stream);
    if (s->own_stream && s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    if (s->h_count) hipHostFree(s->h_count);
    if (s->h_nbits) hipHostFree(s->h_nbits);
    if (s->d_nbits) hipFree(s->d_nbits);
    (void)hipGetLastError();
    void *p[] = { s->d_chan, s->d_frames, s->d_hdr, s->d_gf, s->d_synd, s->d_flen, s->d_ecc, s->d_codes, s->d_meta, s->d_count, s->d_dfm_chan, s->d_dfm_out, s->d_m10_chan, s->d_m10_out };
    for (void *q : p) if (q) hipFree(q);
    delete s;
}

// one pass of the framer (and rs41_ecc() over the frames it completes) over the channels of ch_list (null: all), its records behind the `off` records the
// call already has; counter `which` of d_count
static int softin_pass(sonde_softin_dev *s, hipStream_t st, const int off, const int nblocks, const int *ch_list, const int which) {
    SoftinArgs a = s->args;
    a.frames += (size_t)off * 518; a.flen += off; a.meta += off; a.cap = s->cap - off; a.count = s->d_count + which; a.ch_list = ch_list;
    HIPCHK(hipMemsetAsync(a.count, 0, 4, st));
    if (s->type == SONDE_DFM09) { SoftinDfmArgs d{a, s->d_dfm_chan, s->d_dfm_out + off, s->ecc_level}; hipLaunchKernelGGL(k_softin_dfm, dim3(nblocks), dim3(64), 0, st, d); }
    else if (s->type == SONDE_M10) {
        // LDS for the call's soft decisions: what a call can hold (a modem launch: its capacity; a pushed stream: its length), up to M10_STAGE_MAX
        long long need = a.fsk_chan || a.nbits_ch ? a.ch_stride : a.nbits;
        if (need > M10_STAGE_MAX || need < 0) need = 0;
        SoftinM10Args m{a, s->d_m10_chan, s->d_m10_out + off, (int)need};
        static size_t attr = 0;
        const size_t lds = (size_t)need * sizeof(float);
        if (lds > attr) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_softin_m10), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) attr = lds; else m.stage_cap = 0; }
        hipLaunchKernelGGL(k_softin_m10, dim3(nblocks), dim3(64), m.stage_cap ? lds : 0, st, m);
    }
    else {
        hipLaunchKernelGGL(k_softin_rs41, dim3(nblocks), dim3(64), 0, st, a);
        if (s->ecc_level > 0)
            sonde_launch_rs41_ecc_batch_n(a.frames, a.flen, a.count, a.cap, s->ecc_level, s->d_ecc + off, s->d_codes + 2 * (size_t)off, s->d_synd + 48 * (size_t)off, s->d_gf, s->d_gf + 512, st);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->h_count + which, a.count, 4, hipMemcpyDeviceToHost, st));
    return 0;
}
// records [from, to) of the call to the host
static int softin_copy(sonde_softin_dev *s, hipStream_t st, const int from, const int to) {
    if (to <= from) return 0;
    const size_t n = (size_t)(to - from);
    if (s->type == SONDE_DFM09) HIPCHK(hipMemcpyAsync(s->h_dfm.data() + from, s->d_dfm_out + from, n * sizeof(sonde_dfm_frame_t), hipMemcpyDeviceToHost, st));
    else if (s->type == SONDE_M10) HIPCHK(hipMemcpyAsync(s->h_m10.data() + from, s->d_m10_out + from, n * sizeof(sonde_m10_frame_t), hipMemcpyDeviceToHost, st));
    else {
        HIPCHK(hipMemcpyAsync(s->h_meta.data() + from, s->d_meta + from, n * sizeof(SoftinMeta), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(s->h_ecc.data() + from, s->d_ecc + from, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(s->h_frames.data() + (size_t)from * 518, s->d_frames + (size_t)from * 518, n * 518, hipMemcpyDeviceToHost, st));   // 518 bytes per frame and second: not the soft decisions
    }
    return 0;
}
// enqueue a call: the framer over what args.sd / nbits describe, the block code, the counter and the records a call normally completes (`head`) on their way to the host
static int softin_enqueue(sonde_softin_dev *s, hipStream_t st) {
    if (s->pending) return SONDE_E_ARG;
    { const int rc = softin_pass(s, st, 0, s->C, nullptr, 0); if (rc) return rc; }
    { const int rc = softin_copy(s, st, 0, s->head); if (rc) return rc; }
    s->pending = true; s->pend_stream = st;
    return 0;
}
// the other half: wait, take what the head did not cover, queue the frames and add up the tallies
static int softin_finish(sonde_softin_dev *s) {
    if (!s->pending) return 0;
    hipStream_t st = s->pend_stream;
    s->pending = false;
    HIPCHK(hipStreamSynchronize(st));
    long long n = s->h_count[0];
    if (n > s->cap) { s->dropped += n - s->cap; n = s->cap; }
    const int have = (int)std::min<long long>(n, s->head);
    if (n > have) { const int rc = softin_copy(s, st, have, (int)n); if (rc) return rc; HIPCHK(hipStreamSynchronize(st)); }
    if (s->type == SONDE_DFM09) {
        // (frames of one channel in order: the slots of a call are handed out in completion order per channel, channels interleave)
        for (long long i = 0; i < n; i++) {
            const sonde_dfm_frame_t &f = s->h_dfm[i];
            s->qdfm.push_back(f); s->frames_total++;
            const bool okf = f.ecc[0] >= 0 && f.ecc[1] >= 0 && f.ecc[2] >= 0;
            if (okf) s->ecc_ok_total++;
            if (okf && (f.ecc[0] > 0 || f.ecc[1] > 0 || f.ecc[2] > 0)) { s->repaired_total++; s->symbols_total += __builtin_popcount((unsigned)f.ecc[0]) + __builtin_popcount((unsigned)f.ecc[1]) + __builtin_popcount((unsigned)f.ecc[2]); }
        }
    } else if (s->type == SONDE_M10) {
        for (long long i = 0; i < n; i++) { s->qm10.push_back(s->h_m10[i]); s->frames_total++; if (s->h_m10[i].cs_ok) s->ecc_ok_total++; }
    } else {
        for (long long i = 0; i < n; i++) {
            sonde_frame_t f; memset(&f, 0, sizeof f);
            const SoftinMeta &m = s->h_meta[i];
            f.channel = m.channel; f.len = m.len; f.nbytes = m.nbytes; f.mv = m.mv; f.mv_pos = (uint32_t)m.hdr_bit;
            f.ecc = s->ecc_level > 0 ? s->h_ecc[i] : 0;
            memcpy(f.frame, s->h_frames.data() + (size_t)i * 518, 518);
            s->queue.push_back(f);
            s->frames_total++;
            if (f.ecc >= 0) s->ecc_ok_total++;
            if (f.ecc > 0) { s->repaired_total++; s->symbols_total += f.ecc; }
        }
    }
    return 0;
}

// the modem's last launch (waited for first: its channels' frame counts come from the host's copy of the records, and whatever the modem had to repeat is repeated by
// then) as this call's input: the soft decisions where that launch left them, the bits per channel uploaded
static int softin_bind_fsk(sonde_softin_dev *s, sonde_fsk_t *modem, hipStream_t st) {
    const float *d_sd = nullptr; long long cap = 0; const FskChan *d_chan = nullptr; int bpf = 0, nch = 0; hipStream_t ms = nullptr;
    { const int rc = sonde_fsk_wait(modem); if (rc) return rc; }
    const int rc = sonde_fsk_dev_view(modem, &d_sd, &cap, &d_chan, &bpf, &nch, &ms);
    if (rc) return rc;
    if (nch != s->C) return SONDE_E_ARG;
    if (!s->h_nbits) {
        if (hipHostMalloc((void **)&s->h_nbits, (size_t)s->C * sizeof(int)) != hipSuccess || hipMalloc((void **)&s->d_nbits, (size_t)s->C * sizeof(int)) != hipSuccess) return SONDE_E_NOMEM;
    }
    { const int rc2 = sonde_fsk_host_frames(modem, s->h_nbits); if (rc2) return rc2; }
    for (int c = 0; c < s->C; c++) s->h_nbits[c] = s->h_nbits[c] > 0 ? s->h_nbits[c] * bpf : 0;
    HIPCHK(hipMemcpyAsync(s->d_nbits, s->h_nbits, (size_t)s->C * sizeof(int), hipMemcpyHostToDevice, st));
    SoftinArgs &a = s->args;
    a.sd = d_sd; a.ch_stride = cap; a.fsk_chan = nullptr; a.bits_per_frame = bpf; a.nbits_ch = s->d_nbits; a.nbits = 0;
    return 0;
}
static int softin_own_stream(sonde_softin_dev *s) {
    if (!s->stream) { HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true; }
    return 0;
}
int sonde_softin_dev_push_fsk(sonde_softin_dev_t *s, sonde_fsk_t *modem) {
    if (!s || !modem) return SONDE_E_ARG;
    { const int rc = softin_finish(s); if (rc) return rc; }
    { const int rc = softin_own_stream(s); if (rc) return rc; }
    { const int rc = softin_bind_fsk(s, modem, s->stream); if (rc) return rc; }
    { const int rc = softin_enqueue(s, s->stream); if (rc) return rc; }
    { const int rc = sonde_fsk_dev_reader_done(modem, s->stream); if (rc) return rc; }
    return softin_finish(s);
}
// the same in two halves: submit waits for the modem's launch (sonde_fsk_wait), then enqueues the consumer's kernels and the copies of its frames on the consumer's OWN
// stream and returns; collect waits for them.  In between the modem can be given its next second (sonde_fsk_submit_device): the modem keeps the soft decisions of two
// launches (they alternate between two buffers), so the consumer of second k runs beside the modem of second k + 1.  Collect before the modem's launch after that.
int sonde_softin_dev_submit_fsk(sonde_softin_dev_t *s, sonde_fsk_t *modem) {
    if (!s || !modem) return SONDE_E_ARG;
    { const int rc = softin_finish(s); if (rc) return rc; }
    { const int rc = softin_own_stream(s); if (rc) return rc; }
    { const int rc = softin_bind_fsk(s, modem, s->stream); if (rc) return rc; }
    { const int rc = softin_enqueue(s, s->stream); if (rc) return rc; }
    // the modem's launch that reuses this buffer of soft decisions (the next but one) waits for the consumer's kernels: submitting the modem twice before
    // sonde_softin_dev_collect is slower, not wrong
    return sonde_fsk_dev_reader_done(modem, s->stream);
}
// The consumer over the launch BEFORE the one the modem has in flight: for a caller that hands the modem its next second first (sonde_fsk_wait (k - 1),
// sonde_fsk_submit_device (k), then this for k - 1) — the modem's stream never waits for the host's decoder bookkeeping.  Nothing waits here; with no launch in
// flight this is sonde_softin_dev_submit_fsk.
int sonde_softin_dev_submit_fsk_behind(sonde_softin_dev_t *s, sonde_fsk_t *modem) {
    if (!s || !modem) return SONDE_E_ARG;
    { const int rc = softin_finish(s); if (rc) return rc; }
    { const int rc = softin_own_stream(s); if (rc) return rc; }
    if (!s->h_nbits) {
        if (hipHostMalloc((void **)&s->h_nbits, (size_t)s->C * sizeof(int)) != hipSuccess || hipMalloc((void **)&s->d_nbits, (size_t)s->C * sizeof(int)) != hipSuccess) return SONDE_E_NOMEM;
    }
    const float *d_sd = nullptr; long long cap = 0; int bpf = 0, nch = s->C;
    const int have = sonde_fsk_dev_view_prev(modem, &d_sd, &cap, &bpf, &nch, s->h_nbits);
    if (have < 0) return have;
    if (have == 0) return sonde_softin_dev_submit_fsk(s, modem);
    if (nch != s->C) return SONDE_E_ARG;
    for (int c = 0; c < s->C; c++) s->h_nbits[c] = s->h_nbits[c] > 0 ? s->h_nbits[c] * bpf : 0;
    HIPCHK(hipMemcpyAsync(s->d_nbits, s->h_nbits, (size_t)s->C * sizeof(int), hipMemcpyHostToDevice, s->stream));
    SoftinArgs &a = s->args;
    a.sd = d_sd; a.ch_stride = cap; a.fsk_chan = nullptr; a.bits_per_frame = bpf; a.nbits_ch = s->d_nbits; a.nbits = 0;
    { const int rc = softin_enqueue(s, s->stream); if (rc) return rc; }
    return sonde_fsk_dev_reader_done_prev(modem, s->stream);
}
int sonde_softin_dev_collect(sonde_softin_dev_t *s) {
    if (!s) return SONDE_E_ARG;
    return softin_finish(s);
}

int sonde_softin_dev_push_device(sonde_softin_dev_t *s, const float *d_soft, int64_t ch_stride, int32_t n_bits) {
    if (!s || !d_soft || n_bits < 0 || ch_stride < n_bits) return SONDE_E_ARG;
    { const int rc = softin_finish(s); if (rc) return rc; }
    if (!s->stream) { HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true; }
    SoftinArgs &a = s->args;
    a.sd = d_soft; a.ch_stride = ch_stride; a.fsk_chan = nullptr; a.nbits_ch = nullptr; a.nbits = n_bits;
    { const int rc = softin_enqueue(s, s->stream); if (rc) return rc; }
    return softin_finish(s);
}

int sonde_softin_dev_fetch(sonde_softin_dev_t *s, sonde_frame_t *out, int32_t max) {
    if (s && s->pending) { const int rc_ = softin_finish(s); if (rc_) return rc_; }
    if (!s || (!out && max > 0)) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->queue.size(), (size_t)(max < 0 ? 0 : max));
    for (int i = 0; i < n; i++) out[i] = s->queue[i];
    s->queue.erase(s->queue.begin(), s->queue.begin() + n);
    return n;
}

int sonde_softin_dev_fetch_dfm(sonde_softin_dev_t *s, sonde_dfm_frame_t *out, int32_t max) {
    if (s && s->pending) { const int rc_ = softin_finish(s); if (rc_) return rc_; }
    if (!s || (!out && max > 0)) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->qdfm.size(), (size_t)(max < 0 ? 0 : max));
    for (int i = 0; i < n; i++) out[i] = s->qdfm[i];
    s->qdfm.erase(s->qdfm.begin(), s->qdfm.begin() + n);
    return n;
}

int sonde_softin_dev_fetch_m10(sonde_softin_dev_t *s, sonde_m10_frame_t *out, int32_t max) {
    if (s && s->pending) { const int rc_ = softin_finish(s); if (rc_) return rc_; }
    if (!s || (!out && max > 0)) return SONDE_E_ARG;
    const int n = (int)std::min<size_t>(s->qm10.size(), (size_t)(max < 0 ? 0 : max));
    for (int i = 0; i < n; i++) out[i] = s->qm10[i];
    s->qm10.erase(s->qm10.begin(), s->qm10.begin() + n);
    return n;
}

int sonde_softin_dev_counts(sonde_softin_dev_t *s, int64_t *frames, int64_t *ecc_ok, int64_t *repaired, int64_t *symbols, int64_t *dropped) {
    if (s && s->pending) { const int rc_ = softin_finish(s); if (rc_) return rc_; }
    if (!s) return SONDE_E_ARG;
    if (frames) *frames = s->frames_total;
    if (ecc_ok) *ecc_ok = s->ecc_ok_total;
    if (repaired) *repaired = s->repaired_total;
    if (symbols) *symbols = s->symbols_total;
    if (dropped) *dropped = s->dropped;
    return 0;
}

}  // extern "C"
