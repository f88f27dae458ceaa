// sonde_rs_dev.h — RS(255,231) on the device: the errors-only decoder on ONE wavefront (a polynomial coefficient per lane) and the
// two passes of rs41_ecc() on a workgroup of four wavefronts (k_rs41_ecc_frames).  Behaviour reproduced (not code): bch_ecc_mod.c rs_decode_ErrEra :877-960 with
// nera = 0 (polyGF_lfsr :547-578, poly_divmod :410-454, poly_mul :469-490, Chien loop :926-940, forney :596-612) and rs41mod.c
// rs41_ecc :1703-1769, :1955-1974.  The key equation is solved by the same extended Euclid on (S, x^24) with the same stop rule and the
// same acceptance tests, so a word the reference cannot repair fails here with the same code, and a word it miscorrects is miscorrected
// into the same bytes.
//
// The file is compiled twice: by hipcc into k_rs41_ecc_frames / k_rs41_ecc_batch (sonde_kernels.hip), and by g++ under tests/emu/wave_emu.h, which runs every
// thread of a workgroup as a fiber and turns the cross-lane operations below into rendezvous points (tests/test_rs_dev_emu.py: the
// device code against the compiled reference without a GPU).  Control flow around every rsw_* call is wave-uniform.
#ifndef SONDE_RS_DEV_H
#define SONDE_RS_DEV_H
#include <stdint.h>

#ifndef SONDE_RS_EMU
#define RSW_DEV __device__ __forceinline__
#define RSW_DEV_NOINLINE __device__ __noinline__
// value of lane `src` (wave-uniform src)
static RSW_DEV int rsw_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// value of lane - d, 0 for the first d lanes
static RSW_DEV int rsw_shfl_up(int v, int d, int lane) { const int r = __shfl_up(v, (unsigned)d); return lane >= d ? r : 0; }
static RSW_DEV unsigned long long rsw_ballot(bool p) { return __ballot(p); }
// LDS written by one lane of a wave, read by another lane of the same wave
static RSW_DEV void rsw_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
static RSW_DEV void rsw_syncthreads() { __syncthreads(); }
static RSW_DEV int rsw_clzll(unsigned long long m) { return __clzll((long long)m); }
static RSW_DEV int rsw_popcll(unsigned long long m) { return __popcll(m); }
#endif

struct RsGf { const uint8_t *exp, *log; };            // exp[512] (two periods), log[256]; LDS copies of sonde::gf_exp_table() / gf_log_table()

static RSW_DEV int rs_gf_mul(const RsGf g, int a, int b) { return (a && b) ? g.exp[g.log[a] + g.log[b]] : 0; }
static RSW_DEV int rs_gf_mul_l(const RsGf g, int a, int logb) { return a ? g.exp[g.log[a] + logb] : 0; }      // b != 0 given by its logarithm
static RSW_DEV int rs_gf_inv(const RsGf g, int a) { return a ? g.exp[255 - g.log[a]] : 0; }
// poly_deg (bch_ecc_mod.c:403): highest lane with a non-zero coefficient, -1 for the zero polynomial
static RSW_DEV int rsw_deg(int v) { const unsigned long long m = rsw_ballot(v != 0); return m ? 63 - rsw_clzll(m) : -1; }

// rs_decode() of one codeword on one wave.  syn = S[lane] for lane < 24 (0 above), cw = the 255 codeword bytes in LDS (repaired in place
// on success, untouched otherwise), scr = 64 bytes of LDS owned by this wave.  Returns what rs_decode returns: 0 (clean), the number of
// repaired symbols, -1 (fewer roots than the locator's degree), -2 (Lambda(0) = 0), -3 (deg Omega >= deg Lambda).
static RSW_DEV int rs255_wave_decode(uint8_t *cw, int syn, uint8_t *scr, const RsGf g, int lane) {
    if (rsw_ballot(syn != 0) == 0) return 0;
    // polyGF_lfsr: r0 = S, r1 = x^24, s0 = 1, s1 = 0; while deg r1 >= 12: (quo, rem) = r0 / r1; r0 = r1; r1 = rem; s2 = quo s1 + s0; ..
    int r0 = syn, r1 = (lane == 24) ? 1 : 0, s0 = (lane == 0) ? 1 : 0, s1 = 0;
    int d1 = 24;
    while (d1 >= 12) {
        int dp = rsw_deg(r0);
        int quo = 0, rem = r0;                                        // deg p < deg q: d = 0, r = p (:433)
        if (dp >= d1) {
            const int lqi = g.log[rs_gf_inv(g, rsw_bcast(r1, d1))];   // 1 / lead(q), as a logarithm
            while (dp >= d1) {
                const int c = rs_gf_mul_l(g, rsw_bcast(rem, dp), lqi);                    // lead(rem) is non-zero
                const int sh = dp - d1;
                if (lane == sh) quo = c;
                rem ^= rs_gf_mul_l(g, rsw_shfl_up(r1, sh, lane), g.log[c]);
                dp = rsw_deg(rem);
            }
        }
        r0 = r1; r1 = rem;
        int s2 = s0;                                                  // poly_mul(quo, s1) + s0
        const int dq = rsw_deg(quo);
        for (int i = 0; i <= dq; i++) {
            const int q = rsw_bcast(quo, i);
            const int t = rsw_shfl_up(s1, i, lane);
            if (q) s2 ^= rs_gf_mul_l(g, t, g.log[q]);
        }
        s0 = s1; s1 = s2;
        d1 = rsw_deg(r1);
    }
    const int dL = rsw_deg(s1), dO = d1;                              // Lambda = s1, Omega = r1
    if (dO >= dL) return -3;
    const int gamma = rsw_bcast(s1, 0);
    if (!gamma) return -2;
    const int lgi = g.log[rs_gf_inv(g, gamma)];
    const int lam = rs_gf_mul_l(g, s1, lgi), om = rs_gf_mul_l(g, r1, lgi);
    // Chien search over x = 1 .. 255 in this order (:926), lane l looks at x = l+1, l+65, l+129, l+193; coefficients through LDS
    if (lane < 32) { scr[lane] = (uint8_t)lam; scr[32 + lane] = (uint8_t)om; }
    rsw_wave_sync();
    int y[4] = {0, 0, 0, 0}, lx[4];
    for (int k = 0; k < 4; k++) lx[k] = g.log[(lane + 1 + 64 * k) & 255];
    for (int n = dL; n >= 0; n--) {
        const int c = scr[n];
        for (int k = 0; k < 4; k++) y[k] = rs_gf_mul_l(g, y[k], lx[k]) ^ c;
    }
    int nroots = 0;
    bool root[4];
    for (int k = 0; k < 4; k++) {
        root[k] = (lane + 1 + 64 * k <= 255) && y[k] == 0;
        nroots += rsw_popcll(rsw_ballot(root[k]));
    }
    if (nroots < dL) return -1;                                       // a polynomial of degree dL has at most dL roots: nroots == dL here
    for (int k = 0; k < 4; k++) {
        if (!root[k]) continue;
        // forney (:596): Y = Omega(x) / Lambda'(x) / x for b = 0; Lambda' keeps the odd coefficients, one degree down
        int w = 0, z = 0;
        for (int n = dO; n >= 0; n--) w = rs_gf_mul_l(g, w, lx[k]) ^ scr[32 + n];
        for (int n = dL - 1; n >= 0; n--) z = rs_gf_mul_l(g, z, lx[k]) ^ ((n & 1) ? 0 : scr[n + 1]);
        const int xinv_log = (255 - lx[k]) % 255;
        const int val = z ? rs_gf_mul_l(g, rs_gf_mul(g, w, rs_gf_inv(g, z)), xinv_log) : 0;
        cw[xinv_log] ^= (uint8_t)val;                                 // err_pos = log(1 / x) (:930)
    }
    return nroots;
}

// Horner partial sums of the 48 syndromes of both codewords by a workgroup of nw waves (nw divides 256): wave c covers the 256 / nw
// coefficients from (256 / nw) c on and scales by alpha^(j c 256 / nw); the XOR over the waves is S_j = cw(alpha^j), j < 24 (syndromes :638,
// rs41mod.c:1729-1732).  All threads call it.
static RSW_DEV void rs41_syndrome_partials(const uint8_t (*cw)[256], uint8_t (*part)[48], const RsGf g, int lane, int wave, int nw) {
    if (lane < 48) {
        const int c = lane / 24, jx = lane % 24, ch = 256 / nw;
        int h = 0;
        for (int i = ch - 1; i >= 0; i--) {
            const int n = ch * wave + i;
            h = rs_gf_mul_l(g, h, jx) ^ (n < 255 ? cw[c][n] : 0);     // x = alpha^jx, log x = jx
        }
        part[wave][lane] = (uint8_t)(h ? g.exp[(g.log[h] + (jx * ch * wave) % 255) % 255] : 0);
    }
}

// rs41_ecc() for ecc levels 1 / 2 on a full frame, by a workgroup of nt threads (nt / 64 divides 256; the product uses 256).
// frame = 518 bytes in LDS, de-whitened, bytes from flen on zero already (rs41mod.c:1727); cw = [2][256], part = [nt / 64][48], res = int[4],
// scr = [2][64]: LDS scratch.  synd_in = the 48 first-pass syndromes when the caller has them (k_framesync computes them for every frame), else
// nullptr: computed here.  On return frame holds the bytes rs41_ecc leaves in gpx->frame and res[0..1] the two rs_decode values of the last
// pass.  Returns rs41_ecc's value: corrected symbols of both codewords, or -1 / -2 / -3 = codeword 1 / 2 / both failed.
static RSW_DEV int rs41_ecc_wg(uint8_t *frame, int level, uint8_t (*cw)[256], uint8_t (*part)[48], int *res, uint8_t (*scr)[64],
                               const uint8_t *synd_in, const RsGf g, int tid, int nt) {
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int i = tid; i < 510; i += nt) {                             // two interleaved codewords: 24 parity bytes each, then the message (:1730-1733)
        const int c = i / 255, n = i % 255;
        cw[c][n] = n < 24 ? frame[8 + 24 * c + n] : frame[56 + 2 * (n - 24) + c];
    }
    if (tid < 4) res[tid] = 0;
    rsw_syncthreads();
    for (int pass = 0; pass < 2; pass++) {
        const bool given = pass == 0 && synd_in != nullptr;
        if (!given) { rs41_syndrome_partials(cw, part, g, lane, wave, nw); rsw_syncthreads(); }
        if (wave < 2) {
            int s = 0;
            if (lane < 24) { if (given) s = synd_in[24 * wave + lane]; else for (int w = 0; w < nw; w++) s ^= part[w][24 * wave + lane]; }
            const int e = rs255_wave_decode(cw[wave], s, scr[wave], g, lane);
            if (lane == 0) res[wave] = e;
        }
        rsw_syncthreads();
        if (pass == 1 || level < 2 || (res[0] >= 0 && res[1] >= 0)) break;
        // 2nd pass (:1739-1769): the block ids every RS41 frame has, and the zero tail; message bytes re-read from the frame, parity as the
        // first pass left it
        int ft = 0; { const int b = frame[0x38]; for (int q = 0; q < 4; q++) ft += ((b >> q) & 1) - ((b >> (q + 4)) & 1); }
        rsw_syncthreads();
        for (int i = tid; i < 518; i += nt) {
            int v = frame[i];
            if (ft < -2) { if (i >= 320 + 7 && i < 518 - 2) v = 0; }
            else {
                if (i >= 320) v = 0;
                if (i >= 0x12D && i < 318) v = 0;
                if (i == 0x12B) v = 0x76; if (i == 0x12C) v = 0x11;
                if (i == 318) v = 0xEC; if (i == 319) v = 0xC7;
            }
            switch (i) {
                case 0x039: v = 0x79; break; case 0x03A: v = 0x28; break;
                case 0x065: v = 0x7A; break; case 0x066: v = 0x2A; break;
                case 0x093: v = 0x7C; break; case 0x094: v = 0x1E; break;
                case 0x0B5: v = 0x7D; break; case 0x0B6: v = 0x59; break;
                case 0x112: v = 0x7B; break; case 0x113: v = 0x15; break;
                default: break;
            }
            frame[i] = (uint8_t)v;
        }
        rsw_syncthreads();
        for (int i = tid; i < 510; i += nt) { const int c = i / 255, n = i % 255; if (n >= 24) cw[c][n] = frame[56 + 2 * (n - 24) + c]; }
        rsw_syncthreads();
    }
    for (int i = tid; i < 510; i += nt) {                             // (:1955-1958)
        const int c = i / 255, n = i % 255;
        if (n < 24) frame[8 + 24 * c + n] = cw[c][n]; else frame[56 + 2 * (n - 24) + c] = cw[c][n];
    }
    const int e1 = res[0], e2 = res[1];
    rsw_syncthreads();
    return (e1 < 0 || e2 < 0) ? -((e1 < 0 ? 1 : 0) | (e2 < 0 ? 2 : 0)) : e1 + e2;
}

#endif
