// rs_dev_emu.cpp — test infrastructure: the device RS(255,231) decoder (csrc/sonde_rs_dev.h) compiled for the CPU under wave_emu.h.
//   emu_rs255_decode(cw[255])                      one wave, as k_framesync's waves 0 / 1 run it
//   emu_rs41_ecc(frame[518], flen, level, synd_in[48] | null, nthreads)  the workgroup form (the product runs 256 threads)
#include "wave_emu.h"
#include "../../radiosonde_auto_rx_amd/csrc/sonde_rs_dev.h"
#include <cstring>

static uint8_t g_exp[512], g_log[256];
static void gf_init() {
    static bool ready = false;
    if (ready) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { g_exp[i] = (uint8_t)x; g_log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11D; }   // GF_genTab, f = 0x11D (bch_ecc_mod.c:136)
    for (int i = 255; i < 512; i++) g_exp[i] = g_exp[i - 255];
    g_log[0] = 0;
    ready = true;
}

extern "C" int emu_rs255_decode(uint8_t *cw_io) {
    gf_init();
    const RsGf g{g_exp, g_log};
    uint8_t cw[256]; memcpy(cw, cw_io, 255); cw[255] = 0;
    uint8_t S[24];
    for (int j = 0; j < 24; j++) { int y = 0; for (int n = 254; n >= 0; n--) y = rs_gf_mul_l(g, y, j) ^ cw[n]; S[j] = (uint8_t)y; }
    uint8_t scr[64]; int ret = 0;
    emu::run_workgroup(64, [&](int tid) {
        const int e = rs255_wave_decode(cw, tid < 24 ? S[tid] : 0, scr, g, tid);
        if (tid == 0) ret = e;
    });
    memcpy(cw_io, cw, 255);
    return ret;
}

// synd_in: 48 first-pass syndromes (as k_framesync leaves them in the record) or nullptr = computed by the workgroup itself
extern "C" int emu_rs41_ecc(uint8_t *frame_io, int flen, int level, const uint8_t *synd_in, int nthreads) {
    gf_init();
    const RsGf g{g_exp, g_log};
    uint8_t frame[520]; memcpy(frame, frame_io, 518);
    for (int i = flen; i < 518; i++) frame[i] = 0;
    static uint8_t cw[2][256], part[16][48], scr[2][64]; int res[4], ret = 0;
    emu::run_workgroup(nthreads, [&](int tid) {
        const int e = rs41_ecc_wg(frame, level, cw, part, res, scr, synd_in, g, tid, nthreads);
        if (tid == 0) ret = e;
    });
    memcpy(frame_io, frame, 518);
    return ret;
}
