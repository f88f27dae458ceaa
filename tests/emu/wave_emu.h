// wave_emu.h — test infrastructure: run the threads of ONE workgroup of a HIP kernel on the CPU, each as a ucontext fiber, so that device
// functions written against a handful of cross-lane primitives (radiosonde_auto_rx_amd/csrc/sonde_rs_dev.h: rsw_bcast, rsw_shfl_up,
// rsw_ballot, rsw_wave_sync, rsw_syncthreads) can be checked against the compiled reference without a GPU.  A primitive is a rendezvous of
// the 64 fibers of a wave (or of all fibers for the workgroup barrier): write your value, wait until every lane has, read, wait until
// every lane has read.  Control flow around the primitives must be wave-uniform, as on the hardware.  Not used by the product.
#ifndef WAVE_EMU_H
#define WAVE_EMU_H
#include <ucontext.h>
#include <functional>
#include <vector>
#include <stdexcept>
#include <stdint.h>

namespace emu {

struct Fiber { ucontext_t ctx; std::vector<char> stack; int state = 0; const unsigned *wait_gen = nullptr; unsigned my_gen = 0; };   // 0 ready, 1 waiting, 2 done
struct Wave { int count = 0; unsigned gen = 0; long long buf[64]; };
struct Group {
    int n = 0, cur = 0;
    std::vector<Fiber> f; std::vector<Wave> w;
    int wg_count = 0; unsigned wg_gen = 0;
    ucontext_t main;
    std::function<void(int)> body;
};
static thread_local Group *g_grp = nullptr;

static inline int tid() { return g_grp->cur; }
static inline void yield_() { Group *g = g_grp; swapcontext(&g->f[g->cur].ctx, &g->main); }
static inline void rendezvous(unsigned &gen, int &count, int n) {
    const unsigned my = gen;
    if (++count == n) { count = 0; gen++; return; }
    Fiber &me = g_grp->f[g_grp->cur];
    me.wait_gen = &gen; me.my_gen = my; me.state = 1;
    yield_();
}
static inline Wave &my_wave() { return g_grp->w[g_grp->cur >> 6]; }
static inline void wave_rendezvous() { Wave &w = my_wave(); rendezvous(w.gen, w.count, 64); }

static void tramp() { Group *g = g_grp; g->body(g->cur); g->f[g->cur].state = 2; swapcontext(&g->f[g->cur].ctx, &g->main); }

// run body(tid) for tid = 0 .. nthreads-1 (a multiple of 64) as one workgroup
static inline void run_workgroup(int nthreads, std::function<void(int)> body) {
    if (nthreads % 64) throw std::runtime_error("emu: workgroup size must be a multiple of 64");
    Group g; g.n = nthreads; g.body = body; g.f.resize(nthreads); g.w.resize(nthreads / 64);
    Group *prev = g_grp; g_grp = &g;
    for (int i = 0; i < nthreads; i++) {
        Fiber &f = g.f[i];
        f.stack.resize(64 * 1024);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data(); f.ctx.uc_stack.ss_size = f.stack.size(); f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())tramp, 0);
    }
    for (;;) {
        int done = 0, ran = 0;
        for (int i = 0; i < nthreads; i++) {
            Fiber &f = g.f[i];
            if (f.state == 2) { done++; continue; }
            if (f.state == 1) { if (*f.wait_gen == f.my_gen) continue; f.state = 0; }
            g.cur = i; ran++;
            swapcontext(&g.main, &f.ctx);
        }
        if (done == nthreads) break;
        if (!ran) { g_grp = prev; throw std::runtime_error("emu: deadlock (divergent control flow around a cross-lane primitive?)"); }
    }
    g_grp = prev;
}

}  // namespace emu

// ---- the primitives of sonde_rs_dev.h
#define SONDE_RS_EMU 1
#define RSW_DEV inline
#define RSW_DEV_NOINLINE inline
static inline int rsw_bcast(int v, int src) {
    emu::Wave &w = emu::my_wave(); w.buf[emu::tid() & 63] = v; emu::wave_rendezvous();
    const int r = (int)w.buf[src & 63]; emu::wave_rendezvous(); return r;
}
static inline int rsw_shfl_up(int v, int d, int lane) {
    emu::Wave &w = emu::my_wave(); w.buf[lane] = v; emu::wave_rendezvous();
    const int r = lane >= d ? (int)w.buf[lane - d] : 0; emu::wave_rendezvous(); return r;
}
static inline unsigned long long rsw_ballot(bool p) {
    emu::Wave &w = emu::my_wave(); w.buf[emu::tid() & 63] = p ? 1 : 0; emu::wave_rendezvous();
    unsigned long long m = 0; for (int i = 0; i < 64; i++) m |= (unsigned long long)(w.buf[i] & 1) << i;
    emu::wave_rendezvous(); return m;
}
static inline void rsw_wave_sync() { emu::wave_rendezvous(); }
static inline void rsw_syncthreads() { emu::Group *g = emu::g_grp; emu::rendezvous(g->wg_gen, g->wg_count, g->n); }
static inline int rsw_clzll(unsigned long long m) { return __builtin_clzll(m); }
static inline int rsw_popcll(unsigned long long m) { return __builtin_popcountll(m); }
#endif
