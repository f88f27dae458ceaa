/*
 * host/sonde_broker.c — one resident GPU engine for all decoder shims of a machine (C).
 *
 * auto_rx runs one decoder pipeline per sonde (auto_rx/autorx/decode.py:1489-1529); started on their own, N `fsk_demod` shims mean N
 * HIP contexts, N engines of one channel and N launches per modem frame.  The broker keeps ONE context: clients connect over a UNIX
 * socket (SONDE_BROKER=<path> in the shim's environment, host/broker_proto.h), get a channel of a batched engine whose configuration
 * equals theirs (a "group": Fs, Rs, M, P, nsym, input format, estimator limits / algorithm), and every step of a group demodulates the
 * pending frame of all its clients in one launch.  The shims keep their whole command-line contract — option parsing, the fsk_nin()
 * read loop, --stats / --testframes output stay in the client — so stdout / stderr are byte for byte those of a stand-alone run.
 *
 *     sonde_broker --socket /run/sonde.sock [--slots 64] [--device 0] [--window-us 2000] [--stall-ms 1000] [--idle-exit]
 *
 * The decoder shims themselves (rs41mod / dfm09mod / m10mod / m20mod on FM audio or IF-rate IQ: what decode.py:375-417 pipes into them) work
 * the same way: a group is an engine configuration, a client owns a channel, every step feeds all channels the same number of samples of
 * their pending input blocks in one sonde_engine_process_host() call and routes the frame records back.  A channel is ended with
 * sonde_engine_finish_channel() when its client reaches EOF and re-armed with sonde_engine_restart_channel() for the next client.
 * A decoder group steps when all its clients have a block pending.  A client whose input pauses for longer than --stall-ms (1000) while
 * others wait is parked, not dropped: its channel is ended like at EOF (the frame in progress is delivered with the bits that exist) and
 * starts over with its next block, so one SDR hiccup neither stalls the other decoders nor ends that sonde's decoder process.
 *
 * Step policy of a modem group: run as soon as every connected client has a frame pending; otherwise when the oldest pending frame has
 * waited --window-us (clients whose frame is not there yet simply sit the step out: their channel is fed 0 samples).  A slot freed by
 * a disconnect is reset to the fsk_create_hbr() state before it is handed out again.  SIGTERM / SIGINT: statistics line on stderr
 * (`broker: groups G clients C steps S frames F max_batch B`), socket removed, exit 0.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <poll.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <unistd.h>
#include <sys/socket.h>
#include <sys/un.h>
#include "broker_proto.h"
#include "sonde_hip.h"

#define MAX_CLIENTS 1024
#define MAX_GROUPS  16

typedef struct {
    int fd;                              /* -1 = unused */
    int kind;                            /* BRK_KIND_FSK / BRK_KIND_DEMOD once HELLO has been seen */
    int group, slot;                     /* -1 until HELLO */
    unsigned char *rx; size_t rx_len, rx_cap;
    int pending;                         /* a complete DATA message sits at the head of rx */
    int64_t pending_since_us;
    /* decoder shims: progress inside the pending block and the records collected for the reply */
    uint32_t blk_off;
    unsigned char *out; size_t out_len, out_cap; uint32_t out_count;
    int64_t active_us;                   /* last time the client sent something or was answered */
    int got_data;                        /* decoder shim: it has delivered samples since it took (or re-took) its channel */
    int parked;                          /* decoder shim whose input paused while its peers waited: its channel has been ended (finish_channel) and
                                          * is restarted with its next block; the group steps without it meanwhile */
} client_t;

typedef struct {
    int used;
    sonde_fsk_cfg_t key;
    sonde_fsk_t *eng;
    sonde_fsk_info_t info;
    size_t unit;                         /* bytes per input sample */
    int *slot_client;                    /* [slots] client index or -1 */
    int *slot_dirty;                     /* freed by a disconnect: reset before reuse */
    int n_clients;
    float *sd; uint8_t *bits; float *Sf;
} group_t;

typedef struct {                         /* one engine configuration of decoder shims */
    int used;
    brk_hello_demod_t key;
    sonde_engine_t *eng;
    sonde_info_t info;
    size_t unit, rec_size;               /* bytes per input sample (all components) / per frame record */
    int max_chunk;
    int *slot_client, *slot_dirty;
    int n_clients;
    long calls;                          /* process calls so far: 0 = the engine is still at its origin */
    unsigned char *stage;                /* [slots][max_chunk] samples of one step */
    unsigned char *recs;                 /* fetch buffer */
} dgroup_t;

static client_t g_cl[MAX_CLIENTS];
static group_t g_gr[MAX_GROUPS];
static dgroup_t g_dg[MAX_GROUPS];
static long g_dsteps = 0, g_drecs = 0, g_parked = 0; static int g_dmax_batch = 0;
static int g_slots = 64, g_device = 0, g_idle_exit = 0;
static int64_t g_window_us = 2000, g_stall_us = 1000000;
static volatile sig_atomic_t g_stop = 0;
static long g_steps = 0, g_frames = 0, g_served = 0; static int g_max_batch = 0;

static int64_t now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (int64_t)t.tv_sec * 1000000 + t.tv_nsec / 1000; }
static void on_signal(int s) { (void)s; g_stop = 1; }

static int send_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) { ssize_t k = send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return -1; } c += k; n -= (size_t)k; }
    return 0;
}
static int send_msg(int fd, uint32_t type, const void *a, size_t na, const void *b, size_t nb, const void *c, size_t nc, const void *d, size_t nd) {
    brk_hdr_t h = { BRK_MAGIC, type, (uint32_t)(na + nb + nc + nd) };
    if (send_all(fd, &h, sizeof h)) return -1;
    if (na && send_all(fd, a, na)) return -1;
    if (nb && send_all(fd, b, nb)) return -1;
    if (nc && send_all(fd, c, nc)) return -1;
    if (nd && send_all(fd, d, nd)) return -1;
    return 0;
}
static void send_error(int fd, const char *text) { send_msg(fd, BRK_ERROR, text, strlen(text) + 1, NULL, 0, NULL, 0, NULL, 0); }

static void drop_client(int ci) {
    client_t *c = &g_cl[ci];
    if (c->fd < 0) return;
    close(c->fd); c->fd = -1;
    if (c->group >= 0 && c->kind == BRK_KIND_FSK) {
        group_t *g = &g_gr[c->group];
        g->slot_client[c->slot] = -1; g->slot_dirty[c->slot] = 1; g->n_clients--;
    }
    if (c->group >= 0 && c->kind == BRK_KIND_DEMOD) {
        dgroup_t *g = &g_dg[c->group];
        g->slot_client[c->slot] = -1; g->slot_dirty[c->slot] = 1; g->n_clients--;
    }
    free(c->rx); c->rx = NULL; c->rx_len = c->rx_cap = 0; c->pending = 0; c->group = c->slot = -1; c->kind = 0;
    free(c->out); c->out = NULL; c->out_len = c->out_cap = 0; c->out_count = 0; c->blk_off = 0; c->parked = 0;
}

/* the fields that make two clients batchable */
static int same_cfg(const sonde_fsk_cfg_t *a, const sonde_fsk_cfg_t *b) {
    return a->Fs == b->Fs && a->Rs == b->Rs && a->M == b->M && a->P == b->P && a->nsym == b->nsym && a->format == b->format &&
           a->fsk_lower == b->fsk_lower && a->fsk_upper == b->fsk_upper && a->mask == b->mask && a->tone_spacing == b->tone_spacing &&
           a->burst_mode == b->burst_mode && a->raw_eye == b->raw_eye;
}

static int join_group(int ci, const sonde_fsk_cfg_t *want) {
    int gi = -1;
    for (int i = 0; i < MAX_GROUPS; i++) if (g_gr[i].used && same_cfg(&g_gr[i].key, want)) { gi = i; break; }
    if (want->abi_version != SONDE_ABI_VERSION) { send_error(g_cl[ci].fd, "broker: the shim was built against another ABI version of libsonde_hip"); return -1; }
    if (gi < 0) {
        for (int i = 0; i < MAX_GROUPS; i++) if (!g_gr[i].used) { gi = i; break; }
        if (gi < 0) { send_error(g_cl[ci].fd, "broker: too many modem configurations"); return -1; }
        group_t *g = &g_gr[gi];
        memset(g, 0, sizeof *g);
        g->key = *want;
        g->key.abi_version = SONDE_ABI_VERSION; g->key.device = g_device; g->key.n_channels = g_slots;
        if (want->Fs < 1 || want->Rs < 1 || want->Fs % want->Rs || want->nsym < 1) { send_error(g_cl[ci].fd, "broker: invalid modem configuration"); return -1; }
        g->key.max_chunk = (want->Fs / want->Rs) * (want->nsym + 2);
        const int rc = sonde_fsk_create(&g->key, &g->eng);
        if (rc < 0) { send_error(g_cl[ci].fd, sonde_strerror(rc)); return -1; }
        sonde_fsk_info(g->eng, &g->info);
        g->unit = want->format == SONDE_FSK_CF32 ? 8 : want->format == SONDE_FSK_CS16 ? 4 : 2;
        g->slot_client = (int *)malloc(sizeof(int) * (size_t)g_slots); g->slot_dirty = (int *)calloc((size_t)g_slots, sizeof(int));
        g->sd = (float *)malloc(sizeof(float) * (size_t)g->info.Nbits); g->bits = (uint8_t *)malloc((size_t)g->info.Nbits);
        g->Sf = (float *)malloc(sizeof(float) * (size_t)g->info.Ndft);
        if (!g->slot_client || !g->slot_dirty || !g->sd || !g->bits || !g->Sf) {
            free(g->slot_client); free(g->slot_dirty); free(g->sd); free(g->bits); free(g->Sf); sonde_fsk_destroy(g->eng); memset(g, 0, sizeof *g);
            send_error(g_cl[ci].fd, "broker: out of memory"); return -1;
        }
        for (int s = 0; s < g_slots; s++) g->slot_client[s] = -1;
        g->used = 1;
    }
    group_t *g = &g_gr[gi];
    int slot = -1;
    for (int s = 0; s < g_slots; s++) if (g->slot_client[s] < 0) { slot = s; break; }
    if (slot < 0) { send_error(g_cl[ci].fd, "broker: all channels of this configuration are taken"); return -1; }
    if (g->slot_dirty[slot]) { sonde_fsk_reset_channel(g->eng, slot); g->slot_dirty[slot] = 0; }
    g->slot_client[slot] = ci; g->n_clients++;
    g_cl[ci].group = gi; g_cl[ci].slot = slot; g_cl[ci].kind = BRK_KIND_FSK;
    g_served++;
    return send_msg(g_cl[ci].fd, BRK_INFO, &g->info, sizeof g->info, NULL, 0, NULL, 0, NULL, 0);
}

static void consume(client_t *c, size_t n);

/* ---- decoder shims ---------------------------------------------------------------------------------------------------------- */

static size_t demod_rec_size(int type) {
    return type == SONDE_RS41 ? sizeof(sonde_frame_t) : type == SONDE_DFM09 ? sizeof(sonde_dfm_frame_t) : type == SONDE_M10 ? sizeof(sonde_m10_frame_t) :
           type == SONDE_M20 ? sizeof(sonde_m20_frame_t) : 0;
}

static int same_demod(const brk_hello_demod_t *a, const brk_hello_demod_t *b) {
    const sonde_cfg_t *x = &a->cfg, *y = &b->cfg;
    return x->sample_rate == y->sample_rate && x->bits == y->bits && x->sonde_type == y->sonde_type && x->opt_lp == y->opt_lp && x->opt_min == y->opt_min &&
           x->lpiq_bw == y->lpiq_bw && x->ecc_level == y->ecc_level && x->thres == y->thres && x->input == y->input && x->audio_channels == y->audio_channels &&
           x->audio_select == y->audio_select && x->opt_inv == y->opt_inv && x->m10_noskip == y->m10_noskip && x->opt_auto == y->opt_auto &&
           x->if_rate == y->if_rate && x->if_tune == y->if_tune &&
           a->set_sync == b->set_sync && (!a->set_sync || (a->hdmax == b->hdmax && a->bitofs == b->bitofs));
}

static int join_dgroup(int ci, const brk_hello_demod_t *want) {
    const sonde_cfg_t *w = &want->cfg;
    const size_t rs = demod_rec_size(w->sonde_type);
    if (w->abi_version != SONDE_ABI_VERSION) { send_error(g_cl[ci].fd, "broker: the shim was built against another ABI version of libsonde_hip"); return -1; }
    if ((w->bits != 8 && w->bits != 16 && w->bits != 32) || (w->input == SONDE_IN_AUDIO && w->audio_channels != 1 && w->audio_channels != 2) ||
        (w->input == SONDE_IN_AUDIO && (w->audio_select < 0 || w->audio_select >= w->audio_channels))) {
        send_error(g_cl[ci].fd, "broker: invalid sample format (bits 8 / 16 / 32, one or two audio channels)"); return -1;
    }
    if (!rs || w->input == SONDE_IN_IQ || w->opt_dc || w->opt_iqdc || w->opt_nolut || w->keep_soft || w->sample_rate < 1000 || w->sample_rate > 4000000) {
        send_error(g_cl[ci].fd, "broker: this decoder configuration is not served (FM audio / IF-rate IQ of rs41mod, dfm09mod, m10mod, m20mod without --dc / --iqdc)");
        return -1;
    }
    int gi = -1;
    for (int i = 0; i < MAX_GROUPS; i++) if (g_dg[i].used && same_demod(&g_dg[i].key, want)) { gi = i; break; }
    if (gi < 0) {
        for (int i = 0; i < MAX_GROUPS; i++) if (!g_dg[i].used) { gi = i; break; }
        if (gi < 0) { send_error(g_cl[ci].fd, "broker: too many decoder configurations"); return -1; }
        dgroup_t *g = &g_dg[gi];
        memset(g, 0, sizeof *g);
        g->key = *want;
        sonde_cfg_t *c = &g->key.cfg;
        c->abi_version = SONDE_ABI_VERSION; c->device = g_device; c->n_channels = g_slots; c->max_chunk = c->sample_rate; c->max_frames = 16 * g_slots;
        c->pipeline = 0; c->keep_soft = 0;
        double *fq = (double *)calloc((size_t)g_slots, sizeof(double));
        if (!fq) { send_error(g_cl[ci].fd, "broker: out of memory"); return -1; }
        int rc = sonde_engine_create(c, fq, &g->eng);
        free(fq);
        if (rc >= 0 && want->set_sync) rc = sonde_engine_set_sync(g->eng, want->hdmax, want->bitofs);
        if (rc < 0) { if (g->eng) sonde_engine_destroy(g->eng); send_error(g_cl[ci].fd, sonde_strerror(rc)); return -1; }
        sonde_engine_info(g->eng, &g->info);
        g->unit = (size_t)(c->input == SONDE_IN_AUDIO ? c->audio_channels : 2) * (size_t)(c->bits / 8);
        g->rec_size = rs; g->max_chunk = c->max_chunk;
        g->slot_client = (int *)malloc(sizeof(int) * (size_t)g_slots); g->slot_dirty = (int *)calloc((size_t)g_slots, sizeof(int));
        g->stage = (unsigned char *)calloc((size_t)g_slots * (size_t)g->max_chunk, g->unit);
        g->recs = (unsigned char *)malloc(rs * 256);
        if (!g->slot_client || !g->slot_dirty || !g->stage || !g->recs || !g->unit) {
            free(g->slot_client); free(g->slot_dirty); free(g->stage); free(g->recs); sonde_engine_destroy(g->eng); memset(g, 0, sizeof *g);
            send_error(g_cl[ci].fd, "broker: out of memory"); return -1;
        }
        for (int s2 = 0; s2 < g_slots; s2++) g->slot_client[s2] = -1;
        g->used = 1;
    }
    dgroup_t *g = &g_dg[gi];
    int slot = -1;
    for (int s2 = 0; s2 < g_slots; s2++) if (g->slot_client[s2] < 0) { slot = s2; break; }
    if (slot < 0) { send_error(g_cl[ci].fd, "broker: all channels of this configuration are taken"); return -1; }
    /* a channel that carried another stream, or sat idle while the engine ran, starts over: from here on it is channel 0 of a fresh engine */
    if (g->slot_dirty[slot] || g->calls > 0) {
        const int rc = sonde_engine_restart_channel(g->eng, slot);
        if (rc < 0) { send_error(g_cl[ci].fd, sonde_strerror(rc)); return -1; }
        g->slot_dirty[slot] = 0;
    }
    g->slot_client[slot] = ci; g->n_clients++;
    g_cl[ci].group = gi; g_cl[ci].slot = slot; g_cl[ci].kind = BRK_KIND_DEMOD; g_cl[ci].blk_off = 0;
    g_served++;
    return send_msg(g_cl[ci].fd, BRK_INFO, &g->info, sizeof g->info, NULL, 0, NULL, 0, NULL, 0);
}

/* frame records the engine has ready -> the outboxes of the clients that own the channels */
static void route_records(dgroup_t *g) {
    for (;;) {
        int k;
        const int t = g->key.cfg.sonde_type;
        if (t == SONDE_RS41) k = sonde_engine_fetch_frames(g->eng, (sonde_frame_t *)g->recs, 256);
        else if (t == SONDE_DFM09) k = sonde_engine_fetch_dfm(g->eng, (sonde_dfm_frame_t *)g->recs, 256, 0);
        else if (t == SONDE_M10) k = sonde_engine_fetch_m10(g->eng, (sonde_m10_frame_t *)g->recs, 256, 0);
        else k = sonde_engine_fetch_m20(g->eng, (sonde_m20_frame_t *)g->recs, 256, 0);
        if (k <= 0) return;
        for (int i = 0; i < k; i++) {
            unsigned char *r = g->recs + (size_t)i * g->rec_size;
            int32_t ch; memcpy(&ch, r, sizeof ch);
            if (ch < 0 || ch >= g_slots || g->slot_client[ch] < 0) continue;          /* a channel nobody owns (silence): nothing to report */
            client_t *c = &g_cl[g->slot_client[ch]];
            const int32_t zero = 0; memcpy(r, &zero, sizeof zero);                     /* the client sees itself as channel 0 */
            if (c->out_cap - c->out_len < g->rec_size) {
                const size_t cap = c->out_cap ? 2 * c->out_cap : 16 * g->rec_size;
                unsigned char *p = (unsigned char *)realloc(c->out, cap);
                if (!p) continue;                                                     /* out of memory: this record is lost, the client stays */
                c->out = p; c->out_cap = cap;
            }
            memcpy(c->out + c->out_len, r, g->rec_size); c->out_len += g->rec_size; c->out_count++; g_drecs++;
        }
        /* a short batch does not mean an empty queue (the DFM fetch takes at most 32 hits per call and a partial hit has fewer than 8 frames):
         * go on until a fetch returns nothing */
    }
}

static void parse_client(int ci);

/* the pending block of a client has been consumed: end its stream if it said so, send the records collected for it, look at its next message */
static void reply_dclient(dgroup_t *g, int s2) {
    const int ci = g->slot_client[s2];
    client_t *c = &g_cl[ci];
    brk_hdr_t h; brk_data_t d;
    memcpy(&h, c->rx, sizeof h); memcpy(&d, c->rx + sizeof h, sizeof d);
    if (d.want_stats & BRK_FINISH) { sonde_engine_finish_channel(g->eng, s2); route_records(g); g->slot_dirty[s2] = 1; }
    brk_dresult_t r = { c->out_count, (uint32_t)g->rec_size };
    const int bad = send_msg(c->fd, BRK_RESULT, &r, sizeof r, c->out, c->out_len, NULL, 0, NULL, 0);
    c->out_len = 0; c->out_count = 0; c->blk_off = 0; c->pending = 0; c->active_us = now_us();
    consume(c, sizeof h + h.length);
    if (bad) { drop_client(ci); return; }
    parse_client(ci);
}

/* one process call for the pending blocks of all clients of the group: n = what every one of them still has */
/* a client with nothing left in its block (an empty end-of-stream message) is answered at once, whatever its peers are doing: its channel must
 * not see the samples — silence — of a call made for the others, and its decoder must not wait for them */
static void answer_empty_blocks(dgroup_t *g) {
    for (int s2 = 0; s2 < g_slots; s2++) {
        const int ci = g->slot_client[s2];
        if (ci < 0 || !g_cl[ci].pending) continue;
        brk_data_t d; memcpy(&d, g_cl[ci].rx + sizeof(brk_hdr_t), sizeof d);
        if (d.n_samples == g_cl[ci].blk_off) reply_dclient(g, s2);
    }
}

static void step_dgroup(dgroup_t *g) {
    answer_empty_blocks(g);
    uint32_t n = 0; int batch = 0, active = 0;
    for (int s2 = 0; s2 < g_slots; s2++) {
        const int ci = g->slot_client[s2];
        if (ci < 0 || g_cl[ci].parked) continue;
        active++;
        if (!g_cl[ci].pending) continue;
        brk_data_t d; memcpy(&d, g_cl[ci].rx + sizeof(brk_hdr_t), sizeof d);
        const uint32_t left = d.n_samples - g_cl[ci].blk_off;
        if (left == 0) continue;
        if (!batch || left < n) n = left;
        batch++;
    }
    if (!batch || batch != active) return;                   /* somebody's next block is not here yet */
    const size_t row = (size_t)g->max_chunk * g->unit;
    for (int s2 = 0; s2 < g_slots; s2++) {
        unsigned char *dst = g->stage + (size_t)s2 * row;
        const int ci = g->slot_client[s2];
        if (ci >= 0 && !g_cl[ci].parked) memcpy(dst, g_cl[ci].rx + sizeof(brk_hdr_t) + sizeof(brk_data_t) + (size_t)g_cl[ci].blk_off * g->unit, (size_t)n * g->unit);
        else memset(dst, 0, (size_t)n * g->unit);             /* nobody there, or a parked (ended) channel: silence */
    }
    const int rc = sonde_engine_process_host(g->eng, g->stage, g->max_chunk, (int32_t)n);
    g_dsteps++; g->calls++; if (batch > g_dmax_batch) g_dmax_batch = batch;
    if (rc < 0) {
        for (int s2 = 0; s2 < g_slots; s2++) { const int ci = g->slot_client[s2]; if (ci >= 0) { send_error(g_cl[ci].fd, sonde_strerror(rc)); drop_client(ci); } }
        return;
    }
    route_records(g);
    for (int s2 = 0; s2 < g_slots; s2++) {
        const int ci = g->slot_client[s2];
        if (ci < 0 || g_cl[ci].parked) continue;
        g_cl[ci].blk_off += n;
        brk_data_t d; memcpy(&d, g_cl[ci].rx + sizeof(brk_hdr_t), sizeof d);
        if (g_cl[ci].blk_off >= d.n_samples) reply_dclient(g, s2);   /* otherwise the rest of its block goes into the next call */
    }
}

/* 1 = a complete message is at the head of rx */
static int have_message(const client_t *c, brk_hdr_t *h) {
    if (c->rx_len < sizeof *h) return 0;
    memcpy(h, c->rx, sizeof *h);
    return c->rx_len >= sizeof *h + h->length;
}
static void consume(client_t *c, size_t n) { memmove(c->rx, c->rx + n, c->rx_len - n); c->rx_len -= n; }

static void parse_client(int ci) {
    client_t *c = &g_cl[ci];
    brk_hdr_t h;
    while (c->fd >= 0 && !c->pending && have_message(c, &h)) {
        if (h.magic != BRK_MAGIC) { drop_client(ci); return; }
        c->active_us = now_us();
        uint32_t hello_kind = 0;
        if (h.type == BRK_HELLO && h.length >= sizeof hello_kind) memcpy(&hello_kind, c->rx + sizeof h, sizeof hello_kind);
        if (h.type == BRK_HELLO && c->group < 0 && hello_kind == BRK_KIND_FSK && h.length == sizeof(brk_hello_t)) {
            brk_hello_t hello; memcpy(&hello, c->rx + sizeof h, sizeof hello);
            consume(c, sizeof h + h.length);
            if (join_group(ci, &hello.fsk) < 0) { drop_client(ci); return; }
        } else if (h.type == BRK_HELLO && c->group < 0 && hello_kind == BRK_KIND_DEMOD && h.length == sizeof(brk_hello_demod_t)) {
            brk_hello_demod_t hello; memcpy(&hello, c->rx + sizeof h, sizeof hello);
            consume(c, sizeof h + h.length);
            if (join_dgroup(ci, &hello) < 0) { drop_client(ci); return; }
        } else if (h.type == BRK_DATA && c->group >= 0 && c->kind == BRK_KIND_DEMOD && h.length >= sizeof(brk_data_t)) {
            brk_data_t d; memcpy(&d, c->rx + sizeof h, sizeof d);
            const dgroup_t *g = &g_dg[c->group];
            if (h.length != sizeof d + (size_t)d.n_samples * g->unit || (int)d.n_samples > g->max_chunk) { send_error(c->fd, "broker: malformed DATA"); drop_client(ci); return; }
            if (c->parked) {                                  /* its input is back: the channel starts over with this block */
                const int rc = sonde_engine_restart_channel(g->eng, c->slot);
                if (rc < 0) { send_error(c->fd, sonde_strerror(rc)); drop_client(ci); return; }
                c->parked = 0;
            }
            if (d.n_samples) c->got_data = 1;
            c->pending = 1; c->pending_since_us = now_us(); c->blk_off = 0;
        } else if (h.type == BRK_DATA && c->group >= 0 && c->kind == BRK_KIND_FSK && h.length >= sizeof(brk_data_t)) {
            brk_data_t d; memcpy(&d, c->rx + sizeof h, sizeof d);
            const group_t *g = &g_gr[c->group];
            if (h.length != sizeof d + (size_t)d.n_samples * g->unit || (int)d.n_samples > g->key.max_chunk) { send_error(c->fd, "broker: malformed DATA"); drop_client(ci); return; }
            c->pending = 1; c->pending_since_us = now_us();
        } else { send_error(c->fd, "broker: unexpected message"); drop_client(ci); return; }
    }
}

static void read_client(int ci) {
    client_t *c = &g_cl[ci];
    if (c->rx_cap - c->rx_len < 65536) {
        const size_t cap = c->rx_cap ? c->rx_cap * 2 : 262144;
        unsigned char *p = (unsigned char *)realloc(c->rx, cap);
        if (!p) { send_error(c->fd, "broker: out of memory"); drop_client(ci); return; }
        c->rx = p; c->rx_cap = cap;
    }
    const ssize_t k = recv(c->fd, c->rx + c->rx_len, c->rx_cap - c->rx_len, MSG_DONTWAIT);
    if (k == 0 || (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) { drop_client(ci); return; }
    if (k > 0) c->rx_len += (size_t)k;
    if (c->rx_len >= sizeof(brk_hdr_t)) {                    /* a header that cannot be ours, or a message no client sends: do not buffer it */
        brk_hdr_t h; memcpy(&h, c->rx, sizeof h);
        if (h.magic != BRK_MAGIC || h.length > (48u << 20)) { drop_client(ci); return; }
    }
    parse_client(ci);
}

/* one launch for every pending frame of the group */
static void step_group(group_t *g) {
    static const void *in[4096]; static int32_t n[4096];
    int batch = 0;
    for (int s = 0; s < g_slots; s++) {
        in[s] = NULL; n[s] = 0;
        const int ci = g->slot_client[s];
        if (ci < 0 || !g_cl[ci].pending) continue;
        brk_data_t d; memcpy(&d, g_cl[ci].rx + sizeof(brk_hdr_t), sizeof d);
        in[s] = g_cl[ci].rx + sizeof(brk_hdr_t) + sizeof d; n[s] = (int32_t)d.n_samples;
        batch++;
    }
    if (!batch) return;
    const int rc = sonde_fsk_process_host_var(g->eng, in, n);
    g_steps++; if (batch > g_max_batch) g_max_batch = batch;
    for (int s = 0; s < g_slots; s++) {
        const int ci = g->slot_client[s];
        if (ci < 0 || !g_cl[ci].pending) continue;
        client_t *c = &g_cl[ci];
        brk_hdr_t h; brk_data_t d;
        memcpy(&h, c->rx, sizeof h); memcpy(&d, c->rx + sizeof h, sizeof d);
        c->pending = 0;
        consume(c, sizeof h + h.length);
        if (rc < 0) { send_error(c->fd, sonde_strerror(rc)); drop_client(ci); continue; }
        brk_result_t r; memset(&r, 0, sizeof r);
        int32_t nf = 0;
        const int nb = sonde_fsk_fetch(g->eng, s, g->sd, g->info.Nbits, &r.frame, 1, &nf);
        if (nf == 0) {                   /* fewer samples than fsk_nin(): nothing came out (a well-behaved client never does this) */
            r.nbits = 0;
            if (send_msg(c->fd, BRK_RESULT, &r, sizeof r, NULL, 0, NULL, 0, NULL, 0)) drop_client(ci);
            continue;
        }
        sonde_fsk_fetch_bits(g->eng, s, g->bits, g->info.Nbits);
        r.nbits = (uint32_t)nb; g_frames++;
        brk_eye_t eye; size_t nsf = 0, neye = 0;
        sonde_fsk_stats(g->eng, s, NULL, d.want_stats ? g->Sf : NULL, &r.samples);
        if (d.want_stats) {
            memset(&eye, 0, sizeof eye);
            sonde_fsk_eye(g->eng, s, eye.eye, &eye.neyetr, &eye.neyesamp);
            r.has_stats = 1; nsf = sizeof(float) * (size_t)g->info.Ndft; neye = sizeof eye;
        }
        if (send_msg(c->fd, BRK_RESULT, &r, sizeof r, g->sd, sizeof(float) * (size_t)nb, g->bits, (size_t)nb, g->Sf, nsf) ||
            (neye && send_all(c->fd, &eye, neye))) { drop_client(ci); continue; }
        parse_client(ci);                /* the next DATA may already be buffered */
    }
}

int main(int argc, char **argv) {
    const char *path = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--socket") && i + 1 < argc) path = argv[++i];
        else if (!strcmp(argv[i], "--slots") && i + 1 < argc) g_slots = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) g_device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--window-us") && i + 1 < argc) g_window_us = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--stall-ms") && i + 1 < argc) g_stall_us = 1000 * atoll(argv[++i]);
        else if (!strcmp(argv[i], "--idle-exit")) g_idle_exit = 1;
        else { fprintf(stderr, "usage: %s --socket <path> [--slots 64] [--device 0] [--window-us 2000] [--stall-ms 1000] [--idle-exit]\n", argv[0]); return 1; }
    }
    if (!path || g_slots < 1 || g_slots > 4096) { fprintf(stderr, "sonde_broker: --socket <path> and 1 <= --slots <= 4096 required\n"); return 1; }
    for (int i = 0; i < MAX_CLIENTS; i++) { g_cl[i].fd = -1; g_cl[i].group = g_cl[i].slot = -1; }
    struct sockaddr_un addr; memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (strlen(path) >= sizeof addr.sun_path) { fprintf(stderr, "sonde_broker: socket path too long\n"); return 1; }
    strcpy(addr.sun_path, path);
    unlink(path);
    const int lfd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (lfd < 0 || bind(lfd, (struct sockaddr *)&addr, sizeof addr) < 0 || listen(lfd, 256) < 0) { perror("sonde_broker: socket"); return 1; }
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = on_signal;
    sigaction(SIGTERM, &sa, NULL); sigaction(SIGINT, &sa, NULL);
    fprintf(stderr, "broker: listening on %s, %d channels per modem configuration\n", path, g_slots);

    static struct pollfd pf[MAX_CLIENTS + 1]; static int map[MAX_CLIENTS + 1];
    int had_clients = 0;
    while (!g_stop) {
        int np = 0, live = 0;
        pf[np].fd = lfd; pf[np].events = POLLIN; map[np++] = -1;
        int64_t deadline = -1;
        for (int i = 0; i < MAX_CLIENTS; i++) {
            if (g_cl[i].fd < 0) continue;
            live++;
            if (g_cl[i].pending) {
                /* modem frames wait at most --window-us for their peers; a decoder block waits for its peers' sockets to become readable
                 * (poll blocks) and at most --stall-ms before the late ones are parked */
                const int64_t dl = g_cl[i].pending_since_us + (g_cl[i].kind == BRK_KIND_FSK ? g_window_us : g_stall_us);
                if (deadline < 0 || dl < deadline) deadline = dl;
                continue;
            }
            pf[np].fd = g_cl[i].fd; pf[np].events = POLLIN; map[np++] = i;
        }
        if (live) had_clients = 1;
        if (g_idle_exit && had_clients && !live) break;
        int tmo = 200;
        if (deadline >= 0) { const int64_t left = deadline - now_us(); tmo = left <= 0 ? 0 : (int)((left + 999) / 1000); }
        const int pr = poll(pf, (nfds_t)np, tmo);
        if (pr < 0 && errno != EINTR) break;
        if (pr > 0) {
            if (pf[0].revents & POLLIN) {
                const int fd = accept(lfd, NULL, NULL);
                if (fd >= 0) {
                    int ci = -1;
                    for (int i = 0; i < MAX_CLIENTS; i++) if (g_cl[i].fd < 0) { ci = i; break; }
                    if (ci < 0) { send_error(fd, "broker: too many clients"); close(fd); }
                    else { memset(&g_cl[ci], 0, sizeof g_cl[ci]); g_cl[ci].fd = fd; g_cl[ci].group = g_cl[ci].slot = -1; g_cl[ci].active_us = now_us(); }
                }
            }
            for (int k = 1; k < np; k++) if (pf[k].revents & (POLLIN | POLLHUP | POLLERR)) read_client(map[k]);
        }
        /* run every group whose clients are all waiting, or whose oldest frame has waited long enough */
        const int64_t t = now_us();
        for (int gi = 0; gi < MAX_GROUPS; gi++) {
            group_t *g = &g_gr[gi];
            if (!g->used || !g->n_clients) continue;
            int pend = 0; int64_t oldest = -1;
            for (int s = 0; s < g_slots; s++) {
                const int ci = g->slot_client[s];
                if (ci < 0 || !g_cl[ci].pending) continue;
                pend++;
                if (oldest < 0 || g_cl[ci].pending_since_us < oldest) oldest = g_cl[ci].pending_since_us;
            }
            if (pend && (pend == g->n_clients || t - oldest >= g_window_us)) step_group(g);
        }
        /* decoder groups feed every channel the same number of samples per call: they step when all their (not parked) clients have input
         * pending.  A client that has sent nothing for --stall-ms while others wait is parked: its channel is ended as at EOF (the frame in
         * progress goes out with the bits that exist, with its next reply) and starts over when its next block arrives — the others go on */
        for (int gi = 0; gi < MAX_GROUPS; gi++) {
            dgroup_t *g = &g_dg[gi];
            if (!g->used || !g->n_clients) continue;
            answer_empty_blocks(g);
            int pend = 0, need = 0; int64_t oldest = -1;
            for (int s2 = 0; s2 < g_slots; s2++) {
                const int ci = g->slot_client[s2];
                if (ci < 0 || g_cl[ci].parked) continue;
                need++;
                if (!g_cl[ci].pending) continue;
                pend++;
                if (oldest < 0 || g_cl[ci].pending_since_us < oldest) oldest = g_cl[ci].pending_since_us;
            }
            if (pend && pend < need && t - oldest >= g_stall_us)
                for (int s2 = 0; s2 < g_slots; s2++) {
                    const int ci = g->slot_client[s2];
                    if (ci < 0 || g_cl[ci].pending || g_cl[ci].parked) continue;
                    if (t - g_cl[ci].active_us < g_stall_us) continue;                 /* it is the broker that was busy, not the client that is late */
                    /* a channel that has carried samples ends like at EOF; one that has not seen any yet just sits out until its first block */
                    if (g_cl[ci].got_data) { sonde_engine_finish_channel(g->eng, s2); route_records(g); }
                    g_cl[ci].parked = 1; g_cl[ci].got_data = 0; g_parked++;
                }
            for (int guard = 0; guard < 64 && g->n_clients > 0; guard++) {             /* blocks of unequal length take more than one call */
                int p2 = 0, n2 = 0;
                for (int s2 = 0; s2 < g_slots; s2++) { const int ci = g->slot_client[s2]; if (ci >= 0 && !g_cl[ci].parked) { n2++; if (g_cl[ci].pending) p2++; } }
                if (!n2 || p2 != n2) break;
                const long before = g->calls; const long served = g_drecs + g_served;
                step_dgroup(g);
                if (g->calls == before && g_drecs + g_served == served) {
                    int p3 = 0;
                    for (int s2 = 0; s2 < g_slots; s2++) { const int ci = g->slot_client[s2]; if (ci >= 0 && !g_cl[ci].parked && g_cl[ci].pending) p3++; }
                    if (p3 == p2) break;                                               /* nothing moved */
                }
            }
        }
    }
    int groups = 0, dgroups = 0;
    for (int gi = 0; gi < MAX_GROUPS; gi++) if (g_gr[gi].used) { groups++; sonde_fsk_destroy(g_gr[gi].eng); }
    for (int gi = 0; gi < MAX_GROUPS; gi++) if (g_dg[gi].used) { dgroups++; sonde_engine_destroy(g_dg[gi].eng); }
    fprintf(stderr, "broker: groups %d clients %ld steps %ld frames %ld max_batch %d\n", groups, g_served, g_steps, g_frames, g_max_batch);
    if (dgroups) fprintf(stderr, "broker: decoder_groups %d calls %ld records %ld max_batch %d parked %ld\n", dgroups, g_dsteps, g_drecs, g_dmax_batch, g_parked);
    for (int i = 0; i < MAX_CLIENTS; i++) if (g_cl[i].fd >= 0) close(g_cl[i].fd);
    close(lfd); unlink(path);
    return 0;
}
