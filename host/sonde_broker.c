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
 *     sonde_broker --socket /run/sonde.sock [--slots 64] [--device 0] [--window-us 2000] [--idle-exit]
 *
 * Step policy of a group: run as soon as every connected client has a frame pending; otherwise when the oldest pending frame has
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

#define MAX_CLIENTS 1024
#define MAX_GROUPS  16

typedef struct {
    int fd;                              /* -1 = unused */
    int group, slot;                     /* -1 until HELLO */
    unsigned char *rx; size_t rx_len, rx_cap;
    int pending;                         /* a complete DATA message sits at the head of rx */
    int64_t pending_since_us;
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

static client_t g_cl[MAX_CLIENTS];
static group_t g_gr[MAX_GROUPS];
static int g_slots = 64, g_device = 0, g_idle_exit = 0;
static int64_t g_window_us = 2000;
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
    if (c->group >= 0) {
        group_t *g = &g_gr[c->group];
        g->slot_client[c->slot] = -1; g->slot_dirty[c->slot] = 1; g->n_clients--;
    }
    free(c->rx); c->rx = NULL; c->rx_len = c->rx_cap = 0; c->pending = 0; c->group = c->slot = -1;
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
        for (int s = 0; s < g_slots; s++) g->slot_client[s] = -1;
        g->sd = (float *)malloc(sizeof(float) * (size_t)g->info.Nbits); g->bits = (uint8_t *)malloc((size_t)g->info.Nbits);
        g->Sf = (float *)malloc(sizeof(float) * (size_t)g->info.Ndft);
        g->used = 1;
    }
    group_t *g = &g_gr[gi];
    int slot = -1;
    for (int s = 0; s < g_slots; s++) if (g->slot_client[s] < 0) { slot = s; break; }
    if (slot < 0) { send_error(g_cl[ci].fd, "broker: all channels of this configuration are taken"); return -1; }
    if (g->slot_dirty[slot]) { sonde_fsk_reset_channel(g->eng, slot); g->slot_dirty[slot] = 0; }
    g->slot_client[slot] = ci; g->n_clients++;
    g_cl[ci].group = gi; g_cl[ci].slot = slot;
    g_served++;
    return send_msg(g_cl[ci].fd, BRK_INFO, &g->info, sizeof g->info, NULL, 0, NULL, 0, NULL, 0);
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
        if (h.type == BRK_HELLO && c->group < 0 && h.length == sizeof(brk_hello_t)) {
            brk_hello_t hello; memcpy(&hello, c->rx + sizeof h, sizeof hello);
            consume(c, sizeof h + h.length);
            if (hello.kind != BRK_KIND_FSK) { send_error(c->fd, "broker: unknown client kind"); drop_client(ci); return; }
            if (join_group(ci, &hello.fsk) < 0) { drop_client(ci); return; }
        } else if (h.type == BRK_DATA && c->group >= 0 && h.length >= sizeof(brk_data_t)) {
            brk_data_t d; memcpy(&d, c->rx + sizeof h, sizeof d);
            const group_t *g = &g_gr[c->group];
            if (h.length != sizeof d + (size_t)d.n_samples * g->unit || (int)d.n_samples > g->key.max_chunk) { send_error(c->fd, "broker: malformed DATA"); drop_client(ci); return; }
            c->pending = 1; c->pending_since_us = now_us();
        } else { send_error(c->fd, "broker: unexpected message"); drop_client(ci); return; }
    }
}

static void read_client(int ci) {
    client_t *c = &g_cl[ci];
    if (c->rx_cap - c->rx_len < 65536) { c->rx_cap = c->rx_cap ? c->rx_cap * 2 : 262144; c->rx = (unsigned char *)realloc(c->rx, c->rx_cap); }
    const ssize_t k = recv(c->fd, c->rx + c->rx_len, c->rx_cap - c->rx_len, MSG_DONTWAIT);
    if (k == 0 || (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) { drop_client(ci); return; }
    if (k > 0) c->rx_len += (size_t)k;
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
        else if (!strcmp(argv[i], "--idle-exit")) g_idle_exit = 1;
        else { fprintf(stderr, "usage: %s --socket <path> [--slots 64] [--device 0] [--window-us 2000] [--idle-exit]\n", argv[0]); return 1; }
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
            if (g_cl[i].pending) { const int64_t dl = g_cl[i].pending_since_us + g_window_us; if (deadline < 0 || dl < deadline) deadline = dl; continue; }
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
                    else { g_cl[ci].fd = fd; g_cl[ci].group = g_cl[ci].slot = -1; g_cl[ci].pending = 0; g_cl[ci].rx = NULL; g_cl[ci].rx_len = g_cl[ci].rx_cap = 0; }
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
    }
    int groups = 0;
    for (int gi = 0; gi < MAX_GROUPS; gi++) if (g_gr[gi].used) { groups++; sonde_fsk_destroy(g_gr[gi].eng); }
    fprintf(stderr, "broker: groups %d clients %ld steps %ld frames %ld max_batch %d\n", groups, g_served, g_steps, g_frames, g_max_batch);
    for (int i = 0; i < MAX_CLIENTS; i++) if (g_cl[i].fd >= 0) close(g_cl[i].fd);
    close(lfd); unlink(path);
    return 0;
}
