/*
 * host/broker_client.h — client side of the resident broker for the decoder shims (rs41mod, dfm09mod, m10mod, m20mod).
 *
 * With SONDE_BROKER=<socket path> in the environment a decoder that reads FM audio or IF-rate IQ (the forms auto_rx pipes into it,
 * decode.py:375-417) does not open the GPU itself: its samples go to host/sonde_broker.c, which runs the channels of all such processes in
 * one engine, and the frame records come back (host/broker_proto.h).  Everything else — options, input reading, telemetry, output — stays in
 * the decoder, so stdout is what a stand-alone run prints.  Configurations the broker does not serve (base-rate --IQ, --dc, --iqdc,
 * --ecc3/4) keep using an engine of their own.
 */
#ifndef BROKER_CLIENT_H
#define BROKER_CLIENT_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/socket.h>
#include <sys/un.h>
#include "broker_proto.h"
#include "sonde_hip.h"

typedef struct { int fd; sonde_info_t info; } brk_demod_t;

static int brk_io(int fd, void *p, size_t n, int out) {
    char *c = (char *)p;
    while (n) {
        const ssize_t k = out ? send(fd, c, n, MSG_NOSIGNAL) : recv(fd, c, n, 0);
        if (k <= 0) return -1;
        c += k; n -= (size_t)k;
    }
    return 0;
}

/* the broker serves FM-audio / IF-rate input without --dc / --iqdc; anything else runs on an engine of the process's own */
static int brk_demod_wanted(const sonde_cfg_t *cfg) {
    const char *path = getenv("SONDE_BROKER");
    if (!path || !*path) return 0;
    return cfg->input != SONDE_IN_IQ && !cfg->opt_dc && !cfg->opt_iqdc && !cfg->opt_nolut && !cfg->keep_soft;
}

/* reply header of the wanted type; an ERROR reply is printed like a failed sonde_engine_create() and returns 0 */
static uint32_t brk_reply(int fd, uint32_t want, int *ok) {
    brk_hdr_t h;
    *ok = 0;
    if (brk_io(fd, &h, sizeof h, 0) || h.magic != BRK_MAGIC) { fprintf(stderr, "error: broker connection lost\n"); return 0; }
    if (h.type == BRK_ERROR) {
        char msg[256]; const size_t n = h.length < sizeof msg ? h.length : sizeof msg - 1;
        if (brk_io(fd, msg, n, 0)) msg[0] = 0;
        msg[n] = 0;
        fprintf(stderr, "error: init buffers (%s)\n", msg);
        return 0;
    }
    if (h.type != want) { fprintf(stderr, "error: unexpected broker reply %u\n", h.type); return 0; }
    *ok = 1;
    return h.length;
}

static int brk_demod_open(brk_demod_t *b, const sonde_cfg_t *cfg, int set_sync, int hdmax, int bitofs) {
    const char *path = getenv("SONDE_BROKER");
    struct sockaddr_un addr; memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX; strncpy(addr.sun_path, path, sizeof addr.sun_path - 1);
    b->fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (b->fd < 0 || connect(b->fd, (struct sockaddr *)&addr, sizeof addr) < 0) { fprintf(stderr, "error: cannot reach the broker at %s\n", path); return -1; }
    brk_hdr_t h = { BRK_MAGIC, BRK_HELLO, sizeof(brk_hello_demod_t) };
    brk_hello_demod_t hello; memset(&hello, 0, sizeof hello);
    hello.kind = BRK_KIND_DEMOD; hello.cfg = *cfg; hello.set_sync = set_sync; hello.hdmax = hdmax; hello.bitofs = bitofs;
    if (brk_io(b->fd, &h, sizeof h, 1) || brk_io(b->fd, &hello, sizeof hello, 1)) return -1;
    int ok;
    if (brk_reply(b->fd, BRK_INFO, &ok) != sizeof b->info || !ok || brk_io(b->fd, &b->info, sizeof b->info, 0)) return -1;
    return 0;
}

/* one read of the input loop (n samples of `unit` bytes; finish = end of the stream): every record that comes back goes to emit().
 * Returns the number of records, or -1. */
static int brk_demod_feed(brk_demod_t *b, const void *buf, int n, size_t unit, int finish, size_t rec_size, void (*emit)(const void *rec)) {
    brk_hdr_t h = { BRK_MAGIC, BRK_DATA, (uint32_t)(sizeof(brk_data_t) + (size_t)n * unit) };
    brk_data_t d = { (uint32_t)n, finish ? BRK_FINISH : 0u };
    if (brk_io(b->fd, &h, sizeof h, 1) || brk_io(b->fd, &d, sizeof d, 1) || (n > 0 && brk_io(b->fd, (void *)buf, (size_t)n * unit, 1))) return -1;
    int ok;
    const uint32_t len = brk_reply(b->fd, BRK_RESULT, &ok);
    brk_dresult_t r;
    if (!ok || len < sizeof r || brk_io(b->fd, &r, sizeof r, 0) || r.rec_size != rec_size || len != sizeof r + (size_t)r.count * rec_size) return -1;
    unsigned char rec[1024];
    if (rec_size > sizeof rec) return -1;
    for (uint32_t i = 0; i < r.count; i++) { if (brk_io(b->fd, rec, rec_size, 0)) return -1; emit(rec); }
    return (int)r.count;
}

static void brk_demod_close(brk_demod_t *b) { if (b->fd >= 0) close(b->fd); b->fd = -1; }

#endif
