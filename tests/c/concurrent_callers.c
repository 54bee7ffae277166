/*
 * tests/c/concurrent_callers.c — the threading contract of the boundary (include/chd_spatial.h "every entry point is
 * thread-safe").  In the reference GetChannelId / QueryChannelIds are called from thousands of goroutines at once — every
 * entity channel's message handler (message_spatial.go:59,236,354), Notify's decision per update (spatial.go:611) — while the
 * GLOBAL channel's goroutine ticks.  Here, in plain C11 + pthreads, as cgo would issue it from many OS threads:
 *
 *   16 caller threads   chd_get_channel_ids with 1, 7, 16 (the lock-free host path) and 17, 300 points (the device path),
 *                       chd_notify_decide, chd_query_channel_ids — each result compared with the answer the SAME call gave
 *                       single-threaded before the threads started (a pure function of its arguments);
 *   1 ticker thread     chd_tick + chd_tick_fetch_segments + chd_tick_digest on a world of the same ctx, tick after tick; every
 *                       tick's record digest compared with a reference run of the same ticks made single-threaded on a second
 *                       ctx before the threads started.
 *
 * Exit code 0: no caller ever saw a wrong answer, no tick a wrong digest.  3: no HIP device.  1: a mismatch or an error.
 *   gcc -std=c11 -Wall -Wextra -Werror -pedantic -pthread -Iinclude tests/c/concurrent_callers.c -Lchanneld_amd -lchd_spatial
 */
#define _POSIX_C_SOURCE 200809L
#include <chd_spatial.h>

#include <inttypes.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N_THREADS 16
#define N_ENT 4000u
#define N_CONN 96u
#define TICKS 30
#define N_PTS 300u
#define N_Q 24u

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double uni(uint64_t *s) { return (double)(splitmix(s) >> 11) / 9007199254740992.0; }

static chd_grid_cfg grid_cfg(void) {
    chd_grid_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.grid_width = 2000.0; cfg.grid_height = 2000.0;
    cfg.world_offset_x = -8000.0; cfg.world_offset_z = -8000.0;
    cfg.grid_cols = 8; cfg.grid_rows = 8; cfg.server_cols = 4; cfg.server_rows = 2;
    cfg.server_interest_border_size = 1;
    return cfg;
}

/* the stateless inputs and their single-threaded answers */
static double pt_x[N_PTS], pt_z[N_PTS], old_x[N_PTS], old_z[N_PTS];
static uint32_t want_id[N_PTS], want_src[N_PTS], want_dst[N_PTS];
static uint8_t want_ho[N_PTS];
static chd_aoi_query qs[N_Q];
static uint32_t want_off[N_Q + 1], want_ids[N_Q * 64], want_dists[N_Q * 64];
static int32_t want_status[N_Q];

static chd_ctx *ctx;
static atomic_int stop_flag, failures;
static atomic_ullong n_calls;

#define FAIL(...) do { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); atomic_fetch_add(&failures, 1); } while (0)

static void *caller(void *arg) {
    uint64_t s = 0xC0FFEE ^ ((uint64_t)(uintptr_t)arg * 0x9E3779B97F4A7C15ull);
    static const uint32_t sizes[5] = {1, 7, 16, 17, N_PTS};
    uint32_t ids[N_PTS], src[N_PTS], dst[N_PTS];
    uint8_t ho[N_PTS];
    while (!atomic_load(&stop_flag) && !atomic_load(&failures)) {
        const uint32_t n = sizes[splitmix(&s) % 5], at = (uint32_t)(splitmix(&s) % (N_PTS - n + 1));
        switch (splitmix(&s) % 3) {
        case 0:
            if (chd_get_channel_ids(ctx, pt_x + at, pt_z + at, n, ids) != CHD_OK) { FAIL("chd_get_channel_ids: %s", chd_last_error(ctx)); break; }
            if (memcmp(ids, want_id + at, 4 * n)) FAIL("chd_get_channel_ids(%u points at %u) differs from its single-threaded answer", n, at);
            break;
        case 1:
            if (chd_notify_decide(ctx, old_x + at, old_z + at, pt_x + at, pt_z + at, n, src, dst, ho) != CHD_OK) { FAIL("chd_notify_decide: %s", chd_last_error(ctx)); break; }
            if (memcmp(src, want_src + at, 4 * n) || memcmp(dst, want_dst + at, 4 * n) || memcmp(ho, want_ho + at, n)) FAIL("chd_notify_decide(%u at %u) differs", n, at);
            break;
        default: {
            const uint32_t nq = 1 + (uint32_t)(splitmix(&s) % 4), q0 = (uint32_t)(splitmix(&s) % (N_Q - nq + 1));
            uint32_t off[5], qi[4 * 64], qd[4 * 64];
            int32_t st[4];
            if (chd_query_channel_ids(ctx, qs + q0, nq, NULL, NULL, NULL, 0, off, qi, qd, NULL, 4 * 64, st) != CHD_OK) { FAIL("chd_query_channel_ids: %s", chd_last_error(ctx)); break; }
            for (uint32_t k = 0; k < nq; k++) {
                const uint32_t m = want_off[q0 + k + 1] - want_off[q0 + k];
                if (st[k] != want_status[q0 + k] || off[k + 1] - off[k] != m || memcmp(qi + off[k], want_ids + want_off[q0 + k], 4 * m) ||
                    memcmp(qd + off[k], want_dists + want_off[q0 + k], 4 * m))
                    FAIL("chd_query_channel_ids(query %u) differs from its single-threaded answer", q0 + k);
            }
        }
        }
        atomic_fetch_add(&n_calls, 1);
    }
    return NULL;
}

/* the world both contexts tick: frames are a pure function of the tick index */
struct frame {
    double x[N_ENT], z[N_ENT];
    chd_aoi_query q[N_CONN];
};
static void make_frame(struct frame *f, int t) {
    uint64_t s = 0xABCD0000u + (uint64_t)t;
    for (uint32_t i = 0; i < N_ENT; i++) {
        f->x[i] = (double)(float)(-8000.0 + 15999.0 * uni(&s));
        f->z[i] = (double)(float)(-8000.0 + 15999.0 * uni(&s));
    }
    memset(f->q, 0, sizeof f->q);
    for (uint32_t c = 0; c < N_CONN; c++) {
        f->q[c].shapes = CHD_SHAPE_SPHERE;
        f->q[c].sph_cx = f->x[c * 11]; f->q[c].sph_cz = f->z[c * 11]; f->q[c].sph_r = 2500.0 + 40.0 * c;
    }
}

static int world_setup(chd_ctx *c) {
    chd_world_cfg wc;
    memset(&wc, 0, sizeof wc);
    wc.max_entities = N_ENT; wc.max_subscribers = N_CONN; wc.max_records = 1u << 23;
    wc.flags = CHD_WORLD_CONN_MAJOR_EMIT | CHD_WORLD_ONE_WAVE_EMIT;
    if (chd_world_create(c, &wc) != CHD_OK) return 1;
    static uint32_t chan[N_ENT], snd[N_ENT], conn[N_CONN];
    static struct frame f0;
    make_frame(&f0, 0);
    for (uint32_t i = 0; i < N_ENT; i++) { chan[i] = 0x80000 + i; snd[i] = 1 + (i & 7); }
    for (uint32_t k = 0; k < N_CONN; k++) conn[k] = 500 + k;
    return chd_world_spawn(c, N_ENT, NULL, chan, f0.x, f0.z, NULL, snd) != CHD_OK || chd_subs_add(c, N_CONN, NULL, conn) != CHD_OK;
}

static int one_tick(chd_ctx *c, int t, chd_records_digest *d, chd_segments_out *seg) {
    static _Thread_local struct frame f;
    make_frame(&f, t);
    chd_tick_in in;
    chd_tick_out out;
    memset(&in, 0, sizeof in);
    memset(&out, 0, sizeof out);
    in.now_ns = (int64_t)t * 50000000;
    in.n_updates = N_ENT; in.upd_x = f.x; in.upd_z = f.z;
    in.n_queries = N_CONN; in.queries = f.q;
    if (chd_tick(c, &in, &out) != CHD_OK || out.overflow) return 1;
    if (seg && chd_tick_fetch_segments(c, seg) != CHD_OK) return 1;
    return chd_tick_digest(c, d, NULL) != CHD_OK;
}

static chd_records_digest want_digest[TICKS + 1];

static void *ticker(void *arg) {
    (void)arg;
    chd_segments_out seg;
    memset(&seg, 0, sizeof seg);
    seg.segments_cap = N_CONN * 64; seg.columns_cap = 10 * (N_ENT + 1024); seg.records_cap = 1u << 22;
    seg.segments = malloc(sizeof(chd_fanout_segment) * seg.segments_cap);
    seg.conn_seg_off = malloc(4 * (N_CONN + 1));
    seg.columns = malloc(4 * seg.columns_cap);
    seg.records = malloc(sizeof(chd_fanout_rec) * seg.records_cap);
    seg.conn_rec_off = malloc(8 * (N_CONN + 1));
    for (int t = 1; t <= TICKS && !atomic_load(&failures); t++) {
        chd_records_digest d;
        if (one_tick(ctx, t, &d, &seg)) { FAIL("tick %d: %s", t, chd_last_error(ctx)); break; }
        if (getenv("CHD_TEST_TRACE")) fprintf(stderr, "tick %d done, %llu stateless calls so far\n", t, (unsigned long long)atomic_load(&n_calls));
        if (d.count != want_digest[t].count || d.sum != want_digest[t].sum || d.xor_ != want_digest[t].xor_ || seg.n_records != d.count)
            FAIL("tick %d beside %d caller threads: %" PRIu64 " records, digest differs from the single-threaded run (%" PRIu64 ")", t, N_THREADS,
                 (uint64_t)d.count, (uint64_t)want_digest[t].count);
    }
    free(seg.segments); free(seg.conn_seg_off); free(seg.columns); free(seg.records); free(seg.conn_rec_off);
    atomic_store(&stop_flag, 1);
    return NULL;
}

int main(void) {
    chd_grid_cfg cfg = grid_cfg();
    chd_ctx *ref = NULL;
    int rc = chd_create(&cfg, 0, &ctx);
    if (rc == CHD_E_NO_DEVICE) { printf("no HIP device: chd_create -> CHD_E_NO_DEVICE\n"); return 3; }
    if (rc != CHD_OK || chd_create(&cfg, 0, &ref) != CHD_OK) { fprintf(stderr, "chd_create failed\n"); return 1; }

    /* single-threaded answers */
    uint64_t s = 42;
    for (uint32_t i = 0; i < N_PTS; i++) {
        pt_x[i] = -8200.0 + 16400.0 * uni(&s); pt_z[i] = -8200.0 + 16400.0 * uni(&s);   /* some outside the world */
        old_x[i] = pt_x[i] + 900.0 * (uni(&s) - 0.5); old_z[i] = pt_z[i] + 900.0 * (uni(&s) - 0.5);
    }
    if (chd_get_channel_ids(ctx, pt_x, pt_z, N_PTS, want_id) != CHD_OK || chd_notify_decide(ctx, old_x, old_z, pt_x, pt_z, N_PTS, want_src, want_dst, want_ho) != CHD_OK) return 1;
    for (uint32_t k = 0; k < N_Q; k++) {
        memset(&qs[k], 0, sizeof qs[k]);
        if (k % 3 == 0) { qs[k].shapes = CHD_SHAPE_SPHERE; qs[k].sph_cx = pt_x[k]; qs[k].sph_cz = pt_z[k]; qs[k].sph_r = 1500.0 + 300.0 * k; }
        else if (k % 3 == 1) { qs[k].shapes = CHD_SHAPE_BOX; qs[k].box_cx = pt_x[k]; qs[k].box_cz = pt_z[k]; qs[k].box_ex = 1800.0; qs[k].box_ez = 900.0 + 100.0 * k; }
        else { qs[k].shapes = CHD_SHAPE_CONE; qs[k].cone_cx = pt_x[k]; qs[k].cone_cz = pt_z[k]; qs[k].cone_dx = 0.6; qs[k].cone_dz = -0.8; qs[k].cone_r = 5000.0; qs[k].cone_cos = 0.8660254037844387; }
    }
    if (chd_query_channel_ids(ctx, qs, N_Q, NULL, NULL, NULL, 0, want_off, want_ids, want_dists, NULL, N_Q * 64, want_status) != CHD_OK) return 1;
    uint32_t hit = 0, ho = 0;
    for (uint32_t i = 0; i < N_PTS; i++) { hit += want_id[i] != 0; ho += want_ho[i]; }
    if (hit < N_PTS / 2 || hit == N_PTS || ho < 10 || want_off[N_Q] < 50) { fprintf(stderr, "degenerate inputs (%u in the world, %u handovers, %u ids)\n", hit, ho, want_off[N_Q]); return 1; }
    /* the reference run of the world, alone on its own context */
    if (world_setup(ref) || world_setup(ctx)) { fprintf(stderr, "world setup: %s\n", chd_last_error(ref)); return 1; }
    uint64_t total = 0;
    for (int t = 1; t <= TICKS; t++) {
        if (one_tick(ref, t, &want_digest[t], NULL)) { fprintf(stderr, "reference tick %d: %s\n", t, chd_last_error(ref)); return 1; }
        total += want_digest[t].count;
    }
    chd_destroy(ref);
    if (total < 100000) { fprintf(stderr, "only %" PRIu64 " records in the reference run\n", total); return 1; }

    pthread_t th[N_THREADS], tk;
    for (long i = 0; i < N_THREADS; i++) pthread_create(&th[i], NULL, caller, (void *)i);
    pthread_create(&tk, NULL, ticker, NULL);
    pthread_join(tk, NULL);
    for (int i = 0; i < N_THREADS; i++) pthread_join(th[i], NULL);
    chd_destroy(ctx);
    if (atomic_load(&failures)) return 1;
    printf("concurrent callers ok: %d ticks (%" PRIu64 " records) beside %llu stateless calls from %d threads\n", TICKS, total,
           (unsigned long long)atomic_load(&n_calls), N_THREADS);
    return 0;
}
