/*
 * tests/c/cgo_sequence.c — the call sequence of INTEGRATION.md §2's cgo shim, in plain C11.
 *
 * No Go toolchain exists in the build image, so the shim itself cannot be compiled here.  What cgo generates for it is C
 * calls with C argument types; this program issues EXACTLY those calls — same functions, same order, same buffer ownership
 * and lifetimes — so that the boundary is exercised as the shim would exercise it, by a C compiler that has never seen the
 * C++ mirror or the ctypes binding:
 *
 *   LoadConfig            chd_create(&cfg, device, &ctx)                       (spatial.go:141-159)
 *   GetChannelId          chd_get_channel_ids(ctx, &x, &z, 1, &id)              one point, pointers to stack variables (:161-180)
 *   QueryChannelIds       chd_query_channel_ids(ctx, &q, 1, ...)               one query, Go slices as ids / dists (:182-317)
 *   GetRegions / GetAdjacentChannels                                           SoA out-arrays (:319-381)
 *   start-up              chd_world_create, chd_host_alloc (the tick buffers: C memory, reused), chd_world_spawn, chd_subs_add
 *   Tick (every tick)     inputs in FRESH heap blocks ("Go slices": borrowed for the call only — they are poisoned and freed the
 *                         moment chd_tick returns), chd_tick with the page-locked output buffers, chd_tick_fetch_segments,
 *                         expansion of the segments while "queueing messages" 
 *
 * Checks (no oracle here — this file knows nothing but include/chd_spatial.h): the expansion yields exactly n_records
 * records, every one addressed to its slot's connection, and their digest {count, sum, xor of mix64(conn << 32 | channel)}
 * equals chd_tick_digest's — the device's own fold of the records where they lie; channel ids of points are stable under
 * the single-point and the batched call; the first fan-out of a connection is FULL.
 *
 * Exit code 0: all checks passed ("cgo sequence ok ...").  3: no HIP device (chd_create -> CHD_E_NO_DEVICE, which is what a
 * gateway on a GPU-less host must see: there is no CPU fallback).  1: a check failed.
 *
 *   gcc -std=c11 -Wall -Wextra -Werror -pedantic -Iinclude tests/c/cgo_sequence.c -Lchanneld_amd -lchd_spatial -o cgo_sequence
 */
#include <chd_spatial.h>

#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N_ENT 1500u
#define N_CONN 64u
#define TICKS 8

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, "\n");                        \
            return 1;                                     \
        }                                                 \
    } while (0)
#define OK(call)                                                                                   \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        CHECK(rc_ == CHD_OK, "%s -> %d (%s)", #call, rc_, chd_last_error(ctx) ? chd_last_error(ctx) : "?"); \
    } while (0)

static uint64_t mix64(uint64_t k) {
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    return k ^ (k >> 31);
}

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static double uniform01(void) { /* SplitMix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}

/* a "Go slice": a fresh heap block per tick, poisoned and freed right after the call it was passed to */
static void *slice(size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p) abort();
    return p;
}
static void drop(void *p, size_t bytes) {
    memset(p, 0xA5, bytes);
    free(p);
}

struct digest {
    uint64_t count, sum, xor_;
};
static void fold(struct digest *d, uint32_t conn, uint32_t channel) {
    const uint64_t h = mix64(((uint64_t)conn << 32) | channel);
    d->count++;
    d->sum += h;
    d->xor_ ^= h;
}

int main(void) {
    chd_ctx *ctx = NULL;

    /* ---- LoadConfig (the 4x4 grid of config/spatial_static_4x4.json, the damping table of message_spatial.go:16-29) ---- */
    chd_grid_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.grid_width = 2000.0; cfg.grid_height = 2000.0;
    cfg.world_offset_x = -4000.0; cfg.world_offset_z = -4000.0;
    cfg.grid_cols = 4; cfg.grid_rows = 4; cfg.server_cols = 2; cfg.server_rows = 2;
    cfg.server_interest_border_size = 1;
    cfg.spatial_channel_id_start = 0x10000; cfg.entity_channel_id_start = 0x80000;
    cfg.default_fanout_interval_ms = 20;
    cfg.n_damping = 3;
    cfg.damping_max_dist[0] = 0; cfg.damping_interval_ms[0] = 20;
    cfg.damping_max_dist[1] = 1; cfg.damping_interval_ms[1] = 50;
    cfg.damping_max_dist[2] = 2; cfg.damping_interval_ms[2] = 100;
    cfg.strict_load_config = 1;
    {
        const int rc = chd_create(&cfg, 0, &ctx);
        if (rc == CHD_E_NO_DEVICE) {
            printf("no HIP device: chd_create -> CHD_E_NO_DEVICE (%s)\n", chd_last_error(NULL) ? chd_last_error(NULL) : "");
            return 3;
        }
        CHECK(rc == CHD_OK && ctx, "chd_create -> %d", rc);
    }
    CHECK(chd_abi_version() == CHD_ABI_VERSION, "header ABI %d, library %d", CHD_ABI_VERSION, chd_abi_version());
    {   /* a config LoadConfig rejects (spatial.go:146-157): the error comes back, no context */
        chd_grid_cfg bad = cfg;
        chd_ctx *none = NULL;
        bad.grid_cols = 0;
        CHECK(chd_create(&bad, 0, &none) == CHD_E_CONFIG && none == NULL, "LoadConfig accepted GridCols = 0");
    }

    /* ---- GetChannelId: one point per call, as the reference's callers do ---- */
    {
        double x = -3999.0, z = -3999.0;
        uint32_t id = 7;
        OK(chd_get_channel_ids(ctx, &x, &z, 1, &id));
        CHECK(id == 0x10000, "corner cell: %#x", id);
        x = 1.0; z = 2001.0;
        OK(chd_get_channel_ids(ctx, &x, &z, 1, &id));
        CHECK(id == 0x10000 + 3 * 4 + 2, "cell (2,3): %#x", id);
        x = 4000.0; /* the exclusive upper edge: (0, err) */
        OK(chd_get_channel_ids(ctx, &x, &z, 1, &id));
        CHECK(id == 0, "out of the world: %#x", id);
    }

    /* ---- QueryChannelIds: one query per call; ids / dists are the caller's slices ---- */
    {
        chd_aoi_query q;
        uint32_t off[2] = {9, 9}, *ids = slice(16 * 4), *dists = slice(16 * 4), ivs[16];
        int32_t status = 99;
        memset(&q, 0, sizeof q);
        q.shapes = CHD_SHAPE_SPHERE;
        q.sph_cx = 100.0; q.sph_cz = 100.0; q.sph_r = 2500.0;
        OK(chd_query_channel_ids(ctx, &q, 1, NULL, NULL, NULL, 0, off, ids, dists, ivs, 16, &status));
        CHECK(status == CHD_OK && off[0] == 0 && off[1] >= 2 && off[1] <= 16, "sphere query: status %d, %u ids", status, off[1]);
        for (uint32_t i = 0; i < off[1]; i++) {
            CHECK(ids[i] >= 0x10000 && ids[i] < 0x10010 && (i == 0 || ids[i] > ids[i - 1]), "ids sorted, in the grid");
            CHECK(ivs[i] == (dists[i] == 0 ? 20u : dists[i] == 1 ? 50u : 100u), "damped interval of dist %u: %u", dists[i], ivs[i]);
        }
        q.sph_r = -1.0; /* spatial.go: invalid radius -> the query's error, the call itself succeeds */
        OK(chd_query_channel_ids(ctx, &q, 1, NULL, NULL, NULL, 0, off, ids, dists, NULL, 16, &status));
        CHECK(status == CHD_E_EXTENT && off[1] == 0, "negative radius: status %d", status);
        drop(ids, 16 * 4);
        drop(dists, 16 * 4);
    }
    {   /* GetRegions / GetAdjacentChannels */
        double mnx[16], mnz[16], mxx[16], mxz[16];
        uint32_t chan[16], srv[16], adj[8], cnt = 0, c = 0x10005;
        OK(chd_get_regions(ctx, mnx, mnz, mxx, mxz, chan, srv));
        CHECK(chan[5] == 0x10005 && mnx[5] == -2000.0 && mxx[5] == 0.0 && srv[5] == 0 && srv[15] == 3, "regions");
        OK(chd_get_adjacent_channels(ctx, &c, 1, adj, &cnt));
        CHECK(cnt == 8 && adj[0] == 0x10000 && adj[7] == 0x1000A, "adjacent of an inner cell: %u", cnt);
    }

    /* ---- start-up: world, page-locked tick buffers (C memory, allocated once), population ---- */
    chd_world_cfg wc;
    memset(&wc, 0, sizeof wc);
    wc.max_entities = N_ENT; wc.max_subscribers = N_CONN; wc.max_records = 1u << 22;
    wc.flags = CHD_WORLD_CONN_MAJOR_EMIT | CHD_WORLD_ONE_WAVE_EMIT; /* the descriptor path (what a world of >= 4096 connections takes by itself) */
    OK(chd_world_create(ctx, &wc));
    const uint32_t capq = 16, cap_lists = N_CONN * capq;
    chd_handover_rec *ho = NULL;
    uint32_t *unsub_s = NULL, *unsub_c = NULL, *new_s = NULL, *new_c = NULL, *new_iv = NULL;
    chd_segments_out seg;
    memset(&seg, 0, sizeof seg);
    OK(chd_host_alloc(ctx, sizeof(chd_handover_rec) * N_ENT, (void **)&ho));
    OK(chd_host_alloc(ctx, 4 * cap_lists, (void **)&unsub_s));
    OK(chd_host_alloc(ctx, 4 * cap_lists, (void **)&unsub_c));
    OK(chd_host_alloc(ctx, 4 * cap_lists, (void **)&new_s));
    OK(chd_host_alloc(ctx, 4 * cap_lists, (void **)&new_c));
    OK(chd_host_alloc(ctx, 4 * cap_lists, (void **)&new_iv));
    seg.segments_cap = (uint64_t)N_CONN * capq * 4; seg.columns_cap = 10 * (N_ENT + 1024); seg.records_cap = 1u << 20;
    OK(chd_host_alloc(ctx, sizeof(chd_fanout_segment) * seg.segments_cap, (void **)&seg.segments));
    OK(chd_host_alloc(ctx, 4 * (N_CONN + 1), (void **)&seg.conn_seg_off));
    OK(chd_host_alloc(ctx, 4 * seg.columns_cap, (void **)&seg.columns));
    OK(chd_host_alloc(ctx, sizeof(chd_fanout_rec) * seg.records_cap, (void **)&seg.records));
    OK(chd_host_alloc(ctx, 8 * (N_CONN + 1), (void **)&seg.conn_rec_off));

    double *px = slice(8 * N_ENT), *pz = slice(8 * N_ENT);   /* the gateway's own copy of the positions */
    uint32_t conn_of_slot[N_CONN];
    {
        uint32_t *chan = slice(4 * N_ENT), *snd = slice(4 * N_ENT), *conn = slice(4 * N_CONN);
        for (uint32_t i = 0; i < N_ENT; i++) {
            px[i] = (double)(float)(-4000.0 + 7999.0 * uniform01());
            pz[i] = (double)(float)(-4000.0 + 7999.0 * uniform01());
            chan[i] = 0x80000 + i;
            snd[i] = 1 + (i & 3);
        }
        for (uint32_t s = 0; s < N_CONN; s++) conn[s] = conn_of_slot[s] = 1000 + s;
        OK(chd_world_spawn(ctx, N_ENT, NULL, chan, px, pz, NULL, snd));
        OK(chd_subs_add(ctx, N_CONN, NULL, conn));
        drop(chan, 4 * N_ENT); drop(snd, 4 * N_ENT); drop(conn, 4 * N_CONN);
    }

    /* ---- the tick driver ---- */
    uint64_t total = 0, full_first = 0, n_handovers = 0;
    uint8_t seen_first[N_CONN];
    memset(seen_first, 0, sizeof seen_first);
    for (int t = 1; t <= TICKS; t++) {
        /* 1. drain what the channel goroutines queued: fresh slices, borrowed for the call */
        uint32_t *idx = slice(4 * N_ENT), *qsub = slice(4 * N_CONN);
        double *x = slice(8 * N_ENT), *z = slice(8 * N_ENT);
        chd_aoi_query *q = slice(sizeof(chd_aoi_query) * N_CONN);
        uint32_t n_upd = 0;
        for (uint32_t i = 0; i < N_ENT; i++) {
            if (uniform01() < 0.1) continue; /* not every entity sends an update every tick */
            double nx = px[i] + 300.0 * (uniform01() - 0.5), nz = pz[i] + 300.0 * (uniform01() - 0.5);
            if (nx < -4000.0 || nx >= 4000.0) nx = px[i];
            if (nz < -4000.0 || nz >= 4000.0) nz = pz[i];
            px[i] = (double)(float)nx; pz[i] = (double)(float)nz;
            idx[n_upd] = i; x[n_upd] = px[i]; z[n_upd] = pz[i];
            n_upd++;
        }
        memset(q, 0, sizeof(chd_aoi_query) * N_CONN);
        for (uint32_t s = 0; s < N_CONN; s++) { /* every connection follows an entity */
            qsub[s] = s;
            q[s].shapes = CHD_SHAPE_SPHERE;
            q[s].sph_cx = px[s * 7]; q[s].sph_cz = pz[s * 7]; q[s].sph_r = 2200.0;
        }
        chd_tick_in in;
        memset(&in, 0, sizeof in);
        in.now_ns = (int64_t)t * 50000000;
        in.n_updates = n_upd; in.upd_idx = idx; in.upd_x = x; in.upd_z = z; in.upd_sender = NULL;
        in.n_queries = N_CONN; in.query_sub = qsub; in.queries = q;
        chd_tick_out out;
        memset(&out, 0, sizeof out);
        out.handovers = ho; out.handovers_cap = N_ENT;
        out.unsub_sub = unsub_s; out.unsub_channel = unsub_c; out.unsub_cap = cap_lists;
        out.newsub_sub = new_s; out.newsub_channel = new_c; out.newsub_interval_ms = new_iv; out.newsub_cap = cap_lists;
        OK(chd_tick(ctx, &in, &out)); /* (no records: they stay in HBM) */
        drop(idx, 4 * N_ENT); drop(qsub, 4 * N_CONN); drop(x, 8 * N_ENT); drop(z, 8 * N_ENT); drop(q, sizeof(chd_aoi_query) * N_CONN);
        CHECK(out.overflow == 0 && out.history_overflow == 0, "tick %d: overflow %#x", t, out.overflow);

        /* 2. handovers: the records the reference's message assembly runs for */
        for (uint32_t h = 0; h < out.n_handovers; h++) {
            CHECK(ho[h].entity < N_ENT && ho[h].channel == 0x80000 + ho[h].entity && ho[h].src != ho[h].dst, "handover record %u", h);
            CHECK(ho[h].src >= 0x10000 && ho[h].src < 0x10010 && ho[h].dst >= 0x10000 && ho[h].dst < 0x10010, "handover cells");
            CHECK((ho[h].src_server != ho[h].dst_server) == (((ho[h].src - 0x10000) % 4 / 2 != (ho[h].dst - 0x10000) % 4 / 2) ||
                                                               ((ho[h].src - 0x10000) / 8 != (ho[h].dst - 0x10000) / 8)), "cross-server flag");
        }
        n_handovers += out.n_handovers;
        /* 3. interest diff */
        for (uint32_t i = 0; i < out.n_newsubs; i++)
            CHECK(new_s[i] < N_CONN && new_c[i] >= 0x10000 && new_c[i] < 0x10010 && (new_iv[i] == 20 || new_iv[i] == 50 || new_iv[i] == 100), "new sub %u", i);
        for (uint32_t i = 0; i < out.n_unsubs; i++) CHECK(unsub_s[i] < N_CONN && unsub_c[i] >= 0x10000 && unsub_c[i] < 0x10010, "unsub %u", i);

        /* 4. fan-out: segments, expanded while "queueing messages" */
        OK(chd_tick_fetch_segments(ctx, &seg));
        struct digest mine = {0, 0, 0};
        for (uint32_t s = 0; s < N_CONN; s++) {
            const uint32_t conn = conn_of_slot[s];
            for (uint32_t k = seg.conn_seg_off[s]; k < seg.conn_seg_off[s + 1]; k++) {
                const chd_fanout_segment g = seg.segments[k];
                const uint64_t before = mine.count;
                if (g.n_info & CHD_SEG_EXPLICIT) {
                    const chd_fanout_rec *r = seg.records + seg.conn_rec_off[s] + g.off;
                    for (uint32_t i = 0; i < g.n_records; i++) {
                        CHECK((r[i].conn & ~CHD_REC_FULL) == conn, "explicit record of slot %u addressed to %u", s, r[i].conn);
                        fold(&mine, r[i].conn, r[i].channel);
                    }
                } else {
                    const uint32_t n = CHD_SEG_N(g.n_info), *col = seg.columns + g.off;
                    CHECK((uint64_t)g.off + n <= seg.n_columns, "column range");
                    if (g.n_info & CHD_SEG_FIRST) {
                        fold(&mine, conn | CHD_REC_FULL, g.channel);
                        for (uint32_t i = 0; i < n; i++) fold(&mine, conn | CHD_REC_FULL, col[i]);
                        if (!seen_first[s]) { seen_first[s] = 1; full_first++; }
                    }
                    for (uint32_t j = 0; j < CHD_SEG_NWIN(g.n_info); j++) {
                        if (g.n_info & CHD_SEG_OWN(j)) fold(&mine, conn, g.channel);
                        if (!(g.n_info & CHD_SEG_NONE))
                            for (uint32_t i = 0; i < n; i++) fold(&mine, conn, col[i]);
                    }
                }
                CHECK(mine.count - before == g.n_records, "segment %u of slot %u expands to %" PRIu64 " records, says %u", k, s, mine.count - before, g.n_records);
            }
        }
        chd_records_digest dev;
        OK(chd_tick_digest(ctx, &dev, NULL));
        CHECK(mine.count == seg.n_records && mine.count == dev.count, "tick %d: %" PRIu64 " expanded, %" PRIu64 " fetched, %" PRIu64 " on the device", t,
              mine.count, (uint64_t)seg.n_records, (uint64_t)dev.count);
        CHECK(mine.sum == dev.sum && mine.xor_ == dev.xor_, "tick %d: the expanded records are not the device's records", t);
        total += mine.count;
    }
    CHECK(total > 100000 && full_first == N_CONN && n_handovers > 0, "totals: %" PRIu64 " records, %" PRIu64 " first fan-outs, %" PRIu64 " handovers", total,
          full_first, n_handovers);

    /* ---- ABI v9: which schedule the ticks took, the tick history's own overflow words, the server-connection table, the local
     * "can this process take part in the library's collectives" question ---- */
    {
        chd_tick_stats st, hist[4];
        OK(chd_get_tick_stats(ctx, &st));
        CHECK(st.gate_timeouts == 0 && (st.schedule & ~(CHD_SCHED_OVERLAP_INTEREST | CHD_SCHED_GATED | CHD_SCHED_PIPELINED)) == 0, "schedule %#x, gate time-outs %u",
              st.schedule, st.gate_timeouts);
        OK(chd_get_tick_history(ctx, 4, hist));
        for (int k = 0; k < 4; k++) CHECK(hist[k].overflow == 0 && hist[k].history_overflow == 0 && hist[k].n_records > 0, "tick history %d: overflow %#x", k, hist[k].overflow);
        const uint32_t servers[4] = {901, 902, 903, 904}; /* the 4x4 grid of this program has 2x2 server regions */
        OK(chd_world_set_server_connections(ctx, 4, servers));
        CHECK(chd_world_set_server_connections(ctx, 3, servers) == CHD_E_INVAL, "a server table of the wrong size was accepted");
        OK(chd_world_set_server_connections(ctx, 0, NULL));
        const int avail = chd_shard_comm_available(); /* CHD_OK where librccl.so loads, CHD_E_STATE (with the loader's message) where not: never a hang */
        CHECK(avail == CHD_OK || avail == CHD_E_STATE, "chd_shard_comm_available -> %d", avail);
    }

    drop(px, 8 * N_ENT); drop(pz, 8 * N_ENT);
    OK(chd_host_free(ctx, ho));
    OK(chd_host_free(ctx, seg.segments));
    chd_destroy(ctx);
    printf("cgo sequence ok: %d ticks, %" PRIu64 " records expanded from segments == the device's digests, %" PRIu64 " handovers\n", TICKS, total, n_handovers);
    return 0;
}
