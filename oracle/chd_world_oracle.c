/*
 * chd_world_oracle.c — CPU oracle for the *tick pipeline* (TEST INFRASTRUCTURE
 * ONLY, see chd_oracle.h).  It composes the literal restatements of
 * chd_oracle.c into the engine's tick model:
 *
 *   1. ingest entity updates: Notify decision (spatial.go:612-626) on
 *      (last merged position, new position); accepted handovers move the
 *      entity between the cells' entity maps (spatial.go:703-736); a locked
 *      entity aborts (entity.go:197-224 -> spatial.go:675-679); every update is
 *      appended to the entity channel's update buffer (data.go:149-173).
 *   2. interest updates: QueryChannelIds + damping + Difference
 *      (message_spatial.go:59-128) and SubscribeToChannel /
 *      UnsubscribeFromChannel (subscription.go:34-125) on the spatial channels.
 *   3. fan-out at time t: tickData (data.go:175-291) on every spatial (cell)
 *      channel and on every entity channel.  Model decision:
 *      an entity channel's subscribers are the subscribers of the cell that
 *      holds it, sharing the cell subscription's phase (lastFanOutTime,
 *      hadFirstFanOut) — exact because that state's evolution in tickData
 *      does not depend on the update buffer's contents.
 *
 * Two modes, cross-checked in tests/test_world_oracle.py:
 *   literal = 1: every entity channel and cell channel is an orc_channel and
 *                orc_tick_data (the literal list walk) produces the sends.
 *   literal = 0: the window formulation (while t >= last+interval ...) with the
 *                literal buffer walk per (channel, subscriber, window); this is
 *                also the loop nest timed as bench.py's cpu_baseline ("port").
 */
#include "chd_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define W_INVALID 0xFFFFFFFFu
#define REC_FULL 0x80000000u

typedef struct {
    orc_time arrival;
    uint32_t sender;
} wupd;

typedef struct {
    wupd *v;
    uint32_t head, len, cap;
    uint32_t pushed; /* updates the channel has taken so far: element i of the buffer is update number pushed - len + i */
} wbuf;

typedef struct {
    uint32_t conn; /* bit31 = full */
    uint32_t chan;
    uint32_t mask; /* window mode: which buffered updates the message merges, bit j = the one that arrived with
                      tick (current - j); 0 for the full state */
    uint32_t range; /* ... and as a range of the channel's update numbers (chd_tick_out.record_masks, range form): bit 31 |
                       (count - 1) << 21 | first & 0x1FFFFF over the buffered elements whose arrival lies in the window */
} wrec;

typedef struct {
    uint32_t cell; /* cell index */
    uint32_t interval_ms;
    orc_time last;
    uint8_t had_first, skip_self, access, is_new;
} wpair;

typedef struct orc_world {
    orc_grid g;
    uint32_t C, N, S, capq;
    uint32_t default_interval_ms;
    int32_t default_delay_ms;
    int literal;
    /* entities */
    uint8_t *alive;
    uint32_t *chan_id, *cell, *member, *eflags, *sender;
    uint32_t *group; /* handover group id per entity (0 = never added to a group: GetHandoverEntities returns the entity itself) */
    /* explicit GetHandoverEntities results (entity.go:197-224 as evaluated by a FlatEntityGroupController, oracle/groups.py):
     * hl_has[i] != 0 -> entity i's handover list is hl_mem[i][0..hl_n[i]) (empty: locked or emptied group, no handover) */
    uint8_t *hl_has;
    uint32_t *hl_n;
    uint32_t **hl_mem;
    wbuf *ebuf;     /* entity channel update buffers */
    wbuf *cbuf;     /* cell channel update buffers */
    /* ChannelData.maxFanOutIntervalMs, one per CHANNEL, only grows (subscription.go:83-86: raised when a subscription is
     * CREATED — the branch that merges options into an existing one does not touch it): cmax[c] for spatial channel c; emax[i]
     * for entity channel i, whose subscribers are (tick model) those of the cells that hold it — raised at every update of the
     * entity, before the eviction test (OnUpdate merges first: Notify -> the dst cell's connections are subscribed to the entity
     * channel, spatial.go:797-830; then data.go:165-171), to the maxima of the cell of its last merged position and of the new one */
    uint32_t *cmax, *emax;
    /* subscribers */
    uint8_t *sub_alive;
    uint32_t *conn_id;
    wpair *pairs; /* [S][capq] */
    uint32_t *pair_cnt;
    /* outputs of the last tick */
    wrec *rec; uint64_t nrec, caprec;
    uint32_t *ho_ent, *ho_src, *ho_dst, *ho_srv_src, *ho_srv_dst; uint32_t nho, capho;
    uint32_t *unsub_sub, *unsub_cell; uint32_t nunsub, capunsub;
    /* handover message recipients of the last tick (spatial.go:776-857) */
    uint32_t *rcp_ho, *rcp_conn; uint8_t *rcp_kind; uint64_t nrcp, caprcp;
    /* per recipient: which entities of its handover go out WITH their entityData — `shouldSend` of SubscribeToChannel(entityCh)
     * per (dst connection, entity), spatial.go:797-857; per handover (<= 32 entities each): the cell whose entity map held every
     * handover entity when Notify ran (ho_ent_before[32 h + q], W_INVALID = none) and how many entities the handover has */
    uint32_t *rcp_mask;
    uint32_t *ho_ent_before, *ho_ent_n; uint32_t cap_ho_ent;
    uint8_t *ho_own_unsub; uint32_t cap_ho_own; /* per handover: step 1 unsubscribes the src spatial server (spatial.go:688-694) */
    int32_t *q_status; uint32_t nq_status;
    uint32_t n_locked_abort;
    uint64_t literal_mismatch;
    uint32_t *server_of_cell;
    /* orc_world_set_server_conns: ConnectionId of spatial server k (owner of its cells' channels, and — tick model — of the entity
     * channels those cells hold); NULL: no connection is an owner */
    uint32_t *server_conn; uint32_t n_server_conn;
    int threads;
    struct fan_job_s *jobs; int njobs; /* per-thread record buffers, kept across ticks */
    orc_time stamps[32]; uint32_t nstamps; /* channel times of the last 32 ticks, newest first */
    /* digest mode (window formulation only): records are not stored, only folded into an order-independent
     * digest (full-size parity runs: 10^8..10^9 records per tick) */
    /* spatialDampingSettings (message_spatial.go:16-29) replaced by the host (the table is a package variable there) */
    uint32_t n_damp, damp_dist[8], damp_iv[8];
    int digest_only;
    /* sorted_walk: walk a buffer newest-first and stop at the first element older than the window (see window_has_update_sorted);
     * unsorted = some channel's arrival stamps went backwards: the sorted walk is then not taken */
    int sorted_walk, unsorted;
    uint64_t d_cnt, d_sum, d_xor, d_sum_masked;
    uint64_t *d_conn; /* [S] per subscriber slot: sum of the record hashes */
} orc_world;

/* SplitMix64 finaliser: the record hash of the digests (same function in the device kernel and in tests/) */
static inline uint64_t mix64(uint64_t k) {
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    return k ^ (k >> 31);
}

static void wbuf_push(wbuf *b, orc_time t, uint32_t sender, uint32_t max_interval_ms) {
    /* data.go:159-172 */
    if (b->head + b->len == b->cap) {
        if (b->head > 0) {
            memmove(b->v, b->v + b->head, b->len * sizeof(wupd));
            b->head = 0;
        } else {
            b->cap = b->cap ? b->cap * 2 : 8;
            b->v = (wupd *)realloc(b->v, b->cap * sizeof(wupd));
        }
    }
    b->v[b->head + b->len].arrival = t;
    b->v[b->head + b->len].sender = sender;
    b->len++;
    b->pushed++;
    if (b->len > 512) {
        if (b->v[b->head].arrival + (orc_time)max_interval_ms * 1000000 < t) {
            b->head++;
            b->len--;
        }
    }
}

orc_world *orc_world_new(const orc_grid *g, uint32_t n_entities, uint32_t n_subs,
                         uint32_t capq, uint32_t default_interval_ms,
                         int32_t default_delay_ms, int literal) {
    orc_world *w = (orc_world *)calloc(1, sizeof(orc_world));
    w->g = *g;
    w->C = g->cols * g->rows;
    w->N = n_entities;
    w->S = n_subs;
    w->capq = capq;
    w->default_interval_ms = default_interval_ms;
    w->default_delay_ms = default_delay_ms;
    w->literal = literal;
    w->threads = 1;
    w->alive = (uint8_t *)calloc(n_entities + 1, 1);
    w->chan_id = (uint32_t *)calloc(n_entities + 1, 4);
    w->cell = (uint32_t *)calloc(n_entities + 1, 4);
    w->member = (uint32_t *)calloc(n_entities + 1, 4);
    w->eflags = (uint32_t *)calloc(n_entities + 1, 4);
    w->sender = (uint32_t *)calloc(n_entities + 1, 4);
    w->group = (uint32_t *)calloc(n_entities + 1, 4);
    w->hl_has = (uint8_t *)calloc(n_entities + 1, 1);
    w->hl_n = (uint32_t *)calloc(n_entities + 1, 4);
    w->hl_mem = (uint32_t **)calloc(n_entities + 1, sizeof(uint32_t *));
    w->ebuf = (wbuf *)calloc(n_entities + 1, sizeof(wbuf));
    w->cbuf = (wbuf *)calloc(w->C + 1, sizeof(wbuf));
    w->cmax = (uint32_t *)calloc(w->C + 1, 4);
    w->emax = (uint32_t *)calloc(n_entities + 1, 4);
    w->sub_alive = (uint8_t *)calloc(n_subs + 1, 1);
    w->conn_id = (uint32_t *)calloc(n_subs + 1, 4);
    w->pairs = (wpair *)calloc((size_t)n_subs * capq + 1, sizeof(wpair));
    w->pair_cnt = (uint32_t *)calloc(n_subs + 1, 4);
    w->server_of_cell = (uint32_t *)calloc(w->C + 1, 4);
    {
        double *t = (double *)malloc(sizeof(double) * 4 * (w->C + 1));
        uint32_t *ids = (uint32_t *)malloc(4 * (w->C + 1));
        orc_regions(g, t, t + w->C, t + 2 * w->C, t + 3 * w->C, ids, w->server_of_cell);
        free(t);
        free(ids);
    }
    return w;
}

static void orc__free_jobs(orc_world *w);

void orc_world_free(orc_world *w) {
    if (w) { free(w->rcp_ho); free(w->rcp_conn); free(w->rcp_kind); free(w->rcp_mask); free(w->ho_ent_before); free(w->ho_ent_n); free(w->ho_own_unsub); }
    if (!w) return;
    for (uint32_t i = 0; i < w->N; i++) free(w->ebuf[i].v);
    for (uint32_t i = 0; i < w->C; i++) free(w->cbuf[i].v);
    free(w->alive); free(w->chan_id); free(w->cell); free(w->member);
    for (uint32_t i = 0; i < w->N; i++) free(w->hl_mem[i]);
    free(w->hl_has); free(w->hl_n); free(w->hl_mem);
    free(w->eflags); free(w->sender); free(w->group); free(w->ebuf); free(w->cbuf); free(w->cmax); free(w->emax);
    free(w->sub_alive); free(w->conn_id); free(w->pairs); free(w->pair_cnt);
    free(w->rec); free(w->ho_ent); free(w->ho_src); free(w->ho_dst);
    free(w->ho_srv_src); free(w->ho_srv_dst); free(w->unsub_sub);
    free(w->unsub_cell); free(w->q_status); free(w->server_of_cell); free(w->server_conn); free(w->d_conn);
    orc__free_jobs(w);
    free(w);
}

void orc_world_set_damping(orc_world *w, uint32_t n, const uint32_t *max_dist, const uint32_t *interval_ms) {
    w->n_damp = n > 8 ? 8 : n;
    for (uint32_t i = 0; i < w->n_damp; i++) { w->damp_dist[i] = max_dist[i]; w->damp_iv[i] = interval_ms[i]; }
}
void orc_world_set_server_conns(orc_world *w, uint32_t n, const uint32_t *conn) {
    free(w->server_conn);
    w->server_conn = NULL;
    w->n_server_conn = n;
    if (n) {
        w->server_conn = (uint32_t *)malloc(4 * (size_t)n);
        memcpy(w->server_conn, conn, 4 * (size_t)n);
    }
}
void orc_world_set_threads(orc_world *w, int threads) { w->threads = threads < 1 ? 1 : threads; }

static uint32_t cell_index(const orc_world *w, double x, double z) {
    uint32_t id = orc_channel_id(&w->g, x, z);
    return id ? id - w->g.id_start : W_INVALID;
}

/* spawn: the entity's channel gets its initial data and the entity is added to
 * the cell that contains it (pkg/unreal/message.go:55 -> SpatialChannelData). */
void orc_world_spawn(orc_world *w, uint32_t i, uint32_t chan_id, double x,
                     double z, uint32_t flags, uint32_t sender) {
    w->alive[i] = 1;
    w->chan_id[i] = chan_id;
    w->cell[i] = cell_index(w, x, z);
    w->member[i] = w->cell[i];
    w->eflags[i] = flags;
    w->sender[i] = sender;
    w->ebuf[i].head = w->ebuf[i].len = 0;
    w->emax[i] = 0; /* a new channel */
}

void orc_world_despawn(orc_world *w, uint32_t i) { w->alive[i] = 0; w->member[i] = W_INVALID; }
void orc_world_set_flags(orc_world *w, uint32_t i, uint32_t flags) { w->eflags[i] = flags; }
void orc_world_set_group(orc_world *w, uint32_t i, uint32_t group) { w->group[i] = group; }
/* has = 0: back to "AddToGroup was never called" (the entity itself) */
void orc_world_set_handover_list(orc_world *w, uint32_t i, int has, uint32_t n, const uint32_t *members) {
    free(w->hl_mem[i]);
    w->hl_mem[i] = NULL;
    w->hl_has[i] = has ? 1 : 0;
    w->hl_n[i] = has ? n : 0;
    if (has && n) {
        w->hl_mem[i] = (uint32_t *)malloc(4 * (size_t)n);
        memcpy(w->hl_mem[i], members, 4 * (size_t)n);
    }
}

void orc_world_add_sub(orc_world *w, uint32_t s, uint32_t conn_id) {
    w->sub_alive[s] = 1;
    w->conn_id[s] = conn_id;
    w->pair_cnt[s] = 0;
}
void orc_world_remove_sub(orc_world *w, uint32_t s) { w->sub_alive[s] = 0; w->pair_cnt[s] = 0; }

/* SubscribeToChannel(conn, spatial channel, options) for an explicit SUB_TO_CHANNEL (subscription.go:34-102).  `set`
 * bit 0 DataAccess, 1 FanOutIntervalMs, 2 FanOutDelayMs, 3 SkipSelfUpdateFanOut, 4 SkipFirstFanOut = the fields present.
 * Returns the second result of SubscribeToChannel (should the subscription message be sent), or -1 when the
 * connection does not exist / is closing ((nil, false)), -5 when the list is full. */
int orc_world_set_sub_options(orc_world *w, uint32_t s, uint32_t channel, uint32_t set, uint32_t access, uint32_t interval_ms,
                              int32_t delay_ms, uint32_t skip_self, uint32_t skip_first, orc_time now) {
    if (s >= w->S || !w->sub_alive[s]) return -1;
    const uint32_t c = channel - w->g.id_start;
    wpair *pp = &w->pairs[(size_t)s * w->capq];
    uint32_t n = w->pair_cnt[s], pos = 0;
    while (pos < n && pp[pos].cell < c) pos++;
    if (pos < n && pp[pos].cell == c) { /* :44-57: proto.Merge(&cs.options, options); dataAccessChanged */
        wpair *p = &pp[pos];
        const uint8_t old_access = p->access;
        if (set & 1u) p->access = (uint8_t)access;
        if (set & 2u) p->interval_ms = interval_ms;
        if (set & 8u) p->skip_self = (uint8_t)(skip_self != 0);
        /* FanOutDelayMs / SkipFirstFanOut only live in the stored options: the queue element is untouched */
        /* (maxFanOutIntervalMs is NOT raised here: subscription.go:83-86 sits on the new-subscription branch only) */
        return old_access != p->access;
    }
    if (n >= w->capq) return -5;
    memmove(&pp[pos + 1], &pp[pos], sizeof(wpair) * (n - pos));
    wpair *p = &pp[pos];
    p->cell = c;                                                         /* :59-64 defaults (:21-31) merged with the options */
    p->access = (set & 1u) ? (uint8_t)access : ORC_ACCESS_READ;
    p->interval_ms = (set & 2u) ? interval_ms : w->default_interval_ms;
    p->skip_self = (set & 8u) ? (uint8_t)(skip_self != 0) : 1;
    p->had_first = (set & 16u) ? (uint8_t)(skip_first != 0) : 0;          /* hadFirstFanOut: *cs.options.SkipFirstFanOut (:68) */
    p->last = now + (orc_time)((set & 4u) ? delay_ms : w->default_delay_ms) * 1000000; /* :70 */
    p->is_new = 0;
    if (w->cmax[c] < p->interval_ms) w->cmax[c] = p->interval_ms; /* :83-86 */
    w->pair_cnt[s] = n + 1;
    return 1;
}

/* DataAccess and SkipSelfUpdateFanOut of a subscriber's pairs, list order */
uint32_t orc_world_pair_options(const orc_world *w, uint32_t s, uint8_t *access, uint8_t *skip_self) {
    uint32_t n = w->pair_cnt[s];
    for (uint32_t p = 0; p < n; p++) {
        access[p] = w->pairs[(size_t)s * w->capq + p].access;
        skip_self[p] = w->pairs[(size_t)s * w->capq + p].skip_self;
    }
    return n;
}

static void push_rec(orc_world *w, uint32_t conn, uint32_t chan) {
    if (w->nrec == w->caprec) {
        w->caprec = w->caprec ? w->caprec * 2 : 1024;
        w->rec = (wrec *)realloc(w->rec, w->caprec * sizeof(wrec));
    }
    w->rec[w->nrec].conn = conn;
    w->rec[w->nrec].chan = chan;
    w->nrec++;
}

/* the elements of the buffer whose arrival lies in the window [max(last, 0), next], whoever sent them: as a range word */
static uint32_t window_range(const wbuf *b, orc_time last, orc_time next) {
    const orc_time lo = last >= 0 ? last : 0;
    uint32_t first = 0, cnt = 0;
    for (uint32_t i = 0; i < b->len; i++) {
        const orc_time a = b->v[b->head + i].arrival;
        if (a >= lo && a <= next) {
            if (!cnt) first = b->pushed - b->len + i;
            cnt++;
        }
    }
    if (!cnt) return 0;
    return 0x80000000u | (((cnt > 1024u ? 1024u : cnt) - 1u) << 21) | (first & 0x1FFFFFu);
}

/* literal buffer walk of data.go:225-269 for one (subscriber, window) */
static int window_has_update(const orc_world *w, const wbuf *b, orc_time last, orc_time next,
                             uint32_t conn, int skip_self, uint32_t *mask) {
    *mask = 0;
    if (b->len == 0) return 0; /* bufp == nil */
    orc_time last_update_time = 0;
    if (last >= last_update_time) last_update_time = last;
    int merged = 0;
    for (uint32_t i = 0; i < b->len; i++) {
        const wupd *be = &b->v[b->head + i];
        if (be->sender == conn && skip_self) continue;
        if (be->arrival >= last_update_time && be->arrival <= next) {
            merged = 1; /* data.go:249-256: this element is merged into the accumulated update */
            last_update_time = be->arrival;
            for (uint32_t j = 0; j < w->nstamps; j++)
                if (w->stamps[j] == be->arrival) *mask |= 1u << j;
        }
    }
    return merged;
}

/* The same selection for buffers whose arrival stamps do not decrease (queue order: Channel.PutMessage stamps in enqueue order,
 * channel.go:296-310): `arrival >= lastUpdateTime` then holds for every element behind the first one that passes, so the elements
 * merged are exactly those with max(last, 0) <= arrival <= next from a sender that is not skipped — whichever way the buffer is
 * walked.  Walked newest first the loop can stop at the first element older than the window: O(updates inside the reach) instead
 * of O(buffer), which is what lets tests/golden/make_bench_digests.py run hundreds of full-size ticks with 512-deep buffers.
 * Opt-in (orc_world_set_sorted_walk); tests/test_world_oracle.py holds it against the forward walk. */
static int window_has_update_sorted(const orc_world *w, const wbuf *b, orc_time last, orc_time next,
                                    uint32_t conn, int skip_self, uint32_t *mask) {
    *mask = 0;
    const orc_time lo = last >= 0 ? last : 0;
    int merged = 0;
    for (uint32_t i = b->len; i-- > 0;) {
        const wupd *be = &b->v[b->head + i];
        if (be->arrival < lo) break;
        if (be->arrival > next) continue;
        if (be->sender == conn && skip_self) continue;
        merged = 1;
        for (uint32_t j = 0; j < w->nstamps; j++)
            if (w->stamps[j] == be->arrival) *mask |= 1u << j;
    }
    return merged;
}

typedef struct {
    uint8_t full;
    orc_time last, next;
} wwin;

/* windows of one pair at time t (net effect of tickData's revisit loop) */
static uint32_t pair_windows(wpair *p, orc_time t, wwin **pout, uint32_t *cap) {
    uint32_t n = 0;
    if (p->access == ORC_ACCESS_NO) return 0;
    for (;;) {
        orc_time next = p->last + (orc_time)p->interval_ms * 1000000;
        if (!(t >= next)) break;
        if (n >= *cap) { /* the reference's loop has no bound: neither has this */
            *cap *= 2;
            *pout = (wwin *)realloc(*pout, sizeof(wwin) * *cap);
        }
        wwin *out = *pout;
        if (!p->had_first) {
            out[n].full = 1; out[n].last = p->last; out[n].next = next; n++;
            p->had_first = 1;
            p->last = t;
        } else {
            out[n].full = 0; out[n].last = p->last; out[n].next = next; n++;
            p->last = next;
        }
        if (p->interval_ms == 0) break; /* the reference would spin */
    }
    return n;
}

/* ---- per-cell subscriber buckets ---- */
typedef struct { uint32_t s, p; } spref;

typedef struct fan_job_s {
    orc_world *w;
    orc_time t;
    const uint32_t *cell_off_sub; const spref *cell_subs;
    const uint32_t *cell_off_ent; const uint32_t *cell_ents;
    uint32_t c0, c1;
    wrec *rec; uint64_t nrec, caprec;
    uint64_t d_sum, d_xor, d_sum_masked; uint64_t *d_conn; /* digest mode */
} fan_job;

static void orc__free_jobs(orc_world *w) {
    for (int k = 0; k < w->njobs; k++) { free(w->jobs[k].rec); free(w->jobs[k].d_conn); }
    free(w->jobs);
    w->jobs = NULL;
    w->njobs = 0;
}

static void job_push(fan_job *j, uint32_t slot, uint32_t conn, uint32_t chan, uint32_t mask, uint32_t range) {
    if (j->w->digest_only) {
        const uint64_t h = mix64(((uint64_t)conn << 32) | chan);
        j->d_sum += h;
        j->d_xor ^= h;
        j->d_sum_masked += mix64(h + mask);
        j->d_conn[slot] += h;
        j->nrec++;
        return;
    }
    if (j->nrec == j->caprec) {
        j->caprec = j->caprec ? j->caprec * 2 : 4096;
        j->rec = (wrec *)realloc(j->rec, j->caprec * sizeof(wrec));
    }
    j->rec[j->nrec].conn = conn;
    j->rec[j->nrec].chan = chan;
    j->rec[j->nrec].mask = mask;
    j->rec[j->nrec].range = range;
    j->nrec++;
}

#define MAXWIN 4096 /* initial capacity of the window list (grows) */

/* shared-state fan-out for cells [c0,c1): channel-major like the reference
 * (one tickData per channel, walking that channel's subscribers). */
static void *fan_cells(void *arg) {
    fan_job *j = (fan_job *)arg;
    orc_world *w = j->w;
    uint32_t wcap = MAXWIN;
    wwin *wins = (wwin *)malloc(sizeof(wwin) * wcap);
    for (uint32_t c = j->c0; c < j->c1; c++) {
        uint32_t ns = j->cell_off_sub[c + 1] - j->cell_off_sub[c];
        if (!ns) continue;
        const spref *subs = j->cell_subs + j->cell_off_sub[c];
        /* channel 0: the spatial channel itself; then each entity channel */
        uint32_t ne = j->cell_off_ent[c + 1] - j->cell_off_ent[c];
        for (uint32_t k = 0; k <= ne; k++) {
            const wbuf *b;
            uint32_t chan;
            if (k == 0) { b = &w->cbuf[c]; chan = w->g.id_start + c; }
            else {
                uint32_t e = j->cell_ents[j->cell_off_ent[c] + k - 1];
                b = &w->ebuf[e];
                chan = w->chan_id[e];
            }
            for (uint32_t si = 0; si < ns; si++) {
                wpair tmp = w->pairs[(size_t)subs[si].s * w->capq + subs[si].p];
                uint32_t conn = w->conn_id[subs[si].s];
                uint32_t nw = pair_windows(&tmp, j->t, &wins, &wcap);
                for (uint32_t wi = 0; wi < nw; wi++) {
                    uint32_t mask = 0;
                    if (wins[wi].full) job_push(j, subs[si].s, conn | REC_FULL, chan, 0, 0);
                    else if ((w->sorted_walk && !w->unsorted) ? window_has_update_sorted(w, b, wins[wi].last, wins[wi].next, conn, tmp.skip_self, &mask)
                                                              : window_has_update(w, b, wins[wi].last, wins[wi].next, conn, tmp.skip_self, &mask))
                        job_push(j, subs[si].s, conn, chan, mask, j->w->digest_only ? 0u : window_range(b, wins[wi].last, wins[wi].next));
                }
            }
        }
        /* commit the subscription state (identical for every channel above) */
        for (uint32_t si = 0; si < ns; si++) {
            wpair *p = &w->pairs[(size_t)subs[si].s * w->capq + subs[si].p];
            pair_windows(p, j->t, &wins, &wcap);
        }
    }
    free(wins);
    return NULL;
}

/* oracle-only: force a queue element's phase (model: inherit the cell phase) */
extern void orc__force_state(orc_channel *ch, uint32_t conn_id, orc_time last, int had_first);
extern int orc__get_state(const orc_channel *ch, uint32_t conn_id, orc_time *last, int *had_first);

static void fan_literal(orc_world *w, orc_time t, const uint32_t *cell_off_sub,
                        const spref *cell_subs, const uint32_t *cell_off_ent,
                        const uint32_t *cell_ents) {
    orc_send *sends = (orc_send *)malloc(sizeof(orc_send) * 65536);
    for (uint32_t c = 0; c < w->C; c++) {
        uint32_t ns = cell_off_sub[c + 1] - cell_off_sub[c];
        if (!ns) continue;
        const spref *subs = cell_subs + cell_off_sub[c];
        uint32_t ne = cell_off_ent[c + 1] - cell_off_ent[c];
        orc_channel *first_ch = NULL;
        for (uint32_t k = 0; k <= ne; k++) {
            const wbuf *b;
            uint32_t chan;
            if (k == 0) { b = &w->cbuf[c]; chan = w->g.id_start + c; }
            else {
                uint32_t e = cell_ents[cell_off_ent[c] + k - 1];
                b = &w->ebuf[e];
                chan = w->chan_id[e];
            }
            orc_channel *ch = orc_channel_new();
            orc_init_data(ch);
            for (uint32_t i = 0; i < b->len; i++)
                orc_on_update(ch, b->v[b->head + i].arrival, b->v[b->head + i].sender, i);
            for (uint32_t si = 0; si < ns; si++) {
                const wpair *p = &w->pairs[(size_t)subs[si].s * w->capq + subs[si].p];
                uint32_t conn = w->conn_id[subs[si].s];
                orc_subscribe(ch, conn, 0, p->interval_ms, 0, p->skip_self, 0, p->access);
                orc__force_state(ch, conn, p->last, p->had_first);
            }
            int n = orc_tick_data(ch, t, sends, 65536);
            for (int i = 0; i < n; i++)
                push_rec(w, sends[i].conn_id | (sends[i].full ? REC_FULL : 0), chan);
            if (k == 0) first_ch = ch;
            else {
                /* every channel's subscription state must have evolved alike */
                for (uint32_t si = 0; si < ns; si++) {
                    orc_time l0, l1; int f0, f1;
                    uint32_t conn = w->conn_id[subs[si].s];
                    orc__get_state(first_ch, conn, &l0, &f0);
                    orc__get_state(ch, conn, &l1, &f1);
                    if (l0 != l1 || f0 != f1) w->literal_mismatch++;
                }
                orc_channel_free(ch);
            }
        }
        for (uint32_t si = 0; si < ns; si++) {
            wpair *p = &w->pairs[(size_t)subs[si].s * w->capq + subs[si].p];
            orc_time l; int f;
            orc__get_state(first_ch, w->conn_id[subs[si].s], &l, &f);
            p->last = l;
            p->had_first = (uint8_t)f;
        }
        orc_channel_free(first_ch);
    }
    free(sends);
}

/*
 * One tick.  Updates: n_upd entries (idx NULL => entity i = u), positions,
 * sender (NULL => keep).  Cell-channel updates: n_cu (cell index, sender).
 * Interest updates: n_q entries (q_sub NULL => subscriber s = i).
 * Returns 0.
 */
/* upd_arrival / cu_arrival: arrivalTime of every update (Channel.PutMessage stamps a message when it is ENQUEUED,
 * channel.go:296-310; OnUpdate stores that stamp, data.go:159-164): NULL = the tick's own time.  Updates are applied in array
 * order, so an entity may appear several times (each its own Notify and its own buffer element, as the reference's handler
 * loop). */
int orc_world_tick_arrivals(orc_world *w, orc_time t, uint32_t n_upd, const uint32_t *idx, const double *x, const double *z,
                            const uint32_t *sender, const orc_time *upd_arrival, uint32_t n_cu, const uint32_t *cu_cell,
                            const uint32_t *cu_sender, const orc_time *cu_arrival, uint32_t n_q, const uint32_t *q_sub,
                            const orc_query *queries) {
    w->nrec = 0; w->nho = 0; w->nunsub = 0; w->n_locked_abort = 0; w->nrcp = 0;
    memmove(w->stamps + 1, w->stamps, sizeof(orc_time) * 31);
    w->stamps[0] = t;
    if (w->nstamps < 32) w->nstamps++;
    /* is_new = "subscribed during the latest tick" */
    for (uint32_t s = 0; s < w->S; s++)
        for (uint32_t p = 0; p < w->pair_cnt[s]; p++) w->pairs[(size_t)s * w->capq + p].is_new = 0;

    /* ---- 1. entity updates ---- */
    for (uint32_t u = 0; u < n_upd; u++) {
        uint32_t i = idx ? idx[u] : u;
        if (!w->alive[i]) continue;
        uint32_t src = w->cell[i];
        uint32_t dst = cell_index(w, x[u], z[u]);
        w->cell[i] = dst; /* merged position is now the new one */
        if (sender) w->sender[i] = sender[u];
        {
            const wbuf *eb = &w->ebuf[i];
            const orc_time a = upd_arrival ? upd_arrival[u] : t;
            if (eb->len && eb->v[eb->head + eb->len - 1].arrival > a) w->unsorted = 1;
            if (src != W_INVALID && w->emax[i] < w->cmax[src]) w->emax[i] = w->cmax[src];
            if (dst != W_INVALID && w->emax[i] < w->cmax[dst]) w->emax[i] = w->cmax[dst];
            wbuf_push(&w->ebuf[i], a, w->sender[i], w->emax[i]);
        }
        if (src == W_INVALID || dst == W_INVALID || src == dst) continue; /* spatial.go:613-626 */
        /* GetHandoverEntities (entity.go:197-224): the notifier's handover group; a locked member empties it (:675-679) */
        int any_locked = (w->eflags[i] & 1u) != 0;
        if (w->hl_has[i]) any_locked = any_locked || w->hl_n[i] == 0; /* len(handoverEntities) == 0: "No handover happens" (spatial.go:675-679) */
        else if (w->group[i])
            for (uint32_t m = 0; m < w->N && !any_locked; m++)
                if (w->alive[m] && w->group[m] == w->group[i] && (w->eflags[m] & 1u)) any_locked = 1;
        if (any_locked) { w->n_locked_abort++; continue; }
        if (w->nho == w->capho) {
            w->capho = w->capho ? w->capho * 2 : 256;
            w->ho_ent = (uint32_t *)realloc(w->ho_ent, 4 * w->capho);
            w->ho_src = (uint32_t *)realloc(w->ho_src, 4 * w->capho);
            w->ho_dst = (uint32_t *)realloc(w->ho_dst, 4 * w->capho);
            w->ho_srv_src = (uint32_t *)realloc(w->ho_srv_src, 4 * w->capho);
            w->ho_srv_dst = (uint32_t *)realloc(w->ho_srv_dst, 4 * w->capho);
        }
        w->ho_ent[w->nho] = i;
        w->ho_src[w->nho] = src + w->g.id_start;
        w->ho_dst[w->nho] = dst + w->g.id_start;
        w->ho_srv_src[w->nho] = w->server_of_cell[src];
        w->ho_srv_dst[w->nho] = w->server_of_cell[dst];
        {   /* the handover's entity list, in the order the message carries it (GetHandoverEntities, entity.go:197-224: the
             * notifier alone, or the live members of its list / group), and where each of them was held when Notify ran — the
             * notifier itself is taken as held by src */
            if (w->nho >= w->cap_ho_ent) {
                w->cap_ho_ent = w->cap_ho_ent ? w->cap_ho_ent * 2 : 256;
                w->ho_ent_before = (uint32_t *)realloc(w->ho_ent_before, 4 * 32 * (size_t)w->cap_ho_ent);
                w->ho_ent_n = (uint32_t *)realloc(w->ho_ent_n, 4 * (size_t)w->cap_ho_ent);
            }
            uint32_t *bf = w->ho_ent_before + 32 * (size_t)w->nho, q = 0;
            if (w->hl_has[i]) {
                for (uint32_t k = 0; k < w->hl_n[i] && q < 32; k++) {
                    uint32_t m = w->hl_mem[i][k];
                    if (m == i) bf[q++] = src;
                    else if (m < w->N && w->alive[m]) bf[q++] = w->member[m];
                }
            } else if (w->group[i]) {
                for (uint32_t m = 0; m < w->N && q < 32; m++)
                    if (w->alive[m] && w->group[m] == w->group[i]) bf[q++] = m == i ? src : w->member[m];
            } else {
                bf[q++] = src;
            }
            w->ho_ent_n[w->nho] = q;
        }
        w->nho++;
        if (w->hl_has[i]) { /* RemoveEntity(src) + AddEntity(dst) over handoverEntities — the notifier only if it is one of them */
            for (uint32_t q = 0; q < w->hl_n[i]; q++) {
                uint32_t m = w->hl_mem[i][q];
                if (m == i) w->member[m] = dst; /* (the notifier: RemoveEntity(src) may fail, AddEntity(dst) happens) */
                else if (m < w->N && w->alive[m] && w->member[m] == src) w->member[m] = dst;
            }
            continue;
        }
        w->member[i] = dst; /* RemoveEntity(src) + AddEntity(dst), :703-736 */
        if (w->group[i]) /* ... for every entity of the handover group that is in src's map */
            for (uint32_t m = 0; m < w->N; m++)
                if (m != i && w->alive[m] && w->group[m] == w->group[i] && w->member[m] == src) w->member[m] = dst;
    }
    for (uint32_t u = 0; u < n_cu; u++) {
        const wbuf *cb = &w->cbuf[cu_cell[u]];
        const orc_time a = cu_arrival ? cu_arrival[u] : t;
        if (cb->len && cb->v[cb->head + cb->len - 1].arrival > a) w->unsorted = 1;
        wbuf_push(&w->cbuf[cu_cell[u]], a, cu_sender[u], w->cmax[cu_cell[u]]);
    }

    /* ---- 1b. who receives each handover's ChannelDataHandoverMessage (spatial.go:776-857) ----
     * srcChannelSubConns / dstChannelSubConns as they are now (Notify runs before this tick's
     * interest updates).  Step 4-1: connections of src that are not in dst get the message without
     * per-recipient entity data (kind 0).  Step 4-2: connections of dst are subscribed to the entity
     * channel with DataAccess = WRITE for the entity channel's owner else READ (:812-817); `shouldSend`
     * = newly subscribed OR the merge changed DataAccess (subscription.go:44-57) -> that entity's full
     * data.  In the tick model the entity channel's subscribers are the subscribers of the cell that
     * held it, and its owner is that cell's spatial server — after step 1 of a cross-server handover
     * (:683-700, SetOwner(dstChannel.GetOwner()) for EVERY handover entity) the dst cell's. */
    if (w->nho > w->cap_ho_own) {
        w->cap_ho_own = w->nho + 256;
        w->ho_own_unsub = (uint8_t *)realloc(w->ho_own_unsub, w->cap_ho_own);
    }
    for (uint32_t h = 0; h < w->nho; h++) {
        uint32_t src = w->ho_src[h] - w->g.id_start, dst = w->ho_dst[h] - w->g.id_start;
        /* step 1 of a CROSS-SERVER handover (spatial.go:683-700): `ownerConn := srcChannel.GetOwner(); ownerConn != nil &&
         * !ownerConn.IsClosing() && !ownerConn.HasInterestIn(dstChannelId)` -> UnsubscribeFromChannel(entityCh) + sendUnsubscribed for
         * every handover entity.  HasInterestIn = dst is among the connection's spatialSubscriptions (subscription.go:181-187).  The
         * owner's connection is known where orc_world_set_server_conns named it AND it is registered as a subscriber. */
        w->ho_own_unsub[h] = 0;
        if (w->server_conn && w->server_of_cell[src] != w->server_of_cell[dst] && w->server_of_cell[src] < w->n_server_conn)
            for (uint32_t s = 0; s < w->S; s++) {
                if (!w->sub_alive[s] || w->conn_id[s] != w->server_conn[w->server_of_cell[src]]) continue;
                int in_dst = 0;
                const wpair *pp = &w->pairs[(size_t)s * w->capq];
                for (uint32_t p = 0; p < w->pair_cnt[s]; p++)
                    if (pp[p].cell == dst) in_dst = 1;
                w->ho_own_unsub[h] = !in_dst;
                break;
            }
        for (uint32_t s = 0; s < w->S; s++) {
            if (!w->sub_alive[s]) continue;
            int in_src = 0, in_dst = 0;
            const wpair *pp = &w->pairs[(size_t)s * w->capq];
            for (uint32_t p = 0; p < w->pair_cnt[s]; p++) {
                if (pp[p].cell == src) in_src = 1;
                if (pp[p].cell == dst) in_dst = 1;
            }
            int kind = -1;
            if (in_dst) kind = in_src ? 2 : 1;
            else if (in_src) kind = 0;
            if (kind < 0) continue;
            if (w->nrcp == w->caprcp) {
                w->caprcp = w->caprcp ? w->caprcp * 2 : 1024;
                w->rcp_ho = (uint32_t *)realloc(w->rcp_ho, 4 * w->caprcp);
                w->rcp_conn = (uint32_t *)realloc(w->rcp_conn, 4 * w->caprcp);
                w->rcp_kind = (uint8_t *)realloc(w->rcp_kind, w->caprcp);
                w->rcp_mask = (uint32_t *)realloc(w->rcp_mask, 4 * w->caprcp);
            }
            w->rcp_ho[w->nrcp] = h;
            w->rcp_conn[w->nrcp] = w->conn_id[s];
            w->rcp_kind[w->nrcp] = (uint8_t)kind;
            {   /* step 4-2 per entity: full data iff the connection was not yet subscribed to that entity's channel */
                uint32_t mask = 0;
                if (kind != 0)
                    for (uint32_t q = 0; q < w->ho_ent_n[h]; q++) {
                        uint32_t bc = w->ho_ent_before[32 * (size_t)h + q];
                        int known = 0;
                        if (bc != W_INVALID)
                            for (uint32_t p = 0; p < w->pair_cnt[s]; p++)
                                if (pp[p].cell == bc) { known = 1; break; }
                        int changed = 0; /* dataAccessChanged: the connection is the old or the new owner's, and they differ */
                        if (known && w->server_conn) {
                            const uint32_t old_srv = w->server_of_cell[bc];
                            const uint32_t new_srv = w->server_of_cell[src] != w->server_of_cell[dst] ? w->server_of_cell[dst] : old_srv;
                            const int was = old_srv < w->n_server_conn && w->server_conn[old_srv] == w->conn_id[s];
                            const int is = new_srv < w->n_server_conn && w->server_conn[new_srv] == w->conn_id[s];
                            changed = was != is;
                        }
                        if (!known || changed) mask |= 1u << q;
                    }
                w->rcp_mask[w->nrcp] = mask;
            }
            w->nrcp++;
        }
    }

    /* ---- 2. interest updates (message_spatial.go:59-128) ---- */
    if (w->nq_status < n_q) {
        w->q_status = (int32_t *)realloc(w->q_status, 4 * (n_q + 1));
        w->nq_status = n_q;
    }
    {
        uint32_t *ids = (uint32_t *)malloc(4 * (w->C + 1));
        uint32_t *dists = (uint32_t *)malloc(4 * (w->C + 1));
        wpair *np = (wpair *)malloc(sizeof(wpair) * (w->capq + 1));
        for (uint32_t qi = 0; qi < n_q; qi++) {
            uint32_t s = q_sub ? q_sub[qi] : qi;
            uint32_t n = 0;
            if (!w->sub_alive[s]) { /* GetConnection == nil, message_spatial.go:53-57 */
                w->q_status[qi] = -2;
                continue;
            }
            int rc = orc_query_channel_ids(&w->g, &queries[qi], ids, dists, w->C, &n);
            if (rc == ORC_OK && n > w->capq) rc = ORC_E_CAP;
            w->q_status[qi] = rc;
            if (rc != ORC_OK) continue; /* error: nothing changes (:60-63) */
            wpair *old = &w->pairs[(size_t)s * w->capq];
            uint32_t nold = w->pair_cnt[s];
            /* Difference(existing, new) -> unsub */
            for (uint32_t o = 0; o < nold; o++) {
                int found = 0;
                for (uint32_t k = 0; k < n; k++)
                    if (ids[k] - w->g.id_start == old[o].cell) { found = 1; break; }
                if (found) continue;
                if (w->nunsub == w->capunsub) {
                    w->capunsub = w->capunsub ? w->capunsub * 2 : 256;
                    w->unsub_sub = (uint32_t *)realloc(w->unsub_sub, 4 * w->capunsub);
                    w->unsub_cell = (uint32_t *)realloc(w->unsub_cell, 4 * w->capunsub);
                }
                w->unsub_sub[w->nunsub] = s;
                w->unsub_cell[w->nunsub] = old[o].cell + w->g.id_start;
                w->nunsub++;
            }
            /* every new id is (re)subscribed with the damped interval */
            for (uint32_t k = 0; k < n; k++) {
                uint32_t c = ids[k] - w->g.id_start;
                uint32_t iv = w->default_interval_ms; /* getSpatialDampingSettings: first entry with dist <= MaxDistance, else the default */
                if (w->n_damp == 0) iv = orc_damping_interval(dists[k], w->default_interval_ms);
                else
                    for (uint32_t di = 0; di < w->n_damp; di++)
                        if (dists[k] <= w->damp_dist[di]) { iv = w->damp_iv[di]; break; }
                const wpair *ex = NULL;
                for (uint32_t o = 0; o < nold; o++)
                    if (old[o].cell == c) { ex = &old[o]; break; }
                if (ex) { /* subscription.go:44-57: options merged, state kept */
                    np[k] = *ex;
                    np[k].interval_ms = iv;
                    np[k].is_new = 0;
                } else { /* subscription.go:59-91 */
                    np[k].cell = c;
                    np[k].interval_ms = iv;
                    np[k].last = t + (orc_time)w->default_delay_ms * 1000000;
                    np[k].had_first = 0;   /* SkipFirstFanOut default false */
                    np[k].skip_self = 1;   /* SkipSelfUpdateFanOut default true */
                    np[k].access = ORC_ACCESS_READ;
                    np[k].is_new = 1;
                    if (w->cmax[c] < iv) w->cmax[c] = iv; /* :83-86 (a new subscription only) */
                }
            }
            memcpy(old, np, sizeof(wpair) * n);
            w->pair_cnt[s] = n;
        }
        free(ids); free(dists); free(np);
    }

    /* ---- 3. fan-out (data.go:175-291 on every channel) ---- */
    uint32_t *cell_off_sub = (uint32_t *)calloc(w->C + 2, 4);
    uint32_t *cell_off_ent = (uint32_t *)calloc(w->C + 2, 4);
    for (uint32_t s = 0; s < w->S; s++) {
        if (!w->sub_alive[s]) continue;
        for (uint32_t p = 0; p < w->pair_cnt[s]; p++)
            cell_off_sub[w->pairs[(size_t)s * w->capq + p].cell + 1]++;
    }
    for (uint32_t i = 0; i < w->N; i++)
        if (w->alive[i] && w->member[i] != W_INVALID) cell_off_ent[w->member[i] + 1]++;
    for (uint32_t c = 0; c < w->C; c++) {
        cell_off_sub[c + 1] += cell_off_sub[c];
        cell_off_ent[c + 1] += cell_off_ent[c];
    }
    spref *cell_subs = (spref *)malloc(sizeof(spref) * (cell_off_sub[w->C] + 1));
    uint32_t *cell_ents = (uint32_t *)malloc(4 * (cell_off_ent[w->C] + 1));
    {
        uint32_t *cur = (uint32_t *)malloc(4 * (w->C + 1));
        memcpy(cur, cell_off_sub, 4 * w->C);
        for (uint32_t s = 0; s < w->S; s++) {
            if (!w->sub_alive[s]) continue;
            for (uint32_t p = 0; p < w->pair_cnt[s]; p++) {
                uint32_t c = w->pairs[(size_t)s * w->capq + p].cell;
                cell_subs[cur[c]].s = s;
                cell_subs[cur[c]].p = p;
                cur[c]++;
            }
        }
        memcpy(cur, cell_off_ent, 4 * w->C);
        for (uint32_t i = 0; i < w->N; i++)
            if (w->alive[i] && w->member[i] != W_INVALID) cell_ents[cur[w->member[i]]++] = i;
        free(cur);
    }
    if (w->literal) {
        fan_literal(w, t, cell_off_sub, cell_subs, cell_off_ent, cell_ents);
    } else {
        int nt = w->threads;
        if ((uint32_t)nt > w->C) nt = (int)w->C;
        if (w->njobs != nt) {
            for (int k = 0; k < w->njobs; k++) { free(w->jobs[k].rec); free(w->jobs[k].d_conn); }
            free(w->jobs);
            w->jobs = (fan_job *)calloc((size_t)nt, sizeof(fan_job));
            w->njobs = nt;
        }
        fan_job *jobs = w->jobs;
        pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
        /* static partition of the cells (channels) over the host threads,
         * balanced by work = subs x (entities + 1) */
        uint64_t total = 0;
        for (uint32_t c = 0; c < w->C; c++)
            total += (uint64_t)(cell_off_sub[c + 1] - cell_off_sub[c]) * (cell_off_ent[c + 1] - cell_off_ent[c] + 1);
        uint32_t c = 0;
        uint64_t acc = 0;
        for (int k = 0; k < nt; k++) {
            jobs[k].w = w; jobs[k].t = t; jobs[k].nrec = 0;
            jobs[k].d_sum = jobs[k].d_xor = jobs[k].d_sum_masked = 0;
            if (w->digest_only) {
                if (!jobs[k].d_conn) jobs[k].d_conn = (uint64_t *)malloc(8 * ((size_t)w->S + 1));
                memset(jobs[k].d_conn, 0, 8 * ((size_t)w->S + 1));
            }
            jobs[k].cell_off_sub = cell_off_sub; jobs[k].cell_subs = cell_subs;
            jobs[k].cell_off_ent = cell_off_ent; jobs[k].cell_ents = cell_ents;
            jobs[k].c0 = c;
            uint64_t target = total * (uint64_t)(k + 1) / (uint64_t)nt;
            while (c < w->C && (acc < target || k == nt - 1)) {
                acc += (uint64_t)(cell_off_sub[c + 1] - cell_off_sub[c]) * (cell_off_ent[c + 1] - cell_off_ent[c] + 1);
                c++;
            }
            jobs[k].c1 = c;
        }
        if (nt == 1) fan_cells(&jobs[0]);
        else {
            for (int k = 0; k < nt; k++) pthread_create(&th[k], NULL, fan_cells, &jobs[k]);
            for (int k = 0; k < nt; k++) pthread_join(th[k], NULL);
        }
        w->nrec = 0;
        for (int k = 0; k < nt; k++) w->nrec += jobs[k].nrec;
        if (w->digest_only) {
            if (!w->d_conn) w->d_conn = (uint64_t *)malloc(8 * ((size_t)w->S + 1));
            memset(w->d_conn, 0, 8 * ((size_t)w->S + 1));
            w->d_cnt = w->nrec; w->d_sum = w->d_xor = w->d_sum_masked = 0;
            for (int k = 0; k < nt; k++) {
                w->d_sum += jobs[k].d_sum; w->d_xor ^= jobs[k].d_xor; w->d_sum_masked += jobs[k].d_sum_masked;
                for (uint32_t q = 0; q < w->S; q++) w->d_conn[q] += jobs[k].d_conn[q];
            }
        }
        free(th);
    }
    free(cell_off_sub); free(cell_off_ent); free(cell_subs); free(cell_ents);
    return 0;
}

int orc_world_tick(orc_world *w, orc_time t, uint32_t n_upd, const uint32_t *idx,
                   const double *x, const double *z, const uint32_t *sender,
                   uint32_t n_cu, const uint32_t *cu_cell, const uint32_t *cu_sender,
                   uint32_t n_q, const uint32_t *q_sub, const orc_query *queries) {
    return orc_world_tick_arrivals(w, t, n_upd, idx, x, z, sender, NULL, n_cu, cu_cell, cu_sender, NULL, n_q, q_sub, queries);
}

/* ---- accessors ---- */
uint64_t orc_world_nrec(const orc_world *w) { return w->nrec; }
/* digest mode: the window formulation folds every record into {count, sum, xor of mix64(conn << 32 | channel), sum of
 * mix64(that hash + merged-updates mask)} and per subscriber slot the sum of the hashes, instead of storing it */
void orc_world_set_digest_only(orc_world *w, int on) { w->digest_only = on && !w->literal; }
void orc_world_set_sorted_walk(orc_world *w, int on) { w->sorted_walk = on; }
int orc_world_unsorted(const orc_world *w) { return w->unsorted; }
void orc_world_digest(const orc_world *w, uint64_t out[4], uint64_t *conn_sum) {
    out[0] = w->d_cnt; out[1] = w->d_sum; out[2] = w->d_xor; out[3] = w->d_sum_masked;
    if (conn_sum && w->d_conn) memcpy(conn_sum, w->d_conn, 8 * (size_t)w->S);
}
void orc_world_records(const orc_world *w, uint32_t *conn, uint32_t *chan) {
    if (w->literal) {
        for (uint64_t i = 0; i < w->nrec; i++) { conn[i] = w->rec[i].conn; chan[i] = w->rec[i].chan; }
        return;
    }
    uint64_t o = 0;
    for (int k = 0; k < w->njobs; k++)
        for (uint64_t i = 0; i < w->jobs[k].nrec; i++, o++) { conn[o] = w->jobs[k].rec[i].conn; chan[o] = w->jobs[k].rec[i].chan; }
}
/* window mode only: per record (same order as orc_world_records) the merged-updates mask */
/* window mode only: per record the range form of the merged updates (0 for full-state records) */
void orc_world_record_ranges(const orc_world *w, uint32_t *range) {
    if (w->literal) {
        memset(range, 0, 4 * w->nrec);
        return;
    }
    uint64_t o = 0;
    for (int k = 0; k < w->njobs; k++)
        for (uint64_t i = 0; i < w->jobs[k].nrec; i++, o++) range[o] = w->jobs[k].rec[i].range;
}
void orc_world_record_masks(const orc_world *w, uint32_t *mask) {
    if (w->literal) {
        memset(mask, 0, 4 * w->nrec);
        return;
    }
    uint64_t o = 0;
    for (int k = 0; k < w->njobs; k++)
        for (uint64_t i = 0; i < w->jobs[k].nrec; i++, o++) mask[o] = w->jobs[k].rec[i].mask;
}
uint32_t orc_world_nhandover(const orc_world *w) { return w->nho; }
void orc_world_handovers(const orc_world *w, uint32_t *ent, uint32_t *src, uint32_t *dst,
                         uint32_t *srv_src, uint32_t *srv_dst) {
    memcpy(ent, w->ho_ent, 4 * w->nho); memcpy(src, w->ho_src, 4 * w->nho);
    memcpy(dst, w->ho_dst, 4 * w->nho); memcpy(srv_src, w->ho_srv_src, 4 * w->nho);
    memcpy(srv_dst, w->ho_srv_dst, 4 * w->nho);
}
void orc_world_handover_owner_unsubs(const orc_world *w, uint8_t *flags) { if (w->nho) memcpy(flags, w->ho_own_unsub, w->nho); }
uint64_t orc_world_nrcp(const orc_world *w) { return w->nrcp; }
void orc_world_recipient_masks(const orc_world *w, uint32_t *mask) { memcpy(mask, w->rcp_mask, 4 * w->nrcp); }
void orc_world_recipients(const orc_world *w, uint32_t *ho, uint32_t *conn, uint8_t *kind) {
    memcpy(ho, w->rcp_ho, 4 * w->nrcp); memcpy(conn, w->rcp_conn, 4 * w->nrcp); memcpy(kind, w->rcp_kind, w->nrcp);
}

/* BroadcastType_ADJACENT_CHANNELS, message.go:188-239, for the client connections of the world:
 * GetAdjacentChannels (+ the centre unless ALL_BUT_OWNER), merge of the channels' connections into
 * one set, then the flag filters in the reference's order.  Returns the number of connection ids
 * written (ascending connection slot). */
uint32_t orc_world_adjacent_recipients(const orc_world *w, uint32_t channel, uint32_t broadcast, uint32_t sender_conn,
                                       uint32_t client_conn, uint32_t *out) {
    uint32_t ids[9];
    uint32_t n = orc_adjacent(&w->g, channel, ids);
    if (!(broadcast & 8u)) ids[n++] = channel; /* !ALL_BUT_OWNER.Check -> append the centre (:201-204) */
    uint32_t k = 0;
    for (uint32_t s = 0; s < w->S; s++) {
        if (!w->sub_alive[s]) continue;
        int hit = 0;
        const wpair *pp = &w->pairs[(size_t)s * w->capq];
        for (uint32_t p = 0; p < w->pair_cnt[s] && !hit; p++)
            for (uint32_t j = 0; j < n; j++)
                if (pp[p].cell + w->g.id_start == ids[j]) { hit = 1; break; }
        if (!hit) continue;
        uint32_t cid = w->conn_id[s];
        if ((broadcast & 4u) && cid == sender_conn) continue; /* ALL_BUT_SENDER (:223-225) */
        if (broadcast & 16u) continue;                         /* ALL_BUT_CLIENT: these are client connections (:227-229) */
        if (cid == client_conn) continue;                       /* the client named in the ServerForwardMessage (:235-237) */
        out[k++] = cid;
    }
    return k;
}

uint32_t orc_world_nunsub(const orc_world *w) { return w->nunsub; }
void orc_world_unsubs(const orc_world *w, uint32_t *sub, uint32_t *cell) {
    memcpy(sub, w->unsub_sub, 4 * w->nunsub); memcpy(cell, w->unsub_cell, 4 * w->nunsub);
}
void orc_world_query_status(const orc_world *w, int32_t *out, uint32_t n) { memcpy(out, w->q_status, 4 * n); }
uint32_t orc_world_locked_aborts(const orc_world *w) { return w->n_locked_abort; }
/* update buffer of entity channel i: elements held; the channel's maxFanOutIntervalMs (entity i / spatial channel c) */
uint32_t orc_world_entity_buffer_len(const orc_world *w, uint32_t i) { return i < w->N ? w->ebuf[i].len : 0; }
uint32_t orc_world_entity_max_interval(const orc_world *w, uint32_t i) { return i < w->N ? w->emax[i] : 0; }
uint32_t orc_world_cell_max_interval(const orc_world *w, uint32_t c) { return c < w->C ? w->cmax[c] : 0; }
uint64_t orc_world_literal_mismatch(const orc_world *w) { return w->literal_mismatch; }
void orc_world_entity_state(const orc_world *w, uint32_t *cell, uint32_t *member) {
    memcpy(cell, w->cell, 4 * w->N); memcpy(member, w->member, 4 * w->N);
}
uint32_t orc_world_pairs(const orc_world *w, uint32_t s, uint32_t *cell, uint32_t *interval_ms,
                         int64_t *last, uint8_t *had_first, uint8_t *is_new) {
    uint32_t n = w->pair_cnt[s];
    for (uint32_t p = 0; p < n; p++) {
        const wpair *q = &w->pairs[(size_t)s * w->capq + p];
        cell[p] = q->cell + w->g.id_start; interval_ms[p] = q->interval_ms; last[p] = q->last;
        had_first[p] = q->had_first; is_new[p] = q->is_new;
    }
    return n;
}
