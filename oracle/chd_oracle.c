/*
 * chd_oracle.c — CPU restatement (plain C) of channeld's SpatialChannel hot
 * path.  TEST INFRASTRUCTURE ONLY — see chd_oracle.h.  Sequential loops, the
 * reference's iteration order, no FMA (compile with -ffp-contract=off).
 */
#include "chd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Go math restated                                                    */
/* ------------------------------------------------------------------ */

/* math.Min, Go src/math/dim.go: -Inf wins, then NaN, then signed zeros. */
double orc_go_min(double x, double y) {
    if ((isinf(x) && x < 0) || (isinf(y) && y < 0)) return -INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}

/* math.Max, Go src/math/dim.go: +Inf wins, then NaN, then signed zeros. */
double orc_go_max(double x, double y) {
    if ((isinf(x) && x > 0) || (isinf(y) && y > 0)) return INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}

/* math.Cos for |x| < 2^29, Go src/math/sin.go (Cephes cosl port).
 * Larger arguments use Payne-Hanek in Go; angles never get there, so the
 * oracle falls back to libm there (flagged in the header as unpinned). */
double orc_go_cos(double x) {
    static const double PI4A = 7.85398125648498535156e-1;
    static const double PI4B = 3.77489470793079817668e-8;
    static const double PI4C = 2.69515142907905952645e-15;
    static const double S[6] = {
        1.58962301576546568060e-10, -2.50507477628578072866e-8,
        2.75573136213857245213e-6,  -1.98412698295895385996e-4,
        8.33333333332211858878e-3,  -1.66666666666666307295e-1};
    static const double C[6] = {
        -1.13585365213876817300e-11, 2.08757008419747316778e-9,
        -2.75573141792967388112e-7,  2.48015872888517045348e-5,
        -1.38888888888730564116e-3,  4.16666666666665929218e-2};
    if (isnan(x) || isinf(x)) return NAN;
    int sign = 0;
    x = fabs(x);
    if (x >= (double)(1 << 29)) return cos(x);
    /* 4/Pi as a float64 constant */
    uint64_t j = (uint64_t)(x * 1.2732395447351628);
    double y = (double)j;
    if (j & 1) {
        j++;
        y++;
    }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) {
        j -= 4;
        sign = !sign;
    }
    if (j > 1) sign = !sign;
    double zz = z * z;
    if (j == 1 || j == 2) {
        y = z + z * zz * ((((((S[0] * zz) + S[1]) * zz + S[2]) * zz + S[3]) * zz + S[4]) * zz + S[5]);
    } else {
        y = 1.0 - 0.5 * zz + zz * zz * ((((((C[0] * zz) + C[1]) * zz + C[2]) * zz + C[3]) * zz + C[4]) * zz + C[5]);
    }
    return sign ? -y : y;
}

/* ------------------------------------------------------------------ */
/* config, cell ids                                                    */
/* ------------------------------------------------------------------ */

int orc_validate_config(const orc_grid *g, int *which) {
    int w = 0;
    /* spatial.go:146-157, in this order */
    if (g->grid_w <= 0 || g->grid_h <= 0) w = 1;
    else if (g->cols == 0 || g->rows == 0) w = 2;
    else if (g->server_cols == 0 || g->server_rows == 0) w = 3;
    else if (g->border == 0) w = 4;
    if (which) *which = w;
    return w ? ORC_E_CONFIG : ORC_OK;
}

double orc_grid_size(const orc_grid *g) {
    /* spatial.go:134-139 (cached there; same value every time) */
    if (g->grid_w > 0 && g->grid_h > 0)
        return sqrt(g->grid_w * g->grid_w + g->grid_h * g->grid_h);
    return 0;
}

/* int(math.Floor(v)) followed by `< 0 || >= n`: on amd64 NaN / out-of-range
 * converts to MinInt64 (negative) => error.  Same predicate on the double. */
static int grid_coord(double v, uint32_t n, uint32_t *out) {
    double f = floor(v);
    if (!(f >= 0.0) || !(f < (double)n)) return 0;
    *out = (uint32_t)f;
    return 1;
}

uint32_t orc_channel_id_with_offset(const orc_grid *g, double x, double z,
                                    double off_x, double off_z) {
    uint32_t gx, gy;
    /* spatial.go:170-177: X is tested first, then Z */
    if (!grid_coord((x - off_x) / g->grid_w, g->cols, &gx)) return 0;
    if (!grid_coord((z - off_z) / g->grid_h, g->rows, &gy)) return 0;
    return gx + gy * g->cols + g->id_start;
}

uint32_t orc_channel_id(const orc_grid *g, double x, double z) {
    return orc_channel_id_with_offset(g, x, z, g->off_x, g->off_z);
}

uint32_t orc_channel_id_no_offset(const orc_grid *g, double x, double z) {
    return orc_channel_id_with_offset(g, x, z, 0, 0);
}

void orc_channel_ids(const orc_grid *g, const double *x, const double *z,
                     uint32_t n, uint32_t *out) {
    for (uint32_t i = 0; i < n; i++) out[i] = orc_channel_id(g, x[i], z[i]);
}

/* ------------------------------------------------------------------ */
/* QueryChannelIds                                                     */
/* ------------------------------------------------------------------ */

/* the Go map[ChannelId]uint as a dense table over the grid */
typedef struct {
    uint8_t *present;
    uint32_t *dist;
} qmap;

static double dist2d(double ax, double az, double bx, double bz) {
    /* common.go:44-46 */
    return sqrt((ax - bx) * (ax - bx) + (az - bz) * (az - bz));
}

/* uint(math.Ceil(d)) — Go uint is 64-bit; the oracle keeps the low 32 bits
 * of the saturating conversion (values here are tiny). */
static uint32_t go_uint_ceil(double d) {
    double c = ceil(d);
    if (!(c >= 0.0)) return 0; /* NaN / negative: unspecified in Go, unused */
    if (c >= 4294967295.0) return 4294967295u;
    return (uint32_t)c;
}

#define ORC_MAX_AXIS_STEPS (1u << 22)

int orc_query_channel_ids(const orc_grid *g, const orc_query *q,
                          uint32_t *ids, uint32_t *dists, uint32_t cap,
                          uint32_t *n_out) {
    *n_out = 0;
    if (q == NULL) return ORC_E_NILQUERY; /* spatial.go:183-185 */
    uint32_t ncell = g->cols * g->rows;
    qmap m;
    m.present = (uint8_t *)calloc(ncell ? ncell : 1, 1);
    m.dist = (uint32_t *)calloc(ncell ? ncell : 1, sizeof(uint32_t));
    int rc = ORC_OK;
    const double gsz = orc_grid_size(g);

#define PUT(chid, d)                                  \
    do {                                              \
        uint32_t _i = (chid) - g->id_start;           \
        m.present[_i] = 1;                            \
        m.dist[_i] = (d);                             \
    } while (0)

    if (q->shapes & ORC_SHAPE_SPOTS) { /* spatial.go:189-202 */
        for (uint32_t i = 0; i < q->n_spots; i++) {
            uint32_t ch = orc_channel_id(g, q->spot_x[i], q->spot_z[i]);
            if (!ch) continue;
            if (i < q->n_spot_dists) PUT(ch, q->spot_dist[i]);
            else PUT(ch, 0);
        }
    }

    if (q->shapes & ORC_SHAPE_BOX) { /* spatial.go:204-233 */
        double cx = q->box_cx, cz = q->box_cz;
        double step_z = orc_go_min(q->box_ez, g->grid_h) * 0.5;
        if (step_z <= 0) { rc = ORC_E_EXTENT; goto done; }
        double step_x = orc_go_min(q->box_ex, g->grid_w) * 0.5;
        if (step_x <= 0) { rc = ORC_E_EXTENT; goto done; }
        uint32_t guard_z = 0;
        for (double z = cz - q->box_ez; z <= cz + q->box_ez; z += step_z) {
            if (++guard_z > ORC_MAX_AXIS_STEPS || z + step_z == z) { rc = ORC_E_HANG; goto done; }
            uint32_t guard_x = 0;
            for (double x = cx - q->box_ex; x <= cx + q->box_ex; x += step_x) {
                if (++guard_x > ORC_MAX_AXIS_STEPS || x + step_x == x) { rc = ORC_E_HANG; goto done; }
                uint32_t ch = orc_channel_id(g, x, z);
                if (!ch) continue;
                PUT(ch, go_uint_ceil(dist2d(cx, cz, x, z) / gsz));
            }
        }
        uint32_t cch = orc_channel_id(g, cx, cz);
        if (!cch) { rc = ORC_E_CENTER; goto done; }
        PUT(cch, 0);
    }

    if (q->shapes & ORC_SHAPE_SPHERE) { /* spatial.go:235-268 */
        double r = q->sph_r, cx = q->sph_cx, cz = q->sph_cz;
        double step_z = orc_go_min(r, g->grid_h) * 0.5;
        if (step_z <= 0) { rc = ORC_E_EXTENT; goto done; }
        double step_x = orc_go_min(r, g->grid_w) * 0.5;
        if (step_x <= 0) { rc = ORC_E_EXTENT; goto done; }
        uint32_t guard_z = 0;
        for (double z = cz - r; z <= cz + r; z += step_z) {
            if (++guard_z > ORC_MAX_AXIS_STEPS || z + step_z == z) { rc = ORC_E_HANG; goto done; }
            uint32_t guard_x = 0;
            for (double x = cx - r; x <= cx + r; x += step_x) {
                if (++guard_x > ORC_MAX_AXIS_STEPS || x + step_x == x) { rc = ORC_E_HANG; goto done; }
                if ((x - cx) * (x - cx) + (z - cz) * (z - cz) > r * r) continue;
                uint32_t ch = orc_channel_id(g, x, z);
                if (!ch) continue;
                PUT(ch, go_uint_ceil(dist2d(cx, cz, x, z) / gsz));
            }
        }
        uint32_t cch = orc_channel_id(g, cx, cz);
        if (!cch) { rc = ORC_E_CENTER; goto done; }
        PUT(cch, 0);
    }

    if (q->shapes & ORC_SHAPE_CONE) { /* spatial.go:270-314 */
        double r = q->cone_r, cx = q->cone_cx, cz = q->cone_cz;
        double ddx = q->cone_dx, ddz = q->cone_dz;
        { /* coneDir.Normalize2D(), common.go:56-60 */
            double mag = sqrt(ddx * ddx + ddz * ddz);
            ddx /= mag;
            ddz /= mag;
        }
        double step_z = orc_go_min(r, g->grid_h) * 0.5;
        if (step_z <= 0) { rc = ORC_E_EXTENT; goto done; }
        double step_x = orc_go_min(r, g->grid_w) * 0.5;
        if (step_x <= 0) { rc = ORC_E_EXTENT; goto done; }
        const double world_w = g->grid_w * (double)g->cols;
        const double world_h = g->grid_h * (double)g->rows;
        const double z_hi = orc_go_min(g->off_z + world_h, cz + r);
        const double x_hi = orc_go_min(g->off_x + world_w, cx + r);
        uint32_t guard_z = 0;
        for (double z = orc_go_max(g->off_z, cz - r); z <= z_hi; z += step_z) {
            if (++guard_z > ORC_MAX_AXIS_STEPS || z + step_z == z) { rc = ORC_E_HANG; goto done; }
            uint32_t guard_x = 0;
            for (double x = orc_go_max(g->off_x, cx - r); x <= x_hi; x += step_x) {
                if (++guard_x > ORC_MAX_AXIS_STEPS || x + step_x == x) { rc = ORC_E_HANG; goto done; }
                if ((x - cx) * (x - cx) + (z - cz) * (z - cz) > r * r) continue;
                double vx = x - cx, vz = z - cz;
                { /* dir.Normalize2D() */
                    double mag = sqrt(vx * vx + vz * vz);
                    vx /= mag;
                    vz /= mag;
                }
                double dot = vx * ddx + vz * ddz; /* common.go:48-50 */
                double c = q->use_cone_cos ? q->cone_cos : orc_go_cos(q->cone_angle);
                const double epsilon = 0.0;
                if (dot < c - epsilon) continue; /* NaN passes */
                uint32_t ch = orc_channel_id(g, x, z);
                if (!ch) continue;
                PUT(ch, go_uint_ceil(dist2d(cx, cz, x, z) / gsz));
            }
        }
        uint32_t cch = orc_channel_id(g, cx, cz);
        if (!cch) { rc = ORC_E_CENTER; goto done; }
        PUT(cch, 0);
    }
#undef PUT

    {
        uint32_t n = 0;
        for (uint32_t i = 0; i < ncell; i++) {
            if (!m.present[i]) continue;
            if (n >= cap) { rc = ORC_E_CAP; n = 0; break; }
            ids[n] = i + g->id_start;
            dists[n] = m.dist[i];
            n++;
        }
        *n_out = n;
    }
done:
    if (rc != ORC_OK) *n_out = 0;
    free(m.present);
    free(m.dist);
    return rc;
}

uint32_t orc_damping_interval(uint32_t dist, uint32_t default_interval_ms) {
    /* message_spatial.go:16-38: first entry with dist <= MaxDistance */
    static const uint32_t max_dist[3] = {0, 1, 2};
    static const uint32_t interval[3] = {20, 50, 100};
    for (int i = 0; i < 3; i++)
        if (dist <= max_dist[i]) return interval[i];
    return default_interval_ms; /* nil settings branch :68-72 */
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

void orc_interest_diff(const uint32_t *existing, uint32_t n_existing,
                       const uint32_t *new_ids, uint32_t n_new,
                       uint32_t *to_unsub, uint32_t *n_unsub,
                       uint8_t *is_new) {
    uint32_t nu = 0;
    for (uint32_t i = 0; i < n_existing; i++) { /* util.go:105-113 */
        int found = 0;
        for (uint32_t j = 0; j < n_new; j++)
            if (new_ids[j] == existing[i]) { found = 1; break; }
        if (!found) to_unsub[nu++] = existing[i];
    }
    qsort(to_unsub, nu, sizeof(uint32_t), cmp_u32);
    *n_unsub = nu;
    for (uint32_t j = 0; j < n_new; j++) {
        int found = 0;
        for (uint32_t i = 0; i < n_existing; i++)
            if (new_ids[j] == existing[i]) { found = 1; break; }
        is_new[j] = (uint8_t)!found;
    }
}

/* ------------------------------------------------------------------ */
/* regions, adjacency, server cells, border subscriptions               */
/* ------------------------------------------------------------------ */

static void server_grid_dims(const orc_grid *g, uint32_t *sgc, uint32_t *sgr) {
    /* spatial.go:321-330 / :399-408 */
    *sgc = g->cols / g->server_cols;
    if (g->cols % g->server_cols > 0) (*sgc)++;
    *sgr = g->rows / g->server_rows;
    if (g->rows % g->server_rows > 0) (*sgr)++;
}

void orc_regions(const orc_grid *g, double *min_x, double *min_z,
                 double *max_x, double *max_z, uint32_t *channel_id,
                 uint32_t *server_index) {
    uint32_t sgc, sgr;
    server_grid_dims(g, &sgc, &sgr);
    for (uint32_t y = 0; y < g->rows; y++) {
        for (uint32_t x = 0; x < g->cols; x++) {
            uint32_t index = x + y * g->cols;
            uint32_t sx = x / sgc, sy = y / sgr;
            min_x[index] = g->off_x + g->grid_w * (double)x;
            min_z[index] = g->off_z + g->grid_h * (double)y;
            max_x[index] = g->off_x + g->grid_w * (double)(x + 1);
            max_z[index] = g->off_z + g->grid_h * (double)(y + 1);
            channel_id[index] = g->id_start + index;
            server_index[index] = sx + sy * g->server_cols;
        }
    }
}

uint32_t orc_adjacent(const orc_grid *g, uint32_t channel_id, uint32_t *out) {
    /* spatial.go:358-381 */
    uint32_t index = channel_id - g->id_start;
    int32_t gx = (int32_t)(index % g->cols);
    int32_t gy = (int32_t)(index / g->cols);
    uint32_t n = 0;
    for (int32_t y = gy - 1; y <= gy + 1; y++) {
        if (y < 0 || y > (int32_t)(g->rows - 1)) continue;
        for (int32_t x = gx - 1; x <= gx + 1; x++) {
            if (x < 0 || x > (int32_t)(g->cols - 1)) continue;
            if (x == gx && y == gy) continue;
            out[n++] = (uint32_t)x + (uint32_t)y * g->cols + g->id_start;
        }
    }
    return n;
}

int orc_server_channels(const orc_grid *g, uint32_t server_index,
                        uint32_t *out, uint32_t cap) {
    uint32_t sgc, sgr;
    server_grid_dims(g, &sgc, &sgr);
    if (sgc * sgr > cap) return -1;
    uint32_t sx = server_index % g->server_cols;
    uint32_t sy = server_index / g->server_cols;
    for (uint32_t y = 0; y < sgr; y++) {
        for (uint32_t x = 0; x < sgc; x++) {
            /* spatial.go:414-422 */
            double px = (double)(sx * sgc + x) * g->grid_w;
            double pz = (double)(sy * sgr + y) * g->grid_h;
            uint32_t ch = orc_channel_id_no_offset(g, px, pz);
            if (!ch) return -1;
            out[x + y * sgc] = ch;
        }
    }
    return (int)(sgc * sgr);
}

int orc_border_channels(const orc_grid *g, uint32_t server_index,
                        uint32_t *out, uint32_t cap) {
    if (g->border == 0) return 0; /* spatial.go:482-484 */
    uint32_t sgc, sgr;
    server_grid_dims(g, &sgc, &sgr);
    uint32_t sx = server_index % g->server_cols;
    uint32_t sy = server_index / g->server_cols;
    uint32_t n = 0;
#define ADD(px, pz)                                               \
    do {                                                          \
        uint32_t _c = orc_channel_id_no_offset(g, (px), (pz));    \
        if (!_c) return -1;                                       \
        if (n >= cap) return -1;                                  \
        out[n++] = _c;                                            \
    } while (0)
    /* the server's first cell must resolve too (:496-499) */
    if (!orc_channel_id_no_offset(g, (double)(sx * sgc) * g->grid_w,
                                  (double)(sy * sgr) * g->grid_h))
        return -1;
    if (sx > 0) /* "Right border" :502-522 (cells at lower X) */
        for (uint32_t y = 0; y < sgr; y++)
            for (uint32_t x = 1; x <= g->border; x++)
                ADD((double)(sx * sgc - x) * g->grid_w, (double)(sy * sgr + y) * g->grid_h);
    if (sx < g->server_cols - 1) /* "Left border" :525-545 */
        for (uint32_t y = 0; y < sgr; y++)
            for (uint32_t x = 0; x < g->border; x++)
                ADD((double)((sx + 1) * sgc + x) * g->grid_w, (double)(sy * sgr + y) * g->grid_h);
    if (sy > 0) /* "Top border" :548-567 */
        for (uint32_t y = 1; y <= g->border; y++)
            for (uint32_t x = 0; x < sgc; x++)
                ADD((double)(sx * sgc + x) * g->grid_w, (double)(sy * sgr - y) * g->grid_h);
    if (sy < g->server_rows - 1) /* "Bottom border" :570-588 */
        for (uint32_t y = 0; y < g->border; y++)
            for (uint32_t x = 0; x < sgc; x++)
                ADD((double)(sx * sgc + x) * g->grid_w, (double)((sy + 1) * sgr + y) * g->grid_h);
#undef ADD
    return (int)n;
}

int orc_notify_decision(const orc_grid *g, double old_x, double old_z,
                        double new_x, double new_z, uint32_t *src,
                        uint32_t *dst) {
    /* spatial.go:613-626 */
    *dst = 0;
    *src = orc_channel_id(g, old_x, old_z);
    if (!*src) return 0;
    *dst = orc_channel_id(g, new_x, new_z);
    if (!*dst) return 0;
    return *dst != *src;
}

/* ------------------------------------------------------------------ */
/* fan-out: container/list + tickData                                   */
/* ------------------------------------------------------------------ */

typedef struct foc_node { /* fanOutConnection, data.go:39-44 */
    struct foc_node *prev, *next;
    uint32_t conn_id;
    int closing;
    int had_first;
    orc_time last;
    uint64_t last_msg_index;
} foc_node;

typedef struct { /* ChannelSubscription.options, subscription.go:13-31 */
    uint32_t conn_id;
    int access;
    uint32_t interval_ms;
    int32_t delay_ms;
    int skip_self;
    int skip_first;
    orc_time sub_time;
    foc_node *elem;
} sub_t;

typedef struct { /* updateMsgBufferElement, data.go:46-51 */
    orc_time arrival;
    uint32_t sender;
    uint32_t tag;
    uint64_t index;
} upd_t;

struct orc_channel {
    foc_node *front, *back; /* fanOutQueue */
    uint32_t qlen;
    sub_t *subs; /* subscribedConnections */
    uint32_t nsubs, capsubs;
    upd_t *buf; /* updateMsgBuffer (vector used as FIFO) */
    uint32_t buf_head, buf_len, buf_cap;
    uint32_t max_interval_ms;
    uint64_t msg_index;
    int has_msg; /* ch.data.msg != nil */
};

#define MAX_UPDATE_MSG_BUFFER_SIZE 512 /* data.go:53-55 */

orc_channel *orc_channel_new(void) {
    return (orc_channel *)calloc(1, sizeof(orc_channel));
}

void orc_channel_free(orc_channel *ch) {
    if (!ch) return;
    foc_node *n = ch->front;
    while (n) {
        foc_node *nx = n->next;
        free(n);
        n = nx;
    }
    free(ch->subs);
    free(ch->buf);
    free(ch);
}

static sub_t *find_sub(orc_channel *ch, uint32_t conn_id) {
    for (uint32_t i = 0; i < ch->nsubs; i++)
        if (ch->subs[i].conn_id == conn_id) return &ch->subs[i];
    return NULL;
}

static void list_unlink(orc_channel *ch, foc_node *e) {
    if (e->prev) e->prev->next = e->next; else ch->front = e->next;
    if (e->next) e->next->prev = e->prev; else ch->back = e->prev;
    e->prev = e->next = NULL;
    ch->qlen--;
}

static void list_insert_after(orc_channel *ch, foc_node *e, foc_node *mark) {
    e->prev = mark;
    e->next = mark->next;
    if (mark->next) mark->next->prev = e; else ch->back = e;
    mark->next = e;
    ch->qlen++;
}

static void list_push_front(orc_channel *ch, foc_node *e) {
    e->prev = NULL;
    e->next = ch->front;
    if (ch->front) ch->front->prev = e; else ch->back = e;
    ch->front = e;
    ch->qlen++;
}

static orc_time add_ms_u(orc_time t, uint32_t ms) { /* channel.go:31-33 */
    return t + (orc_time)ms * 1000000;
}
static orc_time offset_ms(orc_time t, int32_t ms) { /* channel.go:35-37 */
    return t + (orc_time)ms * 1000000;
}

int orc_subscribe(orc_channel *ch, uint32_t conn_id, orc_time now,
                  uint32_t interval_ms, int32_t delay_ms, int skip_self,
                  int skip_first, int access) {
    sub_t *cs = find_sub(ch, conn_id);
    if (cs) {
        /* subscription.go:44-57: proto.Merge(&cs.options, options).
         * The caller passes the merged-in values; "absent" is encoded as
         * UINT32_MAX / INT32_MIN / -1 and keeps the old value. */
        const int old_access = cs->access;
        if (interval_ms != UINT32_MAX) cs->interval_ms = interval_ms;
        if (delay_ms != INT32_MIN) cs->delay_ms = delay_ms;
        if (skip_self >= 0) cs->skip_self = skip_self;
        if (skip_first >= 0) cs->skip_first = skip_first;
        if (access >= 0) cs->access = access;
        /* subscription.go:47-57: "shouldSend" of a repeated subscription = dataAccessChanged
         * (subscription_test.go:35-44) */
        return cs->access != old_access;
    }
    if (ch->nsubs == ch->capsubs) {
        /* keep elem pointers valid: nodes are heap objects, subs may move */
        ch->capsubs = ch->capsubs ? ch->capsubs * 2 : 16;
        ch->subs = (sub_t *)realloc(ch->subs, ch->capsubs * sizeof(sub_t));
    }
    cs = &ch->subs[ch->nsubs++];
    cs->conn_id = conn_id;
    cs->access = access < 0 ? ORC_ACCESS_READ : access;
    cs->interval_ms = interval_ms;
    cs->delay_ms = delay_ms;
    cs->skip_self = skip_self < 0 ? 1 : skip_self;
    cs->skip_first = skip_first < 0 ? 0 : skip_first;
    cs->sub_time = now;
    foc_node *e = (foc_node *)calloc(1, sizeof(foc_node));
    e->conn_id = conn_id;
    e->had_first = cs->skip_first;              /* subscription.go:72 */
    e->last = offset_ms(now, cs->delay_ms);     /* subscription.go:74 */
    list_push_front(ch, e);                     /* subscription.go:70 */
    cs->elem = e;
    if (ch->max_interval_ms < cs->interval_ms)  /* subscription.go:84-86 */
        ch->max_interval_ms = cs->interval_ms;
    return 1;
}

/* the subscription's merged options (test accessor); -1 if not subscribed */
int orc_sub_options(const orc_channel *ch, uint32_t conn_id, uint32_t *interval_ms, int32_t *delay_ms,
                    int *skip_self, int *skip_first, int *access) {
    for (uint32_t i = 0; i < ch->nsubs; i++) {
        const sub_t *cs = &ch->subs[i];
        if (cs->conn_id != conn_id) continue;
        *interval_ms = cs->interval_ms; *delay_ms = cs->delay_ms;
        *skip_self = cs->skip_self; *skip_first = cs->skip_first; *access = cs->access;
        return 0;
    }
    return -1;
}

int orc_unsubscribe(orc_channel *ch, uint32_t conn_id) {
    sub_t *cs = find_sub(ch, conn_id);
    if (!cs) return -1;
    /* the element may already have been dropped by tickData (closing) */
    if (cs->elem) {
        list_unlink(ch, cs->elem);
        free(cs->elem);
    }
    *cs = ch->subs[--ch->nsubs];
    return 0;
}

void orc_set_closing(orc_channel *ch, uint32_t conn_id) {
    sub_t *cs = find_sub(ch, conn_id);
    if (cs && cs->elem) cs->elem->closing = 1;
}

void orc_init_data(orc_channel *ch) { ch->has_msg = 1; }

void orc_on_update(orc_channel *ch, orc_time t, uint32_t sender_conn,
                   uint32_t update_tag) {
    /* data.go:149-173 */
    ch->has_msg = 1;
    ch->msg_index++;
    if (ch->buf_head + ch->buf_len == ch->buf_cap) {
        if (ch->buf_head > 0) {
            memmove(ch->buf, ch->buf + ch->buf_head, ch->buf_len * sizeof(upd_t));
            ch->buf_head = 0;
        } else {
            ch->buf_cap = ch->buf_cap ? ch->buf_cap * 2 : 64;
            ch->buf = (upd_t *)realloc(ch->buf, ch->buf_cap * sizeof(upd_t));
        }
    }
    upd_t *u = &ch->buf[ch->buf_head + ch->buf_len++];
    u->arrival = t;
    u->sender = sender_conn;
    u->tag = update_tag;
    u->index = ch->msg_index;
    if (ch->buf_len > MAX_UPDATE_MSG_BUFFER_SIZE) {
        upd_t *oldest = &ch->buf[ch->buf_head];
        if (add_ms_u(oldest->arrival, ch->max_interval_ms) < t) {
            ch->buf_head++;
            ch->buf_len--;
        }
    }
}

int orc_tick_data(orc_channel *ch, orc_time t, orc_send *out, uint32_t cap) {
    if (!ch->has_msg) return 0; /* data.go:176-178 */
    uint32_t nout = 0;
    uint64_t guard = 0;
    foc_node *focp = ch->front;
    while (focp) {
        if (++guard > 50000000ull) return ORC_E_HANG;
        foc_node *foc = focp;
        if (foc->closing) { /* :183-188 */
            foc_node *tmp = focp->next;
            sub_t *owner = find_sub(ch, foc->conn_id);
            if (owner) owner->elem = NULL;
            list_unlink(ch, focp);
            free(focp);
            focp = tmp;
            continue;
        }
        sub_t *cs = find_sub(ch, foc->conn_id);
        if (!cs || cs->access == ORC_ACCESS_NO) { /* :192-195 */
            focp = focp->next;
            continue;
        }
        orc_time next = add_ms_u(foc->last, cs->interval_ms);
        if (t >= next) {
            orc_time latest = next;
            orc_time last_update_time = 0;
            int has_ever_merged = 0;
            orc_send s;
            memset(&s, 0, sizeof s);
            s.conn_id = foc->conn_id;
            if (!foc->had_first) { /* :217-223 */
                if (nout >= cap) return ORC_E_CAP;
                s.full = 1;
                s.win_lo = foc->last;
                s.win_hi = next;
                out[nout++] = s;
                foc->had_first = 1;
                foc->last_msg_index = ch->msg_index;
                latest = t;
            } else if (ch->buf_len > 0) { /* :224-269 */
                if (foc->last >= last_update_time) last_update_time = foc->last;
                s.win_lo = last_update_time;
                s.win_hi = next;
                for (uint32_t bi = 0; bi < ch->buf_len; bi++) {
                    upd_t *be = &ch->buf[ch->buf_head + bi];
                    if (be->sender == foc->conn_id && cs->skip_self) continue;
                    if (be->arrival >= last_update_time && be->arrival <= next) {
                        if (!has_ever_merged) s.first_tag = be->tag;
                        s.last_tag = be->tag;
                        s.n_merged++;
                        has_ever_merged = 1;
                        last_update_time = be->arrival;
                        foc->last_msg_index = be->index;
                    }
                }
                if (has_ever_merged) {
                    if (nout >= cap) return ORC_E_CAP;
                    out[nout++] = s;
                }
            }
            foc->last = latest; /* :271 */

            foc_node *temp = focp->prev; /* :273-286 */
            for (foc_node *be = ch->back; be != NULL; be = be->prev) {
                if (be->last <= foc->last) {
                    if (be != focp) { /* list.MoveAfter: no-op when e == mark */
                        list_unlink(ch, focp);
                        list_insert_after(ch, focp, be);
                    }
                    if (temp != NULL) focp = temp->next;
                    else focp = ch->front;
                    break;
                }
            }
        } else {
            focp = focp->next;
        }
    }
    return (int)nout;
}

uint32_t orc_channel_queue(const orc_channel *ch, uint32_t *conn_ids,
                           orc_time *last, uint8_t *had_first, uint32_t cap) {
    uint32_t n = 0;
    for (foc_node *e = ch->front; e && n < cap; e = e->next, n++) {
        if (conn_ids) conn_ids[n] = e->conn_id;
        if (last) last[n] = e->last;
        if (had_first) had_first[n] = (uint8_t)e->had_first;
    }
    return n;
}

uint32_t orc_channel_buffer_len(const orc_channel *ch) { return ch->buf_len; }
uint32_t orc_channel_max_interval(const orc_channel *ch) { return ch->max_interval_ms; }

/* oracle-only helpers used by chd_world_oracle.c's literal mode */
void orc__force_state(orc_channel *ch, uint32_t conn_id, orc_time last, int had_first) {
    sub_t *cs = find_sub(ch, conn_id);
    if (cs && cs->elem) {
        cs->elem->last = last;
        cs->elem->had_first = had_first;
    }
}

int orc__get_state(const orc_channel *ch, uint32_t conn_id, orc_time *last, int *had_first) {
    sub_t *cs = find_sub((orc_channel *)ch, conn_id);
    if (!cs || !cs->elem) return -1;
    *last = cs->elem->last;
    *had_first = cs->elem->had_first;
    return 0;
}
