/*
 * chd_oracle.h — CPU restatement of channeld's SpatialChannel hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 * The product (libchd_spatial.so) never links, loads or calls it.
 *
 * Every function restates, sequentially and literally, one piece of the
 * reference Go code (paths relative to the channeld tree):
 *   pkg/channeld/spatial.go, pkg/common/common.go,
 *   pkg/channeld/message_spatial.go, pkg/channeld/subscription.go,
 *   pkg/channeld/data.go, pkg/channeld/channel.go, pkg/channeld/util.go.
 * The reference is Go; no Go toolchain exists in the build image and the
 * module's dependencies are not vendored, so the reference cannot be
 * compiled here.  Parity is pinned by the reference's own test vectors
 * (spatial_test.go, data_test.go, subscription_test.go, connection_test.go)
 * transcribed in tests/test_oracle_golden.py / tests/test_wire_oracle.py, and
 * for the wire format by the packets the Go server recorded in
 * examples/replay (tests/golden/cpr_packs.npz).
 * PARITY UNPINNED (no reference test or fixture exists; literal restatement
 * only): Notify's handover outputs, the interest diff, AOI dist values, Spots
 * AOIs, the cone boundary beyond result-set sizes, handover / broadcast
 * recipient lists, the per-record merged-update masks.
 *
 * Third-party arithmetic restated here: Go's math.Cos (Go standard library,
 * src/math/sin.go — the Cephes port; go.mod pins `go 1.25`), math.Min/Max
 * special-case rules, int(float64)/uint(float64) conversion on amd64.
 * No FMA anywhere (build with -ffp-contract=off), as amd64 Go emits none.
 */
#ifndef CHD_ORACLE_H
#define CHD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* StaticGrid2DSpatialController fields, spatial.go:89-124, plus
 * GlobalSettings.SpatialChannelIdStart (settings.go:94). */
typedef struct {
    double grid_w, grid_h;     /* GridWidth, GridHeight */
    double off_x, off_z;       /* WorldOffsetX, WorldOffsetZ */
    uint32_t cols, rows;       /* GridCols, GridRows */
    uint32_t server_cols, server_rows;
    uint32_t border;           /* ServerInterestBorderSize */
    uint32_t id_start;         /* SpatialChannelIdStart, default 0x10000 */
} orc_grid;

/* error codes (negative) */
#define ORC_OK 0
#define ORC_E_CONFIG (-1)
#define ORC_E_NILQUERY (-2)
#define ORC_E_EXTENT (-3)   /* "invalid box extent" / "invalid radius" */
#define ORC_E_CENTER (-4)   /* centre of Box/Sphere/Cone out of world */
#define ORC_E_CAP (-5)      /* caller buffer too small (oracle-only) */
#define ORC_E_HANG (-6)     /* reference would loop forever (v+step==v) */
#define ORC_E_RANGE (-7)

/* LoadConfig validation, spatial.go:141-159.  Returns ORC_OK or ORC_E_CONFIG.
 * *which: 1 grid size, 2 cols/rows, 3 server cols/rows, 4 border<=0. */
int orc_validate_config(const orc_grid *g, int *which);

/* GridSize(), spatial.go:134-139. */
double orc_grid_size(const orc_grid *g);

/* GetChannelIdWithOffset, spatial.go:169-180.  Returns 0 on error
 * (the reference returns (0, err)). */
uint32_t orc_channel_id_with_offset(const orc_grid *g, double x, double z,
                                    double off_x, double off_z);
/* GetChannelId :161 and GetChannelIdNoOffset :165. */
uint32_t orc_channel_id(const orc_grid *g, double x, double z);
uint32_t orc_channel_id_no_offset(const orc_grid *g, double x, double z);
void orc_channel_ids(const orc_grid *g, const double *x, const double *z,
                     uint32_t n, uint32_t *out);

/* Go math helpers restated. */
double orc_go_cos(double x);           /* math.Cos, Go src/math/sin.go */
double orc_go_min(double x, double y); /* math.Min */
double orc_go_max(double x, double y); /* math.Max */

/* SpatialInterestQuery, channeld.proto:386-440 — flattened. */
#define ORC_SHAPE_SPOTS 1u
#define ORC_SHAPE_BOX 2u
#define ORC_SHAPE_SPHERE 4u
#define ORC_SHAPE_CONE 8u
typedef struct {
    uint32_t shapes;          /* bit mask of ORC_SHAPE_* (non-nil members) */
    uint32_t n_spots;         /* len(SpotsAOI.Spots) */
    uint32_t n_spot_dists;    /* len(SpotsAOI.Dists) */
    uint32_t _pad;
    const double *spot_x;     /* Spots[i].X */
    const double *spot_z;     /* Spots[i].Z */
    const uint32_t *spot_dist;/* Dists[i] */
    double box_cx, box_cz, box_ex, box_ez;        /* BoxAOI centre/extent */
    double sph_cx, sph_cz, sph_r;                 /* SphereAOI */
    double cone_cx, cone_cz, cone_dx, cone_dz;    /* ConeAOI centre/direction */
    double cone_r, cone_angle;
    double cone_cos;          /* used instead of go_cos(angle) iff use_cone_cos */
    uint32_t use_cone_cos;
    uint32_t _pad2;
} orc_query;

/* QueryChannelIds, spatial.go:182-317.  Output = the Go map as (cell id,
 * dist) pairs sorted ascending by channel id.  Returns ORC_OK or an error
 * (then *n_out = 0: the reference returns (nil, err)). */
int orc_query_channel_ids(const orc_grid *g, const orc_query *q,
                          uint32_t *ids, uint32_t *dists, uint32_t cap,
                          uint32_t *n_out);

/* getSpatialDampingSettings + the nil branch, message_spatial.go:16-38,66-79.
 * default_interval_ms = SPATIAL DefaultFanOutIntervalMs. */
uint32_t orc_damping_interval(uint32_t dist, uint32_t default_interval_ms);

/* handleUpdateSpatialInterest diff, message_spatial.go:59-128 + util.go:105.
 * existing/new are channel-id lists (sets).  to_unsub = existing - new.
 * Every id of `new` is (re)subscribed, so to_sub == new; is_new[i] tells
 * whether new[i] was absent from `existing`. Outputs sorted ascending. */
void orc_interest_diff(const uint32_t *existing, uint32_t n_existing,
                       const uint32_t *new_ids, uint32_t n_new,
                       uint32_t *to_unsub, uint32_t *n_unsub,
                       uint8_t *is_new);

/* GetRegions, spatial.go:319-356 — SoA outputs of cols*rows entries. */
void orc_regions(const orc_grid *g, double *min_x, double *min_z,
                 double *max_x, double *max_z, uint32_t *channel_id,
                 uint32_t *server_index);

/* GetAdjacentChannels, spatial.go:358-381.  out has room for 8. */
uint32_t orc_adjacent(const orc_grid *g, uint32_t channel_id, uint32_t *out);

/* CreateChannels cell ownership, spatial.go:399-424: channel ids created for
 * server `server_index`.  Returns count, or -1 if a cell falls outside the
 * grid (GetChannelIdNoOffset error :418-421). */
int orc_server_channels(const orc_grid *g, uint32_t server_index,
                        uint32_t *out, uint32_t cap);

/* subToAdjacentChannels, spatial.go:481-590: the border channels server
 * `server_index` subscribes to, in the reference's call order (duplicates
 * preserved).  Returns count or -1 on GetChannelIdNoOffset error. */
int orc_border_channels(const orc_grid *g, uint32_t server_index,
                        uint32_t *out, uint32_t cap);

/* Notify decision, spatial.go:612-626.  Returns 1 if src!=dst and both
 * valid (handover candidate), else 0.  src and dst receive the ids (0 = error). */
int orc_notify_decision(const orc_grid *g, double old_x, double old_z,
                        double new_x, double new_z, uint32_t *src,
                        uint32_t *dst);

/* ---------------- fan-out: data.go / subscription.go ---------------- */

typedef int64_t orc_time; /* ChannelTime, ns, channel.go:28-37 */

#define ORC_ACCESS_NO 0
#define ORC_ACCESS_READ 1
#define ORC_ACCESS_WRITE 2

typedef struct orc_channel orc_channel;

orc_channel *orc_channel_new(void);
void orc_channel_free(orc_channel *ch);

/* SubscribeToChannel, subscription.go:34-102 (new-subscription branch and
 * the already-subscribed merge branch).  now = ch.GetTime().  Returns the
 * reference's second result ("shouldSend"): 1 for a new subscription; for an
 * existing one 1 iff the merged options changed DataAccess (:47-57). */
int orc_subscribe(orc_channel *ch, uint32_t conn_id, orc_time now,
                  uint32_t interval_ms, int32_t delay_ms, int skip_self,
                  int skip_first, int access);
/* UnsubscribeFromChannel, subscription.go:104-125. Returns 0, -1 if absent. */
int orc_sub_options(const orc_channel *ch, uint32_t conn_id, uint32_t *interval_ms, int32_t *delay_ms,
                    int *skip_self, int *skip_first, int *access);
int orc_unsubscribe(orc_channel *ch, uint32_t conn_id);
/* conn.IsClosing() becomes true: tickData drops it (data.go:183-188). */
void orc_set_closing(orc_channel *ch, uint32_t conn_id);
/* InitData: ch.data.msg != nil from now on. */
void orc_init_data(orc_channel *ch);
/* ChannelData.OnUpdate, data.go:149-173.  update_tag identifies the update
 * (stands in for the protobuf payload). */
void orc_on_update(orc_channel *ch, orc_time t, uint32_t sender_conn,
                   uint32_t update_tag);

/* one fanOutDataUpdate call (data.go:293-318) */
typedef struct {
    uint32_t conn_id;
    uint32_t full;       /* 1 = whole channel data (first fan-out) */
    uint32_t n_merged;   /* number of buffered updates accumulated */
    uint32_t first_tag;  /* tag of the first / last update merged */
    uint32_t last_tag;
    orc_time win_lo;     /* window [lastFanOutTime clamp, nextFanOutTime] */
    orc_time win_hi;
} orc_send;

/* tickData, data.go:175-291, literal (linked list walk incl. the
 * move-to-back re-ordering).  Appends to out[0..cap).  Returns number of
 * sends, or ORC_E_CAP / ORC_E_HANG (interval 0 makes the reference spin). */
int orc_tick_data(orc_channel *ch, orc_time t, orc_send *out, uint32_t cap);

/* introspection for tests */
uint32_t orc_channel_queue(const orc_channel *ch, uint32_t *conn_ids,
                           orc_time *last, uint8_t *had_first, uint32_t cap);
uint32_t orc_channel_buffer_len(const orc_channel *ch);
uint32_t orc_channel_max_interval(const orc_channel *ch);

#ifdef __cplusplus
}
#endif
#endif
