/*
 * chd_spatial.h — C-ABI of libchd_spatial.so, the MI355X-native (HIP/gfx950)
 * replacement for channeld's SpatialChannel hot path.
 *
 * The reference (channeldorg/channeld) is pure Go and has no FFI layer; its
 * boundary for this path is the Go interface `SpatialController`
 * (pkg/channeld/spatial.go:17-35) plus the per-channel fan-out tick
 * (pkg/channeld/data.go:175-318).  Each entry point below names the reference
 * symbol it replaces; INTEGRATION.md shows the cgo shim (a second Go type
 * implementing SpatialController) a channeld maintainer would add on top.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary, no torch / HIP types;
 *   - every function returns 0 (CHD_OK) or a negative chd_status;
 *     chd_last_error(ctx) gives a message for the calling thread's last failure;
 *   - all pointers are HOST pointers unless the name starts with d_ (device);
 *     inputs are borrowed for the duration of the call only (cgo pointer rules);
 *     outputs are caller-allocated;
 *   - channel ids are channeld ChannelIds (uint32): 0 = invalid / out of world
 *     (GLOBAL is never a spatial id), spatial ids start at
 *     spatial_channel_id_start (settings.go:94), entity ids are the engine's;
 *   - every entry point is thread-safe: it binds the HIP device of the ctx to
 *     the calling OS thread (goroutines migrate between threads) and serialises
 *     on the ctx's stream with one mutex;
 *   - ALL compute runs in HIP kernels on the ctx's device.  There is no CPU
 *     fallback: without a usable gfx950 device chd_create fails with
 *     CHD_E_NO_DEVICE.
 *   - arithmetic is IEEE float64 with no FMA contraction, matching amd64 Go.
 */
#ifndef CHD_SPATIAL_H
#define CHD_SPATIAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHD_ABI_VERSION 9

typedef struct chd_ctx chd_ctx;

typedef enum {
    CHD_OK = 0,
    CHD_E_CONFIG = -1,      /* LoadConfig validation failed (spatial.go:146-157) */
    CHD_E_INVAL = -2,       /* NULL / out-of-range argument (Go would nil-deref) */
    CHD_E_EXTENT = -3,      /* per-query: invalid box extent / radius (spatial.go:208-215 ...) */
    CHD_E_CENTER = -4,      /* per-query: AOI centre out of the world (spatial.go:228-231 ...) */
    CHD_E_CAPACITY = -5,    /* output / interest-set capacity exceeded */
    CHD_E_HANG = -6,        /* the reference would loop forever on this input */
    CHD_E_TOO_LARGE = -8,   /* per-query, interest updates of chd_tick only: sample lattice (> 256 lines per axis) or cell
                               window (> 4096 cells) beyond the in-kernel limits — such an interest set would exceed
                               max_interest_cells anyway.  chd_query_channel_ids has no such limit (beyond 65 535 lattice
                               lines per axis): larger queries take whole-GPU passes over global memory */
    CHD_E_NO_DEVICE = -9,   /* no usable HIP device / kernel image */
    CHD_E_HIP = -10,        /* HIP runtime error (message in chd_last_error) */
    CHD_E_STATE = -11       /* call sequence error (e.g. tick before world_create) */
} chd_status;

/* The 9 fields of StaticGrid2DSpatialController (spatial.go:89-124) + the id
 * ranges of GlobalSettings (settings.go:94-95) + the SPATIAL channel settings
 * used on the path (settings.go:64-74) + the damping table
 * (message_spatial.go:16-29). */
#define CHD_MAX_DAMPING 8
typedef struct {
    double grid_width, grid_height;
    double world_offset_x, world_offset_z;
    uint32_t grid_cols, grid_rows;
    uint32_t server_cols, server_rows;
    uint32_t server_interest_border_size;
    uint32_t spatial_channel_id_start;   /* 0 => 0x10000 */
    uint32_t entity_channel_id_start;    /* 0 => 0x80000 */
    uint32_t default_fanout_interval_ms; /* SPATIAL DefaultFanOutIntervalMs; 0 => 20 */
    int32_t default_fanout_delay_ms;     /* SPATIAL DefaultFanOutDelayMs */
    uint32_t n_damping;                  /* 0 => the reference's table {0:20,1:50,2:100} */
    uint32_t damping_max_dist[CHD_MAX_DAMPING];
    uint32_t damping_interval_ms[CHD_MAX_DAMPING];
    /* LoadConfig rejects ServerInterestBorderSize == 0 (spatial.go:155) but
     * InitSpatialController drops that error (spatial.go:68) and the shipped
     * configs use 0.  strict_load_config != 0 reproduces the rejection. */
    uint32_t strict_load_config;
} chd_grid_cfg;

/* replaces: InitSpatialController + LoadConfig (spatial.go:40-74,141-159) */
int chd_create(const chd_grid_cfg *cfg, int device, chd_ctx **out);
void chd_destroy(chd_ctx *ctx);
const char *chd_last_error(const chd_ctx *ctx);
int chd_abi_version(void);

/* replaces: GetChannelId (spatial.go:161-180), batched.  out_ids[i] == 0 means
 * "(0, err)": the point is outside the world. */
int chd_get_channel_ids(chd_ctx *ctx, const double *x, const double *z,
                        uint32_t n, uint32_t *out_ids);

/* replaces: the decision part of Notify (spatial.go:612-626), batched and
 * stateless: src/dst ids (0 = error) and handover[i] = 1 iff both valid and
 * different. */
int chd_notify_decide(chd_ctx *ctx, const double *old_x, const double *old_z,
                      const double *new_x, const double *new_z, uint32_t n,
                      uint32_t *src_ids, uint32_t *dst_ids, uint8_t *handover);

/* channeldpb.SpatialInterestQuery (channeld.proto:386-440), flattened.
 * A shape is present iff its bit is set (a nil sub-message in Go).  Spots live
 * in side arrays shared by the batch: spots [spot_off, spot_off+n_spots),
 * of which the first n_spot_dists have an explicit dist.
 * cone_cos = math.Cos(ConeAOI.Angle) evaluated by the caller (Go evaluates
 * its own pure-Go cos; the device never evaluates cos — SURVEY §8c). */
#define CHD_SHAPE_SPOTS 1u
#define CHD_SHAPE_BOX 2u
#define CHD_SHAPE_SPHERE 4u
#define CHD_SHAPE_CONE 8u
typedef struct {
    uint32_t shapes;
    uint32_t spot_off, n_spots, n_spot_dists;
    double box_cx, box_cz, box_ex, box_ez;
    double sph_cx, sph_cz, sph_r;
    double cone_cx, cone_cz, cone_dx, cone_dz, cone_r, cone_cos;
    double _reserved;
} chd_aoi_query; /* 128 bytes */

/* replaces: QueryChannelIds (spatial.go:182-317), batched.  CSR output:
 * query i owns [offsets[i], offsets[i+1]) of ids/dists, sorted by channel id
 * (the Go map is unordered).  status[i] is CHD_OK or the error the reference
 * returns for that query (then its range is empty).  cap = capacity of
 * ids/dists; CHD_E_CAPACITY if the total does not fit.  Also returns the
 * damped fan-out interval per entry (message_spatial.go:31-38,66-79) when
 * intervals_ms != NULL. */
int chd_query_channel_ids(chd_ctx *ctx, const chd_aoi_query *queries, uint32_t nq,
                          const double *spot_x, const double *spot_z,
                          const uint32_t *spot_dist, uint32_t n_spots_total,
                          uint32_t *offsets /* nq+1 */, uint32_t *ids,
                          uint32_t *dists, uint32_t *intervals_ms, uint32_t cap,
                          int32_t *status /* nq */);

/* replaces: GetRegions (spatial.go:319-356).  SoA, grid_cols*grid_rows entries.
 * Y range is the constant [MinY, MaxY] of spatial.go:80-83 and is not returned. */
int chd_get_regions(chd_ctx *ctx, double *min_x, double *min_z, double *max_x,
                    double *max_z, uint32_t *channel_id, uint32_t *server_index);

/* replaces: GetAdjacentChannels (spatial.go:358-381), batched: out[8*i ..] holds
 * counts[i] ids in the reference's row-major order. */
int chd_get_adjacent_channels(chd_ctx *ctx, const uint32_t *channel_ids,
                              uint32_t n, uint32_t *out /* 8*n */,
                              uint32_t *counts /* n */);

/* replaces: the cell-ownership arithmetic of CreateChannels (spatial.go:399-424):
 * the channel ids created for spatial server `server_index`, in creation order.
 * *n_out = count.  CHD_E_CONFIG if a cell falls outside the grid (:418-421). */
int chd_server_channels(chd_ctx *ctx, uint32_t server_index, uint32_t *out,
                        uint32_t cap, uint32_t *n_out);

/* replaces: subToAdjacentChannels (spatial.go:481-590): the border channels
 * server `server_index` is subscribed to, in the reference's call order. */
int chd_border_channels(chd_ctx *ctx, uint32_t server_index, uint32_t *out,
                        uint32_t cap, uint32_t *n_out);

/* ------------------------------------------------------------------ */
/* The batched tick: entity updates -> handover, interest updates ->    */
/* sub/unsub diff, fan-out decisions -> per-connection records.         */
/* ------------------------------------------------------------------ */

typedef struct {
    uint32_t max_entities;       /* entity slots [0, max_entities) */
    uint32_t max_subscribers;    /* subscriber slots */
    uint32_t max_interest_cells; /* per-subscriber interest-set capacity (0 => min(cells,256)) */
    uint64_t max_records;        /* fan-out record capacity per tick (0 => auto) */
    uint32_t max_handovers;      /* handover record capacity per tick (0 => max_entities) */
    uint32_t flags;              /* CHD_WORLD_* */
    uint32_t wire_max_update_len; /* CHD_WORLD_WIRE: largest serialized Any of a channel update (0 => 128) */
    uint32_t wire_max_full_len;   /* CHD_WORLD_WIRE: largest serialized Any of a channel's full state (0 => 1024) */
    /* ChannelData.updateMsgBuffer (data.go:53-55,149-173) per channel, element for element: {arrivalTime, senderConnId} of the
     * last `history_depth` updates, with the reference's own eviction (beyond MaxUpdateMsgBufferSize = 512 elements the oldest
     * goes once it is older than maxFanOutIntervalMs).  0: only the 32-tick bit ring (exact while every update is stamped
     * with its tick's now_ns, a window reaches back at most 32 ticks and a channel has at most two senders inside it; anything
     * else is COUNTED in history_overflow).  >= 32 (1024 covers the reference's 512 with room for the unexpired surplus):
     * exact for any arrival stamps (chd_tick_in.upd_arrival_ns), any number of updates of a channel per tick
     * (upd_round_off), any number of senders and windows as old as the buffer reaches — subscriptions the bit ring cannot
     * answer are served from these buffers by a separate (slower) launch; history_overflow then only counts what the
     * reference's buffer would still hold and this one had to drop.  12 B x history_depth per entity and cell.
     * Region-sharded worlds: with shard_channels (below).
     * maxFanOutIntervalMs of the eviction test is kept per CHANNEL as in the reference (subscription.go:83-86: raised when a
     * subscription is created, not when options are merged into an existing one): per spatial channel; and per entity channel,
     * whose subscribers are — in the tick model — those of the cells that hold it: raised at every update of the entity to the
     * maxima of the cells of its last merged and its new position, before the eviction test (OnUpdate merges first, and the
     * merge's Notify subscribes the dst cell's connections to the entity channel, spatial.go:797-830). */
    uint32_t history_depth;
    /* Region-sharded worlds that keep exact update buffers (history_depth > 0; else 0): the number of entity channels of the WHOLE
     * world, ids entity_channel_id_start .. + shard_channels - 1.  In the reference a channel's update buffer lives in the one
     * gateway process and never moves; a cross-server handover changes who owns the entity, not where its buffer is
     * (spatial.go:683-700).  Here every rank is fed the same whole-world update stream by channel id (chd_shard_ingest:
     * positions, chd_shard_set_update_arrivals / _senders), so every rank keeps every channel's UPDATE LOG — buffer, tick-ring
     * masks, sub-tick offsets, maxFanOutIntervalMs — itself, indexed by channel id: a pure function of that stream and of the
     * cells' maxFanOutIntervalMs, which travel with the emigrant exchange.  An entity that changes ranks, a ghost entry of a
     * neighbour's border cell, a subscriber that regains access a hundred ticks after its entity left: all find the log where
     * they are.  12 B x history_depth x shard_channels per rank (12 GiB at 1 M channels x 1024: HBM is 288 GB). */
    uint32_t shard_channels;
} chd_world_cfg;

/* The fan-out emit kernel has two forms.  Connection-major: one workgroup per connection
 * streams the (L2-resident) cell tables of its due subscriptions — best while the cell
 * tables are small.  Cell-major (grids up to 4096 cells): each cell's entity table is
 * staged once in LDS by a loader wave and streamed to all its due subscribers by
 * load-free streamer waves — best for populous cells (1.6x at 1M entities on 225 cells).
 * Default: cell-major when max_entities / cells >= 1024 — except on a world that keeps exact
 * update buffers where the descriptor path can run (history_depth > 0, >= 4096 subscriber slots
 * or CHD_WORLD_ONE_WAVE_EMIT, no UPDATE_MASKS / WIRE): only the connection-major form keeps the
 * sub-tick arrival offsets, the cell-major form would send every window that cuts through a
 * tick's arrivals to the element walk (70 x slower at 1 M entities).  The flags force one form;
 * chd_tick_stats.schedule says which is in force (CHD_SCHED_CELL_MAJOR, CHD_SCHED_ARRIVAL_OFFSETS). */
#define CHD_WORLD_CONN_MAJOR_EMIT 1u
#define CHD_WORLD_CELL_MAJOR_EMIT 2u
/* also plan, every tick, who receives each handover's ChannelDataHandoverMessage (chd_handover_recipients) */
#define CHD_WORLD_HANDOVER_RECIPIENTS 4u
/* keep what chd_wire_build needs: which channel table entry every fan-out record came from (+4 B per record) */
#define CHD_WORLD_WIRE 8u
/* run the interest updates on a second HIP stream, concurrently with the entity ingest and the cell index build
 * (they touch disjoint state: subscriptions vs entities; on worlds with exact update buffers maxFanOutIntervalMs is kept
 * twice so that the tick's updates are buffered under the value from before the tick's interest updates, as in the serial
 * order), joining before the fan-out plan.  At BASELINE config B: -3 % per tick with the dependencies as HIP events, -9 % with
 * CHD_WORLD_GATED_OVERLAP.  Ignored with CHD_WORLD_HANDOVER_RECIPIENTS. */
#define CHD_WORLD_OVERLAP_INTEREST 16u
/* also say, per fan-out record, WHICH buffered updates the message merges (chd_tick_out.record_masks): the selection
 * of data.go:225-269 — arrival inside the subscription's window and, with SkipSelfUpdateFanOut, sender != connection —
 * as a mask over the tick ring, bit j = the update that arrived with tick (current - j).  The host (or the wire builder)
 * merges exactly those into the accumulated update, so slow subscribers get correct merged deltas.
 * +4 B per record; windows every entity passes are then streamed with their histories instead of as a plain copy. */
#define CHD_WORLD_UPDATE_MASKS 32u

/* connection-major emit: one wave per connection (the pipelined form) also for worlds of fewer than 4096 connections,
 * where four waves per connection is the default.  For tests and for small worlds with few subscriptions per connection. */
#define CHD_WORLD_ONE_WAVE_EMIT 64u

/* Pipeline successive ticks over two HIP streams: the kernel that writes tick t's records (HBM-write bound, ~55 % of a
 * tick of config B) runs on the ctx stream while tick t+1's stages — ingest, index build, interest updates, fan-out plan,
 * the subscriptions' new fan-out state — run beside it on a second stream; what that kernel reads (segment descriptors,
 * record offsets, the cells' entity columns) and the record buffer itself then exist twice, by tick parity (max_records
 * = 0 halves the automatic buffer size accordingly).  Results are those of the serial schedule, tick for tick; the
 * latency of ONE tick does not change, the rate of back-to-back chd_tick_device calls does.
 * Takes effect where the descriptor-driven connection-major emit runs (no cell-major emit, no CHD_WORLD_UPDATE_MASKS,
 * no CHD_WORLD_WIRE, no region sharding) and is ignored elsewhere.  Contract: the device inputs of chd_tick_device
 * must be COMPLETE when the call is made — the stages do not wait for work the caller merely enqueued on the ctx
 * stream.  Everything else (chd_tick, fetch, digest, sync, any other entry point) orders itself after both streams. */
#define CHD_WORLD_PIPELINE_TICKS 128u

/* Serial schedule: run the filtering launch (the subscriptions the plan left to a per-entity decision, and the commit of every
 * subscription's new fan-out state) and the tick's epilogue on a second HIP stream BESIDE the kernel that writes the records,
 * joining before the tick ends — the same pair a pipelined tick runs side by side.  Results are the serial schedule's.  Off by
 * default: at BASELINE config B it does not pay — the record kernel is HBM-write bound and slows down by what the launch beside
 * it would have taken alone (0.273 against 0.264 ms per tick, record kernel 149 against 141 us); it is for worlds whose
 * filtering launch is long and whose record kernel is not the bound.  Takes effect where the descriptor-driven connection-major
 * emit runs and the world keeps no exact update buffers (history_depth); ignored elsewhere, and by pipelined ticks. */
#define CHD_WORLD_OVERLAP_DEFERRED 256u
/* With CHD_WORLD_OVERLAP_INTEREST on the serial schedule: the two dependencies between the tick's stream and the second one —
 * "the interest updates start after the previous tick", "the plan starts after the interest updates" — as device-side flags
 * (one spinning wave each) instead of HIP events: recording an event idles the tick's stream for ~7 us and a cross-queue wait
 * takes ~11 us to resolve, which together cost half of what the overlap saves.  Results unchanged.  As with
 * CHD_WORLD_PIPELINE_TICKS the inputs of chd_tick_device must be COMPLETE when the call is made (the second stream no longer
 * waits for work the caller enqueued on the tick's stream before the call); chd_tick (host pointers) and the first tick after
 * any other call on the context take the event form by themselves.
 * Safe by construction: the join's release is a kernel boundary (a one-wave kernel behind the interest launch raises the flag),
 * so nothing rests on where the dispatcher places workgroups; and the flag is only used where the context's two streams were
 * SEEN to run side by side (probed with a waiter and a raiser at world creation and at every chd_set_stream: with both streams
 * on one hardware queue — GPU_MAX_HW_QUEUES=1 — the raiser would sit behind the waiter; such a world takes HIP events and says
 * so in chd_tick_stats.schedule).  A gate that still is not raised within its spin bound (2^23 polls, ~1 s) ends the wait, reports
 * overflow bit 0x4000 for that tick (its results are not valid) and switches the world to events for good
 * (chd_tick_stats.gate_timeouts). */
#define CHD_WORLD_GATED_OVERLAP 512u
/* A world whose fan-out is consumed in the SEGMENT form only (chd_tick_fetch_segments, chd_tick_segments, chd_tick_segments_begin /
 * _end: what INTEGRATION.md's tick driver does).  On ticks of the descriptor path the plain-copy records — the ones a segment names
 * as "column [off, off + n) once per window" — are then NOT written to HBM at all: the host expands them from the columns anyway.
 * Counts, segments, columns, explicit records (the subscriptions that needed a per-entity decision) and every other output are
 * unchanged; chd_tick's dense record outputs, chd_tick_digest and the wire builder answer CHD_E_STATE on such a world (there are
 * no dense records to pack, digest or assemble).  Not with CHD_WORLD_WIRE / CHD_WORLD_UPDATE_MASKS / the cell-major emit. */
#define CHD_WORLD_SEGMENTS_ONLY 1024u

#define CHD_ENTITY_LOCKED 1u /* member of a non-empty lock group (entity.go:197-224) */

int chd_world_create(chd_ctx *ctx, const chd_world_cfg *cfg);

/* Which form a world of this shape WILL take, without creating it: *schedule = the CHD_SCHED_CELL_MAJOR / CHD_SCHED_ARRIVAL_OFFSETS /
 * CHD_SCHED_PIPELINED bits chd_tick_stats.schedule of such a world reports (the stream-overlap bits depend on the device and are not
 * predicted).  Pure host arithmetic — the one function chd_world_create itself decides with — so that a gateway (and the test-suite)
 * can check that its configuration keeps the streaming path: populous cells (>= 1024 entities per cell) select the cell-major form
 * unless the world keeps exact update buffers (history_depth) and can run the descriptor path, where sub-tick arrival offsets decide
 * the windows that cut through a tick's arrivals.  CHD_E_INVAL for a combination chd_world_create refuses. */
int chd_world_emit_form(uint32_t max_entities, uint32_t max_subscribers, uint32_t n_cells, uint32_t world_flags, uint32_t history_depth,
                        uint32_t *schedule);

/* Entity channel creation + spawn into the cell containing (x,z)
 * (message_spatial.go:231-237, pkg/unreal/message.go:55).  idx = entity slots,
 * chan_id = entity channel ids (NetGUIDs), sender = the connection that will
 * send the entity's updates (channel owner).  flags: CHD_ENTITY_*. */
int chd_world_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *idx,
                    const uint32_t *chan_id, const double *x, const double *z,
                    const uint32_t *flags, const uint32_t *sender);
int chd_world_despawn(chd_ctx *ctx, uint32_t n, const uint32_t *idx);
int chd_world_set_entity_flags(chd_ctx *ctx, uint32_t n, const uint32_t *idx,
                               const uint32_t *flags);

/* Handover groups (entity.go:58-244, FlatEntityGroupController; AddEntityGroupMessage / RemoveEntityGroupMessage stay
 * with the Go host, the engine receives the result): entities with the same non-zero group id cross cells TOGETHER
 * — when one of them hands over (its own old / new position, spatial.go:612-626) every member that is in the src
 * cell's entity map moves to the dst cell's (:703-736 over handoverEntities) — and a locked member (CHD_ENTITY_LOCKED)
 * aborts the handover of the whole group (GetHandoverEntities, entity.go:197-224; counted in n_locked_aborts).  One
 * handover record per notifying entity, as the reference sends one ChannelDataHandoverMessage per Notify.  group 0 =
 * no group (a group of one).  Deviations: a member that is in a THIRD cell's map stays there (the reference would
 * also add it to dst's map, leaving it in two maps); "locked" is the entity's flag, not the notifier's lock-group
 * membership test.  Not available on region-sharded worlds (there: chd_shard_set_handover_lists). */
int chd_world_set_entity_groups(chd_ctx *ctx, uint32_t n, const uint32_t *idx, const uint32_t *group);

/* The exact form: the engine is given what FlatEntityGroupController.GetHandoverEntities (entity.go:197-224) returns for
 * every entity channel, as the host's group controller evaluates it (the reference's per-channel handover / lock group
 * pointers are not equivalence classes: a locked entity does not take over the group it is added to, a removed one keeps an
 * EMPTY group and cannot hand over until it is re-added — entity_test.go:11-105; channeld_amd/groups.py mirrors the
 * controller and produces these arrays).  List k = list_members[list_off[k] .. list_off[k+1]) (entity slots; ids without a
 * live entity are left out by the host); entity idx[i] takes list list_of[i], CHD_NO_HANDOVER_LIST = "AddToGroup was never
 * called: the entity itself".  When entity e crosses cells: an EMPTY list -> no handover (len(handoverEntities) == 0,
 * spatial.go:675-679: a locked member or an emptied group; counted in n_locked_aborts); else one handover record and every
 * list member that is in the src cell's entity map moves to dst's (:703-736) — the notifier itself only if the list names
 * it.  CHD_ENTITY_LOCKED is still honoured (it empties the notifier's own result).  Replaces the WHOLE group state of the
 * world (n_lists == 0 clears it); the later of this call and chd_world_set_entity_groups wins.  Same deviation as above for
 * members in a third cell's map.  Region-sharded worlds: chd_shard_set_handover_lists. */
#define CHD_NO_HANDOVER_LIST 0xFFFFFFFFu
int chd_world_set_handover_lists(chd_ctx *ctx, uint32_t n_lists, const uint32_t *list_off /* n_lists + 1 */,
                                 const uint32_t *list_members, uint32_t n, const uint32_t *idx, const uint32_t *list_of);

/* A client connection with spatial interest (connection.go:106
 * spatialSubscriptions).  Slot -> ConnectionId.  Removing a subscriber drops all
 * its subscriptions (a closing connection, data.go:183-188). */
int chd_subs_add(chd_ctx *ctx, uint32_t n, const uint32_t *slot,
                 const uint32_t *conn_id);
int chd_subs_remove(chd_ctx *ctx, uint32_t n, const uint32_t *slot);

/* channeldpb.ChannelSubscriptionOptions (channeld.proto:216-240) for ONE (connection, spatial channel) pair.  `set` says
 * which fields are present (they are optional in the protobuf: an absent field keeps the stored / default value). */
#define CHD_SUBOPT_ACCESS 1u
#define CHD_SUBOPT_INTERVAL 2u
#define CHD_SUBOPT_DELAY 4u
#define CHD_SUBOPT_SKIP_SELF 8u
#define CHD_SUBOPT_SKIP_FIRST 16u
#define CHD_SUBOPT_FIELD_MASK 32u
#define CHD_ACCESS_NONE 0u  /* ChannelDataAccess_NO_ACCESS: skipped by the fan-out but stays queued (data.go:194-197) */
#define CHD_ACCESS_READ 1u
#define CHD_ACCESS_WRITE 2u
typedef struct {
    uint32_t slot;                    /* connection slot (chd_subs_add) */
    uint32_t channel;                 /* spatial channel id */
    uint32_t set;                     /* CHD_SUBOPT_* */
    uint32_t data_access;             /* CHD_ACCESS_* */
    uint32_t fanout_interval_ms;      /* must not be 0 (the reference's tickData would spin) */
    int32_t fanout_delay_ms;          /* may be negative (channeld.proto:229-233) */
    uint32_t skip_self_update_fanout; /* 0 / 1 */
    uint32_t skip_first_fanout;       /* 0 / 1 */
    /* ChannelSubscriptionOptions.DataFieldMasks (channeld.proto:216-240; fmutils.Filter in fanOutDataUpdate, data.go:294) in
     * the bit form of the world's merge schema (chd_wire_set_merge_schema; the host maps the path strings).  For
     * CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT: bit f (0..5) = "actorState.replicatedMovement.<field f + 1>" (linearVelocity,
     * angularVelocity, location, rotation, bSimulatedPhysicSleep, bRepPhysics) is listed; bit 6 = the masks list only fields
     * outside actorState (every field of the subset is cleared); 0 = no masks (the message goes out whole).  Applied to the
     * UPDATE messages the engine builds; the reference also filters the full state of a first fan-out — in place, on the
     * channel's own data (data.go:219,294) — which stays with the host. */
    uint32_t data_field_mask;
} chd_sub_options;

/* replaces: Connection.SubscribeToChannel(spatial channel, options) (subscription.go:34-102) as reached from an explicit
 * SUB_TO_CHANNEL message (handleSubToChannel) — the spatial servers' own subscriptions (WRITE access, spatial.go:481-590:
 * chd_server_channels / chd_border_channels give the channels) and clients that set DataAccess, SkipSelfUpdateFanOut,
 * SkipFirstFanOut or their own interval / delay.  Records are applied in array order.  Already subscribed: the present
 * fields are merged into the stored options, the fan-out state stays (:44-57) and should_send[i] = dataAccessChanged;
 * else a new subscription with the defaults of :21-31 merged with the options, hadFirstFanOut = SkipFirstFanOut,
 * lastFanOutTime = now_ns + FanOutDelayMs (:59-75) and should_send[i] = 1.  status[i] (optional) = CHD_OK, CHD_E_INVAL (no
 * such connection: the reference returns (nil, false)) or CHD_E_CAPACITY (max_interest_cells).  An interest update
 * (chd_tick) afterwards treats such a subscription like any other: kept with its options and state if the new query
 * still holds the channel (only the interval is overwritten by the damped one, message_spatial.go:66-79), unsubscribed
 * otherwise (Difference over spatialSubscriptions, :82). */
int chd_subs_set_options(chd_ctx *ctx, int64_t now_ns, uint32_t n, const chd_sub_options *opts,
                         uint8_t *should_send /* n, optional */, int32_t *status /* n, optional */);
/* DataAccess (CHD_ACCESS_*) and SkipSelfUpdateFanOut of slot's subscriptions, in the order of chd_subs_get. */
int chd_subs_get_options(chd_ctx *ctx, uint32_t slot, uint8_t *data_access, uint8_t *skip_self, uint32_t *n_out);

/* fan-out record: one fanOutDataUpdate decision (data.go:293-318). */
#define CHD_REC_FULL 0x80000000u /* in .conn: first fan-out, whole channel data */
typedef struct {
    uint32_t conn;    /* ConnectionId (31 bits, settings.go:90) | CHD_REC_FULL */
    uint32_t channel; /* ChannelId: a spatial channel or an entity channel */
} chd_fanout_rec;

typedef struct {
    uint32_t entity;     /* entity slot */
    uint32_t channel;    /* entity channel id */
    uint32_t src, dst;   /* spatial channel ids */
    uint32_t src_server, dst_server; /* ServerIndex of src/dst (spatial.go:336-351);
                                        cross-server iff different (:683) */
} chd_handover_rec;

typedef struct {
    int64_t now_ns; /* ChannelTime of this tick (channel.go:28-37); also the
                       arrival time stamped on this batch's updates (data.go:161) */
    /* entity ChannelDataUpdates merged this tick */
    uint32_t n_updates;
    const uint32_t *upd_idx;    /* entity slots, NULL => slot u = u */
    const double *upd_x, *upd_z;/* new position (SpatialInfo X,Z; Y is ignored) */
    const uint32_t *upd_sender; /* senderConnId, NULL => the entity's owner */
    /* spatial-channel data updates (spawn/destroy merges), optional; applied in array order */
    uint32_t n_cell_updates;
    const uint32_t *cell_upd_channel; /* spatial channel ids */
    const uint32_t *cell_upd_sender;
    /* UPDATE_SPATIAL_INTEREST messages (message_spatial.go:41-129) */
    uint32_t n_queries;
    const uint32_t *query_sub;  /* subscriber slots, NULL => slot i = i */
    const chd_aoi_query *queries;
    const double *spot_x, *spot_z;
    const uint32_t *spot_dist;
    uint32_t n_spots_total;
    /* ---- exact update buffers (chd_world_cfg.history_depth > 0; ignored fields must be 0 / NULL otherwise) ----
     * The reference stamps every update when it is ENQUEUED (Channel.PutMessage: arrivalTime = ch.GetTime(), channel.go:296-310
     * -> handleChannelDataUpdate, message.go:651 -> OnUpdate, data.go:159-164), not when the tick handles it: an update
     * enqueued at 249 ms and handled by the tick at 260 ms belongs to the fan-out window [200, 250].
     * upd_arrival_ns[u] / cell_upd_arrival_ns[u]: that stamp, per update; <= now_ns, >= 0, not decreasing per channel
     * (queue order).  NULL => now_ns (every update arrives with its tick). */
    const int64_t *upd_arrival_ns;
    const int64_t *cell_upd_arrival_ns;
    /* Several updates of one entity channel between two ticks: the reference handles them one after the other (each its own
     * Notify and its own buffer element).  The host hands them over in ROUNDS — round r holds every entity at most once, a
     * channel's r-th update of the tick — as consecutive ranges of the update arrays: round r = [upd_round_off[r],
     * upd_round_off[r+1]), upd_round_off[n_update_rounds] == n_updates; the rounds are applied in order (one ingest launch
     * each).  n_update_rounds == 0 => one round (the precondition of chd_tick_device then covers all of upd_idx).  At most 256
     * rounds per tick (CHD_E_INVAL beyond: a channel with more than 256 updates between two ticks is split over two ticks by
     * the host).  upd_round_off is a HOST array in chd_tick AND chd_tick_device (the library reads it to launch). */
    uint32_t n_update_rounds;
    const uint32_t *upd_round_off;
} chd_tick_in;

typedef struct {
    /* caller-allocated, any of them may be NULL (then that output stays on the
     * device and only the counts are returned) */
    chd_handover_rec *handovers; uint32_t handovers_cap; uint32_t n_handovers;
    uint32_t n_locked_aborts;          /* handovers aborted by a lock (spatial.go:675-679) */
    int32_t *query_status;             /* n_queries */
    uint32_t *unsub_sub, *unsub_channel; uint32_t unsub_cap; uint32_t n_unsubs;
    /* subscriptions created this tick (SubscribeToChannel's new branch,
     * subscription.go:59-91): subscriber slot, spatial channel, damped interval.
     * Re-subscriptions of already subscribed channels only merge the interval
     * (subscription.go:44-57) and are visible through chd_subs_get. */
    uint32_t *newsub_sub, *newsub_channel, *newsub_interval_ms; uint32_t newsub_cap; uint32_t n_newsubs;
    chd_fanout_rec *records; uint64_t records_cap; uint64_t n_records;
    uint64_t *conn_rec_off;            /* max_subscribers+1: slot s owns records
                                          [conn_rec_off[s], conn_rec_off[s]+conn_rec_cnt[s]) */
    uint32_t *conn_rec_cnt;            /* max_subscribers */
    uint32_t overflow;                 /* !=0: chd_tick_fetch returns CHD_E_CAPACITY.  Bits: 1 handover list, 2 unsub list,
                                          4 record buffer, 8 new-sub list truncated (grow the *_cap / max_records);
                                          region-sharded worlds: 16 no free entity slot for an immigrant (it waits, in no
                                          cell table, and takes the first slot that frees up: results are incomplete while
                                          the bit is set - size max_entities with headroom for clustering; 128: more than
                                          max_entities of them waiting, one was dropped), 32 an emigrant did not fit its destination's send
                                          segment (it stays with the wrong owner and is retried next tick), 64 a border
                                          band outgrew its halo segment or a subscription reaches beyond the halo;
                                          256 chd_tick_device: an entity slot twice in one round of updates, or a subscriber
                                          slot twice (the caller's precondition);
                                          0x4000 CHD_WORLD_GATED_OVERLAP: a device-side gate was not raised within its spin bound —
                                          this tick's results are NOT valid; the world orders its streams with HIP events from
                                          the next tick on (chd_tick_stats.gate_timeouts);
                                          0x8000 an internal loop bound tripped (a bug, never a capacity) */
    uint32_t history_overflow;         /* windows reaching beyond the 32-tick update history, or channels
                                          updated by more than two senders inside it (results then inexact);
                                          history_depth > 0: windows reaching an update the exact buffer had to drop */
    uint32_t *record_masks;            /* optional, records_cap entries, parallel to `records`: CHD_WORLD_UPDATE_MASKS
                                          worlds only (else ignored); 0 for CHD_REC_FULL records.  Worlds with history_depth:
                                          a word with bit 31 SET is in range form — the message merges the buffered updates
                                          of the channel numbered [first, first + count), first = bits 20..0 (the channel's
                                          update counter, mod 2^21: how many updates it had taken before that one), count - 1 =
                                          bits 30..21 (the elements of updateMsgBuffer whose arrivalTime lies in the window,
                                          data.go:236-241; with SkipSelfUpdateFanOut minus the recipient's own) — records the
                                          tick ring cannot answer (mid-tick stamps, windows older than 31 ticks, a third
                                          sender); tick-ring words have bit 31 clear there */
} chd_tick_out;

/* replaces, for the whole world in one call: Notify (spatial.go:612-736,
 * decision + entity-map update), handleUpdateSpatialInterest
 * (message_spatial.go:41-129) and tickData on every spatial and entity channel
 * (data.go:175-318).  Order inside a tick: entity updates, interest updates,
 * fan-out at now_ns.  now_ns must not decrease between ticks. */
int chd_tick(chd_ctx *ctx, const chd_tick_in *in, chd_tick_out *out);

/* Same, with every input already resident in device memory (d_* pointers of the
 * same layout) and outputs left on the device; asynchronous on the ctx stream.
 * This is what bench.py times.  Use chd_tick_fetch to read the outputs back.
 * PRECONDITION (chd_tick sees the host arrays and returns CHD_E_INVAL up front): inside one round of updates upd_idx holds
 * no entity slot twice, and query_sub no subscriber slot twice — the host coalesces (keeping the last, as the reference's
 * sequential handlers would leave it) or hands several updates of a channel over in rounds (upd_round_off).  Here the
 * device checks it while it ingests: a repeated slot sets overflow bit 256 (chd_tick_fetch then returns CHD_E_CAPACITY
 * with the mask; that tick's state for the slot is one of the two updates, its results are not to be used). */
int chd_tick_device(chd_ctx *ctx, const chd_tick_in *d_in);
int chd_tick_fetch(chd_ctx *ctx, chd_tick_out *out);
int chd_sync(chd_ctx *ctx);

/* Order-independent digest of the LAST tick's fan-out records, computed on the device over the records
 * where they lie (nothing crosses PCIe but the result): with h(r) = mix64(r.conn << 32 | r.channel), mix64 =
 * SplitMix64's finaliser ((k ^ k>>30) * 0xBF58476D1CE4E5B9, (k ^ k>>27) * 0x94D049BB133111EB, k ^ k>>31),
 * count = number of records, sum = sum of h (mod 2^64), xor_ = xor of h, sum_masked = sum of
 * mix64(h + merged-updates mask) (mask = 0 unless the world has CHD_WORLD_UPDATE_MASKS); conn_sum[s]
 * (optional, max_subscribers entries) = sum of h over connection slot s's records.  A receiver (or a test)
 * that folds the messages it got the same way can check a whole tick without sorting 10^8 records. */
typedef struct {
    uint64_t count, sum, xor_, sum_masked;
} chd_records_digest;
int chd_tick_digest(chd_ctx *ctx, chd_records_digest *total, uint64_t *conn_sum);

/* ------------------------------------------------------------------ */
/* The COMPACT form of a tick's fan-out: what a gateway needs to write  */
/* its sockets, without the expanded records crossing PCIe (8 B per     */
/* message: 644 MB per tick at BASELINE config B, against ~4 MB here).  */
/* ------------------------------------------------------------------ */

/* One segment = one (connection, spatial channel) subscription that fanned out this tick.  Most segments are PLAIN COPIES
 * of the cell's entity-channel column (every entity of the cell has an update inside every due window, or this is the first
 * fan-out): they travel as a reference into `columns` and expand on the host to
 *     CHD_SEG_FIRST : {conn | CHD_REC_FULL, channel}, then {conn | CHD_REC_FULL, columns[off + k]} for k < n
 *     then, for window j < CHD_SEG_NWIN(n_info): {conn, channel} if bit CHD_SEG_OWN(j) is set (the spatial channel's own
 *                     update passes window j), and, unless CHD_SEG_NONE, {conn, columns[off + k]} for k < n
 * (conn = the ConnectionId of the slot, which the host registered itself).  Everything else — subscriptions that needed a
 * per-entity decision — is CHD_SEG_EXPLICIT: records[conn_rec_off[slot] + off .. + n_records) hold its records as they are. */
#define CHD_SEG_N(n_info) ((n_info) & 0x3FFFFFu)            /* entities of the cell (explicit: unused) */
#define CHD_SEG_FIRST (1u << 22)
#define CHD_SEG_NONE (1u << 23)
#define CHD_SEG_EXPLICIT (1u << 24)
#define CHD_SEG_NWIN(n_info) (((n_info) >> 25) & 7u)
#define CHD_SEG_OWN(j) (1u << (28 + (j)))
typedef struct {
    uint32_t channel;   /* the spatial channel of the subscription */
    uint32_t off;       /* first entry in `columns`; CHD_SEG_EXPLICIT: first record, relative to conn_rec_off[slot] */
    uint32_t n_info;    /* CHD_SEG_* */
    uint32_t n_records; /* records the segment expands to */
} chd_fanout_segment;   /* 16 bytes */

typedef struct {
    chd_fanout_segment *segments; uint64_t segments_cap; uint64_t n_segments;
    uint32_t *conn_seg_off;   /* max_subscribers + 1: slot s owns segments [conn_seg_off[s], conn_seg_off[s+1]) */
    uint32_t *columns; uint64_t columns_cap; uint64_t n_columns;  /* the tick's cell-sorted entity channel ids */
    chd_fanout_rec *records; uint64_t records_cap; uint64_t n_explicit;  /* the explicit segments' records, per slot */
    uint64_t *conn_rec_off;   /* max_subscribers + 1: slot s's explicit records start at conn_rec_off[s] */
    uint64_t n_records;       /* what all segments expand to (== chd_tick_out.n_records of the same tick) */
} chd_segments_out;

/* The last tick's fan-out in that form (`columns`: max_entities entries — ten times that on ticks where some live entity skipped
 * an update, when the per-window columns travel too).  All buffers caller-allocated (chd_host_alloc memory makes the copies DMA);
 * CHD_E_CAPACITY (n_* say what is needed) when one is too small.  replaces: the per-message loop of fanOutDataUpdate
 * (data.go:293-318) on the host side of the boundary — the host walks segments instead of 8-byte records. */
int chd_tick_fetch_segments(chd_ctx *ctx, chd_segments_out *out);

/* chd_tick followed by chd_tick_fetch_segments, as ONE call: what a gateway that writes its sockets from segments does every tick.
 * Same results as the two calls; two host synchronisations instead of five (uploads + tick + sizing pass | lists + filling pass +
 * downloads).  `out` as for chd_tick but WITHOUT the dense record outputs (records, conn_rec_off, conn_rec_cnt must be NULL:
 * CHD_E_INVAL).  If `seg`'s buffers are too small the tick is still done and its lists fetched: CHD_E_CAPACITY with the needed
 * sizes in seg->n_*, grow and call chd_tick_fetch_segments. */
int chd_tick_segments(chd_ctx *ctx, const chd_tick_in *in, chd_tick_out *out, chd_segments_out *seg);

/* The same tick and the same outputs as an ASYNCHRONOUS PAIR, for a host loop that prepares tick t+1 while tick t's results travel:
 *     chd_tick_segments_begin(ctx, &in[t+1]);     enqueue only: uploads, the tick, the segment passes — no host wait
 *     chd_tick_segments_end(ctx, &blk);            tick t: wait for its last kernel, ONE copy of its block, pointers into it
 * _begin uploads the inputs on a side stream (while the tick in flight still runs), enqueues the tick on the ctx stream and packs
 * everything the tick hands out — a header with every count and offset, the per-connection offsets, query status, columns,
 * segments, explicit records, handover records, the unsub / new-sub lists — into ONE block in device memory.  _end waits for that
 * tick's last kernel (the next tick's kernels are queued behind it and start at once), reads the sizes from the header (the device
 * wrote it into page-locked memory itself: no sizing round trip), copies exactly the block's bytes with one DMA into its page-locked
 * twin while the next tick runs, and returns pointers into it.  At most TWO ticks in flight (a third _begin: CHD_E_STATE).  _end
 * returns ticks in the order they were begun; its pointers stay valid until the next-but-one _begin (the same parity's).  Results:
 * those of chd_tick_segments on the same inputs — offsets, segments and columns byte for byte; explicit records and lists in the
 * order of the device's atomics, as there.  The block is fixed-size (the columns + 96 MiB: 6M segments or 12M explicit records): a
 * tick beyond it returns CHD_E_CAPACITY from _end with the counts — the world has advanced; chd_tick_fetch +
 * chd_tick_fetch_segments still deliver that tick if called before the next _begin.  Do not interleave with chd_tick /
 * chd_tick_device / chd_tick_segments while a tick is in flight (_end first); not on region-sharded worlds (CHD_E_STATE).
 * replaces: the host side of Channel.tickData → fanOutDataUpdate (data.go:201-233, 235-318) as a loop that never idles the device. */
typedef struct {
    const uint32_t *conn_seg_off;          /* max_subscribers + 1 */
    const uint64_t *conn_rec_off;          /* max_subscribers + 1 */
    const chd_fanout_segment *segments; uint64_t n_segments;
    const uint32_t *columns; uint64_t n_columns;
    const chd_fanout_rec *records; uint64_t n_explicit;
    uint64_t n_records;                    /* what the segments expand to */
    const chd_handover_rec *handovers; uint32_t n_handovers; uint32_t n_locked_aborts;
    const uint32_t *unsub_sub, *unsub_channel; uint32_t n_unsubs;
    uint32_t n_newsubs;
    const uint32_t *newsub_sub, *newsub_channel, *newsub_interval_ms;
    const int32_t *query_status; uint32_t n_queries;
    uint32_t overflow, history_overflow;   /* as chd_tick_out */
    uint32_t reserved;
    const void *block; uint64_t block_bytes; /* the page-locked block all pointers above point into; the bytes of it this tick filled */
    float wait_ms, copy_ms;                /* host time inside _end: blocked until the tick's last kernel | the copy of the block */
    float device_ms;                       /* chd_set_profiling(depth > 0): ctx-stream time of the tick + the segment passes (HIP events); else 0 */
    uint32_t reserved2;
} chd_segments_block;
int chd_tick_segments_begin(chd_ctx *ctx, const chd_tick_in *in);
int chd_tick_segments_end(chd_ctx *ctx, chd_segments_block *out);

/* Read back the interest set of a subscriber slot (the keys of
 * Connection.spatialSubscriptions, with the per-subscription fan-out state of
 * subscription.go:13-31 / data.go:39-44).  Arrays have max_interest_cells
 * entries; *n_out = count. */
int chd_subs_get(chd_ctx *ctx, uint32_t slot, uint32_t *channel,
                 uint32_t *interval_ms, int64_t *last_fanout_ns,
                 uint8_t *had_first, uint8_t *is_new, uint32_t *n_out);
/* entity state: position-derived cell id and the cell whose entity map holds it */
int chd_world_get_entities(chd_ctx *ctx, uint32_t n, const uint32_t *idx,
                           uint32_t *cell_channel, uint32_t *member_channel);

/* ------------------------------------------------------------------ */
/* Region-sharded worlds: one ctx per GPU, rank r owns the cells whose  */
/* ServerIndex (GetRegions, spatial.go:336-351) is r - the partition    */
/* CreateChannels gives spatial server r (spatial.go:399-424).  Every   */
/* rank is created with the SAME global grid config and the SAME       */
/* max_entities (the halo segment capacities are derived from it on     */
/* both sides of every exchange).  Two exchange                         */
/* steps per tick, both all-to-all, both for border traffic only        */
/* (DESIGN.md section 7):                                               */
/*   chd_shard_ingest -> all-to-all(emigrants: cross-server handovers)  */
/*   chd_shard_import -> all-to-all(halo: the border bands of the cell  */
/*                       tables, ServerInterestBorderSize cells wide)   */
/*   chd_shard_fanout                                                   */
/* All pointers are DEVICE pointers (the exchange buffers are owned by  */
/* the caller, e.g. torch tensors handed to RCCL); calls are            */
/* asynchronous on the ctx stream.                                      */
/* ------------------------------------------------------------------ */

/* external != 0: enqueue all later work of this ctx on the caller's HIP stream
 * (hipStream_t; NULL is the legacy default stream, which is what torch's default
 * "current stream" is) so that it orders with the caller's collectives without host
 * synchronisation.  external == 0 restores the ctx's own non-blocking stream. */
int chd_set_stream(chd_ctx *ctx, void *hip_stream, int external);

/* An entity crossing to another rank: the cross-server handover of
 * spatial.go:683-700 carries the entity's whole engine-side state. */
typedef struct {
    uint32_t chan_id;  /* entity channel id */
    uint32_t cell;     /* cell index of the last merged position (0xFFFFFFFF = out of world) */
    uint32_t member;   /* cell index whose entity map holds it */
    uint32_t eflags;
    uint32_t sender;
    uint32_t hist;     /* update history of `sender`, aligned to the tick of the export */
    uint32_t sender_prev, hist_prev; /* the previous sender's updates still buffered */
} chd_entity_state; /* 32 bytes */

/* Sharded worlds address entities by channel id; slots are allocated by the library.
 * Spawns the entities given (the caller passes only those whose cell belongs to this
 * rank; out-of-world entities may live on any one rank).  Not mixable with
 * chd_world_spawn on the same ctx. */
int chd_shard_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id, const double *x,
                    const double *z, const uint32_t *flags, const uint32_t *sender);

/* The entity channels `chan_id` leave the world (RemoveChannel of an entity channel: the entity is destroyed): whichever rank holds one
 * frees its slot; on worlds with an update log by channel id the channel's log is closed (a later chd_shard_log_spawn of the same id
 * starts an empty one).  EVERY rank is given the same list, between two ticks.  chd_world_despawn on unsharded worlds. */
int chd_shard_despawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id);

/* Phase 1.  Starts tick `now_ns`.  Every live entity e of this rank reads its new
 * position from d_x_by_chan/d_z_by_chan[e.chan_id - entity_channel_id_start]
 * (d_has_update, if not NULL, marks which channels carry an update this tick) and goes
 * through the Notify decision like chd_tick's ingest; then every entity whose member cell
 * now belongs to another rank is packed into the send buffer and leaves this rank.
 * d_send = world segments of (cap + 1) records: record 0 of segment `dst` is a header whose
 * chan_id field holds the number of emigrants k, followed by the k states.  Equal-sized
 * segments: one all-to-all moves them.  If a destination's `cap` is exceeded the surplus
 * stays and retries next tick (overflow flag).
 * cap_used (optional): a few hundred entities cross a region border per tick while `cap` is sized for the worst burst, and
 * an all-to-all of full segments moves (cap + 1) x 32 B per peer whatever they hold.  With cap_used != NULL the library
 * picks this tick's segment capacity itself — the caller's `cap` for the first ticks, then four times the largest segment
 * count any rank saw two ticks ago (every header carries its sender's maximum, so every rank derives the SAME value and the
 * all-to-all sizes agree without a collective), a power of two in [256, cap] — lays the segments out at a pitch of
 * (*cap_used + 1) records and returns it: exchange world x (*cap_used + 1) records and pass *cap_used to chd_shard_import.
 * Record 0's other fields are reserved (the library uses `cell`).
 * LIFETIME of d_x_by_chan / d_z_by_chan / d_has_update on a world with the update log by channel id (shard_channels + history_depth):
 * the tick's updates are logged by chd_shard_IMPORT, behind the emigrant exchange (which carries the cells' maxFanOutIntervalMs) — the
 * three arrays must stay valid and unmodified until that call has been enqueued; it releases them.  One import per ingest: a
 * chd_shard_import without a fresh chd_shard_ingest[_pre] answers CHD_E_STATE (it would log the tick's updates twice). */
int chd_shard_ingest(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan,
                     const double *d_z_by_chan, const uint8_t *d_has_update, uint32_t n_chan,
                     uint32_t rank, uint32_t world, chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used);

/* The halo exchange.  Rank s sends rank d the cell tables of the cells of ITS region that lie within
 * ServerInterestBorderSize cells of d's region (the grid config's own parameter for "how much of a neighbour a spatial
 * server sees", spatial.go:114-118,481-590, generalised from the 4-neighbour border rows to a band around the whole
 * region incl. the corners): one fixed-capacity segment per ordered pair of ranks, none for ranks further apart.  The
 * layout is a pure function of (grid config, max_entities, rank): segs[p] = where the segment TO rank p sits in this
 * rank's send buffer and where the segment FROM rank p lands in its receive buffer — the split sizes of one
 * all-to-all(v).  Must be called once with the ctx's own rank before the first tick (it reserves room for the
 * neighbours' border entities); it may be called for other ranks too (a host-staged exchange needs their offsets).
 * A subscription that reaches a cell beyond the halo, or a band that outgrows its segment, sets overflow bit 64
 * (CHD_E_CAPACITY from chd_tick_fetch): widen ServerInterestBorderSize / max_entities. */
typedef struct {
    uint64_t send_off, send_bytes, recv_off, recv_bytes;
} chd_halo_seg;
int chd_shard_halo_layout(chd_ctx *ctx, uint32_t rank, uint32_t world, chd_halo_seg *segs /* world */,
                          uint64_t *send_total, uint64_t *recv_total);

/* Phase 2, after the all-to-all of emigrants: the states in d_recv (same layout as d_send, segment `src` = what rank src
 * sent here; cap = the capacity chd_shard_ingest used this tick) join this rank; the local cell index is rebuilt and the
 * border bands are packed into d_halo_send (chd_shard_halo_layout; may be NULL when world == 1).
 * An immigrant that finds no free slot is NOT lost: it waits in a side list and takes the first slot that frees up at a
 * later import; every tick it waits (in no cell table, visible to nobody) sets overflow bit 16. */
int chd_shard_import(chd_ctx *ctx, const chd_entity_state *d_recv, uint32_t world,
                     uint32_t cap, void *d_halo_send);

/* Optional, any time between chd_shard_ingest and chd_shard_fanout: the interest updates
 * of this rank's connections (the query fields of d_in).  They do not depend on the
 * gathered tables, so a caller can run them while the all-gather is in flight and then
 * pass no queries to chd_shard_fanout (e.g. while the halo exchange is in flight). */
int chd_shard_interest(chd_ctx *ctx, const chd_tick_in *d_in);

/* Phase 3, after the halo all-to-all: the received bands (d_halo_recv, chd_shard_halo_layout) join the own cell tables
 * as ghost entries; then the interest updates of d_in (queries of this rank's connections) and the fan-out of this rank's
 * connections over region + halo.  Of d_in's update fields the ENTITY updates are ignored (they came by channel id with
 * chd_shard_ingest); the SPATIAL CHANNELS' own updates (n_cell_updates, cell_upd_channel / _sender / _arrival_ns: device arrays) are
 * applied — per-cell state that every rank keeps for every cell, so EVERY rank is given the same, whole-world list (a cell's
 * subscribers live on its owner's rank and on the neighbours whose border it is).  Outputs as chd_tick_device. */
int chd_shard_fanout(chd_ctx *ctx, const void *d_halo_recv, uint32_t world, const chd_tick_in *d_in);

/* Who SENT the updates (senderConnId of data.go:159-164; SkipSelfUpdateFanOut compares it, data.go:242-245): a DEVICE array indexed
 * like the positions of chd_shard_ingest (channel id - EntityChannelIdStart) that the following ticks read at ingest — the host
 * keeps it resident and changes entries as owners change (after a cross-server handover the updates come from the new server's
 * connection, spatial.go:683-700).  NULL / n_chan 0: back to every entity's own sender as given at spawn (it migrates with the
 * entity).  chd_tick_in.upd_sender on unsharded worlds. */
int chd_shard_set_update_senders(chd_ctx *ctx, const uint32_t *d_sender_by_chan, uint32_t n_chan);

/* Worlds with an update log by channel id (chd_world_cfg.shard_channels) only.
 * chd_shard_log_spawn: EVERY rank is told about EVERY entity channel that comes to life and where (the whole-world list, the same on
 * every rank, beside chd_shard_spawn's own-region list): the log notes the cell of the channel's first position (what the first
 * update's Notify compares with, and what the channel's maxFanOutIntervalMs starts from) and takes updates for it from then on.
 * chd_shard_set_update_arrivals: WHEN each update was enqueued (arrivalTime = ch.GetTime() in Channel.PutMessage,
 * channel.go:296-310): a DEVICE array of int64 ns indexed like the positions (channel id - EntityChannelIdStart), read by the
 * following ticks' ingest; resident, the host rewrites it between ticks.  NULL: every update is stamped with its tick's now_ns.
 * chd_tick_in.upd_arrival_ns on unsharded worlds.  The senders must come by channel id as well (chd_shard_set_update_senders)
 * — a rank does not hold the entity whose "own sender" it would otherwise fall back to.
 * d_has_update of chd_shard_ingest / chd_shard_tick must only mark channels that exist somewhere (every rank logs them). */
int chd_shard_log_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id, const double *x, const double *z);
int chd_shard_set_update_arrivals(chd_ctx *ctx, const int64_t *d_arrival_ns_by_chan, uint32_t n_chan);
/* The emigrant segments of such a world carry, behind their (cap + 1) records, the sending rank's maxFanOutIntervalMs per
 * spatial channel (4 B per cell): *extra = that many more 32-byte records per segment (0 on other worlds).  A caller that runs
 * the exchange itself lays the segments out at a pitch of (cap_used + 1 + extra) records; chd_shard_tick does so by itself. */
int chd_shard_migrate_extra_records(chd_ctx *ctx, uint32_t *extra);

/* Handover lists on a region-sharded world: the meaning of chd_world_set_handover_lists, keyed by ENTITY CHANNEL ID — slots are
 * the library's here and an entity changes ranks.  List k = the channel ids list_member_chan[list_off[k] .. list_off[k+1]); the
 * entity with channel id chan_id[i] takes list list_of[i] (CHD_NO_HANDOVER_LIST: the entity itself; an EMPTY list: no handover,
 * counted in n_locked_aborts).  Channel ids lie in [entity_channel_id_start, + n_chan).  EVERY rank is given the same, complete
 * arrays (the group controllers are the host's, entity.go:58-244): a list's members may live on any rank — those that move
 * with a handover are the ones in src's entity map (spatial.go:703-736), which is the notifier's rank's by construction; a
 * member that thereby lands in another rank's region emigrates with the tick's exchange like any other entity.  Replaces the
 * whole group state of this rank (n_lists == 0 clears it). */
int chd_shard_set_handover_lists(chd_ctx *ctx, uint32_t n_lists, const uint32_t *list_off /* n_lists + 1 */,
                                 const uint32_t *list_member_chan, uint32_t n, const uint32_t *chan_id, const uint32_t *list_of,
                                 uint32_t n_chan);

/* With lists, a handover can concern ANOTHER rank's entity map: the notifier lives on the rank of the cell whose map holds it, and
 * after a list-mate's handover pulled it across a region border that is not the rank of the cell it stands in (src).  The members
 * to move are in src's map, so the handover travels to src's rank as a 16-byte request and is applied there before anything is
 * exported.  chd_shard_tick does this by itself; a host that drives the exchanges calls, instead of chd_shard_ingest (which
 * refuses on a world with lists and world > 1):
 *   chd_shard_ingest_pre   positions -> cells, handovers, local list members moved; d_req_send = world segments of
 *                          (req_cap + 1) records, record 0 = {count} (more than req_cap: overflow bit 32)
 *   all-to-all of the segments (equal sizes)
 *   chd_shard_ingest_post  the received requests applied, the emigrants exported into d_send as chd_shard_ingest does */
typedef struct chd_handover_request {
    uint32_t list;     /* index into the lists every rank was given (record 0: the count) */
    uint32_t src, dst; /* cell indices */
    uint32_t notifier; /* its entity channel id: it moved, or not, on its own rank */
} chd_handover_request;
#define CHD_SHARD_REQ_CAP 1024u /* what chd_shard_tick uses per peer */
int chd_shard_ingest_pre(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan,
                         const uint8_t *d_has_update, uint32_t n_chan, uint32_t rank, uint32_t world,
                         chd_handover_request *d_req_send, uint32_t req_cap);
int chd_shard_ingest_post(chd_ctx *ctx, const chd_handover_request *d_req_recv, uint32_t req_cap, uint32_t rank, uint32_t world,
                          chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used);

/* ---- Native collectives: the two exchanges inside the library, on RCCL over xGMI (librccl.so, dlopen'ed here: a single-GPU
 * gateway never loads it).  replaces: the transport of the cross-server handover (spatial.go:683-700: the reference sends the
 * handover data to the destination spatial server's connection) and of the border subscriptions (spatial.go:481-590), which in
 * the reference are TCP/KCP messages between the gateway and the spatial servers.  A Go host calls
 *     rank 0:  chd_shard_comm_unique_id(id)            -> ships the 128 bytes to the other gateways over its own control channel
 *     all:     chd_shard_comm_init(ctx, id, rank, world, migrate_cap)   (collective: every rank must call it)
 *     tick:    chd_shard_tick(ctx, now_ns, positions ..., d_in)         (collective)
 * and nothing else: the exchange buffers are the library's.  migrate_cap = capacity of one emigrant segment (entities crossing
 * to ONE other rank in one tick; the library shrinks it to the traffic as chd_shard_ingest's cap_used does).  Same results as
 * the four chd_shard_* stages around caller-run collectives (channeld_amd/dist.py keeps that path for host-staged test
 * transports); outputs as chd_tick_device (chd_tick_fetch, chd_tick_digest, ...). */
#define CHD_COMM_ID_BYTES 128
/* Can this process load RCCL at all?  LOCAL, no collective: every gateway asks this (and tells the others) BEFORE any of them enters
 * chd_shard_comm_init, which is a collective and would wait for a rank that never comes.  CHD_OK, or CHD_E_STATE with the
 * loader's message in chd_last_error(NULL). */
int chd_shard_comm_available(void);
/* (CHD_SHARD_TRANSPORT=hostpipe in rank 0's environment makes the id name a POSIX shared-memory segment instead: a blocking, host-staged
 * TEST transport for running chd_shard_tick with several ranks on one GPU — call sequence, sizes and the gated join, none of RCCL's
 * asynchrony.  It and the library's other test hook, CHD_TEST_DROP_GATE_RAISE, are environment-driven; a production build may compile
 * both out with -DCHD_NO_TEST_HOOKS, and chd_shard_comm_init then refuses such an id.) */
int chd_shard_comm_unique_id(void *id_out /* CHD_COMM_ID_BYTES */);
/* On failure nothing is left behind (no communicator, stream or events) and the call may be repeated with a fresh id. */
int chd_shard_comm_init(chd_ctx *ctx, const void *unique_id, uint32_t rank, uint32_t world, uint32_t migrate_cap);
int chd_shard_comm_destroy(chd_ctx *ctx);
/* One tick: chd_shard_ingest -> all-to-all(emigrants) -> chd_shard_import -> all-to-all(v)(halo) beside chd_shard_interest ->
 * chd_shard_fanout, enqueued on the ctx stream (and the library's second stream) with events between them: no host code, no
 * host synchronisation.  Arguments as chd_shard_ingest (positions by channel id) and chd_shard_fanout (d_in: the queries).
 * On a world created with CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP the interest updates run on the second stream from
 * the tick's START — beside ingest, both exchanges and the index build; they read nothing the front writes — and are joined by a
 * device-side flag inside the ghost-table unpack; the halo exchange then runs in stream order on the ctx stream (no events).
 * Worlds with handover lists (chd_shard_set_handover_lists) take one more small exchange before the export. */
int chd_shard_tick(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan, const uint8_t *d_has_update,
                   uint32_t n_chan, const chd_tick_in *d_in);

/* Live entities of this rank: channel ids and cell / member channel ids (0 = none).
 * Arrays have max_entities room; *n_out = count. */
int chd_shard_get_entities(chd_ctx *ctx, uint32_t *chan_id, uint32_t *cell_channel,
                           uint32_t *member_channel, uint32_t *n_out);

/* ------------------------------------------------------------------ */
/* Recipient planning: WHO gets the messages the reference assembles    */
/* around the path (the protobuf assembly itself     */
/* stays on the host).  Connections = the client connections registered */
/* with chd_subs_add; the spatial servers' own subscriptions are static */
/* (chd_server_channels / chd_border_channels) and handled by the host. */
/* ------------------------------------------------------------------ */

/* Recipients of the ChannelDataHandoverMessage of every handover of the LAST tick
 * (spatial.go:776-857), evaluated on the subscriptions as they were when the handover
 * happened (before that tick's interest updates).  Needs CHD_WORLD_HANDOVER_RECIPIENTS.
 * Handover h (index into chd_tick_out.handovers of the same tick) owns
 * [offsets[h], offsets[h+1]) of conn/kind, ascending connection slot. */
#define CHD_HO_SRC_ONLY 0  /* in src only: the message without per-recipient entity data (:780-787) */
#define CHD_HO_DST_NEW 1   /* in dst, not yet subscribed to the entity channel (= not in src): full entity
                              data, SubscribeToChannel(entity channel) (:797-857, shouldSend) */
#define CHD_HO_DST_KNOWN 2 /* in dst and in src: already subscribed to the entity channel */
int chd_handover_recipients(chd_ctx *ctx, uint32_t *offsets /* n_handovers+1 */, uint32_t *conn,
                            uint8_t *kind, uint64_t cap, uint64_t *n_out);
/* Who OWNS the spatial channels: ConnectionId of spatial server k, k < n_servers = ServerCols x ServerRows (CreateChannels gives
 * server k's connection the cells of region k, spatial.go:399-424; ctl.serverConnections).  The handover loop subscribes every dst
 * connection to every handover entity's channel with DataAccess = WRITE for the entity channel's owner, else READ
 * (spatial.go:812-817), and `shouldSend` — the entity goes out WITH its full data — is also true when that merge CHANGES the
 * connection's DataAccess (subscription.go:44-57): on a cross-server handover the dst server's connection (subscribed through its
 * border interest with READ, now the owner: WRITE) and the src server's, if it keeps interest in dst (WRITE -> READ).  In the tick
 * model an entity channel's owner is the server of the cell that holds it.  Server connections take part as subscribers like any
 * other (chd_subs_add + chd_subs_set_options for their region and border cells); this table only says which ConnectionId is
 * which server.  n_servers == 0 clears it (no connection is an owner: chd_handover_recipients_ex then reports "newly subscribed"
 * alone, as before ABI v9). */
int chd_world_set_server_connections(chd_ctx *ctx, uint32_t n_servers, const uint32_t *conn_ids);

/* The same, EXACT for handover groups.  The reference decides per (destination connection, ENTITY) whether the entity goes
 * out with its entityData: `shouldSend` of conn.SubscribeToChannel(entityCh) inside the loop over handoverEntities
 * (spatial.go:797-857) — the connection was not yet subscribed to THAT entity's channel.  A group's members may sit in different
 * cells (only those in src's entity map move, :703-736), so a connection of dst can know some of them and not others.
 * full_mask[i] (cap entries, parallel to conn / kind): bit q = entity q of the handover's entity list — the order
 * chd_handover_messages / chd_handover_variants write them: the notifier alone, or the live members of its handover list / group
 * in list order (first 32) — carries entityData for recipient i; 0 for CHD_HO_SRC_ONLY.  "Subscribed to the entity's channel"
 * = subscribed to the cell whose entity map held the entity when Notify ran: src for the members that
 * moved with this handover, the cell that still holds them for the others.  kind stays the single-entity classification of the
 * NOTIFIER'S cell pair (DST_NEW <=> not in src). */
int chd_handover_recipients_ex(chd_ctx *ctx, uint32_t *offsets /* n_handovers+1 */, uint32_t *conn, uint8_t *kind,
                               uint32_t *full_mask, uint64_t cap, uint64_t *n_out);
/* The ownership assumption behind full_mask's `dataAccessChanged` part: the DataAccess a connection holds on an entity channel is
 * DERIVED — WRITE exactly for the connection chd_world_set_server_connections names as the spatial server of the cell that holds the
 * entity, READ for every other connection — not the value stored in the subscription (subscription.go:44-57 compares the stored one).
 * A client that subscribed to an entity channel with WRITE explicitly, or an entity channel owned by a connection that is not its cell's
 * spatial server, diverges from the reference here; SURVEY 8c lists the Notify outputs as unpinned by anything the reference holds. */

/* Step 1 of a CROSS-SERVER handover (spatial.go:683-700): `ownerConn := srcChannel.GetOwner(); ownerConn != nil && !ownerConn.IsClosing()
 * && !ownerConn.HasInterestIn(dstChannelId)` -> ownerConn.UnsubscribeFromChannel(entityCh) + sendUnsubscribed, for every handover
 * entity.  flags[h] (handover h of the LAST tick, as chd_handover_recipients) = 1 when a live subscriber slot holds the ConnectionId
 * chd_world_set_server_connections names for src's server, src and dst belong to different servers, and dst is not among that
 * connection's spatial subscriptions (subscription.go:181-187) as they were when the handover happened; else 0 (also when the
 * owner's connection is not registered with chd_subs_add: the engine cannot know its interest).  Needs CHD_WORLD_HANDOVER_RECIPIENTS. */
int chd_handover_src_owner_unsubscribed(chd_ctx *ctx, uint8_t *flags /* cap */, uint32_t cap, uint32_t *n_out);

/* Region-sharded worlds (CHD_WORLD_HANDOVER_RECIPIENTS on every rank).  A handover's recipients are the connections subscribed to its
 * src or dst cell: they live on the rank that owns the cell and on the neighbours whose halo it is in — the placement the fan-out
 * already has.  Each rank detects the handovers of the entities it holds (chd_tick_fetch: handovers); the gateways gather the
 * ranks' records into ONE whole-world list (rank order, the same on every rank: they exchange handover messages anyway) and every
 * rank asks for ITS connections' share:
 *   handovers[n_handovers]   the whole world's handover records of the LAST tick (host memory; records between cells beyond this
 *                            rank's region + halo simply have no recipients here);
 *   offsets / conn / kind / full_mask   as chd_handover_recipients_ex, for this rank's connections, evaluated on the subscriptions
 *                            as they were at the START of the last tick (the library keeps that copy: Notify runs before the tick's
 *                            interest updates).  full_mask: bit 0 only — every handover as its notifier alone; on a world with
 *                            handover lists (chd_shard_set_handover_lists) pass NULL (CHD_E_STATE otherwise: the other members'
 *                            cells are other ranks' state);
 *   src_owner_unsubscribed[n_handovers]  (may be NULL) as chd_handover_src_owner_unsubscribed, for the src server's connection IF
 *                            THIS RANK HOLDS IT (a spatial server's connection is registered on its own gateway): the gateway that
 *                            reads 1 sends the unsubscribe; the union over the ranks is the single world's flag.
 * The union of the ranks' lists is the single world's recipient list (tests/test_gpu_shard.py: 2 and 4 ranks against the
 * single-world CPU restatement).  replaces: spatial.go:738-857 on a deployment of one channeld gateway per spatial server region. */
int chd_shard_handover_recipients(chd_ctx *ctx, uint32_t n_handovers, const chd_handover_rec *handovers, uint32_t *offsets /* n_handovers+1 */,
                                  uint32_t *conn, uint8_t *kind, uint32_t *full_mask, uint8_t *src_owner_unsubscribed, uint64_t cap, uint64_t *n_out);

/* replaces: the connection merge of BroadcastType_ADJACENT_CHANNELS (message.go:188-239):
 * for request r the de-duplicated connections subscribed to spatial channel channel[r] or
 * one of its (up to 8) adjacent channels, filtered by the broadcast flags exactly as the
 * reference does (the centre channel is included unless ALL_BUT_OWNER is set, :201-204;
 * ALL_BUT_SENDER drops sender_conn[r], :223-225; ALL_BUT_CLIENT drops every client
 * connection, :227-229; the connection client_conn[r] of the ServerForwardMessage is
 * always dropped, :235-237).  CSR output, ascending connection slot. */
#define CHD_BROADCAST_ALL_BUT_SENDER 4u
#define CHD_BROADCAST_ALL_BUT_OWNER 8u
#define CHD_BROADCAST_ALL_BUT_CLIENT 16u
#define CHD_BROADCAST_ALL_BUT_SERVER 32u
#define CHD_BROADCAST_ADJACENT_CHANNELS 64u
int chd_adjacent_recipients(chd_ctx *ctx, uint32_t n_req, const uint32_t *channel,
                            const uint32_t *broadcast, const uint32_t *sender_conn,
                            const uint32_t *client_conn, uint32_t *offsets /* n_req+1 */,
                            uint32_t *conns, uint64_t cap);

/* ------------------------------------------------------------------ */
/* Wire-format fan-out buffers: the bytes the gateway's   */
/* flush goroutine would write per connection for the tick's fan-out    */
/* messages, built on the device so that the host does one conn.Write   */
/* per connection instead of three marshals per message.                */
/* ------------------------------------------------------------------ */

/* replaces: anypb.New(updateMsg) results handed to fanOutDataUpdate (data.go:293-302).
 * The host keeps merging channel data (ChannelData.OnUpdate, data.go:149-173) and gives the
 * engine, per channel, the serialized google.protobuf.Any to fan out:
 *   kind CHD_WIRE_ENTITY_UPDATE / CHD_WIRE_ENTITY_FULL : idx = entity slots
 *   kind CHD_WIRE_CELL_UPDATE / CHD_WIRE_CELL_FULL     : idx = spatial channel ids
 * bytes = the payloads back to back, lens[i] bytes each.  A payload stays until replaced.
 * (A window that holds several buffered updates of one channel fans out that channel's
 * CURRENT payload once per window, as the reference sends one merged message per window.) */
#define CHD_WIRE_ENTITY_UPDATE 0
#define CHD_WIRE_ENTITY_FULL 1
#define CHD_WIRE_CELL_UPDATE 2
#define CHD_WIRE_CELL_FULL 3
#define CHD_WIRE_ENTITY_OBJREF 4 /* serialized unrealpb.UnrealObjectRef of the entity (idx = entity slots; at most wire_max_update_len
                                   bytes): what MergeTo puts into SpatialEntityState.objRef (chd_handover_messages) */
int chd_wire_set_payloads(chd_ctx *ctx, int kind, uint32_t n, const uint32_t *idx,
                          const uint32_t *lens, const uint8_t *bytes);

/* Worlds with CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS build the MERGED update of every fan-out message on the device
 * (data.go:225-269: tickData merges the buffered updates a subscriber's window selects — a 100 ms
 * subscriber of a 50 ms world gets two ticks' deltas in one message).  In such a world
 *   - the UPDATE payload kinds (CHD_WIRE_ENTITY_UPDATE, CHD_WIRE_CELL_UPDATE) are the serialized channel-data update
 *     MESSAGES themselves (not wrapped in Any); a payload set between two ticks belongs to the update that arrives
 *     with the NEXT chd_tick; the engine keeps the last 32 ticks' payloads per channel;
 *   - chd_wire_set_type_url gives the Any.type_url of the entity (cell = 0) / spatial (cell = 1) channel data message
 *     ("type.googleapis.com/..." as anypb.New writes it);
 *   - a message's data field is Any{type_url, value = the selected updates, oldest first, concatenated}.  A protobuf
 *     parser reads concatenated messages as their merge (proto.Merge semantics: last scalar wins, repeated fields
 *     append, sub-messages merge), so the receiver decodes what the reference's accumulated message decodes to whenever
 *     the channel data type uses the default merge; the BYTES differ from Go's re-marshalled merge (fields repeat), and
 *     custom Merge implementations / ChannelDataMergeOptions (list limits, removable map entries) stay with the host.
 *   The FULL payload kinds stay whole Any messages. */
int chd_wire_set_type_url(chd_ctx *ctx, int which /* 0 entity data, 1 spatial channel data updates, 2 handover data */,
                          const uint8_t *url, uint32_t len);

/* replaces, for one channel data type: the accumulation of tickData (data.go:249-253: proto.Merge of the first selected
 * update into an empty message, the type's Merge for the rest) + fmutils.Filter (data.go:294) + proto.Marshal — BYTE FOR BYTE.
 * CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS worlds.  With a schema set, an entity update record whose selected updates all lie
 * inside the schema carries Any{type_url, value = the merged message as Go marshals it (fields in field-number order)};
 * a record with an update outside it keeps the generic form (the selected updates concatenated: same decoded message,
 * other bytes).
 *   CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT: tpspb.EntityChannelData restricted to actorState.replicatedMovement
 *   {linearVelocity, angularVelocity, location, rotation: FVector{x, y, z}; bSimulatedPhysicSleep, bRepPhysics}
 *   (examples/channeld-ue-tps/tpspb/tps.proto:22-35, pkg/unrealpb/unreal_common.proto:55-59,161-184; EntityChannelData.Merge,
 *   tpspb/data.go:227-252: objRef dropped from all but the first update — an update WITH an objRef is outside the subset).
 * 0 = none (concatenate). */
#define CHD_MERGE_SCHEMA_NONE 0
#define CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT 1
int chd_wire_set_merge_schema(chd_ctx *ctx, int schema);

/* replaces: the message assembly of Notify (spatial.go:738-773,797-857; HandoverDataMerger.MergeTo,
 * examples/channeld-ue-tps/tpspb/data.go:323-347): for every handover of the LAST tick the two MessagePacks the
 * reference sends — ChannelDataHandoverMessage{srcChannelId, dstChannelId, contextConnId = the src channel's
 * latestDataUpdateConnId, data = Any{type_url (which = 2), SpatialChannelData{entities}}} behind MessageContext{MsgType
 * CHANNEL_DATA_HANDOVER, ChannelId = dstChannelId}:
 *   blob 2h     entities carry their objRef only — for the recipients of kind CHD_HO_SRC_ONLY and CHD_HO_DST_KNOWN;
 *   blob 2h + 1 entities also carry entityData = the entity's full state Any (CHD_WIRE_ENTITY_FULL) — for CHD_HO_DST_NEW
 *               (`shouldSend`: the connection was just subscribed to the entity channel).
 * The message does not depend on the recipient otherwise, so chd_handover_recipients says who gets which blob and the
 * host queues the same bytes for each of them.  The entities of a handover = the notifying entity, or all live
 * members of its handover group / handover list (chd_world_set_entity_groups, chd_world_set_handover_lists), each as one map
 * entry {netId = its entity channel id}.
 * Exactness: the reference decides `fullData` per (dst connection, ENTITY) (spatial.go:797-857: shouldSend of that entity
 * channel's SubscribeToChannel); the two blobs are exact for single-entity handovers — every handover without groups —
 * and for group handovers whose dst connections know all members or none.  A dst connection already subscribed to SOME
 * members of a group gets a mixed message in the reference: chd_handover_recipients_ex + chd_handover_variants below
 * build exactly that one.
 * Needs CHD_WORLD_WIRE, the CHD_WIRE_ENTITY_OBJREF / CHD_WIRE_ENTITY_FULL payloads and the type url (and
 * CHD_WORLD_HANDOVER_RECIPIENTS for chd_handover_recipients, which says who gets which blob).  n_handovers = what the
 * caller sized `offsets` for: 2 * n_handovers + 1 entries; CHD_E_CAPACITY when the last tick had more handovers. */
int chd_handover_messages(chd_ctx *ctx, uint32_t n_handovers, uint32_t *offsets, uint8_t *bytes, uint64_t cap, uint64_t *n_out);
/* ... and the exact form: one MessagePack per requested VARIANT (handover index of the last tick, full mask as
 * chd_handover_recipients_ex reports it) — the host asks for the distinct (handover, full_mask) pairs among a tick's
 * recipients (one or two per handover; a group whose members sit in several cells a few more) and queues blob v for every
 * recipient with that pair.  Variant v = bytes[offsets[v], offsets[v+1]).  replaces, byte for byte, what the loop of
 * spatial.go:797-857 marshals per destination connection (handoverMerger.MergeTo(handoverDataMsg, shouldSend) per entity). */
int chd_handover_variants(chd_ctx *ctx, uint32_t n_var, const uint32_t *var_handover, const uint32_t *var_full_mask,
                          uint32_t *offsets /* n_var + 1 */, uint8_t *bytes, uint64_t cap, uint64_t *n_out);

/* replaces, for every connection at once: queuedMessagePackSender.Send (connection.go:57-83:
 * MessagePack{ChannelId, MsgType: CHANNEL_DATA_UPDATE, MsgBody: ChannelDataUpdateMessage{Data}},
 * packs of 65530 bytes or more are dropped) and flush (connection.go:626-714: greedy Packets of
 * at most 65535 bytes, each behind the tag 'C','H',size_hi,size_lo,compression=0) for the fan-out
 * records of the LAST tick, in the order chd_tick_fetch reports them.  The streams stay on the
 * device (chd_wire_fetch copies them out); *total_bytes, *total_packets, *dropped are optional. */
int chd_wire_build(chd_ctx *ctx, uint64_t *total_bytes, uint64_t *total_packets, uint32_t *dropped);

/* How the last chd_wire_build produced its streams (diagnostics; no reference counterpart).  On ticks that took the
 * descriptor path of the fan-out the streams are concatenations of per-cell message IMAGES (every subscriber of a cell gets
 * the same bytes apart from where the packet tags fall): *n_image_ranges = ranges copied from the images (0: the streams
 * were built record by record), *n_record_path_connections = connections with at least one subscription that was still
 * walked record by record (per-entity decisions, merged updates, a message Send drops). */
int chd_wire_build_info(chd_ctx *ctx, uint64_t *n_image_ranges, uint32_t *n_record_path_connections);

/* conn_off[s] .. conn_off[s+1] = connection slot s's stream inside bytes (max_subscribers+1
 * offsets); conn_packets[s] = its number of packets.  bytes may be NULL (offsets only). */
int chd_wire_fetch(chd_ctx *ctx, uint64_t *conn_off, uint32_t *conn_packets, uint8_t *bytes, uint64_t cap);

/* device memory, for callers that keep their batches on the GPU (bench, tests) */
int chd_dev_alloc(chd_ctx *ctx, uint64_t bytes, void **d_out);
int chd_dev_free(chd_ctx *ctx, void *d_ptr);
int chd_dev_upload(chd_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int chd_dev_download(chd_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);

/* Page-locked host memory for the buffers chd_tick / chd_tick_fetch / chd_wire_fetch fill: the device writes it by
 * DMA at PCIe rate (pageable memory is staged through a bounce buffer at a fraction of it).  A cgo shim allocates
 * its per-tick output buffers here once and reuses them (C memory, not Go memory: no cgo pointer rules apply). */
int chd_host_alloc(chd_ctx *ctx, uint64_t bytes, void **out);
int chd_host_free(chd_ctx *ctx, void *ptr);

/* metrics (channel_tick_duration analogue, channel.go:382-383): GPU time of the last
 * tick per stage in microseconds, measured with HIP events on the ctx stream. */
#define CHD_STAGE_INGEST 0   /* K1 cell assign + handover */
#define CHD_STAGE_INDEX 1    /* K2 cell index build */
#define CHD_STAGE_INTEREST 2 /* K3+K4 AOI query + diff */
#define CHD_STAGE_PLAN 3     /* K5a due plan + scan */
#define CHD_STAGE_EMIT 4     /* K5b fan-out emit */
#define CHD_N_STAGES 5
typedef struct {
    float stage_us[CHD_N_STAGES];
    float total_us;
    float emit_main_us; /* the emit stage's dominant kernel alone (k_fanout_emit_seg when the descriptor path runs — together with the
                           filtered descriptors' kernel on worlds that keep arrival offsets, history_depth —; else = stage_us[4]) */
    uint64_t n_records, n_record_upper_bound;
    uint32_t n_handovers, n_unsubs, n_pairs;
    uint32_t n_deferred_records; /* of n_records: written by the deferred-connection launch, not by the dominant emit kernel */
    uint64_t algorithmic_bytes; /* DESIGN.md §4 byte model for the last tick */
    /* chd_get_tick_history only (saturating at 2^32 - 1), both part of n_deferred_records: written by the filtered descriptors
     * (windows that needed a per-entity decision on arrival stamps / update histories) and by the walk of the exact update
     * buffers (history_depth: irregular channels, long catch-ups) */
    uint32_t n_filtered_records, n_deep_records;
    /* how the world's ticks are scheduled NOW (CHD_SCHED_*), and how many device-side gates have timed out on this world (each one
     * invalidated its tick — overflow bit 0x4000 — and turned CHD_SCHED_GATED off for good) */
    uint32_t schedule, gate_timeouts;
    /* chd_get_tick_history only: the tick's overflow mask and history_overflow count as chd_tick_out reports them for the LAST tick
     * — a host that enqueues several chd_tick_device calls back to back reads the earlier ticks' here */
    uint32_t overflow, history_overflow;
} chd_tick_stats;
#define CHD_SCHED_OVERLAP_INTEREST 1u /* the interest updates run on the second stream */
#define CHD_SCHED_GATED 2u            /* ... forked / joined by device-side flags (CHD_WORLD_GATED_OVERLAP asked for AND the streams were seen to run side by side) */
#define CHD_SCHED_PIPELINED 4u        /* CHD_WORLD_PIPELINE_TICKS in force */
#define CHD_SCHED_CELL_MAJOR 8u       /* the cell-major emit form (asked for, or chosen for cells of >= 1024 entities on a world without exact update buffers) */
#define CHD_SCHED_ARRIVAL_OFFSETS 16u /* history_depth > 0 on the descriptor path: sub-tick arrival offsets decide the windows that cut through a tick's arrivals */
/* depth > 0: record HIP events around the stages of the next ticks, keeping the
 * last `depth` ticks (<= 1024); 0 turns it off. */
int chd_set_profiling(chd_ctx *ctx, int depth);
/* What a profiled tick records.  CHD_PROF_STAGES (default): an event at every stage boundary (a timed event between two
 * short kernels idles the stream for a few microseconds: use it for stage breakdowns and latency runs).
 * CHD_PROF_RECORD_KERNEL: nothing but the pair around the dominant record-writing kernel (chd_tick_stats.emit_main_us;
 * stage_us and total_us read 0): for throughput runs, whose tick time the caller takes from its own clock.  The analogue
 * of not scraping the reference's per-channel `channel_tick_duration` gauge (channel.go:382-383) while load-testing. */
#define CHD_PROF_STAGES 0
#define CHD_PROF_RECORD_KERNEL 1
/* ... and that pair on every n-th tick only (2 <= n <= 1024; the ticks in between record nothing, their emit_main_us reads 0):
 * each of the two events idles the stream for ~7 us beside a kernel of ~140 (profiles/r04t_tick_timeline_*.csv), so a throughput
 * run that wants the kernel's duration AND an undisturbed tick time samples. */
#define CHD_PROF_RECORD_KERNEL_EVERY(n) (CHD_PROF_RECORD_KERNEL | ((n) << 8))
int chd_set_profiling_scope(chd_ctx *ctx, int scope);
/* Turn the tick pipelining of a CHD_WORLD_PIPELINE_TICKS world off (on = 0: serial schedule on the ctx stream, as without
 * the flag) and on again; CHD_E_STATE for a world created without the flag (or where it did not take effect). */
int chd_world_set_pipelining(chd_ctx *ctx, int on);
int chd_get_tick_stats(chd_ctx *ctx, chd_tick_stats *out);
/* Statistics of the last n ticks (out[0] = most recent), n <= 1024.  Counts come
 * from a device-side ring written by every tick; stage times need profiling.
 * Synchronises the stream. */
int chd_get_tick_history(chd_ctx *ctx, uint32_t n, chd_tick_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* CHD_SPATIAL_H */
