// chd_kernels.h — host-callable launchers of the gfx950 kernels.
// Every launcher enqueues on `st` and returns immediately.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chd_spatial.h"
#include "chd_device.h"

// ---- world state (all device pointers, SoA) ----
struct WorldDev {
    uint32_t N, S, capq;
    // entities
    uint32_t *chan_id;    // entity channel id
    uint32_t *cell;       // cell index of the last merged position (Notify's "old")
    uint32_t *member;     // cell whose entity map holds the entity
    uint32_t *eflags;     // EF_*
    uint32_t *sender;     // senderConnId of the entity's updates
    uint32_t *hist;       // bit j: an update arrived at tick (hist_tick - j)
    uint32_t *hist_tick;
    // spatial (cell) channels' own update history
    uint32_t *cell_hist, *cell_hist_tick, *cell_sender;
    // cell index (rebuilt every tick)
    uint32_t nblk;        // histogram blocks
    uint32_t *blk_cnt;    // [ncell*nblk + 1] counts -> exclusive scan (cell-major)
    uint4 *ce;            // [N] sorted by cell: {entity channel id, history aligned to this tick, sender, slot}
    uint32_t *cell_off;   // [ncell+1] cell c owns ce[cell_off[c], cell_off[c+1])
    // what the fan-out kernels read: cell c owns ce_view[cell_start[c], cell_end[c]).  Single GPU: ce_view = ce,
    // cell_start = cell_off, cell_end = cell_off + 1.  Region-sharded: ce_view = the all-gathered tables.
    const uint4 *ce_view;
    const uint32_t *cell_start, *cell_end;
    uint32_t *cell_tab;   // [2*ncell] storage of cell_start/cell_end in sharded mode
    // slot allocator of region-sharded worlds (entities migrate between ranks)
    uint32_t *free_stack; // [N]
    int32_t *free_top;    // number of free slots
    // subscribers
    uint32_t *conn_id;    // [S]
    uint32_t *sub_alive;  // [S]
    uint32_t *sub_tick;   // [S] tick of the subscriber's last interest update
    uint32_t *pair_cnt;   // [S]
    uint32_t *pair_cell;  // [S*capq] cell index
    uint32_t *pair_iv;    // [S*capq] FanOutIntervalMs
    int64_t *pair_last;   // [S*capq] lastFanOutTime
    uint32_t *pair_flags; // [S*capq] PF_*
    uint32_t *pair_rel;   // [S*capq] this tick: segment offset inside the connection's record range
    uint32_t *pair_nrec;  // [S*capq] this tick: records emitted for the subscription
    // fan-out outputs
    uint64_t *rec_ub;     // [S+1] upper bound per subscriber -> exclusive scan = base of its record range
    uint32_t *rec_cnt;    // [S] records emitted for the connection (sum of its pair_nrec)
    chd_fanout_rec *recs; uint64_t recs_cap;
    chd_handover_rec *handovers; uint32_t handovers_cap;
    uint32_t *unsub_sub, *unsub_cell; uint32_t unsub_cap;
    uint32_t *newsub_sub, *newsub_cell, *newsub_iv; uint32_t newsub_cap;
    int32_t *q_status;    // [S]
    uint32_t *counters;   // CTR_COUNT
    uint64_t *tot64;      // [64][16] hashed per-tick totals, one 128-B line per bucket: {records, subscriptions}
    uint64_t *tick_ring;  // [TICK_RING][8] per-tick totals written by the epilogue
};

// ---- stateless ----
void launch_get_channel_ids(hipStream_t st, DevGrid g, const double *x, const double *z,
                            uint32_t n, uint32_t *out);
void launch_notify_decide(hipStream_t st, DevGrid g, const double *ox, const double *oz,
                          const double *nx, const double *nz, uint32_t n, uint32_t *src,
                          uint32_t *dst, uint8_t *handover);
void launch_regions(hipStream_t st, DevGrid g, double *min_x, double *min_z, double *max_x,
                    double *max_z, uint32_t *channel_id, uint32_t *server_index);
void launch_adjacent(hipStream_t st, DevGrid g, const uint32_t *ids, uint32_t n, uint32_t *out,
                     uint32_t *counts);
// mode 0: CreateChannels cells (spatial.go:399-424); mode 1: border subs (:481-590).
// out[cap], *n_out (device), *err (device, set to 1 on GetChannelIdNoOffset error)
void launch_server_cells(hipStream_t st, DevGrid g, uint32_t server_index, int mode,
                         uint32_t *out, uint32_t cap, uint32_t *n_out, uint32_t *err);

// ---- world ----
void launch_spawn(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *idx,
                  const uint32_t *chan_id, const double *x, const double *z,
                  const uint32_t *flags, const uint32_t *sender, uint32_t cur_tick);
void launch_despawn(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *idx);
void launch_set_flags(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *idx,
                      const uint32_t *flags);
void launch_subs_add(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *slot,
                     const uint32_t *conn, int add);

// K1: cell assign + handover detect (+ update history)
void launch_ingest(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *idx,
                   const double *x, const double *z, const uint32_t *sender, uint32_t cur_tick);
void launch_cell_updates(hipStream_t st, DevGrid g, WorldDev w, uint32_t n,
                         const uint32_t *chan, const uint32_t *sender, uint32_t cur_tick);
// pull-mode ingest of region-sharded worlds: every live slot reads its position by entity channel id
void launch_ingest_by_channel(hipStream_t st, DevGrid g, WorldDev w, const double *x_by_chan,
                              const double *z_by_chan, const uint8_t *has_update, uint32_t n_chan,
                              uint32_t entity_id_start, uint32_t cur_tick);
// entities whose member cell belongs to another rank leave (state packed per destination, slot freed)
void launch_export(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                   chd_entity_state *send, uint32_t cap, uint32_t cur_tick);
void launch_import(hipStream_t st, WorldDev w, const chd_entity_state *recv, uint32_t world, uint32_t cap,
                   uint32_t cur_tick);
void launch_spawn_auto(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id,
                       const double *x, const double *z, const uint32_t *flags, const uint32_t *sender,
                       uint32_t cur_tick);
void launch_free_stack_init(hipStream_t st, WorldDev w);
// tables = world x table_bytes; each: N 16-byte entries, then ncell+1 offsets
void launch_cell_table(hipStream_t st, DevGrid g, WorldDev w, const void *tables, uint32_t world,
                       uint64_t table_bytes);
// K2: cell index build
void launch_index_build(hipStream_t st, DevGrid g, WorldDev w, uint32_t cur_tick);
// K3/K4: AOI query (+ interest diff when stateful)
struct AoiLimits {
    uint32_t maxax;   // samples per lattice axis
    uint32_t winmax;  // cells in the per-query table
    uint32_t maxdim;  // max(window width, window height) <= max(cols, rows)
};
void launch_aoi_stateless(hipStream_t st, DevGrid g, AoiLimits lim, const chd_aoi_query *q,
                          uint32_t nq, const double *spot_x, const double *spot_z,
                          const uint32_t *spot_dist, uint32_t stride, uint32_t *cells,
                          uint32_t *dists, uint32_t *ivs, uint32_t *counts, int32_t *status);
void launch_aoi_interest(hipStream_t st, DevGrid g, AoiLimits lim, WorldDev w,
                         const chd_aoi_query *q, uint32_t nq, const uint32_t *q_sub,
                         const double *spot_x, const double *spot_z, const uint32_t *spot_dist,
                         int64_t now_ns, uint32_t cur_tick);
size_t aoi_lds_bytes(AoiLimits lim, uint32_t capq);
// compaction of the fixed-stride stateless output into CSR
void launch_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n);  // exclusive, out[n]=total
void launch_scan_u32_inplace(hipStream_t st, uint32_t *data, uint32_t n);             // data[n] = total
void launch_scan_u64_inplace(hipStream_t st, uint64_t *data, uint32_t n);
void launch_csr_gather(hipStream_t st, uint32_t nq, uint32_t stride, const uint32_t *counts,
                       const uint32_t *offsets, const uint32_t *cells, const uint32_t *dists,
                       const uint32_t *ivs, uint32_t *out_ids, uint32_t *out_dists,
                       uint32_t *out_ivs, uint32_t cap, uint32_t id_start);
// K5: fan-out
void launch_fanout_plan(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring);
void launch_fanout_emit(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring);
#define TICK_RING 1024
void launch_tick_epilogue(hipStream_t st, WorldDev w, uint32_t slot);
