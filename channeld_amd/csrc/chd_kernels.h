// chd_kernels.h — host-callable launchers of the gfx950 kernels.
// Every launcher enqueues on `st` and returns immediately.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chd_spatial.h"
#include "chd_device.h"

// due subscriptions of one cell-major work item (active cell x 256 connection slots), written by
// k_fanout_items and copied to LDS by the emit kernel's loader wave
struct WsItemG {
    uint32_t pi[256];     // connection slot * capq + subscription index
    uint32_t conn[256];
    uint32_t flags[256];  // WSF_*
    uint32_t out16[256];  // segment start in the record buffer, in units of 16 records (128-byte lines)
    uint32_t wm[4][256];  // the first four non-empty fan-out windows as masks over the tick ring
    uint32_t iv[256];     // generic path only (more than four non-empty windows): interval and
    int64_t L[256];       //   lastFanOutTime before this tick
    uint32_t ndue, nsub, c, start, end, ch_hist, ch_hprev, ch_sender, ch_sprev, _pad[7];
};
#define WSF_FIRST 1u        // first fan-out of the subscription: full state
#define WSF_SKIP_SELF 2u
#define WSF_GENERIC 4u      // more than four non-empty windows: walk them from (L, iv)
#define WSF_NWIN_SHIFT 8    // bits 8..10: number of stored windows
#define WSF_OWN_SHIFT 16    // bit 16+j: the spatial channel's own update passes window j

// One fan-out window of a filtered descriptor: an entity's message passes when one of its buffered updates lies inside the
// window — an update of a ring slot the window covers whole (full), or of one of the (at most two) slots a window edge cuts
// through, with its offset inside that slot's bounds (slot index 0xFF = unused; offsets are t[slot] - arrival).
struct FiltWin {
    uint32_t full;       // mask over the tick ring: every update of these slots lies inside the window
    uint32_t slots;      // slot_a | slot_b << 8
    uint32_t a_lo, a_hi; // slot_a: a_lo <= offset <= a_hi (lo > hi: unused)
    uint32_t b_lo, b_hi; // slot_b (only with slot_a)
};
#define CHD_FILT_WINS 8u  // fan-out windows of one filtered descriptor (more non-empty ones: the exact update buffers decide)

// ---- world state (all device pointers, SoA) ----
struct WorldDev {
    uint32_t N, S, capq;
    // entities
    uint32_t *chan_id;    // entity channel id
    uint32_t *cell;       // cell index of the last merged position (Notify's "old")
    uint32_t *member;     // cell whose entity map holds the entity
    uint32_t *eflags;     // EF_*
    uint32_t *sender;     // senderConnId of the entity's updates
    uint32_t *hist;       // bit j: an update of `sender` arrived at tick (hist_tick - j)
    uint32_t *hist_tick;
    uint32_t *sender_prev, *hist_prev;  // the previous sender's updates still inside the history
    // THE UPDATE LOG BY CHANNEL (region-sharded worlds with exact update buffers, chd_world_cfg.shard_channels; log_on): everything
    // push_update writes — sender / hist / hist_tick / sender_prev / hist_prev above, eoff, deep_*, irr_tick, ent_max_iv below —
    // is then indexed by the entity CHANNEL, u = chan_id - log_eid0 < log_n, not by the entity slot, and every rank keeps it for
    // every channel of the whole world (k_log_push: the ranks are fed the same update stream by channel id).  An entity that
    // changes ranks, a neighbour's ghost entry, a window older than the entity's stay on this rank: the log is already here.
    // log_ix() maps a slot to its log.  log_cell[u] = cell of the channel's last merged position (CHD_INVALID: none / out of the
    // world), log_alive[u] = the channel exists (chd_shard_log_spawn).  sh_arrival_by_chan: chd_shard_set_update_arrivals.
    uint32_t log_on, log_n, log_eid0;
    // ce_by_chan (shard_channels != 0): ce_slot[pos] — where a cell-table entry's exact buffer (log_on) and its wire payloads
    // (CHD_WORLD_WIRE) are found — holds the channel index, for own and for ghost entries alike
    uint32_t ce_by_chan;
    uint32_t *log_cell, *log_alive;
    const int64_t *sh_arrival_by_chan;
    uint32_t sh_arrival_n;
    uint32_t *upd_mark;   // [N] (tick, round) of the slot's last update: a second one in the same round is a caller error (OVF_DUPLICATE)
    uint32_t *q_mark;     // [S] tick of the slot's last interest update, likewise
    // handover groups (entity.go:58-244, FlatEntityGroupController): entities that cross cells together.  grp_of[i] =
    // index of entity i's group (CHD_INVALID: a group of one); group k's members are grp_mem[grp_off[k] .. grp_off[k+1]);
    // grp_locked[k] = members that are alive and locked (a locked member aborts the whole handover, entity.go:197-224)
    // grp_exact (chd_world_set_handover_lists): the lists ARE GetHandoverEntities' results as the host's group controller
    // evaluated them — an empty list = no handover (a locked member, or an emptied group), the notifier moves only if it is
    // a member; grp_locked is then unused.
    uint32_t *grp_of, *grp_off, *grp_mem, *grp_locked;
    uint32_t n_groups;
    uint32_t grp_exact;
    // region-sharded worlds (chd_shard_set_handover_lists): the same lists keyed by ENTITY CHANNEL ID — slots are the library's
    // there and change when an entity changes ranks.  sh_list_of[k] = list of channel (sh_eid0 + k) (CHD_NO_HANDOVER_LIST: the
    // entity itself), list l = channel ids sh_list_mem[sh_list_off[l] .. sh_list_off[l+1]); sh_slot_of[k] = the slot that holds
    // the channel on THIS rank (CHD_INVALID: it lives elsewhere), kept by spawn / import / export.  The members that move with a
    // handover are those in src's entity map — by construction on the rank that owns src, i.e. the notifier's.
    const uint32_t *sh_sender_by_chan;  // chd_shard_set_update_senders: senderConnId of this tick's update of channel (eid0 + k); nullptr: the entity's own
    uint32_t sh_sender_n;
    uint32_t sh_nchan, sh_eid0, sh_nlists;
    uint32_t *sh_slot_of, *sh_list_of, *sh_list_off, *sh_list_mem;
    // spatial (cell) channels' own update history
    uint32_t *cell_hist, *cell_hist_tick, *cell_sender, *cell_hist_prev, *cell_sender_prev;
    // Exact update buffers (chd_world_cfg.history_depth = deep_depth > 0): ChannelData.updateMsgBuffer element for element,
    // a ring of {arrivalTime, senderConnId} per entity channel (deep_*) and per spatial channel (cdeep_*): n = updates ever
    // pushed (write index n % D), len = elements held (the newest), drop = arrival of the newest element the ring had to
    // overwrite although the reference would still hold it (INT64_MIN: none).  irr_tick = 1 + the last tick at which the
    // channel took an update the tick-ring masks cannot represent (an arrival stamp off the tick's own, a third sender);
    // cell_irr[c] = this tick some channel of cell c is like that inside the 32-tick mask horizon: its subscriptions go to
    // k_fanout_emit_deep.  cell_max_iv / ent_max_iv = ChannelData.maxFanOutIntervalMs per CHANNEL (only grows; the eviction
    // test of data.go:165-171): raised for spatial channel c when a subscription to it is CREATED (subscription.go:83-86 — the
    // branch that merges options into an existing subscription does not touch it); for an entity channel — whose subscribers
    // are, in the tick model, those of the cells that hold it — at every update, before the eviction test, to the maxima of the
    // cell of its last merged position and of the new one (push_update).
    uint32_t deep_depth;
    int64_t *deep_a, *cdeep_a, *deep_drop, *cdeep_drop;
    uint32_t *deep_s, *cdeep_s, *deep_n, *cdeep_n, *deep_len, *cdeep_len, *irr_tick, *cell_irr_tick;
    uint32_t *cell_irr;
    uint32_t *cell_max_iv;  // [2 * ncell]: [c] what this tick's updates are buffered under, [ncell + c] raised by this tick's interest updates (folded into [c] by the epilogue: the interest updates may then run beside the ingest)
    uint32_t *ent_max_iv;   // [N]
    uint32_t *conn_deep;  // [S] this tick: the connection has PF_DEEP subscriptions
    // SUB-TICK ARRIVAL OFFSETS (off_on: worlds with exact update buffers on the descriptor path).  The reference stamps an
    // update when it is ENQUEUED (Channel.PutMessage: arrivalTime = ch.GetTime(), channel.go:296-310), so a tick's updates
    // carry stamps anywhere inside (previous tick, this tick].  For such REGULAR updates — one per channel and tick, stamp
    // inside the tick's own interval, fewer than 2^32 ns before the tick — the tick-ring mask bit of the update keeps its
    // meaning "an update arrived WITH tick (cur - j)" and the exact stamp is t[j] - offset, offset kept per ring slot for the
    // newest CHD_OFF_SLOTS ticks: eoff (entity slots, aligned to hist_tick like the masks), ce_off (the cell-sorted columns the
    // filtering record kernel streams, aligned to this tick: slot j at j * off_stride), cell_orng (per cell and slot the
    // {min, max} offset over its entities' updates — the plan decides "every update of the slot inside / outside this window"
    // from it and sends only cells that a window edge cuts through to the per-entity compare), cell_ooff (the spatial
    // channels' own updates).  Anything else (a second update of a channel in one tick, a stamp at or before the previous
    // tick's, a third sender) marks the channel irregular as before and its cell's subscriptions walk the element buffers.
    uint32_t off_on;
    uint32_t off_stride;
    int64_t prev_ns;      // stamp of the previous tick (-1 before the first): set per tick by the host
    uint4 *eoff;          // [2 * N]
    uint32_t *ce_off;     // [CHD_OFF_SLOTS * off_stride]
    uint2 *cell_orng;     // [ncell * CHD_OFF_SLOTS]
    uint4 *cell_ooff;     // [2 * ncell]
    // FILTERED descriptors (k_fanout_plan_seg -> k_fanout_emit_filt): due subscriptions with a window that a per-entity compare
    // must decide (an edge cuts through a tick's arrivals, or some entity has no update in it).  Row s holds n_filt[s] entries:
    // filt_desc {segment offset, column start, entries, windows | own-update bits << 8}, filt_desc2 / filt_ln as seg_desc2 /
    // seg_ln; the windows' tests are indexed by the SUBSCRIPTION, filt_win[(s * capq + p) * CHD_FILT_WINS + window] (FiltWin): the plan
    // writes them as it walks the windows, whatever the subscription turns out to be
    uint32_t *n_filt;     // [S]
    uint4 *filt_desc, *filt_desc2;  // [S * capq]
    int64_t *filt_ln;     // [S * capq]
    struct FiltWin *filt_win;  // [S * capq * CHD_FILT_WINS]
    // ... and, for the cell-major filtered kernel (k_fanout_emit_filt_cm: a cell's columns staged once in LDS for all of its
    // filtered descriptors), the descriptors listed per CELL, as self-contained 32-byte entries {segment offset, entries, windows
    // | own bits << 8, subscription index p; connection slot s, -, -, -}: cell_flist[(c * S + i) * 2 ..], i < cell_fcnt[c * 32]
    // (one counter per 128-byte line; the plan appends, the tick epilogue clears); filt_items = the kernel's work list {cell,
    // first list entry, descriptors | tile entries << 8, column start} in chunks of 64 descriptors, n in filt_nitems (k_filt_items)
    uint32_t fcm_on;
    uint32_t *cell_sorted;  // [ncell] (fcm_on, else nullptr) the cell's entries are in the order of this tick's arrival offsets (k_cell_sort0): a window inside the tick's own arrivals is a run of the column
    uint32_t *cell_fcnt;
    uint4 *cell_flist;
    uint4 *filt_items;
    uint32_t *filt_nitems;
    uint32_t filt_target;   // work items per launch of k_fanout_emit_filt_cm that filt_items_block aims for (0: 1024; CHD_FILT_ITEMS_TARGET)
    uint32_t seg_only;      // CHD_WORLD_SEGMENTS_ONLY: on descriptor-path ticks the record kernel of the simple descriptors does not run
    uint32_t late_tot;      // pipelined ticks on off_on worlds: k_fanout_emit_filt_cm counts into tot64[..][8], k_filt_fold adds that to the tick's row
    // cell index (rebuilt every tick)
    uint32_t nblk;        // histogram blocks
    uint32_t *blk_cnt;    // [ncell*nblk + 1] counts -> exclusive scan (cell-major)
    uint4 *ce;            // [N] sorted by cell: {entity channel id, history of `sender` aligned to this tick,
                          //  sender, history of the previous sender}
    uint32_t *ce_sprev;   // [N] previous sender (read only where its history intersects a window)
    uint2 *ce8;           // [N] compact entries {entity channel id, history of any sender} for single-sender cells
    uint32_t *cell_usender;           // [ncell] the one sender of the cell's buffered updates, or CHD_NONUNIFORM
    uint32_t *cell_smin, *cell_smax;  // [ncell] range of the sender ids behind the cell's buffered updates (a connection outside it sent none of them)
    uint32_t *blk_smin, *blk_smax, *blk_hand;  // [ncell*nblk] per-block sender range / AND of histories (index build intermediates)
    uint32_t *cell_hand;              // [ncell] AND of the histories of the cell's entities (aligned to this tick)
    uint32_t *ce_chan;                // [N + 4] the entity channel ids alone, cell-sorted: what an all-pass window copies
    // WINDOW COLUMNS (descriptor path of partially updating worlds): behind the full column array, at (k + 1) * wcol_stride for
    // k < CHD_WCOLS, the cell-sorted channel ids of the entities that have an update inside the ticks of wcol_mask(k) — runs
    // of 1..3 ticks starting at the newest tick, the one before or the one before that: the masks a subscription served every
    // interval produces (its windows are closed intervals: a stamp on an edge lies in two of them; an interest update that
    // changes the damped interval leaves a window that starts at an older stamp) — a subsequence of the cell's column, stored
    // from the cell's own start; cell_wcnt[k * ncell + c] = its length.  A fan-out window with exactly that mask is then a
    // plain copy of that column, as an all-pass window is of the full one.  Built by k_window_columns in the ticks where some
    // entity skipped an update (wcol_on); wcol_stride == 0: not available (region-sharded worlds).
    uint32_t wcol_stride, wcol_on;
    uint32_t *cell_wcnt;
    // wire worlds whose streams are built from the descriptors (WireDev::img_on): the record kernel writes no position words
    // (seg_no_pos; k_wire_layout_img derives them for the few subscriptions it still walks record by record), the window
    // columns also carry the entities' SLOTS (wcol_slot: same indexing as ce_chan's columns 1..9), and the plan leaves every
    // subscription's first descriptor index (pair_desc: index | count << 28, ~0 = no descriptor this tick)
    uint32_t seg_no_pos;
    uint32_t *wcol_slot;
    uint32_t *pair_desc;
    const uint32_t *ce_chan_view;     // (views: always the arrays above since the halo exchange appends the neighbours' entries to them)
    const uint2 *ce8_view;
    uint32_t *cell_off;   // [ncell+1] cell c owns ce[cell_off[c], cell_off[c+1])
    uint32_t *cell_tot;   // [ncell] entities per cell (intermediate of the index build)
    // what the fan-out kernels read: cell c owns ce_view[cell_start[c], cell_end[c]).  Single GPU: ce_view = ce,
    // cell_start = cell_off, cell_end = cell_off + 1.  Region-sharded: the same arrays with the neighbours' border entities
    // appended from index N on (ghosts), cell_start / cell_end = cell_tab.
    const uint4 *ce_view;
    const uint32_t *ce_sprev_view;
    uint32_t ce_sprev_stride;  // sharded: entries of rank o start at o*stride16 (16-B units), its sprev at +N entries
    const uint32_t *cell_start, *cell_end;
    uint32_t *cell_tab;   // [2*ncell] storage of cell_start/cell_end in sharded mode
    uint32_t *cell_cov;   // sharded mode (else nullptr): [ncell] 1 = the cell's table is on this rank (own region or a received halo)
    uint32_t ghost_cap;   // sharded mode: room behind the N own entries of ce / ce_sprev / ce8 / ce_chan for the neighbours' border entities
    // slot allocator of region-sharded worlds (entities migrate between ranks)
    uint32_t *free_stack; // [N]
    int32_t *free_top;    // number of free slots
    chd_entity_state *limbo;  // [2][N] immigrants that found no free slot, by tick parity (allocated with the halo layout)
    uint32_t *limbo_n;        // [2]
    uint32_t *mig_gmax;       // [4] by tick & 3: the largest emigrant segment count of that tick's exchange, over ALL ranks
    // subscribers
    unsigned char *aoi_scratch;  // [S * aoi_scratch_bytes] the interest kernel's long-lattice work areas, per subscriber slot
    uint32_t *conn_id;    // [S]
    uint32_t *sub_alive;  // [S]
    uint32_t *sub_tick;   // [S] tick of the subscriber's last interest update
    uint32_t *pair_cnt;   // [S]
    uint32_t *pair_cell;  // [S*capq] cell index
    uint32_t *pair_iv;    // [S*capq] FanOutIntervalMs
    int64_t *pair_last;   // [S*capq] lastFanOutTime
    uint32_t *pair_flags; // [S*capq] PF_*
    // cell-major fan-out (grids up to 4096 cells): interest bitmap per connection and subscriber count per cell
    uint32_t wb;                    // 64-bit words per bitmap row (0: grid larger than 4096 cells, no bitmap)
    uint32_t cm_emit;               // cell-major emit selected
    uint32_t one_wave_emit;         // connection-major emit: one wave per connection also below 4096 connections
    uint32_t seg_off;               // this tick: not the descriptor path (set per tick by the host, see tick_locked)
    unsigned long long *sub_bits;   // [S*wb]
    uint32_t *cell_ref;             // [ncell] live subscriptions of the cell
    uint32_t *active_cells;         // [ncell] cells with cell_ref > 0 (compacted every tick)
    uint32_t *n_active;             // [1]
    uint32_t emit_grid;             // persistent grid of the cell-major emit kernel (workgroups)
    uint32_t seg_waves;             // persistent waves of k_fanout_emit_seg (connection-major descriptor emit)
    uint32_t emit_act_t1, emit_act_t2;  // records per connection from which a tick runs seg_waves / 1.5 x seg_waves active waves (below: 2 x; CHD_EMIT_ACTIVE_THRESHOLDS="t1,t2")
    uint32_t emit_waves;            // workgroups of a k_fanout_emit_seg launch (2 x seg_waves; CHD_EMIT_WAVES_PER_CU: seg_waves); how many of them are ACTIVE in a tick — seg_waves, 1.5 x or 2 x — k_fanout_scan writes into emit_ticket[32 b + 3]; seg_waves also sizes the filtered kernel's and the wire copy's grids
    uint32_t *emit_ticket;          // [8 x 32] k_fanout_emit_seg's ticket counters, one 128-byte line each (zeroed by k_fanout_plan_seg; word 0 then set to the bank's first free ticket and word 3 to the tick's active waves by k_fanout_scan; word 1: k_fanout_emit_filt's)
    WsItemG *items;                 // [ncell * ceil(S/256)]
    uint32_t *conn_defer; // [S] this tick: the connection has subscriptions left to the deferred emit launch
    // The tick's TAIL LISTS (k_fanout_scan, the single-workgroup pass behind the plan): the connections with deferred subscriptions
    // (defer_list) and with PF_DEEP ones (deep_list), compacted in slot order, and the first connection without room for its records
    // (rec_ub[s + 1] > recs_cap; the exclusive prefix is monotonic, so every later one has none either).  The plan COMMITS a
    // subscription's new fan-out state itself and keeps the old one in seg_ln / seg_desc2.z (filt_ln / filt_desc2.z); the deferred
    // launch walks defer_list and puts the old state back for the connections from tail_ctl[TC_SCAP] on — a launch over a few
    // workgroups instead of one per connection slot (9.7 us for ~0 deferred records at config B).
    uint32_t *defer_list, *deep_list;  // [S]
    uint32_t *tail_ctl;   // [16] TC_*
    // descriptor-driven emit (k_fanout_plan_seg -> k_fanout_emit_seg): per connection the first n_simple[s] entries of its
    // row [s*capq, ...) describe the due subscriptions whose every window is a plain copy of the cell's channel column
    uint32_t *n_simple;   // [S]
    uint4 *seg_desc;      // [S*capq] {segment offset in the connection's range, column start, entries, SD_* | windows}
    uint4 *seg_desc2;     // [S*capq] {cell index, index of the subscription in the connection's list, pair_flags after this tick, -}
    uint4 *seg_wm;        // [S*capq] CHD_WORLD_UPDATE_MASKS only: the descriptor's (up to four) window masks = its records' merged-update masks
    int64_t *seg_ln;      // [S*capq] lastFanOutTime after this tick
    uint32_t *pair_rel;   // [S*capq] this tick: segment offset inside the connection's record range
    uint32_t *pair_nrec;  // [S*capq] this tick: records emitted for the subscription
    // fan-out outputs
    uint64_t *rec_ub;     // [S+1] upper bound per subscriber -> exclusive scan = base of its record range
    uint32_t *rec_cnt;    // [S] records emitted for the connection (sum of its pair_nrec)
    chd_fanout_rec *recs; uint64_t recs_cap;
    uint32_t *rec_mask;   // CHD_WORLD_UPDATE_MASKS only (else nullptr): per record, the buffered updates (ring slots) the message merges
    uint32_t *rec_pos;    // wire mode only (else nullptr): cell-table position of each record's entity, or CHD_POS_CELL | cell
    uint32_t *ce_slot;    // [N] entity slot of each cell-table entry (wire mode: payload lookup)
    chd_handover_rec *handovers; uint32_t handovers_cap;
    // per handover record (worlds that plan recipients, else nullptr): bit q = entity q of the handover's entity list — the
    // notifier alone, or the live members of its list / group in list order — was in src's entity map and moved with it
    // (spatial.go:703-736); the others stayed where they were.  What chd_handover_recipients_ex needs to say, per member, whose
    // entity channel a destination connection was already subscribed to (the subscribers of the cell that HELD the member).
    uint32_t *ho_moved;
    // chd_world_set_server_connections (else nullptr): ConnectionId of spatial server k — the owner of its cells' spatial channels
    // and, in the tick model, of the entity channels those cells hold (what SubscribeToChannel's DataAccess in the handover loop
    // is derived from, spatial.go:812-817)
    const uint32_t *server_conn;
    uint32_t n_server_conn;
    // Unsub / new-sub lists are kept in CHD_LIST_BANKS banks (bank = subscriber slot & 63, list_bank_cap entries
    // each, one tail counter per bank on its own 128-B line): a single list tail is a same-address atomic for
    // every interest update in the tick, and those serialise at L2 (~5 ns each, 22 us per tick at 10K queries).
    // chd_tick_fetch packs the banks into the dense lists the ABI returns.
    uint32_t *unsub_sub, *unsub_cell; uint32_t unsub_cap;
    uint32_t *newsub_sub, *newsub_cell, *newsub_iv; uint32_t newsub_cap;
    uint32_t *list_ctr;     // [2][CHD_LIST_BANKS][32] tails of the tick being built: row 0 unsubs, row 1 new subs
    uint32_t *list_bank_n;  // [2][CHD_LIST_BANKS] tails of the last finished tick (written by the epilogue)
    uint32_t list_bank_cap;
    int32_t *q_status;    // [S]
    uint32_t *counters;   // CTR_COUNT
    unsigned long long *gate_fail;  // CHD_WORLD_GATED_OVERLAP (else nullptr): gates that timed out, ever (sticky: the per-tick counters are cleared by every epilogue)
    uint64_t *tot64;      // [64][16] hashed per-tick totals, one 128-B line per bucket: {records, subscriptions}
    uint64_t *tick_ring;  // [TICK_RING][8] per-tick totals written by the epilogue
};

// where entity slot i's update log lives (WorldDev::log_on)
__device__ __forceinline__ uint32_t log_ix(const WorldDev &w, uint32_t i) { return w.log_on ? w.chan_id[i] - w.log_eid0 : i; }
// what ce_slot holds for entity slot i (WorldDev::ce_by_chan)
__device__ __forceinline__ uint32_t ce_ix(const WorldDev &w, uint32_t i) { return w.ce_by_chan ? w.chan_id[i] - w.log_eid0 : i; }

// a gate's spin bound tripped (OVF_GATE): this tick's mask, and the world's sticky count (the host turns the gates off when it sees it)
__device__ __forceinline__ void gate_timed_out(const WorldDev &w) {
    atomicOr(&w.counters[CTR_OVERFLOW], OVF_GATE);
    if (w.gate_fail) atomicAdd(w.gate_fail, 1ull);
}

// ChannelData.OnUpdate (data.go:159-164) on the entity channel's bit-mask update
// buffer.  Every buffered update keeps its senderConnId (tickData compares it with
// the subscriber for SkipSelfUpdateFanOut, data.go:242-245): bits are kept per sender
// for the current and the previous sender of the channel; a third sender inside the
// 32-tick window folds the oldest bits into the previous one (counted, never silent).
// One element into an exact update buffer (data.go:159-172): ring[(n % D)] = {arrival, sender}; beyond
// MaxUpdateMsgBufferSize elements the oldest goes if it is older than maxFanOutIntervalMs, one per push, as the reference
// does; a ring that is full of elements the reference would still hold overwrites its oldest and remembers that arrival
// (a window that reaches it is reported as history_overflow, never silently short).
__device__ __forceinline__ void deep_push(int64_t *__restrict__ A, uint32_t *__restrict__ S, uint32_t D, uint32_t &n, uint32_t &len,
                                          int64_t &drop, int64_t arrival, uint32_t sender, uint32_t max_iv_ms) {
    if (len == D) {  // no room: the oldest element leaves whatever its age
        drop = A[(n - len) % D];
        len--;
    }
    A[n % D] = arrival;
    S[n % D] = sender;
    n++;
    len++;
    if (len > CHD_MAX_UPDATE_BUFFER && A[(n - len) % D] + (int64_t)max_iv_ms * 1000000 < arrival) len--;
}

// a channel's offsets of the newest CHD_OFF_SLOTS ticks, moved on by `age` ticks: slot j takes what slot j - age held
__device__ __forceinline__ void off_shift(uint32_t (&o)[CHD_OFF_SLOTS], uint32_t age) {
    if (age == 0) return;
    uint32_t r[CHD_OFF_SLOTS];
#pragma unroll
    for (int j = 0; j < (int)CHD_OFF_SLOTS; j++) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < j; k++) v = (age == (uint32_t)(j - k)) ? o[k] : v;
        r[j] = v;
    }
#pragma unroll
    for (int j = 0; j < (int)CHD_OFF_SLOTS; j++) o[j] = r[j];
}

// Every word of channel i's update log that push_update reads, requested TOGETHER and unconditionally (the caller issues this with its
// other loads, before any test: the ingest launches are one short workgroup per CU and as long as their chain of dependent round
// trips — with each of these words behind "alive?", "another sender?", "exact buffers?" that chain was six trips long).
struct UpdPre {
    uint32_t hist_tick, hist, hist_prev, sender, sender_prev;
    uint4 oa, ob;                    // eoff (off_on)
    uint32_t deep_n, deep_len, miv;  // exact buffers (deep_depth)
    int64_t deep_drop;
};
__device__ __forceinline__ UpdPre upd_prefetch(const WorldDev &w, uint32_t i) {
    UpdPre p;
    p.hist_tick = w.hist_tick[i]; p.hist = w.hist[i]; p.hist_prev = w.hist_prev[i]; p.sender = w.sender[i]; p.sender_prev = w.sender_prev[i];
    // (not kept by this world: any readable words, unused — a load behind a uniform test would still be awaited at the join)
    const uint4 *ep = w.off_on ? w.eoff : (const uint4 *)(const void *)w.hist;
    const size_t ei = w.off_on ? 2 * (size_t)i : 0;
    p.oa = ep[ei]; p.ob = ep[ei + 1];
    const uint32_t *dn = w.deep_depth ? w.deep_n : w.hist, *dl = w.deep_depth ? w.deep_len : w.hist, *dm = w.deep_depth ? w.ent_max_iv : w.hist;
    const int64_t *dd = w.deep_depth ? w.deep_drop : (const int64_t *)(const void *)w.hist;
    const uint32_t di = w.deep_depth ? i : 0u;
    p.deep_n = dn[di]; p.deep_len = dl[di]; p.miv = dm[di]; p.deep_drop = dd[di];
    return p;
}

// c_old / c_new: the cells of the entity's last merged position and of this update's (CHD_INVALID: out of the world)
__device__ __forceinline__ void push_update_pre(const WorldDev &w, uint32_t i, const UpdPre &P, uint32_t snd, uint32_t cur_tick, uint32_t c_old, uint32_t c_new,
                                                int64_t arrival = 0, int64_t now = 0) {
    const uint32_t age = cur_tick - P.hist_tick;
    uint32_t h = (age >= CHD_HIST_BITS) ? 0u : (P.hist << age);
    uint32_t hp = (age >= CHD_HIST_BITS) ? 0u : (P.hist_prev << age);
    const uint32_t cur = P.sender;
    bool irregular = arrival != now;  // (the masks stand for "arrived with the tick's own stamp")
    if (w.off_on) {
        // ... or, with the per-slot offsets, "arrived inside the tick's own interval": (previous tick, this tick], one per tick
        const uint64_t off = (uint64_t)(now - arrival);
        uint32_t o[CHD_OFF_SLOTS] = {P.oa.x, P.oa.y, P.oa.z, P.oa.w, P.ob.x, P.ob.y, P.ob.z, P.ob.w};
        off_shift(o, age);
        irregular = !(arrival > w.prev_ns && arrival <= now && off <= 0xFFFFFFFEull);  // (0xFFFFFFFF stands for "no update" in the staged columns)
        if (age == 0 && ((h | hp) & 1u) && o[0] != (uint32_t)off) irregular = true;  // a second update in this tick, another stamp
        o[0] = (uint32_t)off;
        w.eoff[2 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
        w.eoff[2 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    if (snd != cur) {
        const uint32_t prev = P.sender_prev;
        if (snd == prev) {  // the previous sender is back: the two histories swap roles
            const uint32_t t = h;
            h = hp;
            hp = t;
            w.sender_prev[i] = cur;
        } else {
            if (h != 0) {
                if (hp != 0 && prev != cur) {  // a third sender inside the mask horizon
                    if (w.deep_depth) irregular = true;
                    else atomicAdd(&w.counters[CTR_SENDER_OVERFLOW], 1u);
                }
                hp |= h;
                w.sender_prev[i] = cur;
            }
            h = 0;
        }
        w.sender[i] = snd;
    }
    w.hist[i] = h | 1u;
    w.hist_prev[i] = hp;
    w.hist_tick[i] = cur_tick;
    if (w.deep_depth) {
        uint32_t n = P.deep_n, len = P.deep_len;
        int64_t drop = P.deep_drop;
        const size_t at = (size_t)i * w.deep_depth;
        // the entity channel's maxFanOutIntervalMs (WorldDev::ent_max_iv): the subscribers of the cells that hold it are its own
        uint32_t miv = P.miv;
        const uint32_t m_old = c_old != CHD_INVALID ? w.cell_max_iv[c_old] : 0u, m_new = c_new != CHD_INVALID ? w.cell_max_iv[c_new] : 0u;
        if (m_old > miv || m_new > miv) { miv = max(miv, max(m_old, m_new)); w.ent_max_iv[i] = miv; }
        deep_push(w.deep_a + at, w.deep_s + at, w.deep_depth, n, len, drop, arrival, snd, miv);
        w.deep_n[i] = n;
        w.deep_len[i] = len;
        w.deep_drop[i] = drop;
        if (irregular) w.irr_tick[i] = cur_tick + 1u;
    }
}
// (one update on its own: the spawn / by-channel paths)
__device__ __forceinline__ void push_update(const WorldDev &w, uint32_t i, uint32_t snd, uint32_t cur_tick, uint32_t c_old, uint32_t c_new,
                                            int64_t arrival = 0, int64_t now = 0) {
    const UpdPre P = upd_prefetch(w, i);
    push_update_pre(w, i, P, snd, cur_tick, c_old, c_new, arrival, now);
}

// a channel's sub-tick arrival offsets (WorldDev::off_on; u = its log index), aligned to this tick, into the cell-sorted columns
__device__ __forceinline__ void scatter_offsets(const WorldDev &w, uint32_t u, uint32_t pos, uint32_t age) {
    const uint4 a = w.eoff[2 * (size_t)u], b = w.eoff[2 * (size_t)u + 1];
    uint32_t o[CHD_OFF_SLOTS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    off_shift(o, age);
#pragma unroll
    for (uint32_t j = 0; j < CHD_OFF_SLOTS; j++) w.ce_off[(size_t)j * w.off_stride + pos] = o[j];
}

// Is connection slot s subscribed to cell c?  Interest bitmap where it exists, else a binary search of
// the connection's cell-sorted subscription list.
__device__ __forceinline__ bool is_subscribed(const WorldDev &w, uint32_t s, uint32_t c) {
    if (w.wb) return (w.sub_bits[(size_t)s * w.wb + (c >> 6)] >> (c & 63u)) & 1ull;
    const uint32_t *cells = w.pair_cell + (size_t)s * w.capq;
    uint32_t lo = 0, hi = w.pair_cnt[s];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = cells[mid];
        if (v == c) return true;
        if (v < c) lo = mid + 1; else hi = mid;
    }
    return false;
}

// previous sender of the entry at position gpos of the (possibly gathered) cell tables
__device__ __forceinline__ uint32_t load_sprev(const WorldDev &w, uint32_t gpos) {
    if (w.ce_sprev_stride == 0) return w.ce_sprev_view[gpos];
    const uint32_t o = gpos / w.ce_sprev_stride, l = gpos % w.ce_sprev_stride;
    return ((const uint32_t *)(w.ce_view + (size_t)o * w.ce_sprev_stride + w.N))[l];
}

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

void launch_subs_set_options(hipStream_t st, DevGrid g, WorldDev w, const chd_sub_options *opts, const uint32_t *order,
                             const uint32_t *grp_off, uint32_t n_groups, int64_t now_ns, uint8_t *should_send, int32_t *status);
void launch_subs_get_options(hipStream_t st, WorldDev w, uint32_t s, uint8_t *access, uint8_t *skip_self);

void launch_group_locks(hipStream_t st, WorldDev w);
// K1: cell assign + handover detect (+ update history)
void launch_ingest(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *idx,
                   const double *x, const double *z, const uint32_t *sender, uint32_t cur_tick,
                   const int64_t *arrival = nullptr, int64_t now_ns = 0, uint32_t round = 0);
void launch_cell_updates(hipStream_t st, DevGrid g, WorldDev w, uint32_t n,
                         const uint32_t *chan, const uint32_t *sender, uint32_t cur_tick,
                         const int64_t *arrival = nullptr, int64_t now_ns = 0);
// pull-mode ingest of region-sharded worlds: every live slot reads its position by entity channel id
void launch_ingest_by_channel(hipStream_t st, DevGrid g, WorldDev w, const double *x_by_chan, const double *z_by_chan,
                              const uint8_t *has_update, uint32_t n_chan, uint32_t entity_id_start, uint32_t cur_tick,
                              uint32_t rank, uint32_t world, uint4 *req_send, uint32_t req_cap);
void launch_apply_requests(hipStream_t st, WorldDev w, const uint4 *req_recv, uint32_t world, uint32_t req_cap);
// entities whose member cell belongs to another rank leave (state packed per destination, slot freed)
void launch_export(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                   chd_entity_state *send, uint32_t cap, uint32_t cur_tick, uint32_t extra = 0);
// extra > 0 (log_on): behind a segment's (cap + 1) records, `extra` records of the sender's maxFanOutIntervalMs per cell
void launch_import(hipStream_t st, WorldDev w, const chd_entity_state *recv, uint32_t world, uint32_t cap,
                   uint32_t cur_tick, uint32_t extra = 0, uint32_t ncell = 0);
void launch_spawn_auto(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id,
                       const double *x, const double *z, const uint32_t *flags, const uint32_t *sender,
                       uint32_t cur_tick);
void launch_free_stack_init(hipStream_t st, WorldDev w);
void launch_shard_despawn(hipStream_t st, WorldDev w, const uint32_t *gone_sorted, uint32_t n);
void launch_slot_of_rebuild(hipStream_t st, WorldDev w);  // sh_slot_of from the live slots' channel ids
// the update log by channel (WorldDev::log_on): note the channels that come to life; push this tick's updates of EVERY channel
void launch_log_spawn(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id, const double *x, const double *z);
void launch_log_push(hipStream_t st, DevGrid g, WorldDev w, const double *x_by_chan, const double *z_by_chan, const uint8_t *has_update,
                     uint32_t n_chan, uint32_t cur_tick, int64_t now_ns);
// halo exchange: pack this rank's border bands per destination (segments at seg_off[d]), and append the received ones behind
// the own cell tables (ghost entries from index N on) + the cell views
void launch_halo_pack(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo, unsigned char *send,
                      const uint64_t *seg_off);
void launch_halo_unpack(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo, const unsigned char *recv,
                        const uint64_t *seg_off, const uint32_t *ghost_off, uint32_t cur_tick, const unsigned long long *gate_p = nullptr,
                        unsigned long long gate_target = 0);
// K2: cell index build
bool launch_index_build(hipStream_t st, DevGrid g, WorldDev w, uint32_t cur_tick, const unsigned long long *gate_p = nullptr,
                        unsigned long long gate_target = 0, int64_t now_ns = 0);
// the window columns of the cells that are not fully updated (WorldDev::wcol_*); after the index build
void launch_window_columns(hipStream_t st, DevGrid g, WorldDev w);
#define CHD_WCOLS 9
__host__ __device__ inline uint32_t wcol_mask(uint32_t k) {  // runs of 1..3 ticks that start 0, 1 or 2 ticks back: 1 3 7, 2 6 14, 4 12 28
    return ((2u << (k % 3u)) - 1u) << (k / 3u);
}
// K3/K4: AOI query (+ interest diff when stateful)
struct AoiLimits {
    uint32_t maxax;   // samples per lattice axis
    uint32_t winmax;  // cells in the per-query table
    uint32_t maxdim;  // max(window width, window height) <= max(cols, rows)
};
void launch_aoi_stateless(hipStream_t st, DevGrid g, AoiLimits lim, const chd_aoi_query *q,
                          uint32_t nq, const double *spot_x, const double *spot_z,
                          const uint32_t *spot_dist, uint32_t stride, uint32_t *cells,
                          uint32_t *dists, uint32_t *ivs, uint32_t *counts, int32_t *status, unsigned char *scratch);
size_t aoi_scratch_bytes(AoiLimits lim);  // global scratch per query (long-lattice path)
// CHD_WORLD_GATED_OVERLAP: the two cross-stream dependencies of a serial tick whose interest updates run on a second stream,
// as device-side flags instead of HIP events (an event record idles the recording stream ~7 us, a cross-queue wait takes ~11 us
// to resolve: profiles/r04t_tick_timeline_*.csv).  Two monotonic counters on their own 128-byte lines: [GATE_TOP] interest
// launches complete — raised by a one-wave kernel BEHIND the interest launch on its stream, so the release is the kernel
// boundary itself and holds wherever the dispatcher placed the workgroups — and [GATE_EPI] ticks whose epilogue has finished.
#define GATE_TOP 0
#define GATE_EPI 16
#define GATE_FAIL 32
#define GATE_WORDS 48
void launch_aoi_interest(hipStream_t st, DevGrid g, AoiLimits lim, WorldDev w,
                         const chd_aoi_query *q, uint32_t nq, const uint32_t *q_sub,
                         const double *spot_x, const double *spot_z, const uint32_t *spot_dist,
                         int64_t now_ns, uint32_t cur_tick, unsigned long long *gate_p = nullptr, unsigned long long gate_value = 0);
void launch_gate_raise(hipStream_t st, unsigned long long *p, unsigned long long value);
// do kernels of the two streams run side by side (a waiter on one sees a flag raised on the other)?  What the gates need; probed
// once per world with a bounded wait (tens of ms when they do not)
int streams_run_side_by_side(hipStream_t waiter_st, hipStream_t raiser_st, unsigned long long *d_flag, unsigned *d_seen, bool *ok);
// one wave that returns when *p >= target (bounded: a dependency that never resolves raises OVF_GATE instead of hanging)
void launch_gate_wait(hipStream_t st, WorldDev w, const unsigned long long *p, unsigned long long target, bool fork);
size_t aoi_lds_bytes(AoiLimits lim, uint32_t capq);
size_t aoi_lds_limit();  // the most dynamic LDS an AOI launch may ask for (gfx950: 160 KiB per CU)
// compaction of the fixed-stride stateless output into CSR
void launch_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n);  // exclusive, out[n]=total
void launch_scan_u32_inplace(hipStream_t st, uint32_t *data, uint32_t n);             // data[n] = total
void launch_scan_u64_inplace(hipStream_t st, uint64_t *data, uint32_t n);
void launch_scan_u32_inplace_dev(hipStream_t st, uint32_t *data, uint32_t n_max, const uint32_t *n_dev);  // n = min(n_max, *n_dev)
void launch_csr_gather(hipStream_t st, uint32_t nq, uint32_t stride, const uint32_t *counts,
                       const uint32_t *offsets, const uint32_t *cells, const uint32_t *dists,
                       const uint32_t *ivs, uint32_t *out_ids, uint32_t *out_dists,
                       uint32_t *out_ivs, uint32_t cap, uint32_t id_start, const int32_t *status = nullptr);
// One query beyond the in-kernel limits (lattice lines / cell window), on the whole GPU in passes over global memory; `q` is a
// HOST copy, spots are device arrays.  On CHD_OK: *count results (ascending channel id) in freshly hipMalloc'ed device arrays
// the caller frees; else the reference's error for the query (or CHD_E_TOO_LARGE beyond 65 535 lattice lines per axis).
int launch_aoi_big_query(hipStream_t st, DevGrid g, const chd_aoi_query &q, const double *d_spot_x, const double *d_spot_z,
                         const uint32_t *d_spot_dist, uint32_t **d_ids, uint32_t **d_dists, uint32_t **d_ivs, uint32_t *count);
// wire-format fan-out buffers
struct WireDev {
    uint32_t stride[2];                 // payload slot size in bytes: [0] update (delta), [1] full state
    uint8_t *pay_ent[2], *pay_cell[2];  // [N * stride], [ncell * stride]: serialized google.protobuf.Any per channel
    uint32_t *len_ent[2], *len_cell[2]; // [N], [ncell]
    // merged updates (CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS): the UPDATE payloads are the serialized channel
    // data update MESSAGES (not Any), kept per tick in a ring of CHD_HIST_BITS slots (slot = tick & 31); a record's message is
    // Any{type_url, value = the updates its mask selects, oldest first, concatenated} — what a protobuf parser reads as their merge
    uint32_t merge;                     // 0: one current payload per channel (pay_ent[0] / pay_cell[0] hold Any bytes)
    uint32_t schema;                    // merge mode: CHD_MERGE_SCHEMA_* of the entity update messages (0: concatenate the selected updates)
    uint32_t cur_tick;                  // tick of the records being built (ring slot of mask bit j = (cur_tick - j) & 31)
    uint8_t *ring_ent, *ring_cell;      // [N * 32 * stride[0]], [ncell * 32 * stride[0]]
    uint32_t *rlen_ent, *rlen_cell;     // [N * 32], [ncell * 32]
    uint8_t *url[3];                    // type_url of the entity / spatial channel data update message, and of the handover data
    uint32_t url_len[3];
    uint8_t *pay_objref;                // [N * stride[0]] serialized UnrealObjectRef per entity (CHD_WIRE_ENTITY_OBJREF)
    uint32_t *len_objref;               // [N]
    uint32_t *rec_woff;                 // per record: byte offset of its Packet entry in the connection's stream (~0 = dropped)
    uint32_t *rec_wtag;                 // per record: any_len | (packet length << 16, if it opens a packet)
    uint8_t *seg_fast;                  // [S * capq] per subscription segment: every message carries <= 80 payload bytes (k_wire_copy_fast)
    uint8_t *conn_slow;                 // [S] the connection has segments k_wire_copy_fast does not take
    uint8_t *trash;                     // [S * 4 * 16] where k_wire_copy_fast's idle lanes store (unconditional stores)
    uint32_t fast_ok;                   // capq <= 512: the fast kernel's LDS segment list holds a connection's segments
    uint64_t *conn_wlen;                // [S+1] stream length per connection -> exclusive scan = conn_woff
    uint64_t *conn_woff;                // alias of conn_wlen after the scan
    uint32_t *conn_npk;                 // [S] packets per connection
    uint32_t *n_dropped;                // [2] messages dropped by the size check of Send; records without a valid position word
    uint32_t ncell, npos;               // bounds of the position words: cells of the grid, entries of the cell table
    uint32_t npay;                      // entity payload entries: max_entities (indexed by slot) or, region-sharded (WorldDev::ce_by_chan), the world's channels (indexed by channel id - EntityChannelIdStart)
    uint8_t *bytes;                     // the wire arena
    // streams from the fan-out descriptors (k_wire_layout_img): per cell and payload kind f (0 update, 1 full state) the IMAGE
    // of the cell's messages — [own message][entity messages in column order] — built once per tick
    uint32_t img_on;                    // the path exists for this world (no merged updates)
    uint32_t img_ok[2];                 // ... and its arena of kind f
    uint32_t img_ncol;                  // update images per cell: 1 (the full column) or 1 + CHD_WCOLS (+ the window columns; merge mode: one per window mask)
    uint32_t *img_need;                 // [(img_ncol + 1) * ncell] the tick's descriptors copy from this image (the last ncell: the full-state images)
    uint8_t *img[2];
    uint64_t img_cap[2];
    // indexed by image = col * ncell + cell (family 0: col < img_ncol; family 1: col 0)
    uint32_t *img_off[2];               // [images + 1] image offsets (16-byte aligned; the scan of the padded lengths)
    uint32_t *img_len[2], *img_own[2];  // image bytes; bytes of the own message in front
    uint32_t *img_bad[2];               // a message of the image is dropped by Send / names no slot: its subscriptions take the record path
    uint32_t *img_end[2];               // [col * wcol_stride + table position] end of each entity message inside its image (the packet cuts fall on these)
    uint32_t *cell_dcnt;                // [ncell * dpad + 1] connections per cell of their first subscription -> exclusive scan
    uint32_t dpad;                      // counter stride in words (32 = one per 128-byte line, grids up to 64K cells)
    uint32_t *conn_ndesc, *conn_key, *conn_rank;  // [S] copy descriptors of the connection; its first cell; its place in the copy order
    uint32_t *rank_ndesc;               // [S + 1] descriptor counts in copy order -> exclusive scan
    uint4 *cdesc;                       // {dst lo, dst hi, src offset | kind << 31, bytes}: per connection in stream order, connections by rank
    uint64_t bytes_cap, cdesc_cap;      // what the arenas hold (checked by the writing kernels: wire_img_fits)
    uint32_t *slow_list;                // [S] connections with record-path segments (count: n_dropped[4])
    uint32_t *cp_ticket;                // k_wire_copy_img's ticket counter
};
void launch_wire_images(hipStream_t st, DevGrid g, WorldDev w, WireDev x);
void launch_wire_images_fill(hipStream_t st, DevGrid g, WorldDev w, WireDev x);
void launch_wire_layout_img(hipStream_t st, DevGrid g, WorldDev w, WireDev x, bool emit);
void launch_wire_conn_order(hipStream_t st, WorldDev w, WireDev x);
void launch_wire_copy_img(hipStream_t st, WorldDev w, WireDev x, uint32_t waves);
void launch_wire_layout(hipStream_t st, WorldDev w, WireDev x);
// handover message assembly: sizes[2*nh] then, with off = exclusive scan of the sizes, the bytes
// blob b = handover var_h[b] with the entities of var_mask[b] carrying their entityData (var_h == nullptr: the two classic blobs
// per handover, b >> 1 with none / all)
void launch_handover_msg_sizes(hipStream_t st, DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *var_h, const uint32_t *var_mask, uint32_t *sizes);
void launch_handover_msg_write(hipStream_t st, DevGrid g, WorldDev w, WireDev x, uint32_t n_blobs, const uint32_t *var_h, const uint32_t *var_mask,
                               const uint32_t *off, uint8_t *out, uint64_t cap);
void launch_wire_copy(hipStream_t st, WorldDev w, WireDev x, uint32_t n_slow_conns);
void launch_wire_set_payloads(hipStream_t st, WireDev x, int full, int cell, uint32_t n, uint32_t limit, const uint32_t *idx,
                              const uint32_t *lens, const uint64_t *off, const uint8_t *bytes, uint32_t ring_slot);
// recipient planning (decision parts)
void launch_handover_recipients_count(hipStream_t st, DevGrid g, WorldDev w, uint32_t *off, uint32_t *own_unsub = nullptr);
void launch_handover_recipients_fill(hipStream_t st, DevGrid g, WorldDev w, const uint32_t *off, uint32_t *conn,
                                     uint8_t *kind, uint32_t *full_mask, uint64_t cap);
void launch_adjacent_recipients(hipStream_t st, DevGrid g, WorldDev w, uint32_t n_req, const uint32_t *channel,
                                const uint32_t *broadcast, const uint32_t *sender_conn, const uint32_t *client_conn,
                                uint32_t *off, uint32_t *conns, uint64_t cap, int fill);
// K5: fan-out
#define TC_NDEFER 0
#define TC_NDEEP 1
#define TC_SCAP 2
void launch_fanout_plan(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring);
bool fanout_seg_path(const WorldDev &w);
void launch_fanout_emit_main(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring);
// the tick's tail in one launch: deferred subscriptions, connections without room, the element walk (PF_DEEP), the epilogue
void launch_fanout_tail(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring, uint32_t slot, unsigned long long *epi = nullptr,
                        unsigned long long epi_seq = 0);
// the filtered descriptors (WorldDev::filt_*); no-op unless off_on
void launch_fanout_emit_filt(hipStream_t st, DevGrid g, WorldDev w);
void launch_filt_fold(hipStream_t st, WorldDev w, uint32_t ring_slot);
// per cell and ring slot the range of the sub-tick offsets (WorldDev::cell_orng); after the index build, off_on worlds only
void launch_cell_offsets(hipStream_t st, DevGrid g, WorldDev w);
#define TICK_RING 1024
void launch_tick_epilogue(hipStream_t st, WorldDev w, uint32_t slot, uint32_t ncell, unsigned long long *epi = nullptr, unsigned long long epi_seq = 0);
