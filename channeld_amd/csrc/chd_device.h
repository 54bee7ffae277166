// chd_device.h — shared device-side definitions for the gfx950 kernels.
//
// Arithmetic contract: IEEE float64, every operation rounded
// separately (the library is compiled with -ffp-contract=off), IEEE division
// and correctly rounded sqrt — the same results as amd64 Go, which emits no
// FMA.  wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHD_INVALID 0xFFFFFFFFu
#define CHD_ABSENT 0xFFFFFFFFu
#define CHD_POS_CELL 0x80000000u     // rec_pos: the record is the spatial channel's own message, low bits = cell
#define CHD_NONUNIFORM 0xFFFFFFFFu  // cell_usender: the cell's buffered updates come from more than one sender
#define CHD_NOT_A_SENDER 0xFFFFFFFEu  // (emit, per connection) ... but this connection is none of them: no per-sender test needed
#define CHD_WAVE 64
#define CHD_HIST_BITS 32
#define CHD_MAX_UPDATE_BUFFER 512u  // MaxUpdateMsgBufferSize, data.go:53-55
#define CHD_OFF_SLOTS 8u  // ticks back for which a channel's update keeps its sub-tick arrival offset (WorldDev::off_on)

// pair flags (per subscriber x spatial channel subscription state,
// subscription.go:13-31 + data.go:39-44)
#define PF_HAD_FIRST 1u  // fanOutConnection.hadFirstFanOut
#define PF_SKIP_SELF 2u  // options.SkipSelfUpdateFanOut
#define PF_NO_ACCESS 4u  // options.DataAccess == NO_ACCESS
#define PF_NEW 8u        // subscribed during the current tick
#define PF_WRITE 16u     // options.DataAccess == WRITE_ACCESS (READ_ACCESS when neither this nor PF_NO_ACCESS is set)
#define PF_DEFER 32u     // (inside one tick) due, left by the pipelined emit kernel to the deferred launch
#define PF_DEEP 64u      // (inside one tick) due, but the tick-ring masks cannot answer its windows: served from the exact
                         // update buffers by k_fanout_emit_deep (history_depth > 0), which clears the bit

#define PF_FIELD_MASK_SHIFT 8u  // bits 8..15: the subscription's DataFieldMasks in bit form (chd_sub_options.data_field_mask)

// entity flags
#define EF_LOCKED 1u
#define EF_ALIVE 0x80000000u

// device counters (one uint32 each unless noted), zeroed every tick
#define CHD_LIST_BANKS 64u

enum {
    CTR_HANDOVERS = 0,
    CTR_LOCKED = 1,
    CTR_UNSUBS = 2,   // (fetch-side index only: the device keeps per-bank tails, WorldDev::list_ctr)
    CTR_OVERFLOW = 3,
    CTR_HIST_OVERFLOW = 4,
    CTR_PAIRS = 5,
    CTR_NEWSUBS = 6,
    CTR_SENDER_OVERFLOW = 7,
    CTR_SCAN_DONE = 8,        // workgroups of k_index_scan that have finished (last-block pattern)  // a third sender inside a channel's 32-tick update history
    CTR_COUNT = 16
};

#define OVF_HANDOVER 1u
#define OVF_UNSUB 2u
#define OVF_RECORDS 4u
#define OVF_NEWSUB 8u
#define OVF_INTERNAL 0x8000u  // a kernel's defensive loop bound tripped (a bug, never a capacity)
#define OVF_GATE 0x4000u      // CHD_WORLD_GATED_OVERLAP: a device-side gate was not raised within its spin bound (the two streams did not run side by side): this tick's results are not valid; the world takes the event form from the next tick on
#define OVF_SLOTS 16u    // sharded world: no free entity slot for a spawn / an immigrant (the immigrant waits in limbo, k_shard.hip)
#define OVF_MIGRATE 32u  // sharded world: an emigrant did not fit its destination's send segment
#define OVF_HALO 64u     // sharded world: a border band did not fit its halo segment, or a subscription reaches a cell beyond the halo
#define OVF_DUPLICATE 256u  // chd_tick_device's precondition broken: an entity slot twice in one round of updates / a subscriber slot twice
#define OVF_LOST 128u    // sharded world: more immigrants waiting for a slot than the limbo list holds (max_entities): an entity was dropped

struct DevGrid {
    double gw, gh, offx, offz;
    double gsz;               // GridSize() = sqrt(gw*gw + gh*gh), spatial.go:134-139
    double world_w, world_h;  // WorldWidth()/WorldHeight(), spatial.go:126-132
    uint32_t cols, rows, ncell, id_start;
    uint32_t server_cols, server_rows, sgc, sgr;  // server grid dims, spatial.go:321-330
    uint32_t border;
    uint32_t default_interval_ms;
    int32_t default_delay_ms;
    uint32_t n_damp;
    uint32_t damp_dist[8];
    uint32_t damp_iv[8];
};

// ---- halo exchange of region-sharded worlds (k_shard.hip) ----
// What rank s sends rank d every tick: the cell tables of the cells of s's region that lie within `halo` cells of d's
// region (the generalisation of ServerInterestBorderSize, spatial.go:114-118,481-590: the border band a spatial server
// sees of its neighbours) — a rectangle of cells, empty for ranks further apart than the halo.
struct HaloRect {
    uint32_t x0, y0, w, h;  // cells [x0, x0+w) x [y0, y0+h) of the global grid; w*h == 0: nothing to exchange
};
__host__ __device__ inline HaloRect halo_rect(uint32_t cols, uint32_t rows, uint32_t server_cols, uint32_t sgc, uint32_t sgr,
                                              uint32_t halo, uint32_t s, uint32_t d) {
    HaloRect r = {0, 0, 0, 0};
    if (s == d) return r;
    const uint32_t sx0 = (s % server_cols) * sgc, sy0 = (s / server_cols) * sgr;
    const uint32_t dx0 = (d % server_cols) * sgc, dy0 = (d / server_cols) * sgr;
    const uint32_t sx1 = sx0 + sgc < cols ? sx0 + sgc : cols, sy1 = sy0 + sgr < rows ? sy0 + sgr : rows;
    const uint32_t dx1 = dx0 + sgc < cols ? dx0 + sgc : cols, dy1 = dy0 + sgr < rows ? dy0 + sgr : rows;
    const uint32_t ex0 = dx0 > halo ? dx0 - halo : 0u, ey0 = dy0 > halo ? dy0 - halo : 0u;
    const uint32_t ex1 = dx1 + halo < cols ? dx1 + halo : cols, ey1 = dy1 + halo < rows ? dy1 + halo : rows;
    const uint32_t x0 = sx0 > ex0 ? sx0 : ex0, x1 = sx1 < ex1 ? sx1 : ex1;
    const uint32_t y0 = sy0 > ey0 ? sy0 : ey0, y1 = sy1 < ey1 ? sy1 : ey1;
    if (x0 >= x1 || y0 >= y1) return r;
    r.x0 = x0; r.y0 = y0; r.w = x1 - x0; r.h = y1 - y0;
    return r;
}
// entries a halo segment can carry: the band's share of the sender's entity slots, with headroom for clustering
__host__ __device__ inline uint32_t halo_cap_entries(uint32_t n_slots, uint32_t rect_cells, uint32_t region_cells) {
    if (!rect_cells) return 0u;
    const uint64_t c = (uint64_t)n_slots * rect_cells / (region_cells ? region_cells : 1u);
    const uint64_t cap = c + c / 4 + 256u;
    return (uint32_t)(cap < n_slots ? cap : n_slots);
}
// Segment layout (all parts 16-byte aligned): header {n_entries, n_cells, overflow, -} | per rect cell {usender, smin, smax,
// hand} | entries (16 B x cap) | per rect cell count (u32) | previous senders (u32 x cap)
__host__ __device__ inline uint64_t halo_seg_bytes(uint32_t cap, uint32_t rect_cells) {
    if (!rect_cells) return 0ull;
    const uint64_t a = 16ull + 16ull * rect_cells + 16ull * cap;
    const uint64_t b = ((4ull * rect_cells + 15ull) & ~15ull) + ((4ull * cap + 15ull) & ~15ull);
    return a + b;
}

struct TickRing {
    int64_t t[CHD_HIST_BITS];  // t[j] = arrival stamp of tick (cur - j); valid for j < n
    uint32_t n;
    uint32_t cur_tick;
};

// int(math.Floor(v)) then `< 0 || >= n`  (spatial.go:170-177).  NaN/±Inf/huge
// convert to MinInt64 on amd64, i.e. "negative": same outcome as this predicate.
__device__ __forceinline__ bool grid_coord(double v, uint32_t n, uint32_t &out) {
    double f = floor(v);
    if (!(f >= 0.0) || !(f < (double)n)) return false;
    out = (uint32_t)f;
    return true;
}

// GetChannelIdWithOffset (spatial.go:169-180) as a cell INDEX (id - id_start),
// CHD_INVALID for the error return.
__device__ __forceinline__ uint32_t cell_of(const DevGrid &g, double x, double z) {
    uint32_t gx, gy;
    if (!grid_coord((x - g.offx) / g.gw, g.cols, gx)) return CHD_INVALID;
    if (!grid_coord((z - g.offz) / g.gh, g.rows, gy)) return CHD_INVALID;
    return gx + gy * g.cols;
}

__device__ __forceinline__ uint32_t server_of(const DevGrid &g, uint32_t cell) {
    // GetRegions (spatial.go:336-351)
    uint32_t x = cell % g.cols, y = cell / g.cols;
    return x / g.sgc + (y / g.sgr) * g.server_cols;
}

// math.Min / math.Max with Go's special cases (Go src/math/dim.go)
__device__ __forceinline__ double go_min(double x, double y) {
    if ((isinf(x) && x < 0) || (isinf(y) && y < 0)) return -INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}
__device__ __forceinline__ double go_max(double x, double y) {
    if ((isinf(x) && x > 0) || (isinf(y) && y > 0)) return INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}

// uint(math.Ceil(d)) for the small non-negative values of the path
__device__ __forceinline__ uint32_t go_uint_ceil(double d) {
    double c = ceil(d);
    if (!(c >= 0.0)) return 0;
    if (c >= 4294967295.0) return 4294967295u;
    return (uint32_t)c;
}

// getSpatialDampingSettings + nil branch (message_spatial.go:31-38,66-79)
__device__ __forceinline__ uint32_t damping_interval(const DevGrid &g, uint32_t dist) {
    // first matching entry wins; written as a fully unrolled reverse select so that every table read has a
    // constant index (scalar kernarg loads, once per kernel) instead of a per-lane load from the kernarg segment
    uint32_t iv = g.default_interval_ms;
#pragma unroll
    for (int i = CHD_MAX_DAMPING - 1; i >= 0; i--)
        iv = ((uint32_t)i < g.n_damp && dist <= g.damp_dist[i]) ? g.damp_iv[i] : iv;
    return iv;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// number of set bits below this lane
__device__ __forceinline__ uint32_t mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
