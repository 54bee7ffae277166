// k_spatial.hip — cell assignment, handover detection, regions, adjacency.
//
// Replaces (reference file:line): GetChannelIdWithOffset spatial.go:169-180,
// the decision part of Notify spatial.go:612-626 + lock abort :675-679 +
// entity-map move :703-736, GetRegions :319-356, GetAdjacentChannels :358-381,
// CreateChannels cell ownership :399-424, subToAdjacentChannels :481-590.
//
// All kernels are HBM-streaming (one thread per point, SoA f64 loads, coalesced
// 4-byte stores); handover records are compacted per wave with ballot +
// mbcnt and one atomic per wave.
#include "chd_kernels.h"

static inline unsigned nblocks(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

__global__ void __launch_bounds__(256) k_get_channel_ids(DevGrid g, const double *__restrict__ x,
                                                         const double *__restrict__ z, uint32_t n,
                                                         uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t c = cell_of(g, x[i], z[i]);
    out[i] = (c == CHD_INVALID) ? 0u : c + g.id_start;
}

void launch_get_channel_ids(hipStream_t st, DevGrid g, const double *x, const double *z,
                            uint32_t n, uint32_t *out) {
    if (!n) return;
    hipLaunchKernelGGL(k_get_channel_ids, dim3(nblocks(n, 256)), dim3(256), 0, st, g, x, z, n, out);
}

__global__ void __launch_bounds__(256) k_notify_decide(DevGrid g, const double *__restrict__ ox,
                                                       const double *__restrict__ oz,
                                                       const double *__restrict__ nx,
                                                       const double *__restrict__ nz, uint32_t n,
                                                       uint32_t *__restrict__ src,
                                                       uint32_t *__restrict__ dst,
                                                       uint8_t *__restrict__ handover) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    // spatial.go:613-626: src error returns before dst is computed
    uint32_t s = cell_of(g, ox[i], oz[i]);
    uint32_t d = CHD_INVALID;
    if (s != CHD_INVALID) d = cell_of(g, nx[i], nz[i]);
    src[i] = (s == CHD_INVALID) ? 0u : s + g.id_start;
    dst[i] = (d == CHD_INVALID) ? 0u : d + g.id_start;
    handover[i] = (s != CHD_INVALID && d != CHD_INVALID && s != d) ? 1 : 0;
}

void launch_notify_decide(hipStream_t st, DevGrid g, const double *ox, const double *oz,
                          const double *nx, const double *nz, uint32_t n, uint32_t *src,
                          uint32_t *dst, uint8_t *handover) {
    if (!n) return;
    hipLaunchKernelGGL(k_notify_decide, dim3(nblocks(n, 256)), dim3(256), 0, st, g, ox, oz, nx, nz, n,
                       src, dst, handover);
}

__global__ void __launch_bounds__(256) k_regions(DevGrid g, double *min_x, double *min_z,
                                                 double *max_x, double *max_z,
                                                 uint32_t *channel_id, uint32_t *server_index) {
    uint32_t index = blockIdx.x * 256u + threadIdx.x;
    if (index >= g.ncell) return;
    uint32_t x = index % g.cols, y = index / g.cols;
    // spatial.go:340-350
    min_x[index] = g.offx + g.gw * (double)x;
    min_z[index] = g.offz + g.gh * (double)y;
    max_x[index] = g.offx + g.gw * (double)(x + 1);
    max_z[index] = g.offz + g.gh * (double)(y + 1);
    channel_id[index] = g.id_start + index;
    server_index[index] = x / g.sgc + (y / g.sgr) * g.server_cols;
}

void launch_regions(hipStream_t st, DevGrid g, double *min_x, double *min_z, double *max_x,
                    double *max_z, uint32_t *channel_id, uint32_t *server_index) {
    hipLaunchKernelGGL(k_regions, dim3(nblocks(g.ncell, 256)), dim3(256), 0, st, g, min_x, min_z,
                       max_x, max_z, channel_id, server_index);
}

__global__ void __launch_bounds__(256) k_adjacent(DevGrid g, const uint32_t *__restrict__ ids,
                                                  uint32_t n, uint32_t *__restrict__ out,
                                                  uint32_t *__restrict__ counts) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t index = ids[i] - g.id_start;
    int32_t gx = (int32_t)(index % g.cols), gy = (int32_t)(index / g.cols);
    uint32_t k = 0;
    for (int32_t y = gy - 1; y <= gy + 1; y++) {
        if (y < 0 || y > (int32_t)(g.rows - 1)) continue;
        for (int32_t x = gx - 1; x <= gx + 1; x++) {
            if (x < 0 || x > (int32_t)(g.cols - 1)) continue;
            if (x == gx && y == gy) continue;
            out[8u * i + k++] = (uint32_t)x + (uint32_t)y * g.cols + g.id_start;
        }
    }
    counts[i] = k;
}

void launch_adjacent(hipStream_t st, DevGrid g, const uint32_t *ids, uint32_t n, uint32_t *out,
                     uint32_t *counts) {
    if (!n) return;
    hipLaunchKernelGGL(k_adjacent, dim3(nblocks(n, 256)), dim3(256), 0, st, g, ids, n, out, counts);
}

// GetChannelIdNoOffset (spatial.go:165-167)
__device__ __forceinline__ uint32_t cell_no_offset(const DevGrid &g, double x, double z) {
    uint32_t gx, gy;
    if (!grid_coord((x - 0.0) / g.gw, g.cols, gx)) return CHD_INVALID;
    if (!grid_coord((z - 0.0) / g.gh, g.rows, gy)) return CHD_INVALID;
    return gx + gy * g.cols;
}

// One thread: the reference's nested loops in their own order (control plane,
// a few dozen cells; kept on the device so that no grid arithmetic lives on the host).
__global__ void k_server_cells(DevGrid g, uint32_t server_index, int mode, uint32_t *out,
                               uint32_t cap, uint32_t *n_out, uint32_t *err) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t sx = server_index % g.server_cols, sy = server_index / g.server_cols;
    uint32_t n = 0;
    *err = 0;
#define ADD(px, pz)                                   \
    do {                                              \
        uint32_t _c = cell_no_offset(g, (px), (pz));  \
        if (_c == CHD_INVALID) { *err = 1; *n_out = 0; return; } \
        if (n >= cap) { *err = 2; *n_out = 0; return; }          \
        out[n++] = _c + g.id_start;                   \
    } while (0)
    if (mode == 0) {  // spatial.go:410-424
        for (uint32_t y = 0; y < g.sgr; y++)
            for (uint32_t x = 0; x < g.sgc; x++)
                ADD((double)(sx * g.sgc + x) * g.gw, (double)(sy * g.sgr + y) * g.gh);
    } else if (g.border != 0) {  // spatial.go:481-590
        if (cell_no_offset(g, (double)(sx * g.sgc) * g.gw, (double)(sy * g.sgr) * g.gh) == CHD_INVALID) {
            *err = 1; *n_out = 0; return;
        }
        if (sx > 0)
            for (uint32_t y = 0; y < g.sgr; y++)
                for (uint32_t x = 1; x <= g.border; x++)
                    ADD((double)(sx * g.sgc - x) * g.gw, (double)(sy * g.sgr + y) * g.gh);
        if (sx < g.server_cols - 1)
            for (uint32_t y = 0; y < g.sgr; y++)
                for (uint32_t x = 0; x < g.border; x++)
                    ADD((double)((sx + 1) * g.sgc + x) * g.gw, (double)(sy * g.sgr + y) * g.gh);
        if (sy > 0)
            for (uint32_t y = 1; y <= g.border; y++)
                for (uint32_t x = 0; x < g.sgc; x++)
                    ADD((double)(sx * g.sgc + x) * g.gw, (double)(sy * g.sgr - y) * g.gh);
        if (sy < g.server_rows - 1)
            for (uint32_t y = 0; y < g.border; y++)
                for (uint32_t x = 0; x < g.sgc; x++)
                    ADD((double)(sx * g.sgc + x) * g.gw, (double)((sy + 1) * g.sgr + y) * g.gh);
    }
#undef ADD
    *n_out = n;
}

void launch_server_cells(hipStream_t st, DevGrid g, uint32_t server_index, int mode,
                         uint32_t *out, uint32_t cap, uint32_t *n_out, uint32_t *err) {
    hipLaunchKernelGGL(k_server_cells, dim3(1), dim3(64), 0, st, g, server_index, mode, out, cap, n_out, err);
}

// ------------------------------------------------------------------------
// world: spawn / despawn / subscribers
// ------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_spawn(DevGrid g, WorldDev w, uint32_t n,
                                               const uint32_t *__restrict__ idx,
                                               const uint32_t *__restrict__ chan_id,
                                               const double *__restrict__ x,
                                               const double *__restrict__ z,
                                               const uint32_t *__restrict__ flags,
                                               const uint32_t *__restrict__ sender,
                                               uint32_t cur_tick) {
    uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n) return;
    uint32_t i = idx ? idx[u] : u;
    if (i >= w.N) return;
    uint32_t c = cell_of(g, x[u], z[u]);
    w.chan_id[i] = chan_id[u];
    w.cell[i] = c;
    w.member[i] = c;
    w.eflags[i] = (flags ? (flags[u] & ~EF_ALIVE) : 0u) | EF_ALIVE;
    w.sender[i] = sender ? sender[u] : 0u;
    w.hist[i] = 0;
    w.hist_tick[i] = cur_tick;
    w.sender_prev[i] = 0;
    w.hist_prev[i] = 0;
    if (w.deep_depth) {  // a new channel's update buffer starts empty
        w.deep_n[i] = 0;
        w.deep_len[i] = 0;
        w.deep_drop[i] = INT64_MIN;
        w.irr_tick[i] = 0;
        w.ent_max_iv[i] = 0;
    }
}

void launch_spawn(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *idx,
                  const uint32_t *chan_id, const double *x, const double *z,
                  const uint32_t *flags, const uint32_t *sender, uint32_t cur_tick) {
    if (!n) return;
    hipLaunchKernelGGL(k_spawn, dim3(nblocks(n, 256)), dim3(256), 0, st, g, w, n, idx, chan_id, x, z,
                       flags, sender, cur_tick);
}

__global__ void __launch_bounds__(256) k_despawn(WorldDev w, uint32_t n, const uint32_t *idx) {
    uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n) return;
    uint32_t i = idx ? idx[u] : u;
    if (i >= w.N) return;
    w.eflags[i] = 0;
    w.member[i] = CHD_INVALID;
    w.cell[i] = CHD_INVALID;
}

void launch_despawn(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *idx) {
    if (!n) return;
    hipLaunchKernelGGL(k_despawn, dim3(nblocks(n, 256)), dim3(256), 0, st, w, n, idx);
}

__global__ void __launch_bounds__(256) k_set_flags(WorldDev w, uint32_t n, const uint32_t *idx,
                                                   const uint32_t *flags) {
    uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n) return;
    uint32_t i = idx ? idx[u] : u;
    if (i >= w.N) return;
    uint32_t alive = w.eflags[i] & EF_ALIVE;
    w.eflags[i] = (flags[u] & ~EF_ALIVE) | alive;
}

void launch_set_flags(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *idx,
                      const uint32_t *flags) {
    if (!n) return;
    hipLaunchKernelGGL(k_set_flags, dim3(nblocks(n, 256)), dim3(256), 0, st, w, n, idx, flags);
}

__global__ void __launch_bounds__(256) k_subs_add(WorldDev w, uint32_t n, const uint32_t *slot,
                                                  const uint32_t *conn, int add) {
    uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n) return;
    uint32_t s = slot ? slot[u] : u;
    if (s >= w.S) return;
    // a connection that goes away (or is replaced) drops its subscriptions
    // (data.go:183-188): per-cell subscriber counts and its interest bitmap row follow
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        for (uint32_t p = 0; p < cnt; p++) atomicSub(&w.cell_ref[w.pair_cell[(size_t)s * w.capq + p]], 1u);
    }
    if (w.sub_bits)
        for (uint32_t k = 0; k < w.wb; k++) w.sub_bits[(size_t)s * w.wb + k] = 0;
    w.sub_alive[s] = add ? 1u : 0u;
    w.conn_id[s] = add ? conn[u] : 0u;
    w.pair_cnt[s] = 0;
}

void launch_subs_add(hipStream_t st, WorldDev w, uint32_t n, const uint32_t *slot,
                     const uint32_t *conn, int add) {
    if (!n) return;
    hipLaunchKernelGGL(k_subs_add, dim3(nblocks(n, 256)), dim3(256), 0, st, w, n, slot, conn, add);
}

// SubscribeToChannel(conn, spatial channel, options) for explicit SUB_TO_CHANNEL messages (subscription.go:34-102;
// handleSubToChannel, message.go) — the subscriptions an interest update creates carry only the damped interval
// (message_spatial.go:66-79), every other option comes through here: the spatial servers' own subscriptions
// (WRITE access, spatial.go:481-590) and clients that change DataAccess / SkipSelfUpdateFanOut / the interval.
// One wave per connection, its records applied in call order (the reference handles a connection's messages one
// after the other).  Already subscribed: the present fields overwrite the stored options (proto.Merge, :44-57), the
// fan-out state stays, result = dataAccessChanged.  Else a new subscription: defaults (:21-31) merged with the
// options, hadFirstFanOut = SkipFirstFanOut, lastFanOutTime = now + FanOutDelayMs (:59-75), result = true;
// inserted into the connection's list (ascending cell), interest bitmap and per-cell count updated.
__global__ void __launch_bounds__(64) k_subs_set_options(DevGrid g, WorldDev w, const chd_sub_options *__restrict__ opts,
                                                         const uint32_t *__restrict__ order, const uint32_t *__restrict__ grp_off,
                                                         int64_t now_ns, uint8_t *__restrict__ should_send, int32_t *__restrict__ status) {
    const uint32_t lane = lane_id();
    const uint32_t r0 = grp_off[blockIdx.x], r1 = grp_off[blockIdx.x + 1];
    const uint32_t s = opts[order[r0]].slot;
    const size_t pbase = (size_t)s * w.capq;
    const bool alive = w.sub_alive[s] != 0;
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t i = order[r];
        const chd_sub_options o = opts[i];
        if (!alive) {  // IsClosing / no such connection: (nil, false)
            if (lane == 0) { should_send[i] = 0; status[i] = CHD_E_INVAL; }
            continue;
        }
        const uint32_t c = o.channel - g.id_start;
        uint32_t cnt = w.pair_cnt[s];
        // position of the first subscription with cell >= c (the list is ascending)
        uint32_t pos = cnt;
        bool found = false;
        for (uint32_t b = 0; b < cnt; b += 64) {
            const uint32_t v = b + lane < cnt ? w.pair_cell[pbase + b + lane] : 0xFFFFFFFFu;
            const uint64_t m = __ballot(v >= c);
            if (m) {
                pos = b + (uint32_t)__ffsll((unsigned long long)m) - 1u;
                found = (uint32_t)__shfl((int)v, (int)(pos - b)) == c;
                break;
            }
        }
        if (found) {
            if (lane == 0) {
                uint32_t fl = w.pair_flags[pbase + pos];
                const uint32_t acc_old = (fl & PF_NO_ACCESS) ? 0u : (fl & PF_WRITE) ? 2u : 1u;
                uint32_t acc = acc_old;
                if (o.set & CHD_SUBOPT_ACCESS) acc = o.data_access;
                fl &= ~(PF_NO_ACCESS | PF_WRITE);
                fl |= acc == 0 ? PF_NO_ACCESS : acc == 2 ? PF_WRITE : 0u;
                if (o.set & CHD_SUBOPT_SKIP_SELF) fl = o.skip_self_update_fanout ? (fl | PF_SKIP_SELF) : (fl & ~PF_SKIP_SELF);
                if (o.set & CHD_SUBOPT_FIELD_MASK) fl = (fl & ~(0xFFu << PF_FIELD_MASK_SHIFT)) | ((o.data_field_mask & 0xFFu) << PF_FIELD_MASK_SHIFT);
                w.pair_flags[pbase + pos] = fl;
                if (o.set & CHD_SUBOPT_INTERVAL) w.pair_iv[pbase + pos] = o.fanout_interval_ms;
                // (maxFanOutIntervalMs is not raised by a merge: subscription.go:83-86 sits on the new-subscription branch)
                should_send[i] = acc != acc_old ? 1 : 0;  // dataAccessChanged
                status[i] = CHD_OK;
            }
            continue;
        }
        if (cnt >= w.capq) {
            if (lane == 0) { should_send[i] = 0; status[i] = CHD_E_CAPACITY; }
            continue;
        }
        // open a slot at pos: move [pos, cnt) up by one, from the top, 64 entries at a time (read, then write)
        for (uint32_t hi = cnt; hi > pos;) {
            const uint32_t lo = hi - pos > 64 ? hi - 64 : pos;
            const uint32_t k = lo + lane;
            uint32_t vc = 0, vi = 0, vf = 0;
            int64_t vl = 0;
            if (k < hi) { vc = w.pair_cell[pbase + k]; vi = w.pair_iv[pbase + k]; vf = w.pair_flags[pbase + k]; vl = w.pair_last[pbase + k]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (k < hi) { w.pair_cell[pbase + k + 1] = vc; w.pair_iv[pbase + k + 1] = vi; w.pair_flags[pbase + k + 1] = vf; w.pair_last[pbase + k + 1] = vl; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            hi = lo;
        }
        if (lane == 0) {
            const uint32_t acc = (o.set & CHD_SUBOPT_ACCESS) ? o.data_access : 1u;  // default READ_ACCESS
            const bool skip_self = (o.set & CHD_SUBOPT_SKIP_SELF) ? o.skip_self_update_fanout != 0 : true;
            const bool skip_first = (o.set & CHD_SUBOPT_SKIP_FIRST) ? o.skip_first_fanout != 0 : false;
            const int32_t delay = (o.set & CHD_SUBOPT_DELAY) ? o.fanout_delay_ms : g.default_delay_ms;
            w.pair_cell[pbase + pos] = c;
            w.pair_iv[pbase + pos] = (o.set & CHD_SUBOPT_INTERVAL) ? o.fanout_interval_ms : g.default_interval_ms;
            if (w.deep_depth) {  // subscription.go:83-86 (a host call between ticks: in force at once, and kept by the next fold)
                atomicMax(&w.cell_max_iv[c], w.pair_iv[pbase + pos]);
                atomicMax(&w.cell_max_iv[g.ncell + c], w.pair_iv[pbase + pos]);
            }
            w.pair_last[pbase + pos] = now_ns + (int64_t)delay * 1000000;
            w.pair_flags[pbase + pos] = (acc == 0 ? PF_NO_ACCESS : acc == 2 ? PF_WRITE : 0u) | (skip_self ? PF_SKIP_SELF : 0u) |
                                        (skip_first ? PF_HAD_FIRST : 0u) |
                                        ((o.set & CHD_SUBOPT_FIELD_MASK) ? (o.data_field_mask & 0xFFu) << PF_FIELD_MASK_SHIFT : 0u);
            w.pair_cnt[s] = cnt + 1;
            atomicAdd(&w.cell_ref[c], 1u);
            if (w.wb) w.sub_bits[(size_t)s * w.wb + (c >> 6)] |= 1ull << (c & 63u);
            should_send[i] = 1;
            status[i] = CHD_OK;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_subs_set_options(hipStream_t st, DevGrid g, WorldDev w, const chd_sub_options *opts, const uint32_t *order,
                             const uint32_t *grp_off, uint32_t n_groups, int64_t now_ns, uint8_t *should_send, int32_t *status) {
    if (!n_groups) return;
    hipLaunchKernelGGL(k_subs_set_options, dim3(n_groups), dim3(64), 0, st, g, w, opts, order, grp_off, now_ns, should_send, status);
}

// the options of a connection's subscriptions (chd_subs_get_options): 0 NO_ACCESS / 1 READ / 2 WRITE, SkipSelfUpdateFanOut
__global__ void __launch_bounds__(64) k_subs_get_options(WorldDev w, uint32_t s, uint8_t *access, uint8_t *skip_self) {
    const uint32_t cnt = w.pair_cnt[s];
    for (uint32_t k = threadIdx.x; k < cnt; k += 64) {
        const uint32_t fl = w.pair_flags[(size_t)s * w.capq + k];
        access[k] = (fl & PF_NO_ACCESS) ? 0 : (fl & PF_WRITE) ? 2 : 1;
        skip_self[k] = (fl & PF_SKIP_SELF) ? 1 : 0;
    }
}
void launch_subs_get_options(hipStream_t st, WorldDev w, uint32_t s, uint8_t *access, uint8_t *skip_self) {
    hipLaunchKernelGGL(k_subs_get_options, dim3(1), dim3(64), 0, st, w, s, access, skip_self);
}

__global__ void __launch_bounds__(256) k_group_locks(WorldDev w) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= w.n_groups) return;
    uint32_t n = 0;
    for (uint32_t q = w.grp_off[k]; q < w.grp_off[k + 1]; q++) {
        const uint32_t f = w.eflags[w.grp_mem[q]];
        n += ((f & EF_ALIVE) && (f & EF_LOCKED)) ? 1u : 0u;
    }
    w.grp_locked[k] = n;
}
void launch_group_locks(hipStream_t st, WorldDev w) {
    if (!w.n_groups || w.grp_exact) return;
    hipLaunchKernelGGL(k_group_locks, dim3(nblocks(w.n_groups, 256)), dim3(256), 0, st, w);
}

// ------------------------------------------------------------------------
// K1: ingest one batch of entity position updates.
//   src = cell of the last merged position (== GetChannelId(oldInfo), kept as
//         the 4-byte cell index instead of re-reading 16 bytes of old position)
//   dst = GetChannelId(newInfo)
//   either invalid -> no handover (spatial.go:613-622); equal -> nothing (:624);
//   locked -> abort (entity.go:197-224, spatial.go:675-679);
//   else record {entity, src, dst, servers} and move the entity to dst's
//   entity map (spatial.go:703-736).
// Every update also enters the entity channel's update history
// (ChannelData.OnUpdate, data.go:159-164) as bit 0 of a 32-tick bitmask.
// Traffic per update: 16 B (x,z) + 4 B cell R + 4 B cell W + 8 B history R/W
// + 4 B flags; handover records are rare (1-2 %).
// ------------------------------------------------------------------------
#define ING_ITEMS 1
// (`bid` = which 256-update block of the batch: blockIdx.x in k_ingest, a loop variable in the fused front kernel)
__device__ __forceinline__ void ingest_block(const DevGrid &g, const WorldDev &w, uint32_t n,
                                             const uint32_t *__restrict__ idx,
                                             const double *__restrict__ x,
                                             const double *__restrict__ z,
                                             const uint32_t *__restrict__ sender,
                                             uint32_t cur_tick, const int64_t *__restrict__ arrival, int64_t now, uint32_t bid,
                                             uint32_t mark = 0) {
    // handover records are compacted per wave (ballot + mbcnt) and per workgroup
    // (LDS), so the global counter sees ONE atomic per 1024 updates: same-address
    // atomics serialise at ~12 ns each and would otherwise dominate this kernel.
    __shared__ uint32_t s_cnt[4 * ING_ITEMS], s_lock[4 * ING_ITEMS];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    bool ho[ING_ITEMS], locked[ING_ITEMS];
    uint32_t ent[ING_ITEMS], src[ING_ITEMS], dst[ING_ITEMS];
    uint64_t hm[ING_ITEMS];
#pragma unroll
    for (int j = 0; j < ING_ITEMS; j++) {
        const uint32_t u = (bid * ING_ITEMS + j) * 256u + threadIdx.x;
        ho[j] = false; locked[j] = false;
        ent[j] = 0; src[j] = CHD_INVALID; dst[j] = CHD_INVALID;
        if (u < n) {
            const uint32_t i = idx ? idx[u] : u;
            ent[j] = i;
            // every word of the update and of the entity's state, requested before the first test (an index beyond N reads slot 0's
            // and is dropped): see upd_prefetch
            const uint32_t ic = i < w.N ? i : 0u;
            const double px = x[u], pz = z[u];
            const uint32_t efl = w.eflags[ic], cell_old = w.cell[ic];
            const UpdPre P = upd_prefetch(w, ic);
            const uint32_t *sp = sender ? sender : w.sender;
            const uint32_t snd_u = sp[sender ? u : ic];
            const int64_t *ap = arrival ? arrival : (const int64_t *)(const void *)x;  // (no stamps: any readable 8 bytes, unused)
            const int64_t arr_raw = ap[u];
            // chd_tick_device cannot see duplicates on the host: one returning atomic per EXPLICITLY indexed update (the identity
            // mapping cannot repeat a slot) turns a broken precondition into an error instead of a race
            if (idx && i < w.N && atomicExch(&w.upd_mark[i], mark) == mark) atomicOr(&w.counters[CTR_OVERFLOW], OVF_DUPLICATE);
            uint32_t ef = (i < w.N) ? efl : 0u;
            if (ef & EF_ALIVE) {
                dst[j] = cell_of(g, px, pz);
                src[j] = cell_old;
                w.cell[i] = dst[j];
                push_update_pre(w, i, P, snd_u, cur_tick, src[j], dst[j], arrival ? arr_raw : now, now);
                if (src[j] != CHD_INVALID && dst[j] != CHD_INVALID && src[j] != dst[j]) {
                    // GetHandoverEntities (entity.go:197-224): a locked member of the notifier's handover group
                    // empties the list and the handover does not happen (spatial.go:675-679)
                    bool lk = (ef & EF_LOCKED) != 0;
                    if (w.n_groups) {
                        const uint32_t gi = w.grp_of[i];
                        if (gi != CHD_INVALID) {
                            // (exact lists: len(handoverEntities) == 0 -> "No handover happens", spatial.go:675-679)
                            if (w.grp_exact ? w.grp_off[gi + 1] == w.grp_off[gi] : w.grp_locked[gi] != 0) lk = true;
                        }
                    }
                    if (lk) locked[j] = true;
                    else ho[j] = true;
                }
            }
        }
        hm[j] = __ballot(ho[j]);
        uint64_t lm = __ballot(locked[j]);
        if (lane == 0) {
            s_cnt[wave * ING_ITEMS + j] = (uint32_t)__popcll(hm[j]);
            s_lock[wave * ING_ITEMS + j] = (uint32_t)__popcll(lm);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, ltot = 0;
        for (int k = 0; k < 4 * ING_ITEMS; k++) { tot += s_cnt[k]; ltot += s_lock[k]; }
        uint32_t base = tot ? atomicAdd(&w.counters[CTR_HANDOVERS], tot) : 0u;
        if (ltot) atomicAdd(&w.counters[CTR_LOCKED], ltot);
        for (int k = 0; k < 4 * ING_ITEMS; k++) { uint32_t c = s_cnt[k]; s_cnt[k] = base; base += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ING_ITEMS; j++) {
        if (!ho[j]) continue;
        const uint32_t i = ent[j];
        bool self_moves = true;
        uint32_t moved = 1u, qa = 0;  // (WorldDev::ho_moved; no list: the notifier alone, entity 0 of its handover)
        if (w.n_groups) {
            // the whole handover group leaves src's entity map for dst's (spatial.go:703-736 run over handoverEntities):
            // members that are in src's map move; one that is elsewhere stays where it is (RemoveEntity(src) fails for
            // it; the reference would ALSO add it to dst's map, an entity in two maps — not modelled).  A member that
            // hands over by its own update in this tick ends in its own dst either way (its store wins or this CAS fails).
            // Exact lists (chd_world_set_handover_lists): the notifier itself moves only if its list names it.
            const uint32_t gi = w.grp_of[i];
            if (gi != CHD_INVALID) {
                if (w.grp_exact) self_moves = false;
                moved = 0;
                for (uint32_t q = w.grp_off[gi]; q < w.grp_off[gi + 1]; q++) {
                    const uint32_t m = w.grp_mem[q];
                    bool mv = false;
                    if (m == i) { self_moves = true; mv = true; }
                    else if (w.eflags[m] & EF_ALIVE) mv = atomicCAS(&w.member[m], src[j], dst[j]) == src[j];
                    if (m == i || (w.eflags[m] & EF_ALIVE)) {  // (position among the live members: the order the message lists them in)
                        if (mv && qa < 32u) moved |= 1u << qa;
                        qa++;
                    }
                }
            }
        }
        if (self_moves) w.member[i] = dst[j];
        uint32_t pos = s_cnt[wave * ING_ITEMS + j] + mask_rank(hm[j]);
        if (pos < w.handovers_cap && w.ho_moved) w.ho_moved[pos] = moved;
        if (pos < w.handovers_cap) {
            chd_handover_rec r;
            r.entity = i;
            r.channel = w.chan_id[i];
            r.src = src[j] + g.id_start;
            r.dst = dst[j] + g.id_start;
            r.src_server = server_of(g, src[j]);
            r.dst_server = server_of(g, dst[j]);
            w.handovers[pos] = r;
        } else {
            atomicOr(&w.counters[CTR_OVERFLOW], OVF_HANDOVER);
        }
    }
}

__global__ void __launch_bounds__(256) k_ingest(DevGrid g, WorldDev w, uint32_t n,
                                                const uint32_t *__restrict__ idx,
                                                const double *__restrict__ x,
                                                const double *__restrict__ z,
                                                const uint32_t *__restrict__ sender,
                                                uint32_t cur_tick, const int64_t *__restrict__ arrival, int64_t now, uint32_t mark) {
    ingest_block(g, w, n, idx, x, z, sender, cur_tick, arrival, now, blockIdx.x, mark);
}

void launch_ingest(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *idx,
                   const double *x, const double *z, const uint32_t *sender, uint32_t cur_tick,
                   const int64_t *arrival, int64_t now_ns, uint32_t round) {
    if (!n) return;
    // (tick, round): distinct for every ingest launch of the last 2^24 ticks; never 0 = "no update yet"
    const uint32_t mark = ((cur_tick << 8) | (round & 0xFFu)) | 0x80000000u;
    hipLaunchKernelGGL(k_ingest, dim3(nblocks(n, 256 * ING_ITEMS)), dim3(256), 0, st, g, w, n, idx, x, z,
                       sender, cur_tick, arrival, now_ns, mark);
}

// spatial-channel data updates (spawn/destroy merges through OnUpdate).  One thread per
// cell applies the batch's updates of its cell in batch order (same two-sender history
// as the entity channels), so several updates of one cell in a tick are deterministic.
__global__ void __launch_bounds__(256) k_cell_updates(DevGrid g, WorldDev w, uint32_t n,
                                                      const uint32_t *__restrict__ chan,
                                                      const uint32_t *__restrict__ sender,
                                                      uint32_t cur_tick, const int64_t *__restrict__ arrival, int64_t now) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= g.ncell) return;
    bool touched = false, irregular = false;
    uint32_t h = 0, hp = 0, cur = 0, prev = 0;
    uint32_t dn = 0, dlen = 0;
    int64_t ddrop = INT64_MIN;
    const uint32_t max_iv = w.deep_depth ? w.cell_max_iv[c] : 0u;  // (the spatial channel's own maxFanOutIntervalMs)
    uint32_t oo[CHD_OFF_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0};  // (off_on) the channel's sub-tick offsets, aligned to this tick
    for (uint32_t u = 0; u < n; u++) {
        if (chan[u] - g.id_start != c) continue;
        bool first_now = false;
        if (!touched) {
            const uint32_t age = cur_tick - w.cell_hist_tick[c];
            h = (age >= CHD_HIST_BITS) ? 0u : (w.cell_hist[c] << age);
            hp = (age >= CHD_HIST_BITS) ? 0u : (w.cell_hist_prev[c] << age);
            cur = w.cell_sender[c];
            prev = w.cell_sender_prev[c];
            touched = true;
            first_now = !((h | hp) & 1u);
            if (w.deep_depth) { dn = w.cdeep_n[c]; dlen = w.cdeep_len[c]; ddrop = w.cdeep_drop[c]; }
            if (w.off_on) {
                const uint4 a = w.cell_ooff[2 * (size_t)c], b = w.cell_ooff[2 * (size_t)c + 1];
                oo[0] = a.x; oo[1] = a.y; oo[2] = a.z; oo[3] = a.w; oo[4] = b.x; oo[5] = b.y; oo[6] = b.z; oo[7] = b.w;
                off_shift(oo, age);
            }
        }
        const uint32_t snd = sender[u];
        if (w.deep_depth) {
            const int64_t a = arrival ? arrival[u] : now;
            if (w.off_on) {
                // regular: inside the tick's own interval, and the channel's only stamp of this tick (WorldDev::off_on)
                const uint64_t off = (uint64_t)(now - a);
                if (!(a > w.prev_ns && a <= now && off <= 0xFFFFFFFEull)) irregular = true;
                if (!first_now && oo[0] != (uint32_t)off) irregular = true;
                oo[0] = (uint32_t)off;
            } else if (a != now) irregular = true;
            const size_t at = (size_t)c * w.deep_depth;
            deep_push(w.cdeep_a + at, w.cdeep_s + at, w.deep_depth, dn, dlen, ddrop, a, snd, max_iv);
        }
        if (snd != cur) {
            if (snd == prev) {
                const uint32_t t = h;
                h = hp;
                hp = t;
                prev = cur;
            } else {
                if (h != 0) {
                    if (hp != 0 && prev != cur) {
                        if (w.deep_depth) irregular = true;
                        else atomicAdd(&w.counters[CTR_SENDER_OVERFLOW], 1u);
                    }
                    hp |= h;
                    prev = cur;
                }
                h = 0;
            }
            cur = snd;
        }
        h |= 1u;
    }
    if (touched && w.deep_depth) {
        w.cdeep_n[c] = dn; w.cdeep_len[c] = dlen; w.cdeep_drop[c] = ddrop;
        // (cell_irr as well: on region-sharded worlds the spatial channels' updates arrive behind the index build, which is where the
        // flag is otherwise raised from cell_irr_tick)
        if (irregular) { w.cell_irr_tick[c] = cur_tick + 1u; w.cell_irr[c] = 1u; }
        if (w.off_on) {
            w.cell_ooff[2 * (size_t)c] = make_uint4(oo[0], oo[1], oo[2], oo[3]);
            w.cell_ooff[2 * (size_t)c + 1] = make_uint4(oo[4], oo[5], oo[6], oo[7]);
        }
    }
    if (touched) {
        w.cell_hist[c] = h;
        w.cell_hist_prev[c] = hp;
        w.cell_sender[c] = cur;
        w.cell_sender_prev[c] = prev;
        w.cell_hist_tick[c] = cur_tick;
    }
}

void launch_cell_updates(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan,
                         const uint32_t *sender, uint32_t cur_tick, const int64_t *arrival, int64_t now_ns) {
    if (!n) return;
    hipLaunchKernelGGL(k_cell_updates, dim3(nblocks(g.ncell, 256)), dim3(256), 0, st, g, w, n, chan, sender,
                       cur_tick, arrival, now_ns);
}
