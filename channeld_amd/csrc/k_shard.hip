// k_shard.hip — region-sharded worlds: entities live on the rank
// that owns their member cell (ServerIndex of GetRegions, spatial.go:336-351);
// crossing a region border is the reference's cross-server handover
// (spatial.go:683-700) and moves the entity's engine-side state to the other GPU.
//
//   k_ingest_by_channel : K1 in "pull" form — slots are library-managed here, so a
//                         live slot fetches its position by entity channel id
//   k_export            : member cell owned by another rank -> pack the 32-byte
//                         state into the per-destination send segment (header record
//                         = count), free the slot
//   k_import            : received states take free slots
//   k_halo_pack/_unpack : the border bands of the cell tables, per neighbour (see below)
// A few hundred entities migrate per tick (border crossings only), so these kernels
// are latency-trivial; the wire volume is what matters (32 B per emigrant).
#include "chd_kernels.h"

static inline unsigned nblocks(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

__global__ void __launch_bounds__(256) k_free_stack_init(WorldDev w) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < w.N) w.free_stack[i] = w.N - 1 - i;  // pops hand out slot 0, 1, 2, ...
    if (i == 0) *w.free_top = (int32_t)w.N;
}

void launch_free_stack_init(hipStream_t st, WorldDev w) {
    hipLaunchKernelGGL(k_free_stack_init, dim3(nblocks(w.N, 256)), dim3(256), 0, st, w);
}

// where channel `chan` lives on this rank (region-sharded handover lists)
__device__ __forceinline__ void note_slot(const WorldDev &w, uint32_t chan, uint32_t slot) {
    if (w.sh_slot_of) {
        const uint32_t k = chan - w.sh_eid0;
        if (k < w.sh_nchan) w.sh_slot_of[k] = slot;
    }
}

__global__ void __launch_bounds__(256) k_slot_of_rebuild(WorldDev w) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < w.N && (w.eflags[i] & EF_ALIVE)) note_slot(w, w.chan_id[i], i);
}

void launch_slot_of_rebuild(hipStream_t st, WorldDev w) {
    if (!w.N || !w.sh_slot_of) return;
    (void)hipMemsetAsync(w.sh_slot_of, 0xFF, sizeof(uint32_t) * (size_t)w.sh_nchan, st);
    hipLaunchKernelGGL(k_slot_of_rebuild, dim3(nblocks(w.N, 256)), dim3(256), 0, st, w);
}

__device__ __forceinline__ uint32_t pop_slot(const WorldDev &w) {
    int32_t k = atomicSub(w.free_top, 1) - 1;
    if (k < 0) {
        atomicAdd(w.free_top, 1);
        atomicOr(&w.counters[CTR_OVERFLOW], OVF_SLOTS);
        return CHD_INVALID;
    }
    return w.free_stack[k];
}

__global__ void __launch_bounds__(256) k_spawn_auto(DevGrid g, WorldDev w, uint32_t n,
                                                    const uint32_t *__restrict__ chan_id,
                                                    const double *__restrict__ x, const double *__restrict__ z,
                                                    const uint32_t *__restrict__ flags,
                                                    const uint32_t *__restrict__ sender, uint32_t cur_tick) {
    uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n) return;
    uint32_t i = pop_slot(w);
    if (i == CHD_INVALID) return;
    uint32_t c = cell_of(g, x[u], z[u]);
    w.chan_id[i] = chan_id[u];
    w.cell[i] = c;
    w.member[i] = c;
    w.eflags[i] = (flags ? (flags[u] & ~EF_ALIVE) : 0u) | EF_ALIVE;
    if (!w.log_on) {  // (log_on: the update log is the channel's, not the slot's — k_log_spawn)
        w.sender[i] = sender ? sender[u] : 0u;
        w.hist[i] = 0;
        w.hist_tick[i] = cur_tick;
        w.sender_prev[i] = 0;
        w.hist_prev[i] = 0;
    }
    note_slot(w, chan_id[u], i);
}

void launch_spawn_auto(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id,
                       const double *x, const double *z, const uint32_t *flags, const uint32_t *sender,
                       uint32_t cur_tick) {
    if (!n) return;
    hipLaunchKernelGGL(k_spawn_auto, dim3(nblocks(n, 256)), dim3(256), 0, st, g, w, n, chan_id, x, z, flags,
                       sender, cur_tick);
}

// chd_shard_despawn: the entity channels of the (ascending) list `gone` leave the world — the slot that holds one on this rank is freed,
// its update log (log_on) is closed, an immigrant still waiting in limbo is dropped from the list (its state record is blanked).
__device__ __forceinline__ bool in_sorted(const uint32_t *__restrict__ a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] == v) return true;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return false;
}

__global__ void __launch_bounds__(256) k_shard_despawn(WorldDev w, const uint32_t *__restrict__ gone, uint32_t n) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < w.N && (w.eflags[t] & EF_ALIVE) && in_sorted(gone, n, w.chan_id[t])) {
        note_slot(w, w.chan_id[t], CHD_INVALID);
        w.eflags[t] = 0;
        w.member[t] = CHD_INVALID;
        w.cell[t] = CHD_INVALID;
        const int32_t f = atomicAdd(w.free_top, 1);
        w.free_stack[f] = t;
    }
    if (t < n && w.log_on) {
        const uint32_t u = gone[t] - w.log_eid0;
        if (u < w.log_n) { w.log_alive[u] = 0; w.log_cell[u] = CHD_INVALID; }
    }
    if (w.limbo)
        for (uint32_t b = 0; b < 2; b++)
            for (uint32_t q = t; q < min(w.limbo_n[b], w.N); q += gridDim.x * 256u)
                if (in_sorted(gone, n, w.limbo[(size_t)b * w.N + q].chan_id)) w.limbo[(size_t)b * w.N + q].chan_id = CHD_INVALID;
}

void launch_shard_despawn(hipStream_t st, WorldDev w, const uint32_t *gone, uint32_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_shard_despawn, dim3(nblocks(std::max(w.N, n), 256)), dim3(256), 0, st, w, gone, n);
}

// Same decision as k_ingest (spatial.go:612-626,675-679,703-736), one thread per slot.
__global__ void __launch_bounds__(256) k_ingest_by_channel(DevGrid g, WorldDev w,
                                                           const double *__restrict__ xs,
                                                           const double *__restrict__ zs,
                                                           const uint8_t *__restrict__ has_update,
                                                           uint32_t n_chan, uint32_t entity_id_start,
                                                           uint32_t cur_tick, uint32_t rank, uint32_t world,
                                                           uint4 *__restrict__ req_send, uint32_t req_cap) {
    __shared__ uint32_t s_cnt[4], s_lock[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    bool ho = false, locked = false;
    uint32_t src = CHD_INVALID, dst = CHD_INVALID;
    if (i < w.N) {
        const uint32_t ef = w.eflags[i];
        const uint32_t k = w.chan_id[i] - entity_id_start;
        if ((ef & EF_ALIVE) && k < n_chan && (!has_update || has_update[k])) {
            dst = cell_of(g, xs[k], zs[k]);
            src = w.cell[i];
            w.cell[i] = dst;
            // (who sent the update: the owner changes with a cross-server handover — chd_shard_set_update_senders — else the entity's own)
            // log_on: every rank logs every channel's update, this one's included (k_log_push, after the emigrant exchange)
            if (!w.log_on) push_update(w, i, (w.sh_sender_by_chan && k < w.sh_sender_n) ? w.sh_sender_by_chan[k] : w.sender[i], cur_tick, src, dst);
            if (src != CHD_INVALID && dst != CHD_INVALID && src != dst) {
                // GetHandoverEntities (entity.go:197-224) as the host's group controllers evaluated it: an EMPTY list = a locked
                // member or an emptied group -> no handover (spatial.go:675-679)
                bool lk = (ef & EF_LOCKED) != 0;
                if (w.sh_list_of && k < w.sh_nchan) {
                    const uint32_t li = w.sh_list_of[k];
                    if (li != CHD_NO_HANDOVER_LIST && w.sh_list_off[li + 1] == w.sh_list_off[li]) lk = true;
                }
                if (lk) locked = true;
                else ho = true;
            }
        }
    }
    const uint64_t hm = __ballot(ho), lm = __ballot(locked);
    if (lane == 0) { s_cnt[wave] = (uint32_t)__popcll(hm); s_lock[wave] = (uint32_t)__popcll(lm); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, ltot = 0;
        for (int k = 0; k < 4; k++) { tot += s_cnt[k]; ltot += s_lock[k]; }
        uint32_t base = tot ? atomicAdd(&w.counters[CTR_HANDOVERS], tot) : 0u;
        if (ltot) atomicAdd(&w.counters[CTR_LOCKED], ltot);
        for (int k = 0; k < 4; k++) { uint32_t c = s_cnt[k]; s_cnt[k] = base; base += c; }
    }
    __syncthreads();
    if (ho) {
        bool self_moves = true;
        if (w.sh_list_of) {
            // every list member that is in src's entity map moves to dst's (spatial.go:703-736 over handoverEntities); the map of
            // src is this rank's, so its members have slots here (a member that lives on another rank is in another cell's map)
            const uint32_t kk = w.chan_id[i] - w.sh_eid0;
            const uint32_t li = kk < w.sh_nchan ? w.sh_list_of[kk] : CHD_NO_HANDOVER_LIST;
            if (li != CHD_NO_HANDOVER_LIST) {
                self_moves = false;  // (the notifier itself only if its list names it)
                // src is the cell of the notifier's last POSITION; the notifier lives on the rank of the cell whose MAP holds it.
                // After another member's handover pulled it across a region border the two differ: src's map — the members to
                // move — is then another rank's, and the handover travels there as a 16-byte request (k_apply_requests) before
                // anything is exported.
                const uint32_t own = server_of(g, src);
                const bool remote = world > 1 && own != rank && own < world;
                for (uint32_t q = w.sh_list_off[li]; q < w.sh_list_off[li + 1]; q++) {
                    const uint32_t mc = w.sh_list_mem[q];
                    if (mc == w.chan_id[i]) { self_moves = true; continue; }
                    if (remote) continue;
                    const uint32_t mk = mc - w.sh_eid0;
                    const uint32_t j = mk < w.sh_nchan ? w.sh_slot_of[mk] : CHD_INVALID;
                    if (j < w.N && (w.eflags[j] & EF_ALIVE) && w.chan_id[j] == mc) atomicCAS(&w.member[j], src, dst);
                }
                if (remote) {
                    uint4 *seg = req_send + (size_t)own * (req_cap + 1);  // record 0 = {count, 0, 0, 0}
                    const uint32_t r = req_send ? atomicAdd(&seg[0].x, 1u) : req_cap;
                    if (r < req_cap) seg[1 + r] = make_uint4(li, src, dst, w.chan_id[i]);
                    else atomicOr(&w.counters[CTR_OVERFLOW], OVF_MIGRATE);
                }
            }
        }
        if (self_moves) w.member[i] = dst;
        uint32_t pos = s_cnt[wave] + mask_rank(hm);
        if (pos < w.handovers_cap) {
            chd_handover_rec r;
            r.entity = i;
            r.channel = w.chan_id[i];
            r.src = src + g.id_start;
            r.dst = dst + g.id_start;
            r.src_server = server_of(g, src);
            r.dst_server = server_of(g, dst);
            w.handovers[pos] = r;
        } else {
            atomicOr(&w.counters[CTR_OVERFLOW], OVF_HANDOVER);
        }
    }
}

void launch_ingest_by_channel(hipStream_t st, DevGrid g, WorldDev w, const double *x_by_chan,
                              const double *z_by_chan, const uint8_t *has_update, uint32_t n_chan,
                              uint32_t entity_id_start, uint32_t cur_tick, uint32_t rank, uint32_t world, uint4 *req_send,
                              uint32_t req_cap) {
    if (req_send) (void)hipMemsetAsync(req_send, 0, sizeof(uint4) * (size_t)world * (req_cap + 1), st);
    if (!w.N || !n_chan) return;
    hipLaunchKernelGGL(k_ingest_by_channel, dim3(nblocks(w.N, 256)), dim3(256), 0, st, g, w, x_by_chan, z_by_chan,
                       has_update, n_chan, entity_id_start, cur_tick, rank, world, req_send, req_cap);
}

// The other ranks' handovers whose src cell is one of OURS: the list members in src's map follow to dst (they leave with this
// tick's export if dst is not ours either).  One thread per request; a tick carries a handful.
__global__ void __launch_bounds__(256) k_apply_requests(WorldDev w, const uint4 *__restrict__ req_recv, uint32_t world, uint32_t req_cap) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t from = t / req_cap, r = t % req_cap;
    if (from >= world) return;
    const uint4 *seg = req_recv + (size_t)from * (req_cap + 1);
    if (r >= min(seg[0].x, req_cap)) return;
    const uint4 rq = seg[1 + r];  // {list, src, dst, notifier channel}
    if (!w.sh_list_of || rq.x >= w.sh_nlists) return;
    for (uint32_t q = w.sh_list_off[rq.x]; q < w.sh_list_off[rq.x + 1]; q++) {
        const uint32_t mc = w.sh_list_mem[q];
        if (mc == rq.w) continue;
        const uint32_t mk = mc - w.sh_eid0;
        const uint32_t j = mk < w.sh_nchan ? w.sh_slot_of[mk] : CHD_INVALID;
        if (j < w.N && (w.eflags[j] & EF_ALIVE) && w.chan_id[j] == mc) atomicCAS(&w.member[j], rq.y, rq.z);
    }
}

void launch_apply_requests(hipStream_t st, WorldDev w, const uint4 *req_recv, uint32_t world, uint32_t req_cap) {
    if (!req_recv || !world || !req_cap) return;
    hipLaunchKernelGGL(k_apply_requests, dim3(nblocks((uint64_t)world * req_cap, 256)), dim3(256), 0, st, w, req_recv, world, req_cap);
}

__global__ void __launch_bounds__(256) k_export(DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                                                chd_entity_state *__restrict__ send, uint32_t cap,
                                                uint32_t cur_tick, uint32_t extra) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    // (the limbo list this tick's import appends to starts empty: see k_import)
    if (i == 0) w.limbo_n[(cur_tick + 1u) & 1u] = 0;
    if (i >= w.N) return;
    const uint32_t ef = w.eflags[i];
    if (!(ef & EF_ALIVE)) return;
    const uint32_t m = w.member[i];
    if (m == CHD_INVALID) return;  // in no cell: visible to nobody, stays where it is
    const uint32_t dst = server_of(g, m);
    if (dst == rank || dst >= world) return;
    chd_entity_state *seg = send + (size_t)dst * ((size_t)cap + 1 + extra);  // record 0 = header, chan_id = count
    const uint32_t k = atomicAdd(&seg[0].chan_id, 1u);
    if (k >= cap) {
        // no room in this tick's segment: stay (still owned here, still wrong owner) and retry next tick
        atomicSub(&seg[0].chan_id, 1u);
        atomicOr(&w.counters[CTR_OVERFLOW], OVF_MIGRATE);
        return;
    }
    chd_entity_state e;
    e.chan_id = w.chan_id[i];
    e.cell = w.cell[i];
    e.member = m;
    e.eflags = ef;
    if (w.log_on) { e.sender = e.hist = e.hist_prev = e.sender_prev = 0; }  // (the destination has the channel's log already)
    else {
        e.sender = w.sender[i];
        const uint32_t age = cur_tick - w.hist_tick[i];
        e.hist = age >= CHD_HIST_BITS ? 0u : (w.hist[i] << age);
        e.hist_prev = age >= CHD_HIST_BITS ? 0u : (w.hist_prev[i] << age);
        e.sender_prev = w.sender_prev[i];
    }
    seg[1 + k] = e;
    note_slot(w, e.chan_id, CHD_INVALID);
    w.eflags[i] = 0;
    w.member[i] = CHD_INVALID;
    w.cell[i] = CHD_INVALID;
    const int32_t f = atomicAdd(w.free_top, 1);
    w.free_stack[f] = i;
}

// Every header also carries the LARGEST count this rank put into any of its segments (field `cell` of record 0): after the
// all-to-all every rank holds every source's maximum, i.e. the same global maximum — what the adaptive segment capacity of
// the following ticks is derived from (chd_shard_ingest, cap_used), identically on every rank and without a collective.
// extra > 0 (WorldDev::log_on): every segment is followed by `extra` more records that carry THIS rank's maxFanOutIntervalMs per
// spatial channel (cell_max_iv's in-force half: stable during the tick — the interest updates raise the other half), so that after
// the exchange every rank folds the same world-wide maxima before it logs the tick's updates (k_import, k_log_push).
__global__ void __launch_bounds__(64) k_export_finish(WorldDev w, chd_entity_state *__restrict__ send, uint32_t world, uint32_t cap, uint32_t extra,
                                                      uint32_t ncell) {
    const size_t pitch = (size_t)cap + 1 + extra;
    uint32_t m = 0;
    for (uint32_t d = threadIdx.x; d < world; d += 64) m = max(m, min(send[d * pitch].chan_id, cap));
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    for (uint32_t d = threadIdx.x; d < world; d += 64) send[d * pitch].cell = m;
    if (extra)
        for (uint32_t d = 0; d < world; d++) {
            uint32_t *x = (uint32_t *)(send + d * pitch + cap + 1);
            for (uint32_t c = threadIdx.x; c < extra * 8u; c += 64) x[c] = c < ncell ? w.cell_max_iv[c] : 0u;
        }
}

void launch_export(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                   chd_entity_state *send, uint32_t cap, uint32_t cur_tick, uint32_t extra) {
    // zero the segment headers (32 bytes at a pitch of (cap + 1 + extra) records)
    (void)hipMemset2DAsync(send, sizeof(chd_entity_state) * ((size_t)cap + 1 + extra), 0, sizeof(chd_entity_state), world, st);
    if (!w.N) return;
    hipLaunchKernelGGL(k_export, dim3(nblocks(w.N, 256)), dim3(256), 0, st, g, w, rank, world, send, cap, cur_tick, extra);
    hipLaunchKernelGGL(k_export_finish, dim3(1), dim3(64), 0, st, w, send, world, cap, extra, g.ncell);
}

__device__ __forceinline__ void install_entity(const WorldDev &w, uint32_t i, const chd_entity_state &e, uint32_t cur_tick) {
    w.chan_id[i] = e.chan_id;
    w.cell[i] = e.cell;
    w.member[i] = e.member;
    w.eflags[i] = e.eflags | EF_ALIVE;
    if (!w.log_on) {
        w.sender[i] = e.sender;
        w.hist[i] = e.hist;
        w.hist_tick[i] = cur_tick;
        w.hist_prev[i] = e.hist_prev;
        w.sender_prev[i] = e.sender_prev;
    }
    note_slot(w, e.chan_id, i);
}

// An immigrant that finds no free slot is not lost: it waits in LIMBO (a side list of states, two buffers by tick parity)
// and is offered a slot again at every later import, before the tick's own immigrants.  While it waits it is in no cell
// table (visible to nobody) and its history does not advance: pop_slot has set overflow bit 16 for this tick, and does so
// again every tick the entity is still waiting — the host sees a capacity error, never a silently shrinking world.
__device__ __forceinline__ void to_limbo(const WorldDev &w, const chd_entity_state &e, uint32_t cur_tick) {
    const uint32_t nb = (cur_tick + 1u) & 1u;
    const uint32_t k = atomicAdd(&w.limbo_n[nb], 1u);
    if (k < w.N) w.limbo[(size_t)nb * w.N + k] = e;
    else { atomicSub(&w.limbo_n[nb], 1u); atomicOr(&w.counters[CTR_OVERFLOW], OVF_LOST); }
}

// blockIdx.y < world: the segment rank y sent; blockIdx.y == world: the entities waiting in limbo since earlier ticks.
// Also folds the global maximum segment count of this tick's exchange (k_export_finish) into mig_gmax[cur_tick & 3].
__global__ void __launch_bounds__(256) k_import(WorldDev w, const chd_entity_state *__restrict__ recv,
                                                uint32_t world, uint32_t cap, uint32_t cur_tick, uint32_t extra, uint32_t ncell) {
    const uint32_t src = blockIdx.y;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const size_t pitch = (size_t)cap + 1 + extra;
    if (src == world) {
        // log_on: the world-wide maxFanOutIntervalMs of every spatial channel = the maximum over the ranks' own (k_export_finish)
        if (extra)
            for (uint32_t c = k; c < ncell; c += gridDim.x * 256u) {
                uint32_t m = w.cell_max_iv[c];
                for (uint32_t s = 0; s < world; s++) m = max(m, ((const uint32_t *)(recv + s * pitch + cap + 1))[c]);
                w.cell_max_iv[c] = m;
            }
        const uint32_t cb = cur_tick & 1u;
        const uint32_t n = min(w.limbo_n[cb], w.N);
        for (uint32_t q = k; q < n; q += gridDim.x * 256u) {
            chd_entity_state e = w.limbo[(size_t)cb * w.N + q];
            if (e.chan_id == CHD_INVALID) continue;  // (despawned while it waited: chd_shard_despawn)
            // (the histories were aligned to the tick of the export and have aged one tick per tick in limbo)
            e.hist <<= 1; e.hist_prev <<= 1;
            const uint32_t i = pop_slot(w);
            if (i == CHD_INVALID) to_limbo(w, e, cur_tick);
            else install_entity(w, i, e, cur_tick);
        }
        if (k == 0) {
            uint32_t m = 0;
            for (uint32_t s = 0; s < world; s++) m = max(m, recv[s * pitch].cell);
            w.mig_gmax[cur_tick & 3u] = m;
        }
        return;
    }
    const chd_entity_state *seg = recv + src * pitch;
    const uint32_t n = min(seg[0].chan_id, cap);
    if (k >= n) return;
    const chd_entity_state e = seg[1 + k];
    const uint32_t i = pop_slot(w);
    if (i == CHD_INVALID) { to_limbo(w, e, cur_tick); return; }
    install_entity(w, i, e, cur_tick);
}

void launch_import(hipStream_t st, WorldDev w, const chd_entity_state *recv, uint32_t world, uint32_t cap,
                   uint32_t cur_tick, uint32_t extra, uint32_t ncell) {
    if (!world || !cap) return;
    hipLaunchKernelGGL(k_import, dim3(nblocks(cap, 256), world + 1), dim3(256), 0, st, w, recv, world, cap, cur_tick, extra, ncell);
}

// ---------------------------------------------------------------------------
// The update log by channel (WorldDev::log_on).  In the reference a channel's update buffer, its senders and its
// maxFanOutIntervalMs live with the CHANNEL in the one gateway process (data.go:53-55); which spatial server owns the entity is
// another matter (spatial.go:683-700).  Every rank of a region-sharded world is fed the same update stream by channel id, so
// every rank keeps every channel's log: nothing of it has to travel with an emigrant or a border band.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_log_spawn(DevGrid g, WorldDev w, uint32_t n, const uint32_t *__restrict__ chan_id,
                                                   const double *__restrict__ x, const double *__restrict__ z) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t u = chan_id[t] - w.log_eid0;
    if (u >= w.log_n) { atomicOr(&w.counters[CTR_OVERFLOW], OVF_SLOTS); return; }
    w.log_cell[u] = cell_of(g, x[t], z[t]);
    w.log_alive[u] = 1u;
    // a new channel: an empty buffer, no history (a re-used channel id starts over)
    w.hist[u] = 0; w.hist_prev[u] = 0; w.sender[u] = 0; w.sender_prev[u] = 0;
    w.deep_n[u] = 0; w.deep_len[u] = 0; w.deep_drop[u] = INT64_MIN; w.irr_tick[u] = 0; w.ent_max_iv[u] = 0;
    if (w.off_on) { w.eoff[2 * (size_t)u] = make_uint4(0, 0, 0, 0); w.eoff[2 * (size_t)u + 1] = make_uint4(0, 0, 0, 0); }
}

void launch_log_spawn(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id, const double *x, const double *z) {
    if (!n) return;
    hipLaunchKernelGGL(k_log_spawn, dim3(nblocks(n, 256)), dim3(256), 0, st, g, w, n, chan_id, x, z);
}

// ChannelData.OnUpdate (data.go:149-173) for this tick's update of EVERY channel of the world, on every rank alike: the Notify
// decision's two cells (for the channel's maxFanOutIntervalMs, push_update), the sender and the arrival stamp by channel id.
__global__ void __launch_bounds__(256) k_log_push(DevGrid g, WorldDev w, const double *__restrict__ xs, const double *__restrict__ zs,
                                                  const uint8_t *__restrict__ has_update, uint32_t n_chan, uint32_t cur_tick, int64_t now) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n_chan || u >= w.log_n || !w.log_alive[u] || (has_update && !has_update[u])) return;
    const uint32_t c_new = cell_of(g, xs[u], zs[u]), c_old = w.log_cell[u];
    w.log_cell[u] = c_new;
    const uint32_t snd = (w.sh_sender_by_chan && u < w.sh_sender_n) ? w.sh_sender_by_chan[u] : 0u;
    const int64_t a = (w.sh_arrival_by_chan && u < w.sh_arrival_n) ? w.sh_arrival_by_chan[u] : now;
    push_update(w, u, snd, cur_tick, c_old, c_new, a, now);
}

void launch_log_push(hipStream_t st, DevGrid g, WorldDev w, const double *x_by_chan, const double *z_by_chan, const uint8_t *has_update,
                     uint32_t n_chan, uint32_t cur_tick, int64_t now_ns) {
    if (!w.log_on || !n_chan) return;
    hipLaunchKernelGGL(k_log_push, dim3(nblocks(n_chan, 256)), dim3(256), 0, st, g, w, x_by_chan, z_by_chan, has_update, n_chan, cur_tick, now_ns);
}

// ---------------------------------------------------------------------------
// Halo exchange (SURVEY 8e, C2 "border halo"): instead of every rank's whole cell table, rank s sends rank d only the
// cells of its region within `halo` cells of d's region (chd_device.h: halo_rect) — one fixed-capacity segment per
// (s, d) pair, exchanged with ONE all-to-all(v) whose split sizes are static (ranks further apart than the halo
// exchange nothing).  The receiver appends the entries as GHOSTS behind its own cell-sorted tables, so the fan-out
// kernels see one local table covering region + halo — the same arrays, the same fast paths as a single-GPU world.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_halo_pack(DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo,
                                                   unsigned char *__restrict__ send, const uint64_t *__restrict__ seg_off) {
    const uint32_t d = blockIdx.x;
    const HaloRect r = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, halo, rank, d);
    const uint32_t nc = r.w * r.h;
    if (!nc) return;
    const uint32_t cap = halo_cap_entries(w.N, nc, g.sgc * g.sgr);
    unsigned char *seg = send + seg_off[d];
    uint32_t *hdr = (uint32_t *)seg;
    uint4 *agg = (uint4 *)(seg + 16);
    uint4 *ent = (uint4 *)(seg + 16 + 16ull * nc);
    uint32_t *cnt = (uint32_t *)(seg + 16 + 16ull * nc + 16ull * cap);
    uint32_t *sprev = (uint32_t *)((unsigned char *)cnt + ((4ull * nc + 15ull) & ~15ull));
    __shared__ uint32_t row_base[1024];  // entries in front of each rect row (rows <= 1024: grids are far smaller per axis band)
    __shared__ uint32_t total_s;
    // per cell: count + aggregates; per row: its entries are one contiguous run of the cell-sorted table
    for (uint32_t k = threadIdx.x; k < nc; k += 256) {
        const uint32_t c = (r.x0 + k % r.w) + (r.y0 + k / r.w) * g.cols;
        cnt[k] = w.cell_off[c + 1] - w.cell_off[c];
        agg[k] = make_uint4(w.cell_usender[c], w.cell_smin[c], w.cell_smax[c], w.cell_hand[c]);
    }
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t y = 0; y < r.h; y++) {
            const uint32_t c0 = r.x0 + (r.y0 + y) * g.cols;
            if (y < 1024) row_base[y] = acc;
            acc += w.cell_off[c0 + r.w] - w.cell_off[c0];
        }
        total_s = acc;
    }
    __syncthreads();
    const uint32_t total = total_s;
    const bool ovf = total > cap || r.h > 1024;
    if (threadIdx.x == 0) {
        hdr[0] = ovf ? 0u : total; hdr[1] = nc; hdr[2] = ovf ? 1u : 0u; hdr[3] = 0u;
        if (ovf) atomicOr(&w.counters[CTR_OVERFLOW], OVF_HALO);
    }
    if (ovf) return;
    for (uint32_t y = 0; y < r.h; y++) {
        const uint32_t c0 = r.x0 + (r.y0 + y) * g.cols;
        const uint32_t a = w.cell_off[c0], n = w.cell_off[c0 + r.w] - a, base = row_base[y];
        for (uint32_t k = threadIdx.x; k < n; k += 256) {
            ent[base + k] = w.ce[a + k];
            sprev[base + k] = w.ce_sprev[a + k];
        }
    }
}

void launch_halo_pack(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo, unsigned char *send,
                      const uint64_t *seg_off) {
    if (world < 2) return;
    hipLaunchKernelGGL(k_halo_pack, dim3(world), dim3(256), 0, st, g, w, rank, world, halo, send, seg_off);
}

// block s < world: the segment rank s sent; block `world`: this rank's own cells and the cells nobody covers
__global__ void __launch_bounds__(256) k_halo_unpack(DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo,
                                                     const unsigned char *__restrict__ recv, const uint64_t *__restrict__ seg_off,
                                                     const uint32_t *__restrict__ ghost_off, uint32_t cur_tick, const unsigned long long *gate_p,
                                                     unsigned long long gate_target) {
    const uint32_t s = blockIdx.x;
    if (s == world) {
        // own region: the local index; cells of no received band: empty and NOT covered (a subscription to one of them is
        // an error the plan kernels report: the halo is narrower than that connection's AOI reach)
        for (uint32_t c = threadIdx.x; c < g.ncell; c += 256) {
            const uint32_t o = server_of(g, c);
            if (o == rank) {
                w.cell_tab[c] = w.cell_off[c];
                w.cell_tab[g.ncell + c] = w.cell_off[c + 1];
                w.cell_cov[c] = 1;
            } else {
                const HaloRect r = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, halo, o, rank);
                const uint32_t x = c % g.cols, y = c / g.cols;
                const bool in = o < world && r.w && x >= r.x0 && x < r.x0 + r.w && y >= r.y0 && y < r.y0 + r.h;
                if (!in) { w.cell_tab[c] = 0; w.cell_tab[g.ncell + c] = 0; w.cell_cov[c] = 0; }
            }
        }
        // chd_shard_tick with the interest updates on the second stream: this launch also holds the tick's stream until they
        // are complete (GateArgs; as k_index_scatter does on unsharded worlds) — the plan that follows needs them
        if (gate_p && threadIdx.x == 0) {
            uint32_t spins = 0;
            while (__hip_atomic_load(gate_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 23)) { gate_timed_out(w); break; }
            }
        }
        return;
    }
    const HaloRect r = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, halo, s, rank);
    const uint32_t nc = r.w * r.h;
    if (!nc) return;
    const uint32_t cap = halo_cap_entries(w.N, nc, g.sgc * g.sgr);
    const unsigned char *seg = recv + seg_off[s];
    const uint32_t *hdr = (const uint32_t *)seg;
    const uint4 *agg = (const uint4 *)(seg + 16);
    const uint4 *ent = (const uint4 *)(seg + 16 + 16ull * nc);
    const uint32_t *cnt = (const uint32_t *)(seg + 16 + 16ull * nc + 16ull * cap);
    const uint32_t *sprev = (const uint32_t *)((const unsigned char *)cnt + ((4ull * nc + 15ull) & ~15ull));
    const uint32_t base = w.N + ghost_off[s];  // ghosts of rank s: ce[base, base + cap)
    // a segment is only taken when it is consistent: the sender's flag, the cell count of the rectangle, and per-cell counts
    // that add up to the entry count it announces, inside the capacity both sides derive from the SAME max_entities
    __shared__ uint32_t total_s, bad_s;
    if (threadIdx.x == 0) {
        uint32_t b = (hdr[2] != 0 || hdr[1] != nc || hdr[0] > cap) ? 1u : 0u;
        if (!b) {
            uint64_t sum = 0;
            for (uint32_t k = 0; k < nc; k++) sum += cnt[k];
            if (sum != hdr[0]) b = 1u;
        }
        bad_s = b;
        if (b) atomicOr(&w.counters[CTR_OVERFLOW], OVF_HALO);
    }
    __syncthreads();
    const bool bad = bad_s != 0;
    // cell views: exclusive prefix of the counts (serial: a band has a few hundred cells)
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < nc; k++) {
            const uint32_t c = (r.x0 + k % r.w) + (r.y0 + k / r.w) * g.cols;
            const uint32_t n = bad ? 0u : cnt[k];
            w.cell_tab[c] = base + acc;
            w.cell_tab[g.ncell + c] = base + acc + n;
            w.cell_cov[c] = bad ? 0u : 1u;
            if (w.cell_sorted) w.cell_sorted[c] = 0;  // (a neighbour's band: its windows are tested per entity)
            acc += n;
        }
        total_s = acc;
    }
    for (uint32_t k = threadIdx.x; k < nc; k += 256) {
        const uint32_t c = (r.x0 + k % r.w) + (r.y0 + k / r.w) * g.cols;
        const uint4 a = agg[k];
        w.cell_usender[c] = a.x; w.cell_smin[c] = a.y; w.cell_smax[c] = a.z; w.cell_hand[c] = a.w;
    }
    __syncthreads();
    const uint32_t total = total_s;
    for (uint32_t k = threadIdx.x; k < total; k += 256) {
        const uint4 e = ent[k];
        w.ce[base + k] = e;
        w.ce_sprev[base + k] = sprev[k];
        w.ce8[base + k] = make_uint2(e.x, e.y | e.w);  // compact entry {channel, history of any sender}
        w.ce_chan[base + k] = e.x;
        if (w.ce_by_chan && w.ce_slot) {
            // the ghost's update log (log_on) and its wire payloads (CHD_WORLD_WIRE) are HERE, by channel: where the exact buffers
            // and the payload slots are, and its sub-tick offsets into the columns
            const uint32_t u = e.x - w.log_eid0;
            w.ce_slot[base + k] = u;
            if (w.log_on && u < w.log_n) {
                if (w.off_on) scatter_offsets(w, u, base + k, cur_tick - w.hist_tick[u]);
                // ... and a ghost cell that holds a channel the tick-ring masks cannot answer for is flagged as the index build flags
                // its own cells (cell_irr).  Rare: the entry's cell by a binary search over the band's cell starts (ascending in
                // rectangle order; written by thread 0 before the barrier above)
                const uint32_t it = w.irr_tick[u];
                if (it && cur_tick + 1u - it < CHD_HIST_BITS) {
                    uint32_t lo = 0, hi = nc;  // last cell whose first entry is <= k
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        const uint32_t cm = (r.x0 + mid % r.w) + (r.y0 + mid / r.w) * g.cols;
                        if (w.cell_tab[cm] - base <= k) lo = mid; else hi = mid;
                    }
                    w.cell_irr[(r.x0 + lo % r.w) + (r.y0 + lo / r.w) * g.cols] = 1u;
                }
            }
        }
    }
}

void launch_halo_unpack(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world, uint32_t halo, const unsigned char *recv,
                        const uint64_t *seg_off, const uint32_t *ghost_off, uint32_t cur_tick, const unsigned long long *gate_p, unsigned long long gate_target) {
    hipLaunchKernelGGL(k_halo_unpack, dim3(world + 1), dim3(256), 0, st, g, w, rank, world, halo, recv, seg_off, ghost_off, cur_tick, gate_p, gate_target);
}
