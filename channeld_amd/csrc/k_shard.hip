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
//   k_cell_table        : after the all-gather, cell c's entries are the owner
//                         rank's [cell_off_o[c], cell_off_o[c+1]) inside its segment
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
    w.sender[i] = sender ? sender[u] : 0u;
    w.hist[i] = 0;
    w.hist_tick[i] = cur_tick;
    w.sender_prev[i] = 0;
    w.hist_prev[i] = 0;
}

void launch_spawn_auto(hipStream_t st, DevGrid g, WorldDev w, uint32_t n, const uint32_t *chan_id,
                       const double *x, const double *z, const uint32_t *flags, const uint32_t *sender,
                       uint32_t cur_tick) {
    if (!n) return;
    hipLaunchKernelGGL(k_spawn_auto, dim3(nblocks(n, 256)), dim3(256), 0, st, g, w, n, chan_id, x, z, flags,
                       sender, cur_tick);
}

// Same decision as k_ingest (spatial.go:612-626,675-679,703-736), one thread per slot.
__global__ void __launch_bounds__(256) k_ingest_by_channel(DevGrid g, WorldDev w,
                                                           const double *__restrict__ xs,
                                                           const double *__restrict__ zs,
                                                           const uint8_t *__restrict__ has_update,
                                                           uint32_t n_chan, uint32_t entity_id_start,
                                                           uint32_t cur_tick) {
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
            push_update(w, i, w.sender[i], cur_tick);
            if (src != CHD_INVALID && dst != CHD_INVALID && src != dst) {
                if (ef & EF_LOCKED) locked = true;
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
        w.member[i] = dst;
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
                              uint32_t entity_id_start, uint32_t cur_tick) {
    if (!w.N || !n_chan) return;
    hipLaunchKernelGGL(k_ingest_by_channel, dim3(nblocks(w.N, 256)), dim3(256), 0, st, g, w, x_by_chan, z_by_chan,
                       has_update, n_chan, entity_id_start, cur_tick);
}

__global__ void __launch_bounds__(256) k_export(DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                                                chd_entity_state *__restrict__ send, uint32_t cap,
                                                uint32_t cur_tick) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w.N) return;
    const uint32_t ef = w.eflags[i];
    if (!(ef & EF_ALIVE)) return;
    const uint32_t m = w.member[i];
    if (m == CHD_INVALID) return;  // in no cell: visible to nobody, stays where it is
    const uint32_t dst = server_of(g, m);
    if (dst == rank || dst >= world) return;
    chd_entity_state *seg = send + (size_t)dst * (cap + 1);  // record 0 = header, chan_id = count
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
    e.sender = w.sender[i];
    const uint32_t age = cur_tick - w.hist_tick[i];
    e.hist = age >= CHD_HIST_BITS ? 0u : (w.hist[i] << age);
    e.hist_prev = age >= CHD_HIST_BITS ? 0u : (w.hist_prev[i] << age);
    e.sender_prev = w.sender_prev[i];
    seg[1 + k] = e;
    w.eflags[i] = 0;
    w.member[i] = CHD_INVALID;
    w.cell[i] = CHD_INVALID;
    const int32_t f = atomicAdd(w.free_top, 1);
    w.free_stack[f] = i;
}

void launch_export(hipStream_t st, DevGrid g, WorldDev w, uint32_t rank, uint32_t world,
                   chd_entity_state *send, uint32_t cap, uint32_t cur_tick) {
    // zero the segment headers (32 bytes at a pitch of (cap+1) records)
    (void)hipMemset2DAsync(send, sizeof(chd_entity_state) * ((size_t)cap + 1), 0, sizeof(chd_entity_state), world, st);
    if (!w.N) return;
    hipLaunchKernelGGL(k_export, dim3(nblocks(w.N, 256)), dim3(256), 0, st, g, w, rank, world, send, cap, cur_tick);
}

__global__ void __launch_bounds__(256) k_import(WorldDev w, const chd_entity_state *__restrict__ recv,
                                                uint32_t world, uint32_t cap, uint32_t cur_tick) {
    const uint32_t src = blockIdx.y;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (src >= world) return;
    const chd_entity_state *seg = recv + (size_t)src * (cap + 1);
    const uint32_t n = min(seg[0].chan_id, cap);
    if (k >= n) return;
    const chd_entity_state e = seg[1 + k];
    const uint32_t i = pop_slot(w);
    if (i == CHD_INVALID) return;
    w.chan_id[i] = e.chan_id;
    w.cell[i] = e.cell;
    w.member[i] = e.member;
    w.eflags[i] = e.eflags | EF_ALIVE;
    w.sender[i] = e.sender;
    w.hist[i] = e.hist;
    w.hist_tick[i] = cur_tick;
    w.hist_prev[i] = e.hist_prev;
    w.sender_prev[i] = e.sender_prev;
}

void launch_import(hipStream_t st, WorldDev w, const chd_entity_state *recv, uint32_t world, uint32_t cap,
                   uint32_t cur_tick) {
    if (!world || !cap) return;
    hipLaunchKernelGGL(k_import, dim3(nblocks(cap, 256), world), dim3(256), 0, st, w, recv, world, cap, cur_tick);
}

__global__ void __launch_bounds__(256) k_cell_table(DevGrid g, WorldDev w, const unsigned char *__restrict__ tables,
                                                    uint32_t world, uint64_t table_bytes) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= g.ncell) return;
    uint32_t o = server_of(g, c);
    uint32_t a = 0, b = 0;
    if (o < world) {
        const uint32_t *off = (const uint32_t *)(tables + (size_t)o * table_bytes + (sizeof(uint4) + sizeof(uint32_t)) * (size_t)w.N);
        const uint32_t base = (uint32_t)((size_t)o * (table_bytes / sizeof(uint4)));  // in 16-byte entries
        a = base + off[c];
        b = base + off[c + 1];
    }
    w.cell_tab[c] = a;
    w.cell_tab[g.ncell + c] = b;
}

void launch_cell_table(hipStream_t st, DevGrid g, WorldDev w, const void *tables, uint32_t world,
                       uint64_t table_bytes) {
    hipLaunchKernelGGL(k_cell_table, dim3(nblocks(g.ncell, 256)), dim3(256), 0, st, g, w,
                       (const unsigned char *)tables, world, table_bytes);
}
