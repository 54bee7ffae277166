// k_recipients.hip — recipient planning for the messages the reference assembles around
// the hot path (decision parts; protobuf assembly stays on the host).
//
//   handover recipients  spatial.go:776-857: for handover (src cell, dst cell) every
//                        connection subscribed to src only gets the message without
//                        per-recipient entity data (:780-787); every connection subscribed
//                        to dst gets it with entity data — full state, plus a subscription
//                        to the entity channel, if it was not yet subscribed to that channel
//                        (:797-857, `shouldSend`).  In the engine's model the
//                        entity channel's subscribers are the subscribers of the cell that
//                        held the entity, i.e. of src.
//   adjacent broadcast   message.go:188-239: de-duplicated connections of a spatial channel
//                        and its up-to-8 neighbours (spatial.go:358-381), with the reference's
//                        flag filters.
//
// Both are "which connection slots have one of these cells in their interest set" sweeps:
// one workgroup per request walks the connection slots 256 at a time, tests the interest
// bitmap (64-bit words, one per 64 cells) and compacts with ballot + a 4-wave LDS prefix, so the
// lists come out in ascending slot order without atomics.  Two passes (count, scan, fill) keep
// the output dense.  Traffic: ~16 B of bitmap per (request, slot) from L2.
#include "chd_kernels.h"

static inline unsigned nblocks(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

#define RCP_NONE 0xFFu

// block-wide exclusive position of a flag in thread order; returns the block total through `total`
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t *wcnt, uint32_t &total) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t m = __ballot(flag);
    __syncthreads();
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    for (uint32_t k = 0; k < 4; k++) {
        if (k < wave) base += wcnt[k];
        total += wcnt[k];
    }
    return base + mask_rank(m);
}

__device__ __forceinline__ uint8_t handover_kind(const WorldDev &w, uint32_t s, uint32_t src, uint32_t dst) {
    if (s >= w.S || !w.sub_alive[s]) return RCP_NONE;
    const bool in_src = is_subscribed(w, s, src), in_dst = is_subscribed(w, s, dst);
    if (in_dst) return in_src ? CHD_HO_DST_KNOWN : CHD_HO_DST_NEW;
    return in_src ? CHD_HO_SRC_ONLY : RCP_NONE;
}

// Which entities of handover h go out WITH their entityData to destination connection s: the reference decides it per
// (connection, ENTITY) — `shouldSend` of SubscribeToChannel(entityCh, {DataAccess: WRITE for the entity channel's owner, else READ})
// in the loop of spatial.go:797-857, which is true when the connection was not yet subscribed to that entity's channel OR the merge
// changed its DataAccess (subscription.go:44-57).  In the engine's model an entity channel's subscribers are the subscribers of
// the cell whose entity map HELD the entity when Notify ran: src for the members that moved with the handover (ho_moved), the
// cell that still holds them for the others; and its owner is the spatial server of that cell — after step 1 of a CROSS-SERVER
// handover (spatial.go:683-700: entityCh.SetOwner(dstChannel.GetOwner()) for every handover entity) the dst cell's.  So the
// DataAccess of a subscribed connection changes exactly when it is the old or the new owner's connection and the two differ:
// the dst server (READ through its border interest -> WRITE) and the src server if it keeps interest in dst (WRITE -> READ).
// Server connections are known through chd_world_set_server_connections (none set: no connection is an owner, nothing changes).
// Bit q = entity q of the handover's list (the order of chd_handover_messages).
__device__ __forceinline__ bool handover_access_changed(const DevGrid &g, const WorldDev &w, uint32_t conn, uint32_t before, uint32_t src, uint32_t dst) {
    if (!w.server_conn || before == CHD_INVALID) return false;
    const uint32_t old_srv = server_of(g, before);
    const uint32_t new_srv = server_of(g, src) != server_of(g, dst) ? server_of(g, dst) : old_srv;
    if (old_srv == new_srv) return false;
    const bool was = old_srv < w.n_server_conn && w.server_conn[old_srv] == conn;
    const bool is = new_srv < w.n_server_conn && w.server_conn[new_srv] == conn;
    return was != is;
}

__device__ __forceinline__ uint32_t handover_full_mask(const DevGrid &g, const WorldDev &w, uint32_t s, uint32_t h, uint32_t notifier, uint32_t src,
                                                       uint32_t dst, bool in_src) {
    const uint32_t moved = w.ho_moved ? w.ho_moved[h] : 1u;
    const uint32_t gi = w.n_groups ? w.grp_of[notifier] : CHD_INVALID;
    const uint32_t conn = w.conn_id[s];
    if (gi == CHD_INVALID) return (!in_src || handover_access_changed(g, w, conn, src, src, dst)) ? 1u : 0u;
    uint32_t mask = 0, q = 0;
    for (uint32_t k = w.grp_off[gi]; k < w.grp_off[gi + 1] && q < 32u; k++) {
        const uint32_t m = w.grp_mem[k];
        if (!(w.eflags[m] & EF_ALIVE)) continue;
        bool known;
        uint32_t before;
        if ((moved >> q) & 1u) { known = in_src; before = src; }
        else {
            before = w.member[m];
            known = before != CHD_INVALID && is_subscribed(w, s, before);
        }
        if (!known || handover_access_changed(g, w, conn, before, src, dst)) mask |= 1u << q;
        q++;
    }
    return mask;
}

// fill == 0: off[h] = number of recipients of handover h;  fill != 0: write them at off[h]
// own_unsub (count pass, may be NULL): own_unsub[h] = 1 when step 1 of a cross-server handover unsubscribes the src spatial server's
// connection from the handover entities' channels (spatial.go:688-694: `ownerConn := srcChannel.GetOwner(); ... !ownerConn.
// HasInterestIn(dstChannelId)`): a live slot holds the src server's ConnectionId (chd_world_set_server_connections) and dst is not
// among its subscriptions.
__global__ void __launch_bounds__(256) k_handover_recipients(DevGrid g, WorldDev w, uint32_t *off, uint32_t *conn,
                                                             uint8_t *kind, uint32_t *full_mask, uint64_t cap, int fill, uint32_t *own_unsub) {
    __shared__ uint32_t wcnt[4];
    __shared__ uint32_t own_flag;
    const uint32_t n = min(w.counters[CTR_HANDOVERS], w.handovers_cap);
    for (uint32_t h = blockIdx.x; h < n; h += gridDim.x) {
        const chd_handover_rec r = w.handovers[h];
        const uint32_t src = r.src - g.id_start, dst = r.dst - g.id_start;
        uint32_t run = fill ? off[h] : 0u;
        const uint32_t ssrv = server_of(g, src);
        const bool own_on = !fill && own_unsub && w.server_conn && ssrv != server_of(g, dst) && ssrv < w.n_server_conn;
        const uint32_t owner = own_on ? w.server_conn[ssrv] : 0u;
        if (own_unsub && !fill) {
            if (threadIdx.x == 0) own_flag = 0;
            __syncthreads();
        }
        for (uint32_t s0 = 0; s0 < w.S; s0 += 256) {
            const uint32_t s = s0 + threadIdx.x;
            const uint8_t k = handover_kind(w, s, src, dst);
            if (own_on && s < w.S && w.sub_alive[s] && w.conn_id[s] == owner && k != CHD_HO_DST_NEW && k != CHD_HO_DST_KNOWN) own_flag = 1;
            uint32_t total;
            const uint32_t pos = run + block_rank(k != RCP_NONE, wcnt, total);
            if (fill && k != RCP_NONE && pos < cap) {
                conn[pos] = w.conn_id[s];
                kind[pos] = k;
                if (full_mask) full_mask[pos] = k == CHD_HO_SRC_ONLY ? 0u : handover_full_mask(g, w, s, h, r.entity, src, dst, k == CHD_HO_DST_KNOWN);
            }
            run += total;
        }
        if (!fill && threadIdx.x == 0) off[h] = run;
        if (!fill && own_unsub) {
            __syncthreads();
            if (threadIdx.x == 0) own_unsub[h] = own_flag;
            __syncthreads();
        }
    }
}

void launch_handover_recipients_count(hipStream_t st, DevGrid g, WorldDev w, uint32_t *off, uint32_t *own_unsub) {
    const unsigned grid = (unsigned)std::min<uint32_t>(w.handovers_cap ? w.handovers_cap : 1u, 2048u);
    hipLaunchKernelGGL(k_handover_recipients, dim3(grid), dim3(256), 0, st, g, w, off, nullptr, nullptr, nullptr, 0, 0, own_unsub);
}

void launch_handover_recipients_fill(hipStream_t st, DevGrid g, WorldDev w, const uint32_t *off, uint32_t *conn,
                                     uint8_t *kind, uint32_t *full_mask, uint64_t cap) {
    const unsigned grid = (unsigned)std::min<uint32_t>(w.handovers_cap ? w.handovers_cap : 1u, 2048u);
    hipLaunchKernelGGL(k_handover_recipients, dim3(grid), dim3(256), 0, st, g, w, (uint32_t *)off, conn, kind, full_mask, cap, 1, (uint32_t *)nullptr);
}

__global__ void __launch_bounds__(256) k_adjacent_recipients(DevGrid g, WorldDev w, uint32_t n_req,
                                                             const uint32_t *__restrict__ channel,
                                                             const uint32_t *__restrict__ broadcast,
                                                             const uint32_t *__restrict__ sender_conn,
                                                             const uint32_t *__restrict__ client_conn, uint32_t *off,
                                                             uint32_t *conns, uint64_t cap, int fill) {
    __shared__ uint32_t wcnt[4];
    const uint32_t r = blockIdx.x;
    if (r >= n_req) return;
    const uint32_t bc = broadcast[r];
    const uint32_t index = channel[r] - g.id_start;
    // GetAdjacentChannels (spatial.go:358-381) + the centre unless ALL_BUT_OWNER (message.go:201-204)
    uint32_t cells[9], nc = 0;
    if (index < g.ncell) {
        const int32_t gx = (int32_t)(index % g.cols), gy = (int32_t)(index / g.cols);
        for (int32_t y = gy - 1; y <= gy + 1; y++) {
            if (y < 0 || y > (int32_t)(g.rows - 1)) continue;
            for (int32_t x = gx - 1; x <= gx + 1; x++) {
                if (x < 0 || x > (int32_t)(g.cols - 1)) continue;
                if (x == gx && y == gy) continue;
                cells[nc++] = (uint32_t)x + (uint32_t)y * g.cols;
            }
        }
        if (!(bc & CHD_BROADCAST_ALL_BUT_OWNER)) cells[nc++] = index;
    }
    uint32_t run = fill ? off[r] : 0u;
    for (uint32_t s0 = 0; s0 < w.S; s0 += 256) {
        const uint32_t s = s0 + threadIdx.x;
        bool hit = false;
        uint32_t cid = 0;
        if (s < w.S && w.sub_alive[s]) {
            for (uint32_t k = 0; k < nc && !hit; k++) hit = is_subscribed(w, s, cells[k]);
            cid = w.conn_id[s];
            // every connection registered here is a CLIENT connection (message.go:227-233)
            if (bc & CHD_BROADCAST_ALL_BUT_CLIENT) hit = false;
            if ((bc & CHD_BROADCAST_ALL_BUT_SENDER) && cid == sender_conn[r]) hit = false;  // :223-225
            if (cid == client_conn[r]) hit = false;                                        // :235-237
        }
        uint32_t total;
        const uint32_t pos = run + block_rank(hit, wcnt, total);
        if (fill && hit && pos < cap) conns[pos] = cid;
        run += total;
    }
    if (!fill && threadIdx.x == 0) off[r] = run;
}

void launch_adjacent_recipients(hipStream_t st, DevGrid g, WorldDev w, uint32_t n_req, const uint32_t *channel,
                                const uint32_t *broadcast, const uint32_t *sender_conn, const uint32_t *client_conn,
                                uint32_t *off, uint32_t *conns, uint64_t cap, int fill) {
    if (!n_req) return;
    hipLaunchKernelGGL(k_adjacent_recipients, dim3(n_req), dim3(256), 0, st, g, w, n_req, channel, broadcast, sender_conn,
                       client_conn, off, conns, cap, fill);
}
