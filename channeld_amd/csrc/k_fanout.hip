// k_fanout.hip — K5: the per-channel data fan-out tick for every spatial and
// entity channel of the world at once.
//
// Replaces Channel.tickData (data.go:175-291) and the decision of
// fanOutDataUpdate (data.go:293-318).  State per (connection, spatial channel)
// subscription = the reference's fanOutConnection{hadFirstFanOut,
// lastFanOutTime} + ChannelSubscriptionOptions{FanOutIntervalMs, DataAccess,
// SkipSelfUpdateFanOut}; an entity channel's subscribers are the subscribers of
// the cell holding the entity and share that state.
//
// tickData's list walk (move-to-back + revisit) nets out, per subscription, to
//     while t >= last + interval:
//         first time : send the whole channel data,            last = t
//         otherwise  : send iff some buffered update u with
//                      max(last,0) <= u.arrival <= last+interval
//                      (and u.sender != conn when SkipSelfUpdateFanOut),
//                                                               last += interval
// (oracle/chd_oracle.c keeps the literal list walk and tests/ proves the
// equivalence).  Updates enter in per-tick batches stamped with the tick's
// channel time, so a channel's update buffer is a 32-bit history mask over the
// ring of the last 32 tick stamps, and a window is a mask over that ring.
//
//   k_fanout_plan : one wave per connection; upper bound of the records it can
//                   emit this tick (windows x entities of the cell) -> rec_ub
//   scan          : exclusive scan -> each connection's base in the record buffer
//   k_fanout_emit : one wave per connection; walks its subscriptions, streams the
//                   cells' SoA entity arrays (coalesced 4-byte loads), compacts
//                   with ballot/mbcnt and writes 8-byte {conn, channel} records
//                   contiguously (coalesced 512-byte wave stores).  Records of a
//                   connection are contiguous: [rec_ub[s], rec_ub[s]+rec_cnt[s]).
// HBM-bound: ~8 B written + 4-12 B read (L2/MALL resident cell tables) per
// record; no MFMA — this is gather/compaction, not a contraction.
#include "chd_kernels.h"

#define FO_WAVES 4

__device__ __forceinline__ uint32_t window_mask(const TickRing &ring, int64_t lo, int64_t hi) {
    uint32_t lane = lane_id();
    bool in = false;
    if (lane < ring.n) {
        int64_t a = ring.t[lane];
        in = (a >= lo) && (a <= hi);
    }
    return (uint32_t)__ballot(in);
}

// windows [.., hi] with hi < oldest stamp can never select an update: skip them
__device__ __forceinline__ int64_t skippable(const TickRing &ring, int64_t L, int64_t I, int64_t nwin) {
    if (ring.n == 0) return nwin;
    int64_t oldest = ring.t[ring.n - 1];
    if (oldest <= L) return 0;
    int64_t d = oldest - L;
    int64_t k = (d + I - 1) / I - 1;  // windows whose hi = L+(k+1)I stays < oldest
    if (k < 0) k = 0;
    return k < nwin ? k : nwin;
}

__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_plan(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    const uint32_t s = blockIdx.x * FO_WAVES + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = lane_id();
    uint64_t ub = 0;
    uint32_t npairs = 0;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        npairs = cnt;
        const size_t pbase = (size_t)s * w.capq;
        for (uint32_t p = lane; p < cnt; p += 64) {
            uint32_t fl = w.pair_flags[pbase + p];
            if (fl & PF_NO_ACCESS) continue;
            int64_t L = w.pair_last[pbase + p];
            int64_t I = (int64_t)w.pair_iv[pbase + p] * 1000000;
            if (I <= 0 || now < L + I) continue;
            uint32_t c = w.pair_cell[pbase + p];
            uint64_t size = (uint64_t)(w.blk_cnt[(size_t)(c + 1) * w.nblk] - w.blk_cnt[(size_t)c * w.nblk]) + 1;
            if (!(fl & PF_HAD_FIRST)) {
                ub += size;  // one full-state window, then last = now
            } else {
                int64_t nwin = (now - L) / I;
                int64_t lim = 2 * (int64_t)ring.n;  // a stamp lies in at most two windows
                if (nwin > lim) nwin = lim;
                ub += (uint64_t)nwin * size;
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) ub += __shfl_xor((unsigned long long)ub, d);
    if (lane == 0) {
        w.rec_ub[s] = ub;
        if (npairs) atomicAdd(&w.counters[CTR_PAIRS], npairs);
    }
}

void launch_fanout_plan(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_fanout_plan, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w,
                       now_ns, ring);
    launch_scan_u64_inplace(st, w.rec_ub, w.S);
}

__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_emit(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    const uint32_t s = blockIdx.x * FO_WAVES + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = lane_id();
    if (!w.sub_alive[s]) {
        if (lane == 0) w.rec_cnt[s] = 0;
        return;
    }
    const uint64_t base = w.rec_ub[s];
    if (w.rec_ub[s + 1] > w.recs_cap) {
        // no room for this connection's worst case: leave its state untouched, it
        // catches up next tick (the reference's catch-up loop), and say so.
        if (lane == 0) {
            w.rec_cnt[s] = 0;
            if (w.rec_ub[s + 1] > base) atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
        }
        return;
    }
    const uint32_t conn = w.conn_id[s];
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    chd_fanout_rec *__restrict__ out = w.recs + base;
    uint32_t n_out = 0;
    uint32_t hist_ovf = 0;
    for (uint32_t p = 0; p < cnt; p++) {
        uint32_t fl = w.pair_flags[pbase + p];
        if (fl & PF_NO_ACCESS) continue;  // data.go:194-197: skipped, stays queued
        int64_t L = w.pair_last[pbase + p];
        const int64_t I = (int64_t)w.pair_iv[pbase + p] * 1000000;
        if (I <= 0 || now < L + I) continue;
        const uint32_t c = w.pair_cell[pbase + p];
        const uint32_t start = w.blk_cnt[(size_t)c * w.nblk];
        const uint32_t end = w.blk_cnt[(size_t)(c + 1) * w.nblk];
        const bool skip_self = (fl & PF_SKIP_SELF) != 0;
        if (!(fl & PF_HAD_FIRST)) {
            // first fan-out: the whole data of the spatial channel and of every
            // entity channel in it (data.go:217-223); last = t
            if (lane == 0) {
                chd_fanout_rec r;
                r.conn = conn | CHD_REC_FULL;
                r.channel = c + g.id_start;
                out[n_out] = r;
            }
            n_out += 1;
            for (uint32_t b = start; b < end; b += 64) {
                uint32_t pos = b + lane;
                if (pos < end) {
                    chd_fanout_rec r;
                    r.conn = conn | CHD_REC_FULL;
                    r.channel = w.ce_chan[pos];
                    out[n_out + (pos - b)] = r;
                }
                n_out += min(64u, end - b);
            }
            fl |= PF_HAD_FIRST;
            L = now;
        }
        // catch-up windows (data.go:224-271 + the revisit through :273-286)
        if (now >= L + I) {
            int64_t nwin = (now - L) / I;
            int64_t skip = skippable(ring, L, I, nwin);
            if (skip > 0 && ring.n == CHD_HIST_BITS) hist_ovf = 1;
            L += skip * I;
            nwin -= skip;
            const uint32_t ch_hist = [&]() {
                uint32_t age = ring.cur_tick - w.cell_hist_tick[c];
                return age >= CHD_HIST_BITS ? 0u : (w.cell_hist[c] << age);
            }();
            const uint32_t ch_sender = w.cell_sender[c];
            for (int64_t k = 0; k < nwin; k++) {
                const int64_t next = L + I;
                const int64_t lo = L > 0 ? L : 0;  // lastUpdateTime starts at max(last, 0)
                const uint32_t wm = window_mask(ring, lo, next);
                if (wm) {
                    // the spatial channel's own buffered updates
                    if ((ch_hist & wm) && !(skip_self && ch_sender == conn)) {
                        if (lane == 0) {
                            chd_fanout_rec r;
                            r.conn = conn;
                            r.channel = c + g.id_start;
                            out[n_out] = r;
                        }
                        n_out += 1;
                    }
                    for (uint32_t b = start; b < end; b += 64) {
                        uint32_t pos = b + lane;
                        bool pass = false;
                        if (pos < end) {
                            pass = (w.ce_hist[pos] & wm) != 0;
                            if (pass && skip_self) pass = w.ce_sender[pos] != conn;
                        }
                        uint64_t m = __ballot(pass);
                        if (pass) {
                            chd_fanout_rec r;
                            r.conn = conn;
                            r.channel = w.ce_chan[pos];
                            out[n_out + mask_rank(m)] = r;
                        }
                        n_out += (uint32_t)__popcll(m);
                    }
                }
                L = next;
            }
        }
        if (lane == 0) {
            w.pair_last[pbase + p] = L;
            w.pair_flags[pbase + p] = fl;
        }
    }
    if (lane == 0) {
        w.rec_cnt[s] = n_out;
        if (hist_ovf) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    }
}

void launch_fanout_emit(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_fanout_emit, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w,
                       now_ns, ring);
}

// per-tick totals into the device-side history ring (read back by chd_get_tick_history)
__global__ void __launch_bounds__(1024) k_tick_epilogue(WorldDev w, uint32_t slot) {
    __shared__ unsigned long long part[16];
    unsigned long long sum = 0;
    for (uint32_t s = threadIdx.x; s < w.S; s += 1024) sum += w.rec_cnt[s];
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int k = 0; k < 16; k++) tot += part[k];
        uint64_t *r = w.tick_ring + (size_t)slot * 8;
        r[0] = tot;
        r[1] = w.rec_ub[w.S];
        r[2] = w.counters[CTR_HANDOVERS];
        r[3] = w.counters[CTR_LOCKED];
        r[4] = w.counters[CTR_UNSUBS];
        r[5] = w.counters[CTR_NEWSUBS];
        r[6] = w.counters[CTR_PAIRS];
        r[7] = (uint64_t)w.counters[CTR_OVERFLOW] | ((uint64_t)w.counters[CTR_HIST_OVERFLOW] << 32);
    }
}

void launch_tick_epilogue(hipStream_t st, WorldDev w, uint32_t slot) {
    hipLaunchKernelGGL(k_tick_epilogue, dim3(1), dim3(1024), 0, st, w, slot);
}
