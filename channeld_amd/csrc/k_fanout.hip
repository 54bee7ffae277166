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
//   k_fanout_plan : one wave per connection, one lane per subscription: upper
//                   bound of the records each due subscription can emit this tick
//                   (windows x (entities of the cell + 1)), wave prefix sum ->
//                   pair_rel (segment offset inside the connection's range),
//                   total -> rec_ub
//   scan          : exclusive scan -> each connection's base in the record buffer
//   k_fanout_emit : one 4-wave workgroup per connection; the waves pull the
//                   connection's subscriptions from an LDS ticket.  Per due
//                   subscription a wave streams the cell's 16-byte entity entries
//                   {channel, history, sender, slot} (four 16-B loads in flight
//                   per lane, L2-resident), compacts with ballot/mbcnt and writes
//                   8-byte {conn, channel} records contiguously (512-B wave
//                   stores) into the subscription's segment
//                   [rec_ub[s] + pair_rel[s][p], + pair_nrec[s][p]).
// HBM-bound: 8 B written per record (the stream) + the L2/MALL-resident cell
// tables read; no MFMA — this is gather/compaction, not a contraction.
#include "chd_kernels.h"

#define FO_WAVES 4
#define FO_UNROLL 4

// stamp of ring slot `lane` (INT64_MAX for unused slots): loaded once per wave
__device__ __forceinline__ int64_t ring_stamp(const TickRing &ring) {
    uint32_t lane = lane_id();
    return lane < ring.n ? ring.t[lane < CHD_HIST_BITS ? lane : 0] : INT64_MAX;
}

__device__ __forceinline__ uint32_t window_mask(int64_t my_t, int64_t lo, int64_t hi) {
    return (uint32_t)__ballot(my_t >= lo && my_t <= hi);
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
    uint64_t carry = 0;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        for (uint32_t p0 = 0; p0 < cnt; p0 += 64) {
            const uint32_t p = p0 + lane;
            uint64_t ub = 0;
            if (p < cnt) {
                uint32_t fl = w.pair_flags[pbase + p];
                int64_t L = w.pair_last[pbase + p];
                int64_t I = (int64_t)w.pair_iv[pbase + p] * 1000000;
                if (!(fl & PF_NO_ACCESS) && I > 0 && now >= L + I) {
                    uint32_t c = w.pair_cell[pbase + p];
                    uint64_t size = (uint64_t)(w.cell_end[c] - w.cell_start[c]) + 1;
                    if (!(fl & PF_HAD_FIRST)) {
                        ub = size;  // one full-state window, then last = now
                    } else {
                        int64_t nwin = (now - L) / I;
                        int64_t lim = 2 * (int64_t)ring.n;  // a stamp lies in at most two windows
                        if (nwin > lim) nwin = lim;
                        ub = (uint64_t)nwin * size;
                    }
                }
            }
            // exclusive prefix over the subscriptions, in list order
            uint64_t inc = ub;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint64_t o = __shfl_up((unsigned long long)inc, d);
                if ((int)lane >= d) inc += o;
            }
            uint64_t rel = carry + inc - ub;
            if (p < cnt) w.pair_rel[pbase + p] = rel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rel;
            carry += __shfl((unsigned long long)inc, 63);
        }
    }
    // a connection whose worst case does not fit 32-bit segment offsets cannot be
    // served this tick: make its range exceed every capacity (flagged by emit)
    if (lane == 0) w.rec_ub[s] = carry > 0xFFFFFFFFull ? (1ull << 40) : carry;
}

void launch_fanout_plan(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_fanout_plan, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w,
                       now_ns, ring);
    launch_scan_u64_inplace(st, w.rec_ub, w.S);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One window (or the full state) of one cell for one connection: stream the cell's
// entries, 4 x 64 per step.  The four 16-byte loads are issued back to back from
// one asm block (left to itself the compiler splits them into dwords, sinks pieces
// behind the ballot branches and interleaves waits: 2-3 serialised L2 round trips
// per step instead of one).
template <bool FULL>
__device__ __forceinline__ uint32_t emit_cell(const uint4 *__restrict__ ce, uint32_t start, uint32_t end,
                                              uint32_t wm, bool skip_self, uint32_t conn, uint32_t conn_tag,
                                              chd_fanout_rec *__restrict__ out, uint32_t n_out) {
    static_assert(FO_UNROLL == 4, "the load block below names four entries");
    const uint32_t lane = lane_id();
    for (uint32_t b = start; b < end; b += 64 * FO_UNROLL) {
        u32x4 e[FO_UNROLL];
        const uint4 *p[FO_UNROLL];
#pragma unroll
        for (int j = 0; j < FO_UNROLL; j++) {
            uint32_t pos = b + j * 64 + lane;
            p[j] = ce + (pos < end ? pos : end - 1);
        }
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\t"
            "global_load_dwordx4 %1, %5, off\n\t"
            "global_load_dwordx4 %2, %6, off\n\t"
            "global_load_dwordx4 %3, %7, off\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
            : "memory");
#pragma unroll
        for (int j = 0; j < FO_UNROLL; j++) {
            uint32_t pos = b + j * 64 + lane;
            bool pass = pos < end;
            if (!FULL) {
                pass = pass && (e[j].y & wm) != 0;
                if (skip_self) pass = pass && e[j].z != conn;
            }
            uint64_t m = __ballot(pass);
            if (pass) {
                chd_fanout_rec r;
                r.conn = conn_tag;
                r.channel = e[j].x;
                out[n_out + mask_rank(m)] = r;
            }
            n_out += (uint32_t)__popcll(m);
        }
    }
    return n_out;
}

#define FO_TILE 256  // subscriptions staged in LDS per round (= workgroup size)

__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_emit(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    // due subscriptions of this connection, staged once per workgroup so that the
    // streaming waves never wait on per-subscription pointer chasing
    __shared__ uint32_t d_p[FO_TILE], d_fl[FO_TILE], d_c[FO_TILE], d_start[FO_TILE], d_end[FO_TILE], d_rel[FO_TILE],
        d_chh[FO_TILE], d_chs[FO_TILE], d_iv[FO_TILE];
    __shared__ int64_t d_L[FO_TILE];
    __shared__ uint32_t n_due, ticket;
    __shared__ uint32_t wave_total[FO_WAVES];
    const uint32_t s = blockIdx.x;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!w.sub_alive[s]) {
        if (threadIdx.x == 0) w.rec_cnt[s] = 0;
        return;
    }
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const uint64_t base = w.rec_ub[s];
    if (w.rec_ub[s + 1] > w.recs_cap) {
        // no room for this connection's worst case: leave its state untouched, it
        // catches up next tick (the reference's catch-up loop), and say so.
        for (uint32_t p = threadIdx.x; p < cnt; p += 64 * FO_WAVES) w.pair_nrec[pbase + p] = 0;
        if (threadIdx.x == 0) {
            w.rec_cnt[s] = 0;
            if (w.rec_ub[s + 1] > base) atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
        }
        return;
    }
    const uint32_t conn = w.conn_id[s];
    const int64_t my_t = ring_stamp(ring);
    const uint4 *__restrict__ ce = w.ce_view;
    uint32_t total = 0;
    uint32_t hist_ovf = 0;
    for (uint32_t tile = 0; tile < cnt; tile += FO_TILE) {
        if (threadIdx.x == 0) { n_due = 0; ticket = 0; }
        __syncthreads();
        {   // stage: one thread per subscription
            const uint32_t p = tile + threadIdx.x;
            if (p < cnt) {
                const uint32_t fl = w.pair_flags[pbase + p];
                const int64_t L = w.pair_last[pbase + p];
                const uint32_t iv = w.pair_iv[pbase + p];
                const int64_t I = (int64_t)iv * 1000000;
                // data.go:194-197: NO_ACCESS is skipped but stays queued
                if (!(fl & PF_NO_ACCESS) && I > 0 && now >= L + I) {
                    const uint32_t c = w.pair_cell[pbase + p];
                    const uint32_t k = atomicAdd(&n_due, 1u);
                    d_p[k] = p; d_fl[k] = fl; d_L[k] = L; d_iv[k] = iv; d_c[k] = c;
                    d_rel[k] = w.pair_rel[pbase + p];
                    d_start[k] = w.cell_start[c];
                    d_end[k] = w.cell_end[c];
                    const uint32_t age = ring.cur_tick - w.cell_hist_tick[c];
                    d_chh[k] = age < CHD_HIST_BITS ? (w.cell_hist[c] << age) : 0u;
                    d_chs[k] = w.cell_sender[c];
                } else {
                    w.pair_nrec[pbase + p] = 0;
                }
            }
        }
        __syncthreads();
        const uint32_t ndue = n_due;
        for (;;) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&ticket, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= ndue) break;
            const uint32_t p = d_p[k];
            uint32_t fl = d_fl[k];
            int64_t L = d_L[k];
            const int64_t I = (int64_t)d_iv[k] * 1000000;
            const uint32_t c = d_c[k];
            const uint32_t start = d_start[k], end = d_end[k];
            const bool skip_self = (fl & PF_SKIP_SELF) != 0;
            chd_fanout_rec *__restrict__ out = w.recs + base + d_rel[k];
            uint32_t n_out = 0;
            if (!(fl & PF_HAD_FIRST)) {
                // first fan-out: the whole data of the spatial channel and of every
                // entity channel in it (data.go:217-223); last = t
                if (lane == 0) {
                    chd_fanout_rec r;
                    r.conn = conn | CHD_REC_FULL;
                    r.channel = c + g.id_start;
                    out[0] = r;
                }
                n_out = emit_cell<true>(ce, start, end, 0u, false, conn, conn | CHD_REC_FULL, out, 1u);
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
                const uint32_t ch_hist = d_chh[k];
                const uint32_t ch_sender = d_chs[k];
                for (int64_t j = 0; j < nwin; j++) {
                    const int64_t next = L + I;
                    const int64_t lo = L > 0 ? L : 0;  // lastUpdateTime starts at max(last, 0)
                    const uint32_t wm = window_mask(my_t, lo, next);
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
                        n_out = emit_cell<false>(ce, start, end, wm, skip_self, conn, conn, out, n_out);
                    }
                    L = next;
                }
            }
            if (lane == 0) {
                w.pair_last[pbase + p] = L;
                w.pair_flags[pbase + p] = fl;
                w.pair_nrec[pbase + p] = n_out;
            }
            total += n_out;
        }
        __syncthreads();
    }
    if (lane == 0) wave_total[wave] = total;
    if (hist_ovf && lane == 0) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < FO_WAVES; k++) t += wave_total[k];
        w.rec_cnt[s] = t;
        // per-tick totals go through 64 hashed counters, one 128-byte line each: a
        // single word (or words sharing a line) would serialise S atomics at ~12 ns
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16];
        if (t) atomicAdd(slot, (unsigned long long)t);
        if (cnt) atomicAdd(slot + 1, (unsigned long long)cnt);
    }
}

void launch_fanout_emit(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    hipLaunchKernelGGL(k_fanout_emit, dim3(w.S), dim3(64 * FO_WAVES), 0, st, g, w, now_ns, ring);
}

// Per-tick totals into the device-side history ring (read back by chd_tick_fetch /
// chd_get_tick_history), then the per-tick counters are cleared for the next tick.
__global__ void __launch_bounds__(64) k_tick_epilogue(WorldDev w, uint32_t slot) {
    const uint32_t lane = threadIdx.x;
    unsigned long long sum = w.tot64[(size_t)lane * 16], pairs = w.tot64[(size_t)lane * 16 + 1];
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        pairs += __shfl_xor(pairs, d);
    }
    if (lane == 0) {
        uint64_t *r = w.tick_ring + (size_t)slot * 8;
        r[0] = sum;
        r[1] = w.rec_ub[w.S];
        r[2] = w.counters[CTR_HANDOVERS];
        r[3] = w.counters[CTR_LOCKED];
        r[4] = w.counters[CTR_UNSUBS];
        r[5] = w.counters[CTR_NEWSUBS];
        r[6] = pairs;
        r[7] = (uint64_t)w.counters[CTR_OVERFLOW] | ((uint64_t)w.counters[CTR_HIST_OVERFLOW] << 32);
    }
    __syncthreads();
    if (lane < CTR_COUNT) w.counters[lane] = 0;
    w.tot64[(size_t)lane * 16] = 0;
    w.tot64[(size_t)lane * 16 + 1] = 0;
}

void launch_tick_epilogue(hipStream_t st, WorldDev w, uint32_t slot) {
    hipLaunchKernelGGL(k_tick_epilogue, dim3(1), dim3(64), 0, st, w, slot);
}
