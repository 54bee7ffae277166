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
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "chd_kernels.h"

// -DCHD_PROFILE_CONN_EMIT: per-phase cycle counts of a few sampled workgroups of the connection-major emit kernel,
// printed at tick 60 (where do the waves wait?  PC sampling / thread trace are not available on this pool)
#ifdef CHD_PROFILE_CONN_EMIT
#define CE_MARK(acc) do { long long _t = clock64(); (acc) += _t - ce_mark; ce_mark = _t; } while (0)
#define CE_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define CE_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define CE_MARK(acc) do { } while (0)
#define CE_WAIT_VM() do { } while (0)
#define CE_WAIT_LGKM() do { } while (0)
#endif

#ifndef FO_WAVES
#define FO_WAVES 4
#endif
#define CHD_SEG_ALIGN 16u  // records: 16 x 8 B = one 128-byte line
#ifndef FO_UNROLL
#define FO_UNROLL 4
#endif
#ifndef FO_UNROLL8
#define FO_UNROLL8 2
#endif
#ifndef FO_COPY_AHEAD
#define FO_COPY_AHEAD 0
#endif

// stamp of ring slot `lane` (INT64_MAX for unused slots): loaded once per wave
__device__ __forceinline__ int64_t ring_stamp(const TickRing &ring) {
    uint32_t lane = lane_id();
    return lane < ring.n ? ring.t[lane < CHD_HIST_BITS ? lane : 0] : INT64_MAX;
}

__device__ __forceinline__ uint32_t window_mask(int64_t my_t, int64_t lo, int64_t hi) {
    return (uint32_t)__ballot(my_t >= lo && my_t <= hi);
}

// floor(d / (iv ms)) for d >= 0 ns, without a 64-bit division (hundreds of instructions on the GPU, and enough live
// scalars to spill the emit kernel's): ns -> ms is a division by a constant (multiply-high), the rest is 32-bit.
// Clamped to 2^32 - 1 ms (49 days): an UNDERESTIMATE there, which every caller tolerates (it jumps fewer windows
// and comes back).
__device__ __forceinline__ int64_t div_interval(int64_t d, uint32_t iv) {
    const uint64_t ms = (uint64_t)d / 1000000u;
    const uint32_t m32 = ms > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ms;
    return (int64_t)(m32 / iv);
}

// floor(d / (iv ms)) EXACTLY, for d >= 0 ns: floor(floor(d / 10^6) / iv) is the same number, the inner division is by a constant
// (multiply-high) and the outer one 32-bit unless d is beyond 2^32 ms (49 days: a rare per-lane branch keeps the 64-bit form).  A 64-bit
// signed division is ~100 vector instructions on this hardware, and the plan kernels did three of them per subscription.
__device__ __forceinline__ int64_t div_windows(int64_t d, uint32_t iv) {
    const uint64_t ms = (uint64_t)d / 1000000u;
    if (ms <= 0xFFFFFFFFull) return (int64_t)((uint32_t)ms / iv);
    return (int64_t)(ms / iv);
}
// ... given the interval in nanoseconds (I = iv x 10^6 > 0)
__device__ __forceinline__ int64_t div_windows_ns(int64_t d, int64_t I) { return div_windows(d, (uint32_t)((uint64_t)I / 1000000u)); }

// The window [max(L,0), L + I] holds no buffered stamp: how many windows to move `last` on in one go.  Empty
// windows emit nothing, so this is the walk of tickData with its empty iterations folded: straight to the window
// that holds the oldest stamp newer than L + I, or past every due window when there is none.  (A connection
// served after a long pause would otherwise walk every empty interval: 10^6 iterations for 1 ms intervals after
// 1000 s.)  `ts_newer` = that stamp, valid when n_newer != 0.
__device__ __forceinline__ int64_t empty_windows_to(int64_t ts_newer, uint32_t n_newer, int64_t now, int64_t L, uint32_t iv) {
    const int64_t I = (int64_t)iv * 1000000;
    const int64_t left = div_interval(now - L, iv);    // windows still due (>= 1; an underestimate only when clamped)
    int64_t j = left;
    if (n_newer) {
        const int64_t k = div_interval(ts_newer - L, iv);  // window k = [L + kI, L + (k+1)I] holds ts ...
        j = (ts_newer - L) - k * I == 0 ? k - 1 : k;       // ... and so does window k-1 when ts is its upper edge
        if (j > left) j = left;
    }
    return j < 1 ? 1 : j;
}

// wave form: lane j holds the stamp of ring slot j (ring_stamp); stamps do not increase with the slot index, so
// the stamps newer than the window are slots [0, newer)
__device__ __forceinline__ int64_t empty_windows(const TickRing &ring, int64_t my_t, int64_t now, int64_t L, uint32_t iv) {
    const uint32_t newer = (uint32_t)__popcll(__ballot(lane_id() < ring.n && my_t > L + (int64_t)iv * 1000000));
    const uint32_t src = newer ? newer - 1 : 0;
    const int64_t ts = ((int64_t)__shfl((int)(my_t >> 32), (int)src) << 32) | (uint32_t)__shfl((int)my_t, (int)src);
    return empty_windows_to(ts, newer, now, L, iv);
}

// Updates older than the 32-tick history can no longer be selected: a subscription whose next window ends before
// the oldest stamp of a FULL ring may have lost some (reported as history_overflow; the windows themselves are
// empty and folded by empty_windows).
__device__ __forceinline__ bool history_lost(const TickRing &ring, int64_t oldest, int64_t L, int64_t I) {
    return ring.n == CHD_HIST_BITS && oldest > L + I;
}

// Exact update buffers (history_depth > 0): a due subscription whose windows the tick-ring masks cannot answer — some
// channel of its cell took an update the masks do not represent (an arrival stamp off the tick's own, a third sender), or
// its next window ends before the oldest stamp of the full ring — is served from the buffers themselves by
// k_fanout_emit_deep.  (A first fan-out sends full states and starts over at `now`: no window to evaluate.)
__device__ __forceinline__ bool sub_is_deep(const WorldDev &w, const TickRing &ring, int64_t oldest, uint32_t fl, int64_t L, int64_t I,
                                            uint32_t c) {
    // (per-record masks on exact worlds: bit 31 of a record's word says "range form" — chd_tick_out.record_masks — so a window
    // that reaches the ring's 32nd tick is answered from the buffers as well)
    return w.deep_depth != 0 && (fl & PF_HAD_FIRST) &&
           (w.cell_irr[c] != 0 || history_lost(ring, oldest, L, I) || (w.rec_mask && ring.n == CHD_HIST_BITS && (L > 0 ? L : 0) <= oldest));
}
// (the same with cell_irr[c] already loaded)
__device__ __forceinline__ bool sub_is_deep_pre(const WorldDev &w, const TickRing &ring, int64_t oldest, uint32_t fl, int64_t L, int64_t I,
                                                uint32_t cell_irr_c) {
    return w.deep_depth != 0 && (fl & PF_HAD_FIRST) &&
           (cell_irr_c != 0 || history_lost(ring, oldest, L, I) || (w.rec_mask && ring.n == CHD_HIST_BITS && (L > 0 ? L : 0) <= oldest));
}
// worst case of its segment: per channel of the cell one record per due window, and no more than TWICE what the buffer holds
// elements — an arrival stamp that sits exactly on a window edge lies in two windows (both ends are inclusive, data.go:236-241),
// so deep_walk can write two records per element
__device__ __forceinline__ uint64_t deep_upper_bound(const WorldDev &w, int64_t now, int64_t L, int64_t I, uint64_t size) {
    int64_t nwin = div_windows_ns(now - L, I);
    if (nwin > 2 * (int64_t)w.deep_depth) nwin = 2 * (int64_t)w.deep_depth;
    return (uint64_t)nwin * (size + 1);
}

__device__ __forceinline__ uint32_t window_mask_serial(const TickRing &ring, int64_t lo, int64_t hi);
__device__ __forceinline__ int64_t empty_windows_serial(const TickRing &ring, int64_t now, int64_t L, uint32_t iv);

__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_plan(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    const uint32_t s = blockIdx.x * FO_WAVES + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = lane_id();
    uint64_t carry = 0;
    uint32_t any_deep = 0;
    const int64_t oldest = ring.n ? ring.t[ring.n - 1] : INT64_MAX;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        for (uint32_t p0 = 0; p0 < cnt; p0 += 64) {
            const uint32_t p = p0 + lane;
            uint64_t ub = 0;
            bool deep = false;
            if (p < cnt) {
                uint32_t fl = w.pair_flags[pbase + p];
                int64_t L = w.pair_last[pbase + p];
                int64_t I = (int64_t)w.pair_iv[pbase + p] * 1000000;
                if (!(fl & PF_NO_ACCESS) && I > 0 && now >= L + I) {
                    uint32_t c = w.pair_cell[pbase + p];
                    // region-sharded world: the cell's table must be on this rank (own region or a received halo band)
                    if (w.cell_cov && !w.cell_cov[c]) atomicOr(&w.counters[CTR_OVERFLOW], OVF_HALO);
                    uint64_t size = (uint64_t)(w.cell_end[c] - w.cell_start[c]) + 1;
                    if (sub_is_deep(w, ring, oldest, fl, L, I, c)) {
                        deep = true;
                        ub = deep_upper_bound(w, now, L, I, size - 1);
                        w.pair_flags[pbase + p] = fl | PF_DEEP;  // (the emit kernels skip it; k_fanout_emit_deep clears the bit)
                    } else if (!(fl & PF_HAD_FIRST)) {
                        ub = size;  // one full-state window, then last = now
                    } else {
                        int64_t nwin = div_windows_ns(now - L, I);
                        int64_t lim = 2 * (int64_t)ring.n;  // a stamp lies in at most two windows
                        if (nwin > lim) nwin = lim;
                        ub = (uint64_t)nwin * size;
                    }
                }
            }
            // every segment starts on a 128-byte line and is padded to whole lines: a line shared by
            // two segments would be written in two partial pieces by different workgroups, and partial
            // lines cost a read-modify-write at the (ECC) HBM — measured 3.7 vs 5.4 TB/s of stores
            ub = (ub + (CHD_SEG_ALIGN - 1)) & ~(uint64_t)(CHD_SEG_ALIGN - 1);
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
            if (__ballot(deep)) any_deep = 1;
        }
    }
    // a connection whose worst case does not fit 32-bit segment offsets cannot be
    // served this tick: make its range exceed every capacity (flagged by emit)
    if (lane == 0) {
        w.rec_ub[s] = carry > 0xFFFFFFFFull ? (1ull << 40) : carry;
        if (w.deep_depth) w.conn_deep[s] = any_deep;
    }
}

// cells with at least one live subscription, ascending (cell-major emit walks this list)
__global__ void __launch_bounds__(1024) k_active_cells(WorldDev w, uint32_t ncell) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ncell; base += 1024) {
        const uint32_t c = base + threadIdx.x;
        const bool act = c < ncell && w.cell_ref[c] != 0;
        const uint64_t m = __ballot(act);
        if (lane == 0) wtot[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t k = 0; k < wave; k++) off += wtot[k];
        if (act) w.active_cells[off + mask_rank(m)] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = carry_s;
            for (int k = 0; k < 16; k++) t += wtot[k];
            carry_s = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *w.n_active = carry_s;
}

// CHD_EMIT_PIPELINED=0 keeps the first connection-major form everywhere (A/B runs)
static bool seg_path_enabled() {
    static const bool on = [] { const char *e = getenv("CHD_EMIT_PIPELINED"); return !(e && e[0] == '0'); }();
    return on;
}
// the descriptor-driven path: connection-major, one-wave geometry (>= 4096 connections or asked for); with per-record masks
// (CHD_WORLD_UPDATE_MASKS) a window is a plain copy only where its mask is the same for every entity (k_fanout_plan_seg)
static bool seg_path(const WorldDev &w) { return !w.cm_emit && !w.seg_off && (w.S >= 4096 || w.one_wave_emit) && seg_path_enabled(); }

template <bool OFF>
__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_plan_seg(DevGrid g, WorldDev w, int64_t now, TickRing ring);
__global__ void __launch_bounds__(1024) k_fanout_scan(WorldDev w, uint32_t ncell, int seg);

void launch_fanout_plan(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    if (seg_path(w)) {
        if (w.off_on) hipLaunchKernelGGL(k_fanout_plan_seg<true>, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w, now_ns, ring);
        else hipLaunchKernelGGL(k_fanout_plan_seg<false>, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w, now_ns, ring);
    } else
        hipLaunchKernelGGL(k_fanout_plan, dim3((w.S + FO_WAVES - 1) / FO_WAVES), dim3(64 * FO_WAVES), 0, st, g, w,
                           now_ns, ring);
    hipLaunchKernelGGL(k_fanout_scan, dim3(1), dim3(1024), 0, st, w, g.ncell, seg_path(w) ? 1 : 0);
    if (w.cm_emit) hipLaunchKernelGGL(k_active_cells, dim3(1), dim3(1024), 0, st, w, g.ncell);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// fill the rest of the segment's last 128-byte line (never read: beyond the segment's record
// count) so that the line leaves L2 as one full-line write
__device__ __forceinline__ void pad_segment(chd_fanout_rec *__restrict__ out, uint32_t n_out) {
    const uint32_t pad = (0u - n_out) & (CHD_SEG_ALIGN - 1);
    if (lane_id() < pad) {
        chd_fanout_rec r;
        r.conn = 0xFFFFFFFFu;
        r.channel = 0;
        out[n_out + lane_id()] = r;
    }
}

// ... with the pad record materialised where it is stored: under k_fanout_emit_filt_cm's register cap the compiler hoisted the
// constant pair to the kernel's start and SPILLED it — a scratch reload per descriptor, i.e. a wait for all of the wave's own record
// stores (gfx950's in-order vm counter)
__device__ __forceinline__ void pad_segment_here(chd_fanout_rec *__restrict__ out, uint32_t n_out) {
    const uint32_t pad = (0u - n_out) & (CHD_SEG_ALIGN - 1);
    if (lane_id() < pad) {
        uint32_t m1, z;
        asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, 0" : "=v"(m1), "=v"(z));
        chd_fanout_rec r;
        r.conn = m1;
        r.channel = z;
        out[n_out + lane_id()] = r;
    }
}

// the spatial channel's own buffered updates against one window (two-sender history)
__device__ __forceinline__ bool cell_update_passes(uint32_t h, uint32_t snd, uint32_t hp, uint32_t sndp, uint32_t wm,
                                                   bool skip_self, uint32_t conn) {
    const uint32_t a = h & wm, b = hp & wm;
    if (!skip_self) return (a | b) != 0;
    return (a != 0 && snd != conn) || (b != 0 && sndp != conn);
}

// ... and WHICH of them the message merges (CHD_WORLD_UPDATE_MASKS): non-zero exactly when cell_update_passes
__device__ __forceinline__ uint32_t cell_update_mask(uint32_t h, uint32_t snd, uint32_t hp, uint32_t sndp, uint32_t wm,
                                                     bool skip_self, uint32_t conn) {
    const uint32_t a = h & wm, b = hp & wm;
    if (!skip_self) return a | b;
    return (snd != conn ? a : 0u) | (sndp != conn ? b : 0u);
}

// the buffered updates of one 16-byte entry that a window's message merges (data.go:242-256 per element:
// arrival inside the window and, with SkipSelfUpdateFanOut, sender != connection); non-zero exactly when
// entry_passes<false>
__device__ __forceinline__ uint32_t entry_mask(const WorldDev &w, const u32x4 &e, uint32_t pos, bool in_range, uint32_t wm,
                                               bool skip_self, uint32_t conn) {
    const uint32_t a = e.y & wm, b2 = e.w & wm;
    uint32_t m = a | b2;
    if (skip_self) {
        m = e.z != conn ? a : 0u;
        if (__ballot(in_range && b2 != 0)) {
            if (in_range && b2 != 0 && load_sprev(w, pos) != conn) m |= b2;
        }
    }
    return in_range ? m : 0u;
}

// One window (or the full state) of one cell for one connection: stream the cell's
// entries, 4 x 64 per step.  The four 16-byte loads are issued back to back from
// one asm block (left to itself the compiler splits them into dwords, sinks pieces
// behind the ballot branches and interleaves waits: 2-3 serialised L2 round trips
// per step instead of one).
// the per-entry test of one window (two-sender history); `pos` only for the rare side-table load
template <bool FULL>
__device__ __forceinline__ bool entry_passes(const WorldDev &w, const u32x4 &e, uint32_t pos, bool in_range,
                                             uint32_t wm, bool skip_self, uint32_t conn) {
    if (FULL) return in_range;
    const uint32_t a = e.y & wm, b2 = e.w & wm;  // current / previous sender's updates in the window
    if (!skip_self) return in_range && (a | b2) != 0;
    bool pa = a != 0 && e.z != conn;
    if (__ballot(in_range && b2 != 0)) {  // rare: a previous sender's update is still buffered
        if (in_range && b2 != 0) pa = pa || load_sprev(w, pos) != conn;
    }
    return in_range && pa;
}

// One window (or the full state) of one cell for one connection.  A step covers 512 entries:
// every lane owns FOUR PAIRS of adjacent entries (eight 16-byte loads issued back to back from one
// asm block — left to itself the compiler splits them into dwords, sinks pieces behind the ballot
// branches and interleaves waits).  The kernel is instruction-issue bound (SQ counters: the SIMDs
// are ~90 % busy), so the common case is made cheap: when all 128 entries of a pair-row pass, each
// lane writes its two records with ONE 16-byte store at out[n_out + 2*lane] and no rank arithmetic;
// otherwise ballot/mbcnt compaction keeps entry order.
template <bool FULL, bool MASKS = false>
__device__ __forceinline__ uint32_t emit_cell(const WorldDev &w, const uint4 *__restrict__ ce, uint32_t start, uint32_t end,
                                              uint32_t wm, bool skip_self, uint32_t conn, uint32_t conn_tag,
                                              chd_fanout_rec *__restrict__ out, uint32_t *__restrict__ opos, uint32_t n_out,
                                              uint32_t *__restrict__ omask = nullptr) {
    static_assert(FO_UNROLL == 8 || FO_UNROLL == 4, "the load blocks below name eight / four entries");
    const uint32_t lane = lane_id();
    for (uint32_t b = start; b < end; b += 64 * FO_UNROLL) {
        u32x4 e[FO_UNROLL];
        const uint4 *p[FO_UNROLL / 2];
#pragma unroll
        for (int j = 0; j < FO_UNROLL / 2; j++) {
            // row j of the step = entries [b + 128 j, b + 128 j + 128); this lane owns 2*lane and 2*lane + 1.
            // `end - 2` keeps both loads inside the cell (cells of fewer than 2 entries: see below).
            const uint32_t pos = b + j * 128 + 2 * lane;
            p[j] = ce + (pos + 1 < end ? pos : (end >= start + 2 ? end - 2 : start));  // ce has one spare entry
        }
        // On gfx950 the vm counter is in-order: waiting for these loads also drains the record stores
        // of the previous step, so a step should be as large as registers allow.
#if FO_UNROLL == 8
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\t"
            "global_load_dwordx4 %1, %8, off offset:16\n\t"
            "global_load_dwordx4 %2, %9, off\n\t"
            "global_load_dwordx4 %3, %9, off offset:16\n\t"
            "global_load_dwordx4 %4, %10, off\n\t"
            "global_load_dwordx4 %5, %10, off offset:16\n\t"
            "global_load_dwordx4 %6, %11, off\n\t"
            "global_load_dwordx4 %7, %11, off offset:16\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6]), "=&v"(e[7])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
            : "memory");
#else
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\t"
            "global_load_dwordx4 %1, %4, off offset:16\n\t"
            "global_load_dwordx4 %2, %5, off\n\t"
            "global_load_dwordx4 %3, %5, off offset:16\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
            : "v"(p[0]), "v"(p[1])
            : "memory");
#endif
#pragma unroll
        for (int j = 0; j < FO_UNROLL / 2; j++) {
            if (end - b <= (uint32_t)(j * 128)) break;  // uniform
            const uint32_t pos = b + j * 128 + 2 * lane;
            const bool in0 = pos < end, in1 = pos + 1 < end;
            // a lane whose pair was clamped loaded entries q, q+1 with q = end-2 (or start): entry `pos`
            // is then the pair's second entry unless pos == q
            const uint32_t q = in1 ? pos : (end >= start + 2 ? end - 2 : start);
            const u32x4 ea = pos == q ? e[2 * j] : e[2 * j + 1];
            bool pass0, pass1;
            uint32_t mk0 = 0, mk1 = 0;  // MASKS: the buffered updates each message merges
            if (MASKS && !FULL) {
                mk0 = entry_mask(w, ea, pos, in0, wm, skip_self, conn);
                mk1 = entry_mask(w, e[2 * j + 1], pos + 1, in1, wm, skip_self, conn);
                pass0 = mk0 != 0;
                pass1 = mk1 != 0;
            } else {
                pass0 = entry_passes<FULL>(w, ea, pos, in0, wm, skip_self, conn);
                pass1 = entry_passes<FULL>(w, e[2 * j + 1], pos + 1, in1, wm, skip_self, conn);
            }
            const uint64_t m0 = __ballot(pass0), m1 = __ballot(pass1);
            if ((m0 & m1) == ~0ull) {
                // all 128 entries pass: records of the row are contiguous, two per lane
                u32x4 r;
                r.x = conn_tag; r.y = ea.x; r.z = conn_tag; r.w = e[2 * j + 1].x;
                *(u32x4 *)(void *)(out + n_out + 2 * lane) = r;
                if (opos) { opos[n_out + 2 * lane] = pos; opos[n_out + 2 * lane + 1] = pos + 1; }
                if (MASKS) { omask[n_out + 2 * lane] = mk0; omask[n_out + 2 * lane + 1] = mk1; }
                n_out += 128;
            } else {
                // entry order: records before this lane's pair = passing entries of lower lanes
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
                if (pass0) {
                    chd_fanout_rec r;
                    r.conn = conn_tag;
                    r.channel = ea.x;
                    out[at] = r;
                    if (opos) opos[at] = pos;
                    if (MASKS) omask[at] = mk0;
                }
                if (pass1) {
                    chd_fanout_rec r;
                    r.conn = conn_tag;
                    r.channel = e[2 * j + 1].x;
                    out[at + (pass0 ? 1u : 0u)] = r;
                    if (opos) opos[at + (pass0 ? 1u : 0u)] = pos + 1;
                    if (MASKS) omask[at + (pass0 ? 1u : 0u)] = mk1;
                }
                n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            }
        }
    }
    return n_out;
}

// The same stream over COMPACT 8-byte entries {channel, history}: used when all buffered updates of
// the cell come from one sender (the usual case: the spatial server that owns the cell), so the
// SkipSelfUpdateFanOut test is one scalar compare per subscription and the L2 read volume halves.
// A lane's adjacent entry pair is ONE 16-byte load; a step covers 512 entries with four loads.
template <bool FULL, bool MASKS = false>
__device__ __forceinline__ uint32_t emit_cell8(const uint2 *__restrict__ ce8, uint32_t start, uint32_t end, uint32_t wm,
                                               uint32_t conn_tag, chd_fanout_rec *__restrict__ out,
                                               uint32_t *__restrict__ opos, uint32_t n_out,
                                               uint32_t *__restrict__ omask = nullptr) {
    const uint32_t lane = lane_id();
    for (uint32_t b = start; b < end; b += 128 * FO_UNROLL8) {
        u32x4 e[FO_UNROLL8];
        const uint2 *p[FO_UNROLL8];
#pragma unroll
        for (int j = 0; j < FO_UNROLL8; j++) {
            const uint32_t pos = b + j * 128 + 2 * lane;
            p[j] = ce8 + (pos + 1 < end ? pos : (end >= start + 2 ? end - 2 : start));  // ce8 has spare entries
        }
#if FO_UNROLL8 == 4
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\t"
            "global_load_dwordx4 %1, %5, off\n\t"
            "global_load_dwordx4 %2, %6, off\n\t"
            "global_load_dwordx4 %3, %7, off\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
            : "memory");
#else
        asm volatile(
            "global_load_dwordx4 %0, %2, off\n\t"
            "global_load_dwordx4 %1, %3, off\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(e[0]), "=&v"(e[1])
            : "v"(p[0]), "v"(p[1])
            : "memory");
#endif
#pragma unroll
        for (int j = 0; j < FO_UNROLL8; j++) {
            if (end - b <= (uint32_t)(j * 128)) break;  // uniform
            const uint32_t pos = b + j * 128 + 2 * lane;
            const bool in0 = pos < end, in1 = pos + 1 < end;
            const uint32_t q = in1 ? pos : (end >= start + 2 ? end - 2 : start);
            const uint32_t chan_a = pos == q ? e[j].x : e[j].z, hist_a = pos == q ? e[j].y : e[j].w;
            const bool pass0 = in0 && (FULL || (hist_a & wm) != 0);
            const bool pass1 = in1 && (FULL || (e[j].w & wm) != 0);
            const uint64_t m0 = __ballot(pass0), m1 = __ballot(pass1);
            if ((m0 & m1) == ~0ull) {
                u32x4 r;
                r.x = conn_tag; r.y = chan_a; r.z = conn_tag; r.w = e[j].z;
                *(u32x4 *)(void *)(out + n_out + 2 * lane) = r;
                if (opos) { opos[n_out + 2 * lane] = pos; opos[n_out + 2 * lane + 1] = pos + 1; }
                if (MASKS) { omask[n_out + 2 * lane] = FULL ? 0u : (hist_a & wm); omask[n_out + 2 * lane + 1] = FULL ? 0u : (e[j].w & wm); }
                n_out += 128;
            } else {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
                if (pass0) {
                    chd_fanout_rec r;
                    r.conn = conn_tag;
                    r.channel = chan_a;
                    out[at] = r;
                    if (opos) opos[at] = pos;
                    if (MASKS) omask[at] = FULL ? 0u : (hist_a & wm);
                }
                if (pass1) {
                    chd_fanout_rec r;
                    r.conn = conn_tag;
                    r.channel = e[j].z;
                    out[at + (pass0 ? 1u : 0u)] = r;
                    if (opos) opos[at + (pass0 ? 1u : 0u)] = pos + 1;
                    if (MASKS) omask[at + (pass0 ? 1u : 0u)] = FULL ? 0u : (e[j].w & wm);
                }
                n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            }
        }
    }
    return n_out;
}

// Every entity of the cell passes the window (the AND of their histories intersects it and the cell has
// one sender that is not this connection — or this is the full state): no per-entry test, no ballot.
// A lane copies FOUR adjacent channel ids per step (one 16-byte load) into four records (two 16-byte
// stores): ~15 instructions per 256 records instead of ~250, and exactly the algorithmic 12 bytes per
// message of traffic.
__device__ __forceinline__ uint32_t emit_cell_all(const uint32_t *__restrict__ chans, uint32_t start, uint32_t end,
                                                  uint32_t conn_tag, chd_fanout_rec *__restrict__ out,
                                                  uint32_t *__restrict__ opos, uint32_t n_out,
                                                  uint32_t *__restrict__ omask = nullptr, long long *ce_prof = nullptr) {
    const uint32_t lane = lane_id();
    const uint32_t n = end - start;
#ifdef CHD_PROFILE_CONN_EMIT
    if (ce_prof) ce_prof[2] = clock64();  // [0] load wait, [1] store issue, [2] running mark
#endif
    // Measured on MI355X, config B: giving every lane two adjacent records per row so that each store instruction
    // writes one contiguous 1 KiB run was slower (217 vs 200 us: twice the load instructions for the same stores).
    // The lane that holds the cell's last 1-3 entries takes the SAME 16-byte load as everyone else (the column has
    // spare entries behind it) and only narrows its stores: a separate load for it came after the step's record
    // stores and, the vm counter being in-order, waited for all of them to drain (~6.6 K of ~13 K cycles per segment
    // in the per-phase profile, -DCHD_PROFILE_CONN_EMIT).
#if FO_COPY_AHEAD
    // both steps of a cell of up to 512 entries are loaded before the first store, for the same reason
    if (n <= 512) {
        const uint32_t k0 = 4 * lane, k1 = 256 + 4 * lane;
        u32x4 c[2];
        c[0] = *(const u32x4 *)(const void *)(chans + start + (k0 < n ? k0 : 0));
        c[1] = *(const u32x4 *)(const void *)(chans + start + (k1 < n ? k1 : 0));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t k = h ? k1 : k0;
            if (k >= n) continue;
            const u32x4 c4 = c[h];
            const uint32_t m = n - k;
            if (m >= 4) {
                u32x4 r0, r1;
                r0.x = conn_tag; r0.y = c4.x; r0.z = conn_tag; r0.w = c4.y;
                r1.x = conn_tag; r1.y = c4.z; r1.z = conn_tag; r1.w = c4.w;
                u32x4 *o = (u32x4 *)(void *)(out + n_out + k);
                o[0] = r0;
                o[1] = r1;
                if (opos) {
                    u32x4 p4;
                    p4.x = start + k; p4.y = start + k + 1; p4.z = start + k + 2; p4.w = start + k + 3;
                    *(u32x4 *)(void *)(opos + n_out + k) = p4;
                }
                if (omask) {
                    u32x4 z4;
                    z4.x = 0; z4.y = 0; z4.z = 0; z4.w = 0;
                    *(u32x4 *)(void *)(omask + n_out + k) = z4;
                }
            } else {
                const uint32_t cc[3] = {c4.x, c4.y, c4.z};
#pragma unroll
                for (uint32_t q = 0; q < 3; q++) {
                    if (q < m) {
                        chd_fanout_rec r;
                        r.conn = conn_tag;
                        r.channel = cc[q];
                        out[n_out + k + q] = r;
                        if (opos) opos[n_out + k + q] = start + k + q;
                        if (omask) omask[n_out + k + q] = 0;
                    }
                }
            }
        }
        return n_out + n;
    }
#endif
    for (uint32_t b = 0; b < n; b += 256) {
        const uint32_t k = b + 4 * lane;      // this lane's first entry of the step, relative to start
        if (k < n) {
            const u32x4 c4 = *(const u32x4 *)(const void *)(chans + start + k);  // (up to 3 entries past the cell)
#ifdef CHD_PROFILE_CONN_EMIT
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(c4) : "memory");
            if (ce_prof) { long long _t = clock64(); ce_prof[0] += _t - ce_prof[2]; ce_prof[2] = _t; }
#endif
            const uint32_t m = n - k;             // entries of the cell this lane holds (>= 1)
            if (m >= 4) {
                u32x4 r0, r1;
                r0.x = conn_tag; r0.y = c4.x; r0.z = conn_tag; r0.w = c4.y;
                r1.x = conn_tag; r1.y = c4.z; r1.z = conn_tag; r1.w = c4.w;
                u32x4 *o = (u32x4 *)(void *)(out + n_out + k);
                o[0] = r0;
                o[1] = r1;
                if (opos) {
                    u32x4 p4;
                    p4.x = start + k; p4.y = start + k + 1; p4.z = start + k + 2; p4.w = start + k + 3;
                    *(u32x4 *)(void *)(opos + n_out + k) = p4;
                }
                if (omask) {  // (full-state records merge nothing: the only caller that passes masks)
                    u32x4 z4;
                    z4.x = 0; z4.y = 0; z4.z = 0; z4.w = 0;
                    *(u32x4 *)(void *)(omask + n_out + k) = z4;
                }
            } else {
                const uint32_t cc[3] = {c4.x, c4.y, c4.z};
#pragma unroll
                for (uint32_t q = 0; q < 3; q++) {
                    if (q < m) {
                        chd_fanout_rec r;
                        r.conn = conn_tag;
                        r.channel = cc[q];
                        out[n_out + k + q] = r;
                        if (opos) opos[n_out + k + q] = start + k + q;
                        if (omask) omask[n_out + k + q] = 0;
                    }
                }
            }
#ifdef CHD_PROFILE_CONN_EMIT
            if (ce_prof) { long long _t = clock64(); ce_prof[1] += _t - ce_prof[2]; ce_prof[2] = _t; }
#endif
        }
    }
    return n_out + n;
}

// WAVES waves per connection (= subscriptions staged in LDS per round / 64).  Measured at config B (10 K connections,
// ~18 due subscriptions each): 192.8 / 194.4 / 196.8 / 227 us per launch with 1 / 2 / 4 / 8 waves; the launcher takes
// one wave per connection when there are enough connections to fill the chip that way, four otherwise.
// DEFERRED: second launch behind k_fanout_emit_seg — only the subscriptions k_fanout_plan_seg marked PF_DEFER (their
// connections are flagged in conn_defer; every other workgroup exits at once), record counts ADDED to the first launch's.
template <int WAVES, bool MASKS, bool DEFERRED = false>
__device__ __forceinline__ void fanout_emit_conn(const DevGrid &g, const WorldDev &w, int64_t now, const TickRing &ring, const uint32_t s) {
    constexpr uint32_t FO_TILE = 64 * WAVES;
    // due subscriptions of this connection, staged once per workgroup so that the
    // streaming waves never wait on per-subscription pointer chasing
    __shared__ uint32_t d_p[FO_TILE], d_fl[FO_TILE], d_c[FO_TILE], d_start[FO_TILE], d_end[FO_TILE], d_rel[FO_TILE],
        d_chh[FO_TILE], d_chs[FO_TILE], d_iv[FO_TILE], d_chhp[FO_TILE], d_chsp[FO_TILE], d_us[FO_TILE], d_hand[FO_TILE];
    __shared__ int64_t d_L[FO_TILE];
    __shared__ uint32_t n_due, ticket;
    __shared__ uint32_t wave_total[WAVES];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!DEFERRED && !w.sub_alive[s]) {
        if (threadIdx.x == 0) w.rec_cnt[s] = 0;
        return;
    }
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const uint64_t base = w.rec_ub[s];
    // (DEFERRED: k_fanout_emit_tail calls this only for connections that have room — fanout_no_room takes the others)
    if (!DEFERRED && w.rec_ub[s + 1] > w.recs_cap) {
        // no room for this connection's worst case: leave its state untouched, it
        // catches up next tick (the reference's catch-up loop), and say so.
        for (uint32_t p = threadIdx.x; p < cnt; p += 64 * WAVES) w.pair_nrec[pbase + p] = 0;
        if (threadIdx.x == 0) {
            w.rec_cnt[s] = 0;
            if (w.rec_ub[s + 1] > base) atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
        }
        return;
    }
#ifdef CHD_PROFILE_CONN_EMIT
    long long ce_mark = clock64(), ce_pro = 0, ce_stage = 0, ce_ticket = 0, ce_decide = 0, ce_stream = 0, ce_tail = 0, ce_ls[3] = {0, 0, 0};
    uint32_t ce_nseg = 0;
#endif
    const uint32_t conn = w.conn_id[s];
    const int64_t my_t = ring_stamp(ring);
    // (wave-uniform, but read through a dynamic index: keep it on the scalar side)
    const int64_t oldest_v = ring.n ? ring.t[ring.n - 1] : INT64_MAX;
    const int64_t oldest = ((int64_t)__builtin_amdgcn_readfirstlane((int)(oldest_v >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)oldest_v);
    const uint4 *__restrict__ ce = w.ce_view;
    uint32_t total = 0;
    uint32_t hist_ovf = 0;
    CE_WAIT_VM();
    CE_MARK(ce_pro);
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
                if (DEFERRED ? (fl & PF_DEFER) != 0 : (!(fl & (PF_NO_ACCESS | PF_DEEP)) && I > 0 && now >= L + I)) {
                    const uint32_t c = w.pair_cell[pbase + p];
                    const uint32_t k = atomicAdd(&n_due, 1u);
                    d_p[k] = p; d_fl[k] = fl & ~PF_DEFER; d_L[k] = L; d_iv[k] = iv; d_c[k] = c;
                    d_rel[k] = w.pair_rel[pbase + p];
                    d_start[k] = w.cell_start[c];
                    d_end[k] = w.cell_end[c];
                    const uint32_t age = ring.cur_tick - w.cell_hist_tick[c];
                    d_chh[k] = age < CHD_HIST_BITS ? (w.cell_hist[c] << age) : 0u;
                    d_chs[k] = w.cell_sender[c];
                    d_chhp[k] = age < CHD_HIST_BITS ? (w.cell_hist_prev[c] << age) : 0u;
                    d_chsp[k] = w.cell_sender_prev[c];
                    uint32_t us = w.ce8_view ? w.cell_usender[c] : CHD_NONUNIFORM;
                    // Several senders behind the cell's buffered updates, but this connection is none of them (outside
                    // their id range) or does not skip its own: SkipSelfUpdateFanOut cannot drop anything, so the
                    // sender-agnostic copy / compact streams apply.  (Entities keep the sender of the server that
                    // spawned them: after some handovers about half of config B's cells hold two servers' entities,
                    // and the subscribers are clients.)
                    if (w.ce8_view && us == CHD_NONUNIFORM &&
                        (!(fl & PF_SKIP_SELF) || conn < w.cell_smin[c] || conn > w.cell_smax[c]))
                        us = CHD_NOT_A_SENDER;
                    d_us[k] = us;
                    d_hand[k] = w.ce_chan_view ? w.cell_hand[c] : 0u;
                } else if (!DEFERRED && !(fl & PF_DEEP)) {
                    w.pair_nrec[pbase + p] = 0;
                }
            }
        }
        __syncthreads();
        CE_MARK(ce_stage);
        const uint32_t ndue = n_due;
        for (;;) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&ticket, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= ndue) break;
#ifdef CHD_PROFILE_CONN_EMIT
            ce_nseg++;
#endif
            const uint32_t p = d_p[k];
            uint32_t fl = d_fl[k];
            int64_t L = d_L[k];
            const int64_t I = (int64_t)d_iv[k] * 1000000;
            const uint32_t c = d_c[k];
            const uint32_t start = d_start[k], end = d_end[k];
            const bool skip_self = (fl & PF_SKIP_SELF) != 0;
            const uint32_t us = d_us[k];
            chd_fanout_rec *__restrict__ out = w.recs + base + d_rel[k];
            // wire mode: which cell-table entry (or, with bit 31, which spatial channel) each record came from
            uint32_t *__restrict__ opos = w.rec_pos ? w.rec_pos + base + d_rel[k] : nullptr;
            // CHD_WORLD_UPDATE_MASKS: which buffered updates each record's message merges
            uint32_t *__restrict__ omask = MASKS ? w.rec_mask + base + d_rel[k] : nullptr;
            uint32_t n_out = 0;
            CE_WAIT_LGKM();
            CE_MARK(ce_ticket);
            if (!(fl & PF_HAD_FIRST)) {
                // first fan-out: the whole data of the spatial channel and of every
                // entity channel in it (data.go:217-223); last = t
                if (lane == 0) {
                    chd_fanout_rec r;
                    r.conn = conn | CHD_REC_FULL;
                    r.channel = c + g.id_start;
                    out[0] = r;
                    if (opos) opos[0] = CHD_POS_CELL | c;
                    if (MASKS) omask[0] = 0;
                }
                n_out = w.ce_chan_view ? emit_cell_all(w.ce_chan_view, start, end, conn | CHD_REC_FULL, out, opos, 1u, omask)
                                       : emit_cell<true, MASKS>(w, ce, start, end, 0u, false, conn, conn | CHD_REC_FULL, out, opos, 1u, omask);
                fl |= PF_HAD_FIRST;
                L = now;
            }
            // catch-up windows (data.go:224-271 + the revisit through :273-286)
            if (now >= L + I) {
                if (history_lost(ring, oldest, L, I)) hist_ovf = 1;
                const uint32_t ch_hist = d_chh[k];
                const uint32_t ch_sender = d_chs[k];
                while (now >= L + I) {
                    const int64_t next = L + I;
                    const int64_t lo = L > 0 ? L : 0;  // lastUpdateTime starts at max(last, 0)
                    const uint32_t wm = window_mask(my_t, lo, next);
                    if (!wm) {
                        L += empty_windows(ring, my_t, now, L, d_iv[k]) * I;
                        continue;
                    }
                    {
                        // the spatial channel's own buffered updates
                        if (cell_update_passes(ch_hist, ch_sender, d_chhp[k], d_chsp[k], wm, skip_self, conn)) {
                            if (lane == 0) {
                                chd_fanout_rec r;
                                r.conn = conn;
                                r.channel = c + g.id_start;
                                out[n_out] = r;
                                if (opos) opos[n_out] = CHD_POS_CELL | c;
                                if (MASKS) omask[n_out] = cell_update_mask(ch_hist, ch_sender, d_chhp[k], d_chsp[k], wm, skip_self, conn);
                            }
                            n_out += 1;
                        }
                        CE_MARK(ce_decide);
                        if (us == CHD_NONUNIFORM) n_out = emit_cell<false, MASKS>(w, ce, start, end, wm, skip_self, conn, conn, out, opos, n_out, omask);
                        else if (!(skip_self && us == conn))
#ifdef CHD_PROFILE_CONN_EMIT
                            n_out = (!MASKS && (d_hand[k] & wm)) ? emit_cell_all(w.ce_chan_view, start, end, conn, out, opos, n_out, nullptr, ce_ls)
#else
                            n_out = (!MASKS && (d_hand[k] & wm)) ? emit_cell_all(w.ce_chan_view, start, end, conn, out, opos, n_out)
#endif
                                                                 : emit_cell8<false, MASKS>(w.ce8_view, start, end, wm, conn, out, opos, n_out, omask);
                        CE_MARK(ce_stream);
                    }
                    L = next;
                }
            }
            CE_MARK(ce_decide);
            pad_segment(out, n_out);
            if (lane == 0) {
                w.pair_last[pbase + p] = L;
                w.pair_flags[pbase + p] = fl;
                w.pair_nrec[pbase + p] = n_out;
            }
            total += n_out;
            CE_MARK(ce_tail);
        }
        __syncthreads();
    }
#ifdef CHD_PROFILE_CONN_EMIT
    if (lane == 0 && ring.cur_tick == 60 && s % 997u == 0)
        printf("conn_emit s %u wave %u: segs %u records %u | cycles: prologue %lld stage %lld ticket+desc %lld decide %lld stream %lld "
               "(load wait %lld, stores %lld) tail %lld\n", s, wave, ce_nseg, total, ce_pro, ce_stage, ce_ticket, ce_decide, ce_stream,
               ce_ls[0], ce_ls[1], ce_tail);
#endif
    if (lane == 0) wave_total[wave] = total;
    if (hist_ovf && lane == 0) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < WAVES; k++) t += wave_total[k];
        if (DEFERRED) { if (t) atomicAdd(&w.rec_cnt[s], t); }  // (beside the element walk of the same connection: k_fanout_tail)
        else w.rec_cnt[s] = t;
        // per-tick totals go through 64 hashed counters, one 128-byte line each: a
        // single word (or words sharing a line) would serialise S atomics at ~12 ns
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16];
        if (t) atomicAdd(slot, (unsigned long long)t);
        if (t && DEFERRED) atomicAdd(slot + 2, (unsigned long long)t);  // (bench.py: what the descriptor kernel did NOT write)
        if (cnt && !DEFERRED) atomicAdd(slot + 1, (unsigned long long)cnt);
    }
}

template <int WAVES, bool MASKS>
__global__ void __launch_bounds__(64 * WAVES) k_fanout_emit(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    fanout_emit_conn<WAVES, MASKS, false>(g, w, now, ring, blockIdx.x);
}

// Descriptor path: a connection without room for its worst case this tick (rec_ub[s + 1] > recs_cap; the record kernels skip it without
// a word): nothing of it was served, so the fan-out state the plan has already committed for its simple and filtered subscriptions
// goes back to what the descriptors kept (seg_ln / seg_desc2.z), its record counts to zero, and the tick says so.  It catches up
// next tick (the reference's catch-up loop).  One wave.
__device__ __forceinline__ void fanout_no_room(const WorldDev &w, uint32_t s) {
    const uint32_t lane = lane_id();
    const size_t pbase = (size_t)s * w.capq;
    const uint32_t cnt = w.sub_alive[s] ? w.pair_cnt[s] : 0u;
    const uint32_t ns = w.n_simple[s];
    for (uint32_t k = lane; k < ns; k += 64) {
        const uint4 d2 = w.seg_desc2[pbase + k];
        w.pair_last[pbase + d2.y] = w.seg_ln[pbase + k];
        w.pair_flags[pbase + d2.y] = d2.z;
    }
    if (w.off_on) {
        const uint32_t nf = w.n_filt[s];
        for (uint32_t k = lane; k < nf; k += 64) {
            const uint4 d2 = w.filt_desc2[pbase + k];
            w.pair_last[pbase + d2.y] = w.filt_ln[pbase + k];
            w.pair_flags[pbase + d2.y] = d2.z;
        }
    }
    if (w.rec_ub[s + 1] > w.rec_ub[s]) {
        for (uint32_t p = lane; p < cnt; p += 64) w.pair_nrec[pbase + p] = 0;
        if (lane == 0) {
            // (k_fanout_plan_seg has already counted the records it planned for this connection into the tick's total)
            if (w.rec_cnt[s]) atomicAdd((unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16], 0ull - (unsigned long long)w.rec_cnt[s]);
            w.rec_cnt[s] = 0;
            atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
        }
    }
}

// ---------------------------------------------------------------------------
// Connection-major emit, descriptor-driven form (the launcher takes it when one wave per connection would fill the
// chip — or CHD_WORLD_ONE_WAVE_EMIT asks for it — and no per-record masks are wanted).
//
// What bounded k_fanout_emit at config B (measured): not the record stores (a store-only kernel with
// the same 10 K x 18 segment pattern takes ~125 us) and not occupancy, but (i) per-segment latency — window walk
// and two dependent L2 round trips per segment, every load wait draining the wave's own record stores because the
// vm counter is in-order — (ii) ~12 us of dependent gathers per connection before its first store, taken while the
// store stream saturates the memory pipeline, and (iii) a tail: one wave per connection lives ~90 us of a ~170 us
// launch, so the last ~80 us drain at falling occupancy with every remaining wave latency-bound.
//
//   k_fanout_plan_seg : one wave per connection, one lane per subscription, BEFORE the emit (no store stream to
//                   contend with): due test, the whole catch-up window walk of tickData (serial over the tick ring,
//                   up to four window masks), classification.  A subscription is SIMPLE when every non-empty window
//                   is a plain copy of the cell's channel-id column (the AND of the entities' histories intersects
//                   it and SkipSelfUpdateFanOut cannot drop anything) or nothing at all: then its record count is
//                   known exactly (tight segment, no worst-case slot) and it becomes a 16-byte descriptor
//                   {segment offset, column start, entries, windows / own-update bits}.  Everything else (a window
//                   some entity has no update in, a cell whose senders include this connection, more than four
//                   non-empty windows, a cell of more than 512 entities) is marked PF_DEFER with a worst-case slot and left to the
//                   second launch, k_fanout_emit<.., DEFERRED>, whose filtering streams would cost this path 50 VGPRs.
//   k_fanout_emit_seg : WAVES waves per connection take its descriptors round-robin: scalar descriptor load,
//                   the column of the NEXT descriptor requested (four 8-byte loads per lane: pairs of adjacent
//                   entries, so every store instruction writes one contiguous 1 KiB run of records) before the
//                   current segment's records are stored, awaited with a COUNTED s_waitcnt — `vmcnt(K)` with K <= the
//                   stores issued since completes the older loads and leaves those stores in flight.  No LDS, no
//                   barrier, no decision; wave 0 commits the subscriptions' new fan-out state in one coalesced pass.
// Same records, same segment order inside a connection's range, same state as k_fanout_emit: the parity tests run
// both (tests/test_gpu_world.py "conn-major-1w", tests/test_gpu_fullsize.py).
// ---------------------------------------------------------------------------
#define SD_NWIN_MASK 7u
#define SD_FIRST 8u      // first fan-out: the cell's FULL record + every entity's
#define SD_NONE 16u      // SkipSelfUpdateFanOut and the cell's only sender is this connection: no entity record passes
#define SD_NOPAD 32u     // another part of the same subscription follows in the next records: do not pad the line
#define SD_OWN_SHIFT 8   // bit 8 + j: the spatial channel's own buffered update passes window j

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifndef FO_SEG_WAVES
#define FO_SEG_WAVES 2   // waves per connection in k_fanout_emit_seg
#endif
#ifndef FO_SEG_TAIL_PCT
#define FO_SEG_TAIL_PCT 10  // k_fanout_emit_seg: the last FO_SEG_TAIL_PCT % of the connections are cut into 2^FO_SEG_TAIL_SH pieces (CHD_SEG_TAIL)
#define FO_SEG_TAIL_SH 2
#endif
#ifndef FO_SEG_OCC
#define FO_SEG_OCC 4     // waves per SIMD the register allocator is asked for (what limits this kernel is bytes in flight per wave, not waves)
#endif

// ---- sub-tick arrival offsets (WorldDev::off_on): the window walk of tickData against the REAL arrival stamps ----
// A due subscription's windows [max(L,0) + kI, L + (k+1)I] against the ring slots: slot j holds the updates that arrived inside
// (t[j+1], t[j]] at stamp t[j] - offset.  Per window and slot the offsets that lie inside the window are [A, B] = [t[j] - hi,
// t[j] - lo]; with the cell's offset range of the slot (cell_orng) the slot is covered WHOLE (every update of it passes: the
// slot's bit joins the window's mask, as a stamp inside the window does on the tick grid), not at all, or CUT by a window edge —
// then the entities need a per-entity compare (FiltWin) and the subscription becomes a filtered descriptor.  Slots older than
// CHD_OFF_SLOTS keep no offsets: whole / not at all by the slot's own interval, else the exact buffers decide (deep).
struct OffPlan {
    uint32_t full[4];   // the first four non-empty windows' whole-slot masks (what the copy path needs)
    uint32_t nw;        // non-empty windows (<= CHD_FILT_WINS); their tests are in filt_win[(pbase + p) * CHD_FILT_WINS + k]
    uint32_t own;       // bit k: the spatial channel's own buffered update passes window k
    bool need;          // some window is not a plain copy of the cell's column: the per-entity compare
    bool deep;          // undecidable here: too many non-empty windows, a cut through a slot without offsets, ...
    int64_t Lw;         // lastFanOutTime after the walk
};

__device__ __forceinline__ void plan_windows_off(const WorldDev &w, const TickRing &ring, int64_t now, int64_t L, int64_t I, uint32_t c,
                                                 uint32_t hand, uint32_t chh, uint32_t chs, uint32_t chhp, uint32_t chsp, bool skip_self,
                                                 uint32_t conn, FiltWin *__restrict__ fwout, OffPlan &o, uint32_t cell_age,
                                                 const uint4 &r0, const uint4 &r1, const uint4 &r2, const uint4 &r3, const uint4 &q0, const uint4 &q1) {
    o.nw = 0; o.own = 0; o.need = false; o.deep = false; o.Lw = L;
#pragma unroll
    for (int k = 0; k < 4; k++) o.full[k] = 0;
    // (r0..r3: the cell's offset ranges, cell_orng; q0, q1: its own channel's offsets, cell_ooff — requested by the caller with its other
    // per-cell words)
    const uint32_t rmin[CHD_OFF_SLOTS] = {r0.x, r0.z, r1.x, r1.z, r2.x, r2.z, r3.x, r3.z};
    const uint32_t rmax[CHD_OFF_SLOTS] = {r0.y, r0.w, r1.y, r1.w, r2.y, r2.w, r3.y, r3.w};
    // (the spatial channel's offsets are stored aligned to the tick of ITS last update, like its history bits: moved on to this tick —
    // round 4 compared them unaligned, which delivered a cell's update of tick t - 1 once more at tick t to every window that starts
    // exactly at t - 1 whenever the cell had no update at t; found by the sharded worlds' cell-update test, wrong on one GPU as well)
    uint32_t coff[CHD_OFF_SLOTS] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    off_shift(coff, cell_age < CHD_OFF_SLOTS ? cell_age : CHD_OFF_SLOTS);
    const int64_t nwin = div_windows_ns(now - L, I);
    if (nwin > 64) { o.deep = true; return; }  // (a long catch-up: the element walk handles any number of windows)
    // The ring slots ANY lane of the wave can reach: the stamps do not increase with the slot index and no window starts before the
    // lane's max(L, 0), so slots older than the wave's oldest start are skipped by a uniform branch — subscriptions served every
    // tick reach two or three of the eight, and the unrolled slot loop below was what made this kernel instruction-bound (40 us).
    uint32_t jn = 0;
    {
        const int64_t lo0 = L > 0 ? L : 0;
#pragma unroll
        for (int j = 0; j < (int)CHD_OFF_SLOTS; j++)
            if ((uint32_t)j < ring.n && __ballot(ring.t[j] >= lo0) != 0) jn = (uint32_t)j + 1u;
    }
    // THE COMMON CASE IN 32 BITS.  Everything above is wave-uniform or per lane in nanoseconds as int64; when no lane of the wave reaches
    // further back than 2^31 ns (2.1 s — every subscription that is served regularly), the same comparisons are made on 32-bit
    // distances from `now`: dj = now - t[j] and dp = now - t[j + 1] per slot (scalars, saturated), dlo = now - lo and dhi = now - hi
    // per lane and window.  tj < lo <=> dj > dlo; tp >= hi <=> dp <= dhi; the offsets inside the window are [t[j] - hi, t[j] - lo] =
    // [dhi - dj (or 0), dlo - dj].  Half the vector instructions of the 64-bit form, which was what this kernel's time consisted of
    // (40 us with the classification, 15 without).
    const bool fast = __ballot((uint64_t)(now - (L > 0 ? L : 0)) >= (1ull << 31)) == 0;
    uint32_t dj[CHD_OFF_SLOTS], dp[CHD_OFF_SLOTS];
#pragma unroll
    for (int j = 0; j < (int)CHD_OFF_SLOTS; j++) {
        const uint64_t a = (uint32_t)j < ring.n ? (uint64_t)(now - ring.t[j]) : ~0ull;
        const uint64_t b = (uint32_t)(j + 1) < ring.n ? (uint64_t)(now - ring.t[(j + 1) & (CHD_HIST_BITS - 1)]) : ~0ull;  // (last slot: tp = -1, never >= hi)
        dj[j] = a > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a;
        dp[j] = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
    }
    int64_t Lw = L;
    for (int64_t k = 0; k < nwin; k++) {
        const int64_t hi = Lw + I, lo = Lw > 0 ? Lw : 0;
        uint32_t fm = 0, om = 0, sa = 0, sb = 0, alo = 1, ahi = 0, blo = 1, bhi = 0, ncut = 0;
        bool undecided = false;
        // one ring slot against the window, given the offsets [A, B] of the slot's updates that lie inside it (`act`: the slot and the
        // window overlap at all).  BRANCH-FREE on purpose (0 / 1 values combined with & and |, selects): per-lane tests compile to
        // exec-masked branches, a dozen per slot — what this kernel's time consisted of, not the arithmetic (40 us with the
        // classification, 15 without it; halving the arithmetic's width changed nothing).
        auto slot = [&](const int j, const bool act, const uint32_t A, const uint32_t B) {
            const bool has = act & (rmin[j] <= rmax[j]);  // (the cell holds updates of this slot)
            const bool whole = has & (A <= rmin[j]) & (rmax[j] <= B);
            const bool cut = has & !whole & !((rmax[j] < A) | (rmin[j] > B));
            fm |= whole ? 1u << j : 0u;
            const bool c0 = cut & (ncut == 0u), c1 = cut & (ncut == 1u);
            undecided |= cut & (ncut >= 2u);
            sa = c0 ? (uint32_t)j : sa; alo = c0 ? A : alo; ahi = c0 ? B : ahi;
            sb = c1 ? (uint32_t)j : sb; blo = c1 ? A : blo; bhi = c1 ? B : bhi;
            ncut += cut ? 1u : 0u;
            om |= (act & (A <= coff[j]) & (coff[j] <= B)) ? 1u << j : 0u;
        };
        if (fast) {
            const uint32_t dlo = (uint32_t)(now - lo), dhi = (uint32_t)(now - hi);  // (hi <= now: the window is due)
#pragma unroll
            for (int j = 0; j < (int)CHD_OFF_SLOTS; j++) {
                if ((uint32_t)j >= jn) break;  // (uniform)
                // (older than the window: dj > dlo; newer: dp <= dhi; the ring's last slot of a full ring: the evicted stamp below it is unknown)
                const bool overlap = (dj[j] <= dlo) & (dp[j] > dhi);
                const bool blind = (uint32_t)(j + 1) >= ring.n && ring.n == CHD_HIST_BITS;  // (uniform)
                undecided |= overlap & blind;
                slot(j, overlap & !blind, dhi > dj[j] ? dhi - dj[j] : 0u, dlo - dj[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < (int)CHD_OFF_SLOTS; j++) {
                if ((uint32_t)j >= jn) break;  // (uniform)
                const int64_t tj = ring.t[j];
                const bool last = (uint32_t)(j + 1) >= ring.n;
                const int64_t tp = last ? -1 : ring.t[j + 1];  // arrivals of the slot are > tp (the world's first tick: >= 0)
                const bool overlap = (tj >= lo) & (tp < hi);   // (else the whole slot is older / newer than the window)
                const bool blind = last && ring.n == CHD_HIST_BITS;
                undecided |= overlap & blind;
                const int64_t A64 = tj - hi > 0 ? tj - hi : 0, B64 = tj - lo;
                // (offsets are at most 0xFFFFFFFE: WorldDev::off_on)
                slot(j, overlap & !blind & (A64 <= 0xFFFFFFFEll), (uint32_t)A64, B64 > 0xFFFFFFFEll ? 0xFFFFFFFEu : (uint32_t)B64);
            }
        }
        for (uint32_t j = CHD_OFF_SLOTS; j < ring.n; j++) {  // slots without offsets: by their own interval
            const int64_t tj = ring.t[j];
            if (tj < lo) break;
            const bool last = j + 1 >= ring.n;
            const int64_t tp = last ? -1 : ring.t[j + 1];
            if (tp >= hi) continue;
            if (last && ring.n == CHD_HIST_BITS) { undecided = true; continue; }
            // every arrival of the slot inside the window?  (tp, tj] within [lo, hi]
            if (tp + 1 >= lo && tj <= hi) { fm |= 1u << j; om |= 1u << j; }
            else undecided = true;
        }
        if (undecided) { o.deep = true; return; }
        Lw = hi;
        const bool ownp = cell_update_passes(chh, chs, chhp, chsp, om, skip_self, conn);
        if (!fm && !ncut && !ownp) continue;  // nothing can pass this window
        if (o.nw == CHD_FILT_WINS) { o.deep = true; return; }
        FiltWin fw;
        fw.full = fm; fw.slots = sa | (sb << 8); fw.a_lo = alo; fw.a_hi = ahi; fw.b_lo = blo; fw.b_hi = bhi;
        fwout[o.nw] = fw;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if ((uint32_t)q == o.nw) o.full[q] = fm;
        if (ownp) o.own |= 1u << o.nw;
        // a plain copy of the cell's column only when some slot the window covers WHOLE holds an update of every entity
        if (!(hand & fm)) o.need = true;
        o.nw++;
    }
    o.Lw = Lw;
}

template <bool OFF>
__global__ void __launch_bounds__(64 * FO_WAVES) k_fanout_plan_seg(DevGrid g, WorldDev w, int64_t now, TickRing ring) {
    if (blockIdx.x == 0 && threadIdx.x < 24) w.emit_ticket[32u * (threadIdx.x & 7u) + (threadIdx.x >> 3)] = 0;  // (this tick's k_fanout_emit_seg / _filt / _filt_cm start after this kernel)
    const uint32_t s = blockIdx.x * FO_WAVES + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = lane_id();
    uint64_t carry = 0;
    uint32_t n_simple = 0, n_filt = 0, any_deferred = 0, hist_ovf = 0, any_deep = 0;
    unsigned long long rec_simple = 0;
    const uint32_t cnt = w.sub_alive[s] ? w.pair_cnt[s] : 0u;
    const size_t pbase = (size_t)s * w.capq;
    const uint32_t conn = w.conn_id[s];
    const uint32_t *__restrict__ chans = w.ce_chan_view;
    const int64_t oldest = ring.n ? ring.t[ring.n - 1] : INT64_MAX;
    for (uint32_t p0 = 0; p0 < cnt; p0 += 64) {
        const uint32_t p = p0 + lane;
        uint64_t ub = 0;
        bool due = false, simple = false, deep = false, filt = false;
        uint32_t fl = 0, c = 0, size = 0, start = 0, info = 0, count = 0;
        uint32_t nd = 0, wcolp = 0, own = 0, ncol0 = 0;  // descriptors of this subscription; per window its column (4 bits each)
        uint32_t nwin_s = 0;                               // its non-empty windows, when it is cut into chunks (cells beyond 512 entries)
        uint4 wm4 = make_uint4(0u, 0u, 0u, 0u);          // (per-record masks) the windows' masks
        int64_t Lw = 0, Lold = 0;
        OffPlan op;
        op.nw = 0; op.own = 0; op.need = false; op.deep = false; op.Lw = 0;
        // Two round trips for everything a lane may read: the subscription's own words, then — by its cell — every per-cell word
        // (bounds, senders, histories, with offsets: ranges and the cell's own).  All unconditional (a lane without a subscription reads
        // the row's first one, a cell index is clamped): with each of them behind "due?", "deep?", "with offsets?" the kernel was a chain
        // of five dependent trips per wave, and it is as long as that chain (one wave per connection, a round of them per CU).
        const uint32_t pp = p < cnt ? p : 0u;
        const uint32_t fl_raw = w.pair_flags[pbase + pp], iv_raw = w.pair_iv[pbase + pp];
        const int64_t L_raw = w.pair_last[pbase + pp];
        const uint32_t c_raw = min(w.pair_cell[pbase + pp], g.ncell - 1u);
        const uint32_t pc_start = w.cell_start[c_raw], pc_end = w.cell_end[c_raw];
        const uint32_t *irr_p = w.deep_depth ? w.cell_irr : w.cell_start, *cov_p = w.cell_cov ? w.cell_cov : w.cell_start;
        const uint32_t pc_irr = irr_p[c_raw], pc_cov = cov_p[c_raw];
        const uint32_t c_us = w.cell_usender[c_raw], c_smin = w.cell_smin[c_raw], c_smax = w.cell_smax[c_raw], c_hand = w.cell_hand[c_raw];
        const uint32_t c_htick = w.cell_hist_tick[c_raw], c_hist = w.cell_hist[c_raw], c_hprev = w.cell_hist_prev[c_raw];
        const uint32_t chs = w.cell_sender[c_raw], chsp = w.cell_sender_prev[c_raw];
        uint4 o_r0 = make_uint4(0u, 0u, 0u, 0u), o_r1 = o_r0, o_r2 = o_r0, o_r3 = o_r0, o_q0 = o_r0, o_q1 = o_r0;
        if (OFF) {
            const uint4 *rp = (const uint4 *)(const void *)(w.cell_orng + (size_t)c_raw * CHD_OFF_SLOTS);
            o_r0 = rp[0]; o_r1 = rp[1]; o_r2 = rp[2]; o_r3 = rp[3];
            o_q0 = w.cell_ooff[2 * (size_t)c_raw]; o_q1 = w.cell_ooff[2 * (size_t)c_raw + 1];
        }
        if (p < cnt) {
            fl = fl_raw & ~(PF_DEFER | PF_DEEP);
            const int64_t L = L_raw;
            Lold = L;
            const uint32_t iv = iv_raw;
            const int64_t I = (int64_t)iv * 1000000;
            // data.go:194-197: NO_ACCESS is skipped but stays queued
            due = !(fl & PF_NO_ACCESS) && I > 0 && now >= L + I;
            if (due) {
                c = c_raw;
                if (w.cell_cov && !pc_cov) atomicOr(&w.counters[CTR_OVERFLOW], OVF_HALO);  // (see k_fanout_plan)
                start = pc_start;
                size = pc_end - start;
                if (sub_is_deep_pre(w, ring, oldest, fl, L, I, pc_irr)) {  // exact update buffers: k_fanout_emit_deep's
                    deep = true;
                    due = false;
                    ub = deep_upper_bound(w, now, L, I, size);
                    w.pair_flags[pbase + p] = fl | PF_DEEP;
                }
            }
            if (due) {
                uint32_t wms[4] = {0, 0, 0, 0}, nw = 0;
                Lw = L;
                if (!(fl & PF_HAD_FIRST)) {  // data.go:217-223: full state, last = t
                    info |= SD_FIRST;
                    Lw = now;
                    ub = (uint64_t)size + 1;  // (the worst case of a deferred first fan-out, as k_fanout_plan)
                } else {
                    int64_t nwin = div_windows(now - L, iv);
                    const int64_t lim = 2 * (int64_t)ring.n;  // a stamp lies in at most two windows
                    if (nwin > lim) nwin = lim;
                    ub = (uint64_t)nwin * ((uint64_t)size + 1);
                }
                bool hlost = false;
                if (OFF) {
                    // (the walk against the real arrival stamps comes below, once the cell's words are loaded: plan_windows_off)
                } else
                if (now >= Lw + I) {
                    hlost = history_lost(ring, oldest, Lw, I);
                    while (now >= Lw + I) {  // data.go:224-271 + the revisit through :273-286
                        const int64_t next = Lw + I, lo = Lw > 0 ? Lw : 0;
                        // Window [lo, next] over the ring.  The stamps do not increase with the slot index, so the scan
                        // stops at the first stamp older than the window — one or two slots for a subscription that is
                        // served every tick (a full 32-slot scan per window and lane made this kernel VALU-bound: 44 us).
                        uint32_t wm = 0, newer = 0;
                        int64_t ts_newer = 0;
                        for (uint32_t j = 0; j < ring.n; j++) {
                            const int64_t tj = ring.t[j];
                            if (tj < lo) break;
                            if (tj <= next) wm |= 1u << j;
                            else { newer++; ts_newer = tj; }  // (the last one kept = the oldest stamp newer than the window)
                        }
                        if (!wm) {
                            Lw += empty_windows_to(ts_newer, newer, now, Lw, iv) * I;
                            continue;
                        }
                        if (nw < 4) wms[nw] = wm;
                        nw++;
                        Lw = next;
                    }
                }
                uint32_t us = w.ce8_view ? c_us : CHD_NONUNIFORM;
                const bool skip_self = (fl & PF_SKIP_SELF) != 0;
                if (w.ce8_view && us == CHD_NONUNIFORM && (!skip_self || conn < c_smin || conn > c_smax))
                    us = CHD_NOT_A_SENDER;
                const uint32_t hand = chans ? c_hand : 0u;
                const bool none = skip_self && us == conn;  // every buffered entity update is this connection's own
                if (OFF && now >= Lw + I) {
                    // the walk against the real arrival stamps; wms = the slots each window covers whole
                    const uint32_t age = ring.cur_tick - c_htick;
                    const uint32_t chh = age < CHD_HIST_BITS ? (c_hist << age) : 0u, chhp = age < CHD_HIST_BITS ? (c_hprev << age) : 0u;
                    plan_windows_off(w, ring, now, Lw, I, c, hand, chh, chs, chhp, chsp, skip_self, conn, w.filt_win + (pbase + p) * CHD_FILT_WINS, op, age,
                                     o_r0, o_r1, o_r2, o_r3, o_q0, o_q1);
                    Lw = op.Lw;
                    nw = op.nw;
#pragma unroll
                    for (int q = 0; q < 4; q++) wms[q] = op.full[q];
                }
                // (a column of up to 512 entries is four 8-byte loads per lane: a descriptor copies at most that.  A larger cell's windows
                // are cut into CHUNKS of 512 entries, one descriptor per (window, chunk) — `big`; not for a first fan-out (its cell
                // record + FULL-tagged entity records would need a second descriptor kind: the deferred launch keeps those) and not on
                // wire worlds, whose layout reads a descriptor as a whole cell column)
                const bool big = size > 512u;
                simple = chans != nullptr && nw <= 4 && (!big || (!(info & SD_FIRST) && !w.rec_pos && size <= 1024u));  // (two chunks: a row of capq descriptors holds them)
                // Which column every non-empty window copies: the cell's full column when every entity has an update inside
                // the window (the AND of their histories intersects it), else — partially updating worlds — the WINDOW COLUMN
                // of exactly that mask (wcol_mask: runs of 1..3 ticks that start at the newest or the one before — what a
                // subscription served every interval sees).  A descriptor carries one column: windows that agree share one
                // descriptor, a subscription whose windows differ gets one descriptor per window (contiguous parts).
                bool same = true;
                if (OFF && !none) {
                    // any window that is not a plain copy (an edge cuts through the arrivals of a tick, some entity skipped an update,
                    // only the spatial channel's own update passes), more than four windows, or a cell beyond the copy kernel's
                    // 512-entry column image: the filtered kernel
                    const bool need = op.need || nw > 4 || (big && !simple && !(info & SD_FIRST));
                    if (need) { filt = chans != nullptr && us != CHD_NONUNIFORM; simple = false; }
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    if (j >= nw || none || (OFF && !simple)) continue;
                    uint32_t cj = 0xFFFFFFFEu;
                    if (us != CHD_NONUNIFORM) {
                        // (per-record masks: the record's mask is its entity's history inside the window — a constant, the
                        // window's own mask, only where every entity has an update at EVERY stamp of the window)
                        if (w.rec_mask ? (hand & wms[j]) == wms[j] : (hand & wms[j]) != 0u) cj = 0u;
                        else if (w.wcol_on) {
#pragma unroll
                            for (uint32_t k = 0; k < CHD_WCOLS; k++)
                                if (wms[j] == wcol_mask(k)) cj = k + 1u;
                        }
                    }
                    if (cj == 0xFFFFFFFEu || (big && cj != 0u)) simple = false;  // (chunks are cut from the full column only)
                    wcolp |= (cj & 15u) << (4u * j);
                    if (j && cj != (wcolp & 15u)) same = false;
                }
                if (OFF && !(info & SD_FIRST) && (op.deep || (!simple && !filt))) {  // (a first fan-out evaluates no window: the filtering launch takes what is not simple)
                    // not decidable from the masks and offsets (plan_windows_off), or a shape the filtered kernel does not take
                    // (several senders of whom this connection may be one, a cell of more than 512 entities): the element buffers
                    deep = true;
                    due = false;
                    simple = filt = false;
                    ub = deep_upper_bound(w, now, L, I, size);
                    w.pair_flags[pbase + p] = fl | PF_DEEP;
                }
                if (simple || filt) {  // (may still turn false below)
                    // exact record counts: the segment is as long as what will be written
                    const uint32_t age = ring.cur_tick - c_htick;
                    const uint32_t chh = age < CHD_HIST_BITS ? (c_hist << age) : 0u;
                    const uint32_t chhp = age < CHD_HIST_BITS ? (c_hprev << age) : 0u;
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++)
                        if (!OFF && j < nw && cell_update_passes(chh, chs, chhp, chsp, wms[j], skip_self, conn)) own |= 1u << j;
                    if (OFF) own = op.own;
                }
                if (filt) {
                    // worst case per window: every entity of the cell (+ the own update); the kernel writes the count
                    ub = (uint64_t)__popc(own) + (uint64_t)nw * size;
                    if (hlost) hist_ovf = 1;
                }
                if (simple) {
                    if (own && w.rec_mask) simple = false;  // (the spatial channel's own record carries its own mask: the filtering launch)
                    wm4 = make_uint4(wms[0], wms[1], wms[2], wms[3]);
                    info |= nw | (none ? SD_NONE : 0u);
                    if (big && nw != 0 && !none) {
                        // one descriptor per (window, chunk of 512 entries)
                        nwin_s = nw;
                        nd = nw * ((size + 511u) >> 9);
                        count = (uint32_t)__popc(own) + nw * size;
                    } else if ((info & SD_FIRST) || nw == 0 || none || same) {
                        // one descriptor: first fan-out (full column, no window), nothing to send, or windows that agree
                        if ((info & SD_FIRST) || nw == 0 || none) wcolp = 0u;
                        const uint32_t col = wcolp & 15u;
                        ncol0 = col ? w.cell_wcnt[(size_t)(col - 1u) * g.ncell + c] : size;
                        nd = 1;
                        count = ((info & SD_FIRST) ? size + 1u : 0u) + (uint32_t)__popc(own) + (none ? 0u : nw * ncol0);
                    } else {
                        // the windows copy different columns: one descriptor per window, contiguous parts of the segment
                        nd = nw;
                        count = (uint32_t)__popc(own);
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++) {
                            if (j >= nw) continue;
                            const uint32_t col = (wcolp >> (4u * j)) & 15u;
                            count += col ? w.cell_wcnt[(size_t)(col - 1u) * g.ncell + c] : size;
                        }
                    }
                    if (hlost) hist_ovf = 1;  // (a deferred subscription is counted by the deferred launch)
                }
            }
        }
        // descriptor slots of this round: exclusive prefix over the lanes; a connection's row holds capq descriptors — a
        // subscription that would not fit goes to the deferred launch instead (never seen: ~20 subscriptions per connection)
        uint32_t dbase;
        const bool multi = w.wcol_on || __ballot(simple && nd > 1u) != 0;  // (uniform) a subscription may take several descriptors
        if (multi) {
            uint32_t dinc = simple ? nd : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)dinc, d);
                if ((int)lane >= d) dinc += o;
            }
            dbase = n_simple + dinc - (simple ? nd : 0u);
        } else {          // one each: the rank among the simple lanes
            dbase = n_simple + mask_rank(__ballot(simple));
        }
        if (simple && dbase + nd > w.capq) simple = false;
        if (due && !simple && !filt) w.pair_flags[pbase + p] = fl | PF_DEFER;
        if (due && simple) ub = count;
        const uint32_t fbase = OFF ? n_filt + mask_rank(__ballot(due && filt)) : 0u;
        // every segment starts on a 128-byte line and is padded to whole lines (k_fanout_plan)
        ub = (ub + (CHD_SEG_ALIGN - 1)) & ~(uint64_t)(CHD_SEG_ALIGN - 1);
        uint64_t inc = ub;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t o = __shfl_up((unsigned long long)inc, d);
            if ((int)lane >= d) inc += o;
        }
        const uint64_t rel = carry + inc - ub;
        const uint32_t rel32 = rel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rel;
        if (p < cnt) {
            w.pair_rel[pbase + p] = rel32;
            w.pair_nrec[pbase + p] = simple ? count : 0u;  // (a deferred subscription's count comes from the deferred launch)
            if (w.pair_desc) w.pair_desc[pbase + p] = (due && simple) ? (dbase | (nd << 28)) : 0xFFFFFFFFu;
        }
        carry += __shfl((unsigned long long)inc, 63);
        if (__ballot(due && !simple && !filt)) any_deferred = 1;
        if (__ballot(deep)) any_deep = 1;
        if (OFF) {
            if (due && filt) {
                const size_t k = pbase + fbase;
                w.filt_desc[k] = make_uint4(rel32, start, size, op.nw | (own << 8));
                // (the subscription's new fan-out state is committed right here — data.go:217-223,271 — and the OLD one kept in the
                // descriptor: the deferred launch puts it back for a connection that turns out to have no room, WorldDev::tail_ctl)
                w.filt_desc2[k] = make_uint4(c, p, fl, s);
                w.filt_ln[k] = Lold;
                w.pair_last[pbase + p] = Lw;
                w.pair_flags[pbase + p] = fl | PF_HAD_FIRST;
                if (w.fcm_on) {  // ... listed under its cell (k_fanout_emit_filt_cm); the order inside a cell's list does not matter
                    const uint32_t at = atomicAdd(&w.cell_fcnt[32u * c], 1u);
                    if (at < w.S) {
                        uint4 *e = w.cell_flist + ((size_t)c * w.S + at) * 2;
                        e[0] = make_uint4(rel32, size, op.nw | (own << 8), p);
                        e[1] = make_uint4(s, 0u, 0u, 0u);
                    }
                }
            }
            n_filt += (uint32_t)__popcll(__ballot(due && filt));
        }
        if (due && simple) {
            if (nwin_s) {
                uint32_t at = rel32, kk = 0;
                const uint32_t nch = (size + 511u) >> 9;
                for (uint32_t j = 0; j < nwin_s; j++) {
                    const uint32_t o = (own >> j) & 1u;
                    for (uint32_t ch = 0; ch < nch; ch++, kk++) {
                        const uint32_t nc = min(512u, size - 512u * ch), oo = ch == 0u ? o : 0u;
                        const size_t k = pbase + dbase + kk;
                        w.seg_desc[k] = make_uint4(at, start + 512u * ch, nc, 1u | (oo << SD_OWN_SHIFT) | (kk + 1u < nd ? SD_NOPAD : 0u));
                        w.seg_desc2[k] = make_uint4(c, p, fl, 0u);
                        w.seg_ln[k] = Lold;
                        if (w.rec_mask) w.seg_wm[k] = make_uint4(j == 0 ? wm4.x : j == 1 ? wm4.y : j == 2 ? wm4.z : wm4.w, 0u, 0u, 0u);
                        at += oo + nc;
                    }
                }
            } else if (nd == 1) {
                const uint32_t col = wcolp & 15u;
                const size_t k = pbase + dbase;
                w.seg_desc[k] = make_uint4(rel32, start + col * w.wcol_stride, ncol0, info | (own << SD_OWN_SHIFT));
                w.seg_desc2[k] = make_uint4(c, p, fl, 0u);  // (.z and seg_ln: the state BEFORE this tick, see the filtered descriptors below)
                w.seg_ln[k] = Lold;
                if (w.rec_mask) w.seg_wm[k] = wm4;
            } else {
                uint32_t at = rel32;
                for (uint32_t j = 0; j < nd; j++) {  // (rare: partially updating worlds, windows with different masks)
                    const uint32_t col = (wcolp >> (4u * j)) & 15u;
                    const uint32_t ncol = col ? w.cell_wcnt[(size_t)(col - 1u) * g.ncell + c] : size;
                    const uint32_t o = (own >> j) & 1u;
                    const size_t k = pbase + dbase + j;
                    w.seg_desc[k] = make_uint4(at, start + col * w.wcol_stride, ncol, 1u | (o << SD_OWN_SHIFT) | (j + 1u < nd ? SD_NOPAD : 0u));
                    w.seg_desc2[k] = make_uint4(c, p, fl, 0u);
                    w.seg_ln[k] = Lold;
                    at += o + ncol;
                }
            }
            w.pair_last[pbase + p] = Lw;
            w.pair_flags[pbase + p] = fl | PF_HAD_FIRST;
            rec_simple += count;
        }
        if (multi) {
            uint32_t tot = (due && simple) ? nd : 0u;
            for (int d = 32; d >= 1; d >>= 1) tot += (uint32_t)__shfl_xor((int)tot, d);
            n_simple += tot;
        } else {
            n_simple += (uint32_t)__popcll(__ballot(due && simple));
        }
    }
    for (int d = 32; d >= 1; d >>= 1) rec_simple += __shfl_xor(rec_simple, d);
    if (__ballot(hist_ovf) && lane == 0) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    if (lane == 0) {
        w.rec_ub[s] = carry > 0xFFFFFFFFull ? (1ull << 40) : carry;
        w.n_simple[s] = n_simple;
        if (OFF) w.n_filt[s] = n_filt;
        w.conn_defer[s] = any_deferred;
        if (w.deep_depth) w.conn_deep[s] = any_deep;
        w.rec_cnt[s] = (uint32_t)rec_simple;
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16];
        if (rec_simple) atomicAdd(slot, rec_simple);
        if (cnt) atomicAdd(slot + 1, (unsigned long long)cnt);
    }
}

// the records of one copy of a cell column held as four adjacent PAIRS per lane (q[h] = entries 128 h + 2 lane, + 1):
// every store instruction writes ONE contiguous 1 KiB run (8 whole 128-byte lines).  tools/ubench/store_pair.hip: the
// older layout — four entries per lane, two stores of 16-byte pieces at a 32-byte stride, i.e. two byte-masked write
// requests per line — reaches 5.0 TB/s, this one 5.4.  `since` counts the wide stores certainly issued (wave-uniform).
__device__ __forceinline__ uint32_t store_column2(const u32x2 (&q)[4], uint32_t n, uint32_t start, uint32_t conn_tag,
                                                  chd_fanout_rec *__restrict__ out, uint32_t *__restrict__ opos, uint32_t n_out,
                                                  uint32_t &since, uint32_t *__restrict__ omask = nullptr, uint32_t wm = 0) {
    const uint32_t lane = lane_id();
    // (opaque copy: left to itself the compiler builds the {tag, channel} register pairs of every store of every
    // caller up front, from the moment the column registers exist — ~25 VGPRs held across the whole segment loop)
    asm volatile("" : "+v"(conn_tag));
#pragma unroll
    for (int h = 0; h < 4; h++) {
        if (n <= (uint32_t)(128 * h)) break;  // uniform
        const uint32_t k = 128u * h + 2 * lane;
#ifdef FO_PF_NOSTORE  // experiment: everything but the record stores
        if (n == 0xFFFFFFFFu)
#endif
        if (k + 1 < n) {
            u32x4 r;
            r.x = conn_tag; r.y = q[h].x; r.z = conn_tag; r.w = q[h].y;
            *(u32x4 *)(void *)(out + n_out + k) = r;
            if (opos) { opos[n_out + k] = start + k; opos[n_out + k + 1] = start + k + 1; }
            if (omask) { u32x2 m2; m2.x = wm; m2.y = wm; *(u32x2 *)(void *)(omask + n_out + k) = m2; }  // (every entity merges the whole window)
        } else if (k < n) {
            chd_fanout_rec r;
            r.conn = conn_tag;
            r.channel = q[h].x;
            out[n_out + k] = r;
            if (opos) opos[n_out + k] = start + k;
            if (omask) omask[n_out + k] = wm;
        }
        if (n >= (uint32_t)(128 * h + 2)) since += 1;  // lane 0 holds a whole pair: the wide store was issued
    }
    return n_out + n;
}

#ifdef FO_PF_TRACE  // diagnosis builds only: per-connection wall-clock marks of the last launch (100 MHz ticks)
__device__ unsigned long long fo_trace[4 * 16384];
extern "C" int chd_debug_trace(unsigned long long *out, unsigned n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_trace), sizeof(unsigned long long) * 4 * (n < 16384 ? n : 16384));
}
#define PF_TRACE(slot) do { if (threadIdx.x == 0 && s < 16384) fo_trace[4 * s + (slot)] = wall_clock64(); } while (0)
#else
#define PF_TRACE(slot) do { } while (0)
#endif

#ifndef FO_SEG_SPEC
#define FO_SEG_SPEC 16u  // descriptors per wave loaded with the ticket's header, before n_simple is known
#endif
#ifndef FO_SEG_BATCH
#define FO_SEG_BATCH 2   // segments whose columns a wave loads together, between two waits
#endif

// A PERSISTENT grid of single-wave workgroups, far fewer than the chip has wave slots (WorldDev::seg_waves): the record
// stream saturates HBM with ~4 waves per SIMD, and the slots left free are what lets the next tick's stages run beside
// this kernel (CHD_WORLD_PIPELINE_TICKS) — a launch of one workgroup per connection takes every slot that frees up until
// its last workgroup is placed, and a kernel on another stream, whatever its priority, starts only then (measured:
// k_index_hist 154 us instead of 8).  Work is handed out by tickets: ticket T = (connection T / WAVES, role T % WAVES),
// the role's segments are k = role, role + WAVES, ...; eight counters on eight lines, a wave takes from the one of its
// XCD (blockIdx & 7), and requests its NEXT ticket before it starts storing, so the atomic's return is awaited together
// with the first column loads and never drains record stores.
// MASKS (CHD_WORLD_UPDATE_MASKS): also the per-record merged-updates mask — on this path a constant per window, the window's
// own mask (the plan took the subscription only if EVERY entity of the cell has an update at EVERY stamp of the window).
__device__ __forceinline__ void filt_items_block(const WorldDev &w, uint32_t ncell);
template <int WAVES, bool MASKS = false>
__global__ void __launch_bounds__(64, FO_SEG_OCC) k_fanout_emit_seg(DevGrid g, WorldDev w, uint32_t n_tickets, uint32_t s_fine, uint32_t fine_sh) {
    constexpr int B = FO_SEG_BATCH;
    static_assert((WAVES & (WAVES - 1)) == 0, "waves per connection: a power of two");
    constexpr uint32_t WSH = WAVES == 1 ? 0u : WAVES == 2 ? 1u : WAVES == 4 ? 2u : 3u;
    // The LAST connections (slots >= s_fine) are cut into 2^fine_sh pieces instead of WAVES: when the tickets run out every
    // wave still finishes the piece it holds, and the chip drains for the length of one piece — shorter pieces, shorter tail
    // (all of them short would pay the per-piece descriptor walk on every connection: profiles/r02_emit_variant_sweep.json).
    const uint32_t t_fine = s_fine << WSH;
    const uint32_t lane = lane_id();
    const uint32_t bank = blockIdx.x & 7u;
    uint32_t *__restrict__ ctr = w.emit_ticket + 32u * bank;
    // The filtered kernel's work items (worlds with sub-tick offsets) are built HERE, by the launch's last workgroup — as a rule one of
    // those that take no ticket —, beside the record stream: nobody needs them before this kernel has ended, and in k_fanout_scan
    // their one-workgroup pass stood 6.5 us in front of every record (14.4 us against 7.9).
    if (blockIdx.x == gridDim.x - 1u && w.off_on && w.fcm_on && !w.late_tot) filt_items_block(w, g.ncell);
    // (this tick's active waves, k_fanout_scan: the launch has emit_waves workgroups, the ones beyond leave at once; an active
    // workgroup's first ticket is its own index — the banks' counters start behind them)
    if (blockIdx.x >= (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[3])) return;
    uint32_t tk = blockIdx.x >> 3;
    for (;;) {
    const uint32_t T = 8u * tk + bank;
    if (T >= n_tickets) break;
    // the next ticket: requested now, read with the ticket's header words.  (atomicInc, not atomicAdd: the compiler's atomic optimizer
    // turns a uniform-address add into a wave reduction + readfirstlane and waits for the value RIGHT HERE — on the in-order vm counter
    // that also drains the previous ticket's record stores; it leaves the wrapping increment alone)
    uint32_t tk_next = 0;
    if (lane == 0) tk_next = atomicInc(ctr, 0xFFFFFFFFu);
    const bool fine = T >= t_fine;
    const uint32_t wsh = fine ? fine_sh : WSH;
    const uint32_t Tr = fine ? T - t_fine : T;
    const uint32_t s = (fine ? s_fine : 0u) + (Tr >> wsh);
    const uint32_t wave = Tr & ((1u << wsh) - 1u);
    PF_TRACE(0);
    const size_t pbase = (size_t)s * w.capq;
    // ONE round trip for everything the ticket needs before its columns: the connection's words, the next ticket and — speculatively, the
    // row exists whatever n_simple says — the wave's first FO_SEG_SPEC descriptors.  As a chain "ticket -> segments? -> room? -> base,
    // connection -> descriptors" these were five dependent trips in front of every ticket's first column load, each behind the previous
    // ticket's draining stores: nothing at config B, where a ticket is ~30 KB of records (profiles/r08b_ab_emit_seg_header.txt), but
    // most of the kernel's time where tickets are short (cells of 44 entities: 8 KB per ticket, profiles/r08i_ab_emit_seg_header_small.txt).
    // (More than 16 speculative descriptors cost bandwidth: 64 per wave = 40 MB of rows nobody reads, +8 us at config B.)
    const uint32_t kl0 = wave + (lane << wsh);
    const bool spec = lane < FO_SEG_SPEC && kl0 < w.capq;
    uint32_t ns_v = w.n_simple[s];
    uint64_t base_v = w.rec_ub[s], end_v = w.rec_ub[s + 1];
    uint32_t conn_v = w.conn_id[s];
    u32x4 dv0 = {0, 0, 0, 0}, wmv0 = {0, 0, 0, 0};
    uint32_t cv0 = 0;
    if (spec) {
        dv0 = *(const u32x4 *)(const void *)(w.seg_desc + pbase + kl0);
        cv0 = w.seg_desc2[pbase + kl0].x;
        if (MASKS) wmv0 = *(const u32x4 *)(const void *)(w.seg_wm + pbase + kl0);
    }
    asm volatile("" : "+v"(ns_v), "+v"(base_v), "+v"(end_v), "+v"(conn_v), "+v"(dv0), "+v"(cv0), "+v"(wmv0), "+v"(tk_next));  // (all of them issued, then awaited together)
    const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)ns_v), conn = (uint32_t)__builtin_amdgcn_readfirstlane((int)conn_v);
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base_v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base_v);
    const uint64_t end = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(end_v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)end_v);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk_next);
    // (end > recs_cap: no room for this connection's worst case — the deferred launch leaves its state as it was and flags the tick)
    if (wave >= ns || end > w.recs_cap) continue;
    const uint32_t *__restrict__ chans = w.ce_chan_view;
    PF_TRACE(1);
    // Every descriptor of this wave in ONE vector load: lane j holds the wave's j-th segment (k = wave + j WAVES); the
    // streaming loop reads them with v_readlane (scalar operands from there on).  No descriptor load ever sits between
    // record stores: the vm counter is in-order, so ANY load wait also waits for every record store issued before it,
    // and a store takes ~3.5 us to complete while the chip streams.
    const uint32_t mine = (ns - wave + (1u << wsh) - 1u) >> wsh;  // segments of this wave (a connection has at most 64 WAVES due ones per pass)
    for (uint32_t j0 = 0; j0 < mine; j0 += 64) {
        const uint32_t kl = wave + ((j0 + lane) << wsh);
        u32x4 dv = {0, 0, 0, 0}, wmv = {0, 0, 0, 0};
        uint32_t cv = 0;
        if (kl < ns) {
            if (j0 == 0 && lane < FO_SEG_SPEC) { dv = dv0; cv = cv0; wmv = wmv0; }
            else {
                dv = *(const u32x4 *)(const void *)(w.seg_desc + pbase + kl);
                cv = w.seg_desc2[pbase + kl].x;
                if (MASKS) wmv = *(const u32x4 *)(const void *)(w.seg_wm + pbase + kl);
            }
        }
        const uint32_t here = min(mine - j0, 64u);
        // ... and the columns of B segments are loaded TOGETHER, one wait, then B segments' records are stored back to
        // back: B times the bytes in flight per wave for the same number of waits (tools/ubench/store_conn2.hip).
        for (uint32_t jb = 0; jb < here; jb += B) {
            u32x2 col[B][4];
#pragma unroll
            for (int b = 0; b < B; b++) {
                // q[h] = entries 128 h + 2 lane, + 1 of the cell: ONE address register, the four quarters are immediate
                // offsets (lanes beyond the cell read the spare entries behind the column or the next cells': never used)
                const uint32_t jj = jb + b < here ? jb + b : jb;
                const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)dv.y, (int)jj);
                const uint32_t *pa = chans + start + 2 * lane;
                asm volatile(
                    "global_load_dwordx2 %0, %4, off\n\t"
                    "global_load_dwordx2 %1, %4, off offset:512\n\t"
                    "global_load_dwordx2 %2, %4, off offset:1024\n\t"
                    "global_load_dwordx2 %3, %4, off offset:1536"
                    : "=&v"(col[b][0]), "=&v"(col[b][1]), "=&v"(col[b][2]), "=&v"(col[b][3])
                    : "v"(pa)
                    : "memory");
            }
#pragma unroll
            for (int b = 0; b < B; b++)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(col[b][0]), "+v"(col[b][1]), "+v"(col[b][2]), "+v"(col[b][3]) : : "memory");
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (jb + b >= here) break;  // uniform
                uint32_t since = 0;
                const int jj = (int)(jb + b);
                const uint32_t rel = (uint32_t)__builtin_amdgcn_readlane((int)dv.x, jj), start = (uint32_t)__builtin_amdgcn_readlane((int)dv.y, jj);
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)dv.z, jj), info = (uint32_t)__builtin_amdgcn_readlane((int)dv.w, jj);
                const uint32_t cch = (uint32_t)__builtin_amdgcn_readlane((int)cv, jj) + g.id_start;
                chd_fanout_rec *__restrict__ out = w.recs + base + rel;
                uint32_t *__restrict__ opos = (w.rec_pos && !w.seg_no_pos) ? w.rec_pos + base + rel : nullptr;
                uint32_t *__restrict__ omask = MASKS ? w.rec_mask + base + rel : nullptr;
                uint32_t wms[4] = {0, 0, 0, 0};
                if (MASKS) {
                    wms[0] = (uint32_t)__builtin_amdgcn_readlane((int)wmv.x, jj); wms[1] = (uint32_t)__builtin_amdgcn_readlane((int)wmv.y, jj);
                    wms[2] = (uint32_t)__builtin_amdgcn_readlane((int)wmv.z, jj); wms[3] = (uint32_t)__builtin_amdgcn_readlane((int)wmv.w, jj);
                }
                uint32_t n_out = 0;
                if (info & SD_FIRST) {
                    // first fan-out: the whole data of the spatial channel and of every entity channel in it
                    if (lane == 0) {
                        chd_fanout_rec r;
                        r.conn = conn | CHD_REC_FULL;
                        r.channel = cch;
                        out[0] = r;
                        if (opos) opos[0] = CHD_POS_CELL | (cch - g.id_start);
                        if (MASKS) omask[0] = 0;
                    }
                    n_out = store_column2(col[b], n, start, conn | CHD_REC_FULL, out, opos, 1u, since, omask, 0u);
                }
                const uint32_t nw = info & SD_NWIN_MASK;
                for (uint32_t j = 0; j < nw; j++) {
                    if ((info >> (SD_OWN_SHIFT + j)) & 1u) {  // the spatial channel's own buffered updates
                        if (lane == 0) {
                            chd_fanout_rec r;
                            r.conn = conn;
                            r.channel = cch;
                            out[n_out] = r;
                            if (opos) opos[n_out] = CHD_POS_CELL | (cch - g.id_start);
                        }
                        n_out += 1;
                    }
                    if (!(info & SD_NONE)) {  // every entity passes this window (that is what made the subscription simple)
                        n_out = store_column2(col[b], n, start, conn, out, opos, n_out, since, omask, j == 0 ? wms[0] : j == 1 ? wms[1] : j == 2 ? wms[2] : wms[3]);
                    }
                }
                // (a later part of a split subscription starts anywhere inside a line: pad to the line of the segment's END)
                if (!(info & SD_NOPAD)) {
                    const uint32_t pad = (0u - (rel + n_out)) & (CHD_SEG_ALIGN - 1u);
                    if (lane < pad) {
                        chd_fanout_rec r;
                        r.conn = 0xFFFFFFFFu;
                        r.channel = 0;
                        out[n_out + lane] = r;
                    }
                }
            }
        }
    }
    PF_TRACE(2);
#ifdef FO_PF_TRACE
    if (threadIdx.x == 0 && s < 16384) fo_trace[4 * s + 3] = w.rec_cnt[s];
#endif
    }
}

// ---------------------------------------------------------------------------
// The FILTERED descriptors (WorldDev::off_on; k_fanout_plan_seg<true>): subscriptions with a window that needs a per-entity
// decision — a window edge cuts through the arrival stamps of a tick (the reference stamps updates when they are enqueued,
// channel.go:296-310, so every window of a subscription whose phase is off the tick grid does), or some entity of the cell has
// no update inside it.  Per descriptor the wave loads the cell's {channel, history} entries once (adjacent pairs: one 16-byte
// load per lane and 128 entries) and, per window, the offset columns of the one or two ring slots the edges cut (8 bytes per
// lane and 128 entries each); an entity passes when an update of a slot the window covers whole is buffered, or one of a cut
// slot with its offset inside the bounds (FiltWin) — the comparison of data.go:236-241 on exact nanosecond stamps.  Passing
// entries are compacted by ballot / mbcnt in entry order and stored as {conn, channel} records; rows where everything passes
// take one 16-byte store per lane.  Persistent single-wave workgroups, tickets as k_fanout_emit_seg (their own counters).
// ---------------------------------------------------------------------------
#ifndef FO_FILT_WAVES
#define FO_FILT_WAVES 2
#endif
#ifndef FO_FILT_OCC
#define FO_FILT_OCC 4      // waves per SIMD the register allocator is asked for
#endif
#ifndef FO_FILT_PER_CU
#define FO_FILT_PER_CU 16  // persistent waves per CU: the kernel is latency-bound per descriptor (dependent loads), not store-bound
#endif

// the offset columns of one or two ring slots for one 512-entry chunk (row h = entries 128 h + 2 lane, + 1), optionally together
// with the chunk's {channel, history} entries — ONE block with its wait inside: the allocator may copy an asm output before a
// later wait, and the hardware does not interlock a v_mov on an outstanding load
__device__ __forceinline__ void filt_load(const uint2 *pe, const uint32_t *pa, const uint32_t *pb, bool with_e, bool use_a, bool use_b,
                                          u32x4 (&e)[4], u32x2 (&oa)[4], u32x2 (&ob)[4]) {
    if (with_e) {
        if (use_b) {
            asm volatile(
                "global_load_dwordx4 %0, %12, off\n\t"
                "global_load_dwordx4 %1, %12, off offset:1024\n\t"
                "global_load_dwordx4 %2, %12, off offset:2048\n\t"
                "global_load_dwordx4 %3, %12, off offset:3072\n\t"
                "global_load_dwordx2 %4, %13, off\n\t"
                "global_load_dwordx2 %5, %13, off offset:512\n\t"
                "global_load_dwordx2 %6, %13, off offset:1024\n\t"
                "global_load_dwordx2 %7, %13, off offset:1536\n\t"
                "global_load_dwordx2 %8, %14, off\n\t"
                "global_load_dwordx2 %9, %14, off offset:512\n\t"
                "global_load_dwordx2 %10, %14, off offset:1024\n\t"
                "global_load_dwordx2 %11, %14, off offset:1536\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3]),
                  "=&v"(ob[0]), "=&v"(ob[1]), "=&v"(ob[2]), "=&v"(ob[3])
                : "v"(pe), "v"(pa), "v"(pb)
                : "memory");
        } else if (use_a) {
            asm volatile(
                "global_load_dwordx4 %0, %8, off\n\t"
                "global_load_dwordx4 %1, %8, off offset:1024\n\t"
                "global_load_dwordx4 %2, %8, off offset:2048\n\t"
                "global_load_dwordx4 %3, %8, off offset:3072\n\t"
                "global_load_dwordx2 %4, %9, off\n\t"
                "global_load_dwordx2 %5, %9, off offset:512\n\t"
                "global_load_dwordx2 %6, %9, off offset:1024\n\t"
                "global_load_dwordx2 %7, %9, off offset:1536\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3])
                : "v"(pe), "v"(pa)
                : "memory");
        } else {
            asm volatile(
                "global_load_dwordx4 %0, %4, off\n\t"
                "global_load_dwordx4 %1, %4, off offset:1024\n\t"
                "global_load_dwordx4 %2, %4, off offset:2048\n\t"
                "global_load_dwordx4 %3, %4, off offset:3072\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
                : "v"(pe)
                : "memory");
        }
    } else if (use_b) {
        asm volatile(
            "global_load_dwordx2 %0, %8, off\n\t"
            "global_load_dwordx2 %1, %8, off offset:512\n\t"
            "global_load_dwordx2 %2, %8, off offset:1024\n\t"
            "global_load_dwordx2 %3, %8, off offset:1536\n\t"
            "global_load_dwordx2 %4, %9, off\n\t"
            "global_load_dwordx2 %5, %9, off offset:512\n\t"
            "global_load_dwordx2 %6, %9, off offset:1024\n\t"
            "global_load_dwordx2 %7, %9, off offset:1536\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3]), "=&v"(ob[0]), "=&v"(ob[1]), "=&v"(ob[2]), "=&v"(ob[3])
            : "v"(pa), "v"(pb)
            : "memory");
    } else if (use_a) {
        asm volatile(
            "global_load_dwordx2 %0, %4, off\n\t"
            "global_load_dwordx2 %1, %4, off offset:512\n\t"
            "global_load_dwordx2 %2, %4, off offset:1024\n\t"
            "global_load_dwordx2 %3, %4, off offset:1536\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3])
            : "v"(pa)
            : "memory");
    }
}

#define FO_FILT_BATCH 32  // descriptors whose windows a wave stages in LDS at a time
template <int WAVES>
__global__ void __launch_bounds__(64, FO_FILT_OCC) k_fanout_emit_filt(DevGrid g, WorldDev w, uint32_t n_tickets) {
    static_assert((WAVES & (WAVES - 1)) == 0, "waves per connection: a power of two");
    constexpr uint32_t WSH = WAVES == 1 ? 0u : WAVES == 2 ? 1u : WAVES == 4 ? 2u : 3u;
    // the windows' tests of the descriptors in flight: read back wave-uniformly (LDS counts on lgkmcnt — a wait for it does not
    // drain the record stores, as a wait for a global load would: the vm counter is in-order)
    __shared__ uint32_t fw_lds[FO_FILT_BATCH][CHD_FILT_WINS][6];
    const uint32_t lane = lane_id();
    const uint32_t bank = blockIdx.x & 7u;
    uint32_t *__restrict__ ctr = w.emit_ticket + 32u * bank + 1u;
    const uint2 *__restrict__ ce8 = w.ce8_view;
    const uint32_t *__restrict__ offs = w.ce_off;
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(ctr, 1u);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    for (;;) {
        const uint32_t T = 8u * tk + bank;
        if (T >= n_tickets) break;
        uint32_t tk_next = 0;
        if (lane == 0) tk_next = atomicAdd(ctr, 1u);
        const uint32_t s = T >> WSH, role = T & (WAVES - 1u);
        const size_t pbase = (size_t)s * w.capq;
        // the connection's header words and (speculatively: the row exists whatever n_filt says) the first batch of descriptors
        // in ONE round trip
        const uint32_t kl0 = role + (lane << WSH);
        u32x4 dv0 = {0, 0, 0, 0};
        u32x2 cv0 = {0, 0};
        if (lane < FO_FILT_BATCH && kl0 < w.capq) {
            dv0 = *(const u32x4 *)(const void *)(w.filt_desc + pbase + kl0);
            cv0 = *(const u32x2 *)(const void *)(w.filt_desc2 + pbase + kl0);
        }
        const uint32_t nf = w.n_filt[s];
        const uint64_t base = w.rec_ub[s], end = w.rec_ub[s + 1];
        const uint32_t conn = w.conn_id[s];
        tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk_next);
        // (end > recs_cap: no room for this connection's worst case — the deferred launch leaves its state as it was and flags the tick)
        if ((role >= nf) | (end > w.recs_cap)) continue;
        uint32_t total = 0;
        // the descriptors of this wave in ONE round of vector loads — lane q holds the wave's q-th descriptor, read with v_readlane
        // below — and their windows' tests into LDS in a second one
        const uint32_t mine = (nf - role + WAVES - 1u) >> WSH;
        for (uint32_t j0 = 0; j0 < mine; j0 += FO_FILT_BATCH) {
            const uint32_t kl = role + ((j0 + lane) << WSH);
            u32x4 dv = dv0;
            u32x2 cv = cv0;
            if (lane < FO_FILT_BATCH && kl < nf) {
                if (j0) {
                    dv = *(const u32x4 *)(const void *)(w.filt_desc + pbase + kl);
                    cv = *(const u32x2 *)(const void *)(w.filt_desc2 + pbase + kl);
                }
                const uint32_t lnw = dv.w & 15u;
                const u32x4 *fwp = (const u32x4 *)(const void *)(w.filt_win + (pbase + cv.y) * CHD_FILT_WINS);  // (192-byte rows: 16-byte aligned)
                // all of the descriptor's windows in one round of loads (two windows = three 16-byte words), then into LDS
                u32x4 t[12];
#pragma unroll
                for (int q = 0; q < 12; q++) t[q] = u32x4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 12; q++)
                    if ((uint32_t)(2 * (q / 3)) < lnw) t[q] = fwp[q];
                u32x4 *dst = (u32x4 *)(void *)&fw_lds[lane][0][0];
#pragma unroll
                for (int q = 0; q < 12; q++)
                    if ((uint32_t)(2 * (q / 3)) < lnw) dst[q] = t[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const uint32_t here = min(mine - j0, (uint32_t)FO_FILT_BATCH);
            for (uint32_t jj = 0; jj < here; jj++) {
                const int jl = (int)jj;
                const uint32_t rel = (uint32_t)__builtin_amdgcn_readlane((int)dv.x, jl), start = (uint32_t)__builtin_amdgcn_readlane((int)dv.y, jl);
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)dv.z, jl), info = (uint32_t)__builtin_amdgcn_readlane((int)dv.w, jl);
                const uint32_t cch = (uint32_t)__builtin_amdgcn_readlane((int)cv.x, jl) + g.id_start;
                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)cv.y, jl);
                const uint32_t nw = info & 15u, own = (info >> 8) & 0xFFu;
                chd_fanout_rec *__restrict__ out = w.recs + base + rel;
                u32x4 e[4];
                u32x2 oa[4], ob[4];
#pragma unroll
                for (int h = 0; h < 4; h++) { e[h] = u32x4{0, 0, 0, 0}; oa[h] = u32x2{0, 0}; ob[h] = u32x2{0, 0}; }
                uint32_t n_out = 0;
                for (uint32_t j = 0; j < nw; j++) {
                    const uint32_t *fw = fw_lds[jj][j];
                    const uint32_t full = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[0]), slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[1]);
                    const uint32_t a_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[2]), a_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[3]);
                    const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[4]), b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)fw[5]);
                    const uint32_t sa = slots & 0xFFu, sb = (slots >> 8) & 0xFFu;
                    const bool use_a = a_lo <= a_hi, use_b = b_lo <= b_hi;  // (use_b only with use_a)
                    const uint32_t bit_a = use_a ? 1u << sa : 0u, bit_b = use_b ? 1u << sb : 0u;
                    if ((own >> j) & 1u) {  // the spatial channel's own buffered update lies inside this window
                        if (lane == 0) {
                            chd_fanout_rec r;
                            r.conn = conn;
                            r.channel = cch;
                            out[n_out] = r;
                        }
                        n_out += 1;
                    }
                    // 512 entries per step; a cell of up to 512 keeps its entries in registers from the first window on
                    for (uint32_t c0 = 0; c0 < n; c0 += 512) {
                        const bool with_e = j == 0 || n > 512;
                        const uint32_t at0 = start + c0 + 2 * lane;
                        filt_load(ce8 + at0, offs + (size_t)sa * w.off_stride + at0, offs + (size_t)sb * w.off_stride + at0, with_e, use_a, use_b, e, oa, ob);
                        const uint32_t nc = n - c0;  // entries of this step (row h: 128 h ...)
#pragma unroll
                        for (int h = 0; h < 4; h++) {
                            if (nc <= (uint32_t)(128 * h)) break;  // uniform
                            const uint32_t q = 128u * h + 2 * lane;
                            const bool in0 = q < nc, in1 = q + 1 < nc;
                            const uint32_t h0 = e[h].y, h1 = e[h].w;
                            // (no short-circuit evaluation: the compiler turns && / || chains into exec-masked branches)
                            const bool pass0 = in0 & (((h0 & full) != 0) | (((h0 & bit_a) != 0) & (oa[h].x >= a_lo) & (oa[h].x <= a_hi)) |
                                                      (((h0 & bit_b) != 0) & (ob[h].x >= b_lo) & (ob[h].x <= b_hi)));
                            const bool pass1 = in1 & (((h1 & full) != 0) | (((h1 & bit_a) != 0) & (oa[h].y >= a_lo) & (oa[h].y <= a_hi)) |
                                                      (((h1 & bit_b) != 0) & (ob[h].y >= b_lo) & (ob[h].y <= b_hi)));
                            const uint64_t m0 = __ballot(pass0), m1 = __ballot(pass1);
                            if ((m0 & m1) == ~0ull) {
                                u32x4 r;
                                r.x = conn; r.y = e[h].x; r.z = conn; r.w = e[h].z;
                                typedef u32x4 __attribute__((aligned(8))) u32x4_a8;  // (n_out is any record index)
                                *(u32x4_a8 *)(void *)(out + n_out + 2 * lane) = r;
                                n_out += 128;
                            } else {
                                // entry order: records before this lane's pair = passing entries of lower lanes
                                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
                                if (pass0) {
                                    chd_fanout_rec r;
                                    r.conn = conn;
                                    r.channel = e[h].x;
                                    out[at] = r;
                                }
                                if (pass1) {
                                    chd_fanout_rec r;
                                    r.conn = conn;
                                    r.channel = e[h].z;
                                    out[at + (pass0 ? 1u : 0u)] = r;
                                }
                                n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
                            }
                        }
                    }
                }
                pad_segment(out, n_out);
                if (lane == 0) w.pair_nrec[pbase + p] = n_out;
                total += n_out;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // (the next batch overwrites the LDS windows)
        }
        if (lane == 0 && total) {
            atomicAdd(&w.rec_cnt[s], total);
            unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16];
            atomicAdd(slot, (unsigned long long)total);
            atomicAdd(slot + 2, (unsigned long long)total);  // (not written by the dominant emit kernel)
            atomicAdd(slot + 4, (unsigned long long)total);  // (chd_tick_stats.n_filtered_records)
        }
    }
}

// ---------------------------------------------------------------------------
// The filtered descriptors, CELL-MAJOR and wave-specialised (WorldDev::fcm_on; the form that runs by default).
//
// What bounded the connection-major kernel above: every (descriptor, window) is a dependent round trip to the cell's columns
// before its stores, and on gfx950's in-order vm counter each such wait also drains the wave's own record stores — ~5 us of
// wave time per window for ~1.5 KB of records.  But the columns a window reads are its CELL's, and a cell has hundreds of filtered
// descriptors per tick (every subscriber of the cell whose phase is off the tick grid).  So, as the cell-major record kernel
// does for the window masks: work item = (cell, 64 of its filtered descriptors); in a workgroup of FC_WAVES waves wave 0 is the LOADER
// — it gathers the item's descriptor headers (segment base, connection, the windows' tests) and stages the cell's columns
// {channel, history, the offsets of all CHD_OFF_SLOTS ring slots} in LDS, double-buffered — and the others are STREAMERS: they take
// descriptors by an LDS ticket and do the per-entity compare of every window on LDS data only, ballot / mbcnt compaction, record
// stores.  A streamer never issues a global load (rare exceptions: a descriptor with more than FC_LWIN windows, a cell beyond
// the 512-entry tile), so it never waits on the vm counter and its stores stay in flight back to back.
// Same records, same segment layout as k_fanout_emit_filt (the tests run both: CHD_FILT_CELL_MAJOR=0).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier();
// one window of one descriptor over a cell's columns in GLOBAL memory (raw offsets: tested together with the history bits), in
// steps of 512 entries: what k_fanout_emit_filt does per window; the cell-major kernel's path for cells beyond its LDS tile
__device__ __forceinline__ uint32_t filt_window_global(const WorldDev &w, uint32_t start, uint32_t n, uint32_t full, uint32_t sa, uint32_t sb,
                                                       uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi, uint32_t conn,
                                                       chd_fanout_rec *__restrict__ out, uint32_t n_out) {
    // (a RARE path of the cell-major kernel — cells beyond its tile, cuts through ring slots the tile does not hold —: one row of 128
    // entries per trip, so that it holds a handful of registers where the unrolled form held 32 and set the whole kernel's count)
    const uint32_t lane = lane_id();
    const bool use_a = a_lo <= a_hi, use_b = b_lo <= b_hi;
    const uint32_t bit_a = use_a ? 1u << sa : 0u, bit_b = use_b ? 1u << sb : 0u;
    const uint32_t *pa = w.ce_off + (size_t)(use_a ? sa : 0u) * w.off_stride + start, *pb = w.ce_off + (size_t)(use_b ? sb : 0u) * w.off_stride + start;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < n; c0 += 128) {
        const uint32_t q = c0 + 2 * lane;
        const bool in0 = q < n, in1 = q + 1 < n;
        // (the pair may straddle the cell's end: its second entry then belongs to the next cell or the arrays' spare entries, and is dropped)
        const u32x4 e = *(const u32x4 *)(const void *)(w.ce8_view + start + q);
        const u32x2 oa = *(const u32x2 *)(const void *)(pa + q), ob = *(const u32x2 *)(const void *)(pb + q);
        const uint32_t h0 = e.y, h1 = e.w;
        const bool pass0 = in0 & (((h0 & full) != 0) | (((h0 & bit_a) != 0) & (oa.x >= a_lo) & (oa.x <= a_hi)) |
                                  (((h0 & bit_b) != 0) & (ob.x >= b_lo) & (ob.x <= b_hi)));
        const bool pass1 = in1 & (((h1 & full) != 0) | (((h1 & bit_a) != 0) & (oa.y >= a_lo) & (oa.y <= a_hi)) |
                                  (((h1 & bit_b) != 0) & (ob.y >= b_lo) & (ob.y <= b_hi)));
        const uint64_t m0 = __ballot(pass0), m1 = __ballot(pass1);
        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                            __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
        if (pass0) {
            chd_fanout_rec r;
            r.conn = conn;
            r.channel = e.x;
            out[at] = r;
        }
        if (pass1) {
            chd_fanout_rec r;
            r.conn = conn;
            r.channel = e.z;
            out[at + (pass0 ? 1u : 0u)] = r;
        }
        n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
    }
    return n_out;
}

#ifndef FC_DESCS
#define FC_DESCS 64  // (>= 16: filt_items' room)
#endif
// 12 waves (1 loader + 11 streamers), two workgroups per CU: measured on config B with jittered stamps against 8 waves (14
// streamers per CU: emit stage 228 us) and 16 (30 per CU, registers spilled: 318 us) — 216 us.  Round 6, with the two rare paths
// rolled (60 registers, nothing spilled: every shape below fits): 16 waves x 2 workgroups (30 streamers per CU) and 8 waves x 4
// workgroups on a tile of three offset slots (FC_TSLOTS = 3, 38 KB: 28 streamers, four loaders per CU) measure what 12 x 2 does,
// 10 x 3 and 14 x 2 more (profiles/r09p_*, r09q_*): the kernel does not respond to the number of waves either.
#ifndef FC_WAVES
#define FC_WAVES 12
#endif
#ifndef FC_OCC
#define FC_OCC 6
#endif
// ring slots whose sub-tick offsets the LDS tile holds (slot 0 = this tick's arrivals).  A window that cuts through a slot beyond them
// takes the global-memory path (filt_window_global): windows of the shipped intervals reach three slots at most.
#ifndef FC_TSLOTS
#define FC_TSLOTS CHD_OFF_SLOTS
#endif
#ifndef FC_WG_PER_CU
#define FC_WG_PER_CU 2u
#endif
#ifndef FC_ITEMS_DEFAULT
#define FC_ITEMS_DEFAULT 512u
#endif
#define FC_LWIN 4  // windows per descriptor whose tests are held in LDS (the rest, rare, are read from global memory)
#define FC_NORUN 0xFFFFFFFFu
struct FcHead {
    uint32_t out16[FC_DESCS];  // segment start in the record buffer, in units of 16 records (128-byte lines)
    uint32_t n[FC_DESCS], info[FC_DESCS];  // entries of the cell; windows | own-update bits << 8 (0xFFFFFFFF: skip — no room for the connection's worst case)
    uint32_t conn[FC_DESCS], pidx[FC_DESCS], sidx[FC_DESCS];
    uint32_t win[FC_DESCS][FC_LWIN][6];
    // sorted cells (k_cell_arrange): the part of window j that lies inside THIS tick's arrivals (a cut through ring slot 0) as the run
    // [i0, i1) of the cell's column, i0 | i1 << 16; FC_NORUN: none (the per-entity compare decides the whole window)
    uint32_t run[FC_DESCS][FC_LWIN];
    uint32_t nd, cch, start, valid, ticket, sorted, _pad[2];  // sorted: the cell's entries are in the order of this tick's arrival offsets
};
struct FcTile {
    uint32_t chan[512], hist[512], off[FC_TSLOTS][512];
};

// work items: per cell with filtered descriptors, chunks of FC_DESCS of its list (one workgroup; ncell <= 4096)
// k_fanout_emit_filt_cm's grid: two workgroups per CU (LDS: 56 KB each), no more than there can be items
__host__ __device__ __forceinline__ uint32_t fc_grid(const WorldDev &w, uint32_t ncell) {
    const uint64_t max_items = (uint64_t)w.S * w.capq / 16u + ncell, cap = (uint64_t)(w.seg_waves / 8u) * FC_WG_PER_CU;
    return (uint32_t)(max_items < cap ? max_items : cap);
}

__device__ __forceinline__ void filt_items_block(const WorldDev &w, uint32_t ncell) {
    __shared__ uint32_t carry_s, total_s;
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) { carry_s = 0; total_s = 0; }
    __syncthreads();
    // How many descriptors a work item takes.  The kernel runs ~512 workgroups that draw items by ticket and handle 1-3 each: with items
    // of up to FC_DESCS = 64 descriptors cut off the front of each cell's list (64, 64, 12 ...) the launch lasted as long as the
    // workgroups that drew two full ones (profiles/r07l_filt_prof.json: the average workgroup was busy for half of it).  So: about
    // filt_target items per launch (two per workgroup), a cell's list cut into EQUAL parts of at most that size, never below 16
    // descriptors (every item stages its cell's tile: 20 KB).
    {
        uint32_t t = 0;
        for (uint32_t c = threadIdx.x; c < ncell; c += blockDim.x) t += min(w.cell_fcnt[32u * c], w.S);
        for (int d = 32; d >= 1; d >>= 1) t += (uint32_t)__shfl_xor((int)t, d);
        if (lane == 0 && t) atomicAdd(&total_s, t);
    }
    __syncthreads();
    const uint32_t target = (w.filt_target & 0x7FFFFFFFu) ? (w.filt_target & 0x7FFFFFFFu) : FC_ITEMS_DEFAULT;
    const bool lpt = !(w.filt_target >> 31);  // (A/B runs: bit 31 of CHD_FILT_ITEMS_TARGET = the items in no particular order)
    const uint32_t D = min(max((total_s + target - 1u) / target, 16u), (uint32_t)FC_DESCS);
    // The items in order of DESCENDING size (a counting sort over the 64 possible sizes): the workgroups draw them in that order —
    // the first gridDim.x statically —, so the ones that start with a large item draw fewer afterwards and the launch ends when the
    // SMALL items run out (longest-processing-time-first; in cell order the launch lasted 73 us while its average workgroup was
    // busy for 54).  A cell's list is cut into `chunks` parts of floor or ceil(cnt / chunks) descriptors.
    __shared__ uint32_t bucket[FC_DESCS + 2];
    for (uint32_t k = threadIdx.x; k < FC_DESCS + 2u; k += blockDim.x) bucket[k] = 0;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < ncell; c += blockDim.x) {
        const uint32_t cnt = min(w.cell_fcnt[32u * c], w.S);
        if (!cnt) continue;
        const uint32_t chunks = (cnt + D - 1u) / D, lo = cnt / chunks, n_hi = cnt - lo * chunks;  // n_hi parts of lo + 1, the rest of lo
        if (n_hi) atomicAdd(&bucket[lpt ? lo + 1u : 0u], n_hi);
        atomicAdd(&bucket[lpt ? lo : 0u], chunks - n_hi);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix from the largest size down; bucket[k] becomes the first position of size k
        uint32_t run = 0;
        for (int k = (int)FC_DESCS + 1; k >= 0; k--) { const uint32_t n = bucket[k]; bucket[k] = run; run += n; }
        carry_s = run;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < ncell; c += blockDim.x) {
        const uint32_t cnt = min(w.cell_fcnt[32u * c], w.S);
        if (!cnt) continue;
        const uint32_t chunks = (cnt + D - 1u) / D;
        const uint32_t start = w.cell_start[c], tn = min(w.cell_end[c] - start, 512u);
        for (uint32_t q = 0; q < chunks; q++) {
            const uint32_t a = (uint32_t)(((uint64_t)cnt * q) / chunks), b = (uint32_t)(((uint64_t)cnt * (q + 1u)) / chunks);  // (b - a <= D <= FC_DESCS)
            const uint32_t at = atomicAdd(&bucket[lpt ? b - a : 0u], 1u);
            w.filt_items[at] = make_uint4(c, a, (b - a) | (tn << 8), start);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *w.filt_nitems = carry_s;
        w.filt_nitems[16] = fc_grid(w, ncell);  // k_fanout_emit_filt_cm's item ticket: every workgroup's first item is its own index
    }
}

// The single-workgroup pass between the plan and the record kernels: (1) the exclusive scan of the connections' record ranges
// (rec_ub, in place; rec_ub[S] = total), (2) the tick's tail lists (WorldDev::defer_list / deep_list / tail_ctl: the connections
// the deferred and the element-walk launches have work for, in slot order, and the first connection without room), (3) the
// cell-major filtered kernel's work items (filt_items_block).  One launch where there were two (k_scan_excl, k_filt_items), and what
// lets the two tail launches run over a few workgroups instead of one per connection slot.
// inclusive scan over the wave by DPP row shifts + row broadcasts (gfx9: ~8 cycles a step; __shfl_up is an LDS-path bpermute,
// ~100 cycles a step, and this kernel is one chain of such steps)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, BOUND);
}
// inclusive scan inside every row of 16 lanes
__device__ __forceinline__ uint32_t row_incl_scan_dpp(uint32_t v) {
    v = dpp_add<0x111, 0xF, true>(v);   // row_shr:1 (lanes shifted in from outside the row read 0)
    v = dpp_add<0x112, 0xF, true>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF, true>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF, true>(v);   // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v = row_incl_scan_dpp(v);
    v = dpp_add<0x142, 0xA, false>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC, false>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// ... of 64-bit values below 2^44 per lane, as two 32-bit scans (low 20 bits / the rest)
__device__ __forceinline__ uint64_t wave_incl_scan_dpp64(uint64_t v) {
    const uint32_t lo = wave_incl_scan_dpp((uint32_t)v & 0xFFFFFu), hi = wave_incl_scan_dpp((uint32_t)(v >> 20));
    return ((uint64_t)hi << 20) + lo;
}

__global__ void __launch_bounds__(1024) k_fanout_scan(WorldDev w, uint32_t ncell, int seg) {
    __shared__ uint64_t wtot[2][16];
    __shared__ uint32_t ctot[2][16];
    __shared__ uint32_t scap_s;
    const uint32_t lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = w.S;
    const uint64_t cap = w.recs_cap;
    if (threadIdx.x == 0) scap_s = n;
    uint64_t carry = 0;             // (kept by every thread: the totals of the tiles before this one)
    uint32_t dcarry = 0, pcarry = 0;
    // FOUR tiles of 4096 connections are loaded together, then scanned from registers one after the other: the kernel is a chain of
    // dependent round trips to words another kernel has just written (nothing of it is in this CU's caches)
    constexpr int T = 4;
    uint32_t tile = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += T * 4096u) {
        uint64_t v[T][4];
        uint32_t fdp[T];  // bits 0..3: connection i0 + k has deferred subscriptions; bits 4..7: PF_DEEP ones
        {
            // Wide loads, all of them unconditional and issued before anything is used: a thread's four connections are 32 + 16 + 16
            // contiguous bytes (the arrays carry four spare elements: no tail case).  This workgroup owns ONE CU's memory pipeline, and
            // 48 narrow loads per thread (16 waves) kept it busy for 5 us.  A load behind a per-lane condition is awaited before the
            // next one is issued, and so is a value merged behind a uniform branch: what is not wanted is read from a zero word
            // (tail_ctl[8..11]) by every lane.
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 va[T], vb[T];
            u32x4 cd[T], cp[T];
            const uint32_t *pd = seg ? w.conn_defer : w.tail_ctl + 8, *pp = w.deep_depth ? w.conn_deep : w.tail_ctl + 8;
            const uint32_t md = seg ? 0xFFFFFFFFu : 0u, mp = w.deep_depth ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const uint32_t i0 = g0 + (uint32_t)t * 4096u + threadIdx.x * 4u;
                const uint32_t ix = i0 < n ? i0 : 0u;
                va[t] = *(const u64x2 *)(const void *)(w.rec_ub + ix);
                vb[t] = *(const u64x2 *)(const void *)(w.rec_ub + ix + 2);
                cd[t] = *(const u32x4 *)(const void *)(pd + (ix & md));
                cp[t] = *(const u32x4 *)(const void *)(pp + (ix & mp));
            }
#pragma unroll
            for (int t = 0; t < T; t++) {
                const uint32_t i0 = g0 + (uint32_t)t * 4096u + threadIdx.x * 4u;
                const uint64_t vv[4] = {va[t].x, va[t].y, vb[t].x, vb[t].y};
                const uint32_t dd[4] = {cd[t].x, cd[t].y, cd[t].z, cd[t].w}, pq[4] = {cp[t].x, cp[t].y, cp[t].z, cp[t].w};
                fdp[t] = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool in = i0 + k < n;
                    v[t][k] = in ? (vv[k] < (1ull << 40) ? vv[k] : (1ull << 40)) : 0ull;  // (the plan's "does not fit 32-bit offsets" mark, at most)
                    fdp[t] |= (in && dd[k]) ? 1u << k : 0u;
                    fdp[t] |= (in && pq[k]) ? 16u << k : 0u;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; t++, tile++) {
            if (g0 + (uint32_t)t * 4096u >= n) break;  // uniform
            const uint32_t i0 = g0 + (uint32_t)t * 4096u + threadIdx.x * 4u;
            const uint32_t par = tile & 1u;
            const uint64_t sum = v[t][0] + v[t][1] + v[t][2] + v[t][3];  // (a connection's range is at most 2^40: < 2^44 per lane)
            const uint64_t inc = wave_incl_scan_dpp64(sum);
            const uint32_t cown = (uint32_t)__popc(fdp[t] & 15u) | ((uint32_t)__popc(fdp[t] >> 4) << 16);  // (4096 per tile: both counts fit 16 bits)
            const uint32_t cinc = wave_incl_scan_dpp(cown);
            if (lane == 63) { wtot[par][wave] = inc; ctot[par][wave] = cinc; }
            __syncthreads();  // (the only barrier of a tile: the totals' buffers alternate)
            // the 16 waves' totals, scanned by lanes 0..15 of every wave
            uint64_t wt = wtot[par][lane & 15u];
            wt = wt < (1ull << 44) ? wt : (1ull << 44);  // (beyond every capacity: saturate, the scans below are 32-bit)
            const uint32_t ct = ctot[par][lane & 15u];
            const uint32_t wlo = row_incl_scan_dpp((uint32_t)wt & 0xFFFFFu), whi = row_incl_scan_dpp((uint32_t)(wt >> 20)), wc = row_incl_scan_dpp(ct);
            const uint32_t blo = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wlo, wave - 1) : 0u, bhi = wave ? (uint32_t)__builtin_amdgcn_readlane((int)whi, wave - 1) : 0u;
            const uint32_t bc = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wc, wave - 1) : 0u;
            const uint64_t ttot = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)whi, 15) << 20) + (uint32_t)__builtin_amdgcn_readlane((int)wlo, 15);
            const uint32_t tc = (uint32_t)__builtin_amdgcn_readlane((int)wc, 15);
            uint64_t run = carry + ((uint64_t)bhi << 20) + blo + inc - sum;
            const uint32_t crun = bc + cinc - cown;
            uint32_t dpos = dcarry + (crun & 0xFFFFu), ppos = pcarry + (crun >> 16);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k < n) {
                    w.rec_ub[i0 + k] = run;
                    run += v[t][k];
                    if (run > cap) atomicMin(&scap_s, i0 + k);  // rec_ub[s + 1] > recs_cap: no room from here on
                    if ((fdp[t] >> k) & 1u) w.defer_list[dpos++] = i0 + k;
                    if ((fdp[t] >> (4 + k)) & 1u) w.deep_list[ppos++] = i0 + k;
                }
            }
            carry += ttot;
            dcarry += tc & 0xFFFFu;
            pcarry += tc >> 16;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        w.rec_ub[n] = carry;
        w.tail_ctl[TC_NDEFER] = dcarry;
        w.tail_ctl[TC_NDEEP] = pcarry;
        w.tail_ctl[TC_SCAP] = scap_s;
    }
    // k_fanout_emit_seg's waves of THIS tick (of the emit_waves its launch has): the kernel's time follows its bytes where a
    // connection's range is long — 8 persistent waves per CU then, more cost the store stream (config B: 148 us, 12 -> 156) — and its
    // DESCRIPTORS where ranges are short, where twice the waves hide twice the latency (profiles/r08g_ab_emit_waves.txt: R = 1.5 cells,
    // 3.4 K records per connection: 12 waves 79 us, 8: 84; cells of 44 entities, 800 per connection: 16 waves 69 us, 8: 79).  Every
    // bank's line gets the count and its first ticket: workgroup b of the active ones starts with ticket b, without an atomic.
    if (seg && threadIdx.x < 8) {
        const uint64_t per_conn = n ? carry / n : 0ull;
        uint32_t act = per_conn >= w.emit_act_t1 ? w.seg_waves : per_conn >= w.emit_act_t2 ? w.seg_waves + w.seg_waves / 2u : 2u * w.seg_waves;
        act = min(act, w.emit_waves) & ~7u;
        if (act < 8u) act = 8u;
        w.emit_ticket[32u * threadIdx.x] = act >> 3;
        w.emit_ticket[32u * threadIdx.x + 3u] = act;
    }
    // (the filtered kernel's work items: here only where k_fanout_emit_seg cannot build them beside its own work — it does not run, or
    // the tick's epilogue, which clears the cells' counts, runs beside it: pipelined ticks)
    if (seg && w.off_on && w.fcm_on && (w.seg_only || w.late_tot)) filt_items_block(w, ncell);
}

// -DCHD_PROFILE_FILT (diagnosis builds): cycles the loader wave spends preparing items / waiting at the hand-over barrier, and the
// streamer waves working / waiting, summed over the launch; chd_debug_filt_prof reads and clears them
#ifdef CHD_PROFILE_FILT
__device__ unsigned long long fc_wg[4 * 1024];
__device__ unsigned long long fc_prof[8 * 64];  // [counter][hashed slot]: one atomic per wave and counter at the kernel's end
extern "C" int chd_debug_filt_prof(unsigned long long *out) {
    static unsigned long long h[8 * 64], z[8 * 64];
    int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(fc_prof), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fc_prof), z, sizeof z);
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int i = 0; i < 64; i++) out[k] += h[k * 64 + i]; }
    return rc;
}
// per workgroup: wall_clock64 (100 MHz) at its loader's start and end, items it drew — written without atomics
extern "C" int chd_debug_filt_wgs(unsigned long long *out, unsigned n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fc_wg), sizeof(unsigned long long) * 4 * (n < 1024 ? n : 1024));
}
#define FCP_T0() long long fcp_t = clock64(); const unsigned long long fcp_w0 = wall_clock64(); unsigned long long fcp_acc[7] = {0, 0, 0, 0, 0, 0, 0}
#define FCP_ADD(k) do { long long _n = clock64(); fcp_acc[k] += (unsigned long long)(_n - fcp_t); fcp_t = _n; } while (0)
#define FCP_CNT(k, v) do { fcp_acc[k] += (unsigned long long)(v); } while (0)
#define FCP_END() do { const unsigned long long _w1 = wall_clock64(); fcp_acc[6] = _w1 - fcp_w0; if (lane == 0 && wave == 0 && blockIdx.x < 1024) { fc_wg[4 * blockIdx.x] = fcp_w0; fc_wg[4 * blockIdx.x + 1] = _w1; fc_wg[4 * blockIdx.x + 2] = fcp_acc[4]; fc_wg[4 * blockIdx.x + 3] = fcp_acc[5]; } if (lane == 0) for (int _k = 0; _k < 7; _k++) if (fcp_acc[_k]) atomicAdd(&fc_prof[_k * 64 + ((blockIdx.x * FC_WAVES + wave) & 63u)], fcp_acc[_k]); } while (0)
#else
#define FCP_T0() do { } while (0)
#define FCP_ADD(k) do { } while (0)
#define FCP_CNT(k, v) do { } while (0)
#define FCP_END() do { } while (0)
#endif

__global__ void __launch_bounds__(64 * FC_WAVES, FC_OCC) k_fanout_emit_filt_cm(DevGrid g, WorldDev w) {
    __shared__ FcHead heads[2];
    __shared__ FcTile tiles[2];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_items = *w.filt_nitems;
    unsigned long long wave_sum = 0;

    FCP_T0();
    if (wave == 0) {
        // ---- loader ----
        // items by TICKET, not by a static stride: an item is 1..64 descriptors of 1..8 windows over 1..4 rows, and with ~3 items per
        // workgroup a static deal left the launch waiting for its unluckiest workgroup (off-grid ticks: 287 us max against 114 avg)
        auto prepare = [&](uint32_t b, bool own_item) {
            FcHead &H = heads[b];
            FcTile &T = tiles[b];
            // (a workgroup's FIRST item is its own index — the tickets start behind the grid, k_fanout_scan —: one round trip less
            // before its eleven streamer waves have anything to do)
            // (... and the later ones from ONE COUNTER PER XCD — ticket t of bank b = item gridDim + 8 t + b, as k_fanout_emit_seg's: the
            // returning atomics of all eight XCDs on one address serialise at the memory side, ~35 ns each, and the workgroups finish
            // their first items at about the same time)
            uint32_t item = blockIdx.x;
            if (!own_item) {
                if (lane == 0) item = atomicAdd(&w.emit_ticket[32u * (blockIdx.x & 7u) + 2u], 1u);
                item = gridDim.x + 8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)item) + (blockIdx.x & 7u);
            }
            if (item >= n_items) {
                if (lane == 0) H.valid = 0;
                return;
            }
            const uint4 it = w.filt_items[item];
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.x), first = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.y);
            const uint32_t ndt = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.z), start = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.w);
            const uint32_t nd = ndt & 0xFFu, tn = ndt >> 8;
            // second round trip: the descriptors' list entries (one per lane) AND the cell's columns — lane l holds the adjacent
            // entries 128 r + 2 l, + 1 of row r (entries beyond the cell: spare, never used)
            uint4 e0 = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u), e1 = make_uint4(0u, 0u, 0u, 0u);
            if (lane < nd) {
                const uint4 *ep = w.cell_flist + ((size_t)c * w.S + first + lane) * 2;
                e0 = ep[0];
                e1 = ep[1];
            }
            // (row by row: the whole tile at once would take 80 registers of every wave of the kernel)
#pragma unroll 1
            for (uint32_t r = 0; r < 4; r++) {
                if (tn <= 128u * r) break;  // uniform
                const uint32_t i = 128u * r + 2u * lane;
                const u32x4 ce = *(const u32x4 *)(const void *)(w.ce8_view + start + i);
                u32x2 co[FC_TSLOTS];
#pragma unroll
                for (uint32_t j = 0; j < FC_TSLOTS; j++) co[j] = *(const u32x2 *)(const void *)(w.ce_off + (size_t)j * w.off_stride + start + i);
                *(u32x2 *)(void *)&T.chan[i] = u32x2{ce.x, ce.z};
                *(u32x2 *)(void *)&T.hist[i] = u32x2{ce.y, ce.w};
                // (an entity WITHOUT an update in slot j gets the offset 0xFFFFFFFF: outside every window's bounds)
#pragma unroll
                for (uint32_t j = 0; j < FC_TSLOTS; j++)
                    *(u32x2 *)(void *)&T.off[j][i] = u32x2{((ce.y >> j) & 1u) ? co[j].x : 0xFFFFFFFFu, ((ce.w >> j) & 1u) ? co[j].y : 0xFFFFFFFFu};
            }
            // third: the connection's words and the windows' tests
            uint32_t info = 0xFFFFFFFFu, out16 = 0, conn = 0;
            const uint32_t sidx = e1.x, pidx = sidx * w.capq + e0.w, nn = e0.y;
            if (lane < nd) {
                const uint64_t base = w.rec_ub[sidx], end = w.rec_ub[sidx + 1];
                conn = w.conn_id[sidx];
                const uint32_t lnw = min(e0.z & 15u, (uint32_t)FC_LWIN);
                const u32x4 *fwp = (const u32x4 *)(const void *)(w.filt_win + (size_t)pidx * CHD_FILT_WINS);  // (192-byte rows: 16-byte aligned)
                u32x4 t[6];
#pragma unroll
                for (int q = 0; q < 6; q++) t[q] = u32x4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if ((uint32_t)(2 * (q / 3)) < lnw) t[q] = fwp[q];
                if (end <= w.recs_cap) {
                    info = e0.z;
                    out16 = (uint32_t)((base + e0.x) / CHD_SEG_ALIGN);
                }
                u32x4 *dst = (u32x4 *)(void *)&H.win[lane][0][0];
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if ((uint32_t)(2 * (q / 3)) < lnw) dst[q] = t[q];
                // the runs of this descriptor's windows: two binary searches each over the staged slot-0 offsets (ascending on a sorted
                // cell, 0xFFFFFFFF = no update last; this wave wrote them just above: LDS operations of one wave execute in order)
                const bool srt = w.cell_sorted && w.cell_sorted[c] != 0u && tn == nn;  // (tn == nn: the whole cell is in the tile)
                const uint32_t tw[24] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w, t[2].x, t[2].y, t[2].z, t[2].w,
                                         t[3].x, t[3].y, t[3].z, t[3].w, t[4].x, t[4].y, t[4].z, t[4].w, t[5].x, t[5].y, t[5].z, t[5].w};
#pragma unroll
                for (int j = 0; j < FC_LWIN; j++) {
                    uint32_t run = FC_NORUN;
                    const uint32_t slots = tw[6 * j + 1], a_lo = tw[6 * j + 2], a_hi = tw[6 * j + 3];
                    // (a second cut through a slot the tile does not hold: no run, the window goes to the global-memory path whole)
                    const bool b_in_tile = tw[6 * j + 4] > tw[6 * j + 5] || ((slots >> 8) & 0xFFu) < FC_TSLOTS;
                    if (srt && (uint32_t)j < lnw && a_lo <= a_hi && (slots & 0xFFu) == 0u && b_in_tile) {
                        uint32_t i0 = 0, i1 = 0;
#pragma unroll
                        for (uint32_t step = 512u; step; step >>= 1) {
                            const uint32_t p0 = i0 + step, p1 = i1 + step;
                            if (p0 <= tn && T.off[0][p0 - 1u] < a_lo) i0 = p0;
                            if (p1 <= tn && T.off[0][p1 - 1u] <= a_hi) i1 = p1;
                        }
                        run = i0 | (i1 << 16);
                    }
                    H.run[lane][j] = run;
                }
            }
            if (lane < FC_DESCS) {
                H.out16[lane] = out16; H.n[lane] = nn; H.info[lane] = info; H.conn[lane] = conn; H.pidx[lane] = pidx; H.sidx[lane] = sidx;
            }
            if (lane == 0) { H.nd = nd; H.cch = c + g.id_start; H.start = start; H.valid = 1; H.ticket = 0; H.sorted = w.cell_sorted ? w.cell_sorted[c] : 0u; }
        };
        prepare(0, true);
        FCP_ADD(0);
        lds_barrier();
        FCP_ADD(1);
        for (uint32_t u = 0;; u++) {
            if (!heads[u & 1u].valid) break;
            FCP_CNT(4, 1); FCP_CNT(5, heads[u & 1u].nd);
            prepare((u + 1u) & 1u, false);
            FCP_ADD(0);
            lds_barrier();
            FCP_ADD(1);
        }
    } else {
        // ---- streamers ----
        lds_barrier();
        FCP_ADD(3);
        for (uint32_t u = 0;; u++) {
            FcHead &H = heads[u & 1u];
            const FcTile &T = tiles[u & 1u];
            if (!H.valid) break;
            const uint32_t nd = H.nd, cch = H.cch, start = H.start;
            const bool sorted = __builtin_amdgcn_readfirstlane((int)H.sorted) != 0;
            for (;;) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&H.ticket, 1u);
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                if (k >= nd) break;
                const uint32_t info = H.info[k];
                if (info == 0xFFFFFFFFu) continue;  // (the deferred launch leaves the connection's state as it was and flags the tick)
                const uint32_t n = H.n[k], conn = H.conn[k], pidx = H.pidx[k];
                const uint32_t nw = info & 15u, own = (info >> 8) & 0xFFu;
                // (a scalar: the segment's start is the same for every lane — kept in SGPRs, not as a per-lane pointer that would be spilled
                // and reloaded with a wait on the wave's own stores)
                chd_fanout_rec *__restrict__ out = w.recs + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)H.out16[k]) * CHD_SEG_ALIGN;
                uint32_t n_out = 0;
                uint32_t lane2 = 2u * lane;  // (opaque per descriptor: `recs + 16 * lane` hoisted to the kernel's start was spilled — see pad_segment_here)
                asm volatile("" : "+v"(lane2));
                for (uint32_t j = 0; j < nw; j++) {
                    // (the values become scalars INSIDE each branch: a vector register that merges a global load with an LDS read
                    // would make the compiler wait on the vm counter — i.e. drain the record stores — on the common path too)
                    uint32_t full, slots, a_lo, a_hi, b_lo, b_hi;
                    if (j < FC_LWIN) {
                        const uint32_t *f = H.win[k][j];
                        full = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[0]); slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[1]);
                        a_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[2]); a_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[3]);
                        b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[4]); b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[5]);
                    } else {  // (rare: more than FC_LWIN windows)
                        const uint32_t *f = (const uint32_t *)(const void *)(w.filt_win + (size_t)pidx * CHD_FILT_WINS + j);
                        full = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[0]); slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[1]);
                        a_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[2]); a_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[3]);
                        b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[4]); b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)f[5]);
                    }
                    const uint32_t sa = slots & 0xFFu, sb = (slots >> 8) & 0xFFu;
                    const bool use_a = a_lo <= a_hi, use_b = b_lo <= b_hi;  // (use_b only with use_a)
                    if ((own >> j) & 1u) {  // the spatial channel's own buffered update lies inside this window
                        if (lane == 0) {
                            chd_fanout_rec r;
                            r.conn = conn;
                            r.channel = cch;
                            out[n_out] = r;
                        }
                        n_out += 1;
                    }
                    const uint32_t run = (sorted && j < FC_LWIN) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)H.run[k][j]) : FC_NORUN;
                    if (n <= 512 && run != FC_NORUN) {
                        // THE WINDOW'S PART INSIDE THIS TICK'S OWN ARRIVALS, on a cell whose entries are in arrival order (k_cell_arrange): the
                        // entities it selects are the run [i0, i1) of the cell's column (the loader found the two ends) — a plain copy, two
                        // records per lane and store
                        const uint32_t i0 = run & 0xFFFFu, i1 = run >> 16, len = i1 - i0;
                        typedef u32x4 __attribute__((aligned(8))) u32x4_a8;  // (n_out is any record index)
                        for (uint32_t b0 = 0; b0 < len; b0 += 128) {
                            const uint32_t q = b0 + lane2, ea = min(i0 + q, 511u), eb = min(i0 + q + 1u, 511u);
                            const uint32_t c0 = T.chan[ea], c1 = T.chan[eb];
                            if (q + 1 < len) {
                                u32x4 r;
                                r.x = conn; r.y = c0; r.z = conn; r.w = c1;
                                *(u32x4_a8 *)(void *)(out + n_out + q) = r;
                            } else if (q < len) {
                                chd_fanout_rec r;
                                r.conn = conn;
                                r.channel = c0;
                                out[n_out + q] = r;
                            }
                        }
                        n_out += len;
                        // ... and whatever else the window selects — a ring slot it covers whole, the cut through an older slot's arrivals —
                        // by the per-entity compare over the entries OUTSIDE the run (an entity inside it has its record)
                        if (full | (use_b ? 1u : 0u)) {
                            const uint32_t rest = n - len, b_rng = b_hi - b_lo;
                            const uint32_t *ob_col = T.off[use_b ? sb : 0u];
                            for (uint32_t b0 = 0; b0 < rest; b0 += 128) {
                                const uint32_t q0 = b0 + 2 * lane, q1 = q0 + 1u;
                                const uint32_t e0 = min(q0 < i0 ? q0 : q0 + len, 511u), e1 = min(q1 < i0 ? q1 : q1 + len, 511u);
                                const uint32_t h0 = T.hist[e0], h1 = T.hist[e1], o0 = ob_col[e0], o1 = ob_col[e1];
                                const bool p0 = (q0 < rest) & (((h0 & full) != 0) | (use_b & (o0 - b_lo <= b_rng)));
                                const bool p1 = (q1 < rest) & (((h1 & full) != 0) | (use_b & (o1 - b_lo <= b_rng)));
                                const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
                                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
                                if (p0 & p1) {
                                    u32x4 r;
                                    r.x = conn; r.y = T.chan[e0]; r.z = conn; r.w = T.chan[e1];
                                    *(u32x4_a8 *)(void *)(out + at) = r;
                                } else if (p0 | p1) {
                                    chd_fanout_rec r;
                                    r.conn = conn;
                                    r.channel = p0 ? T.chan[e0] : T.chan[e1];
                                    out[at] = r;
                                }
                                n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
                            }
                        }
                    } else if (n <= 512 && (!use_a || sa < FC_TSLOTS) && (!use_b || sb < FC_TSLOTS)) {
                        // THE PER-ENTITY PATH, LDS only (since the cells come in arrival order a RARE one: windows without a run — a cell too
                        // crowded to sort, a window that lies in older slots only; ~1 % of the windows, profiles/r09m_filt_paths.txt — so it
                        // goes row by row and holds a handful of registers; unrolled over the four rows it held 32 and set the kernel's
                        // count).  Per row of 128 entries the lanes do two and + compare for the whole-slot mask and two subtract + compare
                        // per cut slot — the staged offsets of entities WITHOUT an update in a slot are 0xFFFFFFFF, so the range test alone
                        // decides (bounds never exceed 0xFFFFFFFE) — and everything else runs on the scalar unit over the ballots.
                        typedef u32x4 __attribute__((aligned(8))) u32x4_a8;  // (n_out is any record index)
                        const uint32_t a_rng = a_hi - a_lo, b_rng = b_hi - b_lo;
                        const uint32_t *oa_col = T.off[use_a ? sa : 0u] + lane2, *ob_col = T.off[use_b ? sb : 0u] + lane2;
#pragma unroll 1
                        for (uint32_t r0 = 0; r0 < n; r0 += 128) {
                            const uint32_t left = n - r0;  // entries of this row and beyond
                            const uint64_t in0 = left >= 127u ? ~0ull : ((1ull << ((left + 1u) >> 1)) - 1ull);
                            const uint64_t in1 = left >= 128u ? ~0ull : ((1ull << (left >> 1)) - 1ull);
                            const u32x2 chv = *(const u32x2 *)(const void *)&T.chan[r0 + lane2];
                            uint64_t a0 = 0, a1 = 0;
                            if (full) {
                                const u32x2 hv = *(const u32x2 *)(const void *)&T.hist[r0 + lane2];
                                a0 = __ballot((hv.x & full) != 0); a1 = __ballot((hv.y & full) != 0);
                            }
                            if (use_a) {
                                const u32x2 o = *(const u32x2 *)(const void *)&oa_col[r0];
                                a0 |= __ballot(o.x - a_lo <= a_rng);
                                a1 |= __ballot(o.y - a_lo <= a_rng);
                            }
                            if (use_b) {
                                const u32x2 o = *(const u32x2 *)(const void *)&ob_col[r0];
                                a0 |= __ballot(o.x - b_lo <= b_rng);
                                a1 |= __ballot(o.y - b_lo <= b_rng);
                            }
                            const uint64_t m0 = a0 & in0, m1 = a1 & in1;
                            // entry order: records before this lane's pair = passing entries of lower lanes; a lane's two records are
                            // adjacent: one 16-byte store when both pass
                            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                                                __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, n_out))));
                            const bool pass0 = (m0 >> lane) & 1ull, pass1 = (m1 >> lane) & 1ull;
                            if (pass0 & pass1) {
                                u32x4 r;
                                r.x = conn; r.y = chv.x; r.z = conn; r.w = chv.y;
                                *(u32x4_a8 *)(void *)(out + at) = r;
                            } else if (pass0 | pass1) {
                                chd_fanout_rec r;
                                r.conn = conn;
                                r.channel = pass0 ? chv.x : chv.y;
                                out[at] = r;
                            }
                            n_out += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
                        }
                    } else {
                        // (rare: a cell beyond the tile — its columns from global memory, step by step, tested with the history bits)
                        n_out = filt_window_global(w, start, n, full, sa, sb, a_lo, a_hi, b_lo, b_hi, conn, out, n_out);
                    }
                }
                pad_segment_here(out, n_out);
                if (lane == 0) {
                    w.pair_nrec[pidx] = n_out;
                    if (n_out) atomicAdd(&w.rec_cnt[H.sidx[k]], n_out);
                }
                wave_sum += n_out;
            }
            FCP_ADD(2);
            lds_barrier();
            FCP_ADD(3);
        }
    }
    FCP_END();
    if (lane == 0 && wave_sum) {
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)((blockIdx.x * FC_WAVES + wave) & 63u) * 16];
        if (w.late_tot) atomicAdd(slot + 8, wave_sum);  // (pipelined: the epilogue runs beside this kernel — k_filt_fold)
        else {
            atomicAdd(slot, wave_sum);
            atomicAdd(slot + 2, wave_sum);  // (not written by the dominant emit kernel)
            atomicAdd(slot + 4, wave_sum);  // (chd_tick_stats.n_filtered_records)
        }
    }
}

// Pipelined ticks on worlds with sub-tick offsets: the filtered kernel's record count joins the tick's row of the history ring — total,
// deferred share, filtered share, as the epilogue would have written them had it run behind that kernel (serial schedule)
__global__ void __launch_bounds__(64) k_filt_fold(WorldDev w, uint32_t slot) {
    const uint32_t lane = threadIdx.x;
    unsigned long long v = w.tot64[(size_t)lane * 16 + 8];
    w.tot64[(size_t)lane * 16 + 8] = 0;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0 && v) {
        uint64_t *r = w.tick_ring + (size_t)slot * 8;
        auto hi_add = [](uint64_t x, unsigned long long a) {
            const unsigned long long h = (x >> 32) + a;
            return (x & 0xFFFFFFFFull) | ((h > 0xFFFFFFFFull ? 0xFFFFFFFFull : h) << 32);
        };
        r[0] += v;
        r[3] = hi_add(r[3], v);
        r[6] = hi_add(r[6], v);
    }
}

void launch_filt_fold(hipStream_t st, WorldDev w, uint32_t ring_slot) { hipLaunchKernelGGL(k_filt_fold, dim3(1), dim3(64), 0, st, w, ring_slot); }

void launch_fanout_emit_filt(hipStream_t st, DevGrid g, WorldDev w) {
    if (!w.S || !w.off_on || !seg_path(w)) return;
    if (w.fcm_on) {
        hipLaunchKernelGGL(k_fanout_emit_filt_cm, dim3(fc_grid(w, g.ncell)), dim3(64 * FC_WAVES), 0, st, g, w);
        return;
    }
    const uint32_t n_tickets = w.S * FO_FILT_WAVES;
    static const uint32_t per_cu = [] { const char *e = getenv("CHD_FILT_WAVES_PER_CU"); return e ? (uint32_t)std::min(std::max(atoi(e), 1), 32) : (uint32_t)FO_FILT_PER_CU; }();
    const uint32_t waves = w.seg_waves / 8u * per_cu;  // (seg_waves = 8 per CU unless CHD_EMIT_WAVES_PER_CU says otherwise)
    const dim3 grid(n_tickets < waves ? n_tickets : waves);
    hipLaunchKernelGGL((k_fanout_emit_filt<FO_FILT_WAVES>), grid, dim3(64), 0, st, g, w, n_tickets);
}

// ---------------------------------------------------------------------------
// Cell-major emit (grids up to 4096 cells), wave-specialised and persistent.
//
// Work item = (active cell c, chunk of 256 connection slots); unit = (item, tile of
// WS_TILE entity entries of c).  A persistent grid strides over the items.  In every
// workgroup wave 0 is the LOADER and waves 1-3 are STREAMERS, double-buffered
// through LDS:
//   loader   new item: one lane per connection slot — interest-bitmap test, rank in
//            the bitmap row (= index of the subscription in the connection's
//            cell-sorted list), due test of tickData (data.go:175-291) -> list of due
//            subscriptions in LDS (ballot-compacted, deterministic order);
//            every unit: the tile's 16-byte entries -> LDS (SoA)
//   streamer each due subscription x each non-empty fan-out window: stream the LDS
//            tile, ballot/mbcnt compaction, 512-byte contiguous wave stores of
//            {conn, channel} records; fan-out state write-back on the last tile
// The streamers never issue a global load, so they never execute an s_waitcnt vmcnt:
// on gfx950 the vm counter is in-order, and a wave that waits for a load also drains
// every record store it issued before it (measured: that drain + the pointer chasing
// of the due test kept the store stream at 55 % of what a plain store kernel reaches).
// The workgroup barrier between units orders LDS only.  Each cell table is read from
// L2 once per (cell, chunk) instead of once per subscription; the HBM traffic that
// remains is the 8 B/record stream.
// ---------------------------------------------------------------------------
#define WS_TILE 512
#define WS_SUBS 256
#define WS_ROUNDS (WS_SUBS / 64)
#define WS_WAVES 8
#define WS_STREAMERS (WS_WAVES - 1)

struct WsList {  // the due subscriptions of one item (LDS copy of WsItemG's first ndue entries)
    uint32_t pi[WS_SUBS], conn[WS_SUBS], flags[WS_SUBS], out16[WS_SUBS], nout[WS_SUBS];
    uint32_t wm[4][WS_SUBS];
    uint32_t ndue, c, item;
};

struct WsTile {  // one tile of the item's cell
    uint32_t chan[WS_TILE], hist[WS_TILE], snd[WS_TILE], hprev[WS_TILE], sprev[WS_TILE];
    uint32_t valid, tn, gpos, first, last, list_buf, any_prev, ticket;
};

__device__ __forceinline__ void lds_barrier() {
    // orders LDS only: __syncthreads() would add s_waitcnt vmcnt(0), i.e. drain the record stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The streamers' inner loop.  Everything that is wave-uniform stays on the scalar unit: the
// bounds mask comes from the tile length, the window / sender tests are ballots combined as
// 64-bit scalars, the write position is mbcnt(mask) + n_out with n_out as mbcnt's addend.
// Per 64 records that leaves ~6 vector ALU instructions, 3 LDS reads and one 512-byte store.
// PREV = some entry of the tile still buffers updates of a previous sender (rare): only then
// the two-sender test (and its side-table load) is compiled in.
template <bool FULL, bool PREV>
__device__ __forceinline__ uint32_t emit_tile(const WorldDev &w, const WsTile &T, uint32_t wm, bool skip_self,
                                              uint32_t conn, uint32_t conn_tag, chd_fanout_rec *__restrict__ out,
                                              uint32_t *__restrict__ opos, uint32_t n_out) {
    const uint32_t lane = lane_id();
    const uint32_t tn = T.tn;
    for (uint32_t b = 0; b < tn; b += 256) {
        uint32_t chan[4], ha[4], hb[4], snd[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // reads beyond tn stay inside the tile arrays (WS_TILE is a multiple of 256) and are masked below
            const uint32_t idx = b + j * 64 + lane;
            chan[j] = T.chan[idx];
            if (!FULL) {
                ha[j] = T.hist[idx];
                snd[j] = T.snd[idx];
                if (PREV) hb[j] = T.hprev[idx];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (tn - b <= (uint32_t)(j * 64)) break;  // uniform
            bool pl = b + j * 64 + lane < tn;
            if (!FULL) {
                bool pa = (ha[j] & wm) != 0;
                if (skip_self) pa = pa && snd[j] != conn;
                if (PREV) {
                    const bool pb = pl && (hb[j] & wm) != 0;
                    if (__ballot(pb)) {
                        bool other = pb;
                        if (pb && skip_self) other = T.sprev[b + j * 64 + lane] != conn;
                        pa = pa || other;
                    }
                }
                pl = pl && pa;
            }
            const uint64_t m = __ballot(pl);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, n_out));
            if (pl) {
                chd_fanout_rec r;
                r.conn = conn_tag;
                r.channel = chan[j];
                out[pos] = r;
                if (opos) opos[pos] = T.gpos + b + j * 64 + lane;
            }
            n_out += (uint32_t)__popcll(m);
        }
    }
    return n_out;
}

// window [lo, hi] as a mask over the tick ring (bit j = the stamp of tick cur-j lies in it)
__device__ __forceinline__ uint32_t window_mask_serial(const TickRing &ring, int64_t lo, int64_t hi) {
    uint32_t m = 0;
    for (uint32_t j = 0; j < ring.n; j++) m |= (ring.t[j] >= lo && ring.t[j] <= hi) ? (1u << j) : 0u;
    return m;
}

// empty_windows for a single thread (the stamps come from the kernel argument)
__device__ __forceinline__ int64_t empty_windows_serial(const TickRing &ring, int64_t now, int64_t L, uint32_t iv) {
    uint32_t newer = 0;
    for (uint32_t j = 0; j < ring.n; j++) newer += ring.t[j] > L + (int64_t)iv * 1000000 ? 1u : 0u;
    return empty_windows_to(newer ? ring.t[newer - 1] : 0, newer, now, L, iv);
}

// K5a': per work item (active cell, chunk of 256 connection slots) the list of due
// subscriptions, in global memory.  One thread per connection slot: interest-bitmap test,
// rank in the bitmap row (= index of the subscription in the connection's cell-sorted
// list), due test of tickData (data.go:194-199) and the whole catch-up window walk
// (data.go:224-271 + the revisit through :273-286): the non-empty windows become masks
// over the tick ring, the subscription's lastFanOutTime / hadFirstFanOut are advanced here,
// so that the emit kernel's streamers only replay masks.  Ballot + LDS compaction keeps the
// list order deterministic.  Subscribed but not due -> its record count is 0.
__global__ void __launch_bounds__(WS_SUBS) k_fanout_items(DevGrid g, WorldDev w, int64_t now, TickRing ring,
                                                          uint32_t chunks) {
    __shared__ uint32_t wcnt[WS_SUBS / 64], wsub[WS_SUBS / 64];
    const uint32_t n_items = *w.n_active * chunks;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t c = w.active_cells[item / chunks];
        const uint32_t s = (item % chunks) * WS_SUBS + threadIdx.x;
        const uint32_t wi = c >> 6, bit = c & 63u;
        uint32_t ch_hist = 0, ch_hprev = 0;
        {
            const uint32_t age = ring.cur_tick - w.cell_hist_tick[c];
            if (age < CHD_HIST_BITS) { ch_hist = w.cell_hist[c] << age; ch_hprev = w.cell_hist_prev[c] << age; }
        }
        const uint32_t ch_sender = w.cell_sender[c], ch_sprev = w.cell_sender_prev[c];
        bool sub = false, due = false;
        uint32_t fl = 0, iv = 0, conn = 0, flags = 0, wms[4] = {0, 0, 0, 0};
        int64_t L = 0;
        uint64_t outpos = 0;
        size_t pi = 0;
        if (s < w.S) {
            const unsigned long long *row = w.sub_bits + (size_t)s * w.wb;
            const unsigned long long word = row[wi];
            uint32_t pre = 0;
            for (uint32_t k = 0; k < wi; k++) pre += (uint32_t)__popcll(row[k]);
            const uint32_t alive = w.sub_alive[s];
            const uint64_t ub0 = w.rec_ub[s], ub1 = w.rec_ub[s + 1];
            conn = w.conn_id[s];
            sub = alive && ((word >> bit) & 1ull);
            if (sub) {
                pi = (size_t)s * w.capq + pre + (uint32_t)__popcll(word & ((1ull << bit) - 1ull));
                fl = w.pair_flags[pi];
                L = w.pair_last[pi];
                iv = w.pair_iv[pi];
                const int64_t I = (int64_t)iv * 1000000;
                due = !(fl & (PF_NO_ACCESS | PF_DEEP)) && I > 0 && now >= L + I;  // (PF_DEEP: k_fanout_plan gave it to k_fanout_emit_deep)
                if (due && w.cell_cov && !w.cell_cov[c]) atomicOr(&w.counters[CTR_OVERFLOW], OVF_HALO);  // (see k_fanout_plan)
                if (due && ub1 > w.recs_cap) {
                    // no room for this connection's worst case: state untouched, it catches up next tick
                    atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
                    due = false;
                }
                if (due) {
                    outpos = ub0 + w.pair_rel[pi];
                    const bool skip_self = (fl & PF_SKIP_SELF) != 0;
                    int64_t Lw = L;
                    if (!(fl & PF_HAD_FIRST)) {  // data.go:217-223: full state, last = t
                        flags |= WSF_FIRST;
                        Lw = now;
                    }
                    if (skip_self) flags |= WSF_SKIP_SELF;
                    uint32_t nw = 0;
                    if (now >= Lw + I) {
                        if (history_lost(ring, ring.n ? ring.t[ring.n - 1] : INT64_MAX, Lw, I)) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
                        while (now >= Lw + I) {
                            const int64_t next = Lw + I;
                            const uint32_t wm = window_mask_serial(ring, Lw > 0 ? Lw : 0, next);
                            if (wm) {
                                if (nw < 4) {
                                    wms[nw] = wm;
                                    if (cell_update_passes(ch_hist, ch_sender, ch_hprev, ch_sprev, wm, skip_self, conn))
                                        flags |= 1u << (WSF_OWN_SHIFT + nw);
                                }
                                nw++;
                            } else {
                                Lw += empty_windows_serial(ring, now, Lw, iv) * I;
                                continue;
                            }
                            Lw = next;
                        }
                    }
                    if (nw > 4) flags |= WSF_GENERIC;
                    flags |= (nw > 4 ? 4u : nw) << WSF_NWIN_SHIFT;
                    w.pair_last[pi] = Lw;
                    w.pair_flags[pi] = fl | PF_HAD_FIRST;
                } else if (!(fl & PF_DEEP)) {
                    w.pair_nrec[pi] = 0;
                }
            }
        }
        const uint64_t m = __ballot(due), sm = __ballot(sub);
        __syncthreads();
        if (lane == 0) { wcnt[wave] = (uint32_t)__popcll(m); wsub[wave] = (uint32_t)__popcll(sm); }
        __syncthreads();
        uint32_t base = 0, total = 0, nsub = 0;
        for (uint32_t k = 0; k < WS_SUBS / 64; k++) {
            if (k < wave) base += wcnt[k];
            total += wcnt[k];
            nsub += wsub[k];
        }
        WsItemG &G = w.items[item];
        if (due) {
            const uint32_t k = base + mask_rank(m);
            G.pi[k] = (uint32_t)pi; G.conn[k] = conn; G.flags[k] = flags; G.out16[k] = (uint32_t)(outpos / CHD_SEG_ALIGN);
            G.wm[0][k] = wms[0]; G.wm[1][k] = wms[1]; G.wm[2][k] = wms[2]; G.wm[3][k] = wms[3];
            G.iv[k] = iv; G.L[k] = L;
        }
        if (threadIdx.x == 0) {
            G.ndue = total;
            G.nsub = nsub;
            G.c = c;
            G.ch_hist = ch_hist; G.ch_hprev = ch_hprev; G.ch_sender = ch_sender; G.ch_sprev = ch_sprev;
            G.start = w.cell_start[c];
            G.end = w.cell_end[c];
        }
    }
}

// loader: the item's due list global -> LDS (the first ndue entries of each array)
__device__ __forceinline__ uint32_t ws_load_item(const WsItemG &G, uint32_t item, WsList &Lst, uint32_t &start,
                                                 uint32_t &end, uint32_t &n_subscribed) {
    const uint32_t lane = lane_id();
    const uint32_t nd = G.ndue;
    n_subscribed += G.nsub;
    if (nd == 0) return 0;
    start = G.start;
    end = G.end;
    for (uint32_t k = lane; k < nd; k += 64) {
        Lst.pi[k] = G.pi[k]; Lst.conn[k] = G.conn[k]; Lst.flags[k] = G.flags[k]; Lst.out16[k] = G.out16[k];
        Lst.wm[0][k] = G.wm[0][k]; Lst.wm[1][k] = G.wm[1][k]; Lst.wm[2][k] = G.wm[2][k]; Lst.wm[3][k] = G.wm[3][k];
        Lst.nout[k] = 0;
    }
    if (lane == 0) { Lst.ndue = nd; Lst.c = G.c; Lst.item = item; }
    return nd;
}

__global__ void __launch_bounds__(64 * WS_WAVES) k_fanout_emit_ws(DevGrid g, WorldDev w, int64_t now, TickRing ring,
                                                                 uint32_t chunks) {
    __shared__ WsList lists[2];
    __shared__ WsTile tiles[2];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_items = *w.n_active * chunks;
    const int64_t my_t = ring_stamp(ring);
    // Consume the stamp once, here: the compiler's waitcnt pass otherwise keeps its load "pending" at
    // the window loop's header and puts an s_waitcnt vmcnt(0) INSIDE the loop, where it drains the
    // streamer's record stores once per window.
    const uint32_t hist_ovf = __ballot(my_t == INT64_MIN) ? 1u : 0u;  // (always 0)
    // loader state (uniform)
    uint32_t item = blockIdx.x, item_seq = 0, tile0 = 0, cell_end = 0;
    bool have_item = false;
    uint32_t n_subscribed = 0;
    // streamer accumulators
    unsigned long long wave_sum = 0;

    // prepares the next unit into tiles[tb] (and, for a new item, its due list)
    auto prepare = [&](uint32_t tb) {
        WsTile &T = tiles[tb];
        for (;;) {
            if (!have_item) {
                if (item >= n_items) {
                    if (lane == 0) T.valid = 0;
                    return;
                }
                const uint32_t nd = ws_load_item(w.items[item], item, lists[item_seq & 1u], tile0, cell_end, n_subscribed);
                item += gridDim.x;
                if (nd == 0) continue;  // nobody in this chunk is due for this cell: no unit at all
                have_item = true;
                item_seq++;
                if (lane == 0) { T.first = 1; }
            } else if (lane == 0) {
                T.first = 0;
            }
            const uint32_t tn = min(cell_end - tile0, (uint32_t)WS_TILE);
            uint32_t prev_or = 0;
            {
                static_assert(WS_TILE == 512, "eight 16-byte loads per lane");
                u32x4 e[8];
                const uint4 *q[8];
                const uint32_t last_i = tn ? tn - 1u : 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = w.ce_view + tile0 + min((uint32_t)(j * 64) + lane, last_i);
                // one round trip for the whole tile (the compiler would wait after every load)
                asm volatile(
                    "global_load_dwordx4 %0, %8, off\n\t"
                    "global_load_dwordx4 %1, %9, off\n\t"
                    "global_load_dwordx4 %2, %10, off\n\t"
                    "global_load_dwordx4 %3, %11, off\n\t"
                    "global_load_dwordx4 %4, %12, off\n\t"
                    "global_load_dwordx4 %5, %13, off\n\t"
                    "global_load_dwordx4 %6, %14, off\n\t"
                    "global_load_dwordx4 %7, %15, off\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6]), "=&v"(e[7])
                    : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
                    : "memory");
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t idx = j * 64 + lane;
                    if (idx < tn) {
                        T.chan[idx] = e[j].x; T.hist[idx] = e[j].y; T.snd[idx] = e[j].z; T.hprev[idx] = e[j].w;
                        prev_or |= e[j].w;
                    }
                }
                if (__ballot(prev_or != 0)) {  // rare: previous senders' ids for the two-sender test
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t idx = j * 64 + lane;
                        if (idx < tn) T.sprev[idx] = load_sprev(w, tile0 + idx);
                    }
                }
            }
            const uint64_t any_prev = __ballot(prev_or != 0);
            if (lane == 0) {
                T.any_prev = any_prev ? 1u : 0u;
                T.ticket = 0;
                T.valid = 1;
                T.tn = tn;
                T.gpos = tile0;
                T.last = tile0 + tn >= cell_end ? 1u : 0u;
                T.list_buf = (item_seq - 1u) & 1u;
            }
            tile0 += tn;
            if (tile0 >= cell_end) have_item = false;
            return;
        }
    };

    // The two roles run in SEPARATE loops that meet only at the workgroup barrier (same number of
    // barriers on both sides).  With one shared loop the compiler merges the loader's pending-load
    // state into the streamers' code and protects their registers with s_waitcnt vmcnt(0) — which,
    // on the in-order vm counter, drains the streamers' record stores once per subscription.
#ifdef CHD_PROFILE_EMIT
    long long t_work = 0, t_wait = 0, t_mark = clock64();
    uint32_t n_units = 0;
#define WS_MARK(acc) do { long long _t = clock64(); acc += _t - t_mark; t_mark = _t; } while (0)
#else
#define WS_MARK(acc) do { } while (0)
#endif
    if (wave == 0) {
        prepare(0);
        WS_MARK(t_work);
        lds_barrier();
        WS_MARK(t_wait);
        for (uint32_t u = 0;; u++) {
            if (!tiles[u & 1u].valid) break;
            prepare((u + 1u) & 1u);
            WS_MARK(t_work);
            lds_barrier();
            WS_MARK(t_wait);
        }
    } else {
        lds_barrier();
        WS_MARK(t_wait);
        for (uint32_t u = 0;; u++) {
            const WsTile &T = tiles[u & 1u];
            if (!T.valid) break;
#ifdef CHD_PROFILE_EMIT
            n_units++;
#endif
            WsList &Lst = lists[T.list_buf];
            const uint32_t ndue = Lst.ndue, c = Lst.c;
            const bool first_tile = T.first != 0, last_tile = T.last != 0;
            for (;;) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&tiles[u & 1u].ticket, 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                if (k >= ndue) break;
                const uint32_t flags = Lst.flags[k];
                const uint32_t conn = Lst.conn[k];
                const bool skip_self = (flags & WSF_SKIP_SELF) != 0;
                chd_fanout_rec *__restrict__ out = w.recs + (size_t)Lst.out16[k] * CHD_SEG_ALIGN;
                uint32_t *__restrict__ opos = w.rec_pos ? w.rec_pos + (size_t)Lst.out16[k] * CHD_SEG_ALIGN : nullptr;
                uint32_t n_out = Lst.nout[k];
                if (flags & WSF_FIRST) {
                    // first fan-out: whole data of the spatial channel and of every entity channel in it
                    if (first_tile) {
                        if (lane == 0) {
                            chd_fanout_rec r;
                            r.conn = conn | CHD_REC_FULL;
                            r.channel = c + g.id_start;
                            out[0] = r;
                            if (opos) opos[0] = CHD_POS_CELL | c;
                        }
                        n_out = 1;
                    }
                    n_out = emit_tile<true, false>(w, T, 0u, false, conn, conn | CHD_REC_FULL, out, opos, n_out);
                }
                if (!(flags & WSF_GENERIC)) {
                    const uint32_t nw = (flags >> WSF_NWIN_SHIFT) & 7u;
                    for (uint32_t j = 0; j < nw; j++) {
                        const uint32_t wm = Lst.wm[j][k];
                        if (first_tile && ((flags >> (WSF_OWN_SHIFT + j)) & 1u)) {  // the spatial channel's own update
                            if (lane == 0) {
                                chd_fanout_rec r;
                                r.conn = conn;
                                r.channel = c + g.id_start;
                                out[n_out] = r;
                                if (opos) opos[n_out] = CHD_POS_CELL | c;
                            }
                            n_out += 1;
                        }
                        n_out = T.any_prev ? emit_tile<false, true>(w, T, wm, skip_self, conn, conn, out, opos, n_out)
                                           : emit_tile<false, false>(w, T, wm, skip_self, conn, conn, out, opos, n_out);
                    }
                } else {
                    // more than four non-empty windows (a subscription that was not served for several ticks):
                    // walk them from the state before this tick (the only global loads a streamer ever issues)
                    const WsItemG &G = w.items[Lst.item];
                    int64_t L = (flags & WSF_FIRST) ? now : G.L[k];
                    const int64_t I = (int64_t)G.iv[k] * 1000000;
                    if (now >= L + I) {
                        while (now >= L + I) {
                            const int64_t next = L + I;
                            const uint32_t wm = window_mask(my_t, L > 0 ? L : 0, next);
                            if (wm) {
                                if (first_tile && cell_update_passes(G.ch_hist, G.ch_sender, G.ch_hprev, G.ch_sprev, wm, skip_self, conn)) {
                                    if (lane == 0) {
                                        chd_fanout_rec r;
                                        r.conn = conn;
                                        r.channel = c + g.id_start;
                                        out[n_out] = r;
                                        if (opos) opos[n_out] = CHD_POS_CELL | c;
                                    }
                                    n_out += 1;
                                }
                                n_out = T.any_prev ? emit_tile<false, true>(w, T, wm, skip_self, conn, conn, out, opos, n_out)
                                                   : emit_tile<false, false>(w, T, wm, skip_self, conn, conn, out, opos, n_out);
                            } else {
                                L += empty_windows(ring, my_t, now, L, G.iv[k]) * I;
                                continue;
                            }
                            L = next;
                        }
                    }
                }
                if (last_tile) pad_segment(out, n_out);
                if (lane == 0) {
                    Lst.nout[k] = n_out;
                    if (last_tile) {
                        w.pair_nrec[Lst.pi[k]] = n_out;
                        wave_sum += n_out;
                    }
                }
            }
            WS_MARK(t_work);
            lds_barrier();
            WS_MARK(t_wait);
        }
    }
#ifdef CHD_PROFILE_EMIT
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 517) && ring.cur_tick == 60)
        printf("emit_ws block %u wave %u: units %u work %lld wait %lld (records %llu)\n", blockIdx.x, wave, n_units, t_work,
               t_wait, wave_sum);
#endif
    if (lane == 0) {
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)((blockIdx.x * WS_WAVES + wave) & 63u) * 16];
        if (wave_sum) atomicAdd(slot, wave_sum);
        if (n_subscribed) atomicAdd(slot + 1, (unsigned long long)n_subscribed);
        if (hist_ovf) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    }
}

bool fanout_seg_path(const WorldDev &w) { return seg_path(w); }

// the kernel that writes (nearly) all records of the tick
void launch_fanout_emit_main(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring) {
    if (!w.S) return;
    if (w.cm_emit) {
        const uint32_t chunks = (w.S + WS_SUBS - 1) / WS_SUBS;
        const uint64_t max_items = (uint64_t)g.ncell * chunks;
        const uint32_t grid = (uint32_t)(max_items < w.emit_grid ? max_items : w.emit_grid);
        const uint32_t pgrid = (uint32_t)(max_items < 8u * w.emit_grid ? max_items : 8u * w.emit_grid);
        hipLaunchKernelGGL(k_fanout_items, dim3(pgrid), dim3(WS_SUBS), 0, st, g, w, now_ns, ring, chunks);
        hipLaunchKernelGGL(k_fanout_emit_ws, dim3(grid), dim3(64 * WS_WAVES), 0, st, g, w, now_ns, ring, chunks);
        return;
    }
    // one wave per connection when the connections alone fill the chip (or when asked to: CHD_WORLD_ONE_WAVE_EMIT),
    // four waves per connection otherwise
    const bool one_wave = w.S >= 4096 || w.one_wave_emit;
    if (seg_path(w)) {
        // CHD_WORLD_SEGMENTS_ONLY: the simple descriptors' records are plain copies of columns the host expands itself from the segment
        // form (chd_tick_fetch_segments): nobody reads them from HBM.  (The deferred and filtered subscriptions' records ARE the
        // segment form's explicit records: their kernels run as ever.)
        if (w.seg_only) return;
        // (k_fanout_plan_seg has decided everything; see launch_fanout_plan)
        // CHD_SEG_TAIL="<percent>,<log2 pieces>": the last <percent> % of the connection slots in 2^<log2 pieces> pieces each
        static const uint32_t tail_pct = [] { const char *e = getenv("CHD_SEG_TAIL"); return e ? (uint32_t)atoi(e) : FO_SEG_TAIL_PCT; }();
        static const uint32_t tail_sh = [] { const char *e = getenv("CHD_SEG_TAIL"); const char *c = e ? strchr(e, ',') : nullptr; return c ? (uint32_t)atoi(c + 1) : FO_SEG_TAIL_SH; }();
        const uint32_t pct = tail_pct > 100u ? 100u : tail_pct, fsh = tail_sh < 1u ? 1u : tail_sh > 4u ? 4u : tail_sh;
        const uint32_t s_fine = w.S - (uint32_t)((uint64_t)w.S * pct / 100u);
        const uint32_t n_tickets = s_fine * FO_SEG_WAVES + ((w.S - s_fine) << fsh);
        const dim3 grid(n_tickets < w.emit_waves ? n_tickets : w.emit_waves);
        if (w.rec_mask) hipLaunchKernelGGL((k_fanout_emit_seg<FO_SEG_WAVES, true>), grid, dim3(64), 0, st, g, w, n_tickets, s_fine, fsh);
        else hipLaunchKernelGGL((k_fanout_emit_seg<FO_SEG_WAVES, false>), grid, dim3(64), 0, st, g, w, n_tickets, s_fine, fsh);
    } else if (w.rec_mask) {
        if (one_wave) hipLaunchKernelGGL((k_fanout_emit<1, true>), dim3(w.S), dim3(64), 0, st, g, w, now_ns, ring);
        else hipLaunchKernelGGL((k_fanout_emit<4, true>), dim3(w.S), dim3(256), 0, st, g, w, now_ns, ring);
    } else if (one_wave) hipLaunchKernelGGL((k_fanout_emit<1, false>), dim3(w.S), dim3(64), 0, st, g, w, now_ns, ring);
    else hipLaunchKernelGGL((k_fanout_emit<4, false>), dim3(w.S), dim3(256), 0, st, g, w, now_ns, ring);
}


// ---------------------------------------------------------------------------
// Exact update buffers (chd_world_cfg.history_depth): tickData's buffer walk itself (data.go:225-269), for the subscriptions
// the plan marked PF_DEEP.  One wave per connection.  Per subscription the due windows are k = 0 .. nwin-1, window k =
// [max(L + k I, 0), L + (k+1) I] (both ends inclusive: `be.arrivalTime >= lastUpdateTime && be.arrivalTime <= nextFanOutTime`
// with lastUpdateTime starting at max(lastFanOutTime, 0)); a channel gets one record per window that holds at least one
// buffered update from a sender the subscription does not skip — an update at arrival a lies in window floor((a - L) / I)
// and, when a sits exactly on that window's lower edge, in the one before as well.  One lane per channel of the cell walks
// that channel's buffer from its newest element back to the first one older than the subscription's reach: O(updates inside
// the reach), not O(buffer).  Two passes (count, wave prefix sum, write) keep a channel's records contiguous; the order of a
// segment's records is not the hot path's (window-major there, channel-major here) — a connection's records are a multiset.
// ---------------------------------------------------------------------------
// Which buffered updates a deep record's message merges (CHD_WORLD_UPDATE_MASKS on an exact world): the elements of the channel's
// buffer whose arrival lies inside the record's window, as a RANGE of the channel's update sequence numbers — bit 31 set, bits
// 30..21 = count - 1, bits 20..0 = sequence number (mod 2^21) of the oldest element of the range; sequence number = how many
// updates the channel had taken before that one.  (With SkipSelfUpdateFanOut the message merges the range's elements from other
// senders; the host knows who sent what.)
__device__ __forceinline__ uint32_t deep_range_word(uint32_t first_seq, uint32_t cnt) {
    return 0x80000000u | (((cnt > 1024u ? 1024u : cnt) - 1u) << 21) | (first_seq & 0x1FFFFFu);
}

template <bool WRITE>
__device__ __forceinline__ uint32_t deep_walk(const int64_t *__restrict__ A, const uint32_t *__restrict__ S, uint32_t D, uint32_t n,
                                              uint32_t len, int64_t drop, int64_t L, int64_t I, int64_t nwin, bool skip_self,
                                              uint32_t conn, uint32_t chan, uint32_t pos, chd_fanout_rec *__restrict__ out,
                                              uint32_t *__restrict__ opos, uint32_t *__restrict__ omask, uint32_t at, uint32_t &lost) {
    const int64_t lo0 = L > 0 ? L : 0, hi_all = L + nwin * I;
    if (drop >= lo0) lost = 1;  // an update the reference would still hold, inside this subscription's reach, is gone
    // Windows are met newest first; an element lies in window k = floor((a - L) / I) and, when it sits exactly on k's lower edge,
    // in k - 1 as well — so two adjacent windows can be open at a time (elements with EQUAL stamps on an edge alternate between
    // them).  Per open window: how many elements it holds so far and the oldest one's sequence number (the record's range), and
    // the record it got once an element from a sender the subscription does not skip turned up.
    int64_t wa = -1, wb = -1;                  // open windows (wb = wa - 1 when both are open); -1: none
    uint32_t na = 0, nb = 0;                   // elements held so far (the newest one examined is the oldest: the record's range)
    uint32_t ra = 0xFFFFFFFFu, rb = 0xFFFFFFFFu;  // record index (0xFFFFFFFF: none yet)
    uint32_t cnt = 0;
    for (uint32_t q = 0; q < len; q++) {
        const uint32_t idx = (n - 1u - q) % D, seq = n - 1u - q;
        const int64_t a = A[idx];
        if (a < lo0) break;  // (arrival order: everything further back is older still)
        if (a > hi_all) continue;
        const bool passes = !(skip_self && S[idx] == conn);
        const int64_t k = (a - L) / I;
        const bool edge = k >= 1 && (a - L) - k * I == 0;  // on the lower edge of window k = the upper edge of window k-1
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int64_t x = t == 0 ? k : k - 1;
            if (t == 0 ? k >= nwin : !edge) continue;
            bool is_a = true;
            if (x == wa) is_a = true;
            else if (x == wb) is_a = false;
            else {  // an older window: the open ones are complete
                wa = x; wb = x - 1;
                na = nb = 0; ra = rb = 0xFFFFFFFFu;
            }
            const uint32_t nn = (is_a ? na : nb) + 1u;
            uint32_t rr = is_a ? ra : rb;
            if (passes && rr == 0xFFFFFFFFu) {
                rr = at + cnt;
                if (WRITE) {
                    out[at + cnt].conn = conn; out[at + cnt].channel = chan;
                    if (opos) opos[at + cnt] = pos;
                }
                cnt++;
            }
            if (is_a) { na = nn; ra = rr; } else { nb = nn; rb = rr; }
            if (WRITE && omask && rr != 0xFFFFFFFFu) omask[rr] = deep_range_word(seq, nn);
        }
    }
    return cnt;
}

__device__ __forceinline__ void fanout_deep_conn(const DevGrid &g, const WorldDev &w, int64_t now, const TickRing &ring, const uint32_t s) {
    const uint32_t lane = lane_id();
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const uint64_t base = w.rec_ub[s];
    const bool room = w.rec_ub[s + 1] <= w.recs_cap;
    const uint32_t conn = w.conn_id[s];
    const uint32_t D = w.deep_depth;
    uint32_t total = 0, lost = 0;
    for (uint32_t p = 0; p < cnt; p++) {
        const uint32_t fl = w.pair_flags[pbase + p];
        if (!(fl & PF_DEEP)) continue;  // (uniform)
        if (!room) {
            // no room for this connection's worst case: its state stays, it catches up next tick (flagged)
            if (lane == 0) {
                w.pair_flags[pbase + p] = fl & ~PF_DEEP;
                w.pair_nrec[pbase + p] = 0;
                atomicOr(&w.counters[CTR_OVERFLOW], OVF_RECORDS);
            }
            continue;
        }
        const int64_t L = w.pair_last[pbase + p];
        const int64_t I = (int64_t)w.pair_iv[pbase + p] * 1000000;
        const int64_t nwin = div_windows_ns(now - L, I);  // >= 1: the plan found it due
        const uint32_t c = w.pair_cell[pbase + p];
        const uint32_t start = w.cell_start[c], end = w.cell_end[c];
        const bool skip_self = (fl & PF_SKIP_SELF) != 0;
        const uint32_t rel = w.pair_rel[pbase + p];
        chd_fanout_rec *__restrict__ out = w.recs + base + rel;
        uint32_t *__restrict__ opos = w.rec_pos ? w.rec_pos + base + rel : nullptr;
        uint32_t *__restrict__ omask = w.rec_mask ? w.rec_mask + base + rel : nullptr;
        uint32_t n_out = 0;
        // the spatial channel's own buffer: lane 0
        {
            uint32_t k0 = 0;
            const size_t at = (size_t)c * D;
            if (lane == 0)
                k0 = deep_walk<true>(w.cdeep_a + at, w.cdeep_s + at, D, w.cdeep_n[c], w.cdeep_len[c], w.cdeep_drop[c], L, I, nwin, skip_self,
                                     conn, c + g.id_start, CHD_POS_CELL | c, out, opos, omask, 0u, lost);
            n_out = (uint32_t)__shfl((int)k0, 0);
        }
        for (uint32_t b = start; b < end; b += 64) {
            const uint32_t pos = b + lane;
            const bool in = pos < end;
            uint32_t e = 0, chan = 0, k = 0;
            if (in) {
                e = w.ce_slot[pos];
                chan = w.ce_chan_view[pos];
                const size_t at = (size_t)e * D;
                k = deep_walk<false>(w.deep_a + at, w.deep_s + at, D, w.deep_n[e], w.deep_len[e], w.deep_drop[e], L, I, nwin, skip_self,
                                     conn, chan, pos, out, opos, omask, 0u, lost);
            }
            uint32_t inc = k;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
                if ((int)lane >= d) inc += o;
            }
            if (in && k) {
                const size_t at = (size_t)e * D;
                (void)deep_walk<true>(w.deep_a + at, w.deep_s + at, D, w.deep_n[e], w.deep_len[e], w.deep_drop[e], L, I, nwin, skip_self,
                                      conn, chan, pos, out, opos, omask, n_out + inc - k, lost);
            }
            n_out += (uint32_t)__shfl((int)inc, 63);
        }
        pad_segment(out, n_out);
        if (lane == 0) {
            w.pair_last[pbase + p] = L + nwin * I;
            w.pair_flags[pbase + p] = fl & ~PF_DEEP;
            w.pair_nrec[pbase + p] = n_out;
        }
        total += n_out;
    }
    if (__ballot(lost != 0) && lane == 0) atomicAdd(&w.counters[CTR_HIST_OVERFLOW], 1u);
    if (lane == 0 && total) {
        atomicAdd(&w.rec_cnt[s], total);
        unsigned long long *slot = (unsigned long long *)&w.tot64[(size_t)(s & 63u) * 16];
        atomicAdd(slot, (unsigned long long)total);
        atomicAdd(slot + 2, (unsigned long long)total);  // (not written by the dominant emit kernel)
        atomicAdd(slot + 3, (unsigned long long)total);  // (chd_tick_stats.n_deep_records)
    }
}

// Per-tick totals into the device-side history ring (read back by chd_tick_fetch /
// chd_get_tick_history), then the per-tick counters are cleared for the next tick.
static_assert(CHD_LIST_BANKS == 64, "one epilogue lane per list bank");
__device__ __forceinline__ void tick_epilogue_wave(const WorldDev &w, uint32_t slot, uint32_t ncell, unsigned long long *epi, unsigned long long epi_seq) {
    const uint32_t lane = threadIdx.x;
    if (w.deep_depth) {  // (set again by the next tick's index build)
        // ... and the spatial channels' maxFanOutIntervalMs as this tick's interest updates left it: what the NEXT tick's updates
        // are buffered under
        for (uint32_t c = lane; c < ncell; c += 64) {
            w.cell_irr[c] = 0;
            const uint32_t m = w.cell_max_iv[ncell + c];
            if (m > w.cell_max_iv[c]) w.cell_max_iv[c] = m;
        }
    }
    if (w.off_on && w.fcm_on)  // (the next tick's plan appends to the cells' lists of filtered descriptors)
        for (uint32_t c = lane; c < ncell; c += 64) w.cell_fcnt[32u * c] = 0;
    unsigned long long sum = w.tot64[(size_t)lane * 16], pairs = w.tot64[(size_t)lane * 16 + 1], deferred = w.tot64[(size_t)lane * 16 + 2];
    unsigned long long deepr = w.tot64[(size_t)lane * 16 + 3], filtr = w.tot64[(size_t)lane * 16 + 4];
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        pairs += __shfl_xor(pairs, d);
        deferred += __shfl_xor(deferred, d);
        deepr += __shfl_xor(deepr, d);
        filtr += __shfl_xor(filtr, d);
    }
    // unsub / new-sub bank tails: totals for the ring, per-bank counts kept for chd_tick_fetch
    uint32_t un = w.list_ctr[lane * 32u], nn = w.list_ctr[(CHD_LIST_BANKS + lane) * 32u];
    un = un < w.list_bank_cap ? un : w.list_bank_cap;
    nn = nn < w.list_bank_cap ? nn : w.list_bank_cap;
    w.list_bank_n[lane] = un;
    w.list_bank_n[CHD_LIST_BANKS + lane] = nn;
    w.list_ctr[lane * 32u] = 0;
    w.list_ctr[(CHD_LIST_BANKS + lane) * 32u] = 0;
    for (int d = 32; d >= 1; d >>= 1) {
        un += __shfl_xor(un, d);
        nn += __shfl_xor(nn, d);
    }
    if (lane == 0) {
        uint64_t *r = w.tick_ring + (size_t)slot * 8;
        r[0] = sum;
        r[1] = w.rec_ub[w.S];
        // (high halves, saturating: the records written by the element-buffer walk / by the filtered descriptors)
        r[2] = (uint64_t)w.counters[CTR_HANDOVERS] | ((deepr > 0xFFFFFFFFull ? 0xFFFFFFFFull : deepr) << 32);
        r[3] = (uint64_t)w.counters[CTR_LOCKED] | ((filtr > 0xFFFFFFFFull ? 0xFFFFFFFFull : filtr) << 32);
        r[4] = un;
        r[5] = nn;
        r[6] = (pairs & 0xFFFFFFFFull) | ((deferred > 0xFFFFFFFFull ? 0xFFFFFFFFull : deferred) << 32);
        r[7] = (uint64_t)w.counters[CTR_OVERFLOW] |
               ((uint64_t)(w.counters[CTR_HIST_OVERFLOW] + w.counters[CTR_SENDER_OVERFLOW]) << 32);
    }
    __syncthreads();
    if (lane < CTR_COUNT) w.counters[lane] = 0;
    w.tot64[(size_t)lane * 16] = 0;
    w.tot64[(size_t)lane * 16 + 1] = 0;
    w.tot64[(size_t)lane * 16 + 2] = 0;
    w.tot64[(size_t)lane * 16 + 3] = 0;
    w.tot64[(size_t)lane * 16 + 4] = 0;
    if (epi) {
        // CHD_WORLD_GATED_OVERLAP: the tick is over, and says so to the second stream's gate (the next tick's interest updates)
        __syncthreads();
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(epi, epi_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64) k_tick_epilogue(WorldDev w, uint32_t slot, uint32_t ncell, unsigned long long *epi, unsigned long long epi_seq) {
    tick_epilogue_wave(w, slot, ncell, epi, epi_seq);
}

void launch_tick_epilogue(hipStream_t st, WorldDev w, uint32_t slot, uint32_t ncell, unsigned long long *epi, unsigned long long epi_seq) {
    hipLaunchKernelGGL(k_tick_epilogue, dim3(1), dim3(64), 0, st, w, slot, ncell, epi, epi_seq);
}

// ---------------------------------------------------------------------------
// The tick's TAIL, behind the record kernels, as ONE launch (a launch of this size costs ~5 us whatever it does; the three that
// were here — the filtering launch with the state commit over one workgroup per connection slot, the element walk likewise, the
// epilogue — took 16 us at config B for ~0 records, 23 us with exact update buffers):
//   * the subscriptions k_fanout_plan_seg left to the filtering streams (PF_DEFER; their connections come compacted in
//     WorldDev::defer_list from k_fanout_scan) — descriptor path only;
//   * the connections without room for their records (slots tail_ctl[TC_SCAP] ..: fanout_no_room) — descriptor path only, the
//     one-launch forms handle theirs in place;
//   * the subscriptions the tick-ring masks cannot answer (PF_DEEP, WorldDev::deep_list), from the exact update buffers;
//   * the tick epilogue, by the workgroup that finishes last (each working workgroup releases its writes at agent scope and takes
//     a ticket; workgroups without an item leave at once and take none).
// Single-wave workgroups stride over the items; the grid is what the host can afford without knowing the lists' lengths.
// ---------------------------------------------------------------------------
#define TC_TICKET 4
template <bool MASKS>
__global__ void __launch_bounds__(64) k_fanout_tail(DevGrid g, WorldDev w, int64_t now, TickRing ring, int seg, uint32_t slot, uint32_t ncell,
                                                    unsigned long long *epi, unsigned long long epi_seq) {
    const uint32_t nd = seg ? w.tail_ctl[TC_NDEFER] : 0u, scap = seg ? w.tail_ctl[TC_SCAP] : w.S;
    const uint32_t np = w.deep_depth ? w.tail_ctl[TC_NDEEP] : 0u;
    const uint32_t n1 = nd + (w.S - scap), total = n1 + np;
    const uint32_t workers = total < gridDim.x ? total : gridDim.x;
    if (blockIdx.x >= (workers ? workers : 1u)) return;
    for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
        if (i < nd) {
            const uint32_t s = w.defer_list[i];
            if (s < scap) fanout_emit_conn<1, MASKS, true>(g, w, now, ring, s);
        } else if (i < n1) {
            fanout_no_room(w, scap + (i - nd));
        } else {
            fanout_deep_conn(g, w, now, ring, w.deep_list[i - n1]);
        }
        __syncthreads();  // (the connection's LDS staging is reused)
    }
    if (workers > 1u) {
        __threadfence();
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(&w.tail_ctl[TC_TICKET], 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t != workers - 1u) return;
        __threadfence();
    }
    if (threadIdx.x == 0) w.tail_ctl[TC_TICKET] = 0;
    tick_epilogue_wave(w, slot, ncell, epi, epi_seq);
}

static uint32_t tail_grid() {  // (CHD_TAIL_GRID: A/B runs)
    static const uint32_t n = [] { const char *e = getenv("CHD_TAIL_GRID"); return e ? (uint32_t)std::max(atoi(e), 1) : 512u; }();
    return n;
}
void launch_fanout_tail(hipStream_t st, DevGrid g, WorldDev w, int64_t now_ns, TickRing ring, uint32_t slot, unsigned long long *epi, unsigned long long epi_seq) {
    if (!w.S) { launch_tick_epilogue(st, w, slot, g.ncell, epi, epi_seq); return; }
    const int seg = seg_path(w) ? 1 : 0;
    const dim3 grid(std::min(std::max(w.S, 1u), tail_grid()));
    if (w.rec_mask) hipLaunchKernelGGL((k_fanout_tail<true>), grid, dim3(64), 0, st, g, w, now_ns, ring, seg, slot, g.ncell, epi, epi_seq);
    else hipLaunchKernelGGL((k_fanout_tail<false>), grid, dim3(64), 0, st, g, w, now_ns, ring, seg, slot, g.ncell, epi, epi_seq);
}
