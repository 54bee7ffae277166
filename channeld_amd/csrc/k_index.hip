// k_index.hip — K2: the cell -> entity index, i.e. the per-cell entity maps
// the reference keeps in SpatialChannelData.Entities
// (pkg/unrealpb/extension.go:32-86), rebuilt every tick as a stable counting
// sort of the entity slots by member cell:
//
//   k_index_hist   : per-block histogram of member cells in LDS (one LDS
//                    atomic per entity), written cell-major to blk_cnt[c*nblk+b]
//   k_index_scan   : one wave per cell: blk_cnt[c*nblk+b] -> entities of c in blocks < b;
//                    the last workgroup scans the cell totals into cell_off
//   k_index_scatter: stable scatter — ranks inside a block come from wave
//                    ballots (peer masks by key bits), never from atomics, so the
//                    per-cell order is by entity slot and the output is
//                    deterministic.  Writes what the emit kernel streams: one
//                    16-byte entry per entity {entity channel id, update history of
//                    its sender aligned to this tick, sender, history of the previous
//                    sender} (+ the previous sender in a side array), and cell_off.
//
// Also here: the generic exclusive scans (single workgroup, 1024 lanes,
// wave-shuffle scan + LDS carry).
#include <algorithm>

#include "chd_kernels.h"

#define IDX_BLOCK 256
// entities per lane of the histogram / scatter workgroups.  2 rather than 4: at 100 K entities 196 workgroups
// instead of 98 on 256 CUs (index build 34 -> 27 us); the per-cell block-count rows grow accordingly.
#ifndef IDX_ITEMS
#define IDX_ITEMS 2
#endif
#define IDX_TILE (IDX_BLOCK * IDX_ITEMS)
#define IDX_MAX_LDS_CELLS 4096  // 4 waves x 4096 x 4 B = 64 KiB of dynamic LDS

uint32_t index_num_blocks(uint32_t N) { return (N + IDX_TILE - 1) / IDX_TILE; }

__device__ __forceinline__ void index_hist_block(const WorldDev &w, uint32_t ncell, uint32_t cur_tick, uint32_t bid, unsigned char *smem) {
    uint32_t *h = (uint32_t *)smem;
    // per cell also the range of sender ids whose updates are still buffered: a cell whose entities all
    // have ONE sender lets the emit kernel stream 8-byte entries and test the sender once per subscription
    // ... and the AND of the entities' update histories: a window that intersects it is hit by EVERY entity of the cell
    uint32_t *smin = h + ncell, *smax = h + 2 * ncell, *hand = h + 3 * ncell;
    for (uint32_t c = threadIdx.x; c < ncell; c += IDX_BLOCK) { h[c] = 0; smin[c] = 0xFFFFFFFFu; smax[c] = 0; hand[c] = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t base = bid * IDX_TILE;
    // Every word of the workgroup's entities is requested up front, unconditionally (an entity beyond N or dead reads slot 0 and
    // is masked afterwards): the launch is one workgroup per CU, as long as its chain of dependent round trips, and loads behind
    // per-lane tests — alive? in a cell? a previous sender? — were five such trips.  Only the update log by CHANNEL (region-sharded
    // exact worlds, log_ix) costs a second one.
    uint32_t ef[IDX_ITEMS], mem[IDX_ITEMS], snd[IDX_ITEMS], htk[IDX_ITEMS], hpv[IDX_ITEMS], hcu[IDX_ITEMS], spv[IDX_ITEMS], irr[IDX_ITEMS];
    const uint32_t *irr_p = w.deep_depth ? w.irr_tick : w.hist_tick;  // (no exact buffers: any readable word, unused)
    // (two copies of the loads, chosen by a uniform branch around ALL of them: a value merged behind a branch is awaited at the join)
    if (!w.log_on) {
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            // wave w handles the contiguous chunk [base + w*256 + r*64, +64)
            const uint32_t i = base + (threadIdx.x >> 6) * (IDX_ITEMS * 64) + r * 64 + (threadIdx.x & 63);
            const uint32_t u = i < w.N ? i : 0u;
            ef[r] = w.eflags[u]; mem[r] = w.member[u];
            snd[r] = w.sender[u]; htk[r] = w.hist_tick[u]; hpv[r] = w.hist_prev[u]; hcu[r] = w.hist[u]; spv[r] = w.sender_prev[u];
            irr[r] = irr_p[u];
            if (i >= w.N) ef[r] = 0u;
        }
    } else {
        uint32_t lu[IDX_ITEMS];
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            const uint32_t i = base + (threadIdx.x >> 6) * (IDX_ITEMS * 64) + r * 64 + (threadIdx.x & 63);
            const uint32_t u = i < w.N ? i : 0u;
            ef[r] = w.eflags[u]; mem[r] = w.member[u];
            lu[r] = w.chan_id[u] - w.log_eid0;
            if (i >= w.N) ef[r] = 0u;
        }
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            const uint32_t u = (ef[r] & EF_ALIVE) ? lu[r] : 0u;  // (a dead slot's channel id means nothing: log 0, masked)
            snd[r] = w.sender[u]; htk[r] = w.hist_tick[u]; hpv[r] = w.hist_prev[u]; hcu[r] = w.hist[u]; spv[r] = w.sender_prev[u];
            irr[r] = irr_p[u];
        }
    }
#pragma unroll
    for (int r = 0; r < IDX_ITEMS; r++) {
        const uint32_t m = (ef[r] & EF_ALIVE) ? mem[r] : CHD_INVALID;
        if (m < ncell) {
            atomicAdd(&h[m], 1u);
            atomicMin(&smin[m], snd[r]);
            atomicMax(&smax[m], snd[r]);
            const uint32_t age = cur_tick - htk[r];
            const uint32_t hp = age < CHD_HIST_BITS ? (hpv[r] << age) : 0u;
            const uint32_t hc = age < CHD_HIST_BITS ? (hcu[r] << age) : 0u;
            atomicAnd(&hand[m], hc | hp);
            // exact update buffers: an update the masks cannot represent, still inside their horizon (rare: a plain
            // global atomic; cell_irr is cleared by the tick epilogue)
            if (w.deep_depth && irr[r] && cur_tick + 1u - irr[r] < CHD_HIST_BITS) atomicOr(&w.cell_irr[m], 1u);
            if (hp != 0) {
                atomicMin(&smin[m], spv[r]);
                atomicMax(&smax[m], spv[r]);
            }
        }
    }
    __syncthreads();
    if (w.deep_depth && bid == 0) {  // ... and the spatial channels' own updates
        for (uint32_t c = threadIdx.x; c < ncell; c += IDX_BLOCK) {
            const uint32_t it = w.cell_irr_tick[c];
            if (it && cur_tick + 1u - it < CHD_HIST_BITS) atomicOr(&w.cell_irr[c], 1u);
        }
    }
    for (uint32_t c = threadIdx.x; c < ncell; c += IDX_BLOCK) {
        const size_t k = (size_t)c * w.nblk + bid;
        w.blk_cnt[k] = h[c];
        w.blk_smin[k] = smin[c];
        w.blk_smax[k] = smax[c];
        w.blk_hand[k] = hand[c];
    }
}

__global__ void __launch_bounds__(IDX_BLOCK) k_index_hist(WorldDev w, uint32_t ncell, uint32_t cur_tick) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    index_hist_block(w, ncell, cur_tick, blockIdx.x, smem);
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v);

// Second step: one wave per cell turns the cell's per-block counts into prefixes (entities of the
// cell in earlier blocks) and records the cell's total; the LAST workgroup to finish then scans the
// totals into cell_off (a release/acquire pair at device scope makes the other workgroups' totals
// visible across the XCDs' private L2s).  Replaces a single-workgroup scan over cells x blocks.
// one wave: cell c's per-block counts -> prefixes, its total and aggregates
__device__ __forceinline__ void index_scan_cell(const WorldDev &w, uint32_t ncell, int finalize, uint32_t c) {
    const uint32_t lane = threadIdx.x & 63u;
    if (c < ncell) {
        uint32_t *row = w.blk_cnt + (size_t)c * w.nblk;
        uint32_t carry = 0, lo = 0xFFFFFFFFu, hi = 0, ha = 0xFFFFFFFFu;
        for (uint32_t b0 = 0; b0 < w.nblk; b0 += 64) {
            const uint32_t b = b0 + lane;
            const uint32_t v = b < w.nblk ? row[b] : 0u;
            const uint32_t inc = wave_incl_scan(v);
            if (b < w.nblk) {
                row[b] = carry + inc - v;
                lo = min(lo, w.blk_smin[(size_t)c * w.nblk + b]);
                hi = max(hi, w.blk_smax[(size_t)c * w.nblk + b]);
                ha &= w.blk_hand[(size_t)c * w.nblk + b];
            }
            carry += __shfl(inc, 63);
        }
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
            ha &= (uint32_t)__shfl_xor((int)ha, d);
        }
        if (lane == 0) {
            (finalize ? w.cell_off : w.cell_tot)[c] = carry;  // the cell's total
            w.cell_usender[c] = carry == 0 ? 0u : (lo == hi ? lo : CHD_NONUNIFORM);
            w.cell_smin[c] = lo;  // (empty cell: [0xFFFFFFFF, 0], no connection is inside)
            w.cell_smax[c] = hi;
            w.cell_hand[c] = carry == 0 ? 0u : ha;
        }
    }
}

__global__ void __launch_bounds__(256) k_index_scan(WorldDev w, uint32_t ncell, int finalize) {
    __shared__ uint32_t is_last;
    __shared__ uint32_t part[256];
    index_scan_cell(w, ncell, finalize, blockIdx.x * 4u + (threadIdx.x >> 6));
    // small grids: the scatter workgroups scan the few cell totals themselves (no cross-workgroup step)
    if (!finalize) return;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&w.counters[CTR_SCAN_DONE], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // exclusive scan of the ncell totals (<= 4096): contiguous chunk per thread, then a 256-wide scan
    const uint32_t per = (ncell + 255u) / 256u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, ncell);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += __atomic_load_n(&w.cell_off[i], __ATOMIC_RELAXED);
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < 256; i++) { const uint32_t v = part[i]; part[i] = run; run += v; }
        w.cell_off[ncell] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = w.cell_off[i];
        w.cell_off[i] = run;
        run += v;
    }
}


__device__ __forceinline__ void index_scatter_block(const WorldDev &w, uint32_t ncell, uint32_t key_bits, uint32_t cur_tick, int local_base,
                                                    uint32_t bid, unsigned char *smem) {
    uint32_t *wcnt = (uint32_t *)smem;  // [4][ncell] running per-wave counters
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t c = threadIdx.x; c < 4 * ncell; c += IDX_BLOCK) wcnt[c] = 0;
    uint32_t *mycnt = wcnt + wave * ncell;
    uint32_t base = bid * IDX_TILE + wave * (IDX_ITEMS * 64);
    // EVERY global word the workgroup needs is requested here, unconditionally (an entity beyond N reads slot 0 and is masked; a dead
    // one's words are read and dropped): its entities' flags, cells and entry words, and the first round of the per-cell bases.  The
    // launch is one workgroup per CU and as long as its chain of dependent round trips; with the loads behind the "alive / in a
    // cell" tests and the bases behind two barriers that chain was five trips long.  Only the update log by CHANNEL (region-sharded
    // exact worlds) keeps a second trip.
    uint32_t ef[IDX_ITEMS], mem[IDX_ITEMS], chn[IDX_ITEMS], htk[IDX_ITEMS], hcu[IDX_ITEMS], hpv[IDX_ITEMS], snd[IDX_ITEMS], spv[IDX_ITEMS], lix[IDX_ITEMS];
    uint4 oa[IDX_ITEMS], ob[IDX_ITEMS];
    const uint4 *eoff_p = w.off_on ? w.eoff : (const uint4 *)(const void *)w.ce;  // (no offsets: any readable 32 bytes, unused)
    const uint32_t eoff_m = w.off_on ? 0xFFFFFFFFu : 0u;
    if (!w.log_on) {
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            const uint32_t i = base + r * 64 + lane, u = i < w.N ? i : 0u;
            ef[r] = w.eflags[u]; mem[r] = w.member[u]; chn[r] = w.chan_id[u];
            htk[r] = w.hist_tick[u]; hcu[r] = w.hist[u]; hpv[r] = w.hist_prev[u]; snd[r] = w.sender[u]; spv[r] = w.sender_prev[u];
            oa[r] = eoff_p[2 * (size_t)(u & eoff_m)]; ob[r] = eoff_p[2 * (size_t)(u & eoff_m) + 1];
            lix[r] = u;
            if (i >= w.N) ef[r] = 0u;
        }
    } else {
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            const uint32_t i = base + r * 64 + lane, u = i < w.N ? i : 0u;
            ef[r] = w.eflags[u]; mem[r] = w.member[u]; chn[r] = w.chan_id[u];
            if (i >= w.N) ef[r] = 0u;
        }
#pragma unroll
        for (int r = 0; r < IDX_ITEMS; r++) {
            const uint32_t u = (ef[r] & EF_ALIVE) ? chn[r] - w.log_eid0 : 0u;  // (a dead slot's channel id means nothing: log 0, masked)
            htk[r] = w.hist_tick[u]; hcu[r] = w.hist[u]; hpv[r] = w.hist_prev[u]; snd[r] = w.sender[u]; spv[r] = w.sender_prev[u];
            oa[r] = eoff_p[2 * (size_t)(u & eoff_m)]; ob[r] = eoff_p[2 * (size_t)(u & eoff_m) + 1];
            lix[r] = u;
        }
    }
    // ... the cell bases of this thread's first cell (grids of up to 256 cells: its only one)
    const uint32_t c0 = threadIdx.x < ncell ? threadIdx.x : 0u;
    const uint32_t pre_blk = w.blk_cnt[(size_t)c0 * w.nblk + bid];
    const uint32_t pre_base = local_base ? w.cell_tot[c0] : w.cell_off[c0];
    __syncthreads();
    uint32_t key[IDX_ITEMS], lrank[IDX_ITEMS];
#pragma unroll
    for (int r = 0; r < IDX_ITEMS; r++) {
        uint32_t m = CHD_INVALID;
        if (ef[r] & EF_ALIVE) m = mem[r];
        bool valid = m < ncell;
        key[r] = valid ? m : CHD_INVALID;
        // peers = lanes of this wave holding the same key
        uint64_t peers = __ballot(valid);
        for (uint32_t b = 0; b < key_bits; b++) {
            uint64_t bm = __ballot(valid && ((m >> b) & 1u));
            peers &= ((m >> b) & 1u) ? bm : ~bm;
        }
        uint32_t pre = 0;
        if (valid) {
            uint32_t rk = mask_rank(peers);
            int leader = __ffsll((unsigned long long)peers) - 1;
            if ((int)lane == leader) {
                pre = mycnt[m];
                mycnt[m] = pre + (uint32_t)__popcll(peers);
            }
            pre = __shfl(pre, leader);
            lrank[r] = pre + rk;
        } else {
            lrank[r] = 0;
        }
    }
    __syncthreads();
    // cell bases: either final in cell_off (k_index_scan's last workgroup) or scanned here from the
    // cell totals (ncell <= 1024: four cells per lane, one wave scan, 256 partials)
    uint32_t *cbase = wcnt + 4 * ncell;
    if (local_base) {
        __shared__ uint32_t part[IDX_BLOCK];
        const uint32_t per = (ncell + IDX_BLOCK - 1) / IDX_BLOCK;
        const uint32_t lo = threadIdx.x * per, hi = min(lo + per, ncell);
        uint32_t sum = 0;
        if (per == 1u) sum = lo < hi ? pre_base : 0u;  // (lo == threadIdx.x == c0)
        else for (uint32_t i = lo; i < hi; i++) sum += w.cell_tot[i];
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x < 64) {  // scan of the 256 partials by one wave, 4 per lane
            uint32_t v[4], t = 0;
            for (int k = 0; k < 4; k++) { v[k] = part[threadIdx.x * 4 + k]; t += v[k]; }
            uint32_t inc = wave_incl_scan(t);
            uint32_t run = inc - t;
            for (int k = 0; k < 4; k++) { part[threadIdx.x * 4 + k] = run; run += v[k]; }
        }
        __syncthreads();
        uint32_t run = part[threadIdx.x];
        for (uint32_t i = lo; i < hi; i++) {
            cbase[i] = run;
            run += per == 1u ? pre_base : w.cell_tot[i];
        }
        if (bid == 0) {  // publish the CSR offsets for the fan-out kernels
            uint32_t r2 = part[threadIdx.x];
            for (uint32_t i = lo; i < hi; i++) { w.cell_off[i] = r2; r2 += per == 1u ? pre_base : w.cell_tot[i]; }
            if (hi == ncell && lo < hi) w.cell_off[ncell] = r2;
        }
        __syncthreads();
    }
    // exclusive prefix over the 4 waves, in place: wcnt[w][c] -> entities of c in waves < w
    for (uint32_t c = threadIdx.x; c < ncell; c += IDX_BLOCK) {
        uint32_t a0 = wcnt[c], a1 = wcnt[ncell + c], a2 = wcnt[2 * ncell + c];
        const bool first = c == threadIdx.x;  // (the prefetched round)
        uint32_t g0 = (local_base ? cbase[c] : (first ? pre_base : w.cell_off[c])) + (first ? pre_blk : w.blk_cnt[(size_t)c * w.nblk + bid]);  // cell base + entities of c in earlier blocks
        wcnt[c] = g0;
        wcnt[ncell + c] = g0 + a0;
        wcnt[2 * ncell + c] = g0 + a0 + a1;
        wcnt[3 * ncell + c] = g0 + a0 + a1 + a2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IDX_ITEMS; r++) {
        if (key[r] == CHD_INVALID) continue;
        uint32_t pos = mycnt[key[r]] + lrank[r];
        uint32_t age = cur_tick - htk[r];
        uint32_t h = (age >= CHD_HIST_BITS) ? 0u : (hcu[r] << age);
        uint32_t hp = (age >= CHD_HIST_BITS) ? 0u : (hpv[r] << age);
        w.ce[pos] = make_uint4(chn[r], h, snd[r], hp);
        w.ce8[pos] = make_uint2(chn[r], h | hp);
        w.ce_chan[pos] = chn[r];
        w.ce_sprev[pos] = spv[r];
        if (w.ce_slot) w.ce_slot[pos] = w.ce_by_chan ? chn[r] - w.log_eid0 : base + r * 64 + lane;  // (what the exact buffers / wire payloads are indexed by: the slot — or the channel)
        if (w.off_on) {
            uint32_t o[CHD_OFF_SLOTS] = {oa[r].x, oa[r].y, oa[r].z, oa[r].w, ob[r].x, ob[r].y, ob[r].z, ob[r].w};
            off_shift(o, age);
#pragma unroll
            for (uint32_t j = 0; j < CHD_OFF_SLOTS; j++) w.ce_off[(size_t)j * w.off_stride + pos] = o[j];
        }
    }
}

// gate_p != nullptr (CHD_WORLD_GATED_OVERLAP): the launch also holds the tick's stream until *gate_p >= gate_target — the
// interest updates on the second stream are complete (GateArgs) — by keeping workgroup 0 alive: what follows on the stream (the
// fan-out plan) needs both the index and the subscriptions, and a separate one-wave wait kernel costs ~5 us of launch.
__global__ void __launch_bounds__(IDX_BLOCK) k_index_scatter(WorldDev w, uint32_t ncell, uint32_t key_bits,
                                                             uint32_t cur_tick, int local_base, const unsigned long long *gate_p,
                                                             unsigned long long gate_target) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    index_scatter_block(w, ncell, key_bits, cur_tick, local_base, blockIdx.x, smem);
    if (gate_p && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t spins = 0;
        while (__hip_atomic_load(gate_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 23)) { gate_timed_out(w); break; }  // (a bug, never a capacity)
        }
    }
}

// fallback for grids too large for the LDS counters: global atomics, the order
// inside a cell is then not deterministic (documented in DESIGN.md).
__global__ void __launch_bounds__(256) k_index_hist_global(WorldDev w, uint32_t ncell, uint32_t cur_tick) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (w.deep_depth && i < ncell) {
        const uint32_t it = w.cell_irr_tick[i];
        if (it && cur_tick + 1u - it < CHD_HIST_BITS) atomicOr(&w.cell_irr[i], 1u);
    }
    if (i >= w.N) return;
    if (!(w.eflags[i] & EF_ALIVE)) return;
    uint32_t m = w.member[i];
    if (m < ncell) {
        atomicAdd(&w.blk_cnt[m], 1u);
        if (w.deep_depth) {
            const uint32_t it = w.irr_tick[log_ix(w, i)];
            if (it && cur_tick + 1u - it < CHD_HIST_BITS) atomicOr(&w.cell_irr[m], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) k_index_scatter_global(WorldDev w, uint32_t ncell, uint32_t *cursor,
                                                              uint32_t cur_tick) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w.N) return;
    if (!(w.eflags[i] & EF_ALIVE)) return;
    uint32_t m = w.member[i];
    if (m >= ncell) return;
    uint32_t pos = w.blk_cnt[m] + atomicAdd(&cursor[m], 1u);
    const uint32_t u = log_ix(w, i);
    uint32_t age = cur_tick - w.hist_tick[u];
    uint32_t h = (age >= CHD_HIST_BITS) ? 0u : (w.hist[u] << age);
    uint32_t hp = (age >= CHD_HIST_BITS) ? 0u : (w.hist_prev[u] << age);
    w.ce[pos] = make_uint4(w.chan_id[i], h, w.sender[u], hp);
    w.ce_sprev[pos] = w.sender_prev[u];
    w.ce_chan[pos] = w.chan_id[i];
    if (w.ce_slot) w.ce_slot[pos] = ce_ix(w, i);
    if (w.off_on) scatter_offsets(w, u, pos, age);
}

// ------------------------------------------------------------------------
// exclusive scans: one workgroup of 1024 lanes walks the array in 1024-element
// tiles; in-wave inclusive scan by DPP-style shuffles, 16 wave totals combined
// through LDS.  out[n] receives the grand total.
// ------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}

// elements per lane and pass.  (16 - one pass for the 10 K per-connection totals instead of three - measured slower:
// 11.5 vs 6.9 us, a lane's 16 consecutive u64 are a 128-byte stride between lanes.)
#define SCAN_ITEMS 4
template <typename T>
__global__ void __launch_bounds__(1024) k_scan_excl(const T *in, T *out, uint32_t n, const uint32_t *n_dev) {
    if (n_dev) n = min(n, *n_dev);  // length known only on the device
    __shared__ T wtot[16];
    __shared__ T carry_s;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    // each lane owns SCAN_ITEMS consecutive elements of a (1024 x SCAN_ITEMS)-element tile
    for (uint32_t base = 0; base < n; base += 1024 * SCAN_ITEMS) {
        const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
        T v[SCAN_ITEMS];
        T sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = (i0 + k < n) ? in[i0 + k] : (T)0;
            sum += v[k];
        }
        T inc = wave_incl_scan(sum);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        T carry = carry_s;
        T woff = 0;
        for (uint32_t k = 0; k < wave; k++) woff += wtot[k];
        T run = carry + woff + inc - sum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

void launch_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_excl<uint32_t>), dim3(1), dim3(1024), 0, st, in, out, n, (const uint32_t *)nullptr);
}
void launch_scan_u32_inplace(hipStream_t st, uint32_t *data, uint32_t n) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_excl<uint32_t>), dim3(1), dim3(1024), 0, st, data, data, n, (const uint32_t *)nullptr);
}
void launch_scan_u32_inplace_dev(hipStream_t st, uint32_t *data, uint32_t n_max, const uint32_t *n_dev) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_excl<uint32_t>), dim3(1), dim3(1024), 0, st, data, data, n_max, n_dev);
}
void launch_scan_u64_inplace(hipStream_t st, uint64_t *data, uint32_t n) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_excl<uint64_t>), dim3(1), dim3(1024), 0, st, data, data, n, (const uint32_t *)nullptr);
}

// Window columns (chd_kernels.h: WorldDev::wcol_*).  One workgroup per cell; for every window mask wcol_mask(j) that the
// cell's entities do not ALL cover (cell_hand: then the full column serves), the channel ids of the entities with an update
// inside it, compacted in entry order (ballot ranks inside a wave, wave totals through LDS).
__global__ void __launch_bounds__(256) k_window_columns(WorldDev w, uint32_t ncell) {
    __shared__ uint32_t wtot[CHD_WCOLS][4];
    const uint32_t c = blockIdx.x;
    const uint32_t start = w.cell_off[c], n = w.cell_off[c + 1] - start;
    const uint32_t hand = w.cell_hand[c];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t run[CHD_WCOLS];
    bool build[CHD_WCOLS];
    bool any = false;
#pragma unroll
    for (int j = 0; j < CHD_WCOLS; j++) { run[j] = 0; build[j] = n != 0 && !(hand & wcol_mask(j)); any = any || build[j]; }
    if (!any) {
        if (threadIdx.x < CHD_WCOLS) w.cell_wcnt[threadIdx.x * ncell + c] = n;
        return;
    }
    for (uint32_t b = 0; b < n; b += 256) {
        const uint32_t k = b + threadIdx.x;
        uint2 e = make_uint2(0u, 0u);
        if (k < n) e = w.ce8[start + k];  // {channel, history of any sender, aligned to this tick}
        uint64_t m[CHD_WCOLS];
#pragma unroll
        for (int j = 0; j < CHD_WCOLS; j++) {
            m[j] = __ballot(k < n && (e.y & wcol_mask(j)) != 0);
            if (lane == 0) wtot[j][wave] = (uint32_t)__popcll(m[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CHD_WCOLS; j++) {
            uint32_t before = 0, all = 0;
            for (uint32_t q = 0; q < 4; q++) { const uint32_t t = wtot[j][q]; if (q < wave) before += t; all += t; }
            if (build[j] && k < n && (e.y & wcol_mask(j)) != 0) {
                const size_t at = (size_t)(j + 1) * w.wcol_stride + start + run[j] + before + mask_rank(m[j]);
                w.ce_chan[at] = e.x;
                if (w.wcol_slot) w.wcol_slot[at] = w.ce_slot[start + k];  // (wire worlds: whose payload the column entry stands for)
            }
            run[j] += all;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < CHD_WCOLS; j++)  // (constant indices: the counters stay in registers)
        if (threadIdx.x == (uint32_t)j) w.cell_wcnt[(size_t)j * ncell + c] = build[j] ? run[j] : n;
}

// ---------------------------------------------------------------------------
// k_cell_arrange: worlds that keep sub-tick arrival offsets (off_on), one workgroup per cell over its cell-sorted entries, behind
// the index build (region-sharded: behind the halo unpack, so the neighbours' ghost entries as well).  Two products:
//  (1) per ring slot j < CHD_OFF_SLOTS the {min, max} sub-tick offset over the cell's entities that hold an update of that slot
//      (WorldDev::cell_orng; min > max: none does): what the plan decides "every update of the slot inside / outside this window" from;
//  (2) cells of up to ARR_MAX entries (the filtered record kernel's LDS tile): the entries IN THE ORDER OF THIS TICK'S ARRIVALS —
//      ascending offset of ring slot 0, entities without an update in this tick last (cell_sorted[c] = 1).  The order inside a cell is
//      free (a connection's records are a multiset), and in this one the part of a fan-out window that lies inside the tick's own
//      arrivals — every window of a subscription whose interval is shorter than the tick, one end of every window whose phase is off
//      the tick grid — selects a CONTIGUOUS RUN of the cell's column: k_fanout_emit_filt_cm finds the run's two ends by a binary search
//      over the staged offsets and copies it, instead of testing every entity of the cell against the window.
// One load round trip (every per-entry array of the cell into registers), a counting sort in LDS — 64 buckets over the span of the
// cell's own slot-0 offsets (their {min, max} is product (1)), the rank inside a bucket by comparing with the bucket's other entries
// (offset, then position: deterministic) —, then every array written back permuted, in place (every thread's loads have returned
// before the barrier in front of the first store).  A cell whose arrivals crowd into one bucket (more than ARR_BKT_MAX entries: the
// rank loop would be quadratic) stays as it is, cell_sorted[c] = 0: its windows are tested per entity, which is always correct.
// ---------------------------------------------------------------------------
#define ARR_MAX 512u
#define ARR_BKT 64u
#define ARR_BKT_MAX 48u
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
// min / max over the wave: inside every row of 16 lanes by DPP (quad permutes, then the mirrors), across the four rows on the scalar unit
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_mov<0xB1>(v)); v = min(v, dpp_mov<0x4E>(v)); v = min(v, dpp_mov<0x141>(v)); v = min(v, dpp_mov<0x140>(v));
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_mov<0xB1>(v)); v = max(v, dpp_mov<0x4E>(v)); v = max(v, dpp_mov<0x141>(v)); v = max(v, dpp_mov<0x140>(v));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// LDS staging of one cell: every per-entry word as a column (the counting sort's scratch shares nothing with it)
struct ArrStage {
    uint32_t ce[4][ARR_MAX];              // the 16-byte entry {channel, history, sender, previous sender's history}
    uint32_t chan8[ARR_MAX], hist8[ARR_MAX];  // the compact entry {channel, history of any sender}
    uint32_t chan[ARR_MAX], sprev[ARR_MAX], slot[ARR_MAX];
    uint32_t off[CHD_OFF_SLOTS][ARR_MAX];
};
__global__ void __launch_bounds__(256) k_cell_arrange(WorldDev w, uint32_t ncell, int sort) {
    __shared__ ArrStage L;
    __shared__ uint32_t smn[4][CHD_OFF_SLOTS], smx[4][CHD_OFF_SLOTS];
    __shared__ uint32_t bcnt[ARR_BKT + 1], boff[ARR_BKT + 2], skey[ARR_MAX], wcnt[4], crowded;
    __shared__ unsigned short sidx[ARR_MAX], perm[ARR_MAX];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    const uint32_t start = w.cell_start[c], n = w.cell_end[c] - start;  // (region-sharded: the neighbours' ghost entries as well)
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const bool small = n <= ARR_MAX, arrange = small && sort && w.cell_sorted != nullptr;
    // PHASE A: the cell's words into LDS (small cells; plain loops of unconditional loads: the whole launch is as long as its slowest
    // workgroup, and loads behind per-lane tests — or a large unrolled body kept live in registers — made that 18-28 us), and per
    // thread the {min, max} offset of every ring slot over its entries
    uint32_t mn[CHD_OFF_SLOTS], mx[CHD_OFF_SLOTS];
#pragma unroll
    for (uint32_t j = 0; j < CHD_OFF_SLOTS; j++) { mn[j] = 0xFFFFFFFFu; mx[j] = 0u; }
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t s = start + i;
        const uint32_t h = w.ce8_view[s].y;  // history of any sender, aligned to this tick
        uint32_t o[CHD_OFF_SLOTS];
#pragma unroll
        for (uint32_t q = 0; q < CHD_OFF_SLOTS; q++) o[q] = w.ce_off[(size_t)q * w.off_stride + s];
#pragma unroll
        for (uint32_t q = 0; q < CHD_OFF_SLOTS; q++) {
            const bool has = (h >> q) & 1u;
            mn[q] = min(mn[q], has ? o[q] : 0xFFFFFFFFu);
            mx[q] = max(mx[q], has ? o[q] : 0u);
            if (arrange) L.off[q][i] = o[q];
        }
        if (arrange) L.hist8[i] = h;
    }
    if (arrange) {
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t s = start + i;
            const uint4 e = w.ce_view[s];
            L.ce[0][i] = e.x; L.ce[1][i] = e.y; L.ce[2][i] = e.z; L.ce[3][i] = e.w;
            L.chan8[i] = w.ce8_view[s].x; L.chan[i] = w.ce_chan_view[s]; L.sprev[i] = w.ce_sprev_view[s];
        }
        if (w.ce_slot) for (uint32_t i = tid; i < n; i += 256) L.slot[i] = w.ce_slot[start + i];
    }
#pragma unroll
    for (uint32_t j = 0; j < CHD_OFF_SLOTS; j++) {
        const uint32_t a = wave_min_u32(mn[j]), b = wave_max_u32(mx[j]);
        if (lane == 0) { smn[wave][j] = a; smx[wave][j] = b; }
    }
    if (tid <= ARR_BKT) bcnt[tid] = 0;
    if (tid == 0) crowded = 0;
    __syncthreads();
    if (tid < CHD_OFF_SLOTS) {
        const uint32_t j = tid;
        const uint32_t a = min(min(smn[0][j], smn[1][j]), min(smn[2][j], smn[3][j]));
        const uint32_t b = max(max(smx[0][j], smx[1][j]), max(smx[2][j], smx[3][j]));
        w.cell_orng[(size_t)c * CHD_OFF_SLOTS + j] = make_uint2(a, b);
    }
    if (!w.cell_sorted) return;
    const uint32_t mn0 = min(min(smn[0][0], smn[1][0]), min(smn[2][0], smn[3][0])), mx0 = max(max(smx[0][0], smx[1][0]), max(smx[2][0], smx[3][0]));
    if (!arrange) { if (tid == 0) w.cell_sorted[c] = 0; return; }  // (not asked for / a cell beyond the tile)
    if (n == 0 || mn0 > mx0) { if (tid == 0) w.cell_sorted[c] = 1; return; }  // (nothing to order: no entry holds an update of this tick)
    // PHASE B: counting sort on the slot-0 offset.  bucket = (offset - min) >> sh with the span of the cell's own offsets over the 64
    // buckets; no update in this tick: bucket 64
    uint32_t sh = 0;
    { const uint32_t span = mx0 - mn0; while (sh < 31u && (span >> sh) >= ARR_BKT) sh++; }
    uint32_t key[2], bk[2], lp[2];
    uint64_t inv_m[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t i = tid + 256u * r;
        const bool in = i < n, has = in && (L.hist8[in ? i : 0u] & 1u);
        key[r] = has ? L.off[0][i] : 0xFFFFFFFFu;
        bk[r] = has ? min((key[r] - mn0) >> sh, ARR_BKT - 1u) : ARR_BKT;
        lp[r] = 0;
        if (has) lp[r] = atomicAdd(&bcnt[bk[r]], 1u);
        // (the entities without an update keep their relative order: position among them by ballots, no atomics)
        inv_m[r] = __ballot(in && !has);
    }
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(inv_m[0]) | ((uint32_t)__popcll(inv_m[1]) << 16);
    __syncthreads();
    if (tid < 64) {  // exclusive prefix over the 64 buckets (one wave), the "no update" bucket behind them
        const uint32_t v = bcnt[tid];
        if (v > ARR_BKT_MAX) crowded = 1;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
            if ((int)lane >= d) inc += o;
        }
        boff[tid] = inc - v;
        if (tid == 63) { boff[64] = inc; boff[65] = n; }
    }
    __syncthreads();
    if (crowded) { if (tid == 0) w.cell_sorted[c] = 0; return; }  // (uniform: the arrays have not been touched)
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t i = tid + 256u * r;
        if (i < n && bk[r] < ARR_BKT) { skey[boff[bk[r]] + lp[r]] = key[r]; sidx[boff[bk[r]] + lp[r]] = (unsigned short)i; }
    }
    __syncthreads();
    {
        // invalid entries before this one: round 0's of lower waves and lanes; round 1's come behind all of round 0's
        uint32_t w0 = 0, w1 = 0, a0 = 0;
        for (uint32_t q = 0; q < 4; q++) { const uint32_t t = wcnt[q]; a0 += t & 0xFFFFu; if (q < wave) { w0 += t & 0xFFFFu; w1 += t >> 16; } }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t i = tid + 256u * r;
            if (i >= n) continue;
            uint32_t rank;
            if (bk[r] == ARR_BKT) {
                rank = boff[64] + (r == 0 ? w0 : a0 + w1) + mask_rank(inv_m[r]);
            } else {
                uint32_t before = 0;
                for (uint32_t k = boff[bk[r]]; k < boff[bk[r] + 1u]; k++) {
                    const uint32_t kk = skey[k], ki = sidx[k];
                    before += (kk < key[r] || (kk == key[r] && ki < i)) ? 1u : 0u;
                }
                rank = boff[bk[r]] + before;
            }
            perm[rank] = (unsigned short)i;
        }
    }
    __syncthreads();
    // PHASE C: every array written back in the new order, destination by destination (whole lines per store instruction)
    if (tid == 0) w.cell_sorted[c] = 1;
    uint4 *ce = (uint4 *)w.ce_view; uint2 *ce8 = (uint2 *)w.ce8_view; uint32_t *cc = (uint32_t *)w.ce_chan_view, *cs = (uint32_t *)w.ce_sprev_view;
    for (uint32_t d = tid; d < n; d += 256) {
        const uint32_t i = perm[d], s = start + d;
        ce[s] = make_uint4(L.ce[0][i], L.ce[1][i], L.ce[2][i], L.ce[3][i]);
        ce8[s] = make_uint2(L.chan8[i], L.hist8[i]);
        cc[s] = L.chan[i]; cs[s] = L.sprev[i];
#pragma unroll
        for (uint32_t q = 0; q < CHD_OFF_SLOTS; q++) w.ce_off[(size_t)q * w.off_stride + s] = L.off[q][i];
    }
    if (w.ce_slot) for (uint32_t d = tid; d < n; d += 256) w.ce_slot[start + d] = L.slot[perm[d]];
}

void launch_cell_offsets(hipStream_t st, DevGrid g, WorldDev w) {
    if (!w.N || !w.off_on) return;
    static const int sort = [] { const char *e = getenv("CHD_SORT_ARRIVALS"); return (e && e[0] == '0') ? 0 : 1; }();  // (A/B runs)
    hipLaunchKernelGGL(k_cell_arrange, dim3(g.ncell), dim3(256), 0, st, w, g.ncell, sort);
}

void launch_window_columns(hipStream_t st, DevGrid g, WorldDev w) {
    if (!w.N || !w.wcol_stride) return;
    hipLaunchKernelGGL(k_window_columns, dim3(g.ncell), dim3(256), 0, st, w, g.ncell);
}

static uint32_t bits_for(uint32_t ncell) {
    uint32_t b = 0;
    while ((1u << b) < ncell) b++;
    return b ? b : 1;
}

// returns true when the launch took the gate wait with it (k_index_scatter)
bool launch_index_build(hipStream_t st, DevGrid g, WorldDev w, uint32_t cur_tick, const unsigned long long *gate_p, unsigned long long gate_target, int64_t now_ns) {
    if (!w.N) return false;
    if (g.ncell <= IDX_MAX_LDS_CELLS) {
        hipLaunchKernelGGL(k_index_hist, dim3(w.nblk), dim3(IDX_BLOCK), 4 * g.ncell * 4, st, w, g.ncell, cur_tick);
        const int local_base = g.ncell <= 1024;
        hipLaunchKernelGGL(k_index_scan, dim3((g.ncell + 3) / 4), dim3(256), 0, st, w, g.ncell, !local_base);
        hipLaunchKernelGGL(k_index_scatter, dim3(w.nblk), dim3(IDX_BLOCK), (local_base ? 5 : 4) * g.ncell * 4, st, w,
                           g.ncell, bits_for(g.ncell), cur_tick, local_base, gate_p, gate_target);
        return gate_p != nullptr;
    } else {
        // nblk == 1 layout: blk_cnt[c] then scan -> cell_off; cursor lives behind it
        uint32_t *cursor = w.blk_cnt + (size_t)g.ncell + 1;
        (void)hipMemsetAsync(w.blk_cnt, 0, sizeof(uint32_t) * (2 * (size_t)g.ncell + 2), st);
        (void)hipMemsetAsync(w.cell_usender, 0xFF, sizeof(uint32_t) * (size_t)g.ncell, st);  // CHD_NONUNIFORM
        (void)hipMemsetAsync(w.cell_smin, 0, sizeof(uint32_t) * (size_t)g.ncell, st);      // sender range unknown: everything
        (void)hipMemsetAsync(w.cell_smax, 0xFF, sizeof(uint32_t) * (size_t)g.ncell, st);
        (void)hipMemsetAsync(w.cell_hand, 0, sizeof(uint32_t) * (size_t)g.ncell, st);
        hipLaunchKernelGGL(k_index_hist_global, dim3((std::max(w.N, g.ncell) + 255) / 256), dim3(256), 0, st, w, g.ncell, cur_tick);
        launch_scan_u32_inplace(st, w.blk_cnt, g.ncell);
        (void)hipMemcpyAsync(w.cell_off, w.blk_cnt, sizeof(uint32_t) * ((size_t)g.ncell + 1), hipMemcpyDeviceToDevice, st);
        hipLaunchKernelGGL(k_index_scatter_global, dim3((w.N + 255) / 256), dim3(256), 0, st, w, g.ncell, cursor,
                           cur_tick);
    }
    return false;
}
