// k_front.hip — the FRONT of a tick in one launch: entity ingest -> cell index (histogram, per-cell scan, scatter) beside
// the interest updates.
//
// The two halves are independent (the ingest and the index touch entities and cell tables, the interest updates touch
// subscriptions; the fan-out plan behind them needs both), but as separate launches on one stream they run one after the
// other: ingest 11 + hist 8 + scan 6 + scatter 12 us of launch-latency-bound kernels, then 40 us of VALU-bound interest
// updates.  On two streams the two cross-stream dependencies cost most of what the overlap saves (measured: -8 of a possible
// -37 us).  Here ONE grid holds both: workgroups [0, NB) are the index roles — the four phases of the chain separated by
// grid barriers among exactly those NB workgroups (NB = index tiles <= FRONT_MAX_INDEX_BLOCKS: they are dispatched first and
// are all resident long before the first of them reaches a barrier) — and the workgroups behind them take four interest
// updates each.  The launch lasts as long as its longer half.
//
// This translation unit is a unity build of the three kernel files whose bodies it fuses.
#include "k_spatial.hip"
#include "k_index.hip"
#include "k_aoi.hip"

#define FRONT_MAX_INDEX_BLOCKS 512u
#define FRONT_QUERIES_PER_BLOCK 4u

// Barrier among the first `nb` workgroups of the grid, XCD-hierarchical (MI355X_MICROARCH.md, price list: barrier-xcd ~4-5 us
// against 9-17 us for one flat counter with two __threadfence()): workgroup b belongs to group b % 8 — the workgroups of a
// dispatch are dealt round-robin over the 8 XCDs, so a group shares ONE XCD (which one depends on where the queue's previous
// dispatch stopped; tools/ubench/xcc_map.hip) — ; the LAST arriver of a group does ONE agent-scope release for it (buffer_wbl2
// writes back the whole XCD's L2), arrives on the top counter, waits for the other groups, acquires, and opens the group's
// generation; everybody else polls that generation (relaxed sc1 loads) and then does its own agent-scope acquire.  (Round 3
// read HW_REG_XCC_ID here to check the placement: that s_getreg costs ~20 us per wave under load and compared against the
// wrong thing — the offset of the round-robin is not 0 — so every workgroup flushed its L2 at every barrier; removed in round
// 4.)  Counters only grow (`seq` = 1, 2, 3, ... over all barriers of all launches): no reset pass.  Spins are bounded:
// if the other workgroups never arrive (they cannot all be resident: a bug in the launch condition) the tick is flagged
// (overflow bit 0x8000) and goes on with whatever is there rather than hanging the GPU.
struct FrontBar {  // all fields on their own 128-byte lines
    unsigned long long cnt[8][16], gen[8][16], top[16];
};

__device__ __forceinline__ bool front_spin(const WorldDev &w, const unsigned long long *p, unsigned long long target) {
    uint32_t spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { atomicOr(&w.counters[CTR_OVERFLOW], OVF_INTERNAL); return false; }
    }
    return true;
}

__device__ __forceinline__ void front_barrier(const WorldDev &w, FrontBar *fb, unsigned long long seq, uint32_t nb) {
    __syncthreads();  // (every wave's stores have left the CU: s_waitcnt vmcnt(0) precedes the s_barrier)
    if (threadIdx.x == 0) {
        const uint32_t grp = blockIdx.x & 7u;
        const uint32_t members = (nb - grp + 7u) / 8u, ngroups = nb < 8u ? nb : 8u;
        const unsigned long long arrived = __hip_atomic_fetch_add(&fb->cnt[grp][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        if (arrived == (unsigned long long)members * seq) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2)
            __hip_atomic_fetch_add(&fb->top[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)front_spin(w, &fb->top[0], (unsigned long long)ngroups * seq);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&fb->gen[grp][0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            (void)front_spin(w, &fb->gen[grp][0], seq);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_front(DevGrid g, AoiLimits lim, WorldDev w, uint32_t nb, unsigned long long bar_seq,
                                               uint32_t n_upd, const uint32_t *__restrict__ upd_idx, const double *__restrict__ upd_x,
                                               const double *__restrict__ upd_z, const uint32_t *__restrict__ upd_sender,
                                               const int64_t *__restrict__ upd_arrival, const chd_aoi_query *queries, uint32_t nq,
                                               const uint32_t *q_sub, const double *spot_x, const double *spot_z,
                                               const uint32_t *spot_dist, int64_t now_ns, uint32_t cur_tick, uint32_t key_bits, uint32_t dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= nb) {
        if (dbg & 1u) return;
        // interest role: one query per wave (wave-private LDS areas, no workgroup barrier in there)
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t qi = (blockIdx.x - nb) * FRONT_QUERIES_PER_BLOCK + wave;
        const size_t per = max(aoi_lds_bytes_dev(lim, w.capq), (size_t)AOI_LDS_FLOOR);
        interest_query(g, lim, w, queries, nq, q_sub, spot_x, spot_z, spot_dist, now_ns, cur_tick, qi, smem + wave * per);
        return;
    }
    const uint32_t bid = blockIdx.x;
    if (dbg & 2u) return;
    // K1: the batch of entity updates, 256 per step
    for (uint32_t b = bid; b * 256u < n_upd; b += nb) {
        ingest_block(g, w, n_upd, upd_idx, upd_x, upd_z, upd_sender, cur_tick, upd_arrival, now_ns, b, (cur_tick << 8) | 0x80000000u);
        __syncthreads();  // (its LDS counters are reused by the next step)
    }
    front_barrier(w, (FrontBar *)w.front_bar, bar_seq + 1, nb);
    // K2a: this workgroup's tile of entity slots -> per-cell counts and aggregates
    index_hist_block(w, g.ncell, cur_tick, bid, smem);
    front_barrier(w, (FrontBar *)w.front_bar, bar_seq + 2, nb);
    // K2b: one wave per cell (grids of up to 1024 cells: the scatter scans the cell totals itself)
    for (uint32_t c = bid * 4u + (threadIdx.x >> 6); c < g.ncell; c += nb * 4u) index_scan_cell(w, g.ncell, 0, c);
    front_barrier(w, (FrontBar *)w.front_bar, bar_seq + 3, nb);
    // K2c: stable scatter of the tile
    index_scatter_block(w, g.ncell, key_bits, cur_tick, 1, bid, smem);
}

bool front_fusable(const DevGrid &g, const WorldDev &w) {
    return g.ncell <= 1024 && w.nblk >= 1 && w.nblk <= FRONT_MAX_INDEX_BLOCKS && w.front_bar != nullptr;
}

void launch_front(hipStream_t st, DevGrid g, AoiLimits lim, WorldDev w, unsigned long long launch_seq, uint32_t n_upd,
                  const uint32_t *upd_idx, const double *upd_x, const double *upd_z, const uint32_t *upd_sender,
                  const int64_t *upd_arrival, const chd_aoi_query *queries, uint32_t nq, const uint32_t *q_sub,
                  const double *spot_x, const double *spot_z, const uint32_t *spot_dist, int64_t now_ns, uint32_t cur_tick) {
    const uint32_t nb = w.nblk;
    const size_t per = std::max<size_t>(aoi_lds_bytes(lim, w.capq), (size_t)AOI_LDS_FLOOR);
    const size_t lds = std::max<size_t>(per * FRONT_QUERIES_PER_BLOCK, (size_t)5 * g.ncell * 4);
    aoi_allow_lds(k_front, lds);
    const uint32_t qblocks = (nq + FRONT_QUERIES_PER_BLOCK - 1) / FRONT_QUERIES_PER_BLOCK;
    // (three barriers per launch: their sequence numbers continue from launch to launch)
    hipLaunchKernelGGL(k_front, dim3(nb + qblocks), dim3(256), lds, st, g, lim, w, nb, 3ull * (launch_seq - 1ull), n_upd, upd_idx,
                       upd_x, upd_z, upd_sender, upd_arrival, queries, nq, q_sub, spot_x, spot_z, spot_dist, now_ns, cur_tick,
                       bits_for(g.ncell), (uint32_t)(getenv("CHD_FRONT_DBG") ? atoi(getenv("CHD_FRONT_DBG")) : 0));
}
