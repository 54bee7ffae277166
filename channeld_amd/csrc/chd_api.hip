// chd_api.hip — the C-ABI of libchd_spatial.so (include/chd_spatial.h).
// Host plumbing only: argument validation, device buffers, stream ordering,
// error reporting.  Every computation is a HIP kernel; there is no CPU path.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <random>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

// RCCL's few types and constants, declared here: librccl.so is dlopen'ed by chd_shard_comm_init (a single-GPU gateway never loads
// it), and a single-GPU BUILD does not need its headers either.  CHD_WITH_RCCL_HEADER=1 takes them from <rccl/rccl.h> instead (the
// static_asserts below then check these declarations against the real ones).
#ifdef CHD_WITH_RCCL_HEADER
#include <rccl/rccl.h>
static_assert(ncclSuccess == 0 && ncclUint8 == 1 && sizeof(ncclUniqueId) == 128, "the declarations of the #else branch are RCCL's");
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#include "chd_kernels.h"

extern uint32_t index_num_blocks(uint32_t N);

namespace {

// The ctx lock, FIFO: a std::mutex makes no fairness promise, and under the reference's calling pattern — thousands of goroutines
// in GetChannelId / QueryChannelIds (message_spatial.go:59,236,354; spatial.go:611) beside ONE ticking goroutine — sixteen threads
// re-taking the lock back to back starved the tick for seconds (tests/c/concurrent_callers.c measured 2 ticks in 60 s).  Tickets:
// whoever asked first is served first, so a tick waits for at most one call of every other thread.
// A waiter sleeps on the condition variable of ITS ticket (ticket % 64): an unlock wakes the next ticket's holder and nobody else
// (waiters 64 tickets apart share a variable and re-check) — one variable for all made every unlock wake every waiter, O(n^2)
// wake-ups with thousands of goroutine threads queued.
// The library's two TEST hooks are environment variables: CHD_SHARD_TRANSPORT=hostpipe (a blocking POSIX-shm stand-in for RCCL so that
// chd_shard_tick can run with several ranks on one GPU) and CHD_TEST_DROP_GATE_RAISE (drops one gate raise: the time-out path).  A
// production build compiles them out: -DCHD_NO_TEST_HOOKS (CHD_EXTRA_FLAGS of channeld_amd/build.py).
static inline const char *test_hook_env(const char *name) {
#ifdef CHD_NO_TEST_HOOKS
    (void)name;
    return nullptr;
#else
    return getenv(name);
#endif
}

class FairMutex {
    static constexpr unsigned K = 64;
    std::mutex m;
    std::condition_variable cv[K];
    uint64_t next = 0, serving = 0;

public:
    void lock() {
        std::unique_lock<std::mutex> l(m);
        const uint64_t t = next++;
        cv[t % K].wait(l, [&] { return t == serving; });
    }
    void unlock() {
        uint64_t s;
        {
            std::lock_guard<std::mutex> l(m);
            s = ++serving;
        }
        cv[s % K].notify_all();
    }
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct World {
    bool created = false;
    chd_world_cfg cfg{};
    WorldDev d{};
    std::vector<void *> allocs;
    unsigned char *slab_cur = nullptr;  // (walloc: the slab being filled)
    size_t slab_used = 0;
    chd_fanout_rec *recs_dense = nullptr;
    uint64_t recs_dense_cap = 0;
    uint32_t *list_dense = nullptr;  // fetch-time staging of the packed unsub / new-sub lists
    size_t list_dense_cap = 0;
    uint64_t *rec_off_exact = nullptr;  // [S+1]
    uint32_t *newsub_sub = nullptr, *newsub_cell = nullptr, *newsub_iv = nullptr;
    int64_t last_now = INT64_MIN;
    uint32_t last_nq = 0;
    // which entity slots are live (host mirror of EF_ALIVE for caller-managed slots: spawn / despawn are synchronous host
    // calls) and for how many ticks in a row every one of them has sent an update: what tick_locked picks the emit form by
    std::vector<uint8_t> live;
    uint32_t n_live = 0, full_streak = 0;
    bool ticked = false;
    int slot_mode = 0;  // 0 unset, 1 caller-chosen slots (chd_world_spawn), 2 library-managed (chd_shard_spawn)
    bool wire = false;                 // CHD_WORLD_WIRE
    WireDev x{};
    uint64_t wire_cap = 0;             // bytes allocated for x.bytes
    uint64_t cdesc_cap = 0;            // copy descriptors allocated for x.cdesc
    uint64_t wire_ranges = 0;          // chd_wire_build_info: image ranges / record-path connections of the last build
    uint32_t wire_slow_conns = 0;
    bool wire_built = false;
    // region-sharded worlds: halo exchange layout (chd_shard_halo_layout)
    uint32_t halo_rank = 0, halo_world = 0;
    uint64_t *d_halo_send_off = nullptr, *d_halo_recv_off = nullptr;  // [world] segment offsets on the device
    uint32_t *d_ghost_off = nullptr;                                   // [world] first ghost entry of each source rank
    // adaptive emigrant segments (chd_shard_ingest, cap_used): the global maximum segment count of tick t arrives in pinned
    // host memory by an async copy enqueued by chd_shard_import; tick t + 2 reads it (its event completed long before)
    uint32_t *h_mig_gmax = nullptr;    // [4] pinned, by tick & 3
    hipEvent_t ev_mig[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t mig_tick[4] = {0, 0, 0, 0};  // tick whose maximum the slot holds (0 = none)
    uint32_t mig_cap = 0;                 // capacity the last chd_shard_ingest used
    // the update log by channel (WorldDev::log_on): this tick's update stream as chd_shard_ingest was given it, logged by shard_import_locked
    const double *log_x = nullptr, *log_z = nullptr;
    const uint8_t *log_has = nullptr;
    uint32_t log_nchan = 0;
    bool ingest_pending = false;  // chd_shard_ingest[_pre] ran and chd_shard_import has not yet
    std::vector<uint32_t> group_id;    // host copy: handover group id per entity slot (0 = none), chd_world_set_entity_groups
    void *grp_buf[4] = {nullptr, nullptr, nullptr, nullptr};  // device arrays behind WorldDev::grp_*
    bool plan_recipients = false;      // CHD_WORLD_HANDOVER_RECIPIENTS
    bool overlap_interest = false;     // CHD_WORLD_OVERLAP_INTEREST
    bool join_in_unpack = false;       // chd_shard_tick -> shard_fanout_locked: k_halo_unpack carries the join
    // CHD_WORLD_GATED_OVERLAP: the fork / join of the interest stream as device-side flags (chd_kernels.h: GATE_*).  gate_asked =
    // the world's flag; gated = ... and the context's two streams were SEEN to run side by side (gate_probe; again whenever
    // chd_set_stream changes the pair), and no gate has timed out since (OVF_GATE: note_overflow turns it off for good)
    bool gate_asked = false, gated = false;
    unsigned long long *gate = nullptr;  // [GATE_WORDS] the two counters, then the probe's scratch
    unsigned long long gate_top = 0, gate_epi = 0;
    uint32_t gate_timeouts = 0;
    unsigned long long test_drop_raise = 0;  // CHD_TEST_DROP_GATE_RAISE (tests of the time-out path)
    bool overlap_deferred = false;     // CHD_WORLD_OVERLAP_DEFERRED
    // CHD_WORLD_PIPELINE_TICKS: everything the record-writing kernel reads (and the record buffer) exists twice, by tick parity
    bool pipe_alloc = false, pipe_on = false;
    uint32_t *pb_n_simple[2] = {nullptr, nullptr}, *pb_ce_chan[2] = {nullptr, nullptr}, *pb_ticket[2] = {nullptr, nullptr};
    uint64_t *pb_rec_ub[2] = {nullptr, nullptr};
    uint4 *pb_seg_desc[2] = {nullptr, nullptr}, *pb_seg_desc2[2] = {nullptr, nullptr};
    chd_fanout_rec *pb_recs[2] = {nullptr, nullptr};
    // ... and on worlds with exact update buffers (WorldDev::off_on): what k_fanout_emit_filt_cm(t) reads or writes while the stages
    // of tick t+1 rebuild it — the cells' compact entries and offset columns, the filtered descriptors' lists, items and windows, the
    // per-subscription and per-connection record counts
    bool pipe_exact = false;
    uint2 *pb_ce8[2] = {nullptr, nullptr};
    uint32_t *pb_ce_off[2] = {nullptr, nullptr}, *pb_cell_sorted[2] = {nullptr, nullptr}, *pb_filt_nitems[2] = {nullptr, nullptr};
    uint32_t *pb_pair_nrec[2] = {nullptr, nullptr}, *pb_rec_cnt[2] = {nullptr, nullptr};
    uint4 *pb_cell_flist[2] = {nullptr, nullptr}, *pb_filt_items[2] = {nullptr, nullptr};
    FiltWin *pb_filt_win[2] = {nullptr, nullptr};
    hipEvent_t ev_stages_done = nullptr, ev_stages_all = nullptr, ev_rec_sync = nullptr, ev_emit_done[2] = {nullptr, nullptr};
    bool last_desc = false;            // the last tick took the descriptor path (k_fanout_plan_seg's descriptors are this tick's)
    uint32_t *seg_cnt = nullptr;       // [S + 1] chd_tick_fetch_segments: segments per connection -> offsets
    uint64_t *seg_exp = nullptr;       // [S + 1] explicit records per connection -> offsets
    chd_fanout_segment *seg_stage = nullptr; size_t seg_stage_cap = 0;
    chd_fanout_rec *seg_rec_stage = nullptr; size_t seg_rec_stage_cap = 0;
    // chd_tick_segments_begin / _end: two ticks in flight; everything a tick hands out packed into one block per parity on the
    // device (header | per-connection offsets | query status | columns | segments | explicit records | handovers | lists) and
    // copied into its page-locked twin with ONE copy of exactly its bytes
    struct SegPipeSlot {
        unsigned char *d_blk = nullptr, *h = nullptr;
        unsigned long long *h_dev = nullptr;  // the page-locked block's header as the device addresses it
        hipEvent_t ev_up = nullptr, ev_begin = nullptr, ev_fill = nullptr;
        uint64_t ncol = 0, tick_no = 0;
        uint32_t nq = 0;
        bool timed = false;
    } segp[2];
    DevBuf segp_scratch[2][13];  // the input uploads' device buffers, one bank per parity (swapped into chd_ctx::scratch for the call)
    bool segp_ready = false;
    uint32_t segp_head = 0, segp_tail = 0, segp_pending = 0;
    hipStream_t segp_stream = nullptr;
    size_t segp_o_cnt = 0, segp_o_exp = 0, segp_o_qst = 0, segp_o_var = 0, segp_cap = 0;
    // native collectives (chd_shard_comm_init): the two exchanges of a sharded tick on RCCL inside the library
    ncclComm_t comm = nullptr;
    struct HostPipe *pipe = nullptr;   // CHD_SHARD_TRANSPORT=hostpipe (a TEST transport, see HostPipe): then comm == nullptr
    uint32_t comm_rank = 0, comm_world = 0, comm_cap = 0;
    uint32_t comm_alloc_cap = 0, comm_alloc_world = 0;  // what mig_send / mig_recv / the halo buffers were sized for
    uint32_t comm_try_cap = 0, comm_try_world = 0;      // ... what a call that may have failed half-way began to size them for
    chd_entity_state *mig_send = nullptr, *mig_recv = nullptr;   // [world][cap + 1]
    chd_handover_request *req_send = nullptr, *req_recv = nullptr;  // [world][CHD_SHARD_REQ_CAP + 1] (handover lists only)
    unsigned char *halo_send_buf = nullptr, *halo_recv_buf = nullptr;
    std::vector<chd_halo_seg> halo_segs;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_halo_ready = nullptr, ev_halo_done = nullptr;
    uint32_t *ho_rcp_off = nullptr;    // [handovers_cap + 1] recipients of handover h: [off[h], off[h+1])
    uint32_t *ho_rcp_conn = nullptr;   // connection ids
    uint8_t *ho_rcp_kind = nullptr;    // CHD_HO_*
    uint32_t *ho_rcp_mask = nullptr;   // per recipient: which entities of its handover carry their entityData (chd_handover_recipients_ex)
    uint64_t ho_rcp_cap = 0;
    uint32_t *ho_rcp_own = nullptr;    // [handovers_cap] step 1 unsubscribes the src spatial server's connection (spatial.go:688-694)
    // region-sharded worlds: the subscriptions as they were at the tick's start (chd_shard_handover_recipients plans on them after the
    // tick, when the host has gathered the whole world's handover records), the uploaded records, their count as k_handover_recipients reads it
    unsigned long long *snap_bits = nullptr; uint32_t *snap_cell = nullptr, *snap_cnt = nullptr;
    chd_handover_rec *rcp_recs = nullptr; uint32_t rcp_recs_cap = 0; uint32_t *rcp_ctr = nullptr;
    uint32_t *rcp_off = nullptr, *rcp_own = nullptr;
    uint32_t *server_conn = nullptr;   // chd_world_set_server_connections (device; WorldDev::server_conn points here while a table is set)
};

}  // namespace

#define EV_PER_TICK (CHD_N_STAGES + 5)  // stage boundaries, interest begin/end (second stream), dominant emit kernel end / begin

struct chd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // where work is enqueued (own_stream unless chd_set_stream)
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // interest updates run here, beside ingest + index build on `stream`; pipelined ticks: all stages
    hipStream_t aux2_stream = nullptr;  // pipelined ticks: the interest updates, beside ingest + index build on aux_stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    chd_grid_cfg cfg{};
    DevGrid g{};
    AoiLimits lim{};
    FairMutex mu;  // (every entry point that touches device state holds it for the call)
    std::string err;
    World w;
    TickRing ring{};
    // scratch for the stateless entry points and for chd_tick's staging
    DevBuf scratch[16];
    hipStream_t up_stream = nullptr;    // chd_tick_segments_begin: host_tick_stage's uploads go to this stream
    bool force_device = false;          // CHD_NO_HOST_FAST_PATH=1: small stateless calls go to the device too (tests, measurements)
    bool gchain = false, gchain_prev = false;  // ... was a serial tick with gated overlap (CHD_WORLD_GATED_OVERLAP)
    bool chain = false, chain_prev = false;  // the previous call on this ctx was a pipelined tick (bind() shifts them)
    int prof_depth = 0;                 // 0 = off
    uint32_t prof_every = 1;            // CHD_PROF_RECORD_KERNEL_EVERY(n): that pair on every n-th tick only (the others record nothing)
    bool prof_kernel_only = false;      // chd_set_profiling_scope(CHD_PROF_RECORD_KERNEL): only the event pair around the dominant emit kernel
    std::vector<hipEvent_t> ev;         // [prof_depth][EV_PER_TICK]: stage boundaries on `stream`, then interest begin/end
    std::vector<uint8_t> ev_overlap;    // [prof_depth] the slot's tick ran the interest stage on aux_stream
    chd_tick_stats stats{};
};

namespace {

thread_local std::string tl_err;

int fail(chd_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tl_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(ctx, CHD_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

int ensure(chd_ctx *ctx, int slot, size_t bytes) {
    DevBuf &b = ctx->scratch[slot];
    if (bytes <= b.cap) return CHD_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
    HIPCHK(hipMalloc(&b.p, cap));
    b.cap = cap;
    return CHD_OK;
}

template <typename T>
T *sbuf(chd_ctx *ctx, int slot) { return (T *)ctx->scratch[slot].p; }

int bind(chd_ctx *ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    // every entry point passes here once: a pipelined tick that directly follows a pipelined tick (nothing else was asked
    // of this ctx in between) may start its stages while the previous tick's records are still being written
    ctx->chain_prev = ctx->chain;
    ctx->chain = false;
    ctx->gchain_prev = ctx->gchain;
    ctx->gchain = false;
    return CHD_OK;
}

int up(chd_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return CHD_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->up_stream ? ctx->up_stream : ctx->stream));
    return CHD_OK;
}
int down(chd_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return CHD_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return CHD_OK;
}

int after_launch(chd_ctx *ctx) {
    HIPCHK(hipGetLastError());
    return CHD_OK;
}

#define TRY(x)                \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

// The world's arrays come out of a few large slabs, not one hipMalloc each: the tick's small kernels touch ten to twenty arrays per
// workgroup, and separately mapped allocations of a few hundred KB each cost them an address translation per array (measured:
// k_cell_arrange over 13 arrays of one cell took 18-28 us for a 3 us body).  Arrays of 32 MiB and more keep their own allocation.
// CHD_WORLD_SLABS=0: one allocation per array, as before (A/B runs).
template <typename T>
int walloc(chd_ctx *ctx, T **out, size_t count, bool zero = true) {
    World &W = ctx->w;
    void *p = nullptr;
    size_t bytes = (std::max<size_t>(count * sizeof(T), 256) + 255) & ~(size_t)255;
    static const bool slabs = [] { const char *e = getenv("CHD_WORLD_SLABS"); return !(e && e[0] == '0'); }();
    constexpr size_t SLAB = 64ull << 20;
    if (slabs && bytes < (32ull << 20)) {
        if (!W.slab_cur || W.slab_used + bytes > SLAB) {
            void *sl = nullptr;
            HIPCHK(hipMalloc(&sl, SLAB));
            W.allocs.push_back(sl);
            W.slab_cur = (unsigned char *)sl;
            W.slab_used = 0;
        }
        p = W.slab_cur + W.slab_used;
        W.slab_used += bytes;
    } else {
        HIPCHK(hipMalloc(&p, bytes));
        W.allocs.push_back(p);
    }
    if (zero) HIPCHK(hipMemsetAsync(p, 0, bytes, ctx->stream));
    *out = (T *)p;
    return CHD_OK;
}

}  // namespace

// dense per-connection packing of the emitted records (for the host-facing fetch):
// a connection's records sit in one segment per subscription inside its range
__global__ void __launch_bounds__(256) k_pack_records(WorldDev w, const uint64_t *exact_off, chd_fanout_rec *dense,
                                                      uint32_t *dense_mask) {
    uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    if (w.rec_cnt[s] == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cnt = w.pair_cnt[s];
    const size_t pbase = (size_t)s * w.capq;
    const chd_fanout_rec *base = w.recs + w.rec_ub[s];
    chd_fanout_rec *dst = dense + exact_off[s];
    const uint32_t *mbase = dense_mask ? w.rec_mask + w.rec_ub[s] : nullptr;
    uint32_t *mdst = dense_mask ? dense_mask + exact_off[s] : nullptr;
    for (uint32_t p = 0; p < cnt; p++) {
        const uint32_t n = w.pair_nrec[pbase + p];
        const chd_fanout_rec *src = base + w.pair_rel[pbase + p];
        for (uint32_t k = lane; k < n; k += 64) dst[k] = src[k];
        dst += n;
        if (mdst) {
            const uint32_t *msrc = mbase + w.pair_rel[pbase + p];
            for (uint32_t k = lane; k < n; k += 64) mdst[k] = msrc[k];
            mdst += n;
        }
    }
}

// Order-independent digest of the tick's fan-out records, where they lie (one padded segment per subscription):
// h = mix64(conn << 32 | channel) per record (SplitMix64's finaliser); per connection the sum of its hashes, and
// through 64 hashed buckets {count, sum, xor, sum of mix64(h + merged-updates mask)}.  One wave per connection, two
// records per lane and load (segments start on 128-byte lines).  HBM-bound: reads the 8 B/record stream once.
__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    return k ^ (k >> 31);
}

__global__ void __launch_bounds__(256) k_records_digest(WorldDev w, unsigned long long *conn_sum, unsigned long long *buckets) {
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long sum = 0, x = 0, summ = 0, cnt = 0;
    if (w.sub_alive[s] && w.rec_cnt[s]) {
        const uint32_t np = w.pair_cnt[s];
        const size_t pbase = (size_t)s * w.capq;
        const uint64_t base = w.rec_ub[s];
        for (uint32_t p = 0; p < np; p++) {
            const uint32_t n = w.pair_nrec[pbase + p];
            if (!n) continue;
            const uint64_t at = base + w.pair_rel[pbase + p];
            const chd_fanout_rec *src = w.recs + at;
            const uint32_t *msrc = w.rec_mask ? w.rec_mask + at : nullptr;
            for (uint32_t k = 2 * lane; k < n; k += 128) {
                const uint4 v = *(const uint4 *)(const void *)(src + k);  // (the pad of the last line is readable)
                const unsigned long long h0 = mix64(((unsigned long long)v.x << 32) | v.y);
                sum += h0; x ^= h0; cnt++;
                summ += mix64(h0 + (msrc ? msrc[k] : 0u));
                if (k + 1 < n) {
                    const unsigned long long h1 = mix64(((unsigned long long)v.z << 32) | v.w);
                    sum += h1; x ^= h1; cnt++;
                    summ += mix64(h1 + (msrc ? msrc[k + 1] : 0u));
                }
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        x ^= __shfl_xor(x, d);
        summ += __shfl_xor(summ, d);
        cnt += __shfl_xor(cnt, d);
    }
    if (lane == 0) {
        if (conn_sum) conn_sum[s] = sum;
        if (cnt) {
            unsigned long long *b = buckets + (size_t)(s & 63u) * 16;  // one 128-byte line per bucket
            atomicAdd(b, cnt);
            atomicAdd(b + 1, sum);
            atomicXor(b + 2, x);
            atomicAdd(b + 3, summ);
        }
    }
}

__global__ void __launch_bounds__(256) k_rec_cnt(WorldDev w) {
    uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (s >= w.S) return;
    uint32_t sum = 0;
    if (w.sub_alive[s]) {
        const uint32_t cnt = w.pair_cnt[s];
        for (uint32_t p = threadIdx.x & 63u; p < cnt; p += 64) sum += w.pair_nrec[(size_t)s * w.capq + p];
    }
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63u) == 0) w.rec_cnt[s] = sum;
}

// ---- chd_tick_fetch_segments: the fan-out as segment descriptors (see include/chd_spatial.h) ----
// pass 1 (fill == 0): per connection the number of segments and of explicit records; pass 2: write them.
// One wave per connection.  Simple subscriptions = the descriptors of k_fanout_plan_seg (have_desc); explicit = every other
// subscription that emitted records this tick.  LDS: one bit per subscription index (capq <= 65534).
#define SEG_BITMAP_WORDS 2048
// The header of a chd_tick_segments_begin block (u64 words): counts, the tick's row of the history ring, and the byte offsets of
// the block's sections — written by k_seg_scan into the device block and into the page-locked copy the host reads first
enum { SEGH_NSEG = 0, SEGH_NEXP = 1, SEGH_ROW = 2 /* ..9 */, SEGH_GATE_FAIL = 10, SEGH_REC_UB = 11, SEGH_TOTAL = 12, SEGH_OFF_COL = 13,
       SEGH_OFF_SEG = 14, SEGH_OFF_REC = 15, SEGH_OFF_HO = 16, SEGH_OFF_UN = 17 /* sub, channel */, SEGH_OFF_NEW = 19 /* sub, channel, interval */,
       SEGH_NCOL = 22, SEGH_NQ = 23, SEGH_CAP = 24, SEGH_NHO = 25, SEGH_NUN = 26, SEGH_NNEW = 27, SEGH_FITS = 28, SEGH_TICK = 29, SEGH_WORDS = 32 };

__global__ void __launch_bounds__(64) k_segments(DevGrid g, WorldDev w, int have_desc, int fill, uint32_t *nseg, unsigned long long *nexp,
                                                 chd_fanout_segment *seg_out, chd_fanout_rec *rec_out, unsigned char *blk = nullptr) {
    __shared__ uint32_t simple_bits[SEG_BITMAP_WORDS];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    if (blk) {  // chd_tick_segments_begin: the outputs' places in the tick's block are in its header (k_seg_scan)
        const unsigned long long *hdr = (const unsigned long long *)blk;
        if (!hdr[SEGH_FITS]) return;
        seg_out = (chd_fanout_segment *)(blk + hdr[SEGH_OFF_SEG]);
        rec_out = (chd_fanout_rec *)(blk + hdr[SEGH_OFF_REC]);
    }
    const uint32_t cnt = w.sub_alive[s] ? w.pair_cnt[s] : 0u;
    const size_t pbase = (size_t)s * w.capq;
    const bool served = w.rec_ub[s + 1] <= w.recs_cap;  // (else the connection was skipped this tick and flagged)
    const uint32_t ns = (have_desc && served && cnt) ? w.n_simple[s] : 0u;
    for (uint32_t k = lane; k < (cnt + 31) / 32; k += 64) simple_bits[k] = 0;
    __syncthreads();
    const uint32_t seg0 = fill ? nseg[s] : 0u;
    const unsigned long long exp0 = fill ? nexp[s] : 0ull;
    for (uint32_t k = lane; k < ns; k += 64) {
        const uint4 d = w.seg_desc[pbase + k], d2 = w.seg_desc2[pbase + k];
        atomicOr(&simple_bits[d2.y >> 5], 1u << (d2.y & 31u));
        if (fill) {
            const uint32_t nw = d.w & 7u;
            uint32_t info = d.z & 0x3FFFFFu, nrec = (d.w & 8u) ? d.z + 1u : 0u;
            if (d.w & 8u) info |= CHD_SEG_FIRST;
            if (d.w & 16u) info |= CHD_SEG_NONE;
            info |= nw << 25;
            for (uint32_t j = 0; j < nw; j++) {
                if ((d.w >> (8 + j)) & 1u) { info |= CHD_SEG_OWN(j); nrec += 1u; }
                if (!(d.w & 16u)) nrec += d.z;
            }
            chd_fanout_segment o;
            o.channel = d2.x + g.id_start; o.off = d.y; o.n_info = info; o.n_records = nrec;
            seg_out[seg0 + k] = o;
        }
    }
    __syncthreads();
    uint32_t n_explicit_seg = 0;
    unsigned long long n_explicit_rec = 0;
    for (uint32_t p0 = 0; p0 < cnt; p0 += 64) {
        const uint32_t p = p0 + lane;
        uint32_t n = 0;
        if (p < cnt && served && !((simple_bits[p >> 5] >> (p & 31u)) & 1u)) n = w.pair_nrec[pbase + p];
        const uint64_t m = __ballot(n != 0);
        // exclusive prefix of the record counts over the wave
        unsigned long long inc = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d);
            if ((int)lane >= d) inc += o;
        }
        if (fill && n) {
            const unsigned long long at = n_explicit_rec + inc - n;
            chd_fanout_segment o;
            o.channel = w.pair_cell[pbase + p] + g.id_start;
            o.off = (uint32_t)at; o.n_info = CHD_SEG_EXPLICIT; o.n_records = n;
            seg_out[seg0 + ns + n_explicit_seg + mask_rank(m)] = o;
        }
        if (fill) {
            // the wave copies the segments' records together, one segment after the other
            for (uint64_t mm = m; mm; mm &= mm - 1) {
                const int src = __ffsll((unsigned long long)mm) - 1;
                const uint32_t pn = (uint32_t)__shfl((int)n, src);
                const unsigned long long pat = n_explicit_rec + __shfl(inc, src) - pn;
                const chd_fanout_rec *from = w.recs + w.rec_ub[s] + w.pair_rel[pbase + p0 + (uint32_t)src];
                for (uint32_t k = lane; k < pn; k += 64) rec_out[exp0 + pat + k] = from[k];
            }
        }
        n_explicit_seg += (uint32_t)__popcll(m);
        n_explicit_rec += __shfl(inc, 63);
    }
    if (!fill && lane == 0) {
        nseg[s] = ns + n_explicit_seg;
        nexp[s] = n_explicit_rec;
    }
}

__global__ void __launch_bounds__(256) k_widen(const uint32_t *in, uint64_t *out, uint32_t n) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = in[i];
}

namespace {
// CHD_WORLD_GATED_OVERLAP holds only where a kernel on the tick's stream can WAIT for a kernel on the second stream: probed with the
// real thing (streams_run_side_by_side), at world creation and whenever the stream pair changes.  Resets the counters: the
// probe leaves the flags' lines dirty, and a changed pair starts a new chain anyway.
static int gate_probe(chd_ctx *ctx) {
    World &W = ctx->w;
    W.gated = false;
    if (!W.gate_asked || !W.gate || W.gate_timeouts) return CHD_OK;
    bool ok = false;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux_stream));
    // (the serial schedule's pair, and the pipelined schedule's: its stages run on aux_stream, its interest updates on aux2_stream)
    bool ok2 = true;
    if (streams_run_side_by_side(ctx->stream, ctx->aux_stream, W.gate + GATE_WORDS, (unsigned *)(W.gate + GATE_WORDS + 16), &ok) != 0 ||
        (ok && W.pipe_alloc && streams_run_side_by_side(ctx->aux_stream, ctx->aux2_stream, W.gate + GATE_WORDS, (unsigned *)(W.gate + GATE_WORDS + 16), &ok2) != 0))
        return fail(ctx, CHD_E_HIP, "CHD_WORLD_GATED_OVERLAP: the stream probe failed");
    ok = ok && ok2;
    HIPCHK(hipMemsetAsync(W.gate, 0, sizeof(unsigned long long) * (GATE_WORDS + 32), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    W.gate_top = W.gate_epi = 0;
    ctx->gchain = ctx->gchain_prev = false;
    W.gated = ok;
    return CHD_OK;
}

static uint32_t schedule_bits(const chd_ctx *ctx) {
    const World &W = ctx->w;
    return (W.overlap_interest ? CHD_SCHED_OVERLAP_INTEREST : 0u) | (W.gated ? CHD_SCHED_GATED : 0u) | (W.pipe_on ? CHD_SCHED_PIPELINED : 0u) |
           (W.d.cm_emit ? CHD_SCHED_CELL_MAJOR : 0u) | (W.d.off_on ? CHD_SCHED_ARRIVAL_OFFSETS : 0u);
}

// Every call that synchronises with the device and looks at a tick's results passes here: a gate that timed out (the sticky
// count behind WorldDev::gate_fail; the tick it happened in carries overflow bit 0x4000 in its own row of the tick history) means
// the two streams did not run side by side after all — the world takes HIP events from now on.
static int gate_poll_begin(chd_ctx *ctx, unsigned long long *fails) {  // (enqueue; the caller's own synchronisation completes it)
    World &W = ctx->w;
    *fails = 0;
    if (W.gate && W.gated) HIPCHK(hipMemcpyAsync(fails, W.gate + GATE_FAIL, sizeof *fails, hipMemcpyDeviceToHost, ctx->stream));
    return CHD_OK;
}
static void gate_poll_end(chd_ctx *ctx, unsigned long long fails) {
    World &W = ctx->w;
    if (W.gated && fails > W.gate_timeouts) {
        W.gate_timeouts = (uint32_t)fails;
        W.gated = false;
        ctx->gchain = ctx->gchain_prev = false;
        fprintf(stderr, "chd: CHD_WORLD_GATED_OVERLAP: a device-side gate timed out (overflow bit 0x4000 in that tick): this world orders its streams with HIP events from now on\n");
    }
}
static int gate_check(chd_ctx *ctx) {
    unsigned long long fails = 0;
    TRY(gate_poll_begin(ctx, &fails));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    gate_poll_end(ctx, fails);
    return CHD_OK;
}

}  // namespace

extern "C" {

int chd_abi_version(void) { return CHD_ABI_VERSION; }

const char *chd_last_error(const chd_ctx *ctx) {
    if (!tl_err.empty()) return tl_err.c_str();
    return ctx ? ctx->err.c_str() : "";
}

int chd_create(const chd_grid_cfg *cfg, int device, chd_ctx **out) {
    chd_ctx *ctx = nullptr;
    if (!cfg || !out) return fail(nullptr, CHD_E_INVAL, "chd_create: NULL argument");
    *out = nullptr;
    // LoadConfig validation, spatial.go:146-157 (same order)
    if (!(cfg->grid_width > 0) || !(cfg->grid_height > 0))
        return fail(nullptr, CHD_E_CONFIG, "GridWidth and GridHeight should be positive");
    if (cfg->grid_cols == 0 || cfg->grid_rows == 0)
        return fail(nullptr, CHD_E_CONFIG, "GridCols and GridRows should be positive");
    if (cfg->server_cols == 0 || cfg->server_rows == 0)
        return fail(nullptr, CHD_E_CONFIG, "ServerCols and ServerRows should be positive");
    if (cfg->strict_load_config && cfg->server_interest_border_size == 0)
        return fail(nullptr, CHD_E_CONFIG, "ServerInterestBorderSize should be positive");
    if ((uint64_t)cfg->grid_cols * cfg->grid_rows > 0x7FFFFFFFull)
        return fail(nullptr, CHD_E_CONFIG, "grid has too many cells");
    if (cfg->n_damping > CHD_MAX_DAMPING) return fail(nullptr, CHD_E_INVAL, "n_damping > %d", CHD_MAX_DAMPING);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CHD_E_NO_DEVICE, "no HIP device visible: libchd_spatial has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, CHD_E_NO_DEVICE, "device %d of %d", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(nullptr, CHD_E_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CHD_E_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only",
                    device, prop.gcnArchName);

    ctx = new chd_ctx();
    ctx->device = device;
    ctx->cfg = *cfg;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, CHD_E_HIP, "cannot create a stream on device %d", device);
    }
    ctx->stream = ctx->own_stream;
    // the second stream outranks the first: with CHD_WORLD_PIPELINE_TICKS its small, latency-bound kernels run beside a
    // record kernel whose 10^4 workgroups would otherwise take every wave slot that frees up until its last one is placed
    // (measured: k_ingest 176 us instead of 10 when it starts beside k_fanout_emit_seg at equal priority)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (const char *e = getenv("CHD_AUX_PRIORITY")) prio_hi = atoi(e);  // (experiments)
    if (hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->aux2_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return fail(nullptr, CHD_E_HIP, "cannot create the second stream on device %d", device);
    }
    DevGrid &g = ctx->g;
    g.gw = cfg->grid_width;
    g.gh = cfg->grid_height;
    g.offx = cfg->world_offset_x;
    g.offz = cfg->world_offset_z;
    // GridSize(): sqrt is correctly rounded on the host and on amd64 Go alike; the
    // two products and the sum are separate roundings (this file is built with
    // -ffp-contract=off).
    {
        volatile double a = g.gw * g.gw, b = g.gh * g.gh;
        volatile double s = a + b;
        g.gsz = std::sqrt(s);
    }
    g.world_w = g.gw * (double)cfg->grid_cols;
    g.world_h = g.gh * (double)cfg->grid_rows;
    g.cols = cfg->grid_cols;
    g.rows = cfg->grid_rows;
    g.ncell = cfg->grid_cols * cfg->grid_rows;
    g.id_start = cfg->spatial_channel_id_start ? cfg->spatial_channel_id_start : 0x10000u;
    g.server_cols = cfg->server_cols;
    g.server_rows = cfg->server_rows;
    g.sgc = cfg->grid_cols / cfg->server_cols + (cfg->grid_cols % cfg->server_cols ? 1 : 0);
    g.sgr = cfg->grid_rows / cfg->server_rows + (cfg->grid_rows % cfg->server_rows ? 1 : 0);
    g.border = cfg->server_interest_border_size;
    g.default_interval_ms = cfg->default_fanout_interval_ms ? cfg->default_fanout_interval_ms : 20;
    g.default_delay_ms = cfg->default_fanout_delay_ms;
    if (cfg->n_damping == 0) {  // message_spatial.go:16-29
        g.n_damp = 3;
        const uint32_t d[3] = {0, 1, 2}, iv[3] = {20, 50, 100};
        for (int i = 0; i < 3; i++) { g.damp_dist[i] = d[i]; g.damp_iv[i] = iv[i]; }
    } else {
        g.n_damp = cfg->n_damping;
        for (uint32_t i = 0; i < cfg->n_damping; i++) {
            g.damp_dist[i] = cfg->damping_max_dist[i];
            g.damp_iv[i] = cfg->damping_interval_ms[i];
            if (g.damp_iv[i] == 0) {
                (void)hipStreamDestroy(ctx->own_stream);
                delete ctx;
                return fail(nullptr, CHD_E_INVAL, "fan-out interval 0 makes the reference's tickData spin forever");
            }
        }
    }
    { const char *e = getenv("CHD_NO_HOST_FAST_PATH"); ctx->force_device = e && e[0] == '1'; }
    ctx->lim.maxax = 256;
    // (experiments: the long-lattice path's sample buffers are 6 of the 11.8 KB of LDS a query holds on config B and bound
    // the resident queries per CU at 13; CHD_AOI_MAXAX=64 keeps queries of up to 64 samples per axis — every shape of the
    // bench — and returns CHD_E_TOO_LARGE for longer lattices.  Not a product setting.)
    if (const char *e = getenv("CHD_AOI_MAXAX")) ctx->lim.maxax = (uint32_t)std::min(std::max(atoi(e), 64), 256);
    ctx->lim.winmax = std::min<uint32_t>(std::max<uint32_t>(g.ncell, 64), 4096);
    ctx->lim.maxdim = std::min<uint32_t>(ctx->lim.winmax, std::max(g.cols, g.rows));
    if (aoi_lds_bytes(ctx->lim, 1) > aoi_lds_limit()) {
        (void)hipStreamDestroy(ctx->own_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux2_stream);
        delete ctx;
        return fail(nullptr, CHD_E_CONFIG, "grid of %u x %u cells: the per-query work area exceeds the LDS of a CU", g.cols, g.rows);
    }
    ctx->ring.n = 0;
    ctx->ring.cur_tick = 0;
    *out = ctx;
    tl_err.clear();
    return CHD_OK;
}

void chd_destroy(chd_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)chd_shard_comm_destroy(ctx);  // (no-op without a communicator)
    if (ctx->w.ev_stages_done) {
        (void)hipEventDestroy(ctx->w.ev_stages_done);
        (void)hipEventDestroy(ctx->w.ev_rec_sync);
        (void)hipEventDestroy(ctx->w.ev_stages_all);
        for (auto &e : ctx->w.ev_emit_done) (void)hipEventDestroy(e);
    }
    for (void *p : ctx->w.allocs) (void)hipFree(p);
    if (ctx->w.h_mig_gmax) (void)hipHostFree(ctx->w.h_mig_gmax);
    for (auto &e : ctx->w.ev_mig) if (e) (void)hipEventDestroy(e);
    for (void *b : ctx->w.grp_buf) if (b) (void)hipFree(b);
    if (ctx->w.recs_dense) (void)hipFree(ctx->w.recs_dense);
    if (ctx->w.seg_stage) (void)hipFree(ctx->w.seg_stage);
    if (ctx->w.seg_rec_stage) (void)hipFree(ctx->w.seg_rec_stage);
    if (ctx->w.segp_stream) {
        (void)hipStreamSynchronize(ctx->w.segp_stream);
        (void)hipStreamDestroy(ctx->w.segp_stream);
    }
    for (auto &sl : ctx->w.segp) {
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d_blk) (void)hipFree(sl.d_blk);
        if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
        if (sl.ev_begin) (void)hipEventDestroy(sl.ev_begin);
        if (sl.ev_fill) (void)hipEventDestroy(sl.ev_fill);
    }
    for (auto &bank : ctx->w.segp_scratch)
        for (auto &b : bank)
            if (b.p) (void)hipFree(b.p);
    if (ctx->w.list_dense) (void)hipFree(ctx->w.list_dense);
    if (ctx->w.x.bytes) (void)hipFree(ctx->w.x.bytes);
    if (ctx->w.x.cdesc) (void)hipFree(ctx->w.x.cdesc);
    for (auto &b : ctx->scratch)
        if (b.p) (void)hipFree(b.p);
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
    (void)hipStreamDestroy(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux2_stream);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int chd_sync(chd_ctx *ctx) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    if (ctx->w.created) return gate_check(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

// Single-point callers (GetChannelId from handleCreateEntityChannel, handleQuerySpatialChannel, handleUnrealSpawnObject:
// one call per message) must not pay a device round trip (mutex + H2D + launch + sync, ~20 us) for two
// divisions: up to CHD_HOST_FAST_PATH points are answered on the host, lock-free, with the SAME arithmetic —
// IEEE double subtract / divide / floor, this file is built with -ffp-contract=off like the kernels — so the
// result is bit-identical to the kernel's (tests/test_gpu_parity.py compares the two paths point by point).
// This is not a fallback: the context still requires a gfx950 device, and every batched call runs on it.
#define CHD_HOST_FAST_PATH 16u
namespace {
inline bool host_coord(double v, uint32_t n, uint32_t &out) {  // grid_coord (chd_device.h)
    const double f = std::floor(v);
    if (!(f >= 0.0) || !(f < (double)n)) return false;
    out = (uint32_t)f;
    return true;
}
inline uint32_t host_channel_id(const DevGrid &g, double x, double z) {  // GetChannelIdWithOffset, spatial.go:169-180
    uint32_t gx, gy;
    volatile double dx = x - g.offx, dz = z - g.offz;  // (volatile: each operation rounded on its own, no re-association)
    volatile double qx = dx / g.gw, qz = dz / g.gh;
    if (!host_coord(qx, g.cols, gx)) return 0u;
    if (!host_coord(qz, g.rows, gy)) return 0u;
    return gx + gy * g.cols + g.id_start;
}
}  // namespace

int chd_get_channel_ids(chd_ctx *ctx, const double *x, const double *z, uint32_t n, uint32_t *out_ids) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (n && (!x || !z || !out_ids)) return fail(ctx, CHD_E_INVAL, "chd_get_channel_ids: NULL buffer");
    if (!n) return CHD_OK;
    if (n <= CHD_HOST_FAST_PATH && !ctx->force_device) {
        for (uint32_t i = 0; i < n; i++) out_ids[i] = host_channel_id(ctx->g, x[i], z[i]);
        return CHD_OK;
    }
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, sizeof(double) * n));
    TRY(ensure(ctx, 1, sizeof(double) * n));
    TRY(ensure(ctx, 2, sizeof(uint32_t) * n));
    TRY(up(ctx, sbuf<double>(ctx, 0), x, sizeof(double) * n));
    TRY(up(ctx, sbuf<double>(ctx, 1), z, sizeof(double) * n));
    launch_get_channel_ids(ctx->stream, ctx->g, sbuf<double>(ctx, 0), sbuf<double>(ctx, 1), n, sbuf<uint32_t>(ctx, 2));
    TRY(after_launch(ctx));
    TRY(down(ctx, out_ids, sbuf<uint32_t>(ctx, 2), sizeof(uint32_t) * n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_notify_decide(chd_ctx *ctx, const double *old_x, const double *old_z, const double *new_x,
                      const double *new_z, uint32_t n, uint32_t *src_ids, uint32_t *dst_ids, uint8_t *handover) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (n && (!old_x || !old_z || !new_x || !new_z || !src_ids || !dst_ids || !handover))
        return fail(ctx, CHD_E_INVAL, "chd_notify_decide: NULL buffer");
    if (!n) return CHD_OK;
    if (n <= CHD_HOST_FAST_PATH && !ctx->force_device) {  // (see chd_get_channel_ids)
        for (uint32_t i = 0; i < n; i++) {
            src_ids[i] = host_channel_id(ctx->g, old_x[i], old_z[i]);
            dst_ids[i] = src_ids[i] ? host_channel_id(ctx->g, new_x[i], new_z[i]) : 0u;  // (:613-617: the src error returns first)
            handover[i] = (src_ids[i] && dst_ids[i] && src_ids[i] != dst_ids[i]) ? 1 : 0;  // spatial.go:613-626
        }
        return CHD_OK;
    }
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    for (int k = 0; k < 4; k++) TRY(ensure(ctx, k, sizeof(double) * n));
    TRY(ensure(ctx, 4, sizeof(uint32_t) * n));
    TRY(ensure(ctx, 5, sizeof(uint32_t) * n));
    TRY(ensure(ctx, 6, n));
    const double *src[4] = {old_x, old_z, new_x, new_z};
    for (int k = 0; k < 4; k++) TRY(up(ctx, sbuf<double>(ctx, k), src[k], sizeof(double) * n));
    launch_notify_decide(ctx->stream, ctx->g, sbuf<double>(ctx, 0), sbuf<double>(ctx, 1), sbuf<double>(ctx, 2),
                         sbuf<double>(ctx, 3), n, sbuf<uint32_t>(ctx, 4), sbuf<uint32_t>(ctx, 5), sbuf<uint8_t>(ctx, 6));
    TRY(after_launch(ctx));
    TRY(down(ctx, src_ids, sbuf<uint32_t>(ctx, 4), sizeof(uint32_t) * n));
    TRY(down(ctx, dst_ids, sbuf<uint32_t>(ctx, 5), sizeof(uint32_t) * n));
    TRY(down(ctx, handover, sbuf<uint8_t>(ctx, 6), n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_query_channel_ids(chd_ctx *ctx, const chd_aoi_query *queries, uint32_t nq, const double *spot_x,
                          const double *spot_z, const uint32_t *spot_dist, uint32_t n_spots_total,
                          uint32_t *offsets, uint32_t *ids, uint32_t *dists, uint32_t *intervals_ms,
                          uint32_t cap, int32_t *status) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (!offsets || !status || (nq && !queries)) return fail(ctx, CHD_E_INVAL, "chd_query_channel_ids: NULL buffer (a nil query is an error in the reference, spatial.go:183)");
    if (cap && (!ids || !dists)) return fail(ctx, CHD_E_INVAL, "chd_query_channel_ids: NULL output");
    if (n_spots_total && (!spot_x || !spot_z)) return fail(ctx, CHD_E_INVAL, "chd_query_channel_ids: NULL spots");
    for (uint32_t i = 0; i < nq; i++) {
        if ((queries[i].shapes & CHD_SHAPE_SPOTS) &&
            ((uint64_t)queries[i].spot_off + queries[i].n_spots > n_spots_total || queries[i].n_spot_dists > queries[i].n_spots))
            return fail(ctx, CHD_E_INVAL, "query %u: spot range out of bounds", i);
    }
    offsets[0] = 0;
    if (!nq) return CHD_OK;
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    const uint32_t stride = std::min<uint32_t>(ctx->g.ncell, ctx->lim.winmax);
    TRY(ensure(ctx, 0, sizeof(chd_aoi_query) * nq));
    TRY(ensure(ctx, 1, sizeof(double) * std::max(n_spots_total, 1u)));
    TRY(ensure(ctx, 2, sizeof(double) * std::max(n_spots_total, 1u)));
    TRY(ensure(ctx, 3, sizeof(uint32_t) * std::max(n_spots_total, 1u)));
    TRY(ensure(ctx, 4, sizeof(uint32_t) * (size_t)nq * stride));
    TRY(ensure(ctx, 5, sizeof(uint32_t) * (size_t)nq * stride));
    TRY(ensure(ctx, 6, sizeof(uint32_t) * (size_t)nq * stride));
    TRY(ensure(ctx, 7, sizeof(uint32_t) * (nq + 1)));   // counts
    TRY(ensure(ctx, 8, sizeof(uint32_t) * (nq + 1)));   // offsets
    TRY(ensure(ctx, 9, sizeof(int32_t) * nq));
    TRY(ensure(ctx, 10, sizeof(uint32_t) * std::max(cap, 1u)));
    TRY(ensure(ctx, 11, sizeof(uint32_t) * std::max(cap, 1u)));
    TRY(ensure(ctx, 12, sizeof(uint32_t) * std::max(cap, 1u)));
    TRY(ensure(ctx, 13, aoi_scratch_bytes(ctx->lim) * (size_t)nq));
    TRY(up(ctx, sbuf<void>(ctx, 0), queries, sizeof(chd_aoi_query) * nq));
    TRY(up(ctx, sbuf<void>(ctx, 1), spot_x, sizeof(double) * n_spots_total));
    TRY(up(ctx, sbuf<void>(ctx, 2), spot_z, sizeof(double) * n_spots_total));
    if (spot_dist) TRY(up(ctx, sbuf<void>(ctx, 3), spot_dist, sizeof(uint32_t) * n_spots_total));
    else HIPCHK(hipMemsetAsync(sbuf<void>(ctx, 3), 0, sizeof(uint32_t) * std::max(n_spots_total, 1u), ctx->stream));
    launch_aoi_stateless(ctx->stream, ctx->g, ctx->lim, sbuf<chd_aoi_query>(ctx, 0), nq, sbuf<double>(ctx, 1),
                         sbuf<double>(ctx, 2), sbuf<uint32_t>(ctx, 3), stride, sbuf<uint32_t>(ctx, 4),
                         sbuf<uint32_t>(ctx, 5), sbuf<uint32_t>(ctx, 6), sbuf<uint32_t>(ctx, 7), sbuf<int32_t>(ctx, 9),
                         sbuf<unsigned char>(ctx, 13));
    TRY(after_launch(ctx));
    TRY(down(ctx, status, sbuf<void>(ctx, 9), sizeof(int32_t) * nq));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // Queries beyond the in-kernel limits (a lattice of more than 256 lines per axis, a window of more than `winmax` cells:
    // the reference has no such limit, spatial.go:217-226) take the whole GPU one at a time (launch_aoi_big_query); their
    // counts join the offsets scan, their rows are copied behind the gather.
    struct Big { uint32_t qi, n; uint32_t *ids, *dists, *ivs; };
    std::vector<Big> big;
    auto free_big = [&]() { for (auto &b : big) { (void)hipFree(b.ids); (void)hipFree(b.dists); (void)hipFree(b.ivs); } };
    for (uint32_t i = 0; i < nq; i++) {
        if (status[i] != CHD_E_TOO_LARGE) continue;
        Big b{i, 0, nullptr, nullptr, nullptr};
        status[i] = launch_aoi_big_query(ctx->stream, ctx->g, queries[i], sbuf<double>(ctx, 1), sbuf<double>(ctx, 2), sbuf<uint32_t>(ctx, 3),
                                         &b.ids, &b.dists, &b.ivs, &b.n);
        if (status[i] == CHD_E_HIP) { free_big(); return fail(ctx, CHD_E_HIP, "chd_query_channel_ids: query %u: out of device memory for its cell window", i); }
        if (status[i] == CHD_OK) {
            big.push_back(b);
            HIPCHK(hipMemcpyAsync(sbuf<uint32_t>(ctx, 7) + i, &big.back().n, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));  // (the count lives in a vector element)
        }
    }
    launch_scan_u32(ctx->stream, sbuf<uint32_t>(ctx, 7), sbuf<uint32_t>(ctx, 8), nq);
    launch_csr_gather(ctx->stream, nq, stride, sbuf<uint32_t>(ctx, 7), sbuf<uint32_t>(ctx, 8), sbuf<uint32_t>(ctx, 4),
                      sbuf<uint32_t>(ctx, 5), sbuf<uint32_t>(ctx, 6), sbuf<uint32_t>(ctx, 10), sbuf<uint32_t>(ctx, 11),
                      sbuf<uint32_t>(ctx, 12), cap, ctx->g.id_start, sbuf<int32_t>(ctx, 9));
    if (hipGetLastError() != hipSuccess) { free_big(); return fail(ctx, CHD_E_HIP, "chd_query_channel_ids: launch failed"); }
    if (hipMemcpyAsync(offsets, sbuf<void>(ctx, 8), sizeof(uint32_t) * (nq + 1), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { free_big(); return fail(ctx, CHD_E_HIP, "chd_query_channel_ids: download failed"); }
    uint32_t total = offsets[nq];
    if (total > cap) { free_big(); return fail(ctx, CHD_E_CAPACITY, "chd_query_channel_ids: %u results, capacity %u", total, cap); }
    for (auto &b : big) {
        const uint32_t at = offsets[b.qi];
        (void)hipMemcpyAsync(sbuf<uint32_t>(ctx, 10) + at, b.ids, 4 * (size_t)b.n, hipMemcpyDeviceToDevice, ctx->stream);
        (void)hipMemcpyAsync(sbuf<uint32_t>(ctx, 11) + at, b.dists, 4 * (size_t)b.n, hipMemcpyDeviceToDevice, ctx->stream);
        (void)hipMemcpyAsync(sbuf<uint32_t>(ctx, 12) + at, b.ivs, 4 * (size_t)b.n, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (!big.empty()) (void)hipStreamSynchronize(ctx->stream);
    free_big();
    TRY(down(ctx, ids, sbuf<void>(ctx, 10), sizeof(uint32_t) * total));
    TRY(down(ctx, dists, sbuf<void>(ctx, 11), sizeof(uint32_t) * total));
    if (intervals_ms) TRY(down(ctx, intervals_ms, sbuf<void>(ctx, 12), sizeof(uint32_t) * total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_get_regions(chd_ctx *ctx, double *min_x, double *min_z, double *max_x, double *max_z,
                    uint32_t *channel_id, uint32_t *server_index) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (!min_x || !min_z || !max_x || !max_z || !channel_id || !server_index)
        return fail(ctx, CHD_E_INVAL, "chd_get_regions: NULL buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    const uint32_t n = ctx->g.ncell;
    for (int k = 0; k < 4; k++) TRY(ensure(ctx, k, sizeof(double) * n));
    TRY(ensure(ctx, 4, sizeof(uint32_t) * n));
    TRY(ensure(ctx, 5, sizeof(uint32_t) * n));
    launch_regions(ctx->stream, ctx->g, sbuf<double>(ctx, 0), sbuf<double>(ctx, 1), sbuf<double>(ctx, 2),
                   sbuf<double>(ctx, 3), sbuf<uint32_t>(ctx, 4), sbuf<uint32_t>(ctx, 5));
    TRY(after_launch(ctx));
    double *dst[4] = {min_x, min_z, max_x, max_z};
    for (int k = 0; k < 4; k++) TRY(down(ctx, dst[k], sbuf<void>(ctx, k), sizeof(double) * n));
    TRY(down(ctx, channel_id, sbuf<void>(ctx, 4), sizeof(uint32_t) * n));
    TRY(down(ctx, server_index, sbuf<void>(ctx, 5), sizeof(uint32_t) * n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_get_adjacent_channels(chd_ctx *ctx, const uint32_t *channel_ids, uint32_t n, uint32_t *out, uint32_t *counts) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (n && (!channel_ids || !out || !counts)) return fail(ctx, CHD_E_INVAL, "chd_get_adjacent_channels: NULL buffer");
    for (uint32_t i = 0; i < n; i++)
        if (channel_ids[i] < ctx->g.id_start || channel_ids[i] - ctx->g.id_start >= ctx->g.ncell)
            return fail(ctx, CHD_E_INVAL, "channel id %u is not a spatial channel of this grid", channel_ids[i]);
    if (!n) return CHD_OK;
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, sizeof(uint32_t) * n));
    TRY(ensure(ctx, 1, sizeof(uint32_t) * 8 * (size_t)n));
    TRY(ensure(ctx, 2, sizeof(uint32_t) * n));
    TRY(up(ctx, sbuf<void>(ctx, 0), channel_ids, sizeof(uint32_t) * n));
    HIPCHK(hipMemsetAsync(sbuf<void>(ctx, 1), 0, sizeof(uint32_t) * 8 * (size_t)n, ctx->stream));
    launch_adjacent(ctx->stream, ctx->g, sbuf<uint32_t>(ctx, 0), n, sbuf<uint32_t>(ctx, 1), sbuf<uint32_t>(ctx, 2));
    TRY(after_launch(ctx));
    TRY(down(ctx, out, sbuf<void>(ctx, 1), sizeof(uint32_t) * 8 * (size_t)n));
    TRY(down(ctx, counts, sbuf<void>(ctx, 2), sizeof(uint32_t) * n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

static int server_cells_common(chd_ctx *ctx, uint32_t server_index, int mode, uint32_t *out, uint32_t cap, uint32_t *n_out) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (!n_out || (cap && !out)) return fail(ctx, CHD_E_INVAL, "NULL buffer");
    if (server_index >= ctx->g.server_cols * ctx->g.server_rows)
        return fail(ctx, CHD_E_INVAL, "all %u grids are allocated to %u servers", ctx->g.ncell,
                    ctx->g.server_cols * ctx->g.server_rows);  // spatial.go:390-392
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, sizeof(uint32_t) * std::max(cap, 1u)));
    TRY(ensure(ctx, 1, 2 * sizeof(uint32_t)));
    launch_server_cells(ctx->stream, ctx->g, server_index, mode, sbuf<uint32_t>(ctx, 0), cap, sbuf<uint32_t>(ctx, 1),
                        sbuf<uint32_t>(ctx, 1) + 1);
    TRY(after_launch(ctx));
    uint32_t res[2] = {0, 0};
    TRY(down(ctx, res, sbuf<void>(ctx, 1), sizeof res));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *n_out = 0;
    if (res[1] == 1) return fail(ctx, CHD_E_CONFIG, "server %u: a cell falls outside the grid (GetChannelIdNoOffset error)", server_index);
    if (res[1] == 2) return fail(ctx, CHD_E_CAPACITY, "server %u: output capacity %u too small", server_index, cap);
    TRY(down(ctx, out, sbuf<void>(ctx, 0), sizeof(uint32_t) * res[0]));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *n_out = res[0];
    return CHD_OK;
}

int chd_server_channels(chd_ctx *ctx, uint32_t server_index, uint32_t *out, uint32_t cap, uint32_t *n_out) {
    return server_cells_common(ctx, server_index, 0, out, cap, n_out);
}
int chd_border_channels(chd_ctx *ctx, uint32_t server_index, uint32_t *out, uint32_t cap, uint32_t *n_out) {
    return server_cells_common(ctx, server_index, 1, out, cap, n_out);
}

// ---------------------------------------------------------------------------
// world
// ---------------------------------------------------------------------------

// WHICH FORM a world's fan-out takes — everything chd_world_create decides from the world's shape, in one place (round 5 found
// BASELINE config C with the reference's stamps 70 x slow because a 40-line `if` inside the creation picked the cell-major form for
// it; tests/test_emit_form.py walks this function over (entities / cell, history_depth, subscriber slots, flags)).
//   cell-major emit: grids up to 4096 cells (64 bitmap words per connection; also bounded by the memory of the per-item due lists:
//     ~40 B per cell x connection slot), asked for — or chosen for populous cells (>= 1024 entities per cell) unless the world keeps
//     EXACT UPDATE BUFFERS and the descriptor path can run: that path keeps sub-tick arrival offsets and decides a window that cuts
//     through a tick's arrivals in the plan / the filtered kernel, the cell-major form has no such columns and sends every such
//     subscription to the element walk (config C, every update stamped at its enqueue time: 155 ms per tick cell-major, 2.2 ms
//     connection-major, profiles/r06j_kernel_stats_c1m_aj{,_cm}.csv);
//   arrival offsets: where the descriptor path runs every tick — connection-major, one wave per connection, no per-record masks, no
//     wire positions, at most 4096 cells — on a world with history_depth;
//   tick pipelining: the descriptor path without masks / wire / exact buffers.
struct EmitForm {
    bool cm_possible, cm_emit, one_wave, off_on, pipe;
    const char *refusal;  // nullptr: fine
};
static EmitForm select_emit_form(uint64_t N, uint64_t S, uint64_t C, uint32_t flags, uint32_t history_depth) {
    EmitForm f{};
    const uint64_t n_items_max = C * ((S + 255) / 256);
    f.cm_possible = C <= 4096 && n_items_max * sizeof(WsItemG) <= (2ull << 30);
    const bool masks = (flags & CHD_WORLD_UPDATE_MASKS) != 0, wire = (flags & CHD_WORLD_WIRE) != 0;
    f.one_wave = (flags & CHD_WORLD_ONE_WAVE_EMIT) != 0;
    const bool desc_geometry = S >= 4096 || f.one_wave;  // one wave per connection: the descriptor path's shape
    const bool off_possible = history_depth && !masks && !wire && desc_geometry && C <= 4096;
    const bool cm_wanted = (flags & CHD_WORLD_CELL_MAJOR_EMIT) || (!(flags & CHD_WORLD_CONN_MAJOR_EMIT) && N / C >= 1024 && !off_possible);
    if ((flags & CHD_WORLD_CELL_MAJOR_EMIT) && !f.cm_possible) f.refusal = "cell-major emit needs a grid of at most 4096 cells (and cells x subscribers x 40 B <= 2 GiB)";
    // update masks: written by the connection-major form only (the cell-major streamers replay precomputed window masks from LDS and
    // would need the cell's own histories per window)
    else if (masks && (flags & CHD_WORLD_CELL_MAJOR_EMIT)) f.refusal = "CHD_WORLD_UPDATE_MASKS is implemented by the connection-major emit only";
    f.cm_emit = f.cm_possible && cm_wanted && !masks;
    f.off_on = history_depth && !f.cm_emit && !masks && !wire && desc_geometry && C <= 4096;
    // (exact update buffers: where the sub-tick offsets AND the cell-major filtered kernel exist — what that kernel reads beside the
    // next tick's stages then exists once per tick parity too, chd_world_create)
    f.pipe = (flags & CHD_WORLD_PIPELINE_TICKS) && !f.cm_emit && !masks && !wire && desc_geometry && (!history_depth || (f.off_on && C * S <= (1ull << 25)));
    return f;
}

int chd_world_emit_form(uint32_t max_entities, uint32_t max_subscribers, uint32_t n_cells, uint32_t world_flags, uint32_t history_depth, uint32_t *schedule) {
    if (!max_entities || !max_subscribers || !n_cells || !schedule) return CHD_E_INVAL;
    const EmitForm f = select_emit_form(max_entities, max_subscribers, n_cells, world_flags, history_depth);
    if (f.refusal) return CHD_E_INVAL;
    *schedule = (f.cm_emit ? CHD_SCHED_CELL_MAJOR : 0u) | (f.off_on ? CHD_SCHED_ARRIVAL_OFFSETS : 0u) | (f.pipe ? CHD_SCHED_PIPELINED : 0u);
    return CHD_OK;
}

int chd_world_create(chd_ctx *ctx, const chd_world_cfg *cfg) {
    if (!ctx || !cfg) return fail(ctx, CHD_E_INVAL, "chd_world_create: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    if (W.created) return fail(ctx, CHD_E_STATE, "world already created");
    if (cfg->max_entities == 0 || cfg->max_subscribers == 0)
        return fail(ctx, CHD_E_INVAL, "max_entities and max_subscribers must be positive");
    W.cfg = *cfg;
    WorldDev &d = W.d;
    const DevGrid &g = ctx->g;
    d.N = cfg->max_entities;
    d.S = cfg->max_subscribers;
    d.capq = cfg->max_interest_cells ? cfg->max_interest_cells : std::min<uint32_t>(g.ncell, 256);
    if (d.capq > ctx->lim.winmax) d.capq = ctx->lim.winmax;
    if (d.capq > 0xFFFE) return fail(ctx, CHD_E_INVAL, "max_interest_cells too large");
    if (aoi_lds_bytes(ctx->lim, d.capq) > aoi_lds_limit())
        return fail(ctx, CHD_E_CONFIG, "max_interest_cells %u on this grid needs %zu bytes of LDS per query (limit %zu)", d.capq,
                    aoi_lds_bytes(ctx->lim, d.capq), aoi_lds_limit());
    const size_t N = d.N, S = d.S, P = (size_t)d.S * d.capq, C = g.ncell;
    // chd_world_cfg.shard_channels: the UPDATE LOG — everything push_update writes — is kept per entity channel id of the whole
    // world, on every rank, instead of per entity slot (WorldDev::log_on); LN = how many logs there are
    if (cfg->shard_channels && !cfg->history_depth && !(cfg->flags & CHD_WORLD_WIRE))
        return fail(ctx, CHD_E_INVAL, "shard_channels is for region-sharded worlds with exact update buffers (history_depth > 0) or wire buffers (CHD_WORLD_WIRE)");
    if (cfg->shard_channels && (cfg->flags & CHD_WORLD_PIPELINE_TICKS))
        return fail(ctx, CHD_E_INVAL, "shard_channels: a region-sharded world (no pipelined ticks)");
    d.ce_by_chan = cfg->shard_channels ? 1u : 0u;
    // (a world with by-channel arrays is region-sharded from its creation on: chd_world_spawn / chd_tick / chd_tick_device index those
    // arrays by entity SLOT and would write past them whenever shard_channels < max_entities — they answer CHD_E_STATE)
    W.slot_mode = cfg->shard_channels ? 2 : 0;
    d.log_on = (cfg->shard_channels && cfg->history_depth) ? 1u : 0u;
    d.log_n = d.log_on ? cfg->shard_channels : d.N;
    d.log_eid0 = ctx->cfg.entity_channel_id_start ? ctx->cfg.entity_channel_id_start : 0x80000u;
    const size_t LN = d.log_n;
    TRY(walloc(ctx, &d.chan_id, N));
    TRY(walloc(ctx, &d.cell, N));
    TRY(walloc(ctx, &d.member, N));
    TRY(walloc(ctx, &d.eflags, N));
    TRY(walloc(ctx, &d.sender, LN));
    TRY(walloc(ctx, &d.hist, LN));
    TRY(walloc(ctx, &d.hist_tick, LN));
    TRY(walloc(ctx, &d.sender_prev, LN));
    TRY(walloc(ctx, &d.hist_prev, LN));
    TRY(walloc(ctx, &d.upd_mark, N));
    if (d.log_on) {
        TRY(walloc(ctx, &d.log_cell, LN, false));
        TRY(walloc(ctx, &d.log_alive, LN));
        HIPCHK(hipMemsetAsync(d.log_cell, 0xFF, sizeof(uint32_t) * LN, ctx->stream));
    }
    TRY(walloc(ctx, &d.q_mark, S));
    TRY(walloc(ctx, &d.cell_hist, C));
    TRY(walloc(ctx, &d.cell_hist_tick, C));
    TRY(walloc(ctx, &d.cell_sender, C));
    TRY(walloc(ctx, &d.cell_hist_prev, C));
    TRY(walloc(ctx, &d.cell_sender_prev, C));
    d.nblk = (C <= 4096) ? index_num_blocks(d.N) : 1;
    TRY(walloc(ctx, &d.blk_cnt, std::max(C * d.nblk + 1, 2 * C + 2)));
    TRY(walloc(ctx, &d.ce, N + 1));  // + 1: the emit kernel loads entries in adjacent pairs
    TRY(walloc(ctx, &d.ce_sprev, N));
    TRY(walloc(ctx, &d.ce8, N + 520));  // (+ 520: the filtered record kernel loads a whole 512-entry image at any cell start)
    TRY(walloc(ctx, &d.cell_usender, C));
    TRY(walloc(ctx, &d.cell_smin, C));
    TRY(walloc(ctx, &d.cell_smax, C));
    HIPCHK(hipMemset(d.cell_smax, 0xFF, sizeof(uint32_t) * C));  // until the first index build: every id may be a sender
    TRY(walloc(ctx, &d.blk_smin, C * d.nblk + 1));
    TRY(walloc(ctx, &d.blk_smax, C * d.nblk + 1));
    TRY(walloc(ctx, &d.blk_hand, C * d.nblk + 1));
    TRY(walloc(ctx, &d.cell_hand, C));
    // + 512: the pipelined emit loads a whole 512-entry column image at any cell start.  Behind the full column array: the four
    // window columns (WorldDev::wcol_*), same stride.
    d.wcol_stride = (uint32_t)((N + 520 + 63) & ~(size_t)63);
    TRY(walloc(ctx, &d.ce_chan, (size_t)(CHD_WCOLS + 1) * d.wcol_stride + 520));
    TRY(walloc(ctx, &d.cell_wcnt, (size_t)CHD_WCOLS * C));
    d.wcol_on = 0;
    TRY(walloc(ctx, &d.cell_off, C + 1));
    TRY(walloc(ctx, &d.cell_tot, C));
    TRY(walloc(ctx, &d.cell_ref, C));
    TRY(walloc(ctx, &d.active_cells, C));
    TRY(walloc(ctx, &d.n_active, 1));
    const size_t n_items_max = C * ((S + 255) / 256);
    const EmitForm form = select_emit_form(N, S, C, cfg->flags, cfg->history_depth);
    if (form.refusal) return fail(ctx, CHD_E_INVAL, "%s", form.refusal);
    // the interest bitmap (one bit per cell and connection) exists for every grid of up to 4096 cells:
    // the recipient planners use it too; larger grids fall back to searching the sorted subscription lists
    d.wb = C <= 4096 ? (uint32_t)((C + 63) / 64) : 0u;
    d.cm_emit = form.cm_emit ? 1u : 0u;
    d.one_wave_emit = form.one_wave ? 1u : 0u;
    d.sub_bits = nullptr;
    d.items = nullptr;
    if (d.wb) TRY(walloc(ctx, &d.sub_bits, S * d.wb));
    if (d.cm_emit) TRY(walloc(ctx, &d.items, n_items_max, false));
    W.plan_recipients = (cfg->flags & CHD_WORLD_HANDOVER_RECIPIENTS) != 0;
    if (cfg->flags & CHD_WORLD_SEGMENTS_ONLY) {
        if ((cfg->flags & (CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS)) || form.cm_emit)
            return fail(ctx, CHD_E_INVAL, "CHD_WORLD_SEGMENTS_ONLY: not with wire buffers, per-record masks or the cell-major emit (they are made of dense records)");
        d.seg_only = 1u;
    }
    // CHD_WORLD_FORCE_FLAGS (tests): schedule-only flags OR-ed into every world of the process, so that the parity suite can be run on them
    uint32_t wflags = cfg->flags;
    if (const char *e = getenv("CHD_WORLD_FORCE_FLAGS")) wflags |= (uint32_t)strtoul(e, nullptr, 0) & (CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_OVERLAP_DEFERRED | CHD_WORLD_GATED_OVERLAP);
    W.overlap_interest = (wflags & CHD_WORLD_OVERLAP_INTEREST) != 0;  // (exact update buffers: WorldDev::cell_max_iv is double-buffered for this)
    W.overlap_deferred = (wflags & CHD_WORLD_OVERLAP_DEFERRED) != 0 && !cfg->history_depth;
    W.gate_asked = W.overlap_interest && (wflags & CHD_WORLD_GATED_OVERLAP) != 0;
    W.gated = false;
    if (W.gate_asked) {  // (probed at the end of this function)
        TRY(walloc(ctx, &W.gate, GATE_WORDS + 32));
        d.gate_fail = W.gate + GATE_FAIL;
    }
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
        d.emit_grid = (uint32_t)std::max(prop.multiProcessorCount, 1) * 4u;  // 4 workgroups of ~37 KB LDS per CU
        // k_fanout_emit_seg: persistent waves per CU (of the 32 wave slots; the rest stays free for the next tick's stages)
        uint32_t per_cu = 8;   // (measured on config B: 8 -> 141 us, 16 -> 154 us, 32 -> 159 us; 6 -> 160 us)
        if (const char *e = getenv("CHD_EMIT_WAVES_PER_CU")) per_cu = (uint32_t)std::min(std::max(atoi(e), 1), 64);
        d.seg_waves = (uint32_t)std::max(prop.multiProcessorCount, 1) * per_cu;
        // ... the launch has twice that; how many of them take tickets is decided per tick on the device, from the records per connection
        // (k_fanout_scan: 8, 12 or 16 per CU)
        d.emit_waves = getenv("CHD_EMIT_WAVES_PER_CU") ? d.seg_waves : 2u * d.seg_waves;
        d.emit_act_t1 = 5000; d.emit_act_t2 = 1500;
        if (const char *e = getenv("CHD_EMIT_ACTIVE_THRESHOLDS")) {  // (A/B runs)
            d.emit_act_t1 = (uint32_t)strtoul(e, nullptr, 0);
            if (const char *c = strchr(e, ',')) d.emit_act_t2 = (uint32_t)strtoul(c + 1, nullptr, 0);
        }
    }
    TRY(walloc(ctx, &d.cell_tab, 2 * C));
    d.cell_cov = nullptr;  // (region-sharded worlds allocate it with the ghost room, chd_shard_halo_layout)
    d.ghost_cap = 0;
    TRY(walloc(ctx, &d.free_stack, N));
    TRY(walloc(ctx, &d.free_top, 1));
    d.limbo = nullptr;  // (chd_shard_halo_layout)
    TRY(walloc(ctx, &d.limbo_n, 2));
    TRY(walloc(ctx, &d.mig_gmax, 4));
    d.ce_view = d.ce;
    d.ce8_view = d.ce8;
    d.ce_chan_view = d.ce_chan;
    d.ce_sprev_view = d.ce_sprev;
    d.ce_sprev_stride = 0;
    d.cell_start = d.cell_off;
    d.cell_end = d.cell_off + 1;
    TRY(walloc(ctx, &d.aoi_scratch, aoi_scratch_bytes(ctx->lim) * S, false));
    TRY(walloc(ctx, &d.conn_id, S));
    TRY(walloc(ctx, &d.sub_alive, S));
    TRY(walloc(ctx, &d.sub_tick, S));
    TRY(walloc(ctx, &d.pair_cnt, S));
    TRY(walloc(ctx, &d.pair_cell, P));
    TRY(walloc(ctx, &d.pair_iv, P));
    TRY(walloc(ctx, &d.pair_last, P));
    TRY(walloc(ctx, &d.pair_flags, P));
    TRY(walloc(ctx, &d.conn_defer, S + 4));  // (+ 4: k_fanout_scan reads four connections' words with one load)
    TRY(walloc(ctx, &d.defer_list, S));
    TRY(walloc(ctx, &d.deep_list, S));
    TRY(walloc(ctx, &d.tail_ctl, 16));
    TRY(walloc(ctx, &d.n_simple, S));
    TRY(walloc(ctx, &d.emit_ticket, 8 * 32));
    TRY(walloc(ctx, &d.seg_desc, P, false));
    TRY(walloc(ctx, &d.seg_desc2, P, false));
    TRY(walloc(ctx, &d.seg_ln, P, false));
    TRY(walloc(ctx, &d.pair_rel, P));
    TRY(walloc(ctx, &d.pair_nrec, P));
    TRY(walloc(ctx, &d.rec_ub, S + 1 + 4));
    TRY(walloc(ctx, &d.rec_cnt, S));
    TRY(walloc(ctx, &W.rec_off_exact, S + 1));
    d.handovers_cap = cfg->max_handovers ? cfg->max_handovers : d.N;
    TRY(walloc(ctx, &d.handovers, d.handovers_cap, false));
    if (cfg->flags & CHD_WORLD_HANDOVER_RECIPIENTS) {
        W.ho_rcp_cap = std::min<uint64_t>((uint64_t)d.handovers_cap * S, 1ull << 25);
        TRY(walloc(ctx, &W.ho_rcp_off, (size_t)d.handovers_cap + 1));
        TRY(walloc(ctx, &W.ho_rcp_conn, W.ho_rcp_cap, false));
        TRY(walloc(ctx, &W.ho_rcp_kind, W.ho_rcp_cap, false));
        TRY(walloc(ctx, &W.ho_rcp_mask, W.ho_rcp_cap, false));
        TRY(walloc(ctx, &W.ho_rcp_own, (size_t)d.handovers_cap + 1));
        TRY(walloc(ctx, &d.ho_moved, d.handovers_cap));
    }
    // banked lists: a bank holds the worst case of the subscriber slots that map to it
    d.list_bank_cap = (uint32_t)std::min<size_t>((size_t)((S + CHD_LIST_BANKS - 1) / CHD_LIST_BANKS) * d.capq, 0x7FFFFFFFu / CHD_LIST_BANKS);
    d.unsub_cap = d.list_bank_cap * CHD_LIST_BANKS;
    TRY(walloc(ctx, &d.list_ctr, 2 * CHD_LIST_BANKS * 32));
    TRY(walloc(ctx, &d.list_bank_n, 2 * CHD_LIST_BANKS));
    TRY(walloc(ctx, &d.unsub_sub, d.unsub_cap, false));
    TRY(walloc(ctx, &d.unsub_cell, d.unsub_cap, false));
    d.newsub_cap = d.unsub_cap;
    TRY(walloc(ctx, &d.newsub_sub, d.newsub_cap, false));
    TRY(walloc(ctx, &d.newsub_cell, d.newsub_cap, false));
    TRY(walloc(ctx, &d.newsub_iv, d.newsub_cap, false));
    TRY(walloc(ctx, &d.q_status, S));
    TRY(walloc(ctx, &d.counters, CTR_COUNT));
    TRY(walloc(ctx, &d.tot64, 64 * 16));
    TRY(walloc(ctx, &d.tick_ring, (size_t)TICK_RING * 8));
    uint64_t nrec = cfg->max_records;
    if (!nrec) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        nrec = std::min<uint64_t>((uint64_t)(free_b * 0.5) / sizeof(chd_fanout_rec), 4000000000ull);
    }
    W.wire = (cfg->flags & CHD_WORLD_WIRE) != 0;
    if (W.wire && !cfg->max_records) nrec = std::min<uint64_t>(nrec / 3, 1000000000ull);  // three more 4-byte arrays per record
    const bool masks = (cfg->flags & CHD_WORLD_UPDATE_MASKS) != 0;
    if (masks && !cfg->max_records) nrec = nrec * 2 / 3;  // one more 4-byte array per record
    // tick pipelining: only with the descriptor-driven connection-major emit (its record kernel reads descriptors, offsets
    // and one column array: all of them, and the record buffer, then exist once per tick parity)
    // (not with exact update buffers: maxFanOutIntervalMs is written by the interest updates and read by the ingest)
    W.pipe_alloc = form.pipe;
    W.pipe_on = W.pipe_alloc;
    if (W.pipe_alloc && !cfg->max_records) nrec /= 2;
    d.recs_cap = nrec;
    TRY(walloc(ctx, &d.recs, nrec, false));
    if (W.pipe_alloc) {
        W.pb_n_simple[0] = d.n_simple; W.pb_rec_ub[0] = d.rec_ub; W.pb_seg_desc[0] = d.seg_desc; W.pb_seg_desc2[0] = d.seg_desc2;
        W.pb_ce_chan[0] = d.ce_chan; W.pb_recs[0] = d.recs; W.pb_ticket[0] = d.emit_ticket;
        TRY(walloc(ctx, &W.pb_ticket[1], 8 * 32));
        TRY(walloc(ctx, &W.pb_n_simple[1], S));
        TRY(walloc(ctx, &W.pb_rec_ub[1], S + 1 + 4));
        TRY(walloc(ctx, &W.pb_seg_desc[1], P, false));
        TRY(walloc(ctx, &W.pb_seg_desc2[1], P, false));
        TRY(walloc(ctx, &W.pb_ce_chan[1], (size_t)(CHD_WCOLS + 1) * d.wcol_stride + 520));
        TRY(walloc(ctx, &W.pb_recs[1], nrec, false));
        HIPCHK(hipEventCreateWithFlags(&W.ev_stages_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&W.ev_rec_sync, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&W.ev_stages_all, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&W.ev_emit_done[k], hipEventDisableTiming));
    }
    d.rec_mask = nullptr;
    d.seg_wm = nullptr;
    if (masks) {
        TRY(walloc(ctx, &d.rec_mask, nrec, false));
        TRY(walloc(ctx, &d.seg_wm, P, false));
    }
    d.rec_pos = nullptr;
    d.ce_slot = nullptr;
    d.seg_no_pos = 0;
    d.wcol_slot = nullptr;
    d.pair_desc = nullptr;
    // exact update buffers (history_depth): ChannelData.updateMsgBuffer per entity and per spatial channel
    d.deep_depth = cfg->history_depth;
    if (d.deep_depth) {
        if (d.deep_depth < CHD_HIST_BITS) return fail(ctx, CHD_E_INVAL, "history_depth must be 0 or at least %d (1024 covers the reference's 512-element buffers)", CHD_HIST_BITS);
        const size_t D = d.deep_depth;
        TRY(walloc(ctx, &d.deep_a, LN * D, false));
        TRY(walloc(ctx, &d.deep_s, LN * D, false));
        TRY(walloc(ctx, &d.deep_n, LN));
        TRY(walloc(ctx, &d.deep_len, LN));
        TRY(walloc(ctx, &d.deep_drop, LN, false));
        TRY(walloc(ctx, &d.irr_tick, LN));
        TRY(walloc(ctx, &d.cdeep_a, C * D, false));
        TRY(walloc(ctx, &d.cdeep_s, C * D, false));
        TRY(walloc(ctx, &d.cdeep_n, C));
        TRY(walloc(ctx, &d.cdeep_len, C));
        TRY(walloc(ctx, &d.cdeep_drop, C, false));
        TRY(walloc(ctx, &d.cell_irr_tick, C));
        TRY(walloc(ctx, &d.cell_irr, C));
        TRY(walloc(ctx, &d.cell_max_iv, 2 * C));
        TRY(walloc(ctx, &d.ent_max_iv, LN));
        TRY(walloc(ctx, &d.conn_deep, S + 4));
        TRY(walloc(ctx, &d.ce_slot, N + 2));
        // drop = INT64_MIN ("nothing was ever dropped"): the byte pattern 0x80 repeated is a very negative int64 as well
        HIPCHK(hipMemsetAsync(d.deep_drop, 0x80, sizeof(int64_t) * LN, ctx->stream));
        HIPCHK(hipMemsetAsync(d.cdeep_drop, 0x80, sizeof(int64_t) * std::max<size_t>(C, 32), ctx->stream));
        // Sub-tick arrival offsets (WorldDev::off_on): where the descriptor path runs every tick — connection-major, one wave per
        // connection, no per-record masks, no wire positions — the stamps of regular updates are kept per ring slot and the
        // fan-out decides on them; the other forms keep "regular = stamped with the tick's own time".  CHD_ARRIVAL_OFFSETS=0: off (A/B).
        d.off_on = form.off_on ? 1u : 0u;
        if (const char *e = getenv("CHD_ARRIVAL_OFFSETS")) if (e[0] == '0') d.off_on = 0;
        if (d.off_on) {
            d.off_stride = (uint32_t)((N + 520 + 63) & ~(size_t)63);
            TRY(walloc(ctx, &d.eoff, 2 * LN));
            TRY(walloc(ctx, &d.ce_off, (size_t)CHD_OFF_SLOTS * d.off_stride + 520));
            TRY(walloc(ctx, &d.cell_orng, C * CHD_OFF_SLOTS));
            TRY(walloc(ctx, &d.cell_ooff, 2 * C));
            TRY(walloc(ctx, &d.n_filt, S));
            TRY(walloc(ctx, &d.filt_desc, P, false));
            TRY(walloc(ctx, &d.filt_desc2, P, false));
            TRY(walloc(ctx, &d.filt_ln, P, false));
            TRY(walloc(ctx, &d.filt_win, P * CHD_FILT_WINS, false));
            // the per-cell lists of the cell-major filtered kernel: 4 B per (cell, connection) — where that is too much the
            // connection-major kernel takes the filtered descriptors (CHD_FILT_CELL_MAJOR=0: A/B)
            d.fcm_on = C * S <= (1ull << 25) ? 1u : 0u;  // (32 B per entry: at most 1 GiB)
            if (const char *e = getenv("CHD_FILT_CELL_MAJOR")) if (e[0] == '0') d.fcm_on = 0;
            if (d.fcm_on) {
                // the cells' entries in the order of the tick's arrivals (k_cell_arrange, the launch that also takes the cells' offset
                // ranges) and the part of a window inside the tick's own arrivals as a run of the column (k_fanout_emit_filt_cm);
                // CHD_SORT_ARRIVALS=0 keeps the entries in slot order (A/B runs)
                TRY(walloc(ctx, &d.cell_sorted, C));
                TRY(walloc(ctx, &d.cell_fcnt, C * 32));
                TRY(walloc(ctx, &d.cell_flist, 2 * C * S, false));
                TRY(walloc(ctx, &d.filt_items, P / 16 + C + 1, false));  // (work items of >= 16 descriptors: FC_DESCS)
                TRY(walloc(ctx, &d.filt_nitems, 32));
                d.filt_target = 0;
                if (const char *e = getenv("CHD_FILT_ITEMS_TARGET")) d.filt_target = (uint32_t)strtoul(e, nullptr, 0);  // (A/B runs)
            }
        }
        d.prev_ns = -1;
        if (W.pipe_alloc && !(d.off_on && d.fcm_on)) W.pipe_alloc = W.pipe_on = false;  // (CHD_ARRIVAL_OFFSETS=0 / CHD_FILT_CELL_MAJOR=0: A/B runs)
        if (W.pipe_alloc) {
            W.pipe_exact = true;
            W.pb_ce8[0] = d.ce8; W.pb_ce_off[0] = d.ce_off; W.pb_cell_sorted[0] = d.cell_sorted; W.pb_filt_nitems[0] = d.filt_nitems;
            W.pb_pair_nrec[0] = d.pair_nrec; W.pb_rec_cnt[0] = d.rec_cnt; W.pb_cell_flist[0] = d.cell_flist; W.pb_filt_items[0] = d.filt_items;
            W.pb_filt_win[0] = d.filt_win;
            TRY(walloc(ctx, &W.pb_ce8[1], N + 520));
            TRY(walloc(ctx, &W.pb_ce_off[1], (size_t)CHD_OFF_SLOTS * d.off_stride + 520));
            TRY(walloc(ctx, &W.pb_cell_sorted[1], C));
            TRY(walloc(ctx, &W.pb_filt_nitems[1], 32));
            TRY(walloc(ctx, &W.pb_pair_nrec[1], P));
            TRY(walloc(ctx, &W.pb_rec_cnt[1], S));
            TRY(walloc(ctx, &W.pb_cell_flist[1], 2 * C * S, false));
            TRY(walloc(ctx, &W.pb_filt_items[1], P / 16 + C + 1, false));
            TRY(walloc(ctx, &W.pb_filt_win[1], P * CHD_FILT_WINS, false));
        }
    }
    if (W.wire) {
        WireDev &x = W.x;
        TRY(walloc(ctx, &d.rec_pos, nrec, false));
        if (!d.ce_slot) TRY(walloc(ctx, &d.ce_slot, N + 2));
        TRY(walloc(ctx, &x.rec_woff, nrec, false));
        TRY(walloc(ctx, &x.rec_wtag, nrec, false));
        TRY(walloc(ctx, &x.seg_fast, P));
        TRY(walloc(ctx, &x.conn_slow, S));
        TRY(walloc(ctx, &x.trash, S * 4 * 16));
        x.fast_ok = d.capq <= 512 ? 1u : 0u;
        if (const char *e = getenv("CHD_WIRE_FAST")) x.fast_ok = (e[0] != '0' && x.fast_ok) ? 1u : 0u;  // (A/B runs)
        x.stride[0] = ((cfg->wire_max_update_len ? cfg->wire_max_update_len : 128u) + 15u) & ~15u;
        x.stride[1] = ((cfg->wire_max_full_len ? cfg->wire_max_full_len : 1024u) + 15u) & ~15u;
        // entity payloads: by slot — or, region-sharded (shard_channels), by channel id on every rank: the host feeds every rank the
        // whole world's update payloads as it feeds it the positions, so a neighbour's ghost entry finds its payload here
        const size_t PN = cfg->shard_channels ? cfg->shard_channels : N;
        x.npay = (uint32_t)PN;
        for (int k = 0; k < 2; k++) {
            // (+ 128: k_wire_copy_fast reads 80 bytes of a slot whatever its stride)
            TRY(walloc(ctx, &x.pay_ent[k], PN * x.stride[k] + 128));
            TRY(walloc(ctx, &x.pay_cell[k], C * x.stride[k] + 128));
            TRY(walloc(ctx, &x.len_ent[k], PN));
            TRY(walloc(ctx, &x.len_cell[k], C));
        }
        x.merge = masks ? 1u : 0u;
        if (x.merge) {
            // merged updates: the UPDATE payloads of the last CHD_HIST_BITS ticks per channel, and the Any type_urls
            TRY(walloc(ctx, &x.ring_ent, PN * CHD_HIST_BITS * x.stride[0], false));
            TRY(walloc(ctx, &x.ring_cell, C * CHD_HIST_BITS * x.stride[0], false));
            TRY(walloc(ctx, &x.rlen_ent, PN * CHD_HIST_BITS));
            TRY(walloc(ctx, &x.rlen_cell, C * CHD_HIST_BITS));
            TRY(walloc(ctx, &x.url[0], 256));
            TRY(walloc(ctx, &x.url[1], 256));
        }
        TRY(walloc(ctx, &x.url[2], 256));
        TRY(walloc(ctx, &x.pay_objref, PN * x.stride[0]));
        TRY(walloc(ctx, &x.len_objref, PN));
        TRY(walloc(ctx, &x.conn_wlen, S + 1));
        x.conn_woff = x.conn_wlen;
        TRY(walloc(ctx, &x.conn_npk, S));
        TRY(walloc(ctx, &x.n_dropped, 8 + 64));
        // the descriptor-driven stream builder (k_wire_layout_img): one image of every cell's messages per payload kind
        x.img_on = 1u;
        if (const char *e = getenv("CHD_WIRE_IMAGES")) if (e[0] == '0') x.img_on = 0;  // (A/B runs, tests of the record path)
        if (cfg->shard_channels) x.img_on = 0;  // (region-sharded: the cell images index table positions and window columns the ghost tables do not have; the record path)
        if (x.img_on) {
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            // window columns in wire worlds (partially updating ticks): nine more update images per cell, if they fit
            x.img_ncol = 1u + CHD_WCOLS;
            for (int k = 0; k < 2; k++) {
                // worst case: every channel's payload at its slot size + the three nested headers, images padded to 16 bytes
                // (merge mode: an update message carries up to three ticks' updates behind the Any's type url; the images that are
                // built are the ones the tick's descriptors ask for — room for four of the nine per cell, checked on the device)
                const uint64_t per = (k == 0 && x.merge) ? 3ull * x.stride[0] + 256u + 48u : x.stride[k] + 32u;
                const uint64_t one = (uint64_t)(N + C) * per + 16ull * C + 4096ull;
                uint64_t cap = one * (k == 0 ? (x.merge ? 4u : x.img_ncol) : 1u);
                if (k == 0 && !(cap < (1ull << 31) && cap < free_b / 8)) { x.img_ncol = 1; cap = one; if (x.merge) cap = 1ull << 40; /* (no merge images) */ }
                x.img_ok[k] = (cap < (1ull << 31) && cap < free_b / 8) ? 1u : 0u;  // (a descriptor's source offset has 31 bits)
                if (!x.img_ok[k]) continue;
                const size_t ni = (size_t)(k == 0 ? x.img_ncol : 1u) * C;
                TRY(walloc(ctx, &x.img[k], cap, false));
                x.img_cap[k] = cap;
                TRY(walloc(ctx, &x.img_off[k], ni + 1));
                TRY(walloc(ctx, &x.img_len[k], ni));
                TRY(walloc(ctx, &x.img_own[k], ni));
                TRY(walloc(ctx, &x.img_bad[k], ni));
                TRY(walloc(ctx, &x.img_end[k], k == 0 ? (size_t)x.img_ncol * d.wcol_stride + 520 : N + 2));
                if (k == 0) TRY(walloc(ctx, &x.img_need, ni + C));
            }
            if (!x.img_ok[0] || (x.merge && x.img_ncol == 1u)) x.img_on = 0;  // (the update images are the point; full states alone are not worth the path)
            if (x.img_on) {
                // (see WorldDev::seg_no_pos) — and with them the window columns of partially updating ticks stay available
                d.seg_no_pos = 1;
                TRY(walloc(ctx, &d.wcol_slot, (size_t)(CHD_WCOLS + 1) * d.wcol_stride + 520, false));
                TRY(walloc(ctx, &d.pair_desc, P, false));
            }
            x.dpad = C <= 65536 ? 32u : 1u;
            TRY(walloc(ctx, &x.cell_dcnt, C * x.dpad + 1));
            TRY(walloc(ctx, &x.conn_ndesc, S));
            TRY(walloc(ctx, &x.conn_key, S));
            TRY(walloc(ctx, &x.conn_rank, S));
            TRY(walloc(ctx, &x.rank_ndesc, S + 1));
            TRY(walloc(ctx, &x.slow_list, S));
            TRY(walloc(ctx, &x.cp_ticket, 32));
        }
        x.ncell = (uint32_t)C;
        x.npos = (uint32_t)N + 2u;
        if (const char *e = getenv("CHD_DEBUG_POISON")) if (e[0] == '1') {  // (debugging: unwritten per-record words are recognisable)
            HIPCHK(hipMemsetAsync(d.rec_pos, 0x7F, sizeof(uint32_t) * nrec, ctx->stream));
            HIPCHK(hipMemsetAsync(x.rec_woff, 0x7F, sizeof(uint32_t) * nrec, ctx->stream));
            HIPCHK(hipMemsetAsync(x.rec_wtag, 0x7F, sizeof(uint32_t) * nrec, ctx->stream));
        }
    }
    launch_free_stack_init(ctx->stream, d);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (W.gate_asked) TRY(gate_probe(ctx));
    if (const char *e = test_hook_env("CHD_TEST_DROP_GATE_RAISE")) W.test_drop_raise = strtoull(e, nullptr, 0);
    W.created = true;
    return CHD_OK;
}

#define NEED_WORLD()                                                              \
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");                       \
    if (!ctx->w.created) return fail(ctx, CHD_E_STATE, "chd_world_create has not been called")

int chd_world_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *idx, const uint32_t *chan_id, const double *x,
                    const double *z, const uint32_t *flags, const uint32_t *sender) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!chan_id || !x || !z) return fail(ctx, CHD_E_INVAL, "chd_world_spawn: NULL buffer");
    if (!idx && n > ctx->w.d.N) return fail(ctx, CHD_E_INVAL, "chd_world_spawn: n > max_entities");
    if (idx)
        for (uint32_t i = 0; i < n; i++)
            if (idx[i] >= ctx->w.d.N) return fail(ctx, CHD_E_INVAL, "entity slot %u out of range", idx[i]);
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (ctx->w.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_world_spawn on a world whose slots are library-managed (chd_shard_spawn)");
    ctx->w.slot_mode = 1;
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, 4 * (size_t)n)); TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(ensure(ctx, 2, 8 * (size_t)n)); TRY(ensure(ctx, 3, 8 * (size_t)n));
    TRY(ensure(ctx, 4, 4 * (size_t)n)); TRY(ensure(ctx, 5, 4 * (size_t)n));
    if (idx) TRY(up(ctx, sbuf<void>(ctx, 0), idx, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), chan_id, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 2), x, 8 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 3), z, 8 * (size_t)n));
    if (flags) TRY(up(ctx, sbuf<void>(ctx, 4), flags, 4 * (size_t)n));
    if (sender) TRY(up(ctx, sbuf<void>(ctx, 5), sender, 4 * (size_t)n));
    launch_spawn(ctx->stream, ctx->g, ctx->w.d, n, idx ? sbuf<uint32_t>(ctx, 0) : nullptr, sbuf<uint32_t>(ctx, 1),
                 sbuf<double>(ctx, 2), sbuf<double>(ctx, 3), flags ? sbuf<uint32_t>(ctx, 4) : nullptr,
                 sender ? sbuf<uint32_t>(ctx, 5) : nullptr, ctx->ring.cur_tick);
    launch_group_locks(ctx->stream, ctx->w.d);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    {
        World &W = ctx->w;
        if (W.live.size() != W.d.N) W.live.assign(W.d.N, 0);
        for (uint32_t i = 0; i < n; i++) {
            uint8_t &l = W.live[idx ? idx[i] : i];
            if (!l) { l = 1; W.n_live++; }
        }
        W.full_streak = 0;
    }
    return CHD_OK;
}

int chd_world_despawn(chd_ctx *ctx, uint32_t n, const uint32_t *idx) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!idx) return fail(ctx, CHD_E_INVAL, "chd_world_despawn: NULL idx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 0), idx, 4 * (size_t)n));
    launch_despawn(ctx->stream, ctx->w.d, n, sbuf<uint32_t>(ctx, 0));
    launch_group_locks(ctx->stream, ctx->w.d);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    {
        World &W = ctx->w;
        for (uint32_t i = 0; i < n; i++)
            if (idx[i] < W.live.size() && W.live[idx[i]]) { W.live[idx[i]] = 0; W.n_live--; }
    }
    return CHD_OK;
}

int chd_world_set_entity_flags(chd_ctx *ctx, uint32_t n, const uint32_t *idx, const uint32_t *flags) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!idx || !flags) return fail(ctx, CHD_E_INVAL, "chd_world_set_entity_flags: NULL buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, 4 * (size_t)n)); TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 0), idx, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), flags, 4 * (size_t)n));
    launch_set_flags(ctx->stream, ctx->w.d, n, sbuf<uint32_t>(ctx, 0), sbuf<uint32_t>(ctx, 1));
    launch_group_locks(ctx->stream, ctx->w.d);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_world_set_entity_groups(chd_ctx *ctx, uint32_t n, const uint32_t *idx, const uint32_t *group) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!idx || !group) return fail(ctx, CHD_E_INVAL, "chd_world_set_entity_groups: NULL buffer");
    World &W = ctx->w;
    WorldDev &d = W.d;
    for (uint32_t i = 0; i < n; i++)
        if (idx[i] >= d.N) return fail(ctx, CHD_E_INVAL, "entity slot %u out of range", idx[i]);
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (W.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_world_set_entity_groups on a region-sharded world: groups are given as lists keyed by channel id there (chd_shard_set_handover_lists)");
    TRY(bind(ctx));
    if (W.group_id.empty() || d.grp_exact) W.group_id.assign(d.N, 0u);  // (after chd_world_set_handover_lists: start over)
    d.grp_exact = 0;
    for (uint32_t i = 0; i < n; i++) W.group_id[idx[i]] = group[i];
    // CSR of the groups (rare control-plane call: rebuilt on the host, O(N log N))
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < d.N; i++)
        if (W.group_id[i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return W.group_id[a] < W.group_id[b]; });
    std::vector<uint32_t> of(d.N, CHD_INVALID), off;
    for (size_t k = 0; k < order.size(); k++) {
        if (k == 0 || W.group_id[order[k]] != W.group_id[order[k - 1]]) off.push_back((uint32_t)k);
        of[order[k]] = (uint32_t)off.size() - 1u;
    }
    const uint32_t G = (uint32_t)off.size();
    off.push_back((uint32_t)order.size());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux_stream));  // (pipelined ticks: k_ingest reads the group arrays on the stage streams)
    HIPCHK(hipStreamSynchronize(ctx->aux2_stream));
    for (void *&b : W.grp_buf) { if (b) HIPCHK(hipFree(b)); b = nullptr; }
    d.n_groups = 0;
    d.grp_of = d.grp_off = d.grp_mem = d.grp_locked = nullptr;
    if (G) {
        const size_t sz[4] = {4 * (size_t)d.N, 4 * (size_t)(G + 1), 4 * std::max<size_t>(order.size(), 1), 4 * (size_t)G};
        const void *src[4] = {of.data(), off.data(), order.data(), nullptr};
        for (int k = 0; k < 4; k++) {
            HIPCHK(hipMalloc(&W.grp_buf[k], sz[k]));
            if (src[k]) HIPCHK(hipMemcpy(W.grp_buf[k], src[k], sz[k], hipMemcpyHostToDevice));
        }
        d.grp_of = (uint32_t *)W.grp_buf[0]; d.grp_off = (uint32_t *)W.grp_buf[1];
        d.grp_mem = (uint32_t *)W.grp_buf[2]; d.grp_locked = (uint32_t *)W.grp_buf[3];
        d.n_groups = G;
        launch_group_locks(ctx->stream, d);
        TRY(after_launch(ctx));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return CHD_OK;
}

int chd_world_set_handover_lists(chd_ctx *ctx, uint32_t n_lists, const uint32_t *list_off, const uint32_t *list_members,
                                 uint32_t n, const uint32_t *idx, const uint32_t *list_of) {
    NEED_WORLD();
    if (n_lists && !list_off) return fail(ctx, CHD_E_INVAL, "chd_world_set_handover_lists: NULL list_off");
    if (n && (!idx || !list_of)) return fail(ctx, CHD_E_INVAL, "chd_world_set_handover_lists: NULL idx / list_of");
    World &W = ctx->w;
    WorldDev &d = W.d;
    const uint32_t total = n_lists ? list_off[n_lists] : 0u;
    if (n_lists && list_off[0] != 0) return fail(ctx, CHD_E_INVAL, "chd_world_set_handover_lists: list_off[0] must be 0");
    for (uint32_t k = 0; k < n_lists; k++)
        if (list_off[k + 1] < list_off[k]) return fail(ctx, CHD_E_INVAL, "chd_world_set_handover_lists: list_off decreases at %u", k);
    if (total && !list_members) return fail(ctx, CHD_E_INVAL, "chd_world_set_handover_lists: NULL list_members");
    for (uint32_t q = 0; q < total; q++)
        if (list_members[q] >= d.N) return fail(ctx, CHD_E_INVAL, "handover list member %u out of range", list_members[q]);
    for (uint32_t i = 0; i < n; i++) {
        if (idx[i] >= d.N) return fail(ctx, CHD_E_INVAL, "entity slot %u out of range", idx[i]);
        if (list_of[i] != CHD_NO_HANDOVER_LIST && list_of[i] >= n_lists) return fail(ctx, CHD_E_INVAL, "entity slot %u: list %u of %u", idx[i], list_of[i], n_lists);
    }
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (W.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_world_set_handover_lists on a region-sharded world: the lists are keyed by channel id there (chd_shard_set_handover_lists)");
    TRY(bind(ctx));
    std::vector<uint32_t> of(d.N, CHD_INVALID);
    for (uint32_t i = 0; i < n; i++) of[idx[i]] = list_of[i] == CHD_NO_HANDOVER_LIST ? CHD_INVALID : list_of[i];
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux_stream));
    HIPCHK(hipStreamSynchronize(ctx->aux2_stream));
    for (void *&b : W.grp_buf) { if (b) HIPCHK(hipFree(b)); b = nullptr; }
    W.group_id.clear();
    d.n_groups = 0;
    d.grp_exact = 0;
    d.grp_of = d.grp_off = d.grp_mem = d.grp_locked = nullptr;
    if (n_lists) {
        const size_t sz[4] = {4 * (size_t)d.N, 4 * (size_t)(n_lists + 1), 4 * std::max<size_t>(total, 1), 4 * (size_t)n_lists};
        const void *src[4] = {of.data(), list_off, total ? list_members : nullptr, nullptr};
        for (int k = 0; k < 4; k++) {
            HIPCHK(hipMalloc(&W.grp_buf[k], sz[k]));
            if (src[k]) HIPCHK(hipMemcpy(W.grp_buf[k], src[k], k == 2 ? 4 * (size_t)total : sz[k], hipMemcpyHostToDevice));
            else HIPCHK(hipMemset(W.grp_buf[k], 0, sz[k]));
        }
        d.grp_of = (uint32_t *)W.grp_buf[0]; d.grp_off = (uint32_t *)W.grp_buf[1];
        d.grp_mem = (uint32_t *)W.grp_buf[2]; d.grp_locked = (uint32_t *)W.grp_buf[3];
        d.n_groups = n_lists;
        d.grp_exact = 1;
    }
    return CHD_OK;
}

static int subs_common(chd_ctx *ctx, uint32_t n, const uint32_t *slot, const uint32_t *conn, int add) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (add && !conn) return fail(ctx, CHD_E_INVAL, "chd_subs_add: NULL conn_id");
    if (!slot && n > ctx->w.d.S) return fail(ctx, CHD_E_INVAL, "n > max_subscribers");
    if (slot)
        for (uint32_t i = 0; i < n; i++)
            if (slot[i] >= ctx->w.d.S) return fail(ctx, CHD_E_INVAL, "subscriber slot %u out of range", slot[i]);
    if (add)
        for (uint32_t i = 0; i < n; i++)
            if (conn[i] & CHD_REC_FULL) return fail(ctx, CHD_E_INVAL, "connection id %u exceeds 31 bits (settings.go:90)", conn[i]);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, 4 * (size_t)n)); TRY(ensure(ctx, 1, 4 * (size_t)n));
    if (slot) TRY(up(ctx, sbuf<void>(ctx, 0), slot, 4 * (size_t)n));
    if (add) TRY(up(ctx, sbuf<void>(ctx, 1), conn, 4 * (size_t)n));
    launch_subs_add(ctx->stream, ctx->w.d, n, slot ? sbuf<uint32_t>(ctx, 0) : nullptr, sbuf<uint32_t>(ctx, 1), add);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_subs_add(chd_ctx *ctx, uint32_t n, const uint32_t *slot, const uint32_t *conn_id) {
    return subs_common(ctx, n, slot, conn_id, 1);
}
int chd_subs_remove(chd_ctx *ctx, uint32_t n, const uint32_t *slot) { return subs_common(ctx, n, slot, nullptr, 0); }

int chd_subs_set_options(chd_ctx *ctx, int64_t now_ns, uint32_t n, const chd_sub_options *opts, uint8_t *should_send, int32_t *status) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!opts) return fail(ctx, CHD_E_INVAL, "chd_subs_set_options: NULL options");
    WorldDev &d = ctx->w.d;
    for (uint32_t i = 0; i < n; i++) {
        const chd_sub_options &o = opts[i];
        if (o.slot >= d.S) return fail(ctx, CHD_E_INVAL, "option %u: subscriber slot %u out of range", i, o.slot);
        if (o.channel < ctx->g.id_start || o.channel - ctx->g.id_start >= ctx->g.ncell)
            return fail(ctx, CHD_E_INVAL, "option %u: %u is not a spatial channel of this grid", i, o.channel);
        if ((o.set & CHD_SUBOPT_ACCESS) && o.data_access > CHD_ACCESS_WRITE) return fail(ctx, CHD_E_INVAL, "option %u: DataAccess %u", i, o.data_access);
        if ((o.set & CHD_SUBOPT_FIELD_MASK) && o.data_field_mask > 0xFFu) return fail(ctx, CHD_E_INVAL, "option %u: data_field_mask has 8 bits", i);
        if ((o.set & CHD_SUBOPT_INTERVAL) && o.fanout_interval_ms == 0)
            return fail(ctx, CHD_E_INVAL, "option %u: fan-out interval 0 makes the reference's tickData spin forever", i);
    }
    // one wave per connection, its records in call order: stable grouping by slot
    std::vector<uint32_t> order(n), grp;
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return opts[a].slot < opts[b].slot; });
    for (uint32_t r = 0; r < n; r++)
        if (r == 0 || opts[order[r]].slot != opts[order[r - 1]].slot) grp.push_back(r);
    const uint32_t ngrp = (uint32_t)grp.size();
    grp.push_back(n);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, sizeof(chd_sub_options) * (size_t)n));
    TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(ensure(ctx, 2, 4 * (size_t)(ngrp + 1)));
    TRY(ensure(ctx, 3, (size_t)n));
    TRY(ensure(ctx, 4, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 0), opts, sizeof(chd_sub_options) * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), order.data(), 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 2), grp.data(), 4 * (size_t)(ngrp + 1)));
    launch_subs_set_options(ctx->stream, ctx->g, d, sbuf<chd_sub_options>(ctx, 0), sbuf<uint32_t>(ctx, 1), sbuf<uint32_t>(ctx, 2), ngrp,
                            now_ns, sbuf<uint8_t>(ctx, 3), sbuf<int32_t>(ctx, 4));
    TRY(after_launch(ctx));
    std::vector<int32_t> st(n);
    std::vector<uint8_t> ss(n);
    TRY(down(ctx, ss.data(), sbuf<void>(ctx, 3), n));
    TRY(down(ctx, st.data(), sbuf<void>(ctx, 4), 4 * (size_t)n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int rc = CHD_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (should_send) should_send[i] = ss[i];
        if (status) status[i] = st[i];
        if (st[i] == CHD_E_CAPACITY) rc = CHD_E_CAPACITY;
    }
    if (rc) return fail(ctx, rc, "chd_subs_set_options: a connection's subscription list is full (max_interest_cells)");
    return CHD_OK;
}

int chd_subs_get_options(chd_ctx *ctx, uint32_t slot, uint8_t *data_access, uint8_t *skip_self, uint32_t *n_out) {
    NEED_WORLD();
    if (!data_access || !skip_self || !n_out) return fail(ctx, CHD_E_INVAL, "chd_subs_get_options: NULL buffer");
    WorldDev &d = ctx->w.d;
    if (slot >= d.S) return fail(ctx, CHD_E_INVAL, "subscriber slot %u out of range", slot);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, d.capq));
    TRY(ensure(ctx, 1, d.capq));
    uint32_t cnt = 0;
    TRY(down(ctx, &cnt, d.pair_cnt + slot, 4));
    launch_subs_get_options(ctx->stream, d, slot, sbuf<uint8_t>(ctx, 0), sbuf<uint8_t>(ctx, 1));
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    TRY(down(ctx, data_access, sbuf<void>(ctx, 0), cnt));
    TRY(down(ctx, skip_self, sbuf<void>(ctx, 1), cnt));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *n_out = cnt;
    return CHD_OK;
}

// ---- the device-side tick, in stages; caller holds the mutex ----
static int tick_begin(chd_ctx *ctx, int64_t now_ns) {
    World &W = ctx->w;
    if (now_ns < W.last_now) return fail(ctx, CHD_E_INVAL, "now_ns went backwards (%lld < %lld)", (long long)now_ns, (long long)W.last_now);
    W.last_now = now_ns;
    TickRing &r = ctx->ring;
    for (int j = CHD_HIST_BITS - 1; j > 0; j--) r.t[j] = r.t[j - 1];
    r.t[0] = now_ns;
    if (r.n < CHD_HIST_BITS) r.n++;
    r.cur_tick++;
    return CHD_OK;
}

static int check_queries(chd_ctx *ctx, const chd_tick_in *in) {
    if (in->n_queries && !in->queries) return fail(ctx, CHD_E_INVAL, "tick: NULL queries");
    if (in->n_queries > ctx->w.d.S) return fail(ctx, CHD_E_INVAL, "tick: n_queries > max_subscribers (one interest update per connection per tick)");
    return CHD_OK;
}

// `in` carries DEVICE pointers.
static int tick_locked(chd_ctx *ctx, const chd_tick_in *in) {
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (in->n_updates && (!in->upd_x || !in->upd_z)) return fail(ctx, CHD_E_INVAL, "tick: NULL update positions");
    if (!in->upd_idx && in->n_updates > d.N) return fail(ctx, CHD_E_INVAL, "tick: n_updates > max_entities");
    if (!in->upd_idx && in->n_update_rounds > 1) return fail(ctx, CHD_E_INVAL, "tick: update rounds need upd_idx");
    TRY(check_queries(ctx, in));
    if (in->n_cell_updates && (!in->cell_upd_channel || !in->cell_upd_sender)) return fail(ctx, CHD_E_INVAL, "tick: NULL cell updates");
    if (W.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_tick on a region-sharded world: use chd_shard_ingest/import/fanout");
    if (!d.deep_depth && (in->upd_arrival_ns || in->cell_upd_arrival_ns || in->n_update_rounds > 1))
        return fail(ctx, CHD_E_STATE, "tick: arrival stamps / update rounds need a world with history_depth > 0 (the 32-tick bit ring stamps a batch with its tick)");
    if (in->n_update_rounds) {
        if (!in->upd_round_off) return fail(ctx, CHD_E_INVAL, "tick: NULL upd_round_off");
        // (the duplicate-slot mark of a round is (tick, round & 0xFF): rounds r and r + 256 would look like a repeated slot)
        if (in->n_update_rounds > 256) return fail(ctx, CHD_E_INVAL, "tick: more than 256 update rounds (updates of ONE channel between two ticks)");
        if (in->upd_round_off[0] != 0 || in->upd_round_off[in->n_update_rounds] != in->n_updates)
            return fail(ctx, CHD_E_INVAL, "tick: upd_round_off must run from 0 to n_updates");
        for (uint32_t r = 0; r < in->n_update_rounds; r++)
            if (in->upd_round_off[r + 1] < in->upd_round_off[r]) return fail(ctx, CHD_E_INVAL, "tick: upd_round_off decreases at %u", r);
    }
    const bool chained = ctx->chain_prev;
    TRY(tick_begin(ctx, in->now_ns));
    TickRing &r = ctx->ring;
    d.prev_ns = r.n > 1 ? r.t[1] : -1;  // (sub-tick arrival offsets: a regular update arrived after the previous tick)
    // CHD_WORLD_PIPELINE_TICKS: the record-writing kernel of this tick runs on `stream` while the NEXT tick's stages
    // (ingest ... plan, commit, the deferred subscriptions, epilogue) run on the second stream: they write the other parity's
    // copies of what that kernel reads.  Stream order: stages(t) -> records(t) on `stream`; stages(t+1) after stages(t)
    // and after records(t-1), whose buffers they reuse.
    // Which connection-major form this tick takes.  The descriptor path (k_fanout_plan_seg + k_fanout_emit_seg + the deferred
    // launch) pays when the windows are plain copies of the cells' channel-id columns, i.e. when EVERY entity of a cell has
    // an update in every tick a window reaches back to; an entity that skipped a tick sends its cell's subscriptions to the
    // deferred (filtering) launch, and a world where most do is faster in the one-launch form (measured on config B with 98 %
    // / 50 % of the entities updating per tick: emit stage 245 / 235 us against 201 / 187 us).  So: the descriptor path once
    // every live entity has sent an update in this tick and in the one before.  Both forms write the same records, segment
    // order and state (the parity tests run both; the full-size tests cross from one to the other after the first tick);
    // worlds that ASK for the one-wave geometry (CHD_WORLD_ONE_WAVE_EMIT) always take the descriptor path.
    W.full_streak = (W.n_live && in->n_updates >= W.n_live && in->n_update_rounds <= 1) ? std::min(W.full_streak + 1u, 1u << 20) : 0u;
    // Partially updating worlds keep the descriptor path where the WINDOW COLUMNS exist (WorldDev::wcol_*: per cell the entities
    // updated within the last 1..4 ticks, compacted once per tick by k_window_columns): a window over exactly those ticks is
    // then a plain copy again.  (Not in wire mode: its records carry cell-table positions.)
    d.wcol_on = (W.full_streak < 2u && d.wcol_stride && (!d.rec_pos || (d.seg_no_pos && W.x.img_ncol > 1u && W.x.img_ok[0])) && !d.cm_emit && !d.rec_mask) ? 1u : 0u;
    d.seg_off = (!d.one_wave_emit && W.full_streak < 2u && !d.wcol_on) ? 1u : 0u;
    if (const char *e = getenv("CHD_WINDOW_COLUMNS")) if (e[0] == '0') {  // (A/B runs: the per-tick choice of round 2)
        d.wcol_on = 0;
        d.seg_off = (!d.one_wave_emit && W.full_streak < 2u) ? 1u : 0u;
    }
    if (d.off_on) {  // sub-tick arrival offsets: always the descriptor path, whose filtered descriptors take every partial window
        d.wcol_on = 0;
        d.seg_off = 0;
        if (!fanout_seg_path(d)) return fail(ctx, CHD_E_STATE, "tick: CHD_EMIT_PIPELINED=0 on a world that keeps arrival offsets (set CHD_ARRIVAL_OFFSETS=0 as well)");
    }
    const bool pipe = W.pipe_on && fanout_seg_path(d);
    d.late_tot = (pipe && d.off_on) ? 1u : 0u;  // (the filtered kernel's count reaches the tick's row by k_filt_fold: the epilogue runs beside it)
    W.last_desc = fanout_seg_path(d);
    hipStream_t st = ctx->stream;
    hipStream_t bs = pipe ? ctx->aux_stream : st;
    const uint32_t par = r.cur_tick & 1u;
    if (pipe) {
        d.n_simple = W.pb_n_simple[par]; d.rec_ub = W.pb_rec_ub[par]; d.seg_desc = W.pb_seg_desc[par];
        d.seg_desc2 = W.pb_seg_desc2[par]; d.ce_chan = W.pb_ce_chan[par]; d.recs = W.pb_recs[par]; d.emit_ticket = W.pb_ticket[par];
        if (W.pipe_exact) {
            d.ce8 = W.pb_ce8[par]; d.ce_off = W.pb_ce_off[par]; d.cell_sorted = W.pb_cell_sorted[par]; d.filt_nitems = W.pb_filt_nitems[par];
            d.pair_nrec = W.pb_pair_nrec[par]; d.rec_cnt = W.pb_rec_cnt[par]; d.cell_flist = W.pb_cell_flist[par]; d.filt_items = W.pb_filt_items[par];
            d.filt_win = W.pb_filt_win[par];
        }
        if (chained) HIPCHK(hipStreamWaitEvent(bs, W.ev_emit_done[par], 0));
        else {  // something else was enqueued on `stream` since the last tick (or this is the first one): after all of it
            HIPCHK(hipEventRecord(W.ev_rec_sync, st));
            HIPCHK(hipStreamWaitEvent(bs, W.ev_rec_sync, 0));
        }
    }
    d.ce_view = d.ce;
    d.ce8_view = d.ce8;
    d.ce_chan_view = d.ce_chan;
    d.ce_sprev_view = d.ce_sprev;
    d.ce_sprev_stride = 0;
    d.cell_start = d.cell_off;
    d.cell_end = d.cell_off + 1;
    // (sampled record-kernel pairs: the ticks in between record nothing and say so — ev_overlap bit 8)
    const bool prof_skip = ctx->prof_depth > 0 && ctx->prof_kernel_only && ctx->prof_every > 1 && r.cur_tick % ctx->prof_every != 0;
    if (prof_skip) ctx->ev_overlap[r.cur_tick % (uint32_t)ctx->prof_depth] = 8;
    const bool prof = ctx->prof_depth > 0 && !prof_skip;
    hipEvent_t *ev = prof ? &ctx->ev[(size_t)(r.cur_tick % (uint32_t)ctx->prof_depth) * EV_PER_TICK] : nullptr;
    // CHD_WORLD_OVERLAP_INTEREST: the interest updates touch subscriptions only, ingest + index build entities
    // only, so the two can run side by side on two streams and join before the fan-out plan.  (Not with handover
    // recipients: those are planned on the subscriptions as they were BEFORE this tick's interest updates.)
    // Pipelined ticks always do, on a third stream (their stages are latency-bound and run beside an HBM-saturating kernel).
    const bool overlap = (W.overlap_interest || pipe) && !W.plan_recipients && in->n_queries > 0;
    const bool gated = overlap && W.gated && (pipe || !(W.overlap_deferred && fanout_seg_path(d)));
    // stage events: the serial schedule marks every stage boundary; the pipelined one only the begin and end of the stage
    // stream's work (a timed event between two small kernels costs ~5 us of idle stream) — stage_times() reports that
    // span as stage 0
    // chd_set_profiling_scope(CHD_PROF_RECORD_KERNEL): nothing but the pair around the record kernel (throughput runs)
    const bool prof_ends = prof && !ctx->prof_kernel_only;
    const bool prof_stages = prof_ends && !pipe;
    if (prof) {
        ctx->ev_overlap[r.cur_tick % (uint32_t)ctx->prof_depth] = (overlap ? 1 : 0) | (pipe ? 2 : 0) | (prof_ends ? 0 : 4);
        if (prof_ends) HIPCHK(hipEventRecord(ev[0], bs));
    }
    {
        if (overlap) {
            hipStream_t ax = pipe ? ctx->aux2_stream : ctx->aux_stream;
            // fork: the second stream starts after everything enqueued on this one so far — or, gated and directly behind a gated
            // tick, after that tick's epilogue said so (no event on the tick's stream)
            if (gated && ctx->gchain_prev) launch_gate_wait(ax, d, W.gate + GATE_EPI, W.gate_epi, true);
            else {
                HIPCHK(hipEventRecord(ctx->ev_fork, bs));
                HIPCHK(hipStreamWaitEvent(ax, ctx->ev_fork, 0));
            }
            if (prof_stages) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 1], ax));
            if (gated) W.gate_top++;  // (the join: a one-wave kernel behind the interest launch raises the flag, k_gate_raise)
            // (CHD_TEST_DROP_GATE_RAISE=k, tests only: the k-th gated tick of the world never raises its flag — the time-out path)
            const bool drop_raise = gated && W.test_drop_raise && W.gate_top == W.test_drop_raise;
            launch_aoi_interest(ax, ctx->g, ctx->lim, d, in->queries, in->n_queries, in->query_sub, in->spot_x, in->spot_z,
                                in->spot_dist, in->now_ns, r.cur_tick, (gated && !drop_raise) ? W.gate + GATE_TOP : nullptr, W.gate_top);
            if (prof_stages) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 2], ax));
            if (!gated) HIPCHK(hipEventRecord(ctx->ev_join, ax));
        }
        {
            // one ingest launch per round of updates (a channel's r-th update of this tick: chd_tick_in.upd_round_off)
            const uint32_t one[2] = {0u, in->n_updates};
            const uint32_t nr = in->n_update_rounds ? in->n_update_rounds : 1u;
            const uint32_t *off = in->n_update_rounds ? in->upd_round_off : one;
            for (uint32_t k = 0; k < nr; k++) {
                const uint32_t a = off[k], n = off[k + 1] - a;
                launch_ingest(bs, ctx->g, d, n, in->upd_idx ? in->upd_idx + a : nullptr, in->upd_x + a, in->upd_z + a,
                              in->upd_sender ? in->upd_sender + a : nullptr, r.cur_tick, in->upd_arrival_ns ? in->upd_arrival_ns + a : nullptr,
                              in->now_ns, k);
            }
        }
        launch_cell_updates(bs, ctx->g, d, in->n_cell_updates, in->cell_upd_channel, in->cell_upd_sender, r.cur_tick,
                            in->cell_upd_arrival_ns, in->now_ns);
        if (W.plan_recipients) {
            // who receives each handover's message: on the subscriptions as they are NOW, before this tick's
            // interest updates (the reference sends from Notify, spatial.go:776-857)
            launch_handover_recipients_count(bs, ctx->g, d, W.ho_rcp_off, W.ho_rcp_own);
            launch_scan_u32_inplace_dev(bs, W.ho_rcp_off, d.handovers_cap, d.counters + CTR_HANDOVERS);
            launch_handover_recipients_fill(bs, ctx->g, d, W.ho_rcp_off, W.ho_rcp_conn, W.ho_rcp_kind, W.ho_rcp_mask, W.ho_rcp_cap);
        }
        if (prof_stages) HIPCHK(hipEventRecord(ev[1], bs));
        const bool gate_in_index = launch_index_build(bs, ctx->g, d, r.cur_tick, (overlap && gated) ? W.gate + GATE_TOP : nullptr, W.gate_top, in->now_ns);
        if (d.wcol_on && fanout_seg_path(d)) launch_window_columns(bs, ctx->g, d);
        launch_cell_offsets(bs, ctx->g, d);
        if (prof_stages) HIPCHK(hipEventRecord(ev[2], bs));
        if (overlap && gated) {  // join: every group of the interest launch complete (the index build's last launch waited for it where it could)
            if (!gate_in_index) launch_gate_wait(bs, d, W.gate + GATE_TOP, W.gate_top, false);
        }
        else if (overlap) HIPCHK(hipStreamWaitEvent(bs, ctx->ev_join, 0));
        else
            launch_aoi_interest(bs, ctx->g, ctx->lim, d, in->queries, in->n_queries, in->query_sub, in->spot_x, in->spot_z,
                                in->spot_dist, in->now_ns, r.cur_tick);
    }
    if (prof_stages) HIPCHK(hipEventRecord(ev[3], bs));
    launch_fanout_plan(bs, ctx->g, d, in->now_ns, r);
    if (prof_stages) HIPCHK(hipEventRecord(ev[4], bs));
    if (pipe) {
        // the record kernel needs the plan and the scan; the deferred subscriptions (+ the state commit) and the epilogue finish beside it
        HIPCHK(hipEventRecord(W.ev_stages_done, bs));
        HIPCHK(hipStreamWaitEvent(st, W.ev_stages_done, 0));
        if (prof) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 4], st));
        launch_fanout_emit_main(st, ctx->g, d, in->now_ns, r);
        if (prof && !d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));
        launch_fanout_emit_filt(st, ctx->g, d);
        if (prof && d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));  // (emit_main_us: both record-writing kernels)
        if (prof_ends) HIPCHK(hipEventRecord(ev[5], st));
        HIPCHK(hipEventRecord(W.ev_emit_done[par], st));
        if (W.gated) launch_fanout_tail(bs, ctx->g, d, in->now_ns, r, r.cur_tick % TICK_RING, W.gate + GATE_EPI, ++W.gate_epi);
        else launch_fanout_tail(bs, ctx->g, d, in->now_ns, r, r.cur_tick % TICK_RING);
        if (prof_ends) HIPCHK(hipEventRecord(ev[4], bs));
        // ... and whatever is enqueued on `stream` after this tick comes after ALL of it
        HIPCHK(hipEventRecord(W.ev_stages_all, bs));
        HIPCHK(hipStreamWaitEvent(st, W.ev_stages_all, 0));
        if (d.late_tot) launch_filt_fold(st, d, r.cur_tick % TICK_RING);  // (behind the epilogue, which wrote the row, and the filtered kernel)
    } else if (W.overlap_deferred && fanout_seg_path(d) && !d.deep_depth) {
        // CHD_WORLD_OVERLAP_DEFERRED: the filtering launch (the subscriptions the plan deferred + the state commit) and the epilogue
        // on the second stream, beside the record kernel — the pair a pipelined tick already runs side by side: the record kernel
        // reads descriptors, offsets and columns and writes records, nothing the other two touch.  Measured on config B
        // (profiles/r03zz_overlap_deferred_ab.json): NOT a gain there, the record kernel takes 149 instead of 141 us and the tick
        // 0.273 instead of 0.264 ms; off unless the world asks for it.
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
        if (prof) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 4], st));
        launch_fanout_emit_main(st, ctx->g, d, in->now_ns, r);
        if (prof) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));
        launch_fanout_tail(ctx->aux_stream, ctx->g, d, in->now_ns, r, r.cur_tick % TICK_RING);
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
        if (prof_ends) HIPCHK(hipEventRecord(ev[5], st));
    } else {
        if (prof) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 4], st));
        launch_fanout_emit_main(st, ctx->g, d, in->now_ns, r);
        if (prof && !d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));
        launch_fanout_emit_filt(st, ctx->g, d);
        if (prof && d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));  // (emit_main_us: both record-writing kernels)
        // (the deferred subscriptions, the element walk and the epilogue: one launch)
        if (W.gated) launch_fanout_tail(st, ctx->g, d, in->now_ns, r, r.cur_tick % TICK_RING, W.gate + GATE_EPI, ++W.gate_epi);
        else launch_fanout_tail(st, ctx->g, d, in->now_ns, r, r.cur_tick % TICK_RING);
        if (prof_ends) HIPCHK(hipEventRecord(ev[5], st));
    }
    TRY(after_launch(ctx));
    ctx->chain = pipe;
    ctx->gchain = W.gated && (pipe || !(W.overlap_deferred && fanout_seg_path(d)));  // (this tick's epilogue raised the flag)
    W.last_nq = in->n_queries;
    W.ticked = true;
    W.wire_built = false;
    return CHD_OK;
}

int chd_tick_device(chd_ctx *ctx, const chd_tick_in *d_in) {
    NEED_WORLD();
    if (!d_in) return fail(ctx, CHD_E_INVAL, "chd_tick_device: NULL input");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    return tick_locked(ctx, d_in);
}

// Banked list (bank b = entries [b*bank_cap, b*bank_cap + bank_n[b])) -> dense, banks in order.  One block per bank.
__global__ void __launch_bounds__(256) k_list_pack(const uint32_t *bank_n, uint32_t bank_cap, const uint32_t *a,
                                                   const uint32_t *b, const uint32_t *c, uint32_t *da, uint32_t *db,
                                                   uint32_t *dc) {
    const uint32_t bank = blockIdx.x;
    size_t off = 0;
    for (uint32_t i = 0; i < bank; i++) off += bank_n[i];
    const uint32_t n = bank_n[bank];
    const size_t src = (size_t)bank * bank_cap;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        da[off + i] = a[src + i];
        db[off + i] = b[src + i];
        if (c) dc[off + i] = c[src + i];
    }
}

#define SEG_SCAN_ITEMS 4
__device__ __forceinline__ unsigned long long seg_wave_incl_scan(unsigned long long v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// chd_tick_segments_begin, between k_segments' two passes: the per-connection segment and explicit-record counts (inside the block,
// at o_cnt / o_exp) become offsets, and — every size of the tick being known on the device now — the block's header: counts, the
// tick's row of the history ring, where each section starts.  One workgroup; the header goes to the device block (what the next
// kernels read) and to the page-locked copy (what the host reads before it sizes the one copy of the block).
__global__ void __launch_bounds__(1024) k_seg_scan(WorldDev w, uint32_t ring_slot, unsigned char *blk, unsigned long long *hdr_host, uint32_t o_cnt,
                                                   uint32_t o_exp, unsigned long long o_var, unsigned long long ncol, uint32_t nq, unsigned long long cap,
                                                   const unsigned long long *gate_fail, unsigned long long tick_no) {
    __shared__ unsigned long long wtot[2][16];
    __shared__ unsigned long long carry_s[2];
    uint32_t *cnt = (uint32_t *)(blk + o_cnt);
    unsigned long long *exp = (unsigned long long *)(blk + o_exp);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = w.S;
    if (threadIdx.x < 2) carry_s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024 * SEG_SCAN_ITEMS) {
        const uint32_t i0 = base + threadIdx.x * SEG_SCAN_ITEMS;
        uint32_t a[SEG_SCAN_ITEMS];
        unsigned long long b[SEG_SCAN_ITEMS], sa = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < SEG_SCAN_ITEMS; k++) {
            a[k] = (i0 + k < n) ? cnt[i0 + k] : 0u;
            b[k] = (i0 + k < n) ? exp[i0 + k] : 0ull;
            sa += a[k]; sb += b[k];
        }
        const unsigned long long ia = seg_wave_incl_scan(sa), ib = seg_wave_incl_scan(sb);
        if (lane == 63) { wtot[0][wave] = ia; wtot[1][wave] = ib; }
        __syncthreads();
        unsigned long long ra = carry_s[0] + ia - sa, rb = carry_s[1] + ib - sb;
        for (uint32_t k = 0; k < wave; k++) { ra += wtot[0][k]; rb += wtot[1][k]; }
#pragma unroll
        for (int k = 0; k < SEG_SCAN_ITEMS; k++) {
            if (i0 + k < n) { cnt[i0 + k] = (uint32_t)ra; exp[i0 + k] = rb; }
            ra += a[k]; rb += b[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_s[0] = ra; carry_s[1] = rb; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { cnt[n] = (uint32_t)carry_s[0]; exp[n] = carry_s[1]; }
    if (threadIdx.x < 64) {
        const uint32_t t = threadIdx.x;
        unsigned long long v = 0;
        const uint64_t *row = w.tick_ring + (size_t)ring_slot * 8;
        const unsigned long long nseg = carry_s[0], nexp = carry_s[1];
        const unsigned long long nho = min((unsigned long long)(uint32_t)row[2], (unsigned long long)w.handovers_cap),
                                 nun = min((unsigned long long)(uint32_t)row[4], (unsigned long long)w.unsub_cap),
                                 nnew = min((unsigned long long)(uint32_t)row[5], (unsigned long long)w.newsub_cap);
        auto al = [](unsigned long long x) { return (x + 255ull) & ~255ull; };
        const unsigned long long off_col = o_var, off_seg = off_col + al(4 * ncol), off_rec = off_seg + al(sizeof(chd_fanout_segment) * nseg),
                                 off_ho = off_rec + al(sizeof(chd_fanout_rec) * nexp), off_un = off_ho + al(sizeof(chd_handover_rec) * nho),
                                 off_new = off_un + 2 * al(4 * nun), total = off_new + 3 * al(4 * nnew);
        if (t == SEGH_NSEG) v = nseg;
        else if (t == SEGH_NEXP) v = nexp;
        else if (t >= SEGH_ROW && t < SEGH_ROW + 8) v = row[t - SEGH_ROW];
        else if (t == SEGH_GATE_FAIL) v = gate_fail ? *gate_fail : 0ull;
        else if (t == SEGH_REC_UB) v = w.rec_ub[w.S];
        else if (t == SEGH_TOTAL) v = total;
        else if (t == SEGH_OFF_COL) v = off_col;
        else if (t == SEGH_OFF_SEG) v = off_seg;
        else if (t == SEGH_OFF_REC) v = off_rec;
        else if (t == SEGH_OFF_HO) v = off_ho;
        else if (t == SEGH_OFF_UN) v = off_un;
        else if (t == SEGH_OFF_UN + 1) v = off_un + al(4 * nun);
        else if (t >= SEGH_OFF_NEW && t < SEGH_OFF_NEW + 3) v = off_new + (t - SEGH_OFF_NEW) * al(4 * nnew);
        else if (t == SEGH_NCOL) v = ncol;
        else if (t == SEGH_NQ) v = nq;
        else if (t == SEGH_CAP) v = cap;
        else if (t == SEGH_NHO) v = nho;
        else if (t == SEGH_NUN) v = nun;
        else if (t == SEGH_NNEW) v = nnew;
        else if (t == SEGH_FITS) v = total <= cap ? 1ull : 0ull;
        else if (t == SEGH_TICK) v = tick_no;
        if (t < SEGH_WORDS) {
            ((unsigned long long *)blk)[t] = v;
            hdr_host[t] = v;
        }
    }
}

// chd_tick_segments_begin, last kernel: what the NEXT tick's kernels overwrite — the entity-channel columns, the query status, the
// handover records, the two banked lists (packed as k_list_pack does) — copied into the tick's block at the header's offsets
__global__ void __launch_bounds__(256) k_seg_stage(WorldDev w, unsigned char *blk, uint32_t o_qst) {
    const unsigned long long *hdr = (const unsigned long long *)blk;
    if (!hdr[SEGH_FITS]) return;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    {
        const size_t n = hdr[SEGH_NCOL], n4 = n >> 2;  // (both sides 16-byte aligned)
        const uint4 *src = (const uint4 *)w.ce_chan_view;
        uint4 *dst = (uint4 *)(blk + hdr[SEGH_OFF_COL]);
        for (size_t i = tid; i < n4; i += nth) dst[i] = src[i];
        for (size_t i = (n4 << 2) + tid; i < n; i += nth) ((uint32_t *)dst)[i] = w.ce_chan_view[i];
    }
    {
        const size_t n = hdr[SEGH_NQ];
        int32_t *dst = (int32_t *)(blk + o_qst);
        for (size_t i = tid; i < n; i += nth) dst[i] = w.q_status[i];
    }
    {
        const size_t n = hdr[SEGH_NHO] * (sizeof(chd_handover_rec) / 8);
        const uint2 *src = (const uint2 *)w.handovers;
        uint2 *dst = (uint2 *)(blk + hdr[SEGH_OFF_HO]);
        for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
    }
    for (uint32_t job = blockIdx.x; job < 2u * CHD_LIST_BANKS; job += gridDim.x) {
        const uint32_t row = job / CHD_LIST_BANKS, bank = job % CHD_LIST_BANKS;
        const uint32_t *bank_n = w.list_bank_n + row * CHD_LIST_BANKS;
        size_t off = 0;
        for (uint32_t i = 0; i < bank; i++) off += bank_n[i];
        const uint32_t n = bank_n[bank];
        const size_t src = (size_t)bank * w.list_bank_cap;
        const size_t lim = row ? hdr[SEGH_NNEW] : hdr[SEGH_NUN];  // (the counts the header was sized with: the banks hold no more)
        const uint32_t *a = row ? w.newsub_sub : w.unsub_sub, *b = row ? w.newsub_cell : w.unsub_cell;
        uint32_t *da = (uint32_t *)(blk + hdr[row ? SEGH_OFF_NEW : SEGH_OFF_UN]), *db = (uint32_t *)(blk + hdr[(row ? SEGH_OFF_NEW : SEGH_OFF_UN) + 1]);
        uint32_t *dc = row ? (uint32_t *)(blk + hdr[SEGH_OFF_NEW + 2]) : nullptr;
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            if (off + i >= lim) break;
            da[off + i] = a[src + i];
            db[off + i] = b[src + i];
            if (dc) dc[off + i] = w.newsub_iv[src + i];
        }
    }
}

// Stage times of one profiled tick.  With the interest stage on the second stream its time is its own
// begin/end pair; the tick total is always first-to-last event on the main stream.
static void stage_times(chd_ctx *ctx, uint32_t tick, chd_tick_stats &s) {
    const uint32_t slot = tick % (uint32_t)ctx->prof_depth;
    hipEvent_t *ev = &ctx->ev[(size_t)slot * EV_PER_TICK];
    float ms = 0;
    if (ctx->ev_overlap[slot] & 8) {  // a tick between two sampled ones: nothing was recorded
        for (int k = 0; k < CHD_N_STAGES; k++) s.stage_us[k] = 0.f;
        s.total_us = s.emit_main_us = 0.f;
        return;
    }
    if (ctx->ev_overlap[slot] & 4) {
        // CHD_PROF_RECORD_KERNEL: only the record kernel's own pair was taken
        for (int k = 0; k < CHD_N_STAGES; k++) s.stage_us[k] = 0.f;
        s.total_us = 0.f;
        (void)hipEventElapsedTime(&ms, ev[CHD_N_STAGES + 4], ev[CHD_N_STAGES + 3]);
        s.emit_main_us = ms * 1000.f;
        return;
    }
    if (ctx->ev_overlap[slot] & 2) {
        // pipelined tick: stage 0 = everything on the stage stream (ingest ... plan, commit, deferred subscriptions, epilogue),
        // stage 4 = from there to the end of the record kernel (includes waiting for the previous tick's record kernel)
        for (int k = 0; k < CHD_N_STAGES; k++) s.stage_us[k] = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[4]);
        s.stage_us[0] = ms * 1000.f;
        ms = 0;
        (void)hipEventElapsedTime(&ms, ev[4], ev[5]);
        s.stage_us[4] = ms * 1000.f;
    } else {
        for (int k = 0; k < CHD_N_STAGES; k++) {
            ms = 0;
            (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
            s.stage_us[k] = ms * 1000.f;
        }
        if (ctx->ev_overlap[slot] & 1) {
            ms = 0;
            (void)hipEventElapsedTime(&ms, ev[CHD_N_STAGES + 1], ev[CHD_N_STAGES + 2]);
            s.stage_us[2] = ms * 1000.f;
        }
    }
    ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[CHD_N_STAGES]);
    s.total_us = ms * 1000.f;
    ms = 0;
    (void)hipEventElapsedTime(&ms, ev[CHD_N_STAGES + 4], ev[CHD_N_STAGES + 3]);
    s.emit_main_us = ms * 1000.f;
}

// the small lists of a tick — handover records, per-query status, the packed unsub / new-sub lists — enqueued for download once the
// counts are known (out->n_* filled from the tick's row of the history ring); rc becomes CHD_E_CAPACITY where a list was cut
static int fetch_lists_enqueue(chd_ctx *ctx, chd_tick_out *out, int &rc) {
    World &W = ctx->w;
    WorldDev &d = W.d;
    hipStream_t st = ctx->stream;
    if (out->handovers) {
        uint32_t n = std::min(out->n_handovers, out->handovers_cap);
        if (n < out->n_handovers) { out->overflow |= OVF_HANDOVER; rc = CHD_E_CAPACITY; }
        TRY(down(ctx, out->handovers, d.handovers, sizeof(chd_handover_rec) * n));
    }
    if (out->query_status) TRY(down(ctx, out->query_status, d.q_status, sizeof(int32_t) * W.last_nq));
    const bool want_un = out->unsub_sub && out->unsub_channel && out->n_unsubs;
    const bool want_new = out->newsub_sub && out->newsub_channel && out->n_newsubs;
    if (want_un || want_new) {
        // the device keeps both lists in banks (WorldDev::list_ctr): pack them into dense staging first
        const size_t nu = want_un ? out->n_unsubs : 0, nn = want_new ? out->n_newsubs : 0, need = 2 * nu + 3 * nn;
        if (W.list_dense_cap < need) {
            if (W.list_dense) HIPCHK(hipFree(W.list_dense));
            W.list_dense = nullptr;
            W.list_dense_cap = need + need / 4 + 1024;
            HIPCHK(hipMalloc((void **)&W.list_dense, W.list_dense_cap * sizeof(uint32_t)));
        }
        uint32_t *du = W.list_dense, *dn = W.list_dense + 2 * nu;
        if (want_un)
            hipLaunchKernelGGL(k_list_pack, dim3(CHD_LIST_BANKS), dim3(256), 0, st, d.list_bank_n, d.list_bank_cap,
                               (const uint32_t *)d.unsub_sub, (const uint32_t *)d.unsub_cell, (const uint32_t *)nullptr,
                               du, du + nu, (uint32_t *)nullptr);
        if (want_new)
            hipLaunchKernelGGL(k_list_pack, dim3(CHD_LIST_BANKS), dim3(256), 0, st, d.list_bank_n + CHD_LIST_BANKS,
                               d.list_bank_cap, (const uint32_t *)d.newsub_sub, (const uint32_t *)d.newsub_cell,
                               (const uint32_t *)d.newsub_iv, dn, dn + nn, dn + 2 * nn);
        if (want_un) {
            uint32_t n = std::min(out->n_unsubs, out->unsub_cap);
            if (n < out->n_unsubs) { out->overflow |= OVF_UNSUB; rc = CHD_E_CAPACITY; }
            TRY(down(ctx, out->unsub_sub, du, 4 * (size_t)n));
            TRY(down(ctx, out->unsub_channel, du + nu, 4 * (size_t)n));
        }
        if (want_new) {
            uint32_t n = std::min(out->n_newsubs, out->newsub_cap);
            if (n < out->n_newsubs) { out->overflow |= OVF_NEWSUB; rc = CHD_E_CAPACITY; }
            TRY(down(ctx, out->newsub_sub, dn, 4 * (size_t)n));
            TRY(down(ctx, out->newsub_channel, dn + nn, 4 * (size_t)n));
            if (out->newsub_interval_ms) TRY(down(ctx, out->newsub_interval_ms, dn + 2 * nn, 4 * (size_t)n));
        }
    }
    return CHD_OK;
}

// the tick's counts from its row of the history ring (written by the epilogue)
static void fetch_counts_from_row(chd_ctx *ctx, const uint64_t *ringrow, chd_tick_out *out, uint32_t *pairs) {
    const WorldDev &d = ctx->w.d;
    out->n_handovers = std::min((uint32_t)ringrow[2], d.handovers_cap);
    out->n_locked_aborts = (uint32_t)ringrow[3];
    out->n_unsubs = std::min((uint32_t)ringrow[4], d.unsub_cap);
    out->n_newsubs = std::min((uint32_t)ringrow[5], d.newsub_cap);
    out->overflow = (uint32_t)(ringrow[7] & 0xFFFFFFFFu);
    out->history_overflow = (uint32_t)(ringrow[7] >> 32);
    *pairs = (uint32_t)ringrow[6];
}

static int fetch_locked(chd_ctx *ctx, chd_tick_out *out) {
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick to fetch");
    if (d.seg_only && (out->records || out->record_masks)) return fail(ctx, CHD_E_STATE, "CHD_WORLD_SEGMENTS_ONLY: the world writes no dense records (chd_tick_fetch_segments)");
    hipStream_t st = ctx->stream;
    uint32_t ctr[CTR_COUNT] = {0};
    uint64_t ringrow[8];
    // per-connection record counts (sum of its subscriptions' segments), then exact offsets
    hipLaunchKernelGGL(k_rec_cnt, dim3((d.S + 3) / 4), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_widen, dim3((d.S + 255) / 256), dim3(256), 0, st, d.rec_cnt, W.rec_off_exact, d.S);
    launch_scan_u64_inplace(st, W.rec_off_exact, d.S);
    // the per-tick counters were folded into the history ring by the tick's epilogue
    TRY(down(ctx, ringrow, d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    uint64_t total = 0;
    TRY(down(ctx, &total, W.rec_off_exact + d.S, sizeof total));
    unsigned long long gate_fails = 0;
    TRY(gate_poll_begin(ctx, &gate_fails));
    HIPCHK(hipStreamSynchronize(st));
    gate_poll_end(ctx, gate_fails);
    ctr[CTR_HANDOVERS] = (uint32_t)ringrow[2];
    ctr[CTR_LOCKED] = (uint32_t)ringrow[3];
    ctr[CTR_UNSUBS] = (uint32_t)ringrow[4];
    ctr[CTR_NEWSUBS] = (uint32_t)ringrow[5];
    ctr[CTR_PAIRS] = (uint32_t)ringrow[6];
    ctr[CTR_OVERFLOW] = (uint32_t)(ringrow[7] & 0xFFFFFFFFu);
    ctr[CTR_HIST_OVERFLOW] = (uint32_t)(ringrow[7] >> 32);
    out->n_handovers = std::min(ctr[CTR_HANDOVERS], d.handovers_cap);
    out->n_locked_aborts = ctr[CTR_LOCKED];
    out->n_unsubs = std::min(ctr[CTR_UNSUBS], d.unsub_cap);
    out->overflow = ctr[CTR_OVERFLOW];
    out->history_overflow = ctr[CTR_HIST_OVERFLOW];
    out->n_records = total;
    int rc = CHD_OK;
    out->n_newsubs = std::min(ctr[CTR_NEWSUBS], d.newsub_cap);
    TRY(fetch_lists_enqueue(ctx, out, rc));
    if (out->conn_rec_off) TRY(down(ctx, out->conn_rec_off, W.rec_off_exact, sizeof(uint64_t) * (d.S + 1)));
    if (out->conn_rec_cnt) TRY(down(ctx, out->conn_rec_cnt, d.rec_cnt, sizeof(uint32_t) * d.S));
    if (out->records && total) {
        if (total > out->records_cap) {
            out->overflow |= OVF_RECORDS;
            rc = CHD_E_CAPACITY;
        } else {
            if (W.recs_dense_cap < total) {
                if (W.recs_dense) HIPCHK(hipFree(W.recs_dense));
                W.recs_dense = nullptr;
                W.recs_dense_cap = total + total / 4;
                // (+ half as much again for the masks of a CHD_WORLD_UPDATE_MASKS world)
                HIPCHK(hipMalloc((void **)&W.recs_dense, W.recs_dense_cap * (sizeof(chd_fanout_rec) + (d.rec_mask ? 4 : 0))));
            }
            uint32_t *dmask = (d.rec_mask && out->record_masks) ? (uint32_t *)(W.recs_dense + W.recs_dense_cap) : nullptr;
            hipLaunchKernelGGL(k_pack_records, dim3((d.S + 3) / 4), dim3(256), 0, st, d, W.rec_off_exact, W.recs_dense, dmask);
            TRY(after_launch(ctx));
            TRY(down(ctx, out->records, W.recs_dense, sizeof(chd_fanout_rec) * total));
            if (dmask) TRY(down(ctx, out->record_masks, dmask, sizeof(uint32_t) * total));
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    if (out->overflow && rc == CHD_OK) rc = CHD_E_CAPACITY;
    // stats
    chd_tick_stats &s = ctx->stats;
    s.n_records = total;
    uint64_t ub = 0;
    HIPCHK(hipMemcpy(&ub, d.rec_ub + d.S, sizeof ub, hipMemcpyDeviceToHost));
    s.n_record_upper_bound = ub;
    s.n_handovers = out->n_handovers;
    s.n_unsubs = out->n_unsubs;
    s.n_pairs = ctr[CTR_PAIRS];
    if (ctx->prof_depth > 0) stage_times(ctx, ctx->ring.cur_tick, s);
    if (rc == CHD_E_CAPACITY) return fail(ctx, rc, "tick output truncated (overflow mask 0x%x)", out->overflow);
    return rc;
}

int chd_tick_fetch(chd_ctx *ctx, chd_tick_out *out) {
    NEED_WORLD();
    if (!out) return fail(ctx, CHD_E_INVAL, "chd_tick_fetch: NULL output");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    return fetch_locked(ctx, out);
}

int chd_tick_fetch_segments(chd_ctx *ctx, chd_segments_out *out) {
    NEED_WORLD();
    if (!out || !out->conn_seg_off || !out->conn_rec_off) return fail(ctx, CHD_E_INVAL, "chd_tick_fetch_segments: NULL output");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick to fetch");
    if (d.capq > 32u * SEG_BITMAP_WORDS) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_fetch_segments: max_interest_cells %u > %u", d.capq, 32u * SEG_BITMAP_WORDS);
    if ((uint64_t)d.S * d.capq > 0xFFFFFFFFull) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_fetch_segments: more than 2^32 subscriptions");
    hipStream_t st = ctx->stream;
    if (!W.seg_cnt) {
        TRY(walloc(ctx, &W.seg_cnt, (size_t)d.S + 1));
        TRY(walloc(ctx, &W.seg_exp, (size_t)d.S + 1));
    }
    hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, W.last_desc ? 1 : 0, 0, W.seg_cnt, (unsigned long long *)W.seg_exp,
                       (chd_fanout_segment *)nullptr, (chd_fanout_rec *)nullptr);
    launch_scan_u32_inplace(st, W.seg_cnt, d.S);
    launch_scan_u64_inplace(st, W.seg_exp, d.S);
    TRY(after_launch(ctx));
    uint32_t nseg = 0;
    uint64_t nexp = 0, ringrow[8];
    TRY(down(ctx, &nseg, W.seg_cnt + d.S, sizeof nseg));
    TRY(down(ctx, &nexp, W.seg_exp + d.S, sizeof nexp));
    TRY(down(ctx, ringrow, d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(st));
    // the columns a descriptor may point into: the own cell tables (+ the neighbours' border entities on a sharded rank)
    const uint64_t ncol = !W.last_desc ? 0ull : d.wcol_on ? (uint64_t)(CHD_WCOLS + 1) * d.wcol_stride : (uint64_t)d.N + d.ghost_cap;
    out->n_segments = nseg;
    out->n_explicit = nexp;
    out->n_columns = ncol;
    out->n_records = ringrow[0];
    if (nseg > out->segments_cap || nexp > out->records_cap || ncol > out->columns_cap || (nseg && !out->segments) || (nexp && !out->records) ||
        (ncol && !out->columns))
        return fail(ctx, CHD_E_CAPACITY, "chd_tick_fetch_segments: %u segments, %llu explicit records, %llu column entries needed", nseg,
                    (unsigned long long)nexp, (unsigned long long)ncol);
    if (W.seg_stage_cap < nseg) {
        if (W.seg_stage) HIPCHK(hipFree(W.seg_stage));
        W.seg_stage = nullptr;
        W.seg_stage_cap = nseg + nseg / 4 + 1024;
        HIPCHK(hipMalloc((void **)&W.seg_stage, W.seg_stage_cap * sizeof(chd_fanout_segment)));
    }
    if (W.seg_rec_stage_cap < nexp) {
        if (W.seg_rec_stage) HIPCHK(hipFree(W.seg_rec_stage));
        W.seg_rec_stage = nullptr;
        W.seg_rec_stage_cap = nexp + nexp / 4 + 1024;
        HIPCHK(hipMalloc((void **)&W.seg_rec_stage, W.seg_rec_stage_cap * sizeof(chd_fanout_rec)));
    }
    hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, W.last_desc ? 1 : 0, 1, W.seg_cnt, (unsigned long long *)W.seg_exp,
                       W.seg_stage, W.seg_rec_stage);
    TRY(after_launch(ctx));
    TRY(down(ctx, out->conn_seg_off, W.seg_cnt, sizeof(uint32_t) * ((size_t)d.S + 1)));
    TRY(down(ctx, out->conn_rec_off, W.seg_exp, sizeof(uint64_t) * ((size_t)d.S + 1)));
    TRY(down(ctx, out->segments, W.seg_stage, sizeof(chd_fanout_segment) * (size_t)nseg));
    TRY(down(ctx, out->records, W.seg_rec_stage, sizeof(chd_fanout_rec) * nexp));
    if (ncol) TRY(down(ctx, out->columns, d.ce_chan_view, sizeof(uint32_t) * ncol));
    unsigned long long gate_fails = 0;
    TRY(gate_poll_begin(ctx, &gate_fails));
    HIPCHK(hipStreamSynchronize(st));
    gate_poll_end(ctx, gate_fails);
    if (ringrow[7] & 0xFFFFFFFFull) return fail(ctx, CHD_E_CAPACITY, "tick output truncated (overflow mask 0x%x)", (uint32_t)ringrow[7]);
    return CHD_OK;
}

int chd_tick_digest(chd_ctx *ctx, chd_records_digest *total, uint64_t *conn_sum) {
    NEED_WORLD();
    if (!total) return fail(ctx, CHD_E_INVAL, "chd_tick_digest: NULL output");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick to digest");
    if (d.seg_only) return fail(ctx, CHD_E_STATE, "CHD_WORLD_SEGMENTS_ONLY: the world writes no dense records");
    hipStream_t st = ctx->stream;
    TRY(ensure(ctx, 14, sizeof(uint64_t) * 64 * 16));
    TRY(ensure(ctx, 15, sizeof(uint64_t) * std::max<size_t>(d.S, 1)));
    HIPCHK(hipMemsetAsync(sbuf<void>(ctx, 14), 0, sizeof(uint64_t) * 64 * 16, st));
    hipLaunchKernelGGL(k_rec_cnt, dim3((d.S + 3) / 4), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_records_digest, dim3((d.S + 3) / 4), dim3(256), 0, st, d, sbuf<unsigned long long>(ctx, 15),
                       sbuf<unsigned long long>(ctx, 14));
    TRY(after_launch(ctx));
    uint64_t b[64 * 16];
    TRY(down(ctx, b, sbuf<void>(ctx, 14), sizeof b));
    if (conn_sum) TRY(down(ctx, conn_sum, sbuf<void>(ctx, 15), sizeof(uint64_t) * d.S));
    unsigned long long gate_fails = 0;
    TRY(gate_poll_begin(ctx, &gate_fails));
    HIPCHK(hipStreamSynchronize(st));
    gate_poll_end(ctx, gate_fails);
    memset(total, 0, sizeof *total);
    for (int k = 0; k < 64; k++) {
        total->count += b[k * 16];
        total->sum += b[k * 16 + 1];
        total->xor_ ^= b[k * 16 + 2];
        total->sum_masked += b[k * 16 + 3];
    }
    return CHD_OK;
}

// chd_tick's host half: the arguments checked as the reference's callers would have (one update per entity and round, one interest
// update per connection, ids in range, stamps inside [0, now]), then every input array staged into device scratch on the ctx stream
// (asynchronous where the caller's memory is page-locked); din = the same tick with DEVICE pointers
static int host_tick_stage(chd_ctx *ctx, const chd_tick_in *in, chd_tick_in &din) {
    din = *in;
    const size_t nu = in->n_updates, nq = in->n_queries, nc = in->n_cell_updates, ns = in->n_spots_total;
    if (nu && (!in->upd_x || !in->upd_z)) return fail(ctx, CHD_E_INVAL, "tick: NULL update positions");
    if (nq && !in->queries) return fail(ctx, CHD_E_INVAL, "tick: NULL queries");
    if (nc && (!in->cell_upd_channel || !in->cell_upd_sender)) return fail(ctx, CHD_E_INVAL, "tick: NULL cell updates");
    if (ns && (!in->spot_x || !in->spot_z)) return fail(ctx, CHD_E_INVAL, "tick: NULL spot coordinates");
    if (in->upd_idx)
        for (size_t i = 0; i < nu; i++)
            if (in->upd_idx[i] >= ctx->w.d.N) return fail(ctx, CHD_E_INVAL, "tick: entity slot %u out of range", in->upd_idx[i]);
    if (in->query_sub)
        for (size_t i = 0; i < nq; i++)
            if (in->query_sub[i] >= ctx->w.d.S) return fail(ctx, CHD_E_INVAL, "tick: subscriber slot %u out of range", in->query_sub[i]);
    // One update per entity and one interest update per connection per tick: the kernels rewrite an entity's /
    // a connection's state in place, one thread / wave per input record, so a repeated slot would race (and the
    // reference applies them one after the other: the host coalesces, keeping the last).  O(n) bitmaps.
    if (in->n_update_rounds && (!in->upd_round_off || in->upd_round_off[in->n_update_rounds] != in->n_updates))
        return fail(ctx, CHD_E_INVAL, "tick: upd_round_off must run from 0 to n_updates");
    if (in->upd_idx && nu > 1) {
        // (inside one ROUND of updates: chd_tick_in.upd_round_off hands a channel's several updates of a tick over in rounds)
        std::vector<uint64_t> seen(((size_t)ctx->w.d.N + 63) / 64, 0);
        const uint32_t one[2] = {0u, (uint32_t)nu};
        const uint32_t nr = in->n_update_rounds ? in->n_update_rounds : 1u;
        const uint32_t *off = in->n_update_rounds ? in->upd_round_off : one;
        for (uint32_t r = 0; r < nr; r++) {
            if (r) std::fill(seen.begin(), seen.end(), 0ull);
            for (size_t i = off[r]; i < off[r + 1] && i < nu; i++) {
                const uint32_t v = in->upd_idx[i];
                if (seen[v >> 6] & (1ull << (v & 63)))
                    return fail(ctx, CHD_E_INVAL, "tick: entity slot %u is updated twice in one round (coalesce the updates of a tick on the host, or hand them over in rounds: upd_round_off)", v);
                seen[v >> 6] |= 1ull << (v & 63);
            }
        }
    }
    for (size_t i = 0; in->upd_arrival_ns && i < nu; i++)
        if (in->upd_arrival_ns[i] < 0 || in->upd_arrival_ns[i] > in->now_ns)
            return fail(ctx, CHD_E_INVAL, "tick: update %zu arrives at %lld, outside [0, now_ns]", i, (long long)in->upd_arrival_ns[i]);
    for (size_t i = 0; in->cell_upd_arrival_ns && i < nc; i++)
        if (in->cell_upd_arrival_ns[i] < 0 || in->cell_upd_arrival_ns[i] > in->now_ns)
            return fail(ctx, CHD_E_INVAL, "tick: cell update %zu arrives at %lld, outside [0, now_ns]", i, (long long)in->cell_upd_arrival_ns[i]);
    if (in->query_sub && nq > 1) {
        std::vector<uint64_t> seen(((size_t)ctx->w.d.S + 63) / 64, 0);
        for (size_t i = 0; i < nq; i++) {
            const uint32_t v = in->query_sub[i];
            if (seen[v >> 6] & (1ull << (v & 63))) return fail(ctx, CHD_E_INVAL, "tick: subscriber slot %u sends two interest updates (keep the last one)", v);
            seen[v >> 6] |= 1ull << (v & 63);
        }
    }
    for (size_t i = 0; i < nq; i++)
        if ((in->queries[i].shapes & CHD_SHAPE_SPOTS) &&
            ((uint64_t)in->queries[i].spot_off + in->queries[i].n_spots > ns || in->queries[i].n_spot_dists > in->queries[i].n_spots))
            return fail(ctx, CHD_E_INVAL, "tick: query %zu spot range out of bounds", i);
    for (size_t i = 0; i < nc; i++)
        if (in->cell_upd_channel[i] < ctx->g.id_start || in->cell_upd_channel[i] - ctx->g.id_start >= ctx->g.ncell)
            return fail(ctx, CHD_E_INVAL, "tick: cell update %zu is not a spatial channel", i);
    auto stage = [&](int slot, const void *src, size_t bytes, const void **dst) -> int {
        *dst = nullptr;
        if (!src || !bytes) return CHD_OK;
        TRY(ensure(ctx, slot, bytes));
        TRY(up(ctx, ctx->scratch[slot].p, src, bytes));
        *dst = ctx->scratch[slot].p;
        return CHD_OK;
    };
    TRY(stage(0, in->upd_idx, 4 * nu, (const void **)&din.upd_idx));
    TRY(stage(1, in->upd_x, 8 * nu, (const void **)&din.upd_x));
    TRY(stage(2, in->upd_z, 8 * nu, (const void **)&din.upd_z));
    TRY(stage(3, in->upd_sender, 4 * nu, (const void **)&din.upd_sender));
    TRY(stage(4, in->cell_upd_channel, 4 * nc, (const void **)&din.cell_upd_channel));
    TRY(stage(5, in->cell_upd_sender, 4 * nc, (const void **)&din.cell_upd_sender));
    TRY(stage(6, in->query_sub, 4 * nq, (const void **)&din.query_sub));
    TRY(stage(7, in->queries, sizeof(chd_aoi_query) * nq, (const void **)&din.queries));
    TRY(stage(8, in->spot_x, 8 * ns, (const void **)&din.spot_x));
    TRY(stage(9, in->spot_z, 8 * ns, (const void **)&din.spot_z));
    TRY(stage(10, in->spot_dist, 4 * ns, (const void **)&din.spot_dist));
    TRY(stage(11, in->upd_arrival_ns, 8 * nu, (const void **)&din.upd_arrival_ns));
    TRY(stage(12, in->cell_upd_arrival_ns, 8 * nc, (const void **)&din.cell_upd_arrival_ns));
    if (ns && !in->spot_dist) {
        TRY(ensure(ctx, 10, 4 * ns));
        HIPCHK(hipMemsetAsync(ctx->scratch[10].p, 0, 4 * ns, ctx->up_stream ? ctx->up_stream : ctx->stream));
        din.spot_dist = (const uint32_t *)ctx->scratch[10].p;
    }
    // The staging uploads above were enqueued on `stream`; a CHAINED pipelined tick starts its stages on the second stream
    // after the previous tick's record kernel only — older than those uploads (with pinned host buffers they really are
    // asynchronous).  Host-pointer ticks therefore always take the un-chained path: the stage stream waits for everything
    // enqueued on `stream` so far.  (chd_tick_device keeps the chained fast path: its inputs must be complete at call time.)
    ctx->chain_prev = false;
    ctx->gchain_prev = false;
    return CHD_OK;
}

int chd_tick(chd_ctx *ctx, const chd_tick_in *in, chd_tick_out *out) {
    NEED_WORLD();
    if (!in || !out) return fail(ctx, CHD_E_INVAL, "chd_tick: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    chd_tick_in din;
    TRY(host_tick_stage(ctx, in, din));
    TRY(tick_locked(ctx, &din));
    return fetch_locked(ctx, out);
}

// chd_tick + chd_tick_fetch_segments as ONE call with two synchronisations instead of five: [uploads, the tick, the segment sizing
// pass, the counts' download] sync [the small lists, the segment filling pass, every download] sync.  What a gateway that writes its
// sockets from segments calls per tick.  `out` takes no dense records (records / conn_rec_off / conn_rec_cnt must be NULL).
int chd_tick_segments(chd_ctx *ctx, const chd_tick_in *in, chd_tick_out *out, chd_segments_out *seg) {
    NEED_WORLD();
    if (!in || !out || !seg || !seg->conn_seg_off || !seg->conn_rec_off) return fail(ctx, CHD_E_INVAL, "chd_tick_segments: NULL argument");
    if (out->records || out->conn_rec_off || out->conn_rec_cnt) return fail(ctx, CHD_E_INVAL, "chd_tick_segments: the dense record outputs belong to chd_tick");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (d.capq > 32u * SEG_BITMAP_WORDS) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_segments: max_interest_cells %u > %u", d.capq, 32u * SEG_BITMAP_WORDS);
    if ((uint64_t)d.S * d.capq > 0xFFFFFFFFull) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_segments: more than 2^32 subscriptions");
    chd_tick_in din;
    TRY(host_tick_stage(ctx, in, din));
    TRY(tick_locked(ctx, &din));
    hipStream_t st = ctx->stream;
    if (!W.seg_cnt) {
        TRY(walloc(ctx, &W.seg_cnt, (size_t)d.S + 1));
        TRY(walloc(ctx, &W.seg_exp, (size_t)d.S + 1));
    }
    // ---- sizing, and everything whose size the host must know first ----
    hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, W.last_desc ? 1 : 0, 0, W.seg_cnt, (unsigned long long *)W.seg_exp,
                       (chd_fanout_segment *)nullptr, (chd_fanout_rec *)nullptr);
    launch_scan_u32_inplace(st, W.seg_cnt, d.S);
    launch_scan_u64_inplace(st, W.seg_exp, d.S);
    TRY(after_launch(ctx));
    uint32_t nseg = 0, pairs = 0;
    uint64_t nexp = 0, ub = 0, ringrow[8];
    TRY(down(ctx, &nseg, W.seg_cnt + d.S, sizeof nseg));
    TRY(down(ctx, &nexp, W.seg_exp + d.S, sizeof nexp));
    TRY(down(ctx, &ub, d.rec_ub + d.S, sizeof ub));
    TRY(down(ctx, ringrow, d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(st));
    fetch_counts_from_row(ctx, ringrow, out, &pairs);
    out->n_records = ringrow[0];
    const uint64_t ncol = !W.last_desc ? 0ull : d.wcol_on ? (uint64_t)(CHD_WCOLS + 1) * d.wcol_stride : (uint64_t)d.N + d.ghost_cap;
    seg->n_segments = nseg;
    seg->n_explicit = nexp;
    seg->n_columns = ncol;
    seg->n_records = ringrow[0];
    const bool seg_fits = !(nseg > seg->segments_cap || nexp > seg->records_cap || ncol > seg->columns_cap || (nseg && !seg->segments) || (nexp && !seg->records) ||
                            (ncol && !seg->columns));
    // ---- the lists and the segments ----
    int rc = CHD_OK;
    TRY(fetch_lists_enqueue(ctx, out, rc));
    if (seg_fits) {
        if (W.seg_stage_cap < nseg) {
            if (W.seg_stage) HIPCHK(hipFree(W.seg_stage));
            W.seg_stage = nullptr;
            W.seg_stage_cap = nseg + nseg / 4 + 1024;
            HIPCHK(hipMalloc((void **)&W.seg_stage, W.seg_stage_cap * sizeof(chd_fanout_segment)));
        }
        if (W.seg_rec_stage_cap < nexp) {
            if (W.seg_rec_stage) HIPCHK(hipFree(W.seg_rec_stage));
            W.seg_rec_stage = nullptr;
            W.seg_rec_stage_cap = nexp + nexp / 4 + 1024;
            HIPCHK(hipMalloc((void **)&W.seg_rec_stage, W.seg_rec_stage_cap * sizeof(chd_fanout_rec)));
        }
        hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, W.last_desc ? 1 : 0, 1, W.seg_cnt, (unsigned long long *)W.seg_exp,
                           W.seg_stage, W.seg_rec_stage);
        TRY(after_launch(ctx));
        TRY(down(ctx, seg->conn_seg_off, W.seg_cnt, sizeof(uint32_t) * ((size_t)d.S + 1)));
        TRY(down(ctx, seg->conn_rec_off, W.seg_exp, sizeof(uint64_t) * ((size_t)d.S + 1)));
        TRY(down(ctx, seg->segments, W.seg_stage, sizeof(chd_fanout_segment) * (size_t)nseg));
        TRY(down(ctx, seg->records, W.seg_rec_stage, sizeof(chd_fanout_rec) * nexp));
        if (ncol) TRY(down(ctx, seg->columns, d.ce_chan_view, sizeof(uint32_t) * ncol));
    }
    unsigned long long gate_fails = 0;
    TRY(gate_poll_begin(ctx, &gate_fails));
    HIPCHK(hipStreamSynchronize(st));
    gate_poll_end(ctx, gate_fails);
    chd_tick_stats &stt = ctx->stats;
    stt.n_records = ringrow[0];
    stt.n_record_upper_bound = ub;
    stt.n_handovers = out->n_handovers;
    stt.n_unsubs = out->n_unsubs;
    stt.n_pairs = pairs;
    if (ctx->prof_depth > 0) stage_times(ctx, ctx->ring.cur_tick, stt);
    if (out->overflow && rc == CHD_OK) rc = CHD_E_CAPACITY;
    if (rc == CHD_E_CAPACITY) return fail(ctx, rc, "tick output truncated (overflow mask 0x%x)", out->overflow);
    // (the tick itself is done: a caller whose segment buffers were too small grows them and calls chd_tick_fetch_segments)
    if (!seg_fits)
        return fail(ctx, CHD_E_CAPACITY, "chd_tick_segments: %u segments, %llu explicit records, %llu column entries needed", nseg,
                    (unsigned long long)nexp, (unsigned long long)ncol);
    return rc;
}

// ---------------------------------------------------------------------------
// region-sharded worlds
// ---------------------------------------------------------------------------

// ---- chd_tick_segments_begin / _end: the tick and its segment output as an asynchronous pair, two ticks in flight ----
// Stream plan (measured, profiles/r07f_*: a barrier packet that WAITS in a second hardware queue while kernels run in the first one
// delays those kernels by ~40 us every ~55 us, so the copies are not chained behind the tick with hipStreamWaitEvent):
//   _begin(t):  side stream: the input uploads into the scratch bank of t's parity (tick t-1's kernels are still running)
//               ctx stream:  [wait uploads] the tick | k_segments (sizes) | k_seg_scan (offsets + header, also into the page-locked
//                            block) | k_segments (fill) | k_seg_stage (columns, query status, handovers, lists) | event
//   _end(t):    host waits for that event (tick t+1's kernels are queued behind it and start at once), reads the sizes from the
//               page-locked header, enqueues ONE copy of exactly the block's bytes on the side stream, waits for it.
static int segp_setup(chd_ctx *ctx) {
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (W.segp_ready) return CHD_OK;
    const size_t S = d.S;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    W.segp_o_cnt = al(SEGH_WORDS * 8);
    W.segp_o_exp = W.segp_o_cnt + al(4 * (S + 1));
    W.segp_o_qst = W.segp_o_exp + al(8 * (S + 1));
    W.segp_o_var = W.segp_o_qst + al(4 * S);
    const size_t colcap = d.wcol_stride ? (size_t)(CHD_WCOLS + 1) * d.wcol_stride : (size_t)d.N + d.ghost_cap;
    // the variable part: the columns + 96 MiB (6M segments, or 12M explicit records)
    W.segp_cap = W.segp_o_var + al(4 * colcap) + (96ull << 20);
    if (hipStreamCreateWithFlags(&W.segp_stream, hipStreamNonBlocking) != hipSuccess) return fail(ctx, CHD_E_HIP, "chd_tick_segments_begin: no side stream");
    for (auto &sl : W.segp) {
        HIPCHK(hipMalloc((void **)&sl.d_blk, W.segp_cap));
        HIPCHK(hipHostMalloc((void **)&sl.h, W.segp_cap, hipHostMallocDefault));
        HIPCHK(hipHostGetDevicePointer((void **)&sl.h_dev, sl.h, 0));
        memset(sl.h, 0, SEGH_WORDS * 8);
        HIPCHK(hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&sl.ev_begin));  // (with timestamps: chd_set_profiling worlds report the tick's device time)
        HIPCHK(hipEventCreate(&sl.ev_fill));
    }
    W.segp_ready = true;
    return CHD_OK;
}

int chd_tick_segments_begin(chd_ctx *ctx, const chd_tick_in *in) {
    NEED_WORLD();
    if (!in) return fail(ctx, CHD_E_INVAL, "chd_tick_segments_begin: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (d.capq > 32u * SEG_BITMAP_WORDS) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_segments_begin: max_interest_cells %u > %u", d.capq, 32u * SEG_BITMAP_WORDS);
    if ((uint64_t)d.S * d.capq > 0xFFFFFFFFull) return fail(ctx, CHD_E_TOO_LARGE, "chd_tick_segments_begin: more than 2^32 subscriptions");
    if (W.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_tick_segments_begin on a region-sharded world");
    TRY(segp_setup(ctx));
    if (W.segp_pending >= 2) return fail(ctx, CHD_E_STATE, "chd_tick_segments_begin: two ticks are in flight already: chd_tick_segments_end first");
    World::SegPipeSlot &sl = W.segp[W.segp_head];
    hipStream_t st = ctx->stream;
    // the uploads: on the side stream, into this parity's bank of scratch buffers (the other bank may still be read by the tick in
    // flight; this one was last read by the tick two back, whose _end has returned)
    for (int i = 0; i < 13; i++) std::swap(ctx->scratch[i], W.segp_scratch[W.segp_head][i]);
    ctx->up_stream = W.segp_stream;
    chd_tick_in din;
    int rc = host_tick_stage(ctx, in, din);
    ctx->up_stream = nullptr;
    auto unbank = [&]() { for (int i = 0; i < 13; i++) std::swap(ctx->scratch[i], W.segp_scratch[W.segp_head][i]); };
    if (rc != CHD_OK) { unbank(); return rc; }
    if (hipEventRecord(sl.ev_up, W.segp_stream) != hipSuccess || hipStreamWaitEvent(st, sl.ev_up, 0) != hipSuccess) { unbank(); return fail(ctx, CHD_E_HIP, "chd_tick_segments_begin: event"); }
    sl.timed = ctx->prof_depth > 0;
    if (sl.timed) (void)hipEventRecord(sl.ev_begin, st);
    rc = tick_locked(ctx, &din);
    unbank();
    if (rc != CHD_OK) return rc;
    const int have_desc = W.last_desc ? 1 : 0;
    uint32_t *cnt = (uint32_t *)(sl.d_blk + W.segp_o_cnt);
    unsigned long long *exp = (unsigned long long *)(sl.d_blk + W.segp_o_exp);
    sl.ncol = !W.last_desc ? 0ull : d.wcol_on ? (uint64_t)(CHD_WCOLS + 1) * d.wcol_stride : (uint64_t)d.N + d.ghost_cap;
    sl.nq = in->n_queries;
    hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, have_desc, 0, cnt, exp, (chd_fanout_segment *)nullptr, (chd_fanout_rec *)nullptr,
                       (unsigned char *)nullptr);
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, st, d, ctx->ring.cur_tick % TICK_RING, sl.d_blk, sl.h_dev, (uint32_t)W.segp_o_cnt, (uint32_t)W.segp_o_exp,
                       (unsigned long long)W.segp_o_var, (unsigned long long)sl.ncol, sl.nq, (unsigned long long)W.segp_cap,
                       (const unsigned long long *)((W.gate && W.gated) ? W.gate + GATE_FAIL : nullptr), (unsigned long long)ctx->ring.cur_tick);
    hipLaunchKernelGGL(k_segments, dim3(d.S), dim3(64), 0, st, ctx->g, d, have_desc, 1, cnt, exp, (chd_fanout_segment *)nullptr, (chd_fanout_rec *)nullptr, sl.d_blk);
    hipLaunchKernelGGL(k_seg_stage, dim3(128), dim3(256), 0, st, d, sl.d_blk, (uint32_t)W.segp_o_qst);
    TRY(after_launch(ctx));
    HIPCHK(hipEventRecord(sl.ev_fill, st));
    sl.tick_no = ctx->ring.cur_tick;
    W.segp_pending++;
    W.segp_head ^= 1u;
    return CHD_OK;
}

int chd_tick_segments_end(chd_ctx *ctx, chd_segments_block *out) {
    NEED_WORLD();
    if (!out) return fail(ctx, CHD_E_INVAL, "chd_tick_segments_end: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    if (!W.segp_pending) return fail(ctx, CHD_E_STATE, "chd_tick_segments_end without a chd_tick_segments_begin");
    World::SegPipeSlot &sl = W.segp[W.segp_tail];
    memset(out, 0, sizeof *out);
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventSynchronize(sl.ev_fill));
    const auto t1 = std::chrono::steady_clock::now();
    W.segp_pending--;
    W.segp_tail ^= 1u;
    const unsigned long long *hdr = (const unsigned long long *)sl.h;  // (k_seg_scan wrote it there)
    if (hdr[SEGH_TICK] != sl.tick_no) return fail(ctx, CHD_E_HIP, "chd_tick_segments_end: the block's header is of tick %llu, not %llu", hdr[SEGH_TICK], (unsigned long long)sl.tick_no);
    const uint64_t nseg = hdr[SEGH_NSEG], nexp = hdr[SEGH_NEXP], nho = hdr[SEGH_NHO], nun = hdr[SEGH_NUN], nnew = hdr[SEGH_NNEW], total = hdr[SEGH_TOTAL];
    const unsigned long long *row = hdr + SEGH_ROW;
    if (!hdr[SEGH_FITS])
        return fail(ctx, CHD_E_CAPACITY, "chd_tick_segments_end: %llu segments, %llu explicit records, %llu handovers, %llu / %llu list entries: %llu bytes exceed the "
                    "block's %llu: take this tick with chd_tick_fetch + chd_tick_fetch_segments", (unsigned long long)nseg, (unsigned long long)nexp,
                    (unsigned long long)nho, (unsigned long long)nun, (unsigned long long)nnew, (unsigned long long)total, (unsigned long long)W.segp_cap);
    // ONE copy of exactly the block's bytes (behind the header the host has already): the next tick's kernels run meanwhile
    const size_t skip = SEGH_WORDS * 8;
    HIPCHK(hipMemcpyAsync(sl.h + skip, sl.d_blk + skip, total - skip, hipMemcpyDeviceToHost, W.segp_stream));
    HIPCHK(hipStreamSynchronize(W.segp_stream));
    out->wait_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    out->copy_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
    out->conn_seg_off = (const uint32_t *)(sl.h + W.segp_o_cnt);
    out->conn_rec_off = (const uint64_t *)(sl.h + W.segp_o_exp);
    out->columns = (const uint32_t *)(sl.h + hdr[SEGH_OFF_COL]); out->n_columns = hdr[SEGH_NCOL];
    out->segments = (const chd_fanout_segment *)(sl.h + hdr[SEGH_OFF_SEG]); out->n_segments = nseg;
    out->records = (const chd_fanout_rec *)(sl.h + hdr[SEGH_OFF_REC]); out->n_explicit = nexp;
    out->n_records = row[0];
    out->handovers = (const chd_handover_rec *)(sl.h + hdr[SEGH_OFF_HO]); out->n_handovers = (uint32_t)nho; out->n_locked_aborts = (uint32_t)row[3];
    out->unsub_sub = (const uint32_t *)(sl.h + hdr[SEGH_OFF_UN]); out->unsub_channel = (const uint32_t *)(sl.h + hdr[SEGH_OFF_UN + 1]); out->n_unsubs = (uint32_t)nun;
    out->newsub_sub = (const uint32_t *)(sl.h + hdr[SEGH_OFF_NEW]); out->newsub_channel = (const uint32_t *)(sl.h + hdr[SEGH_OFF_NEW + 1]);
    out->newsub_interval_ms = (const uint32_t *)(sl.h + hdr[SEGH_OFF_NEW + 2]); out->n_newsubs = (uint32_t)nnew;
    out->query_status = (const int32_t *)(sl.h + W.segp_o_qst); out->n_queries = (uint32_t)hdr[SEGH_NQ];
    out->overflow = (uint32_t)(row[7] & 0xFFFFFFFFull); out->history_overflow = (uint32_t)(row[7] >> 32);
    out->block = sl.h; out->block_bytes = total;
    if (sl.timed) (void)hipEventElapsedTime(&out->device_ms, sl.ev_begin, sl.ev_fill);
    gate_poll_end(ctx, hdr[SEGH_GATE_FAIL]);
    chd_tick_stats &stt = ctx->stats;
    stt.n_records = row[0];
    stt.n_record_upper_bound = hdr[SEGH_REC_UB];
    stt.n_handovers = (uint32_t)nho;
    stt.n_unsubs = (uint32_t)nun;
    stt.n_pairs = (uint32_t)row[6];
    if (out->overflow) return fail(ctx, CHD_E_CAPACITY, "tick output truncated (overflow mask 0x%x)", out->overflow);
    return CHD_OK;
}

int chd_set_stream(chd_ctx *ctx, void *hip_stream, int external) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = external ? (hipStream_t)hip_stream : ctx->own_stream;
    if (ctx->w.created) TRY(gate_probe(ctx));  // (CHD_WORLD_GATED_OVERLAP: does THIS pair of streams run side by side?)
    return CHD_OK;
}

int chd_shard_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id, const double *x, const double *z,
                    const uint32_t *flags, const uint32_t *sender) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!chan_id || !x || !z) return fail(ctx, CHD_E_INVAL, "chd_shard_spawn: NULL buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (ctx->w.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_spawn on a world with caller-chosen slots (chd_world_spawn)");
    if (ctx->w.d.deep_depth && !ctx->w.d.log_on)
        return fail(ctx, CHD_E_STATE, "history_depth on a region-sharded world needs chd_world_cfg.shard_channels (the update log by channel id)");
    if (ctx->w.d.ce_by_chan) {  // (tables by channel id: the log, the wire payloads — an id outside them would index past their ends)
        const uint32_t eid0 = ctx->w.d.log_eid0, nch = ctx->w.cfg.shard_channels;
        for (uint32_t i = 0; i < n; i++)
            if (chan_id[i] < eid0 || chan_id[i] - eid0 >= nch)
                return fail(ctx, CHD_E_INVAL, "chd_shard_spawn: channel id %#x outside entity_channel_id_start .. + shard_channels (%u)", chan_id[i], nch);
    }
    ctx->w.slot_mode = 2;
    TRY(bind(ctx));
    TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(ensure(ctx, 2, 8 * (size_t)n)); TRY(ensure(ctx, 3, 8 * (size_t)n));
    TRY(ensure(ctx, 4, 4 * (size_t)n)); TRY(ensure(ctx, 5, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), chan_id, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 2), x, 8 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 3), z, 8 * (size_t)n));
    if (flags) TRY(up(ctx, sbuf<void>(ctx, 4), flags, 4 * (size_t)n));
    if (sender) TRY(up(ctx, sbuf<void>(ctx, 5), sender, 4 * (size_t)n));
    launch_spawn_auto(ctx->stream, ctx->g, ctx->w.d, n, sbuf<uint32_t>(ctx, 1), sbuf<double>(ctx, 2), sbuf<double>(ctx, 3),
                      flags ? sbuf<uint32_t>(ctx, 4) : nullptr, sender ? sbuf<uint32_t>(ctx, 5) : nullptr,
                      ctx->ring.cur_tick);
    TRY(after_launch(ctx));
    uint32_t ovf = 0;
    TRY(down(ctx, &ovf, ctx->w.d.counters + CTR_OVERFLOW, 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ovf & OVF_SLOTS) return fail(ctx, CHD_E_CAPACITY, "chd_shard_spawn: more entities than max_entities slots");
    return CHD_OK;
}

int chd_shard_despawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!chan_id) return fail(ctx, CHD_E_INVAL, "chd_shard_despawn: NULL ids");
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (ctx->w.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_despawn on a world with caller-chosen slots (chd_world_despawn)");
    TRY(bind(ctx));
    std::vector<uint32_t> gone(chan_id, chan_id + n);
    std::sort(gone.begin(), gone.end());
    TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), gone.data(), 4 * (size_t)n));
    launch_shard_despawn(ctx->stream, ctx->w.d, sbuf<uint32_t>(ctx, 1), n);
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // (the staging vector goes out of scope)
    return CHD_OK;
}

// records behind each emigrant segment's (cap + 1): the sender's maxFanOutIntervalMs per cell, 8 cells per 32-byte record (log_on)
static uint32_t migrate_extra(const chd_ctx *ctx) { return ctx->w.d.log_on ? (ctx->g.ncell + 7u) / 8u : 0u; }

// ghost room behind the own entries: reallocates the cell-sorted tables (rebuilt every tick, nothing to preserve)
static int shard_reserve_ghosts(chd_ctx *ctx, uint32_t ghosts) {
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (ghosts <= d.ghost_cap && d.cell_cov) return CHD_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t)d.N + ghosts;
    TRY(walloc(ctx, &d.ce, n + 1));
    TRY(walloc(ctx, &d.ce_sprev, n));
    TRY(walloc(ctx, &d.ce8, n + 2));
    TRY(walloc(ctx, &d.ce_chan, n + 520));
    if (d.ce_slot) TRY(walloc(ctx, &d.ce_slot, n + 2));
    W.x.npos = (uint32_t)n + 2u;  // (wire worlds: a record's position word may name a ghost entry)
    if (d.off_on) {  // (the offset columns run beside the entries, ghosts included)
        d.off_stride = (uint32_t)((n + 520 + 63) & ~(size_t)63);
        TRY(walloc(ctx, &d.ce_off, (size_t)CHD_OFF_SLOTS * d.off_stride + 520));
    }
    d.wcol_stride = 0;  // (no window columns on region-sharded worlds: the tables are rebuilt with ghost room, one column array)
    if (!d.cell_cov) TRY(walloc(ctx, &d.cell_cov, ctx->g.ncell));
    d.ghost_cap = ghosts;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

static int shard_halo_layout_locked(chd_ctx *ctx, uint32_t rank, uint32_t world, chd_halo_seg *segs, uint64_t *send_total, uint64_t *recv_total) {
    const DevGrid &g = ctx->g;
    TRY(bind(ctx));
    World &W = ctx->w;
    const uint32_t halo = g.border, region = g.sgc * g.sgr, N = W.d.N;
    uint64_t so = 0, ro = 0;
    uint32_t ghosts = 0;
    std::vector<uint64_t> soff(world), roff(world);
    std::vector<uint32_t> goff(world);
    for (uint32_t p = 0; p < world; p++) {
        const HaloRect out = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, halo, rank, p);
        const HaloRect in = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, halo, p, rank);
        const uint32_t ocap = halo_cap_entries(N, out.w * out.h, region), icap = halo_cap_entries(N, in.w * in.h, region);
        segs[p].send_off = soff[p] = so;
        segs[p].send_bytes = halo_seg_bytes(ocap, out.w * out.h);
        segs[p].recv_off = roff[p] = ro;
        segs[p].recv_bytes = halo_seg_bytes(icap, in.w * in.h);
        so += segs[p].send_bytes;
        ro += segs[p].recv_bytes;
        goff[p] = ghosts;
        ghosts += icap;
    }
    *send_total = so;
    *recv_total = ro;
    if (rank == W.halo_rank && world == W.halo_world && W.d_ghost_off) return CHD_OK;  // (asked again)
    // first call for this (rank, world) — or a query about ANOTHER rank's layout (host-staged exchanges need the
    // senders' offsets): only this rank's own layout is installed, by the call that names its rank first
    if (W.halo_world == 0) {
        TRY(shard_reserve_ghosts(ctx, ghosts));
        TRY(walloc(ctx, &W.d_halo_send_off, world));
        TRY(walloc(ctx, &W.d_halo_recv_off, world));
        TRY(walloc(ctx, &W.d_ghost_off, world));
        if (world > 1) {
            TRY(walloc(ctx, &W.d.limbo, 2 * (size_t)N, false));
            HIPCHK(hipHostMalloc((void **)&W.h_mig_gmax, 4 * sizeof(uint32_t), hipHostMallocDefault));
            for (auto &e : W.ev_mig) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        HIPCHK(hipMemcpy(W.d_halo_send_off, soff.data(), 8 * (size_t)world, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(W.d_halo_recv_off, roff.data(), 8 * (size_t)world, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(W.d_ghost_off, goff.data(), 4 * (size_t)world, hipMemcpyHostToDevice));
        W.halo_rank = rank;
        W.halo_world = world;
    }
    return CHD_OK;
}

int chd_shard_halo_layout(chd_ctx *ctx, uint32_t rank, uint32_t world, chd_halo_seg *segs, uint64_t *send_total, uint64_t *recv_total) {
    NEED_WORLD();
    if (!segs || !send_total || !recv_total) return fail(ctx, CHD_E_INVAL, "chd_shard_halo_layout: NULL output");
    const DevGrid &g = ctx->g;
    if (!world || rank >= world) return fail(ctx, CHD_E_INVAL, "chd_shard_halo_layout: rank %u of %u", rank, world);
    if (world != g.server_cols * g.server_rows)
        return fail(ctx, CHD_E_INVAL, "chd_shard_halo_layout: %u ranks but the grid has %u server regions", world, g.server_cols * g.server_rows);
    std::lock_guard<FairMutex> lk(ctx->mu);
    return shard_halo_layout_locked(ctx, rank, world, segs, send_total, recv_total);
}

// Segment capacity of this tick's emigrant exchange.  Every rank must use the same value (the all-to-all's sizes), so it is a
// pure function of a quantity every rank holds identically: the global maximum segment count of tick t - 2 (carried in the
// segment headers, k_export_finish / k_import).  Four times that maximum plus a floor, as a power of two; the caller's `cap`
// until two exchanges have been seen.  A burst beyond it sets overflow bit 32 for the tick (the surplus stays and retries).
static uint32_t migrate_cap_for(World &W, uint32_t cur_tick, uint32_t cap_max) {
    if (cur_tick < 3) return cap_max;
    const uint32_t slot = (cur_tick - 2u) & 3u;
    if (!W.h_mig_gmax || W.mig_tick[slot] != cur_tick - 2u) return cap_max;
    (void)hipEventSynchronize(W.ev_mig[slot]);  // (recorded two ticks ago)
    const uint64_t want = 4ull * W.h_mig_gmax[slot] + 64ull;
    uint64_t c = 256;
    while (c < want) c <<= 1;
    return (uint32_t)std::min<uint64_t>(c, cap_max);
}

// The ingest in two halves.  pre: positions -> cells, handovers, list members on THIS rank moved, handovers whose src map is another
// rank's written to d_req_send (handover lists only).  post: the requests received applied, then the emigrants exported.
static int shard_ingest_pre_locked(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan, const uint8_t *d_has_update,
                                   uint32_t n_chan, uint32_t rank, uint32_t world, chd_handover_request *d_req_send, uint32_t req_cap) {
    if (ctx->w.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_ingest on a world with caller-chosen slots");
    ctx->w.slot_mode = 2;
    TRY(bind(ctx));
    World &W = ctx->w;
    if (W.d.log_on && n_chan && !W.d.sh_sender_by_chan)
        return fail(ctx, CHD_E_STATE, "a world with an update log by channel id takes its updates' senders by channel id too (chd_shard_set_update_senders)");
    if (W.d.log_on && n_chan > W.d.log_n) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest: %u channels, the world was created for %u (shard_channels)", n_chan, W.d.log_n);
    TRY(tick_begin(ctx, now_ns));
    if (W.plan_recipients) {
        // Notify runs before the tick's interest updates (spatial.go:776-857 reads the subscriptions as they are then): keep them
        WorldDev &d = W.d;
        if (d.wb) {
            if (!W.snap_bits) TRY(walloc(ctx, &W.snap_bits, (size_t)d.S * d.wb, false));
            HIPCHK(hipMemcpyAsync(W.snap_bits, d.sub_bits, sizeof(uint64_t) * (size_t)d.S * d.wb, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            if (!W.snap_cell) { TRY(walloc(ctx, &W.snap_cell, (size_t)d.S * d.capq, false)); TRY(walloc(ctx, &W.snap_cnt, d.S, false)); }
            HIPCHK(hipMemcpyAsync(W.snap_cell, d.pair_cell, sizeof(uint32_t) * (size_t)d.S * d.capq, hipMemcpyDeviceToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(W.snap_cnt, d.pair_cnt, sizeof(uint32_t) * d.S, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    W.d.prev_ns = ctx->ring.n > 1 ? ctx->ring.t[1] : -1;  // (sub-tick arrival offsets: a regular update arrived after the previous tick)
    const uint32_t eid0 = ctx->cfg.entity_channel_id_start ? ctx->cfg.entity_channel_id_start : 0x80000u;
    launch_ingest_by_channel(ctx->stream, ctx->g, ctx->w.d, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, eid0,
                             ctx->ring.cur_tick, rank, world, (uint4 *)d_req_send, req_cap);
    // (log_on: the tick's updates are logged after the emigrant exchange — shard_import_locked — which also carries the cells' maxFanOutIntervalMs)
    W.log_x = d_x_by_chan; W.log_z = d_z_by_chan; W.log_has = d_has_update; W.log_nchan = n_chan;
    W.ingest_pending = true;  // (chd_shard_import consumes it: one import per ingest — a second one would log the tick's updates twice)
    return CHD_OK;
}

static int shard_ingest_post_locked(chd_ctx *ctx, const chd_handover_request *d_req_recv, uint32_t req_cap, uint32_t rank, uint32_t world,
                                    chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used) {
    World &W = ctx->w;
    const uint32_t use = (cap_used && world > 1) ? migrate_cap_for(W, ctx->ring.cur_tick, cap) : cap;
    if (cap_used) *cap_used = use;
    W.mig_cap = use;
    if (world > 1 && d_req_recv) launch_apply_requests(ctx->stream, W.d, (const uint4 *)d_req_recv, world, req_cap);
    if (world > 1) launch_export(ctx->stream, ctx->g, ctx->w.d, rank, world, d_send, use, ctx->ring.cur_tick, migrate_extra(ctx));
    TRY(after_launch(ctx));
    return CHD_OK;
}

static int shard_ingest_locked(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan, const uint8_t *d_has_update, uint32_t n_chan,
                               uint32_t rank, uint32_t world, chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used) {
    if (world > 1 && ctx->w.d.sh_list_of)
        return fail(ctx, CHD_E_STATE, "chd_shard_ingest on a world with handover lists: a handover may concern another rank's entity map — "
                                      "chd_shard_ingest_pre, exchange the requests, chd_shard_ingest_post (or chd_shard_tick)");
    TRY(shard_ingest_pre_locked(ctx, now_ns, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, rank, world, nullptr, 0));
    return shard_ingest_post_locked(ctx, nullptr, 0, rank, world, d_send, cap, cap_used);
}

static int shard_args(chd_ctx *ctx, const char *fn, uint32_t rank, uint32_t world) {
    if (!world || rank >= world) return fail(ctx, CHD_E_INVAL, "%s: rank %u of %u", fn, rank, world);
    if (world != ctx->g.server_cols * ctx->g.server_rows)
        return fail(ctx, CHD_E_INVAL, "%s: %u ranks but the grid has %u server regions", fn, world, ctx->g.server_cols * ctx->g.server_rows);
    return CHD_OK;
}

int chd_shard_ingest_pre(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan, const uint8_t *d_has_update,
                         uint32_t n_chan, uint32_t rank, uint32_t world, chd_handover_request *d_req_send, uint32_t req_cap) {
    NEED_WORLD();
    if (n_chan && (!d_x_by_chan || !d_z_by_chan)) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest_pre: NULL positions");
    TRY(shard_args(ctx, "chd_shard_ingest_pre", rank, world));
    if (world > 1 && (!d_req_send || !req_cap)) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest_pre: NULL request buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(shard_ingest_pre_locked(ctx, now_ns, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, rank, world, world > 1 ? d_req_send : nullptr, req_cap));
    return after_launch(ctx);
}

int chd_shard_ingest_post(chd_ctx *ctx, const chd_handover_request *d_req_recv, uint32_t req_cap, uint32_t rank, uint32_t world,
                          chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used) {
    NEED_WORLD();
    TRY(shard_args(ctx, "chd_shard_ingest_post", rank, world));
    if (world > 1 && (!d_send || !cap || !d_req_recv || !req_cap)) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest_post: NULL buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (ctx->w.slot_mode != 2) return fail(ctx, CHD_E_STATE, "chd_shard_ingest_post before chd_shard_ingest_pre");
    return shard_ingest_post_locked(ctx, d_req_recv, req_cap, rank, world, d_send, cap, cap_used);
}

int chd_shard_ingest(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan,
                     const uint8_t *d_has_update, uint32_t n_chan, uint32_t rank, uint32_t world,
                     chd_entity_state *d_send, uint32_t cap, uint32_t *cap_used) {
    NEED_WORLD();
    if (n_chan && (!d_x_by_chan || !d_z_by_chan)) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest: NULL positions");
    if (!world || rank >= world) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest: rank %u of %u", rank, world);
    if (world != ctx->g.server_cols * ctx->g.server_rows)
        return fail(ctx, CHD_E_INVAL, "chd_shard_ingest: %u ranks but the grid has %u server regions", world,
                    ctx->g.server_cols * ctx->g.server_rows);
    if (world > 1 && (!d_send || !cap)) return fail(ctx, CHD_E_INVAL, "chd_shard_ingest: NULL send buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    return shard_ingest_locked(ctx, now_ns, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, rank, world, d_send, cap, cap_used);
}

static int shard_import_locked(chd_ctx *ctx, const chd_entity_state *d_recv, uint32_t world, uint32_t cap, void *d_halo_send) {
    if (ctx->w.slot_mode != 2 || !ctx->w.ingest_pending) return fail(ctx, CHD_E_STATE, "chd_shard_import before chd_shard_ingest (one import per ingest)");
    World &W = ctx->w;
    if (world > 1 && (W.halo_world != world || !d_halo_send)) return fail(ctx, CHD_E_STATE, "chd_shard_import: call chd_shard_halo_layout(rank, world) first and pass the halo send buffer");
    TRY(bind(ctx));
    WorldDev &d = W.d;
    hipStream_t st = ctx->stream;
    // (the fan-out of the previous tick read the combined views; the index build works on the own tables)
    d.ce_view = d.ce; d.ce8_view = d.ce8; d.ce_chan_view = d.ce_chan; d.ce_sprev_view = d.ce_sprev; d.ce_sprev_stride = 0;
    d.cell_start = d.cell_off;
    d.cell_end = d.cell_off + 1;
    if (world > 1) {
        if (cap != W.mig_cap) return fail(ctx, CHD_E_INVAL, "chd_shard_import: cap %u, but this tick's chd_shard_ingest used %u", cap, W.mig_cap);
        if (!d.limbo) return fail(ctx, CHD_E_STATE, "chd_shard_import: chd_shard_halo_layout has not installed this rank's layout");
        launch_import(st, d, d_recv, world, cap, ctx->ring.cur_tick, migrate_extra(ctx), ctx->g.ncell);
        const uint32_t slot = ctx->ring.cur_tick & 3u;
        HIPCHK(hipMemcpyAsync(W.h_mig_gmax + slot, d.mig_gmax + slot, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(W.ev_mig[slot], st));
        W.mig_tick[slot] = ctx->ring.cur_tick;
    }
    // log_on: ChannelData.OnUpdate for every channel of the world, under the cells' world-wide maxFanOutIntervalMs (just folded)
    launch_log_push(st, ctx->g, d, W.log_x, W.log_z, W.log_has, W.log_nchan, ctx->ring.cur_tick, W.last_now);
    W.log_x = W.log_z = nullptr; W.log_has = nullptr; W.log_nchan = 0;  // (the caller's arrays are its own again)
    W.ingest_pending = false;
    launch_index_build(st, ctx->g, d, ctx->ring.cur_tick, nullptr, 0, W.last_now);
    if (world > 1) launch_halo_pack(st, ctx->g, d, W.halo_rank, world, ctx->g.border, (unsigned char *)d_halo_send, W.d_halo_send_off);
    TRY(after_launch(ctx));
    return CHD_OK;
}

int chd_shard_import(chd_ctx *ctx, const chd_entity_state *d_recv, uint32_t world, uint32_t cap, void *d_halo_send) {
    NEED_WORLD();
    if (world > 1 && !d_recv) return fail(ctx, CHD_E_INVAL, "chd_shard_import: NULL receive buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    return shard_import_locked(ctx, d_recv, world, cap, d_halo_send);
}

static int shard_interest_locked(chd_ctx *ctx, const chd_tick_in *d_in) {
    if (ctx->w.slot_mode != 2) return fail(ctx, CHD_E_STATE, "chd_shard_interest before chd_shard_ingest");
    TRY(bind(ctx));
    TRY(check_queries(ctx, d_in));
    launch_aoi_interest(ctx->stream, ctx->g, ctx->lim, ctx->w.d, d_in->queries, d_in->n_queries, d_in->query_sub,
                        d_in->spot_x, d_in->spot_z, d_in->spot_dist, ctx->w.last_now, ctx->ring.cur_tick);
    TRY(after_launch(ctx));
    ctx->w.last_nq = d_in->n_queries;
    return CHD_OK;
}

int chd_shard_interest(chd_ctx *ctx, const chd_tick_in *d_in) {
    NEED_WORLD();
    if (!d_in) return fail(ctx, CHD_E_INVAL, "chd_shard_interest: NULL input");
    std::lock_guard<FairMutex> lk(ctx->mu);
    return shard_interest_locked(ctx, d_in);
}

static int shard_fanout_locked(chd_ctx *ctx, const void *d_halo_recv, uint32_t world, const chd_tick_in *d_in) {
    if (ctx->w.slot_mode != 2) return fail(ctx, CHD_E_STATE, "chd_shard_fanout before chd_shard_ingest");
    TRY(bind(ctx));
    TRY(check_queries(ctx, d_in));
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (world > 1 && W.halo_world != world) return fail(ctx, CHD_E_STATE, "chd_shard_fanout: call chd_shard_halo_layout(rank, world) first");
    if (!d.cell_cov) TRY(shard_reserve_ghosts(ctx, 0));
    hipStream_t st = ctx->stream;
    const TickRing &r = ctx->ring;
    const int64_t now = W.last_now;
    // stage events: ingest/index ran in the earlier phases (their slots read 0 here)
    const bool prof_skip = ctx->prof_depth > 0 && ctx->prof_kernel_only && ctx->prof_every > 1 && r.cur_tick % ctx->prof_every != 0;
    if (prof_skip) ctx->ev_overlap[r.cur_tick % (uint32_t)ctx->prof_depth] = 8;  // (CHD_PROF_RECORD_KERNEL_EVERY: nothing recorded)
    const bool prof = ctx->prof_depth > 0 && !prof_skip;
    hipEvent_t *ev = prof ? &ctx->ev[(size_t)(r.cur_tick % (uint32_t)ctx->prof_depth) * EV_PER_TICK] : nullptr;
    const bool prof_ends = prof && !ctx->prof_kernel_only;  // (chd_set_profiling_scope)
    if (prof) {
        ctx->ev_overlap[r.cur_tick % (uint32_t)ctx->prof_depth] = prof_ends ? 0 : 4;
        if (prof_ends) for (int k = 0; k <= 2; k++) HIPCHK(hipEventRecord(ev[k], st));
    }
    // the neighbours' border bands join the own tables as ghost entries: ONE local table over region + halo, so the
    // fan-out takes the same kernels (and fast paths) as on a single GPU
    // the spatial channels' OWN updates (a cell's entity map changed: spawn, destroy, handover — ChannelData.OnUpdate on the spatial
    // channel, data.go:149-173): per-cell state, so every rank applies the same list (the cells a rank fans out are its region's and
    // its neighbours' border cells alike); given in d_in like chd_tick_device's
    if (d_in->n_cell_updates) {
        if (!d_in->cell_upd_channel || !d_in->cell_upd_sender) return fail(ctx, CHD_E_INVAL, "chd_shard_fanout: NULL cell updates");
        if (!d.deep_depth && d_in->cell_upd_arrival_ns) return fail(ctx, CHD_E_STATE, "chd_shard_fanout: arrival stamps need a world with history_depth > 0");
        launch_cell_updates(st, ctx->g, d, d_in->n_cell_updates, d_in->cell_upd_channel, d_in->cell_upd_sender, r.cur_tick, d_in->cell_upd_arrival_ns, now);
    }
    d.seg_off = 0;
    W.last_desc = fanout_seg_path(d);
    launch_halo_unpack(st, ctx->g, d, world > 1 ? W.halo_rank : 0u, world, ctx->g.border, (const unsigned char *)d_halo_recv,
                       W.d_halo_recv_off, W.d_ghost_off, r.cur_tick, W.join_in_unpack ? W.gate + GATE_TOP : nullptr, W.gate_top);
    W.join_in_unpack = false;
    d.ce_view = d.ce; d.ce8_view = d.ce8; d.ce_chan_view = d.ce_chan; d.ce_sprev_view = d.ce_sprev; d.ce_sprev_stride = 0;
    d.cell_start = d.cell_tab;
    d.cell_end = d.cell_tab + ctx->g.ncell;
    launch_cell_offsets(st, ctx->g, d);  // (off_on: per cell and ring slot the range of the sub-tick offsets, ghost cells included)
    launch_aoi_interest(st, ctx->g, ctx->lim, d, d_in->queries, d_in->n_queries, d_in->query_sub, d_in->spot_x,
                        d_in->spot_z, d_in->spot_dist, now, r.cur_tick);
    if (prof_ends) HIPCHK(hipEventRecord(ev[3], st));
    launch_fanout_plan(st, ctx->g, d, now, r);
    if (prof_ends) HIPCHK(hipEventRecord(ev[4], st));
    if (prof) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 4], st));
    launch_fanout_emit_main(st, ctx->g, d, now, r);
    if (prof && !d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));
    launch_fanout_emit_filt(st, ctx->g, d);
    if (prof && d.off_on) HIPCHK(hipEventRecord(ev[CHD_N_STAGES + 3], st));  // (emit_main_us: both record-writing kernels)
    if (W.gated) launch_fanout_tail(st, ctx->g, d, now, r, r.cur_tick % TICK_RING, W.gate + GATE_EPI, ++W.gate_epi);
    else launch_fanout_tail(st, ctx->g, d, now, r, r.cur_tick % TICK_RING);
    if (prof_ends) HIPCHK(hipEventRecord(ev[5], st));
    TRY(after_launch(ctx));
    if (d_in->n_queries) W.last_nq = d_in->n_queries;
    W.ticked = true;
    W.wire_built = false;
    return CHD_OK;
}

int chd_shard_fanout(chd_ctx *ctx, const void *d_halo_recv, uint32_t world, const chd_tick_in *d_in) {
    NEED_WORLD();
    if (!d_in) return fail(ctx, CHD_E_INVAL, "chd_shard_fanout: NULL input");
    if (!world) return fail(ctx, CHD_E_INVAL, "chd_shard_fanout: world = 0");
    if (world > 1 && !d_halo_recv) return fail(ctx, CHD_E_INVAL, "chd_shard_fanout: NULL halo receive buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    return shard_fanout_locked(ctx, d_halo_recv, world, d_in);
}

int chd_shard_set_update_senders(chd_ctx *ctx, const uint32_t *d_sender_by_chan, uint32_t n_chan) {
    NEED_WORLD();
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    if (ctx->w.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_set_update_senders on a world with caller-chosen slots (chd_tick_in.upd_sender)");
    ctx->w.d.sh_sender_by_chan = n_chan ? d_sender_by_chan : nullptr;
    ctx->w.d.sh_sender_n = d_sender_by_chan ? n_chan : 0u;
    return CHD_OK;
}

int chd_shard_set_update_arrivals(chd_ctx *ctx, const int64_t *d_arrival_ns_by_chan, uint32_t n_chan) {
    NEED_WORLD();
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    if (!ctx->w.d.log_on) return fail(ctx, CHD_E_STATE, "chd_shard_set_update_arrivals: the world keeps no update log by channel id (chd_world_cfg.shard_channels)");
    ctx->w.d.sh_arrival_by_chan = n_chan ? d_arrival_ns_by_chan : nullptr;
    ctx->w.d.sh_arrival_n = d_arrival_ns_by_chan ? n_chan : 0u;
    return CHD_OK;
}

int chd_shard_log_spawn(chd_ctx *ctx, uint32_t n, const uint32_t *chan_id, const double *x, const double *z) {
    NEED_WORLD();
    if (!n) return CHD_OK;
    if (!chan_id || !x || !z) return fail(ctx, CHD_E_INVAL, "chd_shard_log_spawn: NULL buffer");
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!ctx->w.d.log_on) return fail(ctx, CHD_E_STATE, "chd_shard_log_spawn: the world keeps no update log by channel id (chd_world_cfg.shard_channels)");
    // (the ids are checked HERE, on the host, as chd_shard_spawn does: the tick's sticky overflow mask is no place to look — an earlier
    // slot overflow would fail a valid call, and a bit set here would show up in the next tick as a bogus "no free slot")
    for (uint32_t k = 0; k < n; k++)
        if (chan_id[k] < ctx->w.d.log_eid0 || chan_id[k] - ctx->w.d.log_eid0 >= ctx->w.d.log_n)
            return fail(ctx, CHD_E_CAPACITY, "chd_shard_log_spawn: channel id %u outside entity_channel_id_start .. + shard_channels (%u .. %u)", chan_id[k],
                        ctx->w.d.log_eid0, ctx->w.d.log_eid0 + ctx->w.d.log_n - 1u);
    TRY(bind(ctx));
    TRY(ensure(ctx, 1, 4 * (size_t)n));
    TRY(ensure(ctx, 2, 8 * (size_t)n)); TRY(ensure(ctx, 3, 8 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), chan_id, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 2), x, 8 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 3), z, 8 * (size_t)n));
    launch_log_spawn(ctx->stream, ctx->g, ctx->w.d, n, sbuf<uint32_t>(ctx, 1), sbuf<double>(ctx, 2), sbuf<double>(ctx, 3));
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // (the staging buffers are the context's)
    return CHD_OK;
}

int chd_shard_migrate_extra_records(chd_ctx *ctx, uint32_t *extra) {
    NEED_WORLD();
    if (!extra) return fail(ctx, CHD_E_INVAL, "chd_shard_migrate_extra_records: NULL output");
    *extra = migrate_extra(ctx);
    return CHD_OK;
}

int chd_shard_set_handover_lists(chd_ctx *ctx, uint32_t n_lists, const uint32_t *list_off, const uint32_t *list_member_chan, uint32_t n,
                                 const uint32_t *chan_id, const uint32_t *list_of, uint32_t n_chan) {
    NEED_WORLD();
    if ((n_lists && (!list_off || (list_off[n_lists] && !list_member_chan))) || (n && (!chan_id || !list_of)))
        return fail(ctx, CHD_E_INVAL, "chd_shard_set_handover_lists: NULL buffer");
    if (!n_chan) return fail(ctx, CHD_E_INVAL, "chd_shard_set_handover_lists: n_chan = 0");
    const uint32_t eid0 = ctx->cfg.entity_channel_id_start ? ctx->cfg.entity_channel_id_start : 0x80000u;
    for (uint32_t k = 0; k < n_lists; k++)
        if (list_off[k + 1] < list_off[k]) return fail(ctx, CHD_E_INVAL, "chd_shard_set_handover_lists: list_off decreases at %u", k);
    for (uint32_t i = 0; i < n; i++) {
        if (chan_id[i] - eid0 >= n_chan) return fail(ctx, CHD_E_INVAL, "chd_shard_set_handover_lists: channel %u outside [%u, %u)", chan_id[i], eid0, eid0 + n_chan);
        if (list_of[i] != CHD_NO_HANDOVER_LIST && list_of[i] >= n_lists) return fail(ctx, CHD_E_INVAL, "chd_shard_set_handover_lists: list %u of %u", list_of[i], n_lists);
    }
    std::lock_guard<FairMutex> lk(ctx->mu);
    World &W = ctx->w;
    WorldDev &d = W.d;
    if (W.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_set_handover_lists on a world with caller-chosen slots (use chd_world_set_handover_lists)");
    W.slot_mode = 2;
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (d.sh_nchan != n_chan) {  // (first call, or another id range: the tables are rebuilt)
        d.sh_nchan = n_chan;
        d.sh_eid0 = eid0;
        TRY(walloc(ctx, &d.sh_slot_of, n_chan, false));
        TRY(walloc(ctx, &d.sh_list_of, n_chan, false));
        launch_slot_of_rebuild(ctx->stream, d);
    }
    std::vector<uint32_t> of(n_chan, CHD_NO_HANDOVER_LIST);
    for (uint32_t i = 0; i < n; i++) of[chan_id[i] - eid0] = list_of[i];
    const uint32_t nm = n_lists ? list_off[n_lists] : 0u;
    uint32_t *doff = nullptr, *dmem = nullptr;
    HIPCHK(hipStreamSynchronize(ctx->aux_stream));
    for (int k = 1; k <= 2; k++) { if (W.grp_buf[k]) HIPCHK(hipFree(W.grp_buf[k])); W.grp_buf[k] = nullptr; }  // (the previous call's lists)
    HIPCHK(hipMalloc(&W.grp_buf[1], 4 * ((size_t)n_lists + 1)));
    HIPCHK(hipMalloc(&W.grp_buf[2], 4 * std::max<size_t>(nm, 1)));
    doff = (uint32_t *)W.grp_buf[1];
    dmem = (uint32_t *)W.grp_buf[2];
    HIPCHK(hipMemset(doff, 0, 4 * ((size_t)n_lists + 1)));
    HIPCHK(hipMemcpy(d.sh_list_of, of.data(), 4 * (size_t)n_chan, hipMemcpyHostToDevice));
    if (n_lists) HIPCHK(hipMemcpy(doff, list_off, 4 * ((size_t)n_lists + 1), hipMemcpyHostToDevice));
    if (nm) HIPCHK(hipMemcpy(dmem, list_member_chan, 4 * (size_t)nm, hipMemcpyHostToDevice));
    d.sh_list_off = doff;
    d.sh_list_mem = dmem;
    d.sh_nlists = n_lists;
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

// ---- native collectives: RCCL inside the library (include/chd_spatial.h: chd_shard_comm_*) ----
namespace {
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// librccl.so, loaded on first use: CHD_RCCL_LIB, else the soname as the process already has it (a host that also runs
// torch.distributed has loaded its copy) or as the loader finds it, else ROCm's own
int rccl_load(chd_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return CHD_OK;
    const char *names[] = {getenv("CHD_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *so = nullptr;
    for (const char *n : names)
        if (n && n[0] && (so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!so) return fail(ctx, CHD_E_STATE, "chd_shard_comm: librccl.so not found (%s)", dlerror());
    Rccl r;
    r.so = so;
#define SYM(field, name) if (!(*(void **)(&r.field) = dlsym(so, name))) return fail(ctx, CHD_E_STATE, "chd_shard_comm: librccl.so lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = r;
    return CHD_OK;
}
#define NCCLCHK(call)                                                                                                   \
    do {                                                                                                                \
        ncclResult_t _r = (call);                                                                                       \
        if (_r != ncclSuccess) return fail(ctx, CHD_E_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)
}  // namespace

// ---- CHD_SHARD_TRANSPORT=hostpipe: a TEST transport for chd_shard_tick ----
// RCCL refuses two ranks on one device, so on a one-GPU box the library's own collectives — the segment sizes every rank must
// derive alike, the request exchange in front of the export, the order of the stages, the gated join — would only ever run with ONE
// rank.  This transport carries the same ncclSend / ncclRecv groups between PROCESSES through a POSIX shared-memory segment: one
// mailbox per ordered pair of ranks, a message staged device -> mailbox -> device by blocking copies at group end.  Same entry points,
// same call sequence, same buffers as the RCCL path; none of its asynchrony (the host blocks in every group) and none of its
// speed: for tests (tests/test_gpu_shard.py) and for nothing else.  Chosen by rank 0's environment when it draws the unique id
// (the id names the segment), so every rank of a world takes the same transport.
namespace {
struct HostPipe {
    struct Box { std::atomic<uint64_t> written, read; unsigned char _pad[112]; };  // one 128-byte header per ordered pair
    struct Op { bool send; void *dev; size_t bytes; uint32_t peer; hipStream_t st; };
    int fd = -1;
    unsigned char *base = nullptr;
    size_t total = 0, box_bytes = 0;
    uint32_t rank = 0, world = 0;
    std::string name;
    std::vector<Op> ops;
    std::vector<uint64_t> sent, rcvd;  // messages so far, per peer
    Box *hdr(uint32_t from, uint32_t to) { return (Box *)(base + 128 * ((size_t)from * world + to)); }
    unsigned char *payload(uint32_t from, uint32_t to) { return base + 128 * (size_t)world * world + box_bytes * ((size_t)from * world + to); }
};
constexpr char PIPE_MAGIC[8] = {'C', 'H', 'D', 'P', 'I', 'P', 'E', '1'};

int pipe_open(chd_ctx *ctx, const unsigned char *id, uint32_t rank, uint32_t world, size_t box_bytes) {
    World &W = ctx->w;
    auto *p = new HostPipe();
    p->rank = rank; p->world = world;
    p->box_bytes = (box_bytes + 127) & ~(size_t)127;
    p->total = 128 * (size_t)world * world + p->box_bytes * (size_t)world * world + 128;
    char nm[64];
    snprintf(nm, sizeof nm, "/chd_pipe_%02x%02x%02x%02x%02x%02x%02x%02x", id[8], id[9], id[10], id[11], id[12], id[13], id[14], id[15]);
    p->name = nm;
    p->fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
    if (p->fd < 0) { delete p; return fail(ctx, CHD_E_STATE, "hostpipe: shm_open(%s) failed", nm); }
    // (a rank that fails here takes the name with it: its peers' attach then times out instead of finding a half-made segment)
    if (ftruncate(p->fd, (off_t)p->total) != 0) { close(p->fd); shm_unlink(nm); delete p; return fail(ctx, CHD_E_STATE, "hostpipe: ftruncate(%s) failed", nm); }
    p->base = (unsigned char *)mmap(nullptr, p->total, PROT_READ | PROT_WRITE, MAP_SHARED, p->fd, 0);
    if (p->base == MAP_FAILED) { close(p->fd); shm_unlink(nm); delete p; return fail(ctx, CHD_E_STATE, "hostpipe: mmap failed"); }
    p->sent.assign(world, 0); p->rcvd.assign(world, 0);
    // the last rank to attach unlinks the name: nothing is left behind whatever happens later
    auto *attached = (std::atomic<uint32_t> *)(p->base + p->total - 128);
    if (attached->fetch_add(1) + 1 == world) shm_unlink(nm);
    W.pipe = p;
    return CHD_OK;
}

void pipe_close(World &W) {
    if (!W.pipe) return;
    if (W.pipe->base) munmap(W.pipe->base, W.pipe->total);
    if (W.pipe->fd >= 0) close(W.pipe->fd);
    delete W.pipe;
    W.pipe = nullptr;
}

// blocks until `cond` holds; a peer that never comes is an error after CHD_HOSTPIPE_TIMEOUT_S (default 60) seconds, not a hang
bool pipe_wait(const std::atomic<uint64_t> &a, uint64_t want) {
    auto cond = [&] { return a.load(std::memory_order_acquire) == want; };
    static const double limit = [] { const char *e = getenv("CHD_HOSTPIPE_TIMEOUT_S"); return e ? atof(e) : 60.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t k = 0; !cond(); k++) {
        if ((k & 1023u) == 1023u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
            std::this_thread::yield();
        }
    }
    return true;
}

int pipe_group_end(chd_ctx *ctx) {
    HostPipe &p = *ctx->w.pipe;
    std::vector<HostPipe::Op> ops;
    ops.swap(p.ops);
    for (auto &o : ops) HIPCHK(hipStreamSynchronize(o.st));  // (what is sent has been written; what is received into is no longer read)
    for (auto &o : ops) {
        if (!o.send) continue;
        if (o.bytes + 8 > p.box_bytes) return fail(ctx, CHD_E_CAPACITY, "hostpipe: a message of %zu bytes, mailboxes hold %zu", o.bytes, p.box_bytes - 8);
        auto *h = p.hdr(p.rank, o.peer);
        const uint64_t n = p.sent[o.peer];
        if (!pipe_wait(h->read, n)) return fail(ctx, CHD_E_STATE, "hostpipe: rank %u never took message %llu of rank %u", o.peer, (unsigned long long)n, p.rank);
        HIPCHK(hipMemcpy(p.payload(p.rank, o.peer), o.dev, o.bytes, hipMemcpyDeviceToHost));
        ((uint64_t *)(p.payload(p.rank, o.peer) + p.box_bytes))[-1] = o.bytes;  // (the last 8 bytes of the box: the size, checked by the receiver)
        h->written.store(n + 1, std::memory_order_release);
        p.sent[o.peer] = n + 1;
    }
    for (auto &o : ops) {
        if (o.send) continue;
        auto *h = p.hdr(o.peer, p.rank);
        const uint64_t n = p.rcvd[o.peer];
        if (!pipe_wait(h->written, n + 1)) return fail(ctx, CHD_E_STATE, "hostpipe: message %llu of rank %u never came (rank %u waits)", (unsigned long long)n, o.peer, p.rank);
        const uint64_t got = ((const uint64_t *)(p.payload(o.peer, p.rank) + p.box_bytes))[-1];
        if (got != o.bytes) return fail(ctx, CHD_E_STATE, "hostpipe: rank %u sent %llu bytes, rank %u expects %zu: the ranks disagree about a segment size", o.peer, (unsigned long long)got, p.rank, o.bytes);
        HIPCHK(hipMemcpy(o.dev, p.payload(o.peer, p.rank), o.bytes, hipMemcpyHostToDevice));
        h->read.store(n + 1, std::memory_order_release);
        p.rcvd[o.peer] = n + 1;
    }
    return CHD_OK;
}

// the transport of chd_shard_tick: RCCL, or the test transport
int xp_group_start(chd_ctx *ctx) {
    if (ctx->w.pipe) { ctx->w.pipe->ops.clear(); return CHD_OK; }
    NCCLCHK(g_rccl.GroupStart());
    return CHD_OK;
}
int xp_send(chd_ctx *ctx, const void *buf, size_t bytes, uint32_t peer, hipStream_t st) {
    if (ctx->w.pipe) { ctx->w.pipe->ops.push_back({true, const_cast<void *>(buf), bytes, peer, st}); return CHD_OK; }
    NCCLCHK(g_rccl.Send(buf, bytes, ncclUint8, (int)peer, ctx->w.comm, st));
    return CHD_OK;
}
int xp_recv(chd_ctx *ctx, void *buf, size_t bytes, uint32_t peer, hipStream_t st) {
    if (ctx->w.pipe) { ctx->w.pipe->ops.push_back({false, buf, bytes, peer, st}); return CHD_OK; }
    NCCLCHK(g_rccl.Recv(buf, bytes, ncclUint8, (int)peer, ctx->w.comm, st));
    return CHD_OK;
}
int xp_group_end(chd_ctx *ctx) {
    if (ctx->w.pipe) return pipe_group_end(ctx);
    NCCLCHK(g_rccl.GroupEnd());
    return CHD_OK;
}
}  // namespace

// the communicator, its stream and events go (the exchange buffers are the world's and stay); every handle is nulled
static int comm_teardown(chd_ctx *ctx) {
    World &W = ctx->w;
    int rc = CHD_OK;
    if (W.comm) {
        const ncclResult_t r = g_rccl.CommDestroy(W.comm);
        if (r != ncclSuccess) rc = fail(ctx, CHD_E_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
        W.comm = nullptr;
    }
    pipe_close(W);
    if (W.comm_stream) { (void)hipStreamDestroy(W.comm_stream); W.comm_stream = nullptr; }
    if (W.ev_halo_ready) { (void)hipEventDestroy(W.ev_halo_ready); W.ev_halo_ready = nullptr; }
    if (W.ev_halo_done) { (void)hipEventDestroy(W.ev_halo_done); W.ev_halo_done = nullptr; }
    W.comm_world = W.comm_rank = W.comm_cap = 0;
    ctx->gchain = ctx->gchain_prev = false;
    return rc;
}

int chd_shard_comm_available(void) {
    chd_ctx *ctx = nullptr;
    if (const char *e = test_hook_env("CHD_SHARD_TRANSPORT")) if (!strcmp(e, "hostpipe")) return CHD_OK;  // (the test transport needs no RCCL)
    return rccl_load(ctx);
}

int chd_shard_comm_unique_id(void *id_out) {
    if (!id_out) return fail(nullptr, CHD_E_INVAL, "chd_shard_comm_unique_id: NULL output");
    chd_ctx *ctx = nullptr;
    if (const char *e = test_hook_env("CHD_SHARD_TRANSPORT")) if (!strcmp(e, "hostpipe")) {  // (the TEST transport: the id names its shared-memory segment)
        memset(id_out, 0, CHD_COMM_ID_BYTES);
        memcpy(id_out, PIPE_MAGIC, 8);
        std::random_device rd;
        const uint64_t r = ((uint64_t)rd() << 32) ^ rd() ^ ((uint64_t)getpid() << 17);
        memcpy((char *)id_out + 8, &r, 8);
        return CHD_OK;
    }
    TRY(rccl_load(ctx));
    static_assert(sizeof(ncclUniqueId) == CHD_COMM_ID_BYTES, "CHD_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return CHD_OK;
}

int chd_shard_comm_init(chd_ctx *ctx, const void *unique_id, uint32_t rank, uint32_t world, uint32_t migrate_cap) {
    NEED_WORLD();
    if (!unique_id) return fail(ctx, CHD_E_INVAL, "chd_shard_comm_init: NULL unique id");
    if (!world || rank >= world || !migrate_cap) return fail(ctx, CHD_E_INVAL, "chd_shard_comm_init: rank %u of %u, capacity %u", rank, world, migrate_cap);
    if (world != ctx->g.server_cols * ctx->g.server_rows)
        return fail(ctx, CHD_E_INVAL, "chd_shard_comm_init: %u ranks but the grid has %u server regions", world, ctx->g.server_cols * ctx->g.server_rows);
    const bool hostpipe = !memcmp(unique_id, PIPE_MAGIC, 8);  // (CHD_SHARD_TRANSPORT=hostpipe where the id was drawn: the TEST transport)
#ifdef CHD_NO_TEST_HOOKS
    if (hostpipe) return fail(ctx, CHD_E_STATE, "chd_shard_comm_init: the id names the hostpipe TEST transport, which this build leaves out (CHD_NO_TEST_HOOKS)");
#endif
    if (!hostpipe) TRY(rccl_load(ctx));
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    if (W.comm || W.pipe) return fail(ctx, CHD_E_STATE, "chd_shard_comm_init: the ctx already has a communicator");
    if (W.slot_mode == 1) return fail(ctx, CHD_E_STATE, "chd_shard_comm_init on a world with caller-chosen slots (chd_world_spawn)");
    // this rank's halo layout (installs it) and the exchange buffers: the library owns them
    W.halo_segs.assign(world, chd_halo_seg{0, 0, 0, 0});
    uint64_t st = 0, rt = 0;
    TRY(shard_halo_layout_locked(ctx, rank, world, W.halo_segs.data(), &st, &rt));
    // the communicator FIRST (a collective: every rank is inside it, or none gets out); only a rank that has one allocates the
    // exchange buffers, the second stream and the events — a failed init leaves nothing behind and may be retried
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (hostpipe) {
        // one mailbox per ordered pair of ranks, as large as the largest message of the protocol: an emigrant segment, a request
        // segment, any pair's halo segment (a pure function of the grid and max_entities: every rank computes the same)
        const DevGrid &g = ctx->g;
        size_t box = std::max(((size_t)migrate_cap + 1 + migrate_extra(ctx)) * sizeof(chd_entity_state), ((size_t)CHD_SHARD_REQ_CAP + 1) * sizeof(chd_handover_request));
        for (uint32_t a = 0; a < world; a++)
            for (uint32_t b = 0; b < world; b++) {
                const HaloRect r = halo_rect(g.cols, g.rows, g.server_cols, g.sgc, g.sgr, g.border, a, b);
                box = std::max<size_t>(box, halo_seg_bytes(halo_cap_entries(W.d.N, r.w * r.h, g.sgc * g.sgr), r.w * r.h));
            }
        TRY(pipe_open(ctx, (const unsigned char *)unique_id, rank, world, box + 8));
    } else {
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        ncclComm_t comm = nullptr;
        NCCLCHK(g_rccl.CommInitRank(&comm, (int)world, id, (int)rank));
        W.comm = comm;
    }
    W.comm_rank = rank;
    W.comm_world = world;
    W.comm_cap = migrate_cap;
    const size_t seg = (size_t)migrate_cap + 1 + migrate_extra(ctx);
    int rc = CHD_OK;
    if (!W.comm_alloc_cap) {  // (world allocations: kept across chd_shard_comm_destroy / a second init of the same size)
        // each buffer on its own pointer: a call that failed half-way is repeated from where it stopped, and the sizes are noted
        // only once all four exist
        if (W.mig_send && (W.comm_try_cap != migrate_cap || W.comm_try_world != world))
            rc = fail(ctx, CHD_E_STATE, "chd_shard_comm_init: an earlier, failed call began to size the exchange buffers for %u ranks x %u emigrants", W.comm_try_world, W.comm_try_cap);
        W.comm_try_cap = migrate_cap; W.comm_try_world = world;
        if (rc == CHD_OK && !W.mig_send) rc = walloc(ctx, &W.mig_send, (size_t)world * seg);
        if (rc == CHD_OK && !W.mig_recv) rc = walloc(ctx, &W.mig_recv, (size_t)world * seg);
        if (rc == CHD_OK && !W.halo_send_buf) rc = walloc(ctx, &W.halo_send_buf, std::max<uint64_t>(st, 16));
        if (rc == CHD_OK && !W.halo_recv_buf) rc = walloc(ctx, &W.halo_recv_buf, std::max<uint64_t>(rt, 16));
        if (rc == CHD_OK) { W.comm_alloc_cap = migrate_cap; W.comm_alloc_world = world; }
    } else if (W.comm_alloc_cap != migrate_cap || W.comm_alloc_world != world) {
        rc = fail(ctx, CHD_E_STATE, "chd_shard_comm_init: the context's exchange buffers were sized for %u ranks x %u emigrants", W.comm_alloc_world, W.comm_alloc_cap);
    }
    if (rc == CHD_OK && hipStreamCreateWithFlags(&W.comm_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(ctx, CHD_E_HIP, "chd_shard_comm_init: no second stream");
    if (rc == CHD_OK && (hipEventCreateWithFlags(&W.ev_halo_ready, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&W.ev_halo_done, hipEventDisableTiming) != hipSuccess))
        rc = fail(ctx, CHD_E_HIP, "chd_shard_comm_init: no events");
    if (rc != CHD_OK) {
        const std::string why = ctx->err;
        comm_teardown(ctx);
        return fail(ctx, rc, "%s", why.c_str());
    }
    return CHD_OK;
}

int chd_shard_comm_destroy(chd_ctx *ctx) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    World &W = ctx->w;
    if (!W.comm && !W.pipe) return CHD_OK;
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (W.comm_stream) HIPCHK(hipStreamSynchronize(W.comm_stream));
    return comm_teardown(ctx);
}

// One tick of a region-sharded world with both exchanges inside: no host code between the stages, everything ordered by
// streams, events and device-side flags.  Default form — ctx stream: ingest + export -> all-to-all of the emigrants (ncclSend /
// ncclRecv group, equal segments of this tick's capacity) -> import + cell index + halo pack -> [event] -> interest updates ->
// [wait] -> fan-out; the halo all-to-all(v) runs on the library's second stream between the two events, beside the interest
// updates (which do not read the neighbours' tables).  Gated form (CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP): the
// interest updates on the second stream from the tick's start, everything else — the halo exchange included — in order on the ctx
// stream, the join as a flag that k_halo_unpack waits for (GateArgs).
int chd_shard_tick(chd_ctx *ctx, int64_t now_ns, const double *d_x_by_chan, const double *d_z_by_chan, const uint8_t *d_has_update,
                   uint32_t n_chan, const chd_tick_in *d_in) {
    NEED_WORLD();
    if (!d_in) return fail(ctx, CHD_E_INVAL, "chd_shard_tick: NULL input");
    if (n_chan && (!d_x_by_chan || !d_z_by_chan)) return fail(ctx, CHD_E_INVAL, "chd_shard_tick: NULL positions");
    std::lock_guard<FairMutex> lk(ctx->mu);
    World &W = ctx->w;
    if (!W.comm && !W.pipe) return fail(ctx, CHD_E_STATE, "chd_shard_tick before chd_shard_comm_init");
    TRY(check_queries(ctx, d_in));  // (before anything is enqueued: a refused tick leaves the world — and the gate counters — untouched)
    const uint32_t world = W.comm_world, rank = W.comm_rank;
    hipStream_t st = ctx->stream;
    uint32_t use = W.comm_cap;
    // CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP: the interest updates — they read nothing this tick's front writes —
    // run on the second stream from the tick's START, beside ingest, both exchanges and the index build, and are joined by a
    // device-side flag before the plan; the halo exchange then needs no stream of its own (and no events).  `chained`: the call
    // before this one on the context was such a tick too (read before the stages below pass through bind()).
    const bool g_on = W.gated && !W.plan_recipients;
    const bool chained = g_on && ctx->gchain;
    if (world > 1 && W.d.sh_list_of) {
        // handover lists: a handover whose src map is another rank's goes there as a request before anything is exported
        // (k_shard.hip: k_apply_requests) — one more small exchange, only on worlds with lists, every rank alike
        const uint32_t rc = CHD_SHARD_REQ_CAP;
        if (!W.req_send) {
            TRY(walloc(ctx, &W.req_send, (size_t)world * (rc + 1)));
            TRY(walloc(ctx, &W.req_recv, (size_t)world * (rc + 1)));
        }
        TRY(shard_ingest_pre_locked(ctx, now_ns, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, rank, world, W.req_send, rc));
        TRY(xp_group_start(ctx));
        for (uint32_t p = 0; p < world; p++) {
            TRY(xp_send(ctx, W.req_send + p * (rc + 1), (rc + 1) * sizeof(chd_handover_request), p, st));
            TRY(xp_recv(ctx, W.req_recv + p * (rc + 1), (rc + 1) * sizeof(chd_handover_request), p, st));
        }
        TRY(xp_group_end(ctx));
        TRY(shard_ingest_post_locked(ctx, W.req_recv, rc, rank, world, W.mig_send, W.comm_cap, &use));
    } else {
        TRY(shard_ingest_pre_locked(ctx, now_ns, d_x_by_chan, d_z_by_chan, d_has_update, n_chan, rank, world, nullptr, 0));
        TRY(shard_ingest_post_locked(ctx, nullptr, 0, rank, world, W.mig_send, W.comm_cap, &use));
    }
    if (g_on && d_in->n_queries) {
        hipStream_t ax = ctx->aux_stream;
        if (chained) launch_gate_wait(ax, W.d, W.gate + GATE_EPI, W.gate_epi, true);  // after the previous tick's epilogue
        else {
            HIPCHK(hipEventRecord(ctx->ev_fork, st));  // (recorded behind this tick's ingest: harmless, the two are independent)
            HIPCHK(hipStreamWaitEvent(ax, ctx->ev_fork, 0));
        }
        W.gate_top++;
        launch_aoi_interest(ax, ctx->g, ctx->lim, W.d, d_in->queries, d_in->n_queries, d_in->query_sub, d_in->spot_x, d_in->spot_z,
                            d_in->spot_dist, W.last_now, ctx->ring.cur_tick, W.gate + GATE_TOP, W.gate_top);
    }
    const size_t seg = (size_t)use + 1 + migrate_extra(ctx);
    {   // the cross-server handovers (spatial.go:683-700): every rank's segment for every other rank, 32 B per emigrant.
        // (One rank: its own segment to itself — nothing to move, but the tick keeps its shape and the transport is exercised.)
        TRY(xp_group_start(ctx));
        for (uint32_t p = 0; p < world; p++) {
            TRY(xp_send(ctx, W.mig_send + p * seg, seg * sizeof(chd_entity_state), p, st));
            TRY(xp_recv(ctx, W.mig_recv + p * seg, seg * sizeof(chd_entity_state), p, st));
        }
        TRY(xp_group_end(ctx));
    }
    TRY(shard_import_locked(ctx, W.mig_recv, world, use, W.halo_send_buf));
    bool halo = false;
    for (uint32_t p = 0; p < world; p++) halo = halo || W.halo_segs[p].send_bytes || W.halo_segs[p].recv_bytes;
    if (halo && g_on) {  // (nothing to overlap it with on this stream any more: in stream order, no events)
        TRY(xp_group_start(ctx));
        for (uint32_t p = 0; p < world; p++) {
            const chd_halo_seg &h = W.halo_segs[p];
            if (h.send_bytes) TRY(xp_send(ctx, W.halo_send_buf + h.send_off, h.send_bytes, p, st));
            if (h.recv_bytes) TRY(xp_recv(ctx, W.halo_recv_buf + h.recv_off, h.recv_bytes, p, st));
        }
        TRY(xp_group_end(ctx));
    } else if (halo) {
        HIPCHK(hipEventRecord(W.ev_halo_ready, st));
        HIPCHK(hipStreamWaitEvent(W.comm_stream, W.ev_halo_ready, 0));
        TRY(xp_group_start(ctx));
        for (uint32_t p = 0; p < world; p++) {
            const chd_halo_seg &h = W.halo_segs[p];
            if (h.send_bytes) TRY(xp_send(ctx, W.halo_send_buf + h.send_off, h.send_bytes, p, W.comm_stream));
            if (h.recv_bytes) TRY(xp_recv(ctx, W.halo_recv_buf + h.recv_off, h.recv_bytes, p, W.comm_stream));
        }
        TRY(xp_group_end(ctx));
        HIPCHK(hipEventRecord(W.ev_halo_done, W.comm_stream));
    }
    if (g_on) {
        W.join_in_unpack = d_in->n_queries != 0;  // join: the ghost unpack (first launch of shard_fanout_locked) waits for the interest updates
    } else {
        TRY(shard_interest_locked(ctx, d_in));
        if (halo) HIPCHK(hipStreamWaitEvent(st, W.ev_halo_done, 0));
    }
    chd_tick_in rest = *d_in;
    rest.n_queries = 0;  // (the interest updates ran above)
    rest.queries = nullptr;
    const uint32_t nq = d_in->n_queries;
    TRY(shard_fanout_locked(ctx, W.halo_recv_buf, world, &rest));
    W.last_nq = nq;
    ctx->gchain = g_on;  // (shard_fanout_locked's epilogue raised the flag the next tick's second stream waits for)
    return CHD_OK;
}

int chd_shard_get_entities(chd_ctx *ctx, uint32_t *chan_id, uint32_t *cell_channel, uint32_t *member_channel, uint32_t *n_out) {
    NEED_WORLD();
    if (!chan_id || !n_out) return fail(ctx, CHD_E_INVAL, "chd_shard_get_entities: NULL buffer");
    WorldDev &d = ctx->w.d;
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    std::vector<uint32_t> ch(d.N), cell(d.N), mem(d.N), fl(d.N);
    TRY(down(ctx, ch.data(), d.chan_id, 4 * (size_t)d.N));
    TRY(down(ctx, cell.data(), d.cell, 4 * (size_t)d.N));
    TRY(down(ctx, mem.data(), d.member, 4 * (size_t)d.N));
    TRY(down(ctx, fl.data(), d.eflags, 4 * (size_t)d.N));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint32_t n = 0;
    for (uint32_t i = 0; i < d.N; i++) {
        if (!(fl[i] & EF_ALIVE)) continue;
        chan_id[n] = ch[i];
        if (cell_channel) cell_channel[n] = cell[i] == CHD_INVALID ? 0u : cell[i] + ctx->g.id_start;
        if (member_channel) member_channel[n] = mem[i] == CHD_INVALID ? 0u : mem[i] + ctx->g.id_start;
        n++;
    }
    *n_out = n;
    return CHD_OK;
}

// ---------------------------------------------------------------------------
// wire-format fan-out buffers
// ---------------------------------------------------------------------------

int chd_wire_set_payloads(chd_ctx *ctx, int kind, uint32_t n, const uint32_t *idx, const uint32_t *lens, const uint8_t *bytes) {
    NEED_WORLD();
    World &W = ctx->w;
    if (!W.wire) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_WIRE");
    if (kind < 0 || kind > CHD_WIRE_ENTITY_OBJREF) return fail(ctx, CHD_E_INVAL, "chd_wire_set_payloads: kind %d", kind);
    if (!n) return CHD_OK;
    if (!idx || !lens) return fail(ctx, CHD_E_INVAL, "chd_wire_set_payloads: NULL buffer");
    const bool objref = kind == CHD_WIRE_ENTITY_OBJREF;
    const int full = objref ? 0 : (kind & 1), cell = objref ? 0 : (kind >> 1);
    std::vector<uint64_t> off(n);
    std::vector<uint32_t> ix(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (lens[i] > W.x.stride[full])
            return fail(ctx, CHD_E_CAPACITY, "payload %u: %u bytes exceed wire_max_%s_len (%u)", i, lens[i], full ? "full" : "update", W.x.stride[full]);
        uint32_t v = idx[i];
        if (cell) {
            if (v < ctx->g.id_start || v - ctx->g.id_start >= ctx->g.ncell) return fail(ctx, CHD_E_INVAL, "payload %u: %u is not a spatial channel", i, v);
            v -= ctx->g.id_start;
        } else if (v >= W.x.npay) {
            return fail(ctx, CHD_E_INVAL, "payload %u: entity %s %u out of range", i, W.d.ce_by_chan ? "channel index" : "slot", v);
        }
        ix[i] = v;
        off[i] = total;
        total += lens[i];
    }
    if (total && !bytes) return fail(ctx, CHD_E_INVAL, "chd_wire_set_payloads: NULL bytes");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(ensure(ctx, 0, 4 * (size_t)n)); TRY(ensure(ctx, 1, 4 * (size_t)n)); TRY(ensure(ctx, 2, 8 * (size_t)n));
    TRY(ensure(ctx, 3, std::max<uint64_t>(total, 16)));
    TRY(up(ctx, sbuf<void>(ctx, 0), ix.data(), 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 1), lens, 4 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 2), off.data(), 8 * (size_t)n));
    TRY(up(ctx, sbuf<void>(ctx, 3), bytes, total));
    // (merge mode: an UPDATE payload set now belongs to the update that arrives with the NEXT tick)
    launch_wire_set_payloads(ctx->stream, W.x, objref ? 2 : full, cell, n, cell ? ctx->g.ncell : W.x.npay, sbuf<uint32_t>(ctx, 0),
                             sbuf<uint32_t>(ctx, 1), sbuf<uint64_t>(ctx, 2), sbuf<uint8_t>(ctx, 3),
                             (ctx->ring.cur_tick + 1u) & (CHD_HIST_BITS - 1u));
    TRY(after_launch(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging vectors go out of scope
    return CHD_OK;
}

int chd_wire_set_type_url(chd_ctx *ctx, int which, const uint8_t *url, uint32_t len) {
    NEED_WORLD();
    World &W = ctx->w;
    if (!W.wire) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_WIRE");
    if (which < 0 || which > 2) return fail(ctx, CHD_E_INVAL, "chd_wire_set_type_url: which = %d", which);
    if (which < 2 && !W.x.merge) return fail(ctx, CHD_E_STATE, "update type urls belong to worlds with CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS");
    if (len > 255 || (len && !url)) return fail(ctx, CHD_E_INVAL, "chd_wire_set_type_url: at most 255 bytes");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (len) HIPCHK(hipMemcpy(W.x.url[which], url, len, hipMemcpyHostToDevice));
    W.x.url_len[which] = len;
    return CHD_OK;
}

int chd_wire_set_merge_schema(chd_ctx *ctx, int schema) {
    NEED_WORLD();
    if (schema != CHD_MERGE_SCHEMA_NONE && schema != CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT) return fail(ctx, CHD_E_INVAL, "chd_wire_set_merge_schema: unknown schema %d", schema);
    std::lock_guard<FairMutex> lk(ctx->mu);
    World &W = ctx->w;
    if (!W.wire || !W.x.merge) return fail(ctx, CHD_E_STATE, "merge schemas belong to worlds with CHD_WORLD_WIRE | CHD_WORLD_UPDATE_MASKS");
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    W.x.schema = (uint32_t)schema;
    W.wire_built = false;
    return CHD_OK;
}

int chd_handover_messages(chd_ctx *ctx, uint32_t n_handovers, uint32_t *offsets, uint8_t *bytes, uint64_t cap, uint64_t *n_out) {
    NEED_WORLD();
    if (!offsets || !n_out || (cap && !bytes)) return fail(ctx, CHD_E_INVAL, "chd_handover_messages: NULL buffer");
    World &W = ctx->w;
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!W.wire) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_WIRE");
    if (W.d.ce_by_chan) return fail(ctx, CHD_E_STATE, "handover message assembly stays with the host on region-sharded worlds (the entity data is looked up by slot)");
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick yet");
    TRY(bind(ctx));
    uint64_t ringrow[8];
    TRY(down(ctx, ringrow, W.d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t nh = std::min<uint32_t>((uint32_t)ringrow[2], W.d.handovers_cap);
    offsets[0] = 0;
    *n_out = 0;
    if (nh > n_handovers)
        return fail(ctx, CHD_E_CAPACITY, "chd_handover_messages: the last tick had %u handovers, offsets has room for %u", nh, n_handovers);
    if (!nh) return CHD_OK;
    TRY(ensure(ctx, 0, sizeof(uint32_t) * (2 * (size_t)nh + 1)));
    launch_handover_msg_sizes(ctx->stream, ctx->g, W.d, W.x, 2 * nh, nullptr, nullptr, sbuf<uint32_t>(ctx, 0));
    launch_scan_u32_inplace(ctx->stream, sbuf<uint32_t>(ctx, 0), 2 * nh);
    TRY(after_launch(ctx));
    TRY(down(ctx, offsets, sbuf<void>(ctx, 0), sizeof(uint32_t) * (2 * (size_t)nh + 1)));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = offsets[2 * nh];
    *n_out = total;
    if (total > cap) return fail(ctx, CHD_E_CAPACITY, "chd_handover_messages: %llu bytes, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
    TRY(ensure(ctx, 1, std::max<uint64_t>(total, 16)));
    launch_handover_msg_write(ctx->stream, ctx->g, W.d, W.x, 2 * nh, nullptr, nullptr, sbuf<uint32_t>(ctx, 0), sbuf<uint8_t>(ctx, 1), total);
    TRY(after_launch(ctx));
    TRY(down(ctx, bytes, sbuf<void>(ctx, 1), total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_handover_variants(chd_ctx *ctx, uint32_t n_var, const uint32_t *var_handover, const uint32_t *var_full_mask, uint32_t *offsets,
                          uint8_t *bytes, uint64_t cap, uint64_t *n_out) {
    NEED_WORLD();
    if (!offsets || !n_out || (n_var && (!var_handover || !var_full_mask)) || (cap && !bytes)) return fail(ctx, CHD_E_INVAL, "chd_handover_variants: NULL buffer");
    World &W = ctx->w;
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!W.wire) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_WIRE");
    if (W.d.ce_by_chan) return fail(ctx, CHD_E_STATE, "handover message assembly stays with the host on region-sharded worlds (the entity data is looked up by slot)");
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick yet");
    TRY(bind(ctx));
    uint64_t ringrow[8];
    TRY(down(ctx, ringrow, W.d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t nh = std::min<uint32_t>((uint32_t)ringrow[2], W.d.handovers_cap);
    offsets[0] = 0;
    *n_out = 0;
    for (uint32_t v = 0; v < n_var; v++)
        if (var_handover[v] >= nh) return fail(ctx, CHD_E_INVAL, "chd_handover_variants: variant %u names handover %u of %u", v, var_handover[v], nh);
    if (!n_var) return CHD_OK;
    TRY(ensure(ctx, 0, sizeof(uint32_t) * ((size_t)n_var + 1)));
    TRY(ensure(ctx, 2, sizeof(uint32_t) * (size_t)n_var));
    TRY(ensure(ctx, 3, sizeof(uint32_t) * (size_t)n_var));
    TRY(up(ctx, sbuf<void>(ctx, 2), var_handover, sizeof(uint32_t) * (size_t)n_var));
    TRY(up(ctx, sbuf<void>(ctx, 3), var_full_mask, sizeof(uint32_t) * (size_t)n_var));
    launch_handover_msg_sizes(ctx->stream, ctx->g, W.d, W.x, n_var, sbuf<uint32_t>(ctx, 2), sbuf<uint32_t>(ctx, 3), sbuf<uint32_t>(ctx, 0));
    launch_scan_u32_inplace(ctx->stream, sbuf<uint32_t>(ctx, 0), n_var);
    TRY(after_launch(ctx));
    TRY(down(ctx, offsets, sbuf<void>(ctx, 0), sizeof(uint32_t) * ((size_t)n_var + 1)));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = offsets[n_var];
    *n_out = total;
    if (total > cap) return fail(ctx, CHD_E_CAPACITY, "chd_handover_variants: %llu bytes, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
    TRY(ensure(ctx, 1, std::max<uint64_t>(total, 16)));
    launch_handover_msg_write(ctx->stream, ctx->g, W.d, W.x, n_var, sbuf<uint32_t>(ctx, 2), sbuf<uint32_t>(ctx, 3), sbuf<uint32_t>(ctx, 0), sbuf<uint8_t>(ctx, 1), total);
    TRY(after_launch(ctx));
    TRY(down(ctx, bytes, sbuf<void>(ctx, 1), total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_wire_build(chd_ctx *ctx, uint64_t *total_bytes, uint64_t *total_packets, uint32_t *dropped) {
    NEED_WORLD();
    World &W = ctx->w;
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!W.wire) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_WIRE");
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick to build the wire buffers of");
    if (W.slot_mode == 2 && !W.d.ce_by_chan)
        return fail(ctx, CHD_E_STATE, "wire buffers on a region-sharded world need chd_world_cfg.shard_channels (payloads by channel id)");
    TRY(bind(ctx));
    WorldDev &d = W.d;
    hipStream_t st = ctx->stream;
    // (a second call for the same tick returns the streams already built: the layout pass rewrote the records' position words)
    const bool again = W.wire_built;
    // streams from the fan-out descriptors where the tick has them (k_wire_layout_img), else from the records (k_wire_layout)
    const bool img = W.x.img_on && W.last_desc && !d.cm_emit;
    auto write_img = [&]() {  // (device-guarded: does nothing unless both arenas hold what the sizing pass found)
        W.x.bytes_cap = W.wire_cap;
        W.x.cdesc_cap = W.cdesc_cap;
        launch_wire_layout_img(st, ctx->g, d, W.x, true);
        launch_wire_copy_img(st, d, W.x, d.seg_waves ? d.seg_waves : 2048u);
    };
    if (!again) {
        HIPCHK(hipMemsetAsync(W.x.n_dropped, 0, 8 * sizeof(uint32_t), st));
        W.x.cur_tick = ctx->ring.cur_tick;
        if (img) {
            HIPCHK(hipMemsetAsync(W.x.cp_ticket, 0, 32 * sizeof(uint32_t), st));
            HIPCHK(hipMemsetAsync(W.x.cell_dcnt, 0, ((size_t)W.x.ncell * W.x.dpad + 1) * sizeof(uint32_t), st));
            launch_wire_images(st, ctx->g, d, W.x);
            launch_wire_layout_img(st, ctx->g, d, W.x, false);
            launch_wire_images_fill(st, ctx->g, d, W.x);
            launch_wire_conn_order(st, d, W.x);
        } else {
            launch_wire_layout(st, d, W.x);
        }
        launch_scan_u64_inplace(st, W.x.conn_wlen, d.S);
        // no host round trip between sizing and writing on the descriptor path: the writing kernels are enqueued for the
        // arenas as they are and check on the device that they suffice
        if (img) write_img();
        TRY(after_launch(ctx));
    }
    uint64_t total = 0;
    uint32_t ndrop2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ndesc = 0;
    TRY(down(ctx, &total, W.x.conn_wlen + d.S, sizeof total));
    TRY(down(ctx, ndrop2, W.x.n_dropped, sizeof ndrop2));
    if (img && !again) TRY(down(ctx, &ndesc, W.x.rank_ndesc + d.S, sizeof ndesc));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t ndrop = ndrop2[0];
    if (ndrop2[1] || ndrop2[2]) return fail(ctx, CHD_E_STATE, "chd_wire_build: %u records of this tick carry no valid position word, %u name no entity slot", ndrop2[1], ndrop2[2]);
    if (ndrop2[6]) return fail(ctx, CHD_E_STATE, "chd_wire_build: the packet split of %u connections made no progress (a bug)", ndrop2[6]);
    if (!again) {
        const bool grow_bytes = total > W.wire_cap, grow_desc = img && ndesc > W.cdesc_cap;
        if (grow_bytes) {
            if (W.x.bytes) HIPCHK(hipFree(W.x.bytes));
            W.x.bytes = nullptr;
            W.wire_cap = total + total / 2 + 4096;  // head-room: the arena is only re-allocated when a tick outgrows it
            HIPCHK(hipMalloc((void **)&W.x.bytes, W.wire_cap));
        }
        if (grow_desc) {
            if (W.x.cdesc) HIPCHK(hipFree(W.x.cdesc));
            W.x.cdesc = nullptr;
            W.cdesc_cap = (uint64_t)ndesc + ndesc / 2 + 4096;
            HIPCHK(hipMalloc((void **)&W.x.cdesc, W.cdesc_cap * sizeof(uint4)));
        }
        if (img) {
            if (grow_bytes || grow_desc) write_img();  // (the guarded launches above did nothing)
        } else {
            launch_wire_copy(st, d, W.x, ndrop2[4]);
        }
        W.wire_ranges = img ? ndesc : 0;
        W.wire_slow_conns = ndrop2[4];
        TRY(after_launch(ctx));
        uint32_t bad = ndrop2[3];
        if (!img || grow_bytes || grow_desc) {  // (else everything ran before the synchronisation above)
            TRY(down(ctx, &bad, W.x.n_dropped + 3, sizeof bad));
            HIPCHK(hipStreamSynchronize(st));
        }
        if (bad) {
            uint32_t g[16] = {0};
            TRY(down(ctx, g, W.x.n_dropped + 8, sizeof g));
            HIPCHK(hipStreamSynchronize(st));
            return fail(ctx, CHD_E_STATE, "chd_wire_build: %u chunks of records lie outside their connection's stream; first: conn %u lo %u hi %u len %u count %u "
                        "live %08x%08x woff0 %u | conn %u lo %u hi %u len %u count %u live %08x%08x woff0 %u", bad, g[0], g[1], g[2], g[3], g[4], g[6], g[5], g[7],
                        g[8], g[9], g[10], g[11], g[12], g[14], g[13], g[15]);
        }
        W.wire_built = true;
    }
    if (total_bytes) *total_bytes = total;
    if (dropped) *dropped = ndrop;
    if (total_packets) {
        std::vector<uint32_t> npk(d.S);
        TRY(down(ctx, npk.data(), W.x.conn_npk, sizeof(uint32_t) * d.S));
        HIPCHK(hipStreamSynchronize(st));
        uint64_t t = 0;
        for (uint32_t v : npk) t += v;
        *total_packets = t;
    }
    return CHD_OK;
}

int chd_wire_build_info(chd_ctx *ctx, uint64_t *n_image_ranges, uint32_t *n_record_path_connections) {
    NEED_WORLD();
    World &W = ctx->w;
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!W.wire_built) return fail(ctx, CHD_E_STATE, "chd_wire_build has not run for the last tick");
    if (n_image_ranges) *n_image_ranges = W.wire_ranges;
    if (n_record_path_connections) *n_record_path_connections = W.wire_slow_conns;
    return CHD_OK;
}

int chd_wire_fetch(chd_ctx *ctx, uint64_t *conn_off, uint32_t *conn_packets, uint8_t *bytes, uint64_t cap) {
    NEED_WORLD();
    World &W = ctx->w;
    if (!conn_off) return fail(ctx, CHD_E_INVAL, "chd_wire_fetch: NULL conn_off");
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (!W.wire_built) return fail(ctx, CHD_E_STATE, "chd_wire_build has not run for the last tick");
    TRY(bind(ctx));
    TRY(down(ctx, conn_off, W.x.conn_woff, sizeof(uint64_t) * ((size_t)W.d.S + 1)));
    if (conn_packets) TRY(down(ctx, conn_packets, W.x.conn_npk, sizeof(uint32_t) * W.d.S));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = conn_off[W.d.S];
    if (bytes) {
        if (total > cap) return fail(ctx, CHD_E_CAPACITY, "chd_wire_fetch: %llu bytes, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
        TRY(down(ctx, bytes, W.x.bytes, total));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return CHD_OK;
}

int chd_world_set_server_connections(chd_ctx *ctx, uint32_t n_servers, const uint32_t *conn_ids) {
    NEED_WORLD();
    if (n_servers && !conn_ids) return fail(ctx, CHD_E_INVAL, "chd_world_set_server_connections: NULL ids");
    if (n_servers && n_servers != ctx->g.server_cols * ctx->g.server_rows)
        return fail(ctx, CHD_E_INVAL, "chd_world_set_server_connections: %u ids but the grid has %u server regions", n_servers, ctx->g.server_cols * ctx->g.server_rows);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    World &W = ctx->w;
    HIPCHK(hipStreamSynchronize(ctx->stream));  // (a planned tick may still read the old table)
    if (!n_servers) { W.d.server_conn = nullptr; W.d.n_server_conn = 0; return CHD_OK; }
    if (!W.server_conn) TRY(walloc(ctx, &W.server_conn, ctx->g.server_cols * ctx->g.server_rows));
    TRY(up(ctx, W.server_conn, conn_ids, sizeof(uint32_t) * n_servers));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    W.d.server_conn = W.server_conn;
    W.d.n_server_conn = n_servers;
    return CHD_OK;
}

int chd_handover_recipients(chd_ctx *ctx, uint32_t *offsets, uint32_t *conn, uint8_t *kind, uint64_t cap, uint64_t *n_out) {
    return chd_handover_recipients_ex(ctx, offsets, conn, kind, nullptr, cap, n_out);
}

int chd_handover_recipients_ex(chd_ctx *ctx, uint32_t *offsets, uint32_t *conn, uint8_t *kind, uint32_t *full_mask, uint64_t cap, uint64_t *n_out) {
    NEED_WORLD();
    if (!offsets || !n_out || (cap && (!conn || !kind))) return fail(ctx, CHD_E_INVAL, "chd_handover_recipients: NULL buffer");
    World &W = ctx->w;
    if (!W.plan_recipients) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_HANDOVER_RECIPIENTS");
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick yet");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    uint64_t ringrow[8];
    TRY(down(ctx, ringrow, W.d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t nh = std::min<uint32_t>((uint32_t)ringrow[2], W.d.handovers_cap);
    TRY(down(ctx, offsets, W.ho_rcp_off, sizeof(uint32_t) * ((size_t)nh + 1)));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = offsets[nh];
    *n_out = total;
    if (total > W.ho_rcp_cap) return fail(ctx, CHD_E_CAPACITY, "handover recipients: %llu exceed the engine capacity %llu", (unsigned long long)total, (unsigned long long)W.ho_rcp_cap);
    if (total > cap) return fail(ctx, CHD_E_CAPACITY, "handover recipients: %llu, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
    TRY(down(ctx, conn, W.ho_rcp_conn, sizeof(uint32_t) * total));
    TRY(down(ctx, kind, W.ho_rcp_kind, total));
    if (full_mask) TRY(down(ctx, full_mask, W.ho_rcp_mask, sizeof(uint32_t) * total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_handover_src_owner_unsubscribed(chd_ctx *ctx, uint8_t *flags, uint32_t cap, uint32_t *n_out) {
    NEED_WORLD();
    if (!n_out || (cap && !flags)) return fail(ctx, CHD_E_INVAL, "chd_handover_src_owner_unsubscribed: NULL buffer");
    World &W = ctx->w;
    if (!W.plan_recipients) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_HANDOVER_RECIPIENTS");
    if (W.slot_mode == 2) return fail(ctx, CHD_E_STATE, "chd_handover_src_owner_unsubscribed on a region-sharded world: chd_shard_handover_recipients");
    if (!W.ticked) return fail(ctx, CHD_E_STATE, "no tick yet");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    uint64_t ringrow[8];
    TRY(down(ctx, ringrow, W.d.tick_ring + (size_t)(ctx->ring.cur_tick % TICK_RING) * 8, sizeof ringrow));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t nh = std::min<uint32_t>((uint32_t)ringrow[2], W.d.handovers_cap);
    *n_out = nh;
    if (nh > cap) return fail(ctx, CHD_E_CAPACITY, "chd_handover_src_owner_unsubscribed: %u handovers, capacity %u", nh, cap);
    std::vector<uint32_t> own(nh);
    TRY(down(ctx, own.data(), W.ho_rcp_own, sizeof(uint32_t) * (size_t)nh));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t h = 0; h < nh; h++) flags[h] = own[h] ? 1 : 0;
    return CHD_OK;
}

int chd_shard_handover_recipients(chd_ctx *ctx, uint32_t n_handovers, const chd_handover_rec *handovers, uint32_t *offsets, uint32_t *conn, uint8_t *kind,
                                  uint32_t *full_mask, uint8_t *src_owner_unsubscribed, uint64_t cap, uint64_t *n_out) {
    NEED_WORLD();
    if (!offsets || !n_out || (n_handovers && !handovers) || (cap && (!conn || !kind))) return fail(ctx, CHD_E_INVAL, "chd_shard_handover_recipients: NULL buffer");
    World &W = ctx->w;
    if (!W.plan_recipients) return fail(ctx, CHD_E_STATE, "the world was created without CHD_WORLD_HANDOVER_RECIPIENTS");
    if (W.slot_mode != 2 || !W.ticked) return fail(ctx, CHD_E_STATE, "chd_shard_handover_recipients: no tick of a region-sharded world yet");
    if (full_mask && W.d.sh_list_of) return fail(ctx, CHD_E_STATE, "chd_shard_handover_recipients: full_mask on a world with handover lists (the members' cells are other ranks' state)");
    for (uint32_t h = 0; h < n_handovers; h++)
        if (handovers[h].src < ctx->g.id_start || handovers[h].src - ctx->g.id_start >= ctx->g.ncell || handovers[h].dst < ctx->g.id_start ||
            handovers[h].dst - ctx->g.id_start >= ctx->g.ncell)
            return fail(ctx, CHD_E_INVAL, "chd_shard_handover_recipients: handover %u is not between two spatial channels", h);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    offsets[0] = 0;
    *n_out = 0;
    if (!n_handovers) return CHD_OK;
    if (W.rcp_recs_cap < n_handovers) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        W.rcp_recs_cap = n_handovers + n_handovers / 4 + 256;
        TRY(walloc(ctx, &W.rcp_recs, W.rcp_recs_cap, false));
        TRY(walloc(ctx, &W.rcp_off, (size_t)W.rcp_recs_cap + 1));
        TRY(walloc(ctx, &W.rcp_own, (size_t)W.rcp_recs_cap + 1));
        if (!W.rcp_ctr) TRY(walloc(ctx, &W.rcp_ctr, CTR_COUNT));
    }
    const uint64_t rcap = std::min<uint64_t>((uint64_t)n_handovers * W.d.S, 1ull << 25);
    if (W.ho_rcp_cap < rcap || !W.ho_rcp_conn) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        W.ho_rcp_cap = rcap;
        TRY(walloc(ctx, &W.ho_rcp_conn, W.ho_rcp_cap, false));
        TRY(walloc(ctx, &W.ho_rcp_kind, W.ho_rcp_cap, false));
        TRY(walloc(ctx, &W.ho_rcp_mask, W.ho_rcp_cap, false));
    }
    hipStream_t st = ctx->stream;
    TRY(up(ctx, W.rcp_recs, handovers, sizeof(chd_handover_rec) * (size_t)n_handovers));
    HIPCHK(hipMemsetAsync(W.rcp_ctr, 0, sizeof(uint32_t) * CTR_COUNT, st));
    HIPCHK(hipMemcpyAsync(W.rcp_ctr + CTR_HANDOVERS, &n_handovers, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    // the kernels of the single-GPU world over: the given records, this rank's subscriptions as they were at the tick's start
    WorldDev v = W.d;
    v.handovers = W.rcp_recs; v.handovers_cap = n_handovers; v.counters = W.rcp_ctr;
    v.ho_moved = nullptr; v.n_groups = 0;  // (every handover as its notifier alone: bit 0 of full_mask)
    if (v.wb) v.sub_bits = W.snap_bits ? W.snap_bits : v.sub_bits;
    else if (W.snap_cell) { v.pair_cell = W.snap_cell; v.pair_cnt = W.snap_cnt; }
    launch_handover_recipients_count(st, ctx->g, v, W.rcp_off, W.rcp_own);
    launch_scan_u32_inplace(st, W.rcp_off, n_handovers);
    launch_handover_recipients_fill(st, ctx->g, v, W.rcp_off, W.ho_rcp_conn, W.ho_rcp_kind, W.ho_rcp_mask, W.ho_rcp_cap);
    TRY(after_launch(ctx));
    TRY(down(ctx, offsets, W.rcp_off, sizeof(uint32_t) * ((size_t)n_handovers + 1)));
    std::vector<uint32_t> own(src_owner_unsubscribed ? n_handovers : 0);
    if (src_owner_unsubscribed) TRY(down(ctx, own.data(), W.rcp_own, sizeof(uint32_t) * (size_t)n_handovers));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t h = 0; h < n_handovers && src_owner_unsubscribed; h++) src_owner_unsubscribed[h] = own[h] ? 1 : 0;
    const uint64_t total = offsets[n_handovers];
    *n_out = total;
    if (total > W.ho_rcp_cap) return fail(ctx, CHD_E_CAPACITY, "handover recipients: %llu exceed the engine capacity %llu", (unsigned long long)total, (unsigned long long)W.ho_rcp_cap);
    if (total > cap) return fail(ctx, CHD_E_CAPACITY, "handover recipients: %llu, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
    TRY(down(ctx, conn, W.ho_rcp_conn, sizeof(uint32_t) * total));
    TRY(down(ctx, kind, W.ho_rcp_kind, total));
    if (full_mask) TRY(down(ctx, full_mask, W.ho_rcp_mask, sizeof(uint32_t) * total));
    HIPCHK(hipStreamSynchronize(st));
    return CHD_OK;
}

int chd_adjacent_recipients(chd_ctx *ctx, uint32_t n_req, const uint32_t *channel, const uint32_t *broadcast,
                            const uint32_t *sender_conn, const uint32_t *client_conn, uint32_t *offsets, uint32_t *conns,
                            uint64_t cap) {
    NEED_WORLD();
    if (!offsets) return fail(ctx, CHD_E_INVAL, "chd_adjacent_recipients: NULL offsets");
    offsets[0] = 0;
    if (!n_req) return CHD_OK;
    if (!channel || !broadcast || !sender_conn || !client_conn || (cap && !conns))
        return fail(ctx, CHD_E_INVAL, "chd_adjacent_recipients: NULL buffer");
    for (uint32_t i = 0; i < n_req; i++) {
        if (channel[i] < ctx->g.id_start || channel[i] - ctx->g.id_start >= ctx->g.ncell)
            return fail(ctx, CHD_E_INVAL, "request %u: BroadcastType_ADJACENT_CHANNELS only works for a spatial channel (message.go:190-193)", i);
    }
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    const size_t nb = sizeof(uint32_t) * (size_t)n_req;
    for (int k = 0; k < 4; k++) TRY(ensure(ctx, k, nb));
    TRY(ensure(ctx, 4, nb + sizeof(uint32_t)));
    TRY(ensure(ctx, 5, sizeof(uint32_t) * std::max<uint64_t>(cap, 1)));
    TRY(up(ctx, sbuf<void>(ctx, 0), channel, nb));
    TRY(up(ctx, sbuf<void>(ctx, 1), broadcast, nb));
    TRY(up(ctx, sbuf<void>(ctx, 2), sender_conn, nb));
    TRY(up(ctx, sbuf<void>(ctx, 3), client_conn, nb));
    WorldDev &d = ctx->w.d;
    launch_adjacent_recipients(ctx->stream, ctx->g, d, n_req, sbuf<uint32_t>(ctx, 0), sbuf<uint32_t>(ctx, 1), sbuf<uint32_t>(ctx, 2),
                               sbuf<uint32_t>(ctx, 3), sbuf<uint32_t>(ctx, 4), nullptr, 0, 0);
    launch_scan_u32_inplace(ctx->stream, sbuf<uint32_t>(ctx, 4), n_req);
    launch_adjacent_recipients(ctx->stream, ctx->g, d, n_req, sbuf<uint32_t>(ctx, 0), sbuf<uint32_t>(ctx, 1), sbuf<uint32_t>(ctx, 2),
                               sbuf<uint32_t>(ctx, 3), sbuf<uint32_t>(ctx, 4), sbuf<uint32_t>(ctx, 5), cap, 1);
    TRY(after_launch(ctx));
    TRY(down(ctx, offsets, sbuf<void>(ctx, 4), nb + sizeof(uint32_t)));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t total = offsets[n_req];
    if (total > cap) return fail(ctx, CHD_E_CAPACITY, "chd_adjacent_recipients: %llu recipients, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
    TRY(down(ctx, conns, sbuf<void>(ctx, 5), sizeof(uint32_t) * total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_subs_get(chd_ctx *ctx, uint32_t slot, uint32_t *channel, uint32_t *interval_ms, int64_t *last_fanout_ns,
                 uint8_t *had_first, uint8_t *is_new, uint32_t *n_out) {
    NEED_WORLD();
    if (!n_out) return fail(ctx, CHD_E_INVAL, "chd_subs_get: NULL n_out");
    WorldDev &d = ctx->w.d;
    if (slot >= d.S) return fail(ctx, CHD_E_INVAL, "subscriber slot %u out of range", slot);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    uint32_t cnt = 0, stick = 0;
    TRY(down(ctx, &cnt, d.pair_cnt + slot, 4));
    TRY(down(ctx, &stick, d.sub_tick + slot, 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const bool fresh = (stick == ctx->ring.cur_tick);
    std::vector<uint32_t> fl(cnt), cell(cnt);
    const size_t pb = (size_t)slot * d.capq;
    TRY(down(ctx, cell.data(), d.pair_cell + pb, 4 * (size_t)cnt));
    TRY(down(ctx, fl.data(), d.pair_flags + pb, 4 * (size_t)cnt));
    if (interval_ms) TRY(down(ctx, interval_ms, d.pair_iv + pb, 4 * (size_t)cnt));
    if (last_fanout_ns) TRY(down(ctx, last_fanout_ns, d.pair_last + pb, 8 * (size_t)cnt));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < cnt; i++) {
        if (channel) channel[i] = cell[i] + ctx->g.id_start;
        if (had_first) had_first[i] = (fl[i] & PF_HAD_FIRST) ? 1 : 0;
        if (is_new) is_new[i] = (fresh && (fl[i] & PF_NEW)) ? 1 : 0;
    }
    *n_out = cnt;
    return CHD_OK;
}

int chd_world_get_entities(chd_ctx *ctx, uint32_t n, const uint32_t *idx, uint32_t *cell_channel, uint32_t *member_channel) {
    NEED_WORLD();
    WorldDev &d = ctx->w.d;
    if (!idx && n > d.N) return fail(ctx, CHD_E_INVAL, "n > max_entities");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    std::vector<uint32_t> cell(d.N), mem(d.N);
    TRY(down(ctx, cell.data(), d.cell, 4 * (size_t)d.N));
    TRY(down(ctx, mem.data(), d.member, 4 * (size_t)d.N));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t u = 0; u < n; u++) {
        uint32_t i = idx ? idx[u] : u;
        if (i >= d.N) return fail(ctx, CHD_E_INVAL, "entity slot %u out of range", i);
        if (cell_channel) cell_channel[u] = cell[i] == CHD_INVALID ? 0u : cell[i] + ctx->g.id_start;
        if (member_channel) member_channel[u] = mem[i] == CHD_INVALID ? 0u : mem[i] + ctx->g.id_start;
    }
    return CHD_OK;
}

int chd_dev_alloc(chd_ctx *ctx, uint64_t bytes, void **d_out) {
    if (!ctx || !d_out) return fail(ctx, CHD_E_INVAL, "chd_dev_alloc: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipMalloc(d_out, bytes ? bytes : 256));
    return CHD_OK;
}
int chd_dev_free(chd_ctx *ctx, void *d_ptr) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_ptr));
    return CHD_OK;
}
int chd_dev_upload(chd_ctx *ctx, void *d_dst, const void *src, uint64_t bytes) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(up(ctx, d_dst, src, bytes));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}
int chd_dev_download(chd_ctx *ctx, void *dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    TRY(down(ctx, dst, d_src, bytes));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CHD_OK;
}

int chd_host_alloc(chd_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx || !out) return fail(ctx, CHD_E_INVAL, "chd_host_alloc: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipHostMalloc(out, bytes ? bytes : 256, hipHostMallocDefault));
    return CHD_OK;
}
int chd_host_free(chd_ctx *ctx, void *ptr) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipHostFree(ptr));
    return CHD_OK;
}

int chd_set_profiling(chd_ctx *ctx, int depth) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    if (depth < 0 || depth > TICK_RING) return fail(ctx, CHD_E_INVAL, "profiling depth must be in [0, %d]", TICK_RING);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    ctx->ev.clear();
    ctx->prof_depth = 0;
    if (depth > 0) {
        ctx->ev.resize((size_t)depth * EV_PER_TICK);
        ctx->ev_overlap.assign((size_t)depth, 0);
        for (auto &e : ctx->ev) HIPCHK(hipEventCreate(&e));
        // record every event once so that elapsed-time queries on unused slots are defined
        for (auto &e : ctx->ev) HIPCHK(hipEventRecord(e, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->prof_depth = depth;
    }
    return CHD_OK;
}

int chd_set_profiling_scope(chd_ctx *ctx, int scope) {
    if (!ctx) return fail(nullptr, CHD_E_INVAL, "NULL ctx");
    const int what = scope & 0xFF, every = scope >> 8;
    if ((what != CHD_PROF_STAGES && what != CHD_PROF_RECORD_KERNEL) || every < 0 || every > 1024 || (every && what != CHD_PROF_RECORD_KERNEL))
        return fail(ctx, CHD_E_INVAL, "chd_set_profiling_scope: unknown scope %d", scope);
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->prof_kernel_only = what == CHD_PROF_RECORD_KERNEL;
    ctx->prof_every = every > 1 ? (uint32_t)every : 1u;
    return CHD_OK;
}

int chd_world_set_pipelining(chd_ctx *ctx, int on) {
    NEED_WORLD();
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    if (!ctx->w.pipe_alloc) return fail(ctx, CHD_E_STATE, "chd_world_set_pipelining: the world was not created with CHD_WORLD_PIPELINE_TICKS (or the flag did not take effect)");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux_stream));
    ctx->w.pipe_on = on != 0;
    return CHD_OK;
}

int chd_get_tick_history(chd_ctx *ctx, uint32_t n, chd_tick_stats *out) {
    NEED_WORLD();
    if (!out || n == 0 || n > TICK_RING) return fail(ctx, CHD_E_INVAL, "chd_get_tick_history: bad arguments");
    std::lock_guard<FairMutex> lk(ctx->mu);
    TRY(bind(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> ring((size_t)TICK_RING * 8);
    HIPCHK(hipMemcpy(ring.data(), ctx->w.d.tick_ring, ring.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const uint32_t cur = ctx->ring.cur_tick;
    for (uint32_t k = 0; k < n; k++) {
        chd_tick_stats &s = out[k];
        memset(&s, 0, sizeof s);
        if (k >= cur) continue;  // before the first tick
        const uint32_t tick = cur - k;
        const uint64_t *r = &ring[(size_t)(tick % TICK_RING) * 8];
        s.n_records = r[0];
        s.n_record_upper_bound = r[1];
        s.n_handovers = (uint32_t)r[2];
        s.n_unsubs = (uint32_t)r[4];
        s.n_pairs = (uint32_t)r[6];
        s.n_deferred_records = (uint32_t)(r[6] >> 32);
        s.n_deep_records = (uint32_t)(r[2] >> 32);
        s.n_filtered_records = (uint32_t)(r[3] >> 32);
        s.algorithmic_bytes = 12ull * s.n_records + 32ull * s.n_handovers;
        if (ctx->prof_depth > 0 && k < (uint32_t)ctx->prof_depth) stage_times(ctx, tick, s);
        s.overflow = (uint32_t)r[7];
        s.history_overflow = (uint32_t)(r[7] >> 32);
    }
    TRY(gate_check(ctx));
    for (uint32_t k = 0; k < n; k++) { out[k].schedule = schedule_bits(ctx); out[k].gate_timeouts = ctx->w.gate_timeouts; }
    return CHD_OK;
}

int chd_get_tick_stats(chd_ctx *ctx, chd_tick_stats *out) {
    if (!ctx || !out) return fail(ctx, CHD_E_INVAL, "chd_get_tick_stats: NULL argument");
    std::lock_guard<FairMutex> lk(ctx->mu);
    // byte model of DESIGN.md §4: 12 B per emitted message dominate
    chd_tick_stats s = ctx->stats;
    s.algorithmic_bytes = 12ull * s.n_records + 32ull * s.n_handovers;
    s.schedule = schedule_bits(ctx);
    s.gate_timeouts = ctx->w.gate_timeouts;
    *out = s;
    return CHD_OK;
}

}  // extern "C"
