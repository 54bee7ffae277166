// How should a wave lay out 16-byte (two-record) stores?  The emit kernels' copy path gives every lane FOUR adjacent
// records (one 16-byte load of four channel ids -> two 16-byte stores): each store instruction then writes 16-byte pieces at
// a 32-byte stride (half of every 128-byte line), the second instruction fills the other halves.  The alternative gives
// every lane TWO adjacent records per instruction, so one instruction writes 1 KiB contiguously (8 whole lines).
// Same bytes, same segments (448-record line-aligned segments at scattered positions), only the lane->address map differs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(64) k_seg(u32x4 *out, uint32_t nseg, uint32_t seg, uint32_t pitch) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t sgm = blockIdx.x; sgm < nseg; sgm += gridDim.x) {
        const uint32_t id = (uint32_t)(((uint64_t)sgm * 2654435761u) % nseg);
        u32x4 *p = out + (size_t)id * (pitch / 2);  // in units of two records
        for (uint32_t b = 0; b < seg; b += 256) {  // 256 records per step = 128 pairs
            u32x4 r0 = {id, b + lane, id, b}, r1 = {id, lane, id, b + 1};
            if (MODE == 0) {  // lane owns records 4l..4l+3: pair indices 2l, 2l+1
                const uint32_t q = b / 2 + 2 * lane;
                if (2 * q + 3 < seg) { p[q] = r0; p[q + 1] = r1; }
            } else {          // instruction 0 writes pairs l, instruction 1 pairs 64 + l
                const uint32_t q = b / 2 + lane;
                if (2 * q + 1 < seg) p[q] = r0;
                if (2 * (q + 64) + 1 < seg) p[q + 64] = r1;
            }
        }
    }
}
int main() {
    const size_t bytes = 2ull << 30;
    void *buf; hipMalloc(&buf, bytes + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t seg : {448u, 512u, 4096u}) {
        const uint32_t pitch = seg, nseg = (uint32_t)(bytes / 8 / pitch);
        for (int mode = 0; mode < 2; mode++)
            for (int grid : {4096, 10000, 32768}) {
                auto launch = [&] { if (mode == 0) k_seg<0><<<grid, 64>>>((u32x4 *)buf, nseg, seg, pitch); else k_seg<1><<<grid, 64>>>((u32x4 *)buf, nseg, seg, pitch); };
                launch();
                hipEventRecord(a);
                for (int r = 0; r < 5; r++) launch();
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("%4u-rec segments, %s, grid %5d x 1 wave: %6.0f GB/s\n", seg, mode == 0 ? "4 records per lane (16 B at a 32-B stride per store)" : "2 records per lane (1 KiB contiguous per store)  ", grid,
                       (double)nseg * seg * 8 * 5 / (ms * 1e-3) / 1e9);
            }
    }
    return 0;
}
