// Write-stream rate for the record stream's ACCESS PATTERN: each wave writes whole segments of
// `seg` records (8 B each) with 64-lane, 512-byte contiguous stores; segments start at arbitrary
// 8-byte alignment and are visited in a scattered order (like (cell, connection) segments).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_seg(v2u *out, uint32_t nseg, uint32_t seg, uint32_t pitch, uint32_t shift, int scatter) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    for (uint32_t sgm = wave; sgm < nseg; sgm += nw) {
        uint32_t id = scatter ? (uint32_t)(((uint64_t)sgm * 2654435761u) % nseg) : sgm;
        v2u *p = out + (size_t)id * pitch + shift;
        for (uint32_t b = 0; b < seg; b += 64)
            if (b + lane < seg) p[b + lane] = (v2u)(id, b + lane);
    }
}
int main() {
    const size_t bytes = 2ull << 30;
    void *buf; hipMalloc(&buf, bytes + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct { const char *name; uint32_t seg, pitch, shift; int scatter; } cases[] = {
        {"aligned 512-rec segments, linear", 512, 512, 0, 0},
        {"aligned 512-rec segments, scattered", 512, 512, 0, 1},
        {"445-rec segments at 446 pitch (8-B alignment), linear", 445, 446, 0, 0},
        {"445-rec segments at 446 pitch, scattered", 445, 446, 0, 1},
        {"445-rec segments at 448 pitch (aligned starts), scattered", 445, 448, 0, 1},
        {"512-rec segments shifted by 1 record, scattered", 512, 512, 1, 1},
        {"448-rec segments (445 + pad to the line) at 448 pitch, scattered", 448, 448, 0, 1},
        {"448-rec segments at 448 pitch, linear", 448, 448, 0, 0},
        {"64-rec segments (one store) at 64 pitch, scattered", 64, 64, 0, 1},
        {"4096-rec segments at 4096 pitch, scattered", 4096, 4096, 0, 1},
        {"448-rec segments at 1120 pitch (holes, like worst-case slots), scattered", 448, 1120, 0, 1},
        {"448-rec segments at 1120 pitch, linear", 448, 1120, 0, 0},
    };
    for (auto &c : cases) {
        uint32_t nseg = (uint32_t)(bytes / 8 / c.pitch);
        for (int grid : {1024, 8192}) {
            k_seg<<<grid, 256>>>((v2u *)buf, nseg, c.seg, c.pitch, c.shift, c.scatter);
            hipEventRecord(a);
            for (int r = 0; r < 5; r++) k_seg<<<grid, 256>>>((v2u *)buf, nseg, c.seg, c.pitch, c.shift, c.scatter);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-62s grid %5d: %6.0f GB/s\n", c.name, grid, (double)nseg * c.seg * 8 * 5 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
