// What does the hardware allow for the emit kernel's store stream as a SHORT kernel?  10 000 single-wave workgroups
// (one per connection), each writing `nseg` line-aligned 448-record segments (1 KiB contiguous per store instruction)
// inside its own region of the record buffer, with holes between the segments like the worst-case slots of
// k_fanout_plan — 672 MB per launch, timed launch by launch (ramp and tail included, as in the tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k_conn(u32x4 *out, uint32_t nseg, uint32_t seg, uint32_t pitch, uint32_t jitter) {
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    // connections differ in size: nseg +- jitter (deterministic hash)
    const uint32_t mine = nseg - jitter + (uint32_t)(((uint64_t)s * 2654435761u >> 7) % (2 * jitter + 1));
    u32x4 *base = out + (size_t)s * (nseg + jitter) * (pitch / 2);
    for (uint32_t k = 0; k < mine; k++) {
        u32x4 *p = base + (size_t)k * (pitch / 2);
        for (uint32_t q = lane; q < seg / 2; q += 64) { u32x4 r = {s, q, s, k}; p[q] = r; }
    }
}
int main() {
    const uint32_t S = 10000, seg = 448;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (uint32_t nseg : {18u, 19u})
    for (uint32_t pitch : {448u, 896u, 1344u})
    for (uint32_t jitter : {0u, 6u}) {
        const size_t bytes = (size_t)S * (nseg + jitter) * pitch * 8;
        void *buf; (void)hipMalloc(&buf, bytes + 4096);
        std::vector<float> t;
        for (int r = 0; r < 12; r++) {
            (void)hipEventRecord(a);
            k_conn<<<S, 64>>>((u32x4 *)buf, nseg, seg, pitch, jitter);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (r >= 2) t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        const double mb = (double)S * nseg * seg * 8 / 1e6;
        printf("%u conns x %u(+-%u) segments of %u records, segment pitch %4u: %6.1f MB per launch, median %6.1f us (min %6.1f) -> %5.0f GB/s\n", S, nseg, jitter, seg, pitch, mb,
               t[t.size() / 2], t[0], mb / t[t.size() / 2] * 1e3 / 1e3);
        (void)hipFree(buf);
    }
    return 0;
}
