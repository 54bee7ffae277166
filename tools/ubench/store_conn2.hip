// Bisecting the gap between the store-only pattern (~125 us for 645 MB) and the emit kernels (~175 us): the same
// 10 000 x 18 segment pattern with the real kernel's features added one at a time.
//   SHIFT : every segment opens with ONE 8-byte record written by lane 0, the 1 KiB runs follow 8 bytes off the lines
//   TWICE : the column is written twice per segment (two fan-out windows), second copy again behind a single record
//   LOAD  : the column (448 x 4 B from a 400 KB L2-resident table) is loaded before the segment's stores and waited for
//   PAD   : the last line of the segment is completed with a partial store (pad_segment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct rec { uint32_t a, b; };
template <bool SHIFT, bool TWICE, bool LOAD, bool PAD, int BATCH>
__global__ void __launch_bounds__(64) k_conn(rec *out, const uint32_t *table, uint32_t ncells, uint32_t nseg, uint32_t seg, uint32_t pitch) {
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    rec *base = out + (size_t)s * nseg * pitch;
    for (uint32_t k0 = 0; k0 < nseg; k0 += BATCH) {
        u32x2 col[BATCH][4];
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const uint32_t cell = (s * 31u + (k0 + b) * 7u) % ncells;
            if (LOAD) {
                const uint32_t *pa = table + cell * 448u + 2 * lane;
#pragma unroll
                for (int h = 0; h < 4; h++) col[b][h] = *(const u32x2 *)(pa + 128 * h);
            } else {
#pragma unroll
                for (int h = 0; h < 4; h++) col[b][h] = (u32x2){cell, lane + h};
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            if (k0 + b >= nseg) break;
            rec *o = base + (size_t)(k0 + b) * pitch;
            uint32_t n_out = 0;
            for (int rep = 0; rep < (TWICE ? 2 : 1); rep++) {
                if (SHIFT) { if (lane == 0) o[n_out] = rec{s, 0xFFFFFFFFu}; n_out += 1; }
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t k = 128u * h + 2 * lane;
                    if (k + 1 < seg) { u32x4 r = {s, col[b][h].x, s, col[b][h].y}; *(u32x4 *)(void *)(o + n_out + k) = r; }
                    else if (k < seg) o[n_out + k] = rec{s, col[b][h].x};
                }
                n_out += seg;
            }
            if (PAD) { const uint32_t pad = (0u - n_out) & 15u; if (lane < pad) o[n_out + lane] = rec{0xFFFFFFFFu, 0}; }
        }
    }
}
template <bool SHIFT, bool TWICE, bool LOAD, bool PAD, int BATCH>
void run(const char *name, rec *buf, const uint32_t *table, uint32_t nseg, uint32_t seg, uint32_t pitch) {
    const uint32_t S = 10000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < 12; r++) {
        (void)hipEventRecord(a);
        k_conn<SHIFT, TWICE, LOAD, PAD, BATCH><<<S, 64>>>(buf, table, 225, nseg, seg, pitch);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const double mb = (double)S * nseg * (seg + (SHIFT ? 1 : 0)) * (TWICE ? 2 : 1) * 8 / 1e6;
    printf("%-64s %6.1f MB  median %6.1f us (min %6.1f) -> %5.2f TB/s\n", name, mb, t[t.size() / 2], t[0], mb / t[t.size() / 2] / 1e3 * 1e3 / 1e3);
}
int main() {
    rec *buf; uint32_t *table;
    (void)hipMalloc(&buf, (size_t)10000 * 18 * 1344 * 8 + 4096);
    (void)hipMalloc(&table, 225 * 448 * 4 + 4096);
    (void)hipMemset(table, 1, 225 * 448 * 4);
    run<false, false, false, false, 1>("plain: 18 x 445-record segments, aligned", buf, table, 18, 445, 1344);
    run<true, false, false, false, 1>("+ SHIFT (leading single record, runs 8 B off the lines)", buf, table, 18, 445, 1344);
    run<true, false, false, true, 1>("+ SHIFT + PAD", buf, table, 18, 445, 1344);
    run<false, false, true, false, 1>("+ LOAD (column from L2 before every segment)", buf, table, 18, 445, 1344);
    run<true, false, true, true, 1>("+ SHIFT + PAD + LOAD", buf, table, 18, 445, 1344);
    run<true, false, true, true, 2>("+ SHIFT + PAD + LOAD, 2 segments per wait", buf, table, 18, 445, 1344);
    run<true, false, true, true, 4>("+ SHIFT + PAD + LOAD, 4 segments per wait", buf, table, 18, 445, 1344);
    run<false, true, false, false, 1>("TWICE: 9 x (2 x 445) aligned copies", buf, table, 9, 445, 1344);
    run<true, true, false, true, 1>("TWICE + SHIFT + PAD: 9 segments", buf, table, 9, 445, 1344);
    run<true, true, true, true, 1>("TWICE + SHIFT + PAD + LOAD: 9 segments", buf, table, 9, 445, 1344);
    return 0;
}
