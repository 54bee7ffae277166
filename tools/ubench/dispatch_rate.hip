// How long does the chip take to START a grid?  Empty kernels (one store behind a never-true test), grids of 32 .. 3600 workgroups of
// 64 / 256 / 1024 threads, with and without 16 KB of LDS and with a large by-value argument (the tick's kernels take the ~1.3 KB
// WorldDev struct), back to back on one stream, microseconds per launch.
// hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned long long p[160]; };
template <int LDS, int SALT>
__global__ void k(unsigned *out, unsigned seed) {
    __shared__ unsigned sh[LDS ? LDS / 4 : 1];
    if (LDS) sh[threadIdx.x] = seed;
    if (threadIdx.x + seed + SALT == 0x7FFFFFFFu) out[0] = sh[0];
}
template <int SALT>
__global__ void kbig(Big b, unsigned seed) {
    if (threadIdx.x + seed + SALT == 0x7FFFFFFFu) ((unsigned *)b.p[3])[0] = (unsigned)b.p[100];
}
template <int LDS>
float run(unsigned *out, int grid, int block) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((k<LDS, 0>), dim3(grid), dim3(block), 0, 0, out, 3u);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 200; r++) { hipLaunchKernelGGL((k<LDS, 0>), dim3(grid), dim3(block), 0, 0, out, 3u); hipLaunchKernelGGL((k<LDS, 1>), dim3(grid), dim3(block), 0, 0, out, 3u); }
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 400.f;
}
float runbig(unsigned *out, int grid, int block) {
    Big bb; for (int i = 0; i < 160; i++) bb.p[i] = (unsigned long long)out;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((kbig<0>), dim3(grid), dim3(block), 0, 0, bb, 3u);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 200; r++) { hipLaunchKernelGGL((kbig<0>), dim3(grid), dim3(block), 0, 0, bb, 3u); hipLaunchKernelGGL((kbig<1>), dim3(grid), dim3(block), 0, 0, bb, 3u); }
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 400.f;
}
int main() {
    unsigned *out; (void)hipMalloc(&out, 64);
    printf("us per launch; columns: no LDS | 16 KB LDS | 1.3 KB by-value argument\n");
    for (int block : {64, 256, 1024})
        for (int grid : {1, 32, 64, 128, 225, 450, 900, 1800, 3600, 10000})
            printf("block %4d grid %5d (%6d waves): %6.2f | %6.2f | %6.2f\n", block, grid, grid * block / 64, run<0>(out, grid, block), block <= 1024 ? run<16384>(out, grid, block) : 0.f, runbig(out, grid, block));
    return 0;
}
