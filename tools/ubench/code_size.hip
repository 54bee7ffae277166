// What does COLD CODE cost?  A kernel of N straight-line VALU instructions (executed once per wave, no loop), 256 workgroups of 256
// threads — one per CU, the shape of the tick's small latency-bound kernels — launched in a chain with a different kernel of the
// same size in between (the instruction cache does not keep it), timed with events over 200 launches.  If instruction fetch were
// free, 4 K instructions would cost 4 K x 4 cycles = 7 us per wave of issue time at most; what is measured beyond that is fetch.
// hipcc --offload-arch=gfx950 -O3 -o code_size code_size.hip && ./code_size
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, int SALT>
__global__ void __launch_bounds__(256) k(unsigned *out, unsigned seed) {
    unsigned v = threadIdx.x + seed + SALT;
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v) : "v"(seed));
    if (v == 0x12345u) out[0] = v;
}
template <int N>
float run(unsigned *out, int grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; w++) { hipLaunchKernelGGL((k<N, 0>), dim3(grid), dim3(256), 0, 0, out, 3u); hipLaunchKernelGGL((k<N, 1>), dim3(grid), dim3(256), 0, 0, out, 3u); }
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 100; r++) {
        hipLaunchKernelGGL((k<N, 0>), dim3(grid), dim3(256), 0, 0, out, 3u);
        hipLaunchKernelGGL((k<N, 1>), dim3(grid), dim3(256), 0, 0, out, 3u);
        hipLaunchKernelGGL((k<N, 2>), dim3(grid), dim3(256), 0, 0, out, 3u);
        hipLaunchKernelGGL((k<N, 3>), dim3(grid), dim3(256), 0, 0, out, 3u);
    }
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 400.f;
}
int main() {
    unsigned *out; (void)hipMalloc(&out, 64);
    for (int grid : {256, 1}) {
        printf("grid %d x 256 threads, us per launch (back to back on one stream, four distinct kernels of each size in rotation):\n", grid);
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 16, 16 * 8, run<16>(out, grid));
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 256, 256 * 8, run<256>(out, grid));
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 1024, 1024 * 8, run<1024>(out, grid));
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 2048, 2048 * 8, run<2048>(out, grid));
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 4096, 4096 * 8, run<4096>(out, grid));
        printf("  %5d instructions (%6d bytes of code): %.2f us\n", 8192, 8192 * 8, run<8192>(out, grid));
    }
    return 0;
}
