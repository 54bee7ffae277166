// What delays the START of a workgroup?  225 workgroups of 256 threads (the tick's one-workgroup-per-cell kernels), each does ONE dependent
// global round trip and leaves.  Varied: VGPRs per wave (a live-range of NV registers across the load), static LDS, how many distinct
// fields of a 1.3 KB by-value argument the kernel reads.  us per launch, back to back on one stream.
// hipcc --offload-arch=gfx950 -O3 -o wg_start wg_start.hip && ./wg_start
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { const unsigned *p[160]; };
template <int NV, int LDS, int NF, int SALT>
__global__ void __launch_bounds__(256) k(Big b, unsigned seed) {
    __shared__ unsigned sh[LDS ? LDS / 4 : 1];
    unsigned v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = threadIdx.x * (i + 1) + seed;
    unsigned acc = 0;
#pragma unroll
    for (int f = 0; f < NF; f++) acc += b.p[(f * 7) % 160][blockIdx.x * 256 + threadIdx.x];   // NF distinct pointer fields, one round trip
    if (LDS) sh[threadIdx.x] = acc;
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(acc));
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < NV; i++) s ^= v[i];
    if (s + SALT == 0x7FFFFFFFu) ((unsigned *)b.p[0])[0] = s + (LDS ? sh[0] : 0);
}
template <int NV, int LDS, int NF>
float run(Big bb, int grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((k<NV, LDS, NF, 0>), dim3(grid), dim3(256), 0, 0, bb, 3u);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 200; r++) { hipLaunchKernelGGL((k<NV, LDS, NF, 0>), dim3(grid), dim3(256), 0, 0, bb, 3u); hipLaunchKernelGGL((k<NV, LDS, NF, 1>), dim3(grid), dim3(256), 0, 0, bb, 3u); }
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 400.f;
}
int main() {
    unsigned *buf; (void)hipMalloc(&buf, 64 << 20);
    Big bb; for (int i = 0; i < 160; i++) bb.p[i] = buf + (size_t)i * 100000;
    printf("225 x 256 threads, one dependent load round trip, us per launch\n");
    printf("VGPR live  8, no LDS,  1 field : %.2f\n", run<8, 0, 1>(bb, 225));
    printf("VGPR live  8, no LDS, 13 fields: %.2f\n", run<8, 0, 13>(bb, 225));
    printf("VGPR live 64, no LDS,  1 field : %.2f\n", run<64, 0, 1>(bb, 225));
    printf("VGPR live 64, no LDS, 13 fields: %.2f\n", run<64, 0, 13>(bb, 225));
    printf("VGPR live  8, 4 KB LDS, 1 field : %.2f\n", run<8, 4096, 1>(bb, 225));
    printf("VGPR live 64, 4 KB LDS, 13 fields: %.2f\n", run<64, 4096, 13>(bb, 225));
    printf("VGPR live 64, 32 KB LDS, 13 fields: %.2f\n", run<64, 32768, 13>(bb, 225));
    printf("same, grid 900: %.2f   grid 57: %.2f\n", run<64, 4096, 13>(bb, 900), run<64, 4096, 13>(bb, 57));
    return 0;
}
