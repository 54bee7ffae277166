// Which XCD does workgroup b run on?  Reads HW_REG_XCC_ID per workgroup and compares with b % 8 for a few launch shapes.
// hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip && ./xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned a = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        unsigned b = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((32 - 1) << 11));
        out[2 * blockIdx.x] = a;
        out[2 * blockIdx.x + 1] = b;
    }
}
__global__ void busy(unsigned *out, unsigned spins) {
    unsigned v = threadIdx.x;
    for (unsigned i = 0; i < spins; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) out[blockIdx.x] = v;
}
int main() {
    const unsigned shapes[][3] = {{196, 256, 0}, {2500, 256, 47200}, {10000, 64, 11800}, {2048, 64, 0}};
    for (auto &s : shapes) {
        unsigned nb = s[0], *d;
        hipMalloc(&d, 8 * nb);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(k, dim3(nb), dim3(s[1]), s[2], 0, d);
        std::vector<unsigned> h(2 * nb);
        hipMemcpy(h.data(), d, 8 * nb, hipMemcpyDeviceToHost);
        unsigned same = 0, hist[16] = {0};
        for (unsigned b = 0; b < nb; b++) { same += (h[2 * b] & 7u) == (b & 7u); hist[h[2 * b] & 15u]++; }
        printf("grid %u x %u threads, %u B LDS: xcc(4 bits) == b %% 8 for %u of %u; raw reg of blocks 0..9:", nb, s[1], s[2], same, nb);
        for (unsigned b = 0; b < 10 && b < nb; b++) printf(" %x", h[2 * b + 1]);
        printf("; histogram of the low 4 bits:");
        for (int i = 0; i < 16; i++) printf(" %u", hist[i]);
        printf("\n");
        hipFree(d);
    }
    // the same while another queue keeps the chip busy (does a second dispatch shift the round-robin?)
    {
        hipStream_t a, b;
        hipStreamCreate(&a); hipStreamCreate(&b);
        unsigned nb = 10000, *d, *d2;
        hipMalloc(&d, 8 * nb); hipMalloc(&d2, 8 * 4096);
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(busy, dim3(391 + 97 * rep), dim3(256), 0, a, d2, 40000u);
            hipLaunchKernelGGL(k, dim3(nb), dim3(64), 11800, b, d);
            hipDeviceSynchronize();
            std::vector<unsigned> h(2 * nb);
            hipMemcpy(h.data(), d, 8 * nb, hipMemcpyDeviceToHost);
            unsigned same = 0;
            for (unsigned i = 0; i < nb; i++) same += (h[2 * i] & 7u) == (i & 7u);
            printf("beside a busy kernel of %d workgroups on another stream: xcc == b %% 8 for %u of %u\n", 391 + 97 * rep, same, nb);
        }
    }
    return 0;
}
