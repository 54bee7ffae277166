// Write-stream ceiling on MI355X for the record stream of k_fanout_emit:
// every lane stores 8 B (or 16 B) contiguous, plain vs nontemporal, 1 GiB per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <typename T, bool NT>
__global__ void __launch_bounds__(256) k_store(T *out, size_t n, T v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &out[i]);
        else out[i] = v;
    }
}
template <typename T, bool NT>
float run(void *buf, size_t bytes, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    T v = (T)(1u);
    size_t n = bytes / sizeof(T);
    k_store<T, NT><<<grid, 256>>>((T *)buf, n, v);
    hipEventRecord(a);
    for (int r = 0; r < 10; r++) k_store<T, NT><<<grid, 256>>>((T *)buf, n, v);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return bytes * 10.0 / (ms * 1e-3) / 1e9;
}
__global__ void __launch_bounds__(256) k_copy(const uint4 *in, uint4 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = in[i];
}
int main() {
    size_t bytes = 1ull << 30;
    void *buf, *buf2; hipMalloc(&buf, bytes); hipMalloc(&buf2, bytes);
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d: 8B plain %.0f GB/s, 8B nt %.0f, 16B plain %.0f, 16B nt %.0f\n", grid,
               run<v2u, false>(buf, bytes, grid), run<v2u, true>(buf, bytes, grid),
               run<v4u, false>(buf, bytes, grid), run<v4u, true>(buf, bytes, grid));
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_copy<<<8192, 256>>>((uint4 *)buf, (uint4 *)buf2, bytes / 16);
    hipEventRecord(a);
    for (int r = 0; r < 10; r++) k_copy<<<8192, 256>>>((uint4 *)buf, (uint4 *)buf2, bytes / 16);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("copy 16B: %.0f GB/s read+write\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    return 0;
}
