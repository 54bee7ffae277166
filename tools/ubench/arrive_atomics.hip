// What does "every workgroup bumps its XCD's arrival counter" cost?  10000 single-wave workgroups, 8 counters on separate lines,
// returning / non-returning, agent / workgroup scope (the latter stays in the XCD's own L2).
// hipcc --offload-arch=gfx950 -O2 -o arrive_atomics arrive_atomics.hip && ./arrive_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long *cnt, unsigned long long *sink, unsigned work) {
    unsigned v = threadIdx.x;
    for (unsigned i = 0; i < work; i++) v = v * 1664525u + 1013904223u;
    if (threadIdx.x == 0) {
        unsigned long long *p = cnt + 16u * (blockIdx.x & 7u);
        unsigned long long r = v == 7u;
        if (MODE == 0) r += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) r += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) r += __hip_atomic_fetch_add(cnt + 16u * (blockIdx.x & 127u), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == 0xFFFFFFFFFFFFull) sink[0] = r;
    }
}
int main() {
    unsigned long long *cnt, *sink;
    (void)hipMalloc(&cnt, 128 * 16 * 8); (void)hipMalloc(&sink, 8);
    (void)hipMemset(cnt, 0, 128 * 16 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const char *names[] = {"no atomic", "returning, agent scope, 8 counters", "returning, workgroup scope, 8 counters", "non-returning, agent scope, 8 counters",
                           "returning, agent scope, 128 counters"};
    for (unsigned work : {0u, 20000u}) {
        for (int m = -1; m < 4; m++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                (void)hipEventRecord(a, 0);
                if (m == -1) hipLaunchKernelGGL(k<9>, dim3(10000), dim3(64), 11800, 0, cnt, sink, work);
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(10000), dim3(64), 11800, 0, cnt, sink, work);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(10000), dim3(64), 11800, 0, cnt, sink, work);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(10000), dim3(64), 11800, 0, cnt, sink, work);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(10000), dim3(64), 11800, 0, cnt, sink, work);
                (void)hipEventRecord(b, 0);
                (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("work %5u, %-44s %8.1f us\n", work, names[m + 1], best * 1000.f);
        }
    }
    unsigned long long h[16];
    (void)hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
    printf("counter 0 = %llu\n", h[0]);
    return 0;
}
