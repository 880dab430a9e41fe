// microbenchmark: n global atomicAdds to pseudo-random addresses of a table of `bins` words (device scope), against the
// same number of plain stores; decides whether the index build may histogram with global atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_atomic(uint32_t *t, uint32_t bins, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    atomicAdd(&t[h % bins], 1u);
}
__global__ void k_store(uint32_t *t, uint32_t bins, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    t[h % bins] = i;
}
int main() {
    const uint32_t n = 1u << 20;
    for (uint32_t bins : {1024u, 131072u, 524288u, 2097152u}) {
        uint32_t *t; hipMalloc(&t, bins * 4); hipMemset(t, 0, bins * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int which = 0; which < 2; which++) {
            float best = 1e9;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(a);
                if (which == 0) hipLaunchKernelGGL(k_atomic, dim3(n / 256), dim3(256), 0, 0, t, bins, n);
                else hipLaunchKernelGGL(k_store, dim3(n / 256), dim3(256), 0, 0, t, bins, n);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("%s bins %8u: %.1f us for %u ops\n", which == 0 ? "atomicAdd" : "store    ", bins, best * 1e3, n);
        }
        hipFree(t);
    }
    return 0;
}
