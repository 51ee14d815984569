// gather4.hip -- do several 16-B loads of ONE lane into the same 64-B line cost one random request or several?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <int W> // W consecutive 16-B loads from one random 64-B-aligned line
__global__ void __launch_bounds__(256) k(const ulonglong2 *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t line = (x & mask) & ~3ull;
        ulonglong2 v[W];
#pragma unroll
        for (int k = 0; k < W; k++) v[k] = tab[line + k];
#pragma unroll
        for (int k = 0; k < W; k++) acc += v[k].x ^ v[k].y;
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <int W> static void run(const void *tab, size_t bytes, uint64_t *out) {
    const int blocks = 256 * 8 * 4, iters = 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, (const ulonglong2 *)tab, bytes / 16 - 1, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%d x 16 B per random line: %7.2f G lines/s\n", W, (double)blocks * 256 * iters / best / 1e6); fflush(stdout);
}
int main() {
    const size_t bytes = 8ull << 30;
    void *tab; uint64_t *out; CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, bytes));
    run<1>(tab, bytes, out); run<2>(tab, bytes, out); run<3>(tab, bytes, out); run<4>(tab, bytes, out);
    return 0;
}
