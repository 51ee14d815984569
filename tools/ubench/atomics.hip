// atomics.hip -- rate of random 64-bit atomicMin (no return) vs footprint; does L2 residency help?
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
// window = footprint each BLOCK works in (blocks with equal blockIdx%8 share an XCD under round-robin placement)
__global__ void __launch_bounds__(256) k(unsigned long long *tab, uint64_t total_mask, uint64_t win_mask, int iters, int windowed) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    const uint64_t base = windowed ? (mix64(blockIdx.x / 64) & total_mask & ~win_mask) : 0; // 64 consecutive-ish blocks share a window
    for (int i = 0; i < iters; i++) {
        const uint64_t idx = windowed ? (base | (x & win_mask)) : (x & total_mask);
        atomicMin(&tab[idx], (unsigned long long)x);
        x = mix64(x + i);
    }
}
int main() {
    const size_t bytes = 8ull << 30;
    unsigned long long *tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0xFF, bytes));
    const int blocks = 256 * 8 * 4, iters = 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("mode footprint_or_window_MiB  G_atomics_per_s\n");
    for (int windowed = 0; windowed < 2; windowed++)
        for (size_t fp = 1ull << 20; fp <= bytes; fp <<= (windowed ? 1 : 2)) {
            if (windowed && fp > (64ull << 20)) break;
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, windowed ? bytes / 8 - 1 : fp / 8 - 1, fp / 8 - 1, iters, windowed);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%s %8zu %8.2f\n", windowed ? "window" : "global", fp >> 20, (double)blocks * 256 * iters / best / 1e6); fflush(stdout);
        }
    return 0;
}
