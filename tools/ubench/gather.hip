// gather.hip -- microbenchmark: chip-wide rate of independent random 8-byte gathers as a function of
// the table footprint (TLB reach / HBM random-access rate).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// each lane performs `iters` gathers; DEP=1: address of the next gather depends on the loaded value
template <int DEP>
__global__ void __launch_bounds__(256) k_gather(const uint64_t *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        uint64_t v = tab[x & mask];
        acc += v;
        x = mix64(x + (DEP ? v : 0) + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = 64;
    const size_t max_bytes = (argc > 1 ? atoll(argv[1]) : 32ull) << 30;
    uint64_t *tab, *out;
    if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipMemset(tab, 0, max_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8 * 8; // 8 waves/SIMD-ish resident, 8 rounds
    printf("footprint_MiB dep gathers_per_s(G) GBps_64B_sectors\n");
    for (size_t bytes = 16ull << 20; bytes <= max_bytes; bytes <<= 1) {
        for (int dep = 0; dep < 2; dep++) {
            uint64_t mask = bytes / 8 - 1;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (dep) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
                else hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    double n = (double)blocks * 256 * iters;
                    printf("%8zu %d %8.2f %8.1f\n", bytes >> 20, dep, n / ms / 1e6, n * 64 / ms / 1e6);
                }
            }
        }
        fflush(stdout);
    }
    return 0;
}
