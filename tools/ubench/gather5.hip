// gather5.hip -- what limits random 16-byte gathers over a large table: address translation or the memory system?
// Every wave-level load picks ONE random window of W bytes of a 16 GiB table and its 64 lanes gather at random inside
// it.  W = the whole table is the plain random gather (the copMEM probe pattern); W = 2 MiB / 64 KiB needs one
// translation per wave-level load instead of up to 64.  If translation (UTCL1/UTCL2 misses) were the limit, small
// windows would run much faster than the plain gather.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// tab: ulonglong2 elements (16 B); nelem_mask / win_mask in elements
__global__ void __launch_bounds__(256)
k_gather_win(const ulonglong2 *__restrict__ tab, uint64_t nelem_mask, uint64_t win_mask, int iters, uint64_t *out) {
    const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t base = mix64(wave * 1000003ull + i) & nelem_mask & ~win_mask;   // wave-uniform window
        const ulonglong2 v = tab[base | (x & win_mask)];
        acc += v.x ^ v.y;
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = 64;
    const size_t bytes = (argc > 1 ? atoll(argv[1]) : 16ull) << 30;
    ulonglong2 *tab;
    uint64_t *out;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipMemset(tab, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8 * 8;
    const uint64_t nelem_mask = bytes / 16 - 1;
    printf("table_GiB window_bytes gathers_per_s(G)\n");
    for (int wb = 12; wb <= 34; wb += (wb < 24 ? 2 : 3)) {
        uint64_t wbytes = 1ull << wb;
        if (wbytes > bytes) wbytes = bytes;
        const uint64_t win_mask = wbytes / 16 - 1;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_gather_win, dim3(blocks), dim3(256), 0, 0, tab, nelem_mask, win_mask, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) printf("%6zu %12llu %8.2f\n", bytes >> 30, (unsigned long long)wbytes, (double)blocks * 256 * iters / ms / 1e6);
        }
        fflush(stdout);
        if (wbytes == bytes) break;
    }
    return 0;
}
