// placement.hip -- does the random-gather rate over a large table depend on WHERE the table was allocated?
// In-process A/B runs of identical match kernels (tools/ab_libs.py) are bimodal: contexts whose tables were allocated
// at some moments run 10 % faster than others, run after run.  This measures the plain random 16-byte gather over
// several tables of the same size allocated one after another (hipMalloc).  Run it with and without HSA_MAX_VA_ALIGN
// (the ROCm runtime's cap on the alignment of virtual addresses, as an order of 4-KiB pages; default 9 = 2 MiB).
// Build: hipcc -O3 --offload-arch=gfx950 placement.hip -o placement
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_gather(const ulonglong2 *__restrict__ tab, uint64_t nelem, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const ulonglong2 v = tab[(uint64_t)(((unsigned __int128)x * nelem) >> 64)];
        acc += v.x ^ v.y;
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

static uint64_t *g_out;
static double rate(const void *tab, size_t bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8 * 8, iters = 64;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, (const ulonglong2 *)tab, (uint64_t)(bytes / 16), iters, g_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)blocks * 256 * iters / best / 1e6;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 8.6) * (1ull << 30)) & ~((size_t)(2u << 20) - 1);
    const int ntab = argc > 2 ? atoi(argv[2]) : 8;
    const bool contig = argc > 3 && argv[3][0] == 'c';   // physically contiguous tables (hipDeviceMallocContiguous)
    hipMalloc(&g_out, 64);
    printf("table bytes %zu\n", bytes);
    // some unrelated allocations first, as a real process has (text, reads)
    void *pre1, *pre2;
    hipMalloc(&pre1, 800ull << 20); hipMalloc(&pre2, 4ull << 30);
    printf("-- %s, one after another (all kept)\n", contig ? "hipExtMallocWithFlags(contiguous)" : "hipMalloc");
    std::vector<void *> tabs;
    for (int i = 0; i < ntab; i++) {
        void *p = nullptr;
        const hipError_t ae = contig ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) : hipMalloc(&p, bytes);
        if (ae != hipSuccess) { printf("alloc %d failed: %s\n", i, hipGetErrorString(ae)); (void)hipGetLastError(); break; }
        hipMemset(p, 0, bytes);
        tabs.push_back(p);
        // an allocate/free pair in between, as the index build's sort buffers are
        void *tmp; hipMalloc(&tmp, 3ull << 30); hipFree(tmp);
    }
    for (int round = 0; round < 2; round++)
        for (size_t i = 0; i < tabs.size(); i++)
            printf("hipMalloc #%zu va %p (va %% 8GiB = %4llu MiB)  %.2f G gathers/s\n", i, tabs[i],
                   (unsigned long long)(((uintptr_t)tabs[i] & ((1ull << 33) - 1)) >> 20), rate(tabs[i], bytes));
    for (void *p : tabs) hipFree(p);
    fflush(stdout);

    return 0;
}
