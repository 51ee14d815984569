// gather_modes.hip -- does any cache policy or kind of allocation make a random 16-byte gather cost less than a 128-byte line?
// (round 5: the dual kernel is bound by the LINES it requests -- 2.3 G per launch at ~37 G/s against a measured ceiling of 48 G/s
// = 6.1 TB/s of 128-byte lines -- while it uses 16-32 bytes of each; the counters list TCC_EA0_RD_UNCACHED_32B, i.e. uncached
// reads can leave the L2 as 32-byte requests.)  Every lane loads 16 bytes at random 128-byte-aligned addresses of a 16 GiB
// table; each variant is a kernel of its own name: rate by HIP events here, request counts under rocprofv3 --pmc.
//   k_plain      global_load_dwordx4
//   k_sc0 / k_sc1 / k_sc01 / k_nt / k_sc01nt   the same load with those cache-policy bits
//   the three tables: hipMalloc; hipExtMallocWithFlags(hipDeviceMallocUncached); (…Finegrained)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define LOADER(NAME, BITS)                                                                                            \
    __device__ __forceinline__ v4u NAME(const char *p) {                                                              \
        v4u v;                                                                                                        \
        __asm__ volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory");                      \
        return v;                                                                                                     \
    }
LOADER(ld_plain, "")
LOADER(ld_sc0, "sc0")
LOADER(ld_sc1, "sc1")
LOADER(ld_sc01, "sc0 sc1")
LOADER(ld_nt, "nt")
LOADER(ld_sc01nt, "sc0 sc1 nt")
#define KERNEL(NAME, LD)                                                                                              \
    __global__ void __launch_bounds__(256) NAME(const char *__restrict__ tab, uint64_t lines_mask, int iters, uint64_t *out) { \
        uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);                                                   \
        unsigned acc = 0;                                                                                             \
        for (int i = 0; i < iters; i += 4) {                                                                          \
            const uint64_t x1 = mix64(x + 1), x2 = mix64(x + 2), x3 = mix64(x + 3);                                   \
            v4u a = LD(tab + ((x & lines_mask) << 7)), b = LD(tab + ((x1 & lines_mask) << 7));                        \
            v4u c = LD(tab + ((x2 & lines_mask) << 7)), d = LD(tab + ((x3 & lines_mask) << 7));                       \
            __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");                    \
            acc += a.x ^ a.w ^ b.x ^ b.w ^ c.x ^ c.w ^ d.x ^ d.w;                                                     \
            x = mix64(x3 + i);                                                                                        \
        }                                                                                                             \
        if (acc == 0x12345678u) out[0] = acc;                                                                         \
    }
KERNEL(k_plain, ld_plain)
KERNEL(k_sc0, ld_sc0)
KERNEL(k_sc1, ld_sc1)
KERNEL(k_sc01, ld_sc01)
KERNEL(k_nt, ld_nt)
KERNEL(k_sc01nt, ld_sc01nt)
typedef void (*kern_t)(const char *, uint64_t, int, uint64_t *);
int main(int argc, char **argv) {
    const size_t bytes = 16ull << 30;
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    uint64_t *out; CK(hipMalloc(&out, 64));
    const uint64_t lines = bytes >> 7;
    const int blocks = 256 * 8 * 4, iters = 64;
    const double loads = (double)blocks * 256.0 * iters;
    printf("loads per launch: %.0f (16 B each at a random 128-B-aligned address of a %zu-byte table)\n", loads, bytes);
    const char *tn[3] = {"hipMalloc", "uncached", "finegrained"};
    struct { const char *name; kern_t k; } ks[6] = {{"k_plain", k_plain}, {"k_sc0", k_sc0}, {"k_sc1", k_sc1}, {"k_sc01", k_sc01}, {"k_nt", k_nt}, {"k_sc01nt", k_sc01nt}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int t = 0; t < 3; t++) {
        void *tab = nullptr;
        hipError_t he = t == 0 ? hipMalloc(&tab, bytes) : hipExtMallocWithFlags(&tab, bytes, t == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (he != hipSuccess) { printf("table %s: %s\n", tn[t], hipGetErrorString(he)); (void)hipGetLastError(); continue; }
        CK(hipMemset(tab, 0, bytes));
        CK(hipDeviceSynchronize());
        for (int k = 0; k < 6; k++) {
            float best = 1e30f;
            for (int r = 0; r < reps; r++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(ks[k].k, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("table %-12s %-9s %8.3f ms  %6.1f G loads/s\n", tn[t], ks[k].name, best, loads / best / 1e6);
            fflush(stdout);
        }
        CK(hipFree(tab));
    }
    printf("done\n");
    return 0;
}
