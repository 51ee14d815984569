// filter_mall.hip -- would a 1-bit-per-bucket-pair occupancy filter pay in the dual kernel?  
// Half of the probed bucket pairs are empty in both strands; a probe costs a random 128-byte line of the 16 GiB pair table.
// A filter of 2^29 bits is 64 MiB: too big for the L2s (8 x 4 MiB, and every XCD touches all of it), small enough for the
// 256 MiB memory-side cache -- IF the table's own lines, which stream through that cache at ~5 TB/s, leave it there.
// Every lane makes `iters` probes: mode 0 = a 16-byte load at a random line of the table; mode 1 = a 4-byte load at a random
// word of the filter; mode 2 = the filter word first, then the table line for half of the probes (by a hash bit, so that
// the work is known).  Filter sizes 8 .. 256 MiB.  Prints ms and G probes/s; mode 2 against mode 0 is the answer.
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
template <int MODE>
__global__ void __launch_bounds__(256) k_probe(const char *__restrict__ tab, uint64_t lines_mask, const uint32_t *__restrict__ filt, uint64_t fwords_mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t a = (x & lines_mask) << 7;
        bool go = true;
        if (MODE >= 1) {
            const uint32_t w = filt[(x >> 24) & fwords_mask];
            acc += w;
            go = MODE == 2 && (((x >> 60) ^ w) & 1u);     // (w = 0: half of the probes; the dependence on w keeps the order filter -> table)
        }
        if (MODE != 1 && go) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab + a);
            acc += v.x ^ v.y;
        }
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
int main() {
    const size_t bytes = 16ull << 30;
    void *tab; uint64_t *out; uint32_t *filt;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 64)); CK(hipMalloc((void **)&filt, 256ull << 20));
    CK(hipMemset(tab, 0, bytes)); CK(hipMemset(filt, 0, 256ull << 20));
    const uint64_t lines = bytes >> 7;
    const int blocks = 256 * 8 * 4, iters = 64;
    const double probes = (double)blocks * 256.0 * iters;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int mode, uint64_t fbytes) -> float {
        float best = 1e30f;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(e0, 0));
            const uint64_t fm = fbytes / 4 - 1;
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, filt, fm, iters, out);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, filt, fm, iters, out);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, filt, fm, iters, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    printf("probes per launch: %.0f; table 16 GiB\n", probes);
    const float t0 = run(0, 8ull << 20);
    printf("mode 0 (every probe a table line)              %8.3f ms  %6.1f G probes/s\n", t0, probes / t0 / 1e6);
    for (uint64_t mb = 8; mb <= 256; mb *= 2) {
        const float t1 = run(1, mb << 20), t2 = run(2, mb << 20);
        printf("filter %3llu MiB: mode 1 (filter word only) %8.3f ms %6.1f G/s | mode 2 (filter, then the line for half) %8.3f ms %6.1f G probes/s = %.2fx mode 0\n",
               (unsigned long long)mb, t1, probes / t1 / 1e6, t2, probes / t2 / 1e6, t0 / t2);
        fflush(stdout);
    }
    printf("done\n");
    return 0;
}
