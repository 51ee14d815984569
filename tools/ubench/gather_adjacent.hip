// gather_adjacent.hip -- what does a SECOND 16-byte load cost next to a random one?  (round 4: a verified text window that straddles
// two 128-byte lines turned out far cheaper than two lines somewhere: profiles/r04_text_twin_ab.txt)
// Every lane loads 16 bytes at a random 128-byte-aligned address of a 16 GiB table, and a second 16 bytes at a distance D from it:
//   D = 0 (none), 16 / 64 (the same 128-byte line), 128 (the adjacent line), 256, 4096, 2 MiB, random (another random line).
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
// mode 0: one load; 1: second load at +dist bytes; 2: second load at another random line
__global__ void __launch_bounds__(256) k(const char *__restrict__ tab, uint64_t lines_mask, int mode, uint64_t dist, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t a = (x & lines_mask) << 7;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab + a);
        acc += v.x ^ v.y;
        if (mode == 1) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(tab + a + dist);
            acc += w.x ^ w.y;
        } else if (mode == 2) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(tab + ((mix64(x ^ 0x5555) & lines_mask) << 7));
            acc += w.x ^ w.y;
        }
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
static void run(const char *name, const void *tab, uint64_t lines, int mode, uint64_t dist, uint64_t *out) {
    const int blocks = 256 * 8 * 4, iters = 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, mode, dist, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %7.2f G lanes/s\n", name, (double)blocks * 256 * iters / best / 1e6); fflush(stdout);
}
int main() {
    const size_t bytes = 16ull << 30;
    void *tab; uint64_t *out; CK(hipMalloc(&tab, bytes + (4ull << 20))); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, bytes + (4ull << 20)));
    const uint64_t lines = bytes >> 7;
    run("one 16-B load per lane", tab, lines, 0, 0, out);
    run("+ 16 B at +16 (same 64-B half)", tab, lines, 1, 16, out);
    run("+ 16 B at +64 (same 128-B line)", tab, lines, 1, 64, out);
    run("+ 16 B at +128 (adjacent line)", tab, lines, 1, 128, out);
    run("+ 16 B at +256", tab, lines, 1, 256, out);
    run("+ 16 B at +1024", tab, lines, 1, 1024, out);
    run("+ 16 B at +4096", tab, lines, 1, 4096, out);
    run("+ 16 B at +65536", tab, lines, 1, 65536, out);
    run("+ 16 B at +2 MiB", tab, lines, 1, 2ull << 20, out);
    run("+ 16 B at another random line", tab, lines, 2, 0, out);
    return 0;
}
