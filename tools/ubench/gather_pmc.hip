// gather_pmc.hip -- calibration of the memory-side counters on RANDOM 16-byte gathers: how many bytes does one
// TCC_EA0_RDREQ stand for when every lane loads 16 bytes at a random 128-byte-aligned address of a 16 GiB table?
// Run under `rocprofv3 --pmc <counters> --kernel-trace` (the program directly after `--`): every access pattern is a kernel of its
// own name, launched three times with a KNOWN number of lane loads (printed), so the per-dispatch counter values divide out.
//   k_one        one 16-B load per lane at a random line
//   k_half       + a second 16-B load at +16  (the same 64-byte half of the same 128-byte line)
//   k_line       + a second 16-B load at +64  (the other 64-byte half of the same 128-byte line)
//   k_adjacent   + a second 16-B load at +128 (the adjacent 128-byte line)
//   k_two        + a second 16-B load at another random line
//   k_stream     every lane loads 16 consecutive bytes of a coalesced stream (the guide's calibration case: FETCH_SIZE = 1/2 the bytes)
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
__device__ __forceinline__ void body(const char *__restrict__ tab, uint64_t lines_mask, uint64_t dist, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t a = (x & lines_mask) << 7;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab + a);
        acc += v.x ^ v.y;
        if (MODE == 1) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(tab + a + dist);
            acc += w.x ^ w.y;
        } else if (MODE == 2) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(tab + ((mix64(x ^ 0x5555) & lines_mask) << 7));
            acc += w.x ^ w.y;
        }
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_one(const char *t, uint64_t m, int it, uint64_t *o) { body<0>(t, m, 0, it, o); }
__global__ void __launch_bounds__(256) k_half(const char *t, uint64_t m, int it, uint64_t *o) { body<1>(t, m, 16, it, o); }
__global__ void __launch_bounds__(256) k_line(const char *t, uint64_t m, int it, uint64_t *o) { body<1>(t, m, 64, it, o); }
__global__ void __launch_bounds__(256) k_adjacent(const char *t, uint64_t m, int it, uint64_t *o) { body<1>(t, m, 128, it, o); }
__global__ void __launch_bounds__(256) k_two(const char *t, uint64_t m, int it, uint64_t *o) { body<2>(t, m, 0, it, o); }
__global__ void __launch_bounds__(256) k_stream(const char *t, uint64_t bytes, uint64_t *o) {
    uint64_t acc = 0;
    for (uint64_t a = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; a < bytes; a += (uint64_t)gridDim.x * 256 * 16) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(t + a);
        acc += v.x ^ v.y;
    }
    if (acc == 0x123456789ull) o[0] = acc;
}
int main() {
    const size_t bytes = 16ull << 30;
    void *tab; uint64_t *out; CK(hipMalloc(&tab, bytes + (4ull << 20))); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, bytes + (4ull << 20)));
    const uint64_t lines = bytes >> 7;
    const int blocks = 256 * 8 * 4, iters = 32;
    printf("lanes per launch: %llu (each: one 16-B load at a random 128-B-aligned address of a %zu-byte table; k_half / k_line / k_adjacent / k_two: two loads)\n",
           (unsigned long long)blocks * 256ull * iters, bytes);
    printf("k_stream: %llu bytes read once, 16 B per lane, coalesced\n", (unsigned long long)(4ull << 30));
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_one, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
        hipLaunchKernelGGL(k_half, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
        hipLaunchKernelGGL(k_line, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
        hipLaunchKernelGGL(k_adjacent, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
        hipLaunchKernelGGL(k_two, dim3(blocks), dim3(256), 0, 0, (const char *)tab, lines - 1, iters, out);
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const char *)tab, (uint64_t)(4ull << 30), out);
        CK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
