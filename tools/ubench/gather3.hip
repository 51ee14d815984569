// gather3.hip -- can scalar loads (s_load via the scalar data cache) add to the vector gather ceiling?
// Each iteration every lane has one random 16-B address; the first S lanes of each wave are served by s_load_dwordx4.
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
template <int S>
__global__ void __launch_bounds__(256) k(const ulonglong2 *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
        const uint64_t idx = x & mask;
        ulonglong2 v = make_ulonglong2(0, 0);
        if (lane >= S) v = tab[idx];
#pragma unroll
        for (int l = 0; l < S; l++) {
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)idx, l), hi = __builtin_amdgcn_readlane((uint32_t)(idx >> 32), l);
            const ulonglong2 sv = tab[((uint64_t)hi << 32) | lo];   // uniform address -> s_load_dwordx4
            if (lane == l) v = sv;
        }
        acc += v.x ^ v.y;
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <int S> static void run(const void *tab, size_t bytes, uint64_t *out) {
    const int blocks = 256 * 8 * 4, iters = 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<S>, dim3(blocks), dim3(256), 0, 0, (const ulonglong2 *)tab, bytes / 16 - 1, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("scalar lanes per wave = %2d : %7.2f G lane-addresses/s\n", S, (double)blocks * 256 * iters / best / 1e6); fflush(stdout);
}
int main() {
    const size_t bytes = 8ull << 30;
    void *tab; uint64_t *out; CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, bytes));
    run<0>(tab, bytes, out); run<4>(tab, bytes, out); run<8>(tab, bytes, out); run<12>(tab, bytes, out);
    run<16>(tab, bytes, out); run<24>(tab, bytes, out); run<32>(tab, bytes, out);
    return 0;
}
