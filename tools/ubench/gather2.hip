// gather2.hip -- random-gather rate vs access width and per-lane ILP at a 4 GiB footprint.
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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ uint64_t fold(T v);
template <> __device__ __forceinline__ uint64_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint64_t fold<uint64_t>(uint64_t v) { return v; }
template <> __device__ __forceinline__ uint64_t fold<u32x4>(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T, int ILP>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        T v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) v[k] = tab[mix64(x + k * 0x9E3779B97F4A7C15ull) & mask];
#pragma unroll
        for (int k = 0; k < ILP; k++) acc += fold<T>(v[k]);
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
// scalar path: every lane's address is served by s_load (wave loops over its lanes)
__global__ void __launch_bounds__(256) k_gather_scalar(const uint64_t *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t x = mix64(blockIdx.x * 256ull + threadIdx.x + 1);
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint64_t idx = x & mask;
        uint64_t mine = 0;
        for (int l = 0; l < 64; l += 8) {
            uint64_t a[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t lo = __builtin_amdgcn_readlane((uint32_t)idx, l + k), hi = __builtin_amdgcn_readlane((uint32_t)(idx >> 32), l + k);
                a[k] = ((uint64_t)hi << 32) | lo;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = tab[a[k]]; // uniform address -> s_load_dwordx2
#pragma unroll
            for (int k = 0; k < 8; k++) if ((int)(threadIdx.x & 63) == l + k) mine = v[k];
        }
        acc += mine;
        x = mix64(x + i);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <typename F> static void timeit(const char *name, double n, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-28s %8.2f G lane-loads/s\n", name, n / best / 1e6); fflush(stdout);
}
int main() {
    const size_t bytes = 4ull << 30;
    void *tab; uint64_t *out; CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(tab, 0, bytes));
    const int blocks = 256 * 8 * 4, iters = 32;
    const double base = (double)blocks * 256 * iters;
#define RUN(T, ILP, NAME) timeit(NAME, base * ILP, [&] { hipLaunchKernelGGL((k_gather<T, ILP>), dim3(blocks), dim3(256), 0, 0, (const T *)tab, bytes / sizeof(T) - 1, iters, out); })
    RUN(uint32_t, 1, "4B ilp1"); RUN(uint64_t, 1, "8B ilp1"); RUN(u32x4, 1, "16B ilp1");
    RUN(uint32_t, 4, "4B ilp4"); RUN(uint64_t, 4, "8B ilp4"); RUN(u32x4, 4, "16B ilp4");
    RUN(uint64_t, 8, "8B ilp8");
    timeit("8B scalar (s_load)", base, [&] { hipLaunchKernelGGL(k_gather_scalar, dim3(blocks), dim3(256), 0, 0, (const uint64_t *)tab, bytes / 8 - 1, iters, out); });
    return 0;
}
