// rocPRIM radix sort rate at the copMEM index's size: could a sort-based index build (stable sort of the sampled
// positions by bucket, then one streaming pass) beat the three sweeps of random atomics?
//   a) pairs u32 key (29 bits) + u64 value   b) keys-only u64, bits [32,61)   c) pairs u32 key + u32 value
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdint>
#include <rocprim/device/device_radix_sort.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void fill(uint32_t *k32, uint64_t *k64, uint64_t *v64, uint32_t *v32, uint64_t n, uint32_t bits) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(mix(i + 12345) & ((1u << bits) - 1));
        k32[i] = h; k64[i] = ((uint64_t)h << 32) | (uint32_t)(i * 5); v64[i] = i; v32[i] = (uint32_t)i;
    }
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 375000000ull;
    const uint32_t bits = argc > 2 ? atoi(argv[2]) : 29;
    uint32_t *k32, *k32o, *v32, *v32o; uint64_t *k64, *k64o, *v64, *v64o;
    CK(hipMalloc(&k32, n * 4)); CK(hipMalloc(&k32o, n * 4)); CK(hipMalloc(&v32, n * 4)); CK(hipMalloc(&v32o, n * 4));
    CK(hipMalloc(&k64, n * 8)); CK(hipMalloc(&k64o, n * 8)); CK(hipMalloc(&v64, n * 8)); CK(hipMalloc(&v64o, n * 8));
    fill<<<4096, 256>>>(k32, k64, v64, v32, n, bits);
    CK(hipDeviceSynchronize());
    size_t ta = 0, tb = 0, tc = 0;
    CK(rocprim::radix_sort_pairs(nullptr, ta, k32, k32o, v64, v64o, n, 0, bits));
    CK(rocprim::radix_sort_keys(nullptr, tb, k64, k64o, n, 32, 32 + bits));
    CK(rocprim::radix_sort_pairs(nullptr, tc, k32, k32o, v32, v32o, n, 0, bits));
    size_t tmax = ta > tb ? ta : tb; if (tc > tmax) tmax = tc;
    void *tmp; CK(hipMalloc(&tmp, tmax));
    printf("n=%llu bits=%u temp a=%zu b=%zu c=%zu MiB\n", (unsigned long long)n, bits, ta >> 20, tb >> 20, tc >> 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CK(hipEventRecord(e0)); CK(rocprim::radix_sort_pairs(tmp, ta, k32, k32o, v64, v64o, n, 0, bits)); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("a) pairs u32+u64   %8.2f ms  %6.2f G/s\n", ms, n / ms * 1e-6);
        CK(hipEventRecord(e0)); CK(rocprim::radix_sort_keys(tmp, tb, k64, k64o, n, 32, 32 + bits)); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("b) keys u64        %8.2f ms  %6.2f G/s\n", ms, n / ms * 1e-6);
        CK(hipEventRecord(e0)); CK(rocprim::radix_sort_pairs(tmp, tc, k32, k32o, v32, v32o, n, 0, bits)); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("c) pairs u32+u32   %8.2f ms  %6.2f G/s\n", ms, n / ms * 1e-6);
    }
    // sanity: output sorted?
    return 0;
}
