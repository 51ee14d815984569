// rocPRIM onesweep configurations for the index sort (u32 bucket key, u64 entry, 29 key bits, 375 M records)
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdint>
#include <rocprim/device/device_radix_sort.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void fill(uint32_t *k32, uint64_t *v64, uint64_t n, uint32_t bits) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        k32[i] = (uint32_t)(mix(i + 12345) & ((1u << bits) - 1)); v64[i] = i;
    }
}

static uint32_t *k32, *k32o; static uint64_t *v64, *v64o; static void *tmp; static size_t tmp_cap;
static uint64_t n; static uint32_t bits;

template <class Config>
int run(const char *name) {
    size_t t = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, t, k32, k32o, v64, v64o, n, 0, bits)));
    if (t > tmp_cap) { printf("%s: temp %zu too large\n", name, t); return 0; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CK(hipEventRecord(e0)); CK((rocprim::radix_sort_pairs<Config>(tmp, t, k32, k32o, v64, v64o, n, 0, bits))); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-28s %8.2f ms  %6.2f G/s\n", name, best, n / best * 1e-6);
    return 0;
}

template <unsigned BS, unsigned IPT, unsigned RB>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>>;

int main(int argc, char **argv) {
    n = argc > 1 ? strtoull(argv[1], 0, 10) : 375000000ull;
    bits = argc > 2 ? atoi(argv[2]) : 29;
    CK(hipMalloc(&k32, n * 4)); CK(hipMalloc(&k32o, n * 4)); CK(hipMalloc(&v64, n * 8)); CK(hipMalloc(&v64o, n * 8));
    tmp_cap = n * 16 + (64 << 20); CK(hipMalloc(&tmp, tmp_cap));
    fill<<<4096, 256>>>(k32, v64, n, bits);
    CK(hipDeviceSynchronize());
    run<rocprim::default_config>("default");
    run<cfg<1024, 8, 8>>("1024x8 r8 match");
    run<cfg<1024, 12, 8>>("1024x12 r8 match");
    run<cfg<512, 16, 8>>("512x16 r8 match");
    run<cfg<1024, 6, 8>>("1024x6 r8 match");
    run<cfg<1024, 4, 10>>("1024x4 r10 match");
    run<cfg<1024, 6, 10>>("1024x6 r10 match");
    run<cfg<512, 8, 10>>("512x8 r10 match");
    run<cfg<512, 12, 10>>("512x12 r10 match");
    return 0;
}
