// biggrid.hip -- does a 1-D grid of more than 2^31 threads execute every thread exactly once with the right indexes?
// (round 4: k_seed_hamming_min over 3.7 G hits in one launch returned wrong results; 1 G per launch was right.)
// Every thread adds 1 to a counter per block-index bucket and checks that its global index is below n.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_count(uint64_t n, unsigned long long *total, unsigned long long *xsum, unsigned int *maxblk) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = x < n ? 1ull : 0ull;
    // wave-level reduction, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    unsigned long long xs = x < n ? x : 0ull;
    for (int o = 32; o > 0; o >>= 1) xs += __shfl_xor(xs, o, 64);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(total, mine);
        atomicAdd(xsum, xs);
        atomicMax(maxblk, blockIdx.x);
    }
}

int main(int argc, char **argv) {
    unsigned long long *d;
    unsigned int *m;
    hipMalloc(&d, 16);
    hipMalloc(&m, 4);
    const uint64_t sizes[] = {1ull << 30, (1ull << 31) - 256, 1ull << 31, (1ull << 31) + 256, 3ull << 30, 3700000000ull, (1ull << 32) - 256};
    for (uint64_t n : sizes) {
        hipMemset(d, 0, 16);
        hipMemset(m, 0, 4);
        const uint32_t grid = (uint32_t)((n + 255) / 256);
        hipLaunchKernelGGL(k_count, dim3(grid), dim3(256), 0, 0, n, d, d + 1, m);
        hipError_t e = hipDeviceSynchronize();
        unsigned long long h[2];
        unsigned int mb;
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        hipMemcpy(&mb, m, 4, hipMemcpyDeviceToHost);
        // expected: total = n, xsum = n (n - 1) / 2 mod 2^64
        const unsigned long long want = (unsigned long long)(((__uint128_t)n * (n - 1) / 2));
        printf("n %llu grid %u: err %d threads counted %llu (%s) index sum %s max block %u (%s)\n", (unsigned long long)n, grid, (int)e, h[0],
               h[0] == n ? "ok" : "WRONG", h[1] == want ? "ok" : "WRONG", mb, mb == grid - 1 ? "ok" : "WRONG");
    }
    return 0;
}
