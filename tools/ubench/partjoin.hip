// partjoin.hip -- measurement only (tools/ubench_partjoin.py; built by `make -C pgrc_amd/csrc partjoin` into
// tools/ubench/libpgrc_partjoin.so = the product objects + this file; NOT part of libpgrc_match.so): can PARTITIONED probing beat random probing of the bucket heads?
//
// VERDICT r03 item 7.  The match kernels gather one bucket head per probed seed at the chip's random-request rate.  The
// alternative: write the probes out as records (bucket, probe id), partition them by the top 16 bucket bits with the
// library's own scatter passes (radix.hip), then join every partition against its 8192 heads staged in LDS -- streams
// instead of gathers.  This entry point times exactly that against the plain gather of the same probes, on a table of
// 2^hbits 16-byte heads:
//   ms[0] generating the records        ms[1] the random gather (one 16-byte load per record)
//   ms[2] partitioning (two passes)     ms[3] the join (heads of a partition in LDS, its records streamed past)
// Both ways fold what they read into one checksum, which must agree.  Nothing of the product path calls this.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ctx.h"
#include "devutil.h"

int pgrc_radix_sort_u64(pgrc_match_ctx *c, uint64_t *d_a, uint64_t *d_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi, DevBuf &scratch,
                        uint64_t **sorted);

#define UB_IDBITS 34u

__device__ __forceinline__ uint64_t ub_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_ub_heads(ulonglong2 *__restrict__ head, uint64_t hs) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < hs; h += (uint64_t)gridDim.x * blockDim.x)
        head[h] = make_ulonglong2(ub_mix(h), ub_mix(h ^ 0x5555555555555555ull));
}

__global__ void __launch_bounds__(256) k_ub_records(uint64_t *__restrict__ rec, uint64_t n, uint32_t hbits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        rec[i] = ((ub_mix(i + 1) >> (64 - hbits)) << UB_IDBITS) | i;
}

__device__ __forceinline__ uint64_t ub_fold(const ulonglong2 hd, uint64_t id) { return (hd.x ^ (hd.y >> 7)) + id * 0x9E3779B97F4A7C15ull; }

__global__ void __launch_bounds__(256) k_ub_gather(const uint64_t *__restrict__ rec, uint64_t n, const ulonglong2 *__restrict__ head,
                                                   unsigned long long *__restrict__ sum) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = rec[i];
        acc += ub_fold(head[r >> UB_IDBITS], r & ((1ull << UB_IDBITS) - 1ull));
    }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)acc);
}

// where every partition's records start (they are contiguous after the partitioning): one binary search per partition, all at
// once (the library's own passes get these starts from their count matrices for free)
#define UB_CB 13u
__global__ void __launch_bounds__(256) k_ub_pstart(const uint64_t *__restrict__ rec, uint64_t n, uint32_t np, uint64_t *__restrict__ pstart) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > np) return;
    const uint64_t key = (uint64_t)p << (UB_IDBITS + UB_CB);
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (rec[mid] < key) lo = mid + 1; else hi = mid;
    }
    pstart[p] = p == np ? n : lo;
}

// one block per partition in turn: the partition's 2^cb heads go to LDS as one contiguous run, its records stream past
__global__ void __launch_bounds__(1024) k_ub_join(const uint64_t *__restrict__ rec, const uint64_t *__restrict__ pstart, const ulonglong2 *__restrict__ head,
                                                  uint32_t np, unsigned long long *__restrict__ sum) {
    extern __shared__ ulonglong2 hl[];
    uint64_t acc = 0;
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        __syncthreads();
        const uint64_t range[2] = {pstart[p], pstart[p + 1]};
        for (uint32_t b = threadIdx.x; b < (1u << UB_CB); b += 1024) hl[b] = head[((uint64_t)p << UB_CB) + b];
        __syncthreads();
        for (uint64_t i = range[0] + threadIdx.x; i < range[1]; i += 1024) {
            const uint64_t r = rec[i];
            acc += ub_fold(hl[(r >> UB_IDBITS) & ((1u << UB_CB) - 1u)], r & ((1ull << UB_IDBITS) - 1ull));
        }
    }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)acc);
}

extern "C" int pgrc_match_ubench_partjoin(uint64_t n, uint32_t hbits, float ms[4], uint64_t sums[2]) {
    if (!ms || !sums || hbits < UB_CB + 2 || hbits > 30 || n < 1 || n >= (1ull << 32) - 8192) return PGRC_E_PARAM;
    pgrc_match_ctx c;
    DevBuf head, ra, rb, scratch, sum, pst;
    auto done = [&](int e) { for (DevBuf *b : {&head, &ra, &rb, &scratch, &sum, &pst}) pgrc_buf_free(*b); return e; };
    const uint64_t hs = 1ull << hbits;
    int e;
    if ((e = pgrc_buf_ensure(&c, head, hs * 16)) || (e = pgrc_buf_ensure(&c, ra, n * 8)) || (e = pgrc_buf_ensure(&c, rb, n * 8)) ||
        (e = pgrc_buf_ensure(&c, sum, 16)) || (e = pgrc_buf_ensure(&c, pst, ((hs >> UB_CB) + 2) * 8)))
        return done(e);
    if (hipFuncSetAttribute((const void *)k_ub_join, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((1u << UB_CB) * 16)) != hipSuccess) return done(PGRC_E_DEVICE);
    hipEvent_t ev[5];
    for (auto &x : ev)
        if (hipEventCreate(&x) != hipSuccess) return done(PGRC_E_DEVICE);
    (void)hipMemsetAsync(sum.p, 0, 16, c.stream);
    hipLaunchKernelGGL(k_ub_heads, dim3(4096), dim3(256), 0, c.stream, (ulonglong2 *)head.p, hs);
    (void)hipEventRecord(ev[0], c.stream);
    hipLaunchKernelGGL(k_ub_records, dim3(8192), dim3(256), 0, c.stream, (uint64_t *)ra.p, n, hbits);
    (void)hipEventRecord(ev[1], c.stream);
    hipLaunchKernelGGL(k_ub_gather, dim3(256 * 16), dim3(256), 0, c.stream, (const uint64_t *)ra.p, n, (const ulonglong2 *)head.p, (unsigned long long *)sum.p);
    (void)hipEventRecord(ev[2], c.stream);
    uint64_t *sorted = nullptr;
    if ((e = pgrc_radix_sort_u64(&c, (uint64_t *)ra.p, (uint64_t *)rb.p, n, UB_IDBITS + UB_CB, UB_IDBITS + hbits, scratch, &sorted))) return done(e);
    const uint32_t np = (uint32_t)(hs >> UB_CB);
    hipLaunchKernelGGL(k_ub_pstart, dim3((np + 256) / 256), dim3(256), 0, c.stream, (const uint64_t *)sorted, n, np, (uint64_t *)pst.p);
    (void)hipEventRecord(ev[3], c.stream);
    hipLaunchKernelGGL(k_ub_join, dim3(256), dim3(1024), (1u << UB_CB) * 16, c.stream, (const uint64_t *)sorted, (const uint64_t *)pst.p, (const ulonglong2 *)head.p,
                       np, (unsigned long long *)sum.p + 1);
    (void)hipEventRecord(ev[4], c.stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess) return done(PGRC_E_DEVICE);
    for (int k = 0; k < 4; k++) (void)hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]);
    if (hipMemcpy(sums, sum.p, 16, hipMemcpyDeviceToHost) != hipSuccess) return done(PGRC_E_DEVICE);
    for (auto &x : ev) (void)hipEventDestroy(x);
    return done(PGRC_OK);
}
