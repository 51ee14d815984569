// scanops.h -- a three-kernel block scan with any associative operator (the index build's k_psc_* scan, idxsort.hip, restated
// as templates): per-block folds, one block that scans them, per-block rescan with the carried-in prefix.  The operator need not
// commute (the folds keep the input order).  Used by the Pg-vs-Pg matcher (mem.hip), whose scans the library provided until
// round 5.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCO_TPB 256
#define SCO_EPT 16
#define SCO_EPB (SCO_TPB * SCO_EPT)

// inclusive fold of the block's values in thread order: returns the fold of everything BEFORE this thread (ident for thread 0),
// *total = the fold of the whole block
template <typename Op>
__device__ __forceinline__ uint32_t sco_block_exclusive(uint32_t v, Op op, uint32_t ident, uint32_t *smem, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc = op(u, inc);
    }
    uint32_t before = __shfl_up(inc, 1, 64);
    if (lane == 0) before = ident;
    if (lane == 63) smem[wv] = inc;
    __syncthreads();
    uint32_t woff = ident, tot = ident;
    for (uint32_t k = 0; k < nwv; k++) {
        const uint32_t s = smem[k];
        if (k < wv) woff = op(woff, s);
        tot = op(tot, s);
    }
    __syncthreads();
    *total = tot;
    return op(woff, before);
}

template <typename In, typename Xf, typename Op>
__global__ void __launch_bounds__(SCO_TPB) k_sco_sums(const In *__restrict__ in, uint64_t n, Xf xf, Op op, uint32_t ident, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t smem[SCO_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCO_EPB + (uint64_t)threadIdx.x * SCO_EPT;
    uint32_t s = ident;
    for (int k = 0; k < SCO_EPT; k++)
        if (base + k < n) s = op(s, xf(in[base + k]));
    uint32_t tot;
    sco_block_exclusive(s, op, ident, smem, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one block: bsum[i] = fold of the blocks before i
template <typename Op>
__global__ void __launch_bounds__(SCO_TPB) k_sco_bsums(uint32_t *bsum, uint64_t nb, Op op, uint32_t ident) {
    __shared__ uint32_t smem[SCO_TPB / 64 + 1];
    uint32_t run = ident;
    for (uint64_t b0 = 0; b0 < nb; b0 += SCO_TPB) {
        const uint64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : ident;
        uint32_t tot;
        const uint32_t ex = sco_block_exclusive(v, op, ident, smem, &tot);
        if (i < nb) bsum[i] = op(run, ex);
        run = op(run, tot);
    }
}

template <typename In, typename Xf, typename Op, bool INCLUSIVE>
__global__ void __launch_bounds__(SCO_TPB) k_sco_write(const In *__restrict__ in, uint32_t *__restrict__ out, uint64_t n, Xf xf, Op op, uint32_t ident,
                                                       const uint32_t *__restrict__ bsum) {
    __shared__ uint32_t smem[SCO_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCO_EPB + (uint64_t)threadIdx.x * SCO_EPT;
    uint32_t v[SCO_EPT], s = ident;
#pragma unroll
    for (int k = 0; k < SCO_EPT; k++) {
        v[k] = (base + k < n) ? xf(in[base + k]) : ident;
        s = op(s, v[k]);
    }
    uint32_t tot;
    uint32_t acc = op(bsum[blockIdx.x], sco_block_exclusive(s, op, ident, smem, &tot));
#pragma unroll
    for (int k = 0; k < SCO_EPT; k++) {
        const uint32_t inc = op(acc, v[k]);
        if (base + k < n) out[base + k] = INCLUSIVE ? inc : acc;
        acc = inc;
    }
}

struct ScoIdentity { __device__ uint32_t operator()(uint32_t x) const { return x; } };
struct ScoPlus { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };

static inline uint64_t sco_scratch_words(uint64_t n) { return (n + SCO_EPB - 1) / SCO_EPB + 2; }

// out[i] = fold of xf(in[0 .. i]) (INCLUSIVE) or of xf(in[0 .. i-1]) with ident in front; in and out may be the same array when In
// is uint32_t; d_bsum: sco_scratch_words(n) words.  On `stream`, no synchronisation.
template <bool INCLUSIVE, typename In, typename Xf, typename Op>
static inline hipError_t sco_scan(hipStream_t stream, const In *in, uint32_t *out, uint64_t n, Xf xf, Op op, uint32_t ident, uint32_t *d_bsum) {
    if (!n) return hipSuccess;
    const uint64_t nb = (n + SCO_EPB - 1) / SCO_EPB;
    hipLaunchKernelGGL((k_sco_sums<In, Xf, Op>), dim3((uint32_t)nb), dim3(SCO_TPB), 0, stream, in, n, xf, op, ident, d_bsum);
    hipLaunchKernelGGL((k_sco_bsums<Op>), dim3(1), dim3(SCO_TPB), 0, stream, d_bsum, nb, op, ident);
    hipLaunchKernelGGL((k_sco_write<In, Xf, Op, INCLUSIVE>), dim3((uint32_t)nb), dim3(SCO_TPB), 0, stream, in, out, n, xf, op, ident, (const uint32_t *)d_bsum);
    return hipGetLastError();
}
