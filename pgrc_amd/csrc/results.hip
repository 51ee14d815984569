// results.hip -- result-state plumbing and mismatch extraction.
//
//   init   : DefaultReadsMatcher::initMatching, matching/ReadsMatchers.cpp:97-105, :411-415
//   hist   : matchedCountPerMismatches / matchedReadsCount (ReadsMatchers.h:35,116).  The reference
//            maintains them incrementally (:439-447); at any point they equal the histogram of
//            readMismatchesCount[], which is what this kernel computes once per run.
//   extract: AbstractReadsApproxMatcher::updateEntry :548-559 with fillEntryWithMismatches :40-51 and
//            fillEntryWithReversedMismatches :53-66; code = (val(pg)<<4)+val(read), helper.cpp:358-362.
#include "ctx.h"
#include "devutil.h"

__global__ void __launch_bounds__(256) k_init_results(uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        pos[i] = PGRC_NOT_MATCHED_POS;
        rc[i] = 0;
        mism[i] = PGRC_NOT_MATCHED_CNT;
    }
}

int pgrc_launch_init_results(pgrc_match_ctx *c) {
    if (!c->n) return PGRC_OK;
    uint32_t grid = (uint32_t)((c->n + 255) / 256 < 65536 ? (c->n + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_init_results, dim3(grid), dim3(256), 0, c->stream, (uint64_t *)c->d_pos.p, (uint8_t *)c->d_rc.p,
                       (uint8_t *)c->d_mism.p, c->n);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

__global__ void __launch_bounds__(256) k_hist(const uint8_t *__restrict__ mism, uint64_t n, unsigned long long *hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    // 16 counts per thread and pass keep the LDS counters far from u32 overflow
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[mism[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

int pgrc_launch_hist(pgrc_match_ctx *c) {
    HIP_TRY(c, hipMemsetAsync(c->d_hist.p, 0, 256 * sizeof(uint64_t), c->stream));
    if (c->n) {
        uint32_t grid = (uint32_t)((c->n + 255) / 256 < 2048 ? (c->n + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, c->stream, (const uint8_t *)c->d_mism.p, c->n,
                           (unsigned long long *)c->d_hist.p);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(c->hist, c->d_hist.p, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->matched = c->n - c->hist[PGRC_NOT_MATCHED_CNT];
    return PGRC_OK;
}

// ---------------------------------------------------------------- mismatch extraction

// exclusive u64 scan of per-read mismatch counts (255 -> 0): 3-kernel block scan
#define XS_TPB 256
#define XS_EPT 16
#define XS_EPB (XS_TPB * XS_EPT)

__device__ __forceinline__ uint32_t mcount(uint8_t m) { return m == PGRC_NOT_MATCHED_CNT ? 0u : (uint32_t)m; }

__device__ __forceinline__ uint32_t xs_block_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63) smem[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < XS_TPB / 64; k++) {
        uint32_t s = smem[k];
        if (k < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

__global__ void __launch_bounds__(XS_TPB) k_xs_sums(const uint8_t *__restrict__ mism, uint64_t n, uint64_t *bsum) {
    __shared__ uint32_t smem[XS_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * XS_EPB + (uint64_t)threadIdx.x * XS_EPT;
    uint32_t s = 0;
    for (int k = 0; k < XS_EPT; k++)
        if (base + k < n) s += mcount(mism[base + k]);
    uint32_t tot;
    xs_block_scan(s, smem, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void k_xs_bsums(uint64_t *bsum, uint64_t nb) { // nb is small (n/4096): one thread suffices
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint64_t i = 0; i < nb; i++) {
            uint64_t v = bsum[i];
            bsum[i] = run;
            run += v;
        }
        bsum[nb] = run;
    }
}

__global__ void __launch_bounds__(XS_TPB) k_xs_write(const uint8_t *__restrict__ mism, uint64_t n,
                                                     const uint64_t *__restrict__ bsum, uint64_t nb, uint64_t *cum) {
    __shared__ uint32_t smem[XS_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * XS_EPB + (uint64_t)threadIdx.x * XS_EPT;
    uint32_t v[XS_EPT], s = 0;
#pragma unroll
    for (int k = 0; k < XS_EPT; k++) {
        v[k] = (base + k < n) ? mcount(mism[base + k]) : 0;
        s += v[k];
    }
    uint32_t tot;
    uint64_t off = (uint64_t)xs_block_scan(s, smem, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < XS_EPT; k++) {
        if (base + k < n) cum[base + k] = off;
        off += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cum[n] = bsum[nb];
}

struct ExtractArgs {
    const uint32_t *pg;
    const uint32_t *reads;
    uint64_t n, stride;
    const uint8_t *nflag;
    const uint32_t *nidx;
    const uint8_t *nascii;
    uint64_t nn;
    const uint64_t *pos;
    const uint8_t *rc, *mism, *revflags;
    const uint64_t *cum;
    uint8_t *codes;
    uint16_t *offsets;
    uint32_t L;
};

// value of symbol x of the ORIGINAL read: 0..3, or 4 for 'N' (byte path)
template <bool ASCII>
__device__ __forceinline__ uint32_t read_val(const ExtractArgs &a, uint64_t i, uint64_t t, uint32_t x) {
    if (ASCII) {
        const uint8_t ch = a.nascii[t * a.L + x];
        return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    }
    return (a.reads[(uint64_t)(x >> 4) * a.stride + i] >> (2u * (x & 15u))) & 3u;
}
__device__ __forceinline__ uint32_t compl_val(uint32_t v) { return v < 4u ? 3u - v : 4u; } // N <-> N

template <bool ASCII>
__global__ void __launch_bounds__(256) k_extract(const ExtractArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i;
    if (ASCII) {
        if (t >= a.nn) return;
        i = a.nidx[t];
    } else {
        if (t >= a.n) return;
        i = t;
        if (a.nflag && a.nflag[i]) return;
    }
    const uint32_t cnt = mcount(a.mism[i]);
    if (!cnt) return;
    const uint64_t p = a.pos[i];
    const bool rc = a.rc[i] != 0;
    const bool reversed = a.revflags ? a.revflags[i] != 0 : rc;
    uint64_t o = a.cum[i];
    const uint32_t L = a.L;
    uint32_t emitted = 0;
    for (uint32_t step = 0; step < L && emitted < cnt; step++) {
        const uint32_t x = reversed ? L - 1u - step : step;     // index into the (possibly RC'd) read / Pg window
        uint32_t rv = rc ? compl_val(read_val<ASCII>(a, i, t, L - 1u - x)) : read_val<ASCII>(a, i, t, x);
        const uint64_t g = p + x;
        uint32_t pv = (a.pg[g >> 4] >> (2u * ((uint32_t)g & 15u))) & 3u;
        if (rv != pv) {
            if (reversed) { pv = compl_val(pv); rv = compl_val(rv); }
            a.codes[o] = (uint8_t)((pv << 4) + rv);
            a.offsets[o] = (uint16_t)step; // forward: x; reversed: L-1-x
            o++;
            emitted++;
        }
    }
}

// The lists on the device: d_cum (u64[n + 1], exclusive scan of the reads' mismatch counts), d_codes / d_offs (one entry per
// mismatch), *total.  d_revflags: per-read orientation flags ON THE DEVICE (nullptr: reversed iff the read matched the RC strand).
// Everything is queued on c->stream; the call returns after the scan's total is known (one synchronisation).
int pgrc_extract_lists_device(pgrc_match_ctx *c, const uint8_t *d_revflags, bool lists, DevBuf &d_cum, DevBuf &d_codes, DevBuf &d_offs, uint64_t *total_out) {
    const uint64_t n = c->n;
    DevBuf d_bsum;
    int e;
    const uint64_t nb = (n + XS_EPB - 1) / XS_EPB;
    if ((e = pgrc_buf_ensure(c, d_cum, (n + 1) * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, d_bsum, (nb + 2) * sizeof(uint64_t)))) return e;
    if (n) {
        hipLaunchKernelGGL(k_xs_sums, dim3((uint32_t)nb), dim3(XS_TPB), 0, c->stream, (const uint8_t *)c->d_mism.p, n, (uint64_t *)d_bsum.p);
        hipLaunchKernelGGL(k_xs_bsums, dim3(1), dim3(64), 0, c->stream, (uint64_t *)d_bsum.p, nb);
        hipLaunchKernelGGL(k_xs_write, dim3((uint32_t)nb), dim3(XS_TPB), 0, c->stream, (const uint8_t *)c->d_mism.p, n,
                           (const uint64_t *)d_bsum.p, nb, (uint64_t *)d_cum.p);
    } else {
        (void)hipMemsetAsync(d_cum.p, 0, sizeof(uint64_t), c->stream);
    }
    uint64_t total = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&total, (const uint64_t *)d_cum.p + n, sizeof total, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        pgrc_buf_free(d_bsum);
        c->err = "extract: scan failed";
        return PGRC_E_DEVICE;
    }
    pgrc_buf_free(d_bsum);
    *total_out = total;
    if (!total || !lists) return PGRC_OK;                   // (lists = false: only the counts' scan was asked for)
    if ((e = pgrc_buf_ensure(c, d_codes, total)) || (e = pgrc_buf_ensure(c, d_offs, total * sizeof(uint16_t)))) return e;
    ExtractArgs a;
    a.pg = (const uint32_t *)c->pg2[0].p;
    a.reads = c->reads2;
    a.n = n;
    a.stride = c->stride;
    a.nflag = c->n_nreads ? (const uint8_t *)c->nread_flag.p : nullptr;
    a.nidx = (const uint32_t *)c->nread_idx.p;
    a.nascii = (const uint8_t *)c->nread_ascii.p;
    a.nn = c->n_nreads;
    a.pos = (const uint64_t *)c->d_pos.p;
    a.rc = (const uint8_t *)c->d_rc.p;
    a.mism = (const uint8_t *)c->d_mism.p;
    a.revflags = d_revflags;
    a.cum = (const uint64_t *)d_cum.p;
    a.codes = (uint8_t *)d_codes.p;
    a.offsets = (uint16_t *)d_offs.p;
    a.L = c->prm.read_len;
    hipLaunchKernelGGL(k_extract<false>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    if (c->n_nreads)
        hipLaunchKernelGGL(k_extract<true>, dim3((uint32_t)((c->n_nreads + 255) / 256)), dim3(256), 0, c->stream, a);
    if (hipGetLastError() != hipSuccess) { c->err = "extract: kernel failed"; return PGRC_E_DEVICE; }
    return PGRC_OK;
}

int pgrc_extract_mismatches(pgrc_match_ctx *c, const uint8_t *reversed_flags, uint64_t *cum, uint8_t *codes,
                            uint16_t *offsets) {
    const uint64_t n = c->n;
    DevBuf d_cum, d_codes, d_offs, d_flags;
    int e;
    auto cleanup = [&]() { pgrc_buf_free(d_cum); pgrc_buf_free(d_codes); pgrc_buf_free(d_offs); pgrc_buf_free(d_flags); };
    if (reversed_flags && codes && offsets) {
        if ((e = pgrc_buf_ensure(c, d_flags, n ? n : 1))) { cleanup(); return e; }
        if (n && hipMemcpyAsync(d_flags.p, reversed_flags, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { cleanup(); return PGRC_E_DEVICE; }
    }
    uint64_t total = 0;
    const bool lists = codes && offsets;
    if ((e = pgrc_extract_lists_device(c, (reversed_flags && lists) ? (const uint8_t *)d_flags.p : nullptr, lists, d_cum, d_codes, d_offs, &total))) { cleanup(); return e; }
    bool ok = hipMemcpyAsync(cum, d_cum.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    if (ok && lists && total)
        ok = hipMemcpyAsync(codes, d_codes.p, total, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
             hipMemcpyAsync(offsets, d_offs.p, total * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    cleanup();
    if (!ok) { c->err = "extract: copy failed"; return PGRC_E_DEVICE; }
    return PGRC_OK;
}
