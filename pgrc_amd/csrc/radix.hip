// radix.hip -- a stable LSD radix sort of 64-bit records by a bit field, hand-written for gfx950 (no library call).
//
// Who sorts with it: the position order of the Pg-order export (export.hip: records = position << 32 | read,
// ReadsMatchers.cpp:563-574) and, since round 5, the events of the Pg-vs-Pg matcher (mem.hip: 64-bit keys that carry a 64-bit
// value each -- the PAIRS form of the scatter kernel stages both and moves 40 bytes per record and pass).  (The hits of modes
// d / i / e were sorted with it for part of round 4; they are reduced without any sort now, seedidx.hip.)
//
// One pass over `dbits` <= 8 key bits, tiles of RX_TILE consecutive records:
//   k_rx_hist     every tile counts its records per digit (LDS atomics)        -> cnt[digit][tile]
//   exclusive scan of that matrix in (digit, tile) order (idxsort.hip's k_psc_*) -> where a tile's run of a digit starts
//   k_rx_scatter  ranks inside the tile by ballots (lanes of a group with one digit find each other with `dbits`
//                 ballots; per-wave counters in LDS; waves in order), the tile is laid out by digit in LDS and every
//                 digit's run goes out as one contiguous piece: whole lines, and equal digits keep their input order.
// Exact offsets from the count matrix: no look-back, no dependency between tiles, any dispatch order (idxsweep.hip tells
// why the decoupled look-back lost on this chip).  24 bytes move per record and pass (counted once, scattered once).
// Integer work, HBM-stream bound; records below ~2^32 in number (offsets are 32-bit: reads are counted in 32 bits anyway).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ctx.h"
#include "devutil.h"

#define RX_TPB 1024
#define RX_NW (RX_TPB / 64)
#define RX_E 8
#define RX_TILE (RX_TPB * RX_E)          // 8192 records = 64 KB of LDS staging: one block of 16 waves per CU
#define RX_WSPAN (64 * RX_E)             // consecutive records per wave
#define RX_MAXD 256

__global__ void __launch_bounds__(RX_TPB)
k_rx_hist(const uint64_t *__restrict__ in, uint64_t n, uint32_t shift, uint32_t dmask, uint64_t ntiles, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t hist[RX_MAXD];
    for (uint32_t d = threadIdx.x; d <= dmask; d += RX_TPB) hist[d] = 0;
    __syncthreads();
    const uint64_t tile = blockIdx.x;
    const uint64_t base = tile * RX_TILE;
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const uint64_t x = base + (uint64_t)i * RX_TPB + threadIdx.x;
        if (x < n) atomicAdd(&hist[(uint32_t)(in[x] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= dmask; d += RX_TPB) cnt[(uint64_t)d * ntiles + tile] = hist[d];
}

struct RxLds {
    uint64_t recS[RX_TILE];
    uint16_t hist[RX_NW][RX_MAXD];       // per wave: running count (<= 512), then the wave's offset inside its digit
    uint32_t dstart[RX_MAXD];            // first slot of a digit in the sorted tile
    uint32_t gbase[RX_MAXD];             // where the tile's run of a digit starts in the output
    uint32_t scan_tmp[RX_NW + 1];
};

__device__ __forceinline__ uint32_t rx_block_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63) smem[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < nwv; k++) {
        const uint32_t s = smem[k];
        if (k < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// Records arrive as RX_E groups per wave; group i of wave w holds the tile's records w * 512 + i * 64 + lane, so "earlier
// wave, then earlier group, then lower lane" is the input order -- and the order equal digits leave in.
template <bool PAIRS>
__global__ void __launch_bounds__(RX_TPB)
k_rx_scatter(const uint64_t *__restrict__ in, const uint64_t *__restrict__ vin, uint64_t n, uint32_t shift, uint32_t dbits, uint64_t ntiles,
             const uint32_t *__restrict__ base, uint64_t *__restrict__ out, uint64_t *__restrict__ vout) {
    __shared__ RxLds s;
    extern __shared__ __attribute__((aligned(16))) uint64_t valS[];      // PAIRS: the values, staged beside their keys (64 KB)
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t D = 1u << dbits, dmask = D - 1u;
    const uint64_t tile = blockIdx.x;
    const uint64_t t0 = tile * RX_TILE;
    const uint32_t nvalid = (uint32_t)min((uint64_t)RX_TILE, n - t0);
    uint64_t rec[RX_E];
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
        rec[i] = x < nvalid ? in[t0 + x] : 0ull;
    }
    uint64_t val[PAIRS ? RX_E : 1];
    if (PAIRS) {
#pragma unroll
        for (int i = 0; i < RX_E; i++) {
            const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
            val[PAIRS ? i : 0] = x < nvalid ? vin[t0 + x] : 0ull;
        }
    }
    for (uint32_t x = threadIdx.x; x < RX_NW * RX_MAXD; x += RX_TPB) (&s.hist[0][0])[x] = (uint16_t)0;
    for (uint32_t d = threadIdx.x; d < D; d += RX_TPB) s.gbase[d] = base[(uint64_t)d * ntiles + tile];
    __syncthreads();
    uint32_t rank[RX_E];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const bool valid = wv * RX_WSPAN + (uint32_t)i * 64u + lane < nvalid;
        const uint32_t d = (uint32_t)(rec[i] >> shift) & dmask;
        unsigned long long peers = __ballot(valid);
        for (uint32_t b = 0; b < dbits; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        // the lowest lane of every peer group (always a valid lane) advances the wave's counter of that digit
        const uint32_t leader = valid ? (uint32_t)__ffsll((long long)peers) - 1u : lane;
        uint32_t old = 0;
        if (valid && lane == leader) {
            old = s.hist[wv][d];
            s.hist[wv][d] = (uint16_t)(old + (uint32_t)__popcll(peers));
        }
        old = __shfl(old, leader, 64);
        rank[i] = old + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // per digit: the waves' counts -> their offsets inside the digit; digit totals -> digit starts
    uint32_t tot = 0;
    if (threadIdx.x < D) {
        const uint32_t d = threadIdx.x;
        for (uint32_t w = 0; w < RX_NW; w++) {
            const uint32_t t = s.hist[w][d];
            s.hist[w][d] = (uint16_t)tot;
            tot += t;
        }
    }
    uint32_t all;
    const uint32_t ex = rx_block_scan(tot, s.scan_tmp, &all);
    if (threadIdx.x < D) s.dstart[threadIdx.x] = ex;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const bool valid = wv * RX_WSPAN + (uint32_t)i * 64u + lane < nvalid;
        if (valid) {
            const uint32_t d = (uint32_t)(rec[i] >> shift) & dmask;
            const uint32_t slot = s.dstart[d] + s.hist[wv][d] + rank[i];
            s.recS[slot] = rec[i];
            if (PAIRS) valS[slot] = val[PAIRS ? i : 0];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nvalid; j += RX_TPB) {
        const uint64_t r = s.recS[j];
        const uint32_t d = (uint32_t)(r >> shift) & dmask;
        const uint64_t dest = (uint64_t)s.gbase[d] + (j - s.dstart[d]);
        out[dest] = r;
        if (PAIRS) vout[dest] = valS[j];
    }
}

// ---- segments: every block sorts ONE segment of at most RX_TILE (key, value) pairs completely, by the key bits [bit_lo, bit_hi),
// in LDS -- LSD passes of 8 bits with the ranking of k_rx_scatter, nothing leaves the CU between the passes -- and writes it back
// in place.  For arrays that a few global passes over the TOP bits have cut into many small segments (seedidx.hip: 300 M pairs,
// two global passes, 65 536 segments of ~4 600: 2 + 1 trips through HBM instead of 8).  A segment larger than RX_TILE is left as
// it is and reported: ovl[0] counts them, ovl[1 .. cap] lists their numbers (the caller sorts those ranges with rx_sort).
__global__ void __launch_bounds__(RX_TPB)
k_rx_segments(uint64_t *__restrict__ keys, uint64_t *__restrict__ vals, const uint32_t *__restrict__ seg, uint32_t bit_lo, uint32_t bit_hi,
              uint32_t top_bits, uint32_t *__restrict__ ovl, uint32_t cap) {
    __shared__ RxLds s;
    extern __shared__ __attribute__((aligned(16))) uint64_t valS[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t s0 = seg[blockIdx.x], n = seg[blockIdx.x + 1] - s0;
    if (n < 2u) return;
    if (n > RX_TILE) {
        if (threadIdx.x == 0) {
            const uint32_t k = atomicAdd(&ovl[0], 1u);
            if (k < cap) ovl[1u + k] = blockIdx.x;
        }
        return;
    }
    uint64_t rec[RX_E], val[RX_E];
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
        rec[i] = x < n ? keys[(uint64_t)s0 + x] : 0ull;
        val[i] = x < n ? vals[(uint64_t)s0 + x] : 0ull;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // A field wider than 32 bits: the passes over its TOP 32 bits first.  At most RX_TILE keys that differ anywhere almost surely
    // differ there (two of 8 192 random keys agree in 32 given bits with probability 2^-7 per segment), and keys that are EQUAL
    // need no order: if no neighbours with equal top bits are out of order afterwards, the segment is sorted -- four passes
    // instead of six for a 48-bit field.  Otherwise (rare) all the passes run, from the lowest bit, on what is there by then.
    // (top_bits = 32; tests pass fewer, so that the long way is taken)
    uint32_t first_shift = bit_hi - bit_lo > top_bits ? bit_lo + ((bit_hi - bit_lo - top_bits) & ~7u) : bit_lo;
    __shared__ uint32_t unsorted;
    for (uint32_t shift = first_shift; shift < bit_hi; shift += 8u) {
        const uint32_t dbits = bit_hi - shift < 8u ? bit_hi - shift : 8u, D = 1u << dbits, dmask = D - 1u;
        for (uint32_t x = threadIdx.x; x < RX_NW * RX_MAXD; x += RX_TPB) (&s.hist[0][0])[x] = (uint16_t)0;
        __syncthreads();
        uint32_t rank[RX_E];
        const bool wave_has = wv * RX_WSPAN < n;                  // (a segment fills the first waves only: ~4 600 of 8 192 slots at C3)
#pragma unroll
        for (int i = 0; i < RX_E; i++) {
            rank[i] = 0;
            if (!wave_has || wv * RX_WSPAN + (uint32_t)i * 64u >= n) continue;    // (wave-uniform)
            const bool valid = wv * RX_WSPAN + (uint32_t)i * 64u + lane < n;
            const uint32_t d = (uint32_t)(rec[i] >> shift) & dmask;
            unsigned long long peers = __ballot(valid);
            for (uint32_t b = 0; b < dbits; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t leader = valid ? (uint32_t)__ffsll((long long)peers) - 1u : lane;
            uint32_t old = 0;
            if (valid && lane == leader) {
                old = s.hist[wv][d];
                s.hist[wv][d] = (uint16_t)(old + (uint32_t)__popcll(peers));
            }
            old = __shfl(old, leader, 64);
            rank[i] = old + (uint32_t)__popcll(peers & lt);
        }
        __syncthreads();
        uint32_t tot = 0;
        if (threadIdx.x < D) {
            const uint32_t d = threadIdx.x;
            for (uint32_t w = 0; w < RX_NW; w++) {
                const uint32_t t = s.hist[w][d];
                s.hist[w][d] = (uint16_t)tot;
                tot += t;
            }
        }
        uint32_t all;
        const uint32_t ex = rx_block_scan(tot, s.scan_tmp, &all);
        if (threadIdx.x < D) s.dstart[threadIdx.x] = ex;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RX_E; i++) {
            if (wv * RX_WSPAN + (uint32_t)i * 64u + lane < n) {
                const uint32_t d = (uint32_t)(rec[i] >> shift) & dmask;
                const uint32_t slot = s.dstart[d] + s.hist[wv][d] + rank[i];
                s.recS[slot] = rec[i];
                valS[slot] = val[i];
            }
        }
        __syncthreads();
        const bool last = shift + 8u >= bit_hi;
        if (last && first_shift != bit_lo && threadIdx.x == 0) unsorted = 0u;
#pragma unroll
        for (int i = 0; i < RX_E; i++) {                       // back in the input layout: slot order = the order of this pass
            const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
            if (x < n) {
                rec[i] = s.recS[x];
                val[i] = valS[x];
            }
        }
        __syncthreads();
        if (last && first_shift != bit_lo) {                   // the short way: is the segment sorted by the whole field?
            const uint64_t fmask = (bit_hi >= 64u ? ~0ull : (1ull << bit_hi) - 1ull) & ~((1ull << bit_lo) - 1ull);
            bool bad = false;
#pragma unroll
            for (int i = 0; i < RX_E; i++) {
                const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
                if (x > 0u && x < n && (s.recS[x] & fmask) < (s.recS[x - 1u] & fmask)) bad = true;
            }
            if (bad) unsorted = 1u;
            __syncthreads();
            if (unsorted) {                                    // (block-uniform) all the passes after all
                first_shift = bit_lo;
                shift = bit_lo - 8u;                           // (the loop's increment brings it to bit_lo)
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < RX_E; i++) {
        const uint32_t x = wv * RX_WSPAN + (uint32_t)i * 64u + lane;
        if (x < n) {
            keys[(uint64_t)s0 + x] = rec[i];
            vals[(uint64_t)s0 + x] = val[i];
        }
    }
}

// seg[0 .. nseg]: the bounds of the segments of keys[] / vals[] (device; ascending, seg[nseg] = the number of pairs); ovl: cap + 1
// words on the device, ovl[0] zeroed by the caller.  On c->stream, no synchronisation.
int pgrc_radix_sort_segments_pairs_u64(pgrc_match_ctx *c, uint64_t *keys, uint64_t *vals, const uint32_t *seg, uint32_t nseg, uint32_t bit_lo, uint32_t bit_hi,
                                       uint32_t *ovl, uint32_t cap) {
    const uint32_t top_bits = c->opt.test_segment_top_bits ? c->opt.test_segment_top_bits : 32u;
    if (!nseg || bit_hi <= bit_lo) return PGRC_OK;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_rx_segments), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RX_TILE * sizeof(uint64_t))));
    hipLaunchKernelGGL(k_rx_segments, dim3(nseg), dim3(RX_TPB), RX_TILE * sizeof(uint64_t), c->stream, keys, vals, seg, bit_lo, bit_hi, top_bits, ovl, cap);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// Sorts d_a[0 .. n) by the bits [bit_lo, bit_hi) of every record, stable; d_b: n records of scratch; `scratch` grows as
// needed (count matrix).  *sorted = d_a or d_b, wherever the last pass put the records.  With v_a / v_b (both or neither): every
// record carries a 64-bit value that moves with it; *vsorted = where the values ended up.  All on c->stream, no synchronisation.
static int rx_sort(pgrc_match_ctx *c, uint64_t *d_a, uint64_t *d_b, uint64_t *v_a, uint64_t *v_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi,
                   DevBuf &scratch, uint64_t **sorted, uint64_t **vsorted) {
    *sorted = d_a;
    if (vsorted) *vsorted = v_a;
    if (n < 2 || bit_hi <= bit_lo) return PGRC_OK;
    if (n >= 0xFFFFF000ull) { c->err = "radix sort: too many records"; return PGRC_E_PARAM; }
    const bool pairs = v_a != nullptr;
    const uint64_t ntiles = (n + RX_TILE - 1) / RX_TILE;
    const uint64_t ncnt = (uint64_t)RX_MAXD * ntiles;
    int e;
    if ((e = pgrc_buf_ensure(c, scratch, (ncnt + pgrc_ps_scan_blocks(ncnt) + 2) * sizeof(uint32_t) + 256))) return e;
    if (pairs) HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_rx_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RX_TILE * sizeof(uint64_t))));
    uint32_t *cnt = (uint32_t *)scratch.p, *bsum = cnt + ncnt;
    const uint32_t bits = bit_hi - bit_lo, passes = (bits + 7) / 8;
    uint64_t *src = d_a, *dst = d_b, *vsrc = v_a, *vdst = v_b;
    uint32_t shift = bit_lo;
    for (uint32_t p = 0; p < passes; p++) {
        // digits as even as the field allows (31 bits: 8 8 8 7)
        const uint32_t dbits = (bits - (shift - bit_lo) + (passes - p) - 1) / (passes - p);
        const uint32_t D = 1u << dbits;
        hipLaunchKernelGGL(k_rx_hist, dim3((uint32_t)ntiles), dim3(RX_TPB), 0, c->stream, (const uint64_t *)src, n, shift, D - 1u, ntiles, cnt);
        HIP_TRY(c, hipGetLastError());
        if ((e = pgrc_ps_scan_u32(c, cnt, (uint64_t)D * ntiles, bsum))) return e;
        if (pairs)
            hipLaunchKernelGGL(k_rx_scatter<true>, dim3((uint32_t)ntiles), dim3(RX_TPB), RX_TILE * sizeof(uint64_t), c->stream, (const uint64_t *)src, (const uint64_t *)vsrc, n,
                               shift, dbits, ntiles, (const uint32_t *)cnt, dst, vdst);
        else
            hipLaunchKernelGGL(k_rx_scatter<false>, dim3((uint32_t)ntiles), dim3(RX_TPB), 0, c->stream, (const uint64_t *)src, (const uint64_t *)nullptr, n, shift, dbits,
                               ntiles, (const uint32_t *)cnt, dst, (uint64_t *)nullptr);
        HIP_TRY(c, hipGetLastError());
        std::swap(src, dst);
        std::swap(vsrc, vdst);
        shift += dbits;
    }
    *sorted = src;
    if (vsorted) *vsorted = vsrc;
    return PGRC_OK;
}

int pgrc_radix_sort_u64(pgrc_match_ctx *c, uint64_t *d_a, uint64_t *d_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi, DevBuf &scratch,
                        uint64_t **sorted) {
    return rx_sort(c, d_a, d_b, nullptr, nullptr, n, bit_lo, bit_hi, scratch, sorted, nullptr);
}

int pgrc_radix_sort_pairs_u64(pgrc_match_ctx *c, uint64_t *k_a, uint64_t *k_b, uint64_t *v_a, uint64_t *v_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi,
                              DevBuf &scratch, uint64_t **ksorted, uint64_t **vsorted) {
    return rx_sort(c, k_a, k_b, v_a, v_b, n, bit_lo, bit_hi, scratch, ksorted, vsorted);
}
