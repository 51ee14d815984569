// idxsweep.hip -- the default front end of the copMEM index build (mode c): the sampled
// positions of the text are grouped by the TOP bucket bits in two scatter passes, hand-written for gfx950, and handed to
// the in-LDS partition finish of idxsort.hip.
//
// What is built is the reference's SERIAL index (CopMEMMatcher::processRef / genCumm, matching/copmem/CopMEMMatcher.cpp:
// 140-231; hash: Hashes.h:54-76): bucket h = the 13 smallest sampled positions whose hash & mask is h, ascending.
//
// Why another front end.  The build is a stream: 0.47 GB of text in, 11.6 GB of heads and entries out (C3); everything
// in between is overhead.  Round 2 moved ~40 GB per strand (record generation 5, two library radix passes over 12-byte
// records with their histogram 19.5, bounds 1.5, finish 16).  This one moves ~28 GB:
//   * records are never "generated": the counting pre-pass and the first scatter pass hash the text themselves;
//   * a record shrinks as the passes decide its bucket bits: after pass 1 (bits [cb, cb+b1)) it is 8 bytes + the
//     pass-2 digit in a byte of its own, after pass 2 just 8 bytes: bucket bits below cb | position index t | 22-bit
//     fingerprint -- t = position / k1 is what makes it fit (C3: 13 + 29 + 22 bits);
//   * the counting pre-pass of pass 2 reads only that byte array (0.4 GB), not the records;
//   * pass 2 walks the pass-1 bins tile by tile, so the offsets of the first tile of bin d_lo are, for every d_hi, the
//     number of records of (d_hi, smaller d_lo): the start of partition (d_hi, d_lo); no bounds pass reads the records;
//   * tiles are ranked with LDS atomics, not ballots: the order of the records INSIDE a partition is not kept (the
//     finish kernel orders every bucket by position anyway and caps over-full buckets order-independently), which is
//     an eighth of the ranking work of a stable pass.
// Every tile knows where its digit runs go from an exclusive scan of the (digit, tile) count matrix (48 MB at C3), so
// tiles are independent: no look-back, no spinning, any dispatch order.  (A one-sweep variant with decoupled look-back
// over 8-byte {flag, count} granules was built first and measured: its INC frontier advances by at most
// `tiles looked at per step / granule load latency` ~ 15 tiles/us chip-wide, which capped a pass at ~3 ms at C3 whatever
// the block shape -- profiles/r03_os_lookback_cfgs.txt.  Direct stores without LDS staging were 2-4x slower.)
//
// Round 5, measured and not kept (profiles/r05_index_variants.txt): pass 1 WITHOUT the counting pre-pass -- every bin a fixed
// number of slots, a tile reserves the run of each of its digits with one returning global atomic, the exact road queued behind
// it and skipped unless a bin ran over (device-side flag).  The reserving pass took 2.7 ms where counting (0.67) + scattering
// (1.75) take 2.4: 256 dependent atomics per tile cost more than hashing the text a second time, and the records of a partition
// then arrive in no order at all, which the general finish kernel of the flagged partitions pays for (0.16 -> 1.8 ms).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "ctx.h"
#include "devutil.h"

#define OS_NW 16                          // waves of the largest block
#define OS_MAXD 512u                      // digits per pass: at most 9 bits
#define OS_CB_MAX 13u                     // bucket bits the finish kernel takes (8192 buckets per partition)

struct OsPlan {
    uint32_t hbits, cb, b1, b2;           // pass 1 sorts bucket bits [cb, cb+b1), pass 2 bits [cb+b1, hbits)
    uint32_t tbits, rec_sh;               // bits of t; rec_sh = tbits + 22 = where the bucket bits start in a record
    uint32_t k1, K, mask;
    uint64_t n, ntiles1, ntiles2_max;     // sampled positions, tiles of pass 1, upper bound of the tiles of pass 2
    uint32_t xcd;                         // 1: tile t of a pass goes to the block that shares an XCD (and its L2) with the blocks of tiles t - 1, t + 1
};

// Which tile block b of nb takes.  The dispatcher hands consecutive blocks to the 8 XCDs in turn (b % 8 labels the blocks that
// share an XCD and its L2: cdna_hip_programming.md T1); the tiles of a scatter pass write runs that are NEIGHBOURS in every
// digit's bin, so two tiles in a row leave every line between their runs half written.  With the remap those halves meet in
// one L2 (a speed choice only: any placement gives the same bytes).
__device__ __forceinline__ uint32_t os_tile_of_block(uint32_t b, uint32_t nb, uint32_t xcd) {
    if (!xcd) return b;
    const uint32_t x = b & 7u, q = nb >> 3, r = nb & 7u;
    return x * q + min(x, r) + (b >> 3);
}

__device__ __forceinline__ uint32_t os_block_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
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

// ---------------------------------------------------------------- hashing a tile of the text

// The K-symbol window at sampled position t of a tile whose text words sit in LDS (txt[0] = word w0 of the text).
// KQ = K/4 when known at compile time (7: the default seed 38), 0 = any K.
template <int KQ>
__device__ __forceinline__ void os_hash_at(const uint32_t *txt, uint64_t w0, uint64_t p, uint32_t K, const uint32_t *lut, uint32_t *hash, uint32_t *fp) {
    const uint32_t q = (uint32_t)((p >> 4) - w0);
    const uint32_t sh = ((uint32_t)p & 15u) * 2u;
    const uint32_t a0 = txt[q], a1 = txt[q + 1], a2 = txt[q + 2], a3 = txt[q + 3], a4 = txt[q + 4];
    const uint32_t w[4] = {funnel_r(a0, a1, sh), funnel_r(a1, a2, sh), funnel_r(a2, a3, sh), funnel_r(a3, a4, sh)};
    if (KQ == 0) {
        *hash = copmem_hash32_fp(w[0], w[1], w[2], w[3], K, lut, fp);
        return;
    }
    uint32_t h = 4u * KQ, f = 0, fb = 0;
#pragma unroll
    for (int j = 0; j < KQ; j++) {
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t x = (j < 3) ? lut[b & 63u] : lut[64u + (b & 15u)];
        h = (h ^ (x + (uint32_t)j)) * 171717u;
        const uint32_t width = (j < 3) ? 2u : 4u;
        if (fb + width <= PGRC_FP_BITS) {
            f |= ((j < 3) ? (b >> 6) : (b >> 4)) << fb;
            fb += width;
        }
    }
    *hash = h;
    *fp = f;
}

// The low 24 bits of the same hash (all the counting pre-pass needs when the pass-1 digit lies below bit 24): xor and
// multiply are closed modulo 2^24, and v_mul_u32_u24 runs at full rate where the 32-bit multiply takes four issue slots.
template <int KQ>
__device__ __forceinline__ uint32_t os_hash24_at(const uint32_t *txt, uint64_t w0, uint64_t p, uint32_t K, const uint32_t *lut) {
    const uint32_t q = (uint32_t)((p >> 4) - w0);
    const uint32_t sh = ((uint32_t)p & 15u) * 2u;
    const uint32_t a0 = txt[q], a1 = txt[q + 1], a2 = txt[q + 2], a3 = txt[q + 3], a4 = txt[q + 4];
    const uint32_t w[4] = {funnel_r(a0, a1, sh), funnel_r(a1, a2, sh), funnel_r(a2, a3, sh), funnel_r(a3, a4, sh)};
    uint32_t h = K;
    const int kq = KQ ? KQ : (int)(K >> 2);
#pragma unroll
    for (int j = 0; j < (KQ ? KQ : 14); j++) {
        if (!KQ && j >= kq) break;
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t x = (j < 3) ? lut[b & 63u] : lut[64u + (b & 15u)];
        h = __umul24(h ^ (x + (uint32_t)j), 171717u);
    }
    return h & 0xFFFFFFu;
}

// words of text a tile of `tile` sampled positions needs: tile * k1 symbols + the last window + alignment slack
__host__ __device__ __forceinline__ uint32_t os_txt_words(uint32_t tile, uint32_t k1, uint32_t K) { return (tile * k1 + K + 15u) / 16u + 6u; }

// loads the text words of tile [t0, t0 + TILE) into LDS (coalesced); returns the first word's index
__device__ __forceinline__ uint64_t os_stage_text(const uint32_t *__restrict__ pg, uint64_t pg_words_alloc, uint64_t t0, uint32_t tile, uint32_t k1,
                                                  uint32_t K, uint32_t *txt) {
    const uint64_t w0 = (t0 * k1) >> 4;
    const uint32_t need = os_txt_words(tile, k1, K);
    for (uint32_t w = threadIdx.x; w < need; w += blockDim.x) txt[w] = (w0 + w < pg_words_alloc) ? pg[w0 + w] : 0u;
    return w0;
}

// ---------------------------------------------------------------- counting pre-passes: records per (digit, tile)

// pass 1: tile = TPB * E consecutive sampled positions; cnt[d * ntiles + tile]
template <int KQ, int TPB, int E>
__global__ void __launch_bounds__(TPB)
k_os_count_gen(const uint32_t *__restrict__ pg, uint64_t pg_words_alloc, const OsPlan pl, uint32_t *__restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];   // (64-bit records are staged in it: 8-byte LDS accesses must be aligned)
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t h1[OS_MAXD];
    constexpr uint32_t TILE = TPB * E;
    uint32_t *txt = dyn;
    hash_lut_init(lut);
    const uint32_t D1 = 1u << pl.b1, m1 = D1 - 1u;
    for (uint32_t d = threadIdx.x; d < D1; d += TPB) h1[d] = 0;
    const uint64_t tile = os_tile_of_block(blockIdx.x, gridDim.x, pl.xcd), t0 = tile * TILE;
    const uint64_t w0 = os_stage_text(pg, pg_words_alloc, t0, TILE, pl.k1, pl.K, txt);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint64_t t = t0 + (uint64_t)i * TPB + threadIdx.x;
        if (t < pl.n) {
            uint32_t h;
            if (pl.cb + pl.b1 <= 24u && pl.hbits >= 24u) {
                h = os_hash24_at<KQ>(txt, w0, t * pl.k1, pl.K, lut);
            } else {
                uint32_t fp;
                os_hash_at<KQ>(txt, w0, t * pl.k1, pl.K, lut, &h, &fp);
                h &= pl.mask;
            }
            atomicAdd(&h1[(h >> pl.cb) & m1], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < D1; d += TPB) cnt[(uint64_t)d * pl.ntiles1 + tile] = h1[d];
}

// the pass-2 tiles: every pass-1 bin is cut into tiles of its own (so a tile never mixes two values of d_lo)
struct __attribute__((aligned(16))) OsBinTile {
    uint32_t r0, nvalid;                  // first record, records
    uint32_t bin, first;                  // the pass-1 bin (= d_lo; all ones: no such tile), whether this is the bin's first tile
};

// one thread per pass-2 tile: its descriptor (last d with tile_start[d] <= tile)
__global__ void __launch_bounds__(256)
k_os_tiles(const OsPlan pl, const uint32_t *__restrict__ off1, const uint32_t *__restrict__ tile_start, uint32_t tile_size, OsBinTile *__restrict__ desc) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= pl.ntiles2_max) return;
    const uint32_t D1 = 1u << pl.b1;
    OsBinTile bt = {0u, 0u, 0xFFFFFFFFu, 0u};
    if (tile < tile_start[D1]) {
        uint32_t lo = 0, hi = D1;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile_start[mid] <= tile) lo = mid;
            else hi = mid;
        }
        const uint32_t q = tile - tile_start[lo];
        const uint64_t bin_lo = off1[(uint64_t)lo * pl.ntiles1], bin_hi = lo + 1 < D1 ? off1[(uint64_t)(lo + 1) * pl.ntiles1] : pl.n;
        const uint64_t r0 = bin_lo + (uint64_t)q * tile_size;
        bt.r0 = (uint32_t)r0;
        bt.nvalid = r0 < bin_hi ? (uint32_t)min((uint64_t)tile_size, bin_hi - r0) : 0u;
        bt.bin = lo;
        bt.first = q == 0;
    }
    desc[tile] = bt;
}

// pass 2: counts from the digit bytes alone, 16 of them per thread; cnt[d * ntiles2_max + tile] (tiles past the last one: zeros)
template <int TPB, typename AUX>
__global__ void __launch_bounds__(TPB)
k_os_count_bins(const AUX *__restrict__ aux_in, const OsBinTile *__restrict__ desc, const OsPlan pl, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t h2[OS_MAXD];
    const uint32_t D2 = 1u << pl.b2, tile = blockIdx.x;
    for (uint32_t d = threadIdx.x; d < D2; d += TPB) h2[d] = 0;
    const OsBinTile bt = desc[tile];
    __syncthreads();
    constexpr uint32_t PER = 16 / sizeof(AUX);                  // digits per 16-byte load
    const uint64_t lo = bt.r0, hi = lo + bt.nvalid, a0 = lo & ~(uint64_t)(PER - 1);
    for (uint64_t x = a0 + (uint64_t)threadIdx.x * PER; x < hi; x += (uint64_t)TPB * PER) {
        const uint4 w = *reinterpret_cast<const uint4 *>(aux_in + x);     // (aligned; the buffer has 16 spare entries at its end)
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t dgt = sizeof(AUX) == 1 ? (ww[k >> 2] >> (8 * (k & 3))) & 0xFFu : (ww[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            if (x + k >= lo && x + k < hi) atomicAdd(&h2[dgt], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < D2; d += TPB) cnt[(uint64_t)d * pl.ntiles2_max + tile] = h2[d];
}

// one block, after the scan of the pass-1 counts: the pass-2 tiles of every pass-1 bin, the end of the last partition
__global__ void __launch_bounds__(1024) k_os_prepare(const OsPlan pl, const uint32_t *__restrict__ off1, uint32_t *__restrict__ tile_start,
                                                      uint32_t *__restrict__ pstart, uint32_t tile) {
    __shared__ uint32_t smem[OS_NW + 1];
    const uint32_t D1 = 1u << pl.b1, d = threadIdx.x;
    uint32_t c1 = 0;
    if (d < D1) {
        const uint64_t lo = off1[(uint64_t)d * pl.ntiles1], hi = d + 1 < D1 ? off1[(uint64_t)(d + 1) * pl.ntiles1] : pl.n;
        c1 = (uint32_t)(hi - lo);
    }
    // every pass-1 bin gets at least one pass-2 tile (an empty one still hands on the partition starts of its d_lo)
    const uint32_t nt = d < D1 ? max(1u, (c1 + tile - 1u) / tile) : 0u;
    uint32_t tot;
    const uint32_t et = os_block_scan(nt, smem, &tot);
    if (d < D1) tile_start[d] = et;
    if (d == 0) {
        tile_start[D1] = tot;
        pstart[1u << (pl.hbits - pl.cb)] = (uint32_t)pl.n;
    }
}

// ---------------------------------------------------------------- one tile of a scatter pass

// LDS of a scatter block beyond the staging arrays
struct OsTileLds {
    uint32_t cnt[OS_MAXD];                // records of the tile per digit (the rank counters)
    uint32_t dstart[OS_MAXD];             // first slot of a digit in the staged tile
    uint32_t gbase[OS_MAXD];              // where the tile's run of a digit starts in the output
    uint32_t scan_tmp[OS_NW + 1];
};

// ranks E records per thread by digit (LDS atomics: any order inside a digit), stages the records digit by digit in LDS
// and streams the runs out.  s.gbase[] must be loaded by the caller (before or after: a barrier follows the ranking).
// dig[i] = OS_MAXD for a record that does not exist.
template <int TPB, int E, typename AUX, typename DIG, bool HAS_AUX>
__device__ __forceinline__ void os_scatter_tile(OsTileLds &s, uint64_t *recS, AUX *auxS, DIG *digS, const uint64_t (&rec)[E], const AUX (&aux)[E],
                                                const uint32_t (&dig)[E], uint32_t nvalid, uint32_t D, uint64_t *__restrict__ rec_out,
                                                AUX *__restrict__ aux_out) {
    uint32_t rank[E];
#pragma unroll
    for (int i = 0; i < E; i++) rank[i] = dig[i] < OS_MAXD ? atomicAdd(&s.cnt[dig[i]], 1u) : 0u;
    __syncthreads();
    uint32_t run = 0;                                            // exclusive scan of the digit counts (D <= 2 * TPB)
    {
        const uint32_t per = (D + TPB - 1) / TPB, d0 = threadIdx.x * per;
        uint32_t c[2] = {0, 0};
        for (uint32_t q = 0; q < per; q++) c[q] = d0 + q < D ? s.cnt[d0 + q] : 0u;
        uint32_t tot;
        run = os_block_scan(c[0] + c[1], s.scan_tmp, &tot);
        for (uint32_t q = 0; q < per; q++) {
            if (d0 + q < D) {
                s.dstart[d0 + q] = run;
                s.gbase[d0 + q] -= run;                          // from here on: output index of staged slot j = gbase[digit] + j
            }
            run += c[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) {
        if (dig[i] < OS_MAXD) {
            const uint32_t slot = s.dstart[dig[i]] + rank[i];
            recS[slot] = rec[i];
            if (HAS_AUX) auxS[slot] = aux[i];
            digS[slot] = (DIG)dig[i];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nvalid; j += TPB) {
        const uint64_t dest = (uint64_t)(s.gbase[digS[j]] + j);      // (mod 2^32 arithmetic: the sum is the true index < 2^32)
        rec_out[dest] = recS[j];
        if (HAS_AUX) aux_out[dest] = auxS[j];
    }
}

// ---------------------------------------------------------------- pass 1: text -> records in bins of bucket bits [cb, cb+b1)

template <int KQ, int TPB, int E, int MINB, typename AUX, typename DIG>
__global__ void __launch_bounds__(TPB, (MINB * TPB) / 256)      // (HIP: threads per block, WAVES PER SIMD)
k_os_scatter_gen(const uint32_t *__restrict__ pg, uint64_t pg_words_alloc, const OsPlan pl, const uint32_t *__restrict__ off1,
                 uint64_t *__restrict__ rec_out, AUX *__restrict__ aux_out) {
    constexpr uint32_t TILE = TPB * E;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];   // (64-bit records are staged in it: 8-byte LDS accesses must be aligned)
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ OsTileLds s;
    // dynamic LDS: recS[TILE] u64 | auxS[TILE] | digS[TILE]; the text words of the tile alias recS (dead once hashed)
    uint64_t *recS = reinterpret_cast<uint64_t *>(dyn);
    DIG *digS = reinterpret_cast<DIG *>(recS + TILE);           // (the wider type first: alignment)
    AUX *auxS = reinterpret_cast<AUX *>(digS + TILE);
    uint32_t *txt = dyn;
    hash_lut_init(lut);
    const uint32_t D1 = 1u << pl.b1, m1 = D1 - 1u, cbmask = (1u << pl.cb) - 1u;
    const uint32_t tile = os_tile_of_block(blockIdx.x, gridDim.x, pl.xcd);
    for (uint32_t d = threadIdx.x; d < D1; d += TPB) {
        s.cnt[d] = 0;
        s.gbase[d] = off1[(uint64_t)d * pl.ntiles1 + tile];
    }
    const uint64_t t0 = (uint64_t)tile * TILE;
    const uint64_t w0 = os_stage_text(pg, pg_words_alloc, t0, TILE, pl.k1, pl.K, txt);
    __syncthreads();
    uint64_t rec[E];
    AUX aux[E];
    uint32_t dig[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint64_t t = t0 + (uint64_t)i * TPB + threadIdx.x;
        uint32_t h = 0, fp = 0;
        if (t < pl.n) os_hash_at<KQ>(txt, w0, t * pl.k1, pl.K, lut, &h, &fp);
        h &= pl.mask;
        rec[i] = ((uint64_t)(h & cbmask) << pl.rec_sh) | (t << PGRC_FP_BITS) | fp;
        aux[i] = (AUX)(h >> (pl.cb + pl.b1));
        dig[i] = t < pl.n ? (h >> pl.cb) & m1 : OS_MAXD;
    }
    __syncthreads();                                             // the text words are dead: recS may be written
    const uint32_t nvalid = (uint32_t)min((uint64_t)TILE, pl.n - t0);
    os_scatter_tile<TPB, E, AUX, DIG, true>(s, recS, auxS, digS, rec, aux, dig, nvalid, D1, rec_out, aux_out);
}

// ---------------------------------------------------------------- pass 2: bins -> partitions of bucket bits [cb, hbits)

template <int TPB, int E, int MINB, typename AUX>
__global__ void __launch_bounds__(TPB, (MINB * TPB) / 256)
k_os_scatter_bins(const uint64_t *__restrict__ rec_in, const AUX *__restrict__ aux_in, const OsPlan pl, const OsBinTile *__restrict__ desc,
                  const uint32_t *__restrict__ off2, uint64_t *__restrict__ rec_out, uint32_t *__restrict__ pstart) {
    constexpr uint32_t TILE = TPB * E;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];   // (64-bit records are staged in it: 8-byte LDS accesses must be aligned)
    __shared__ OsTileLds s;
    uint64_t *recS = reinterpret_cast<uint64_t *>(dyn);
    AUX *digS = reinterpret_cast<AUX *>(recS + TILE);
    const uint32_t D2 = 1u << pl.b2, tile = os_tile_of_block(blockIdx.x, gridDim.x, pl.xcd);
    const OsBinTile bt = desc[tile];
    if (bt.bin == 0xFFFFFFFFu) return;
    for (uint32_t d = threadIdx.x; d < D2; d += TPB) {
        s.cnt[d] = 0;
        const uint32_t g = off2[(uint64_t)d * pl.ntiles2_max + tile];
        s.gbase[d] = g;
        // the first tile of a bin: its offsets are, per d_hi, the records of (d_hi, smaller d_lo) = the start of partition (d_hi, d_lo)
        if (bt.first) pstart[(d << pl.b1) | bt.bin] = g;
    }
    uint64_t rec[E];
    AUX aux[E] = {};
    uint32_t dig[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t j = (uint32_t)i * TPB + threadIdx.x;
        const bool ok = j < bt.nvalid;
        rec[i] = ok ? rec_in[(uint64_t)bt.r0 + j] : 0ull;
        dig[i] = ok ? (uint32_t)aux_in[(uint64_t)bt.r0 + j] : OS_MAXD;
    }
    __syncthreads();
    os_scatter_tile<TPB, E, AUX, AUX, false>(s, recS, (AUX *)nullptr, digS, rec, aux, dig, bt.nvalid, D2, rec_out, (AUX *)nullptr);
}

// ---------------------------------------------------------------- driver

static bool os_plan(const pgrc_match_ctx *c, uint32_t hbits, OsPlan *pl) {
    const uint64_t n = c->npos;
    if (!n || n >= 0xFFFFF000ull || hbits < 16 || hbits > 31 || c->cp.k1 > 16 || c->cp.K > 56 || c->cp.K < 4) return false;
    uint32_t tbits = 1;
    while ((1ull << tbits) < n) tbits++;
    if (tbits + PGRC_FP_BITS + 8u > 64u) return false;
    const uint32_t cb = std::min<uint32_t>(OS_CB_MAX, 64u - PGRC_FP_BITS - tbits);
    if (cb < 12u || hbits < cb + 2u) return false;             // (the finish kernel works in rounds of 4096 buckets)
    const uint32_t top = hbits - cb;
    if (top > 18u) return false;                               // two passes of at most 9 bits
    pl->hbits = hbits;
    pl->cb = cb;
    pl->b1 = top / 2;
    pl->b2 = top - pl->b1;
    pl->tbits = tbits;
    pl->rec_sh = tbits + PGRC_FP_BITS;
    pl->k1 = (uint32_t)c->cp.k1;
    pl->K = (uint32_t)c->cp.K;
    pl->mask = (uint32_t)(c->cp.hash_size - 1);
    pl->n = n;
    pl->ntiles1 = pl->ntiles2_max = 0;
    pl->xcd = 1;
    return true;
}

bool pgrc_os_applicable(const pgrc_match_ctx *c, uint32_t hbits) {
    OsPlan pl;
    return os_plan(c, hbits, &pl);
}

template <typename F>
static hipError_t os_allow_lds(F *kernel, size_t bytes) {
    // more than the default 64 KB of dynamic LDS must be asked for
    return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess;
}

struct OsBufs {
    uint32_t *cnt1, *cnt2, *tile_start, *pstart, *slow, *bsum;
    OsBinTile *desc;
    uint64_t *recA, *recB;
    void *aux;
};

// everything between the text and the finish, in one block shape: TPB threads x E records, MINB blocks per CU; AUX = type of the
// pass-2 digit that travels with a pass-1 record, DIG = type of a staged pass-1 digit
template <int TPB, int E, int MINB, typename AUX, typename DIG>
static int os_passes(pgrc_match_ctx *c, int strand, const OsPlan &pl, const OsBufs &b) {
    constexpr uint32_t TILE = TPB * E;
    const uint32_t *pg = (const uint32_t *)c->pg2[strand].p;
    const uint64_t pgw = c->pg_words + PGRC_PG_PAD_WORDS;
    const uint32_t D1 = 1u << pl.b1, D2 = 1u << pl.b2;
    const size_t txt_bytes = (size_t)os_txt_words(TILE, pl.k1, pl.K) * sizeof(uint32_t);
    const size_t lds1 = std::max<size_t>((size_t)TILE * (8 + sizeof(AUX) + sizeof(DIG)), txt_bytes), lds2 = (size_t)TILE * (8 + sizeof(AUX));
    const bool k7 = pl.K == 28;
    hipError_t he = k7 ? os_allow_lds(k_os_scatter_gen<7, TPB, E, MINB, AUX, DIG>, lds1) : os_allow_lds(k_os_scatter_gen<0, TPB, E, MINB, AUX, DIG>, lds1);
    if (he == hipSuccess) he = os_allow_lds(k_os_scatter_bins<TPB, E, MINB, AUX>, lds2);
    if (he == hipSuccess) he = k7 ? os_allow_lds(k_os_count_gen<7, TPB, E>, txt_bytes) : os_allow_lds(k_os_count_gen<0, TPB, E>, txt_bytes);
    if (he != hipSuccess) { c->err = std::string("index build: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
    const dim3 g1((uint32_t)pl.ntiles1), g2((uint32_t)pl.ntiles2_max), blk(TPB);
    int e;
    // pass 1
    if (k7) hipLaunchKernelGGL((k_os_count_gen<7, TPB, E>), g1, blk, txt_bytes, c->stream, pg, pgw, pl, b.cnt1);
    else hipLaunchKernelGGL((k_os_count_gen<0, TPB, E>), g1, blk, txt_bytes, c->stream, pg, pgw, pl, b.cnt1);
    if ((e = pgrc_ps_scan_u32(c, b.cnt1, (uint64_t)D1 * pl.ntiles1, b.bsum))) return e;
    hipLaunchKernelGGL(k_os_prepare, dim3(1), dim3(1024), 0, c->stream, pl, (const uint32_t *)b.cnt1, b.tile_start, b.pstart, TILE);
    if (k7) hipLaunchKernelGGL((k_os_scatter_gen<7, TPB, E, MINB, AUX, DIG>), g1, blk, lds1, c->stream, pg, pgw, pl, (const uint32_t *)b.cnt1, b.recA, (AUX *)b.aux);
    else hipLaunchKernelGGL((k_os_scatter_gen<0, TPB, E, MINB, AUX, DIG>), g1, blk, lds1, c->stream, pg, pgw, pl, (const uint32_t *)b.cnt1, b.recA, (AUX *)b.aux);
    // pass 2
    hipLaunchKernelGGL(k_os_tiles, dim3((uint32_t)((pl.ntiles2_max + 255) / 256)), dim3(256), 0, c->stream, pl, (const uint32_t *)b.cnt1, (const uint32_t *)b.tile_start, TILE, b.desc);
    hipLaunchKernelGGL((k_os_count_bins<512, AUX>), g2, dim3(512), 0, c->stream, (const AUX *)b.aux, (const OsBinTile *)b.desc, pl, b.cnt2);
    if ((e = pgrc_ps_scan_u32(c, b.cnt2, (uint64_t)D2 * pl.ntiles2_max, b.bsum))) return e;
    hipLaunchKernelGGL((k_os_scatter_bins<TPB, E, MINB, AUX>), g2, blk, lds2, c->stream, (const uint64_t *)b.recA, (const AUX *)b.aux, pl,
                       (const OsBinTile *)b.desc, (const uint32_t *)b.cnt2, b.recB, b.pstart);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

int pgrc_os_build_index(pgrc_match_ctx *c, int strand, uint32_t hbits) {
    OsPlan pl;
    if (!os_plan(c, hbits, &pl)) { c->err = "index build (sweep): not applicable"; return PGRC_E_PARAM; }
    // XCD-aware tile order in the passes (os_tile_of_block; round 5): at C3 pass 2 goes 1.83 -> 1.38 ms per strand, pass 1 1.81 ->
    // 1.74, the pair of builds 18.5 -> 17.1 ms in one context (profiles/r05_index_variants.txt).  PGRC_INDEX_CFG=0: off (A/B runs).
    pl.xcd = c->opt.index_cfg == 0 ? 0u : 1u;
    // Block shape of the passes: 1024 threads x 6 records, 61-68 KB of LDS and 64 registers: two blocks per CU, one hashing while
    // the other stores.  Measured at C3 in round 3, per strand (profiles/r03_os_cfgs.txt): pass 1 / pass 2 = 3.0 / 1.8 ms with 1024
    // x 8 (one block per CU), 1.8 / 1.75 with 1024 x 6, 2.2 / 2.1 with 512 x 8 (three per CU), 3.3 / 2.75 with 512 x 16, 3.0 / 2.4
    // with a persistent, prefetching pass 2.  Nine-bit digits in BOTH passes (tables of 2^30 buckets and more: 16-bit digit
    // arrays) take 1024 x 5: two blocks per CU fit (round 5; 1024 x 8, one per CU, until then: the C5 shard's pair of builds 39.8 ->
    // 38.3 ms, P64's 50.8 -> 48.9 in one context -- a tile's run of one of 512 digits is 10 records, 80 bytes, either way).
    const bool aux16 = pl.b2 > 8;
    const uint64_t n = pl.n, tile = aux16 ? 5120u : 6144u;
    const uint32_t D1 = 1u << pl.b1, D2 = 1u << pl.b2, np = 1u << (pl.hbits - pl.cb);
    pl.ntiles1 = (n + tile - 1) / tile;
    pl.ntiles2_max = pl.ntiles1 + D1;
    int e;
    // buffers (grow-only, shared with the other front end): d_sval[0] = pass-1 records, later ent[]; d_sval[1] = pass-2
    // records; d_skey[0] = the pass-2 digit of every pass-1 record; d_sorttmp = count matrices, tables, flags
    if ((e = pgrc_buf_ensure(c, c->d_sval[0], (n + 16) * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, c->d_sval[1], (n + 16) * sizeof(uint64_t))) ||
        (e = pgrc_buf_ensure(c, c->d_skey[0], (n + 16) * sizeof(uint32_t))))
        return e;
    const uint64_t n1 = (uint64_t)D1 * pl.ntiles1, n2 = (uint64_t)D2 * pl.ntiles2_max;
    const uint64_t nbs = pgrc_ps_scan_blocks(std::max(n1, n2)) + 2;
    const uint64_t flag_words = 2ull * np + 2;                 // np flags, the list of flagged partitions, its length (idxsort.hip)
    const uint64_t words = flag_words + (OS_MAXD + 2) + (np + 2) + nbs + n1 + n2 + 4 * (pl.ntiles2_max + 1) + 64;
    if ((e = pgrc_buf_ensure(c, c->d_sorttmp, words * sizeof(uint32_t)))) return e;
    OsBufs b;
    b.slow = (uint32_t *)c->d_sorttmp.p;
    b.tile_start = b.slow + flag_words;
    b.pstart = b.tile_start + OS_MAXD + 2;
    b.bsum = b.pstart + np + 2;
    b.cnt1 = b.bsum + nbs;
    b.cnt2 = b.cnt1 + n1;
    b.desc = (OsBinTile *)(((uintptr_t)(b.cnt2 + n2) + 15) & ~(uintptr_t)15);
    b.recA = (uint64_t *)c->d_sval[0].p;
    b.recB = (uint64_t *)c->d_sval[1].p;
    b.aux = c->d_skey[0].p;
    HIP_TRY(c, hipMemsetAsync(b.slow, 0, flag_words * sizeof(uint32_t), c->stream));
    if (aux16) e = os_passes<1024, 5, 2, uint16_t, uint16_t>(c, strand, pl, b);
    else e = os_passes<1024, 6, 2, uint8_t, uint8_t>(c, strand, pl, b);
    if (e) return e;
    return pgrc_ps_finish_packed(c, b.recB, b.pstart, b.slow, np, pl.cb, pl.rec_sh, b.recA);
}
