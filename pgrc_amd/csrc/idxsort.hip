// idxsort.hip -- the copMEM seed index built by a hand-written partition sort (mode c).
//
// What is built (reference semantics: CopMEMMatcher::processRef / genCumm, matching/copmem/CopMEMMatcher.cpp:140-231,
// SERIAL build): bucket h = the first 13 sampled positions, ascending, whose hash (Hashes.h:54-76) & mask equals h.
// The records (bucket, entry = position << 22 | fingerprint) are generated in position order, so any STABLE
// arrangement by bucket reproduces the serial index.  rocPRIM's general radix sort did that in four 8-bit passes
// (11.7 ms at C3) between a generator kernel and a head-writing kernel (16.3 ms per strand in all).  This build
// knows more about the job than a general sort can:
//   * the keys are hash values: cheap to RECOMPUTE from the text, so the first pass needs no stored input at all
//     (its histogram pre-pass and its scatter both hash the text: k_ps_hist_gen, k_ps_scatter_gen);
//   * a full sort is not needed: after two stable scatter passes over the TOP hbits-13 bucket bits the records of
//     8192 consecutive buckets are contiguous (a "partition", ~5.7 k records at C3), and one block finishes a
//     partition in LDS -- counting, the 13-entry cap, placement, bucket heads -- and streams its 128 KB of heads out
//     (k_ps_finish).  Entries past a bucket's 13th are never written (the reference drops them: "skippedList").
// Passes:  hist(text) -> scan -> scatter(text -> A) -> hist(A keys) -> scan -> scatter(A -> B) -> bounds(B keys) -> finish.
// Bytes moved at C3: 0.5 + 0.5+4.5 + 1.5 + 9.0 + 1.5 + 4.5(+4.5 from L2) + 11.6 = 34 GB (rocPRIM path: 55 GB).
// Everything is deterministic: ranks come from ballots and prefix sums, never from the arrival order of atomics
// (the one place where LDS atomics place records, k_ps_finish, re-sorts a bucket's <= 13 entries by position and
// resolves the cap with a deterministic rank).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "ctx.h"
#include "devutil.h"
#include "headfmt.h"

#define PS_TPB 1024
#define PS_NW (PS_TPB / 64)
#define PS_E 8
#define PS_TILE (PS_TPB * PS_E)          // 8192 records per tile (118 KB of LDS: one block of 16 waves per CU; 4096-record tiles, two blocks per CU, were 4 % slower)
#define PS_WSPAN (64 * PS_E)             // consecutive records per wave
#define PS_MAXD 512                      // digits per scatter pass (<= 9 bits)
#define PS_CB 13u                        // bucket bits finished in LDS: 8192 buckets per partition
#define PS_TXT_WORDS 80                  // text words one wave stages for 64 sampled positions: (15 + 63*16 + 56 + 15)/16 + 5

struct PsPlan {
    uint32_t hbits, b1, b2;              // pass 1 sorts bucket bits [13, 13+b1), pass 2 bits [13+b1, hbits)
    uint64_t n, ntiles;
};

// ---------------------------------------------------------------- exclusive scan u32 -> u32 (counts of one pass)
#define PSC_TPB 256
#define PSC_EPT 16
#define PSC_EPB (PSC_TPB * PSC_EPT)

__device__ __forceinline__ uint32_t psc_block_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
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

__global__ void __launch_bounds__(PSC_TPB) k_psc_sums(const uint32_t *__restrict__ in, uint64_t n, uint32_t *bsum) {
    __shared__ uint32_t smem[PSC_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * PSC_EPB + (uint64_t)threadIdx.x * PSC_EPT;
    uint32_t s = 0;
    if (base + PSC_EPT <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < PSC_EPT / 4; k++) { const uint4 q = p[k]; s += q.x + q.y + q.z + q.w; }
    } else {
        for (int k = 0; k < PSC_EPT; k++)
            if (base + k < n) s += in[base + k];
    }
    uint32_t tot;
    psc_block_scan(s, smem, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(PSC_TPB) k_psc_bsums(uint32_t *bsum, uint64_t nb) {
    __shared__ uint32_t smem[PSC_TPB / 64 + 1];
    uint32_t run = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += PSC_TPB) {
        const uint64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0;
        uint32_t tot;
        const uint32_t ex = psc_block_scan(v, smem, &tot);
        if (i < nb) bsum[i] = run + ex;
        run += tot;
    }
}

__global__ void __launch_bounds__(PSC_TPB) k_psc_write(uint32_t *__restrict__ io, uint64_t n, const uint32_t *__restrict__ bsum) {
    __shared__ uint32_t smem[PSC_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * PSC_EPB + (uint64_t)threadIdx.x * PSC_EPT;
    uint32_t v[PSC_EPT], s = 0;
#pragma unroll
    for (int k = 0; k < PSC_EPT; k++) {
        v[k] = (base + k < n) ? io[base + k] : 0;
        s += v[k];
    }
    uint32_t tot;
    uint32_t off = psc_block_scan(s, smem, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < PSC_EPT; k++) {
        if (base + k < n) io[base + k] = off;
        off += v[k];
    }
}

uint64_t pgrc_ps_scan_blocks(uint64_t n) { return (n + PSC_EPB - 1) / PSC_EPB; }

// in place: counts -> exclusive prefix sums (d_bsum: pgrc_ps_scan_blocks(n) + 1 words of scratch)
static int ps_scan(pgrc_match_ctx *c, uint32_t *d_io, uint64_t n, uint32_t *d_bsum) {
    const uint64_t nb = (n + PSC_EPB - 1) / PSC_EPB;
    hipLaunchKernelGGL(k_psc_sums, dim3((uint32_t)nb), dim3(PSC_TPB), 0, c->stream, (const uint32_t *)d_io, n, d_bsum);
    hipLaunchKernelGGL(k_psc_bsums, dim3(1), dim3(PSC_TPB), 0, c->stream, d_bsum, nb);
    hipLaunchKernelGGL(k_psc_write, dim3((uint32_t)nb), dim3(PSC_TPB), 0, c->stream, d_io, n, (const uint32_t *)d_bsum);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

int pgrc_ps_scan_u32(pgrc_match_ctx *c, uint32_t *d_io, uint64_t n, uint32_t *d_bsum) { return ps_scan(c, d_io, n, d_bsum); }

// ---------------------------------------------------------------- record generation (one wave = 64 consecutive sampled positions)

struct GenArgs {
    const uint32_t *pg;
    uint64_t pg_words_alloc, npos;
    uint32_t k1, K, mask;
};

// hashes sampled position t (valid lanes only compute meaningful values); txt = this wave's staging area
__device__ __forceinline__ void ps_gen64(const GenArgs &g, uint64_t t_first, uint32_t lane, uint32_t *txt, const uint32_t *lut,
                                         uint32_t *key, uint64_t *val) {
    const uint64_t p0 = t_first * g.k1;
    const uint64_t w0 = p0 >> 4;
    const uint32_t need = (uint32_t)((((p0 & 15) + 63ull * g.k1 + g.K + 15) >> 4) + 5);
    for (uint32_t w = lane; w < need; w += 64) txt[w] = (w0 + w < g.pg_words_alloc) ? g.pg[w0 + w] : 0u;
    __syncthreads();
    const uint64_t p = (t_first + lane) * g.k1;
    const uint32_t q = (uint32_t)((p >> 4) - w0);
    const uint32_t sh = ((uint32_t)p & 15u) * 2u;
    const uint32_t a0 = txt[q], a1 = txt[q + 1], a2 = txt[q + 2], a3 = txt[q + 3], a4 = txt[q + 4];
    uint32_t fp;
    *key = copmem_hash32_fp(funnel_r(a0, a1, sh), funnel_r(a1, a2, sh), funnel_r(a2, a3, sh), funnel_r(a3, a4, sh), g.K, lut, &fp) & g.mask;
    *val = (p << PGRC_FP_BITS) | fp;
    __syncthreads();
}

// pass-1 histogram straight from the text: cnt[d * ntiles + tile]
__global__ void __launch_bounds__(PS_TPB)
k_ps_hist_gen(const GenArgs g, uint32_t shift, uint32_t dmask, uint64_t ntiles, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t txt[PS_NW][PS_TXT_WORDS];
    __shared__ uint32_t hist[PS_MAXD];
    hash_lut_init(lut);
    for (uint32_t d = threadIdx.x; d <= dmask; d += PS_TPB) hist[d] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x;
    const uint64_t wbase = tile * PS_TILE + (uint64_t)wv * PS_WSPAN;
    for (int i = 0; i < PS_E; i++) {
        const uint64_t tf = wbase + (uint64_t)i * 64;
        uint32_t key;
        uint64_t val;
        ps_gen64(g, tf, lane, txt[wv], lut, &key, &val);
        if (tf + lane < g.npos) atomicAdd(&hist[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= dmask; d += PS_TPB) cnt[(uint64_t)d * ntiles + tile] = hist[d];
}

// histogram of a later pass from stored keys
__global__ void __launch_bounds__(PS_TPB)
k_ps_hist_keys(const uint32_t *__restrict__ keys, uint64_t n, uint32_t shift, uint32_t dmask, uint64_t ntiles, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t hist[PS_MAXD];
    for (uint32_t d = threadIdx.x; d <= dmask; d += PS_TPB) hist[d] = 0;
    __syncthreads();
    const uint64_t tile = blockIdx.x;
    const uint64_t base = tile * PS_TILE;
    for (int i = 0; i < PS_E; i++) {
        const uint64_t x = base + (uint64_t)i * PS_TPB + threadIdx.x;
        if (x < n) atomicAdd(&hist[(keys[x] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= dmask; d += PS_TPB) cnt[(uint64_t)d * ntiles + tile] = hist[d];
}

// ---------------------------------------------------------------- stable scatter of one tile

struct ScatterLds {
    uint32_t keyS[PS_TILE];
    uint64_t valS[PS_TILE];
    uint16_t hist[PS_NW][PS_MAXD];      // per wave: running count (<= 4096), then the wave's offset inside its digit
    uint32_t dstart[PS_MAXD];           // first slot of a digit in the sorted tile
    uint32_t gbase[PS_MAXD];            // where the tile's run of a digit starts in the output
    uint32_t scan_tmp[PS_NW + 1];
};

// Records arrive as PS_E groups per wave; group i of wave w holds tile-local records w*512 + i*64 + lane, so "earlier
// wave, then earlier group, then lower lane" is the input order.  Ranks: lanes of a group with the same digit find
// each other with `dbits` ballots; the group's lowest such lane bumps the wave's counter for that digit.
template <bool GEN>
__device__ __forceinline__ void ps_scatter_tile(ScatterLds &s, const uint32_t (&key)[PS_E], const uint64_t (&val)[PS_E], uint64_t nvalid,
                                                uint32_t shift, uint32_t dbits, uint64_t ntiles, uint64_t tile,
                                                const uint32_t *__restrict__ base, uint32_t *__restrict__ keys_out,
                                                uint64_t *__restrict__ vals_out) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t D = 1u << dbits, dmask = D - 1u;
    for (uint32_t x = threadIdx.x; x < PS_NW * PS_MAXD; x += PS_TPB) (&s.hist[0][0])[x] = (uint16_t)0;
    for (uint32_t d = threadIdx.x; d < D; d += PS_TPB) s.gbase[d] = base[(uint64_t)d * ntiles + tile];
    __syncthreads();
    uint32_t rank[PS_E];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < PS_E; i++) {
        const bool valid = (uint64_t)wv * PS_WSPAN + (uint64_t)i * 64 + lane < nvalid;
        const uint32_t d = (key[i] >> shift) & dmask;
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
        for (uint32_t w = 0; w < PS_NW; w++) {
            const uint32_t t = s.hist[w][d];
            s.hist[w][d] = (uint16_t)tot;
            tot += t;
        }
    }
    uint32_t all;
    const uint32_t ex = psc_block_scan(tot, s.scan_tmp, &all);
    if (threadIdx.x < D) s.dstart[threadIdx.x] = ex;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PS_E; i++) {
        const bool valid = (uint64_t)wv * PS_WSPAN + (uint64_t)i * 64 + lane < nvalid;
        if (valid) {
            const uint32_t d = (key[i] >> shift) & dmask;
            const uint32_t slot = s.dstart[d] + s.hist[wv][d] + rank[i];
            s.keyS[slot] = key[i];
            s.valS[slot] = val[i];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < (uint32_t)nvalid; j += PS_TPB) {
        const uint32_t k = s.keyS[j];
        const uint32_t d = (k >> shift) & dmask;
        const uint64_t dest = (uint64_t)s.gbase[d] + (j - s.dstart[d]);
        keys_out[dest] = k;
        vals_out[dest] = s.valS[j];
    }
}

// pass 1: records from the text, scattered by their first digit
__global__ void __launch_bounds__(PS_TPB)
k_ps_scatter_gen(const GenArgs g, uint32_t shift, uint32_t dbits, uint64_t ntiles, const uint32_t *__restrict__ base,
                 uint32_t *__restrict__ keys_out, uint64_t *__restrict__ vals_out) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t txt[PS_NW][PS_TXT_WORDS];
    __shared__ ScatterLds s;
    hash_lut_init(lut);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x;
    const uint64_t t0 = tile * PS_TILE;
    uint32_t key[PS_E];
    uint64_t val[PS_E];
#pragma unroll
    for (int i = 0; i < PS_E; i++) ps_gen64(g, t0 + (uint64_t)wv * PS_WSPAN + (uint64_t)i * 64, lane, txt[wv], lut, &key[i], &val[i]);
    const uint64_t nvalid = min((uint64_t)PS_TILE, g.npos - t0);
    ps_scatter_tile<true>(s, key, val, nvalid, shift, dbits, ntiles, tile, base, keys_out, vals_out);
}

// pass 2: stored records, scattered by their second digit
__global__ void __launch_bounds__(PS_TPB)
k_ps_scatter_keys(const uint32_t *__restrict__ keys_in, const uint64_t *__restrict__ vals_in, uint64_t n, uint32_t shift,
                  uint32_t dbits, uint64_t ntiles, const uint32_t *__restrict__ base, uint32_t *__restrict__ keys_out,
                  uint64_t *__restrict__ vals_out) {
    __shared__ ScatterLds s;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x;
    const uint64_t t0 = tile * PS_TILE;
    uint32_t key[PS_E];
    uint64_t val[PS_E];
#pragma unroll
    for (int i = 0; i < PS_E; i++) {
        const uint64_t x = t0 + (uint64_t)wv * PS_WSPAN + (uint64_t)i * 64 + lane;
        key[i] = x < n ? keys_in[x] : 0u;
        val[i] = x < n ? vals_in[x] : 0ull;
    }
    const uint64_t nvalid = min((uint64_t)PS_TILE, n - t0);
    ps_scatter_tile<false>(s, key, val, nvalid, shift, dbits, ntiles, tile, base, keys_out, vals_out);
}

// ---------------------------------------------------------------- partition bounds

// pstart[p] = index of the first record of partition p (records are sorted by partition); untouched for empty ones
__global__ void __launch_bounds__(256) k_ps_bounds(const uint32_t *__restrict__ keys, uint64_t n, uint32_t cb, uint32_t *__restrict__ pstart) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = keys[i] >> cb;
        if (i == 0 || (keys[i - 1] >> cb) != p) pstart[p] = (uint32_t)i;
    }
}

// an empty partition starts where the next non-empty one does; pstart2[np] = n
__global__ void __launch_bounds__(256) k_ps_bounds_fill(const uint32_t *__restrict__ pstart, uint32_t np, uint32_t n, uint32_t *__restrict__ pstart2) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= np; p += gridDim.x * blockDim.x) {
        uint32_t q = p;
        while (q < np && pstart[q] == 0xFFFFFFFFu) q++;
        pstart2[p] = q < np ? pstart[q] : n;
    }
}

// ---------------------------------------------------------------- the last bucket bits, one block per partition

// A partition = the records of 2^cb consecutive buckets (cb = 13 unless the table is larger than 2^29: the scatter
// passes / the library sort cover at most 16 bits), contiguous after the top bits are sorted, in position order.
// One block turns it into its slice of ent[] and its 2^cb bucket heads, 4096 buckets (a "round") at a time:
//   1. count per bucket (LDS atomics; order is irrelevant for counting);
//   2. kept = min(count, 13); exclusive scan -> where a bucket's entries go;
//   3. placement: a record of a bucket with <= 13 entries takes the next free slot of its bucket (any order: step 4
//      sorts); an over-full bucket keeps its 13 smallest entries -- the serial build's `cumm[h] <= 12` cap
//      (CopMEMMatcher.cpp:156-159) -- found by 13 rounds of atomicMin, so the result does not depend on the order in
//      which the records arrive (the one-sweep front end of idxsweep.hip does not keep them in position order);
//   4. every bucket with >= 2 entries puts its <= 13 entries in ascending order (entry order = position order); the
//      heads are built and streamed out as one contiguous run, the entries as another.
// Fast kernel: the partition's records sit in registers (one load burst), entries are staged in LDS, nothing is
// read back from HBM.  Partitions it cannot take (more records than its registers hold, more kept entries than its
// staging area: repeats and low-complexity text) are flagged and finished by the general kernel below.
// TPB threads hold E records each; SUB = 2^SUBBITS buckets per round, CAP staged entries per round
#define PFF_PAD(b) ((b) + ((b) >> 4))    // one pad word per 16: a thread's 8 consecutive words never share banks with its neighbours'
// packed bucket word: bits 0-12 first slot in the staging area, 13-16 kept (14 = more than 13), 17-29 first slot among
// the round's entries that go to ent[]
#define PFF_OFF(w) ((w) & 0x1FFFu)
#define PFF_KEPT(w) (((w) >> 13) & 15u)
#define PFF_XOFF(w) ((w) >> 17)

// PACKED: the records are single 64-bit words (idxsweep.hip): bucket bits below cb | sampled position index t | fingerprint,
// fmt.sh = bits of (t, fingerprint); `vals` is that array, `keys` is unused.
struct PsRecFmt {
    uint32_t sh, k1;
};
__device__ __forceinline__ void ps_unpack(const PsRecFmt f, uint64_t rec, uint32_t *k, uint64_t *v) {
    const uint64_t tv = rec & ((1ull << f.sh) - 1ull);
    *k = (uint32_t)(rec >> f.sh);
    *v = (((tv >> PGRC_FP_BITS) * (uint64_t)f.k1) << PGRC_FP_BITS) | (tv & ((1ull << PGRC_FP_BITS) - 1ull));
}

// Measured (C3, profiles/r03_index_pmc_before.txt): with two rounds of 4096 buckets the kernel issued ~1600 VALU and ~900
// scalar instructions per wave and partition -- every round walks all of a thread's records and 64-bit entry arithmetic was
// redone at each use -- and the SIMDs were busy with them for more than half of its 5 ms; memory was not the limit (storing
// nothing saves 0.6 ms, prefetching the next partition's records into registers nothing).  Hence: ONE round over the
// partition's 2^cb buckets (the counters and a staging area for 6144 entries fit in 84 KB: the register count allows one
// block per CU anyway), bucket and entry of every record computed once, and a persistent grid.
// Round 5 (profiles/r05_index_variants.txt, r05_ubench_headwrite.txt): the kernel is at the ceiling of its WRITE PATTERN, not of
// its instructions.  A strand's build owns one 64-byte half of every 128-byte line of the pair table; writing 8.6 GB that way
// while 3 GB stream in takes 3.1-3.3 ms in a kernel that does nothing else (this one: 3.4).  Measured and not kept: partitions of
// 4096 buckets finished by 512-thread blocks, three per CU (3.5 ms; and pass 1 pays 0.4 ms for its ninth digit bit); the same
// with the records kept packed in registers (80 VGPRs: 3.5 ms; the 8192-bucket form of it: 4.1 ms); a kernel that keeps the
// partition's heads as an IMAGE in LDS which the records write themselves into (half the instructions: 3.2 ms alone, but 80 KB
// of LDS x 2 blocks leave no room for the other strand's kernels: the pair of builds got slower, 18.7 against 16.6 ms).
template <int E, int PFF_TPB, int PFF_SUBBITS, int CAPI, bool PACKED>
__global__ void __launch_bounds__(PFF_TPB)
k_ps_finish_fast(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, const uint32_t *__restrict__ pstart, uint32_t cb, uint32_t np,
                 uint64_t *__restrict__ ent, ulonglong2 *__restrict__ head, uint32_t hsh, uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count, const PsRecFmt fmt) {
    constexpr uint32_t PFF_SUB = 1u << PFF_SUBBITS, PFF_CAP = (uint32_t)CAPI, BPT = PFF_SUB / PFF_TPB;
    static_assert(CAPI <= 8191 && (E == 8 || E == 16) && PFF_SUBBITS <= 13, "staging slots are 13-bit; ranks are packed 8 per register");
    __shared__ uint32_t pk[PFF_SUB + PFF_SUB / 16];
    __shared__ uint64_t entS[PFF_CAP];
    __shared__ uint32_t scan_tmp[PFF_TPB / 64 + 1];
    __shared__ uint32_t flags[2];            // [0] a bucket overflows, [1] bail out
    __shared__ uint16_t big[PFF_SUB];        // the buckets with three or more entries (any order)
    __shared__ uint32_t nbig;
    const uint32_t nb = 1u << cb, cbmask = nb - 1u;          // (nb <= PFF_SUB: the launcher sees to it)
    // (Fetching the NEXT partition's records ahead -- into a second set of registers at the top, or late, while the heads
    //  are written -- was tried twice and lost both times: 127-128 registers, and the wait for the fetched records also
    //  waits for the head stores issued after them.  3.37 -> 3.85 ms at C3.)
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const uint64_t s = pstart[p], e = pstart[p + 1];
        if (e - s > (uint64_t)E * PFF_TPB) {                  // more records than the registers hold: the general kernel's
            if (threadIdx.x == 0) todo[atomicAdd(todo_count, 1u)] = p;      // (the list of the general kernel, any order)
            continue;
        }
        // the partition's records: bucket (all ones: no record) and entry, in registers
        uint32_t k[E];
        uint64_t v[E];
#pragma unroll
        for (int i = 0; i < E; i++) {
            const uint64_t x = s + (uint64_t)i * PFF_TPB + threadIdx.x;
            if (PACKED) {
                ps_unpack(fmt, x < e ? vals[x] : 0ull, &k[i], &v[i]);
            } else {
                k[i] = x < e ? keys[x] & cbmask : 0u;
                v[i] = x < e ? vals[x] : 0ull;
            }
            if (x >= e) k[i] = 0xFFFFFFFFu;
        }
        for (uint32_t b = threadIdx.x; b < PFF_SUB + PFF_SUB / 16; b += PFF_TPB) pk[b] = 0;
        if (threadIdx.x < 2) flags[threadIdx.x] = 0;
        if (threadIdx.x == 2) nbig = 0;
        __syncthreads();
        // 1. count; the count a record finds is its rank in its bucket (4 bits per record: only ranks below 13 are used)
        uint32_t ranks = 0, ranks_lo = 0;
#pragma unroll
        for (int i = 0; i < E; i++) {
            if (k[i] != 0xFFFFFFFFu) ranks |= min(atomicAdd(&pk[PFF_PAD(k[i])], 1u), 15u) << (4 * (i & 7));
            if (E > 8 && i == 7) { ranks_lo = ranks; ranks = 0; }
        }
        __syncthreads();
        // 2. scan of the kept counts (low half) and of the entries that go to ent[] (high half): BPT consecutive buckets per thread
        uint32_t total, xtotal;
        {
            const uint32_t b0 = threadIdx.x * BPT;
            uint32_t c[BPT], sum = 0;
            bool ovf = false;
#pragma unroll
            for (uint32_t q = 0; q < BPT; q++) {
                c[q] = pk[PFF_PAD(b0 + q)];
                ovf |= c[q] > PGRC_BUCKET_CAP;
                const uint32_t kc = min(c[q], PGRC_BUCKET_CAP);
                sum += kc | ((kc > 2u ? kc - 1u : 0u) << 16);
            }
            uint32_t both;
            uint32_t off = psc_block_scan(sum, scan_tmp, &both);
            total = both & 0xFFFFu;
            xtotal = both >> 16;
#pragma unroll
            for (uint32_t q = 0; q < BPT; q++) {
                const uint32_t kc = min(c[q], PGRC_BUCKET_CAP), o = off & 0xFFFFu;
                pk[PFF_PAD(b0 + q)] = (o & 0x1FFFu) | (min(c[q], 14u) << 13) | ((off >> 16) << 17);
                if (c[q] > PGRC_BUCKET_CAP && total <= PFF_CAP)               // an over-full bucket: its slots start as "no entry yet" (step 3)
                    for (uint32_t j = 0; j < PGRC_BUCKET_CAP; j++) entS[o + j] = ~0ull;
                if (kc > 2u) big[atomicAdd(&nbig, 1u)] = (uint16_t)(b0 + q);
                off += kc | ((kc > 2u ? kc - 1u : 0u) << 16);
            }
            if (ovf) flags[0] = 1;
            if (total > PFF_CAP && threadIdx.x == 0) flags[1] = 1;
        }
        __syncthreads();
        if (flags[1]) {                                       // more entries than the staging area holds: the general kernel redoes the partition
            if (threadIdx.x == 0) todo[atomicAdd(todo_count, 1u)] = p;
            __syncthreads();                                  // (flags[] is rewritten for the next partition)
            continue;
        }
        // 3. placement into the staging area: slot = first slot of the bucket + rank.  An over-full bucket keeps its 13
        //    SMALLEST entries (= positions: the serial build's `cumm[h] <= 12` cap, CopMEMMatcher.cpp:156-159) whatever
        //    order the records arrive in: 13 rounds of "the smallest entry not taken yet" (LDS atomicMin on slot r), which
        //    also leaves them in ascending order.
        uint32_t live = 0;
#pragma unroll
        for (int i = 0; i < E; i++) {
            if (k[i] != 0xFFFFFFFFu) {
                const uint32_t w = pk[PFF_PAD(k[i])];
                const uint32_t r = ((E > 8 && i < 8 ? ranks_lo : ranks) >> (4 * (i & 7))) & 15u;
                if (PFF_KEPT(w) <= PGRC_BUCKET_CAP) entS[PFF_OFF(w) + r] = v[i];
                else live |= 1u << i;
            }
        }
        if (flags[0]) {
            __syncthreads();
#pragma unroll 1
            for (uint32_t r = 0; r < PGRC_BUCKET_CAP; r++) {
#pragma unroll
                for (int i = 0; i < E; i++)
                    if ((live >> i) & 1u) atomicMin((unsigned long long *)&entS[PFF_OFF(pk[PFF_PAD(k[i])]) + r], (unsigned long long)v[i]);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < E; i++)
                    if (((live >> i) & 1u) && entS[PFF_OFF(pk[PFF_PAD(k[i])]) + r] == v[i]) live &= ~(1u << i);
                __syncthreads();
            }
        }
        __syncthreads();
        // 4. buckets with three or more entries (3 % of the buckets; one thread each, from the list, so that the lanes of a
        //    wave all have such a bucket -- done inside the loop over ALL buckets, nearly every wave walked its one slow
        //    lane's loops): order the entries (over-full buckets are in order already, step 3), entries 1.. go to ent[]
        //    (a head holds entry 0, and entry 1 of a bucket of two: nothing else is ever read from ent[])
        for (uint32_t x = threadIdx.x; x < nbig; x += PFF_TPB) {
            const uint32_t w = pk[PFF_PAD((uint32_t)big[x])];
            const uint32_t c = min(PFF_KEPT(w), PGRC_BUCKET_CAP), o = PFF_OFF(w);
            if (PFF_KEPT(w) <= PGRC_BUCKET_CAP) {
                for (uint32_t i = 1; i < c; i++) {          // insertion sort of <= 13 values
                    const uint64_t y = entS[o + i];
                    uint32_t j = i;
                    while (j > 0 && entS[o + j - 1] > y) { entS[o + j] = entS[o + j - 1]; j--; }
                    entS[o + j] = y;
                }
            }
            const uint64_t base = s + PFF_XOFF(w);
            for (uint32_t j = 1; j < c; j++) ent[base + j - 1] = entS[o + j];
        }
        __syncthreads();
        // 5. heads (one contiguous run)
        for (uint32_t b = threadIdx.x; b < nb; b += PFF_TPB) {
            const uint32_t w = pk[PFF_PAD(b)];
            const uint32_t c = min(PFF_KEPT(w), PGRC_BUCKET_CAP), o = PFF_OFF(w);
            const uint64_t e0 = c ? entS[o] : HEAD_EMPTY, e1 = c > 1 ? entS[o + 1] : HEAD_EMPTY;     // (slot o < PFF_CAP also for an empty last bucket)
            ulonglong2 hd;
            if (c <= 2) hd = make_ulonglong2(min(e0, e1), c == 2 ? max(e0, e1) : HEAD_EMPTY);      // (c == 0: both HEAD_EMPTY = all ones)
            else hd = make_ulonglong2(e0 | HEAD_OVF, (s + PFF_XOFF(w)) | ((uint64_t)c << 56));     // entries 1.. at ent[base + j - 1]
            head[head_slot(((uint64_t)p << cb) + b, hsh)] = hd;
        }
        __syncthreads();                                      // (the next partition reuses counters and staging area)
        (void)xtotal;
    }
}

// The general kernel: any partition size, entries placed straight into ent[] and read back for the heads.  Runs only
// for the partitions the fast kernel flagged (`todo` = their list; nullptr: every partition), a persistent grid that
// walks the list.  8192 buckets per round.  Three sweeps over the partition's records: count; place (buckets with at
// most 13 records take any free slot); and, for over-full buckets, the selection of their 13 smallest entries:
//   * the FIRST 13 records of such a bucket to arrive (any 13 will do) leave the largest of their position keys in
//     tmax[bucket]: a record whose key is larger has 13 smaller ones before it and is out;
//   * the remaining candidates (normally the bucket's first 13 records and a few neighbours: records arrive roughly in
//     position order) are listed by index, and 13 rounds of "the smallest entry above the last round's" (atomicMin on
//     slot r) over that short list fill the bucket's slots in ascending order.
// Nothing depends on the order in which the records arrive.
#define PF_TPB 512
#define PF_NB 8192u
#define PF_LIST 8192u

template <bool PACKED>
__global__ void __launch_bounds__(PF_TPB)
k_ps_finish(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, const uint32_t *__restrict__ pstart, uint32_t cb,
            uint64_t *__restrict__ ent, ulonglong2 *__restrict__ head, uint32_t hsh, const uint32_t *__restrict__ todo, const uint32_t *__restrict__ todo_count,
            uint32_t np, const PsRecFmt fmt) {
    __shared__ uint32_t cnt[PF_NB];          // 1: count; from 2 on: first slot of the bucket (relative to this round's base)
    __shared__ uint8_t kept[PF_NB];          // min(count, 14): 14 = "more than 13"
    __shared__ uint32_t arrived[PF_NB];      // records of the bucket seen by the placement sweep
    __shared__ uint32_t tmax[PF_NB];         // over-full buckets: largest position key among the first 13 arrivals
    __shared__ uint32_t list[PF_LIST];       // candidates of the over-full buckets (record index in the partition)
    __shared__ uint32_t scan_tmp[PF_TPB / 64 + 1];
    __shared__ uint32_t any_ovf, ncand;
    const uint32_t nlist = todo ? *todo_count : np;
    const uint32_t nb = 1u << cb, cbmask = nb - 1u;
    auto record = [&](uint64_t x, uint32_t *k, uint64_t *v) {
        if (PACKED) ps_unpack(fmt, vals[x], k, v);
        else { *k = keys[x]; *v = vals[x]; }
    };
    auto poskey = [](uint64_t v) -> uint32_t { return (uint32_t)(v >> (PGRC_FP_BITS + 8u)); };   // position / 256: monotone in v
    // entries other threads of the block placed (plain stores, or atomics that execute in the L2): read past this CU's L1
    auto ld = [](const uint64_t *q) -> uint64_t { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (uint32_t li = blockIdx.x; li < nlist; li += gridDim.x) {
        const uint32_t p = todo ? todo[li] : li;
        const uint64_t s = pstart[p], e = pstart[p + 1];
        uint64_t out = s;
        for (uint32_t r0 = 0; r0 < nb; r0 += PF_NB) {
            __syncthreads();
            for (uint32_t b = threadIdx.x; b < PF_NB; b += PF_TPB) { cnt[b] = 0; arrived[b] = 0; tmax[b] = 0; }
            if (threadIdx.x == 0) { any_ovf = 0; ncand = 0; }
            __syncthreads();
            for (uint64_t x = s + threadIdx.x; x < e; x += PF_TPB) {
                uint32_t k;
                uint64_t v;
                record(x, &k, &v);
                const uint32_t b = (k & cbmask) - r0;
                if (b < PF_NB) atomicAdd(&cnt[b], 1u);
            }
            __syncthreads();
            uint32_t total;
            {
                const uint32_t b0 = threadIdx.x * (PF_NB / PF_TPB);
                uint32_t sum = 0;
                for (uint32_t q = 0; q < PF_NB / PF_TPB; q++) {
                    const uint32_t c = cnt[b0 + q];
                    kept[b0 + q] = (uint8_t)min(c, 14u);
                    sum += min(c, PGRC_BUCKET_CAP);
                }
                uint32_t off = psc_block_scan(sum, scan_tmp, &total);
                for (uint32_t q = 0; q < PF_NB / PF_TPB; q++) {
                    cnt[b0 + q] = off;
                    if (kept[b0 + q] > PGRC_BUCKET_CAP) {      // an over-full bucket: its 13 slots start as "no entry yet"
                        for (uint32_t j = 0; j < PGRC_BUCKET_CAP; j++) ent[out + off + j] = ~0ull;
                        any_ovf = 1;
                    }
                    off += min((uint32_t)kept[b0 + q], PGRC_BUCKET_CAP);
                }
            }
            __threadfence();
            __syncthreads();
            for (uint64_t x = s + threadIdx.x; x < e; x += PF_TPB) {
                uint32_t k;
                uint64_t v;
                record(x, &k, &v);
                const uint32_t b = (k & cbmask) - r0;
                if (b < PF_NB) {
                    const uint32_t r = atomicAdd(&arrived[b], 1u);
                    if (kept[b] <= PGRC_BUCKET_CAP) ent[out + cnt[b] + r] = v;
                    else if (r < PGRC_BUCKET_CAP) atomicMax(&tmax[b], poskey(v));
                }
            }
            __syncthreads();
            if (any_ovf) {
                for (uint64_t x = s + threadIdx.x; x < e; x += PF_TPB) {
                    uint32_t k;
                    uint64_t v;
                    record(x, &k, &v);
                    const uint32_t b = (k & cbmask) - r0;
                    if (b < PF_NB && kept[b] > PGRC_BUCKET_CAP && poskey(v) <= tmax[b]) {
                        const uint32_t i = atomicAdd(&ncand, 1u);
                        if (i < PF_LIST) list[i] = (uint32_t)(x - s);
                    }
                }
                __syncthreads();
                const bool listed = ncand <= PF_LIST;        // (else: sweep the whole partition in every round)
                const uint64_t m = listed ? ncand : e - s;
                for (uint32_t r = 0; r < PGRC_BUCKET_CAP; r++) {
                    for (uint64_t i = threadIdx.x; i < m; i += PF_TPB) {
                        const uint64_t x = s + (listed ? (uint64_t)list[i] : i);
                        uint32_t k;
                        uint64_t v;
                        record(x, &k, &v);
                        const uint32_t b = (k & cbmask) - r0;
                        if (b < PF_NB && kept[b] > PGRC_BUCKET_CAP && poskey(v) <= tmax[b]) {
                            uint64_t *slot = ent + out + cnt[b] + r;
                            if (r == 0 || v > ld(slot - 1)) atomicMin((unsigned long long *)slot, (unsigned long long)v);
                        }
                    }
                    __threadfence();
                    __syncthreads();
                }
            }
            __threadfence();
            __syncthreads();
            for (uint32_t b = threadIdx.x; b < PF_NB && r0 + b < nb; b += PF_TPB) {
                const uint32_t c = min((uint32_t)kept[b], PGRC_BUCKET_CAP);
                ulonglong2 hd = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
                if (c) {
                    uint64_t *q = ent + out + cnt[b];
                    uint64_t v0 = ld(q), v1 = c > 1 ? ld(q + 1) : 0;
                    if (c == 2) {
                        if (v1 < v0) { const uint64_t t = v0; v0 = v1; v1 = t; q[0] = v0; q[1] = v1; }
                    } else if (c > 2) {
                        uint64_t w[PGRC_BUCKET_CAP];
                        for (uint32_t i = 0; i < c; i++) w[i] = ld(q + i);
                        for (uint32_t i = 1; i < c; i++) {      // insertion sort of <= 13 values
                            const uint64_t x = w[i];
                            uint32_t j = i;
                            while (j > 0 && w[j - 1] > x) { w[j] = w[j - 1]; j--; }
                            w[j] = x;
                        }
                        for (uint32_t i = 0; i < c; i++) q[i] = w[i];
                        v0 = w[0];
                    }
                    hd.x = v0;
                    if (c == 2) hd.y = v1;
                    else if (c > 2) {
                        hd.x |= HEAD_OVF;
                        hd.y = (out + cnt[b] + 1) | ((uint64_t)c << 56);    // entries 1.. at ent[base + j - 1]
                    }
                }
                head[head_slot(((uint64_t)p << cb) + r0 + b, hsh)] = hd;
            }
            out += total;
        }
    }
}

// ---------------------------------------------------------------- drivers

uint32_t pgrc_ps_partition_bits(uint32_t hbits) { return hbits > PS_CB + 16u ? hbits - 16u : PS_CB; }

bool pgrc_ps_applicable(const pgrc_match_ctx *c, uint32_t hbits) {
    return hbits >= PS_CB + 2 && hbits <= 31 && c->npos > 0 && c->npos < 0xFFFFF000ull && c->cp.k1 <= 16 && c->cp.K <= 56;
}

// The hand-written front end: records straight from the text, two stable scatter passes over the bucket bits
// [cb, hbits).  Output in c->d_skey[1] / c->d_sval[1] (sorted by partition, position order inside); c->d_sval[0] is free.
int pgrc_ps_scatter_front(pgrc_match_ctx *c, int strand, uint32_t hbits, uint32_t cb) {
    const uint32_t K = (uint32_t)c->cp.K, k1 = (uint32_t)c->cp.k1;
    const uint64_t hs = c->cp.hash_size, n = c->npos;
    PsPlan pl;
    pl.hbits = hbits;
    pl.b1 = (hbits - cb) / 2;
    pl.b2 = hbits - cb - pl.b1;
    pl.n = n;
    pl.ntiles = (n + PS_TILE - 1) / PS_TILE;
    const uint32_t D = 1u << pl.b2;                                       // b2 >= b1
    int e;
    for (int k = 0; k < 2; k++)
        if ((e = pgrc_buf_ensure(c, c->d_skey[k], (n + 16) * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, c->d_sval[k], (n + 16) * sizeof(uint64_t)))) return e;
    const uint64_t ncnt = (uint64_t)D * pl.ntiles;
    const uint64_t nbs = (ncnt + PSC_EPB - 1) / PSC_EPB + 1;
    if ((e = pgrc_buf_ensure(c, c->d_sorttmp, (ncnt + nbs) * sizeof(uint32_t) + 256))) return e;
    uint32_t *cnt = (uint32_t *)c->d_sorttmp.p, *bsum = cnt + ncnt;
    uint32_t *kA = (uint32_t *)c->d_skey[0].p, *kB = (uint32_t *)c->d_skey[1].p;
    uint64_t *vA = (uint64_t *)c->d_sval[0].p, *vB = (uint64_t *)c->d_sval[1].p;
    GenArgs g;
    g.pg = (const uint32_t *)c->pg2[strand].p;
    g.pg_words_alloc = c->pg_words + PGRC_PG_PAD_WORDS;
    g.npos = n;
    g.k1 = k1;
    g.K = K;
    g.mask = (uint32_t)(hs - 1);
    const uint32_t grid = (uint32_t)pl.ntiles;
    const uint32_t sh1 = cb, sh2 = cb + pl.b1;
    // pass 1: text -> A; pass 2: A -> B
    hipLaunchKernelGGL(k_ps_hist_gen, dim3(grid), dim3(PS_TPB), 0, c->stream, g, sh1, (1u << pl.b1) - 1u, pl.ntiles, cnt);
    if ((e = ps_scan(c, cnt, ((uint64_t)1 << pl.b1) * pl.ntiles, bsum))) return e;
    hipLaunchKernelGGL(k_ps_scatter_gen, dim3(grid), dim3(PS_TPB), 0, c->stream, g, sh1, pl.b1, pl.ntiles, (const uint32_t *)cnt, kA, vA);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(k_ps_hist_keys, dim3(grid), dim3(PS_TPB), 0, c->stream, (const uint32_t *)kA, n, sh2, (1u << pl.b2) - 1u, pl.ntiles, cnt);
    if ((e = ps_scan(c, cnt, ncnt, bsum))) return e;
    hipLaunchKernelGGL(k_ps_scatter_keys, dim3(grid), dim3(PS_TPB), 0, c->stream, (const uint32_t *)kA, (const uint64_t *)vA, n, sh2, pl.b2,
                       pl.ntiles, (const uint32_t *)cnt, kB, vB);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// The finish of every partition.  pst2[0 .. np] = partition starts (device); slow = np flags + np + 1 words for the list
// of flagged partitions (all zero).  packed_sh != 0: d_vals holds the 64-bit packed records of idxsweep.hip (d_keys
// unused), packed_sh = bits of (t, fingerprint) in a record.
static int ps_launch_finish(pgrc_match_ctx *c, const uint32_t *d_keys, const uint64_t *d_vals, const uint32_t *pst2, uint32_t *slow, uint32_t np,
                            uint32_t cb, uint64_t *d_ent, uint32_t packed_sh) {
    const uint64_t n = c->npos;
    PsRecFmt fmt;
    fmt.sh = packed_sh;
    fmt.k1 = (uint32_t)c->cp.k1;
    const bool packed = packed_sh != 0;
    // the partitions the fast kernel cannot take go straight to the general kernel's list (round 5: no flags + compaction kernel in
    // between -- a 256-block launch that, beside the other strand's build, waited 1.4 ms on average for its turn)
    uint32_t *todo = slow + np, *todo_count = todo + np;
    ulonglong2 *head = c->head_ptr;
    const uint32_t hsh = c->head_sh;
    const uint32_t ggrid = std::min<uint32_t>(np, (uint32_t)c->num_cus * 2u);
    // PGRC_INDEX_FINISH=general: the general finish kernel for every partition (tests)
    // (the fast kernel takes a partition's 2^cb buckets in one round of at most 8192: the stable front end's partitions of
    //  tables beyond 2^29 buckets are larger and all go to the general kernel)
    if (c->opt.index_finish_general || cb > 13u) {
        if (packed) hipLaunchKernelGGL(k_ps_finish<true>, dim3(ggrid), dim3(PF_TPB), 0, c->stream, d_keys, d_vals, pst2, cb, d_ent, head, hsh, (const uint32_t *)nullptr, (const uint32_t *)nullptr, np, fmt);
        else hipLaunchKernelGGL(k_ps_finish<false>, dim3(ggrid), dim3(PF_TPB), 0, c->stream, d_keys, d_vals, pst2, cb, d_ent, head, hsh, (const uint32_t *)nullptr, (const uint32_t *)nullptr, np, fmt);
    } else {
        // registers per thread sized for the mean partition (uniform hash values); whatever is larger is flagged
        const uint64_t need = n / np + n / np / 4 + 512;
        const uint32_t fgrid = std::min<uint32_t>(np, (uint32_t)c->num_cus);     // persistent: one block of 16 waves per CU
        // block shape: 1024 threads x 8 records (16 for larger partitions: tables beyond 2^29 buckets), two 4096-bucket
        // rounds.  Measured at C3 (index build per strand, tools/ab_finish_cfg.sh in the round-2 history): 512 x 16: 15.1 ms,
        // 256 x 32: 16.0, 256 x 32 with 2048-bucket rounds: 16.8, 512 x 16 with 2048-bucket rounds: 16.4, 1024 x 8: 13.5.
#define PFF_LAUNCH(E, TPB, SB, CAP, PK)                                                                                   \
        hipLaunchKernelGGL((k_ps_finish_fast<E, TPB, SB, CAP, PK>), dim3(fgrid), dim3(TPB), 0, c->stream, d_keys, d_vals,  \
                           pst2, cb, np, d_ent, head, hsh, todo, todo_count, fmt)
        // (one round of 2^cb <= 8192 buckets; staging area for 6144 entries -- the mean partition holds 0.7 * 8192 -- resp.
        //  8191 where partitions are larger: tables beyond 2^29 buckets)
        if (packed) {
            if (need <= 8192) PFF_LAUNCH(8, 1024, 13, 6144, true);
            else PFF_LAUNCH(16, 1024, 13, 8191, true);
        } else {
            if (need <= 8192) PFF_LAUNCH(8, 1024, 13, 6144, false);
            else PFF_LAUNCH(16, 1024, 13, 8191, false);
        }
#undef PFF_LAUNCH
        if (packed) hipLaunchKernelGGL(k_ps_finish<true>, dim3(ggrid), dim3(PF_TPB), 0, c->stream, d_keys, d_vals, pst2, cb, d_ent, head, hsh, (const uint32_t *)todo, (const uint32_t *)todo_count, np, fmt);
        else hipLaunchKernelGGL(k_ps_finish<false>, dim3(ggrid), dim3(PF_TPB), 0, c->stream, d_keys, d_vals, pst2, cb, d_ent, head, hsh, (const uint32_t *)todo, (const uint32_t *)todo_count, np, fmt);
    }
    HIP_TRY(c, hipGetLastError());
    c->ent_ptr = d_ent;
    return PGRC_OK;
}

// records sorted by their bucket bits [cb, hbits) -> ent[] (d_ent) and all bucket heads
int pgrc_ps_finish(pgrc_match_ctx *c, const uint32_t *d_keys, const uint64_t *d_vals, uint32_t hbits, uint32_t cb, uint64_t *d_ent) {
    const uint64_t n = c->npos;
    const uint32_t np = 1u << (hbits - cb);
    int e;
    if ((e = pgrc_buf_ensure(c, c->d_sorttmp, (4ull * (np + 2)) * sizeof(uint32_t) + 256))) return e;
    uint32_t *pst = (uint32_t *)c->d_sorttmp.p, *pst2 = pst + np + 2, *slow = pst2 + np + 2;
    HIP_TRY(c, hipMemsetAsync(pst, 0xFF, (size_t)(np + 1) * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(slow, 0, (size_t)(2 * np + 1) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_ps_bounds, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536ull * 4)), dim3(256), 0, c->stream, d_keys, n, cb, pst);
    hipLaunchKernelGGL(k_ps_bounds_fill, dim3((np + 256) / 256), dim3(256), 0, c->stream, (const uint32_t *)pst, np, (uint32_t)n, pst2);
    return ps_launch_finish(c, d_keys, d_vals, (const uint32_t *)pst2, slow, np, cb, d_ent, 0u);
}

// the same behind the one-sweep front end (idxsweep.hip): packed records, partition starts already known; d_slow = 2 np + 1
// zeroed words
int pgrc_ps_finish_packed(pgrc_match_ctx *c, const uint64_t *d_recs, const uint32_t *d_pstart, uint32_t *d_slow, uint32_t np, uint32_t cb,
                          uint32_t rec_sh, uint64_t *d_ent) {
    return ps_launch_finish(c, nullptr, d_recs, d_pstart, d_slow, np, cb, d_ent, rec_sh);
}
