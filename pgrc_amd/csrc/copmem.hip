// copmem.hip -- mode 'c' (the reference's default): copMEM seed index over the pseudogenome
// and the per-read probe + Hamming verification, as HIP kernels for gfx950.
//
// Reference behaviour restated (not translated):
//   index  : CopMEMMatcher::processRef / genCumm, matching/copmem/CopMEMMatcher.cpp:140-231
//            (serial semantics: per bucket the first 13 sampled positions in ascending order)
//   hash   : maRushPrime1HashSparsified<K>, matching/copmem/Hashes.h:54-76
//   query  : CopMEMMatcher::processApproxMatchQueryTight, CopMEMMatcher.cpp:483-566
//   driver : CopMEMReadsApproxMatcher::executeMatching, matching/ReadsMatchers.cpp:421-451
//
// MI355X design
//   * text and reads live 2-bit packed in HBM; a read is 10 dwords (L=150) held in VGPRs.
//   * index build = count (atomics) -> exclusive scan -> scatter (atomics) -> per-bucket
//     13-smallest selection; the racy order of the scatter is erased by the selection, so the
//     result is the canonical serial index, bit for bit.  Pg windows are staged through LDS in
//     coalesced tiles (adjacent sampled positions overlap by K-k1 symbols).
//   * the index is laid out for 64-B HBM sectors, not as the reference's CSR: a 1-bit-per-bucket
//     occupancy bitmap (64 MiB at 2^29 buckets: served from L2 / Infinity Cache) answers the ~50 %
//     of probes that hit an empty bucket; an 8-B bucket head holds a single-entry bucket inline
//     (70 % of the non-empty ones); every entry carries a 24-bit fingerprint of the symbols the
//     sparsified hash ignores, so a false candidate is rejected -- with exactly the reference's
//     head-reject accounting -- without fetching its text window.
//   * match = one read per lane; the reference's sequential per-read state machine (limit
//     tightening, false-candidate budget, early exit) is kept exactly, so results are
//     bit-identical.  The seed window is kept at the low bits of a shifting copy of the read
//     (v_alignbit), so no register array is ever indexed dynamically.  Hamming distance on
//     2-bit words: xor, fold pair bits, v_bcnt (popcount) under head/tail symbol masks.
//   * all of it is HBM/latency bound integer work: no MFMA.
#include <stdlib.h>

#include "ctx.h"
#include "devutil.h"

// ----------------------------------------------------------------------------- index build

#define IDX_TPB 256
#define IDX_TILE_WORDS (IDX_TPB + 16) // 256 positions * k1(<=16) symbols / 16 + K/16 + slack

// One block walks tiles of IDX_TPB consecutive sampled positions; the tile's text words are
// loaded once, coalesced, into LDS; every thread then hashes its K-symbol window from LDS.
template <bool FILL>
__global__ void __launch_bounds__(IDX_TPB)
k_copmem_index_pass(const uint32_t *__restrict__ pg, uint64_t pg_words_alloc, uint64_t npos, uint32_t k1, uint32_t K,
                    uint32_t mask, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ cumm,
                    uint64_t *__restrict__ ent) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t tile[IDX_TILE_WORDS + 8];
    hash_lut_init(lut);
    const uint64_t ntiles = (npos + IDX_TPB - 1) / IDX_TPB;
    for (uint64_t tIdx = blockIdx.x; tIdx < ntiles; tIdx += gridDim.x) {
        const uint64_t t0 = tIdx * IDX_TPB;
        const uint64_t p0 = t0 * k1;
        const uint64_t w0 = p0 >> 4;
        // symbols needed: [p0, p0 + (IDX_TPB-1)*k1 + K)  (+16 for the funnel's upper word)
        const uint32_t need = (uint32_t)((((p0 & 15) + (uint64_t)(IDX_TPB - 1) * k1 + K + 15) >> 4) + 5);
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < need; w += IDX_TPB) tile[w] = (w0 + w < pg_words_alloc) ? pg[w0 + w] : 0u;
        __syncthreads();
        const uint64_t t = t0 + threadIdx.x;
        if (t < npos) {
            const uint64_t p = t * k1;
            const uint32_t q = (uint32_t)((p >> 4) - w0);
            const uint32_t sh = ((uint32_t)p & 15u) * 2u;
            const uint32_t a0 = tile[q], a1 = tile[q + 1], a2 = tile[q + 2], a3 = tile[q + 3], a4 = tile[q + 4];
            uint32_t fp;
            const uint32_t h = copmem_hash32_fp(funnel_r(a0, a1, sh), funnel_r(a1, a2, sh), funnel_r(a2, a3, sh),
                                                funnel_r(a3, a4, sh), K, lut, &fp) & mask;
            if (!FILL) {
                atomicAdd(&cnt[h], 1u);
            } else {
                const uint32_t slot = cumm[h] + atomicAdd(&cnt[h], 1u);
                ent[slot] = (p << PGRC_FP_BITS) | fp; // position in the high bits: u64 order = position order
            }
        }
    }
}

// ---- exclusive scan of u32 counts (optionally capped at 13), 4096 elements per block ----
#define SCAN_TPB 256
#define SCAN_EPT 16
#define SCAN_EPB (SCAN_TPB * SCAN_EPT)

template <bool CAP>
__device__ __forceinline__ uint32_t scan_val(uint32_t v) {
    return CAP ? (v < PGRC_BUCKET_CAP ? v : PGRC_BUCKET_CAP) : v;
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *smem /*[SCAN_TPB/64+1]*/, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63) smem[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < SCAN_TPB / 64; k++) {
        uint32_t s = smem[k];
        if (k < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

template <bool CAP>
__global__ void __launch_bounds__(SCAN_TPB) k_scan_sums(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t smem[SCAN_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_EPB + (uint64_t)threadIdx.x * SCAN_EPT;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_EPT; k++)
        if (base + k < n) s += scan_val<CAP>(in[base + k]);
    uint32_t tot;
    block_exclusive_scan(s, smem, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums in place; writes the grand total to bsum[nb]
__global__ void __launch_bounds__(1024) k_scan_bsums(uint32_t *bsum, uint64_t nb) {
    __shared__ uint32_t smem[1024 / 64 + 1];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nb; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        uint32_t v = i < nb ? bsum[i] : 0;
        // wave scan + cross-wave
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t u = __shfl_up(inc, o, 64);
            if (lane >= (uint32_t)o) inc += u;
        }
        if (lane == 63) smem[wv] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (uint32_t k = 0; k < 16; k++) {
            uint32_t s = smem[k];
            if (k < wv) woff += s;
            tot += s;
        }
        const uint32_t carry = carry_s;
        if (i < nb) bsum[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry_s;
}

// out[i] = exclusive prefix; out[n] = out[n+1] = total  (cumm has hash_size+2 entries)
template <bool CAP>
__global__ void __launch_bounds__(SCAN_TPB)
k_scan_write(const uint32_t *__restrict__ in, uint64_t n, const uint32_t *__restrict__ bsum, uint64_t nb,
             uint32_t *__restrict__ out) {
    __shared__ uint32_t smem[SCAN_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_EPB + (uint64_t)threadIdx.x * SCAN_EPT;
    uint32_t v[SCAN_EPT];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_EPT; k++) {
        v[k] = (base + k < n) ? scan_val<CAP>(in[base + k]) : 0;
        s += v[k];
    }
    uint32_t tot;
    uint32_t off = block_exclusive_scan(s, smem, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_EPT; k++) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n] = bsum[nb];
        out[n + 1] = bsum[nb];
    }
}

// Per bucket: move its min(count,13) smallest entries (= smallest positions), ascending, to the front
// -- this turns the racy scatter order into the reference's serial order (ascending p, later p
// dropped) -- and emit the bucket head and the occupancy bitmap.
#define ENT_MASK ((1ull << 56) - 1)
#define HEAD_COUNT(w0) ((uint32_t)((w0) >> 56) & 15u)

// 16-B bucket head (a random 16-B gather costs the same as an 8-B one on gfx950: the limit is
// lane-addresses per second, tools/ubench/gather2.hip):
//   w0 = entry0 | count << 56          count = min(n, 13); w0 == 0 <=> empty bucket
//   w1 = entry1                        when count == 2
//      = start of the bucket in ent[]  when count >= 3 (entries 1.. are fetched from there)
__global__ void __launch_bounds__(256)
k_bucket_finalize(const uint32_t *__restrict__ cumm, uint64_t hash_size, uint64_t *__restrict__ ent,
                  ulonglong2 *__restrict__ head) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // grid covers hash_size exactly
    const uint32_t lo = cumm[h], n = cumm[h + 1] - lo;
    ulonglong2 hd = make_ulonglong2(0ull, 0ull);
    if (n == 1) {
        hd.x = ent[lo] | (1ull << 56);
    } else if (n == 2) {
        const uint64_t x = ent[lo], y = ent[lo + 1];
        hd.x = min(x, y) | (2ull << 56);
        hd.y = max(x, y);
        ent[lo] = min(x, y);
        ent[lo + 1] = max(x, y);
    } else if (n > 2) {
        uint64_t a[PGRC_BUCKET_CAP];
#pragma unroll
        for (int k = 0; k < (int)PGRC_BUCKET_CAP; k++) a[k] = ~0ull;
        for (uint32_t j = 0; j < n; j++) {
            uint64_t x = ent[lo + j];
#pragma unroll
            for (int k = 0; k < (int)PGRC_BUCKET_CAP; k++) {
                const uint64_t m = min(a[k], x);
                x = max(a[k], x);
                a[k] = m;
            }
        }
#pragma unroll
        for (int k = 0; k < (int)PGRC_BUCKET_CAP; k++)
            if ((uint32_t)k < n) ent[lo + k] = a[k];
        hd.x = a[0] | ((uint64_t)min(n, PGRC_BUCKET_CAP) << 56);
        hd.y = lo;
    }
    head[h] = hd;
}

static int run_scan(pgrc_match_ctx *c, bool cap, const uint32_t *in, uint64_t n, uint32_t *out) {
    const uint64_t nb = (n + SCAN_EPB - 1) / SCAN_EPB;
    int e = pgrc_buf_ensure(c, c->d_scan_tmp, (nb + 2) * sizeof(uint32_t));
    if (e) return e;
    uint32_t *bsum = (uint32_t *)c->d_scan_tmp.p;
    if (cap) {
        hipLaunchKernelGGL(k_scan_sums<true>, dim3((uint32_t)nb), dim3(SCAN_TPB), 0, c->stream, in, n, bsum);
        hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, c->stream, bsum, nb);
        hipLaunchKernelGGL(k_scan_write<true>, dim3((uint32_t)nb), dim3(SCAN_TPB), 0, c->stream, in, n, bsum, nb, out);
    } else {
        hipLaunchKernelGGL(k_scan_sums<false>, dim3((uint32_t)nb), dim3(SCAN_TPB), 0, c->stream, in, n, bsum);
        hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, c->stream, bsum, nb);
        hipLaunchKernelGGL(k_scan_write<false>, dim3((uint32_t)nb), dim3(SCAN_TPB), 0, c->stream, in, n, bsum, nb, out);
    }
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

int pgrc_copmem_build_index(pgrc_match_ctx *c, int strand) {
    const uint64_t G = c->G;
    const uint32_t K = (uint32_t)c->cp.K, k1 = (uint32_t)c->cp.k1;
    const uint64_t hs = c->cp.hash_size;
    c->npos = (G >= K) ? (G - K) / k1 + 1 : 0;
    int e;
    if ((e = pgrc_buf_ensure(c, c->d_cnt, (hs + 2) * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_cumm, (hs + 2) * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_ent, (c->npos + 16) * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_head, hs * 2 * sizeof(uint64_t)))) return e;
    uint32_t *cnt = (uint32_t *)c->d_cnt.p, *cumm = (uint32_t *)c->d_cumm.p;
    uint64_t *ent = (uint64_t *)c->d_ent.p;
    const uint32_t *pg = (const uint32_t *)c->pg2[strand].p;
    const uint64_t pg_alloc = c->pg_words + PGRC_PG_PAD_WORDS;
    HIP_TRY(c, hipMemsetAsync(cnt, 0, (hs + 2) * sizeof(uint32_t), c->stream));
    const uint64_t ntiles = (c->npos + IDX_TPB - 1) / IDX_TPB;
    const uint32_t grid = (uint32_t)(ntiles < 256u * 16u ? (ntiles ? ntiles : 1) : 256u * 16u);
    if (c->npos)
        hipLaunchKernelGGL(k_copmem_index_pass<false>, dim3(grid), dim3(IDX_TPB), 0, c->stream, pg, pg_alloc, c->npos, k1, K,
                           (uint32_t)(hs - 1), cnt, (const uint32_t *)nullptr, (uint64_t *)nullptr);
    if ((e = run_scan(c, false, cnt, hs, cumm))) return e;
    HIP_TRY(c, hipMemsetAsync(cnt, 0, (hs + 2) * sizeof(uint32_t), c->stream));
    if (c->npos)
        hipLaunchKernelGGL(k_copmem_index_pass<true>, dim3(grid), dim3(IDX_TPB), 0, c->stream, pg, pg_alloc, c->npos, k1, K,
                           (uint32_t)(hs - 1), cnt, (const uint32_t *)cumm, ent);
    hipLaunchKernelGGL(k_bucket_finalize, dim3((uint32_t)(hs / 256)), dim3(256), 0, c->stream, (const uint32_t *)cumm, hs, ent,
                       (ulonglong2 *)c->d_head.p);
    HIP_TRY(c, hipGetLastError());
    c->index_strand = strand;
    return PGRC_OK;
}

// canonical layout for tests: capped CSR exactly as the reference leaves cumm / sampledPositions
__global__ void __launch_bounds__(256)
k_export_compact(const uint32_t *__restrict__ cumm_full, const uint32_t *__restrict__ cumm_cap, uint64_t hash_size,
                 const uint64_t *__restrict__ ent, uint32_t *__restrict__ pos_cap) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < hash_size;
         h += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = cumm_full[h], lc = cumm_cap[h], n = cumm_cap[h + 1] - lc;
        for (uint32_t k = 0; k < n; k++) pos_cap[lc + k] = (uint32_t)(ent[lo + k] >> PGRC_FP_BITS);
    }
}

int pgrc_copmem_export_index(pgrc_match_ctx *c, uint32_t *h_cumm, uint32_t *h_positions, uint64_t *count) {
    const uint64_t hs = c->cp.hash_size;
    DevBuf cc, pc;
    int e;
    if ((e = pgrc_buf_ensure(c, cc, (hs + 2) * sizeof(uint32_t)))) return e;
    // after the fill pass d_cnt again holds the raw (uncapped) bucket counts
    if ((e = run_scan(c, true, (const uint32_t *)c->d_cnt.p, hs, (uint32_t *)cc.p))) { pgrc_buf_free(cc); return e; }
    uint32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, (uint32_t *)cc.p + hs, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (count) *count = total;
    if (h_cumm) HIP_TRY(c, hipMemcpy(h_cumm, cc.p, (hs + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (h_positions && total) {
        if ((e = pgrc_buf_ensure(c, pc, (size_t)total * sizeof(uint32_t)))) { pgrc_buf_free(cc); return e; }
        const uint32_t sgrid = (uint32_t)((hs + 255) / 256 < 65536u * 8u ? (hs + 255) / 256 : 65536u * 8u);
        hipLaunchKernelGGL(k_export_compact, dim3(sgrid), dim3(256), 0, c->stream, (const uint32_t *)c->d_cumm.p,
                           (const uint32_t *)cc.p, hs, (const uint64_t *)c->d_ent.p, (uint32_t *)pc.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(h_positions, pc.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    pgrc_buf_free(cc);
    pgrc_buf_free(pc);
    return PGRC_OK;
}

// ----------------------------------------------------------------------------- matching

struct MatchArgs {
    const uint32_t *pg;
    uint64_t G;
    const uint32_t *reads;
    uint64_t n, stride;
    const uint8_t *nflag;
    const ulonglong2 *head;
    const uint64_t *ent;
    uint64_t *pos;
    uint8_t *rc, *mism;
    unsigned long long *counters; // [0] searched [1] candidates [2] probes
    uint32_t L, K, k2, mask, kmax, kmin, strand;
};

#define MATCH_TPB 256

// Per-read state of the reference's sequential query (CopMEMMatcher.cpp:483-566).
struct ReadState {
    uint32_t limit, falses, cur;
    uint64_t best;
    bool done;
};

// One candidate entry against read `rd` (2-bit words).  Order of the checks = the reference's:
// bounds (:517-520), head count vs limit (:523-539, +1 false), tail (:540-551, +2 falses), accept
// (:552-560).  The fingerprint test decides most head rejects without touching the text: its
// symbols are head symbols of the window (mask fpm), so fp mismatches > limit => head count > limit.
template <int NW>
__device__ __forceinline__ void try_candidate(const MatchArgs &a, const uint32_t (&rd)[NW], uint64_t e, uint32_t s,
                                              uint32_t fp_read, uint32_t fpm, int H, ReadState &st,
                                              uint64_t &n_cand) {
    const uint64_t sp = e >> PGRC_FP_BITS;
    if ((uint64_t)s > sp) return;
    const uint64_t p = sp - s;
    if (p + a.L > a.G) return;
    n_cand++;
    const uint32_t x = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
    if ((uint32_t)__popc((x | (x >> 1)) & fpm) > st.limit) { st.falses += 1; return; }
    const uint32_t *src = a.pg + (p >> 4);
    const uint32_t b = ((uint32_t)p & 15u) * 2u;
    uint32_t pw[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; k++) pw[k] = src[k];
    uint32_t mh = 0, mt = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint32_t tw = funnel_r(pw[k], pw[k + 1], b);
        mh += mism2(tw, rd[k], sym_mask(k, 0, H));
        mt += mism2(tw, rd[k], sym_mask(k, H, (int)a.L));
    }
    if (mh > st.limit) { st.falses += 1; return; }
    const uint32_t m = mh + mt;
    if (m > st.limit) { st.falses += 2; return; }
    st.cur = m;
    st.best = p;
    if (m <= a.kmin) { st.done = true; return; }
    st.limit = m - 1u;
}

template <int NW>
__global__ void __launch_bounds__(MATCH_TPB) k_copmem_match(const MatchArgs a) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    hash_lut_init(lut);
    __syncthreads();

    const uint64_t i = (uint64_t)blockIdx.x * MATCH_TPB + threadIdx.x;
    bool active = i < a.n;
    const uint32_t cin = active ? a.mism[i] : 0u;
    if (active && a.nflag && a.nflag[i]) active = false; // 'N' reads: byte-path kernel
    if (cin <= a.kmin) active = false;                    // ReadsMatchers.cpp:430

    uint64_t n_cand = 0, n_probe = 0;
    if (active) {
        const int H = ((int)a.L / 8) * 8; // head = whole 8-symbol groups (CopMEMMatcher.cpp:495)
        uint32_t rd[NW], sh[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) sh[k] = rd[k] = a.reads[(uint64_t)k * a.stride + i];

        ReadState st;
        st.limit = (cin < a.kmax) ? cin - 1u : a.kmax;                 // :488-489
        st.falses = 0;
        st.cur = cin;
        st.best = PGRC_NOT_MATCHED_POS;
        st.done = false;
        const uint32_t budget = (a.L + 1u - a.K) / a.k2;               // :496-498
        const uint32_t sbits = 2u * a.k2;

        for (uint32_t s = 0; s + a.K <= a.L && !st.done; s += a.k2) {  // :503
            uint32_t fp_read;
            const uint32_t h = copmem_hash32_fp(sh[0], NW > 1 ? sh[1 % NW] : 0u, NW > 2 ? sh[2 % NW] : 0u,
                                                NW > 3 ? sh[3 % NW] : 0u, a.K, lut, &fp_read) & a.mask;
            n_probe++;
            const ulonglong2 hd = a.head[h];
            if (hd.x) {
                const uint32_t fpm = fp_head_mask(a.K, s, (uint32_t)H);
                const uint32_t cnt = HEAD_COUNT(hd.x);
                uint32_t nb = cnt;
                if (st.falses > budget) nb = min(nb, PGRC_TRUNC_BUCKET);     // :510-514
                for (uint32_t j = 0; j < nb && !st.done; j++) {
                    const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (cnt == 2 ? hd.y : a.ent[(uint32_t)hd.y + j]);
                    try_candidate<NW>(a, rd, e, s, fp_read, fpm, H, st, n_cand);
                }
            }
            // slide the seed window by k2 symbols
#pragma unroll
            for (int k = 0; k < NW - 1; k++) sh[k] = funnel_r(sh[k], sh[k + 1], sbits);
            sh[NW - 1] >>= sbits;
        }
        if (st.best != PGRC_NOT_MATCHED_POS && st.cur < cin) {         // ReadsMatchers.cpp:437-447
            a.pos[i] = a.strand ? a.G - (st.best + a.L) : st.best;
            a.rc[i] = (uint8_t)a.strand;
            a.mism[i] = (uint8_t)st.cur;
        }
    }
    if (a.counters) {
        const uint64_t s0 = wave_sum_u64(active ? 1ull : 0ull), s1 = wave_sum_u64(n_cand), s2 = wave_sum_u64(n_probe);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)s0);
            atomicAdd(&a.counters[1], (unsigned long long)s1);
            atomicAdd(&a.counters[2], (unsigned long long)s2);
        }
    }
}

// Per-lane state machine version of the same query.  The sequential kernel above makes a whole wave
// wait out up to three dependent memory latencies per seed (bucket head -> bucket entry -> text
// window) whenever ANY of its 64 reads needs them.  Here every lane advances its own read by one
// memory access per iteration -- a head (mode 0), the next entry of a multi-entry bucket (mode 1) or
// a text window to verify (mode 2) -- and all lanes' loads of an iteration are issued together, so an
// iteration costs one latency.  The per-read order of events is exactly the reference's.
#define SM_MAX_SEEDS 240
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U32x4A4 { u32x4 v; }; // 16-B load that only needs 4-B alignment
template <int NW>
__global__ void __launch_bounds__(MATCH_TPB) k_copmem_match_sm(const MatchArgs a) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t fpm_tab[SM_MAX_SEEDS];
    hash_lut_init(lut);
    const int H = ((int)a.L / 8) * 8;
    const uint32_t nseeds = (a.L - a.K) / a.k2 + 1; // seeds s = 0, k2, ... with s + K <= L
    for (uint32_t t = threadIdx.x; t < nseeds && t < SM_MAX_SEEDS; t += blockDim.x)
        fpm_tab[t] = fp_head_mask(a.K, t * a.k2, (uint32_t)H);
    __syncthreads();

    const uint64_t i = (uint64_t)blockIdx.x * MATCH_TPB + threadIdx.x;
    bool active = i < a.n;
    const uint32_t cin = active ? a.mism[i] : 0u;
    if (active && a.nflag && a.nflag[i]) active = false;
    if (cin <= a.kmin) active = false;

    uint64_t n_cand = 0, n_probe = 0, n_ent = 0, n_ver = 0;
    uint32_t rd[NW], sh[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) sh[k] = rd[k] = active ? a.reads[(uint64_t)k * a.stride + i] : 0u;

    ReadState st;
    st.limit = (cin < a.kmax) ? cin - 1u : a.kmax;
    st.falses = 0;
    st.cur = cin;
    st.best = PGRC_NOT_MATCHED_POS;
    st.done = false;
    const uint32_t budget = (a.L + 1u - a.K) / a.k2;
    const uint32_t sbits = 2u * a.k2;

    enum { M_PROBE = 0, M_ENTRY = 1, M_VERIFY = 2, M_FIN = 3, M_ADV = 4 };
    uint32_t mode = active ? M_PROBE : M_FIN;
    uint32_t si = 0;              // seed index: s = si * k2
    uint32_t lo = 0, nb = 0, j = 0, fp_read = 0;
    uint64_t cand_p = 0, e_inline1 = 0;
    bool inline1 = false; // entry 1 of the current bucket sits in e_inline1 (2-entry bucket)
    // the last verified alignment and its head/tail counts: a read accepted with m > 0 mismatches meets
    // its own alignment again at every later seed that is sampled there; the counts cannot change, only
    // the limit they are judged against does -- no need to fetch the text window again
    uint64_t last_p = PGRC_NOT_MATCHED_POS;
    uint32_t last_mh = 0, last_mt = 0;
    constexpr int PWN = ((NW + 1 + 3) / 4) * 4;

    // judge a verified alignment (head count mh, tail count mt) exactly as CopMEMMatcher.cpp:536-560
    auto judge = [&](uint32_t mh, uint32_t mt, uint64_t p) {
        const uint32_t m = mh + mt;
        if (mh > st.limit) st.falses += 1;                       // :536-539
        else if (m > st.limit) st.falses += 2;                   // :542-551 (counted twice)
        else {
            st.cur = m;                                          // :552-555
            st.best = p;
            if (m <= a.kmin) st.done = true;                     // :556-559
            else st.limit = m - 1u;                              // :560
        }
    };

    while (__any(mode != M_FIN)) {
        const uint32_t m0 = mode;
        const uint32_t s = si * a.k2;
        // ---- issue this iteration's loads
        ulonglong2 hd = make_ulonglong2(0ull, 0ull);
        uint64_t v = 0;
        if (m0 == M_PROBE) {
            const uint32_t h = copmem_hash32_fp(sh[0], NW > 1 ? sh[1 % NW] : 0u, NW > 2 ? sh[2 % NW] : 0u,
                                                NW > 3 ? sh[3 % NW] : 0u, a.K, lut, &fp_read) & a.mask;
            hd = a.head[h];
            n_probe++;
        } else if (m0 == M_ENTRY) {
            v = inline1 ? e_inline1 : a.ent[lo + j];
            n_ent += inline1 ? 0 : 1;
        }
        uint32_t pw[PWN];
        const uint32_t b = ((uint32_t)cand_p & 15u) * 2u;
        if (m0 == M_VERIFY) {
            const uint32_t *src = a.pg + (cand_p >> 4); // the text is padded: PWN words are always in bounds
#pragma unroll
            for (int k = 0; k < PWN; k += 4) {
                const u32x4 q = reinterpret_cast<const U32x4A4 *>(src + k)->v;
                pw[k] = q.x; pw[k + 1] = q.y; pw[k + 2] = q.z; pw[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PWN; k++) pw[k] = 0;
        }
        // ---- consume
        uint32_t next = m0;
        if (m0 == M_VERIFY) {
            uint32_t mh = 0, mt = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const uint32_t tw = funnel_r(pw[k], pw[k + 1], b);
                mh += mism2(tw, rd[k], sym_mask(k, 0, H));
                mt += mism2(tw, rd[k], sym_mask(k, H, (int)a.L));
            }
            n_ver++;
            last_p = cand_p;
            last_mh = mh;
            last_mt = mt;
            judge(mh, mt, cand_p);
            next = st.done ? M_FIN : (j < nb ? M_ENTRY : M_ADV);
        } else if (m0 <= M_ENTRY) {
            bool have = false;
            uint64_t e = 0;
            if (m0 == M_PROBE) {
                if (hd.x) {
                    const uint32_t cnt = HEAD_COUNT(hd.x);
                    nb = cnt;
                    if (st.falses > budget) nb = min(nb, PGRC_TRUNC_BUCKET); // :510-514
                    inline1 = cnt == 2;
                    e_inline1 = hd.y;
                    lo = (uint32_t)hd.y;
                    e = hd.x & ENT_MASK;
                    have = true;
                    j = 1;
                } else {
                    next = M_ADV;
                }
            } else {
                e = v;
                have = true;
                j++;
            }
            if (have) {
                next = (j < nb) ? M_ENTRY : M_ADV;
                const uint64_t sp = e >> PGRC_FP_BITS;
                if ((uint64_t)s <= sp && sp - s + a.L <= a.G) {      // :517-520
                    n_cand++;
                    const uint32_t x = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
                    if ((uint32_t)__popc((x | (x >> 1)) & fpm_tab[si]) > st.limit) {
                        st.falses += 1;                              // certain head reject
                    } else if (sp - s == last_p) {
                        judge(last_mh, last_mt, last_p);
                        if (st.done) next = M_FIN;
                    } else {
                        cand_p = sp - s;
                        next = M_VERIFY;
                    }
                }
            }
        }
        if (next == M_ADV) {
            si++;
#pragma unroll
            for (int k = 0; k < NW - 1; k++) sh[k] = funnel_r(sh[k], sh[k + 1], sbits);
            sh[NW - 1] >>= sbits;
            next = (si < nseeds) ? M_PROBE : M_FIN;
        }
        mode = next;
    }
    if (active && st.best != PGRC_NOT_MATCHED_POS && st.cur < cin) {
        a.pos[i] = a.strand ? a.G - (st.best + a.L) : st.best;
        a.rc[i] = (uint8_t)a.strand;
        a.mism[i] = (uint8_t)st.cur;
    }
    if (a.counters) {
        const uint64_t s0 = wave_sum_u64(active ? 1ull : 0ull), s1 = wave_sum_u64(n_cand), s2 = wave_sum_u64(n_probe);
        const uint64_t s3 = wave_sum_u64(n_ent), s4 = wave_sum_u64(n_ver);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)s0);
            atomicAdd(&a.counters[1], (unsigned long long)s1);
            atomicAdd(&a.counters[2], (unsigned long long)s2);
            atomicAdd(&a.counters[3], (unsigned long long)s3);
            atomicAdd(&a.counters[4], (unsigned long long)s4);
        }
    }
}

// Byte path for reads containing 'N' (the reference's N read set, ACGNT-packed; an N never equals a
// Pg symbol and is hashed as the byte 0x4E -- SURVEY.md Appendix A notes).  One read per lane.
__global__ void __launch_bounds__(MATCH_TPB)
k_copmem_match_ascii(const MatchArgs a, const uint32_t *__restrict__ nidx, const uint8_t *__restrict__ nascii, uint64_t nn) {
    const uint64_t t = (uint64_t)blockIdx.x * MATCH_TPB + threadIdx.x;
    bool active = t < nn;
    uint64_t i = active ? nidx[t] : 0;
    uint32_t cin = active ? a.mism[i] : 0u;
    if (cin <= a.kmin) active = false;
    uint64_t n_cand = 0, n_probe = 0;
    if (active) {
        const uint8_t *rd = nascii + t * a.L;
        const uint32_t H = (a.L / 8) * 8;
        uint32_t limit = (cin < a.kmax) ? cin - 1u : a.kmax;
        const uint32_t budget = (a.L + 1u - a.K) / a.k2;
        uint32_t falses = 0, cur = cin;
        uint64_t best = PGRC_NOT_MATCHED_POS;
        bool done = false;
        for (uint32_t s = 0; s + a.K <= a.L && !done; s += a.k2) {
            uint32_t h = a.K;
            for (uint32_t j = 0; j < a.K / 4; j++) {
                const uint8_t *q = rd + s + 4 * j;
                uint32_t w = (uint32_t)q[0] | ((uint32_t)q[1] << 8);
                if (j < 3) w |= (uint32_t)q[2] << 16;
                h = (h ^ (w + j)) * 171717u;
            }
            h &= a.mask;
            n_probe++;
            const ulonglong2 hd = a.head[h];
            if (!hd.x) continue;
            const uint32_t cnt = HEAD_COUNT(hd.x);
            uint32_t nb = cnt;
            if (falses > budget) nb = min(nb, PGRC_TRUNC_BUCKET);
            for (uint32_t j = 0; j < nb; j++) {
                const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (cnt == 2 ? hd.y : a.ent[(uint32_t)hd.y + j]);
                const uint64_t sp = e >> PGRC_FP_BITS;
                if ((uint64_t)s > sp) continue;
                const uint64_t p = sp - s;
                if (p + a.L > a.G) continue;
                n_cand++;
                uint32_t mh = 0, mt = 0;
                for (uint32_t k = 0; k < a.L; k++) {
                    const uint64_t x = p + k;
                    const uint32_t code = (a.pg[x >> 4] >> (2u * ((uint32_t)x & 15u))) & 3u;
                    const uint32_t ne = code2ascii(code) != (uint32_t)rd[k];
                    if (k < H) mh += ne; else mt += ne;
                }
                if (mh > limit) { falses += 1; continue; }
                const uint32_t m = mh + mt;
                if (m > limit) { falses += 2; continue; }
                cur = m;
                best = p;
                if (m <= a.kmin) { done = true; break; }
                limit = m - 1u;
            }
        }
        if (best != PGRC_NOT_MATCHED_POS && cur < cin) {
            a.pos[i] = a.strand ? a.G - (best + a.L) : best;
            a.rc[i] = (uint8_t)a.strand;
            a.mism[i] = (uint8_t)cur;
        }
    }
    if (a.counters) {
        const uint64_t s0 = wave_sum_u64(active ? 1ull : 0ull), s1 = wave_sum_u64(n_cand), s2 = wave_sum_u64(n_probe);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)s0);
            atomicAdd(&a.counters[1], (unsigned long long)s1);
            atomicAdd(&a.counters[2], (unsigned long long)s2);
        }
    }
}

template <int NW>
static void launch_match(pgrc_match_ctx *c, const MatchArgs &a) {
    const uint32_t grid = (uint32_t)((a.n + MATCH_TPB - 1) / MATCH_TPB);
    const char *ev = getenv("PGRC_MATCH_KERNEL"); // tuning knob: "seq" = wave-sequential variant
    if (ev && ev[0] == 's')
        hipLaunchKernelGGL(k_copmem_match<NW>, dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
    else
        hipLaunchKernelGGL(k_copmem_match_sm<NW>, dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
}

int pgrc_copmem_match_pass(pgrc_match_ctx *c, int strand) {
    if (c->n == 0) return PGRC_OK;
    MatchArgs a;
    a.pg = (const uint32_t *)c->pg2[strand].p;
    a.G = c->G;
    a.reads = c->reads2;
    a.n = c->n;
    a.stride = c->stride;
    a.nflag = c->n_nreads ? (const uint8_t *)c->nread_flag.p : nullptr;
    a.head = (const ulonglong2 *)c->d_head.p;
    a.ent = (const uint64_t *)c->d_ent.p;
    a.pos = (uint64_t *)c->d_pos.p;
    a.rc = (uint8_t *)c->d_rc.p;
    a.mism = (uint8_t *)c->d_mism.p;
    a.counters = (unsigned long long *)c->d_counters.p + 8 * strand;
    a.L = c->prm.read_len;
    a.K = (uint32_t)c->cp.K;
    a.k2 = (uint32_t)c->cp.k2;
    a.mask = c->cp.hash_size - 1;
    a.kmax = c->prm.max_mismatches;
    a.kmin = c->prm.min_mismatches;
    a.strand = (uint32_t)strand;
    switch (c->nw) {
#define CASE_NW(N) case N: launch_match<N>(c, a); break;
        CASE_NW(2) CASE_NW(3) CASE_NW(4) CASE_NW(5) CASE_NW(6) CASE_NW(7) CASE_NW(8) CASE_NW(9)
        CASE_NW(10) CASE_NW(11) CASE_NW(12) CASE_NW(13) CASE_NW(14) CASE_NW(15) CASE_NW(16)
#undef CASE_NW
    default:
        c->err = "unsupported read length for mode c";
        return PGRC_E_PARAM;
    }
    HIP_TRY(c, hipGetLastError());
    if (c->n_nreads) {
        const uint32_t grid = (uint32_t)((c->n_nreads + MATCH_TPB - 1) / MATCH_TPB);
        hipLaunchKernelGGL(k_copmem_match_ascii, dim3(grid), dim3(MATCH_TPB), 0, c->stream, a,
                           (const uint32_t *)c->nread_idx.p, (const uint8_t *)c->nread_ascii.p, c->n_nreads);
        HIP_TRY(c, hipGetLastError());
    }
    return PGRC_OK;
}
