// mem.hip -- pseudogenome-vs-pseudogenome exact matching on gfx950.
//
// Reference behaviour restated (not translated):
//   CopMEMMatcher::matchTexts -> processExactMatchQueryTight, matching/copmem/CopMEMMatcher.cpp:333-481, :604-622,
//   driven by SimplePgMatcher::exactMatchPg, matching/SimplePgMatcher.cpp:24-55.
//
// The reference scans the destination text left to right in steps of k2; for every window whose bucket is not
// empty it walks the bucket and (a) drops self matches, (b) jumps ahead when the window lies inside the previous
// match on the same diagonal, (c) pre-filters on two 4-symbol side contexts, (d) extends and, if long enough, records
// a match and jumps ahead.  (b) and the jumps make the scan sequential -- but only (window, entry) pairs whose K-mers
// are EQUAL can record a match or trigger (b) (a window inside a match on its diagonal equals the source there),
// and those pairs are rare.  So:
//   1. k_mem_probe   : every destination window in parallel -- hash, ONE 16-byte head gather, fingerprint reject,
//                      exact K-mer compare -- emits the equal pairs ("events") after filter (a);
//   2. rocPRIM sort  : events by (window, bucket order) = the order the reference meets them;
//   3. k_mem_flags .. k_mem_apply : the side-context test (c) per event; the maximal extension (d) once per RUN of
//                      events that one match produces on its diagonal (linear in the text, however long the match);
//   4. k_mem_replay  : (b), the jumps (which never leave a block of 256 windows in the main loop, :365-424, but carry
//                      through in the tail loop, :428-476) and the acceptance, one thread per block of windows that
//                      holds events, in rounds until every block has seen the last match recorded before it;
//   5. k_mem_emit    : the accepted events, compacted in discovery order -- the only thing the host receives.
// The side-context registers l1/r1/l2/r2 are refreshed only when their 4 bytes lie inside the text (:381-382,
// :401-402): for the handful of events at the text ends the host re-creates the stale value the reference would
// hold by walking back through the windows actually examined (one block's outcomes and bucket lookups from the device
// on demand), tells the device the verdict and lets the replay continue.
//
// All integer work, bound by the random head gathers (one per window): no MFMA.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "scanops.h"

#include "ctx.h"
#include "devutil.h"
#include "headfmt.h"
#include "pgrc_mem.h"

struct pgrc_mem_ctx {
    pgrc_match_ctx *base = nullptr;   // owns the packed source, its reverse complement and the seed index
    uint32_t L = 0;
    int K = 0, k1 = 0, k2 = 0, LK2 = 0, KLK24 = 0;
    const char *src = nullptr;        // borrowed host text
    uint64_t N = 0;
    bool have_src = false;
    DevBuf d_dest, d_nmap, d_stage, d_flag, d_cursor, d_evk[2], d_evv[2], d_tmp, d_scan, d_orun, d_oflag;
    DevBuf d_skey[2], d_sidx[2], d_first, d_runid, d_rstart, d_rend;   // events by (diagonal, window): sort ping-pong, runs
    DevBuf d_rdend, d_outc, d_ebstart, d_ebin, d_ebout, d_ebinc, d_small, d_match;   // the replay: per run, per event, per event block
    hipEvent_t ev[5]{};               // phase timing (created on first use)
    bool have_ev = false;
    pgrc_mem_counters ctr{};
    std::string err;
};

#define MEM_TRY(m, expr)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (m)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return pgrc_hip_code(e__);                                                       \
        }                                                                                    \
    } while (0)

static thread_local std::string g_mem_create_err; // reported by pgrc_mem_last_error(NULL)

// ------------------------------------------------------------------------------------------------ device side

// 16 ASCII symbols -> one 2-bit word + a 16-bit mask of the 'N's among them (N packs as code 0)
__global__ void __launch_bounds__(256)
k_mem_pack(const uint8_t *__restrict__ ascii, uint64_t count, uint32_t *__restrict__ words, uint16_t *__restrict__ nmap,
           uint32_t *flags) {
    const uint64_t nwords = (count + 15) / 16;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t out = 0, nb = 0, fl = 0;
        for (uint32_t k = 0; k < 16 && w * 16 + k < count; k++) {
            const uint32_t c = ascii[w * 16 + k];
            uint32_t x = (c >> 1) & 3u;
            x ^= x >> 1;                                   // A0 C1 G2 T3
            if (c == 'N') { nb |= 1u << k; x = 0; fl |= 2u; }
            else if (c != 'A' && c != 'C' && c != 'G' && c != 'T') fl |= 1u;
            out |= x << (2 * k);
        }
        words[w] = out;
        nmap[w] = (uint16_t)nb;
        if (fl) atomicOr(flags, fl);
    }
}

struct MemArgs {
    const uint32_t *src;
    const uint32_t *dest;
    const uint16_t *nmap;        // nullptr: the destination holds no 'N'
    uint64_t N, N2;
    uint64_t dest_words_alloc;
    const ulonglong2 *head;      // head of bucket h at head[head_slot(h, hsh)] (headfmt.h)
    uint32_t hsh;
    const uint64_t *ent;
    uint32_t mask, K, k2;
    uint32_t LK2, KLK24;
    uint64_t nprobes;
    int dest_is_src, rev_compl;
};

__device__ __forceinline__ uint32_t sym16(const uint32_t *t, uint64_t pos) {   // 16 symbols from an arbitrary position
    const uint32_t *p = t + (pos >> 4);
    return funnel_r(p[0], p[1], ((uint32_t)pos & 15u) * 2u);
}
__device__ __forceinline__ uint32_t nbits16(const uint16_t *nm, uint64_t pos) { // the N flags of the same 16 symbols
    const uint64_t w = pos >> 4;
    const uint32_t v = (uint32_t)nm[w] | ((uint32_t)nm[w + 1] << 16);
    return (v >> ((uint32_t)pos & 15u)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t spread16(uint32_t x) {                      // bit i -> bit 2i
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// per-symbol difference mask (bits at even positions) of 16 source symbols at ps and 16 destination symbols at pd
__device__ __forceinline__ uint32_t diff16(const MemArgs &a, uint64_t ps, uint64_t pd) {
    const uint32_t x = sym16(a.src, ps) ^ sym16(a.dest, pd);
    uint32_t d = (x | (x >> 1)) & 0x55555555u;
    if (a.nmap) d |= spread16(nbits16(a.nmap, pd));
    return d;
}

#define MEM_TPB 256
#define MEM_TILE_WORDS 272   // 255 * k2 (k2 <= 15) + K (<= 56) symbols, + slack
#define MEM_WCAP 256u        // events a wave collects in LDS before it writes them out

// 1. one thread per destination window.  The walk over a bucket is wave-uniform (entry j of every lane's bucket in
// turn; buckets hold one or two entries almost always), so the lanes that found an event take consecutive slots of their
// wave's LDS buffer by ballot: no LDS atomic per event and 16 KB of buffers per block (a block-wide 32 KB buffer halved the
// resident waves: 17.0 -> 14.2 ms for the 625 M windows of the C3 text).  At its end the block takes room for all four
// buffers with ONE global atomic (one per wave made the cursor's address the bottleneck of event-rich texts: 1.2 -> 4.3 ms
// for a 60 Mbp copy of the source); only a wave whose buffer fills up earlier writes it out by itself.
__global__ void __launch_bounds__(MEM_TPB)
k_mem_probe(const MemArgs a, unsigned long long *cursor, uint64_t *__restrict__ evk, uint64_t *__restrict__ evv, uint64_t cap) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t tile[MEM_TILE_WORDS];
    __shared__ uint64_t wk[MEM_TPB / 64][MEM_WCAP], wv[MEM_TPB / 64][MEM_WCAP];
    __shared__ uint32_t wfill[MEM_TPB / 64];
    __shared__ unsigned long long gbase;
    hash_lut_init(lut);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * MEM_TPB;
    const uint64_t q0 = t0 * a.k2;
    const uint64_t w0 = q0 >> 4;
    const uint32_t need = (uint32_t)((((q0 & 15) + (uint64_t)(MEM_TPB - 1) * a.k2 + a.K + 15) >> 4) + 5);
    for (uint32_t w = threadIdx.x; w < need; w += MEM_TPB) tile[w] = (w0 + w < a.dest_words_alloc) ? a.dest[w0 + w] : 0u;
    __syncthreads();
    const uint64_t t = t0 + threadIdx.x;
    const uint64_t q = t * a.k2;
    uint32_t cnt = 0, fp = 0;
    ulonglong2 hd = make_ulonglong2(0, 0);
    uint32_t dw[4] = {0, 0, 0, 0};
    if (t < a.nprobes) {
        const uint32_t x = (uint32_t)((q >> 4) - w0);
        const uint32_t sh = ((uint32_t)q & 15u) * 2u;
        dw[0] = funnel_r(tile[x], tile[x + 1], sh);
        dw[1] = funnel_r(tile[x + 1], tile[x + 2], sh);
        dw[2] = funnel_r(tile[x + 2], tile[x + 3], sh);
        dw[3] = funnel_r(tile[x + 3], tile[x + 4], sh);
        bool has_n = false;
        if (a.nmap)
            for (uint32_t k = 0; k < a.K; k += 16) {
                const uint32_t nb = nbits16(a.nmap, q + k);
                has_n |= (nb & (a.K - k >= 16 ? 0xFFFFu : ((1u << (a.K - k)) - 1u))) != 0;
            }
        if (!has_n) {       // a window with an 'N' equals no source K-mer: it can produce no event
            const uint32_t h = copmem_hash32_fp(dw[0], dw[1], dw[2], dw[3], a.K, lut, &fp) & a.mask;
            hd = a.head[head_slot(h, a.hsh)];
            cnt = head_count(hd);
        }
    }
    const uint64_t base = hd.y & W1_BASE_MASK;
    uint32_t fill = 0;                                    // events in this wave's buffer (wave-uniform)
    auto flush = [&]() {
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(cursor, (unsigned long long)fill);
        gb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
        __builtin_amdgcn_wave_barrier();                  // (the slots were written by other lanes of this wave)
        for (uint32_t i = lane; i < fill; i += 64)
            if (gb + i < cap) { evk[gb + i] = wk[wave][i]; evv[gb + i] = wv[wave][i]; }
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    };
    for (uint32_t j = 0; __ballot(j < cnt) != 0ull; j++) {
        bool ev = false;
        uint64_t p = 0;
        if (j < cnt) {
            const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (cnt == 2 ? hd.y : a.ent[base + j - 1]);
            p = e >> PGRC_FP_BITS;
            const bool self = a.dest_is_src && (a.rev_compl ? a.N2 - p < q : q >= p);                // :389-392
            if (!self && !(((uint32_t)e ^ fp) & ((1u << PGRC_FP_BITS) - 1u))) {                      // else the K-mers differ
                ev = true;
                for (uint32_t k = 0; k < a.K; k += 16) {
                    uint32_t dx = sym16(a.src, p + k) ^ dw[k >> 4];
                    if (a.K - k < 16) dx &= (1u << (2 * (a.K - k))) - 1u;
                    ev &= dx == 0;
                }
            }
        }
        const uint64_t m = __ballot(ev);
        if (m) {
            const uint32_t n = (uint32_t)__popcll(m);
            if (fill + n > MEM_WCAP) flush();
            if (ev) {
                const uint32_t slot = fill + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                wk[wave][slot] = (t << 4) | j;
                wv[wave][slot] = p;
            }
            fill += n;
        }
    }
    // what is left: one reservation for the whole block
    if (lane == 0) wfill[wave] = fill;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wfill[0] + wfill[1] + wfill[2] + wfill[3];
        gbase = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long gb = gbase;
    for (uint32_t w = 0; w < wave; w++) gb += wfill[w];
    for (uint32_t i = lane; i < fill; i += 64)
        if (gb + i < cap) { evk[gb + i] = wk[wave][i]; evv[gb + i] = wv[wave][i]; }
}

// flags of an event
#define MF_L1_OK 1u    // the source side context lies inside the source text (else the register is stale, :401-402)
#define MF_R1_OK 2u
#define MF_L2_OK 4u    // the same for the destination (:381-382)
#define MF_R2_OK 8u
#define MF_L_EQ 16u    // both left contexts fresh and equal
#define MF_R_EQ 32u
#define MF_LONG 64u    // the match of the event's run is long enough to be recorded (:413)

// 3. side contexts and extensions.  Every event of one maximal match (same diagonal, connected by equal symbols) has the
// same extents, and a long match carries one event per lcm(k1, k2) symbols: extending each of them separately would be
// quadratic in the match length (a 10 Mbp duplicate: 10^12 symbol compares).  So the events are ordered by (diagonal,
// window) once, neighbours on a diagonal are tested for being CONNECTED (only the gap between their K-mers is
// compared, usually nothing: the K-mers overlap), a prefix sum numbers the runs, the first event of a run extends to
// the left, the last one to the right, and everybody takes its run's extents: linear in the text.
__global__ void __launch_bounds__(256)
k_mem_flags(const MemArgs a, const uint64_t *__restrict__ evk, const uint64_t *__restrict__ evv, uint64_t nev,
            uint8_t *__restrict__ oflag, uint64_t *__restrict__ qkey, uint64_t *__restrict__ idx, uint32_t *nstale) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nev) return;
    const uint64_t q = (evk[i] >> 4) * a.k2, p = evv[i];
    uint32_t fl = 0;
    if (p >= a.LK2) fl |= MF_L1_OK;
    if (p + a.KLK24 + 4 <= a.N) fl |= MF_R1_OK;
    if (q >= a.LK2) fl |= MF_L2_OK;
    if (q + a.KLK24 + 4 <= a.N2) fl |= MF_R2_OK;
    if ((fl & (MF_L1_OK | MF_L2_OK)) == (MF_L1_OK | MF_L2_OK) && (diff16(a, p - a.LK2, q - a.LK2) & 0xFFu) == 0) fl |= MF_L_EQ;
    if ((fl & (MF_R1_OK | MF_R2_OK)) == (MF_R1_OK | MF_R2_OK) && (diff16(a, p + a.KLK24, q + a.KLK24) & 0xFFu) == 0) fl |= MF_R_EQ;
    oflag[i] = (uint8_t)fl;
    qkey[i] = q;
    idx[i] = i;
    if ((fl & 15u) != 15u) atomicAdd(nstale, 1u);                    // (a handful: only at the ends of the texts)
}

// diagonal of event idx[k] (offset by N so that it is not negative), as the key of the second, stable sort
__global__ void __launch_bounds__(256)
k_mem_diag(const MemArgs a, const uint64_t *__restrict__ evk, const uint64_t *__restrict__ evv, const uint64_t *__restrict__ idx,
           uint64_t nev, uint64_t *__restrict__ dkey) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nev) return;
    const uint64_t i = idx[k];
    dkey[k] = (evk[i] >> 4) * a.k2 + a.N - evv[i];
}

// first[k] = 1 when event idx[k] starts a new run: another diagonal, or a difference in the gap to its predecessor
__global__ void __launch_bounds__(256)
k_mem_connect(const MemArgs a, const uint64_t *__restrict__ evk, const uint64_t *__restrict__ evv, const uint64_t *__restrict__ idx,
              const uint64_t *__restrict__ dkey, uint64_t nev, uint32_t *__restrict__ first) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nev) return;
    uint32_t f = 1;
    if (k > 0 && dkey[k] == dkey[k - 1]) {
        const uint64_t i = idx[k], j = idx[k - 1];
        const uint64_t q1 = (evk[i] >> 4) * a.k2, p1 = evv[i];
        const uint64_t q0 = (evk[j] >> 4) * a.k2, p0 = evv[j];      // q0 < q1 (sorted by window within the diagonal)
        bool same = true;
        for (uint64_t g = q0 + a.K; g < q1 && same; g += 16) {       // the symbols between the two K-mers, if any
            uint32_t d = diff16(a, p0 + (g - q0), g);
            if (q1 - g < 16) d &= (1u << (2 * (uint32_t)(q1 - g))) - 1u;
            same = d == 0;
        }
        (void)p1;
        f = same ? 0u : 1u;
    }
    first[k] = f;
}

// the first event of a run finds the match start, the last one the match end (runid = inclusive sum of first[] - 1)
__global__ void __launch_bounds__(256)
k_mem_run_ends(const MemArgs a, const uint64_t *__restrict__ evk, const uint64_t *__restrict__ evv, const uint64_t *__restrict__ idx,
               const uint32_t *__restrict__ first, const uint32_t *__restrict__ runid, uint64_t nev,
               uint64_t *__restrict__ run_start, uint64_t *__restrict__ run_end, uint64_t *__restrict__ run_dend) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nev) return;
    const bool leader = first[k] != 0, tail = (k + 1 == nev) || first[k + 1] != 0;
    if (!leader && !tail) return;
    const uint64_t i = idx[k];
    const uint64_t q = (evk[i] >> 4) * a.k2, p = evv[i];
    const uint32_t r = runid[k] - 1u;
    if (tail) {
        // right: symbols after the K-mer, up to the first difference or either text end (:405-407)
        const uint64_t rlim = min(a.N - (p + a.K), a.N2 - (q + a.K));
        uint64_t x = 0;
        while (x < rlim) {
            const uint32_t d = diff16(a, p + a.K + x, q + a.K + x);
            if (d) { x += (uint32_t)(__ffs((int)d) - 1) >> 1; break; }
            x += 16;
        }
        run_end[r] = p + a.K + min(x, rlim);                         // source position after the match
        run_dend[r] = q + a.K + min(x, rlim);                        // ... and the destination position after it
    }
    if (leader) {
        // left: symbols before the K-mer (:409-411); the loop of the reference stops ON symbol 0 without consuming it
        const uint64_t llim = min(p, q);
        uint64_t s = 0;
        bool mism = false;
        while (llim - s >= 16) {
            const uint32_t d = diff16(a, p - s - 16, q - s - 16);
            if (d) { s += (uint32_t)__clz((int)d) >> 1; mism = true; break; }
            s += 16;
        }
        while (!mism && s < llim) {
            if (diff16(a, p - s - 1, q - s - 1) & 1u) mism = true;
            else s++;
        }
        // first source symbol of the match as the reference reports it: the same number for every event of the run
        // (they all walk left to the same stop)
        run_start[r] = mism ? p - s : p - s + 1;
    }
}

// every event learns its run and whether that run's match is long enough to be recorded: right - p1 > minMatchLength
// with right - p1 = length + 1 (:413)
__global__ void __launch_bounds__(256)
k_mem_apply(const uint64_t *__restrict__ idx, const uint32_t *__restrict__ runid, uint64_t nev, const uint64_t *__restrict__ run_start,
            const uint64_t *__restrict__ run_end, uint32_t min_len, uint32_t *__restrict__ orun, uint8_t *__restrict__ oflag) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nev) return;
    const uint32_t r = runid[k] - 1u;
    const uint64_t i = idx[k];
    orun[i] = r;
    if (run_end[r] - run_start[r] + 1 > (uint64_t)min_len) oflag[i] |= MF_LONG;
}

// 4. the sequential rules, over the events only.  The reference's scan is sequential for two reasons: a jump skips the
// next K/k1 - 1 windows -- but never beyond its block of 256 windows (main loop; the tail loop is one block of its own),
// so blocks are independent there -- and rule (b) looks at the LAST match recorded, which carries from block to block.
// One wave replays one block's events ("event block": only blocks that hold events exist here).  Round 0 assumes
// "no match recorded before"; then every block takes the last match recorded by the nearest block before it that
// recorded one (an exclusive scan with "rightmost valid"), and is replayed again if that match reaches into it (rule
// (b) could fire) and is not what it assumed: round after round until nothing changes.  After round r the first r
// blocks are final, so the rounds end; usually after two or three (a match that straddles a block boundary costs its
// successor one more replay, ONE 20 Mbp match over 26 000 blocks: two rounds).
// A match is named by its run: every event of a run records the same (source, length, destination) triple, and an event
// on the diagonal of the last match lies inside it (q + K < its end) only if it belongs to its run -- two runs on one
// diagonal are separated by a difference.
#define MO_NONE 0u
#define MO_JUMP 1u             // rule (b): inside the previous match, the scan jumped (:393-399)
#define MO_ACCEPT 2u           // recorded a match and jumped (:413-419)
#define MO_STALE 3u            // reached, could record a match, but a side-context register is stale: the host decides
#define MR_NONE 0xFFFFFFFFu    // "no match recorded so far"
#define MR_DIRTY 0xFFFFFFFEu   // assumed incoming match of a block that has to be replayed whatever comes in

__host__ __device__ __forceinline__ uint64_t mem_block_of(uint64_t t, uint64_t nmain) { return t < nmain ? t >> 8 : (nmain >> 8) + 1; }

__global__ void __launch_bounds__(256)
k_mem_lead(const uint64_t *__restrict__ ek, uint64_t nev, uint64_t nmain, uint32_t *__restrict__ lead) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nev) return;
    lead[i] = (i == 0 || mem_block_of(ek[i] >> 4, nmain) != mem_block_of(ek[i - 1] >> 4, nmain)) ? 1u : 0u;
}
// lid = inclusive sum of lead[]: event block of event i = lid[i] - 1
__global__ void __launch_bounds__(256)
k_mem_eb_start(const uint32_t *__restrict__ lead, const uint32_t *__restrict__ lid, uint64_t nev, uint32_t *__restrict__ eb_start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nev) return;
    if (lead[i]) eb_start[lid[i] - 1u] = (uint32_t)i;
    if (i + 1 == nev) eb_start[lid[i]] = (uint32_t)nev;
}

struct MemReplayArgs {
    const uint64_t *ek;          // events in the reference's order: (window << 4) | bucket order
    const uint32_t *orun;        // run of an event
    const uint8_t *oflag;
    uint8_t *outc;               // MO_* per event
    const uint64_t *run_dend;    // destination position after a run's match
    const uint32_t *eb_start;    // [neb + 1]
    uint32_t *eb_in, *eb_out;    // incoming match a block assumed in its last replay / last match it recorded (MR_NONE: none)
    const uint32_t *eb_inc;      // last match recorded before the block, from the blocks' current eb_out
    uint32_t neb, K, k2, skip;
    uint32_t *changed;
    int first;
};

// One WAVE per event block: 64 events are loaded at once (coalesced) and classified in parallel -- "would record a
// match" and "stale" do not depend on the scan's state, "inside the last match" (b) only on the last match -- and the
// wave then steps through the events that the scan REACHES and that do something (a jump, an acceptance, a stale mark),
// in order: every one of them removes the rest of its window and the next `skip` windows from the candidates with one
// ballot.  The state (last match, its end, window of the last jump) is wave-uniform.
#define MEM_RP_WAVES 4
__global__ void __launch_bounds__(64 * MEM_RP_WAVES)
k_mem_replay(const MemReplayArgs r) {
    const uint32_t eb = blockIdx.x * MEM_RP_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (eb >= r.neb) return;
    const uint32_t i0 = r.eb_start[eb], i1 = r.eb_start[eb + 1];
    uint32_t in = MR_NONE;
    if (!r.first) {
        const uint32_t inc = r.eb_inc[eb];
        if (inc != MR_NONE && (r.ek[i0] >> 4) * r.k2 + r.K < r.run_dend[inc]) in = inc;   // else rule (b) cannot fire in here
        if (in == r.eb_in[eb]) return;
        if (lane == 0) *r.changed = 1u;
    }
    if (lane == 0) r.eb_in[eb] = in;
    uint32_t last = in, out = MR_NONE;
    uint64_t last_dend = in != MR_NONE ? r.run_dend[in] : 0;
    uint64_t jump_end = 0;       // windows below it are finished: the last jump skipped them (or one of their entries jumped)
    for (uint32_t base = i0; base < i1; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < i1;
        const uint64_t t = valid ? r.ek[i] >> 4 : 0;
        const uint32_t run = valid ? r.orun[i] : MR_NONE, fl = valid ? r.oflag[i] : 0u;
        const uint64_t qk = t * r.k2 + r.K;
        const bool is_long = (fl & MF_LONG) != 0, fresh = (fl & 15u) == 15u;
        const bool c_stale = is_long && !fresh, c_acc = is_long && fresh && (fl & (MF_L_EQ | MF_R_EQ)) != 0;
        uint64_t live = __ballot(valid && t >= jump_end);                                 // events of windows the scan still visits
        uint64_t mb = last != MR_NONE ? __ballot(run == last && qk < last_dend) : 0ull;    // (b)
        const uint64_t msa = __ballot(c_stale || c_acc), mstale = __ballot(c_stale);
        uint32_t oc = MO_NONE;
        for (;;) {
            const uint64_t todo = live & (mb | msa);
            if (!todo) break;
            const uint32_t l = (uint32_t)__builtin_ctzll(todo);                            // the next event that does something
            const uint64_t bit = 1ull << l;
            const uint64_t tl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t >> 32), l) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, l);
            if (mb & bit) {                                                               // (b): jump
                if (lane == l) oc = MO_JUMP;
                jump_end = tl + r.skip + 1;
                live &= __ballot(t >= jump_end);
            } else if (mstale & bit) {                                                    // until resolved: as if it failed (c)
                if (lane == l) oc = MO_STALE;
                live &= ~bit;
            } else {                                                                      // (c), (d): record, jump
                if (lane == l) oc = MO_ACCEPT;
                last = out = (uint32_t)__builtin_amdgcn_readlane((int)run, l);
                last_dend = r.run_dend[last];
                jump_end = tl + r.skip + 1;
                live &= __ballot(t >= jump_end);
                mb = __ballot(run == last && qk < last_dend);
            }
        }
        if (valid) r.outc[i] = (uint8_t)oc;
    }
    if (lane == 0) r.eb_out[eb] = out;
}

struct MemLastValid {          // scan operator: the rightmost recorded match
    __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return b != MR_NONE ? b : a; }
};
struct MemIsAccept {
    __device__ uint32_t operator()(uint8_t oc) const { return oc == MO_ACCEPT ? 1u : 0u; }
};

// smallest event index with outcome MO_STALE (everything before it is final)
__global__ void __launch_bounds__(256)
k_mem_first_stale(const uint8_t *__restrict__ outc, uint64_t nev, uint32_t *first) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += (uint64_t)gridDim.x * blockDim.x)
        if (outc[i] == MO_STALE) atomicMin(first, (uint32_t)i);
}
// the host's verdict on event x: its flags now read "all four contexts fresh, left ones equal / different"
__global__ void k_mem_resolve(uint32_t x, int pass, uint8_t *oflag, const uint32_t *lid, uint32_t *eb_in) {
    oflag[x] = (uint8_t)(15u | (pass ? MF_L_EQ : 0u) | (oflag[x] & MF_LONG));
    eb_in[lid[x] - 1u] = MR_DIRTY;
}
// [lo, hi) = the events of the windows [tlo, thi)
__global__ void k_mem_find_range(const uint64_t *__restrict__ ek, uint64_t nev, uint64_t tlo, uint64_t thi, uint64_t *out) {
    for (int k = 0; k < 2; k++) {
        const uint64_t key = (k ? thi : tlo) << 4;
        uint64_t lo = 0, hi = nev;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ek[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        out[k] = lo;
    }
}
// 5. the accepted events in discovery order (slot = inclusive count of accepted events - 1)
__global__ void __launch_bounds__(256)
k_mem_emit(const uint64_t *__restrict__ ek, const uint64_t *__restrict__ ep, const uint32_t *__restrict__ orun,
           const uint8_t *__restrict__ outc, const uint32_t *__restrict__ slot, uint64_t nev, uint32_t k2,
           const uint64_t *__restrict__ run_start, const uint64_t *__restrict__ run_end, pgrc_text_match *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nev || outc[i] != MO_ACCEPT) return;
    const uint32_t r = orun[i];
    const uint64_t ms = run_start[r], q = (ek[i] >> 4) * k2;
    pgrc_text_match tm;
    tm.pos_src = ms;
    tm.length = run_end[r] - ms;
    tm.pos_dest = q - (ep[i] - ms);
    out[slot[i] - 1u] = tm;
}

// ------------------------------------------------------------------------------------------------ host side

// maRushPrime1HashSparsified<K> over ASCII (matching/copmem/Hashes.h:54-76), for the few windows the host
// re-examines; the low 32 bits of the u64 fold are closed under xor / multiply
static uint32_t host_hash(int K, const char *s) {
    uint32_t h = (uint32_t)K;
    for (int j = 0; j < K / 4; j++) {
        uint32_t w;
        memcpy(&w, s + 4 * j, 4);
        w &= (j < 3) ? 0x00FFFFFFu : 0x0000FFFFu;
        h = (h ^ (w + (uint32_t)j)) * 171717u;
    }
    return h;
}
static uint32_t le32(const char *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

// The stale registers.  Near the ends of the texts the reference does not refresh l1 / r1 / l2 / r2 and compares against
// what an earlier bucket entry or window left there (:381-382, :401-402).  The device marks the first such event that the
// scan reaches and that could record a match (MO_STALE); everything before it is final, so the host walks back through
// the windows actually examined -- one block's jumps (the events with outcome MO_JUMP / MO_ACCEPT) and single bucket
// lookups, fetched from the device on demand -- to the entry that wrote the register last.
struct StaleWalk {
    pgrc_mem_ctx *m;
    const char *dest;
    uint64_t N2;
    bool dest_is_src, rev_compl;
    uint64_t skip, nmain;                        // probes jumped over after a hit; probes of the main loop (whole blocks of 256)
    const uint64_t *d_ek;                        // device: the events in the reference's order, their outcomes
    const uint8_t *d_outc;
    uint64_t nev;
    uint64_t *d_range;                           // device scratch, 2 words
    // the probes of one block that ended with a jump: (probe, order of the entry that caused it, 1 = recorded a match)
    struct Jump { uint64_t t; uint32_t order; uint32_t accepted; };
    uint64_t cached = ~0ull;
    std::vector<Jump> jumps;
    std::vector<uint64_t> hk;
    std::vector<uint8_t> ho;
    int lookup_err = 0;

    static bool jump_lt(const Jump &j, uint64_t v) { return j.t < v; }
    // the jumps of the block that probe t belongs to (a jump never leaves its block)
    void load_block(uint64_t t) {
        const uint64_t b = mem_block_of(t, nmain);
        if (b == cached) return;
        cached = b;
        jumps.clear();
        const uint64_t tlo = t < nmain ? t & ~255ull : nmain, thi = t < nmain ? tlo + 256 : (1ull << 59);
        pgrc_match_ctx *c = m->base;
        hipLaunchKernelGGL(k_mem_find_range, dim3(1), dim3(1), 0, c->stream, d_ek, nev, tlo, thi, d_range);
        uint64_t rg[2] = {0, 0};
        if (hipMemcpyAsync(rg, d_range, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { lookup_err = 1; return; }
        const uint64_t cnt = rg[1] - rg[0];
        if (!cnt) return;
        hk.resize(cnt);
        ho.resize(cnt);
        if (hipMemcpy(hk.data(), d_ek + rg[0], cnt * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ho.data(), d_outc + rg[0], cnt, hipMemcpyDeviceToHost) != hipSuccess) { lookup_err = 1; return; }
        for (uint64_t i = 0; i < cnt; i++)
            if (ho[i] == MO_JUMP || ho[i] == MO_ACCEPT) jumps.push_back({hk[i] >> 4, (uint32_t)(hk[i] & 15u), ho[i] == MO_ACCEPT ? 1u : 0u});
    }
    // was probe t visited?  (only asked for probes before the event being resolved: their outcomes are final)
    bool examined(uint64_t t) {
        load_block(t);
        auto it = std::lower_bound(jumps.begin(), jumps.end(), t, jump_lt);     // first jump at a probe >= t
        if (it == jumps.begin()) return true;
        --it;                                        // only the latest jump before t can still cover it
        return !(t <= it->t + skip);
    }
    const Jump *jump_at(uint64_t t) {
        load_block(t);
        auto it = std::lower_bound(jumps.begin(), jumps.end(), t, jump_lt);
        return (it != jumps.end() && it->t == t) ? &*it : nullptr;
    }
    // the bucket of a destination window, from the device (rare path)
    int bucket(uint64_t q, uint64_t *pos) {
        pgrc_match_ctx *c = m->base;
        const uint32_t h = host_hash(m->K, dest + q) & (uint32_t)(c->cp.hash_size - 1);
        unsigned long long hd[2];
        if (hipMemcpy(hd, (const char *)c->head_ptr + (size_t)head_slot(h, c->head_sh) * 16, 16, hipMemcpyDeviceToHost) != hipSuccess) { lookup_err = 1; return 0; }
        if (hd[0] == HEAD_EMPTY) return 0;
        pos[0] = (hd[0] & ENT_MASK) >> PGRC_FP_BITS;
        if (!(hd[0] & HEAD_OVF)) {
            if (hd[1] == HEAD_EMPTY) return 1;
            pos[1] = hd[1] >> PGRC_FP_BITS;
            return 2;
        }
        const int cnt = (int)((hd[1] >> 56) & 15u);
        unsigned long long e[PGRC_BUCKET_CAP];
        if (hipMemcpy(e, c->ent_ptr + (hd[1] & W1_BASE_MASK), (size_t)(cnt - 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) { lookup_err = 1; return 0; }
        for (int j = 1; j < cnt; j++) pos[j] = e[j - 1] >> PGRC_FP_BITS;
        return cnt;
    }
    bool self_filtered(uint64_t q, uint64_t p) const { return dest_is_src && (rev_compl ? N2 - p < q : q >= p); }

    // Value of register l1 (left = true) or r1 when the entry at bucket order `order` of probe t is about to be
    // tested and its own context lies outside the source: what the last entry examined before it left there.
    uint32_t stale_src_reg(bool left, uint64_t t, uint32_t order) {
        uint64_t pos[PGRC_BUCKET_CAP];
        uint64_t tt = t;
        uint32_t upto = order;                       // entries [0, upto) of probe tt were examined before
        for (;;) {
            const uint64_t q = tt * (uint64_t)m->k2;
            const int cnt = bucket(q, pos);
            for (int j = std::min<int>(cnt, (int)upto) - 1; j >= 0; j--) {
                const uint64_t p = pos[j];
                if (self_filtered(q, p)) continue;   // (a): never reached the register update
                if (left ? p >= (uint64_t)m->LK2 : p + (uint64_t)m->KLK24 + 4 <= m->N)
                    return le32(left ? m->src + p - m->LK2 : m->src + p + m->KLK24);
            }
            // previous examined probe
            for (;;) {
                if (tt == 0 || lookup_err) return 0u; // nothing before: the register still holds its initial 0
                tt--;
                if (examined(tt)) break;
            }
            const Jump *jp = jump_at(tt);
            // a probe that jumped: the entry that recorded a match had updated its registers first (:401-413), an
            // entry that triggered rule (b) had not (:393-399)
            upto = jp ? (jp->accepted ? jp->order + 1 : jp->order) : PGRC_BUCKET_CAP;
        }
    }
    // register r2 at probe t when t's own right context lies outside the destination (l2: always 0 there, the
    // windows before LK2 come first)
    uint32_t stale_r2(uint64_t t) {
        uint64_t pos[PGRC_BUCKET_CAP];
        uint64_t tt = t;
        while (tt > 0 && !lookup_err) {
            tt--;
            if (!examined(tt)) continue;
            const uint64_t q = tt * (uint64_t)m->k2;
            if (q + (uint64_t)m->KLK24 + 4 > N2) continue;
            if (bucket(q, pos) == 0) continue;       // empty bucket: the registers are not touched (:376-379)
            return le32(dest + q + m->KLK24);
        }
        return 0u;
    }
};

extern "C" {

const char *pgrc_mem_last_error(const pgrc_mem_ctx *m) { return m ? m->err.c_str() : g_mem_create_err.c_str(); }

int pgrc_mem_create(uint32_t target_len, uint32_t ctor_min_len, int32_t device, pgrc_mem_ctx **out) {
    if (!out) return PGRC_E_PARAM;
    *out = nullptr;
    if (target_len > 255) { g_mem_create_err = "target match length above 255 is not supported"; return PGRC_E_PARAM; }
    if (target_len < 24) { g_mem_create_err = "Minimal matching length too short"; return PGRC_E_SEED_SHORT; }   // CopMEMMatcher.cpp:77-80
    if (ctor_min_len < target_len) { g_mem_create_err = "a constructor minMatchLength below the target length is not supported"; return PGRC_E_PARAM; }
    pgrc_match_params prm;
    memset(&prm, 0, sizeof prm);
    prm.read_len = target_len;
    prm.seed_len = target_len;
    prm.mode = 'c';
    prm.device = device;
    pgrc_match_ctx *base = nullptr;
    int e = pgrc_match_create(&prm, &base);
    if (e) { g_mem_create_err = pgrc_match_last_error(nullptr); return e; }
    pgrc_mem_ctx *m = new pgrc_mem_ctx();
    m->base = base;
    m->L = target_len;
    *out = m;
    return PGRC_OK;
}

void pgrc_mem_destroy(pgrc_mem_ctx *m) {
    if (!m) return;
    {
        PgrcDeviceScope scope(m->base->device);
        (void)hipDeviceSynchronize();          // (its buffers go to the pool of device buffers: nothing may still be running)
    }
    DevBuf *bufs[] = {&m->d_dest, &m->d_nmap, &m->d_stage, &m->d_flag, &m->d_cursor, &m->d_evk[0], &m->d_evk[1], &m->d_evv[0],
                      &m->d_evv[1], &m->d_tmp, &m->d_scan, &m->d_orun, &m->d_oflag, &m->d_skey[0], &m->d_skey[1], &m->d_sidx[0],
                      &m->d_sidx[1], &m->d_first, &m->d_runid, &m->d_rstart, &m->d_rend, &m->d_rdend, &m->d_outc, &m->d_ebstart,
                      &m->d_ebin, &m->d_ebout, &m->d_ebinc, &m->d_small, &m->d_match};
    for (DevBuf *b : bufs) pgrc_buf_free(*b);
    if (m->have_ev)
        for (auto &x : m->ev) (void)hipEventDestroy(x);
    pgrc_match_destroy(m->base);
    delete m;
}

int pgrc_mem_get_counters(pgrc_mem_ctx *m, pgrc_mem_counters *out) {
    if (!m || !out) return PGRC_E_PARAM;
    *out = m->ctr;
    return PGRC_OK;
}

int pgrc_mem_set_src_ascii(pgrc_mem_ctx *m, const char *src, uint64_t n) {
    if (!m || !src) return PGRC_E_PARAM;
    pgrc_match_ctx *c = m->base;
    m->have_src = false;
    PgrcDeviceScope dev_scope__(c->device);
    if (!dev_scope__.ok) { m->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
    const auto t0 = std::chrono::steady_clock::now();
    int e = pgrc_match_set_pg_ascii(c, src, n);
    if (!e) e = pgrc_copmem_build_index(c, 0);
    if (!e && hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "index build failed"; e = PGRC_E_DEVICE; }
    if (e) { m->err = c->err; return e; }
    m->ctr.ms_index = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->K = c->cp.K; m->k1 = c->cp.k1; m->k2 = c->cp.k2;
    m->LK2 = ((int)m->L - m->K) / 2;                       // CopMEMMatcher.cpp:87-89
    m->KLK24 = m->K + m->LK2 - 4;
    m->src = src;
    m->N = n;
    m->have_src = true;
    return PGRC_OK;
}

void pgrc_mem_free_matches(pgrc_text_match *p) { free(p); }

int pgrc_mem_match_texts(pgrc_mem_ctx *m, const char *dest, uint64_t N2, int dest_is_src, int rev_compl, uint32_t min_len,
                         pgrc_text_match **matches, uint64_t *count) {
    if (!m || !dest || !matches || !count) return PGRC_E_PARAM;
    *matches = nullptr;
    *count = 0;
    if (!m->have_src) { m->err = "match_texts: set the source text first"; return PGRC_E_STATE; }
    pgrc_match_ctx *c = m->base;
    if ((int)min_len < m->K) { m->err = "Minimal matching length cannot be smaller than K"; return PGRC_E_PARAM; }   // :606-609
    if (dest_is_src && N2 != m->N) { m->err = "match_texts: dest_is_src with a text of another length"; return PGRC_E_PARAM; }
    if (N2 / (uint64_t)m->k2 + 1 >= (1ull << 40)) { m->err = "destination text too long"; return PGRC_E_PARAM; }
    PgrcDeviceScope dev_scope__(c->device);
    if (!dev_scope__.ok) { m->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
    m->ctr.probes = m->ctr.events = m->ctr.stale_lookups = 0;
    const uint64_t K = (uint64_t)m->K, k2 = (uint64_t)m->k2;
    const uint64_t nprobes = N2 >= K ? (N2 - K) / k2 + 1 : 0;          // windows q = t * k2 with q + K <= N2
    m->ctr.probes = nprobes;
    if (nprobes == 0) return PGRC_OK;
    int e;
    if (!m->have_ev) {
        for (auto &x : m->ev) MEM_TRY(m, hipEventCreate(&x));
        m->have_ev = true;
    }
    hipEvent_t *ev = m->ev;

    // ---- the destination in HBM
    MemArgs a;
    memset(&a, 0, sizeof a);
    const uint64_t dwords = (N2 + 15) / 16;
    if (dest_is_src) {
        if (rev_compl) {
            if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) { m->err = c->err; return e; }
            c->have_rc = true;
        }
        a.dest = (const uint32_t *)c->pg2[rev_compl ? 1 : 0].p;
        a.nmap = nullptr;
        a.dest_words_alloc = c->pg_words + PGRC_PG_PAD_WORDS;
    } else {
        const uint64_t CH = 64ull << 20;
        if ((e = pgrc_buf_ensure(c, m->d_dest, (dwords + PGRC_PG_PAD_WORDS) * 4)) || (e = pgrc_buf_ensure(c, m->d_nmap, (dwords + PGRC_PG_PAD_WORDS) * 2)) ||
            (e = pgrc_buf_ensure(c, m->d_stage, (size_t)std::min(CH, N2))) || (e = pgrc_buf_ensure(c, m->d_flag, 4))) { m->err = c->err; return e; }
        (void)hipMemsetAsync(m->d_dest.p, 0, (dwords + PGRC_PG_PAD_WORDS) * 4, c->stream);
        (void)hipMemsetAsync(m->d_nmap.p, 0, (dwords + PGRC_PG_PAD_WORDS) * 2, c->stream);
        (void)hipMemsetAsync(m->d_flag.p, 0, 4, c->stream);
        for (uint64_t off = 0; off < N2; off += CH) {
            const uint64_t len = std::min(CH, N2 - off);
            hipError_t he = hipMemcpyAsync(m->d_stage.p, dest + off, len, hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) {
                const uint64_t nw = (len + 15) / 16;
                hipLaunchKernelGGL(k_mem_pack, dim3((uint32_t)std::min<uint64_t>((nw + 255) / 256, 65536)), dim3(256), 0, c->stream,
                                   (const uint8_t *)m->d_stage.p, len, (uint32_t *)m->d_dest.p + off / 16, (uint16_t *)m->d_nmap.p + off / 16,
                                   (uint32_t *)m->d_flag.p);
                he = hipStreamSynchronize(c->stream);
            }
            if (he != hipSuccess) { m->err = std::string("destination upload: ") + hipGetErrorString(he); return PGRC_E_DEVICE; }
        }
        uint32_t fl = 0;
        MEM_TRY(m, hipMemcpy(&fl, m->d_flag.p, 4, hipMemcpyDeviceToHost));
        if (fl & 1u) { m->err = "destination text contains a symbol outside ACGNT"; return PGRC_E_SYMBOL; }
        a.dest = (const uint32_t *)m->d_dest.p;
        a.nmap = (fl & 2u) ? (const uint16_t *)m->d_nmap.p : nullptr;
        a.dest_words_alloc = dwords + PGRC_PG_PAD_WORDS;
    }
    a.src = (const uint32_t *)c->pg2[0].p;
    a.N = m->N;
    a.N2 = N2;
    a.head = (const ulonglong2 *)c->head_ptr;
    a.hsh = c->head_sh;
    a.ent = c->ent_ptr;
    a.mask = (uint32_t)(c->cp.hash_size - 1);
    a.K = (uint32_t)m->K;
    a.k2 = (uint32_t)m->k2;
    a.LK2 = (uint32_t)m->LK2;
    a.KLK24 = (uint32_t)m->KLK24;
    a.nprobes = nprobes;
    a.dest_is_src = dest_is_src ? 1 : 0;
    a.rev_compl = rev_compl ? 1 : 0;
    if (c->index_strand != 0) {          // (only if somebody rebuilt the base index in between)
        if ((e = pgrc_copmem_build_index(c, 0))) { m->err = c->err; return e; }
        a.head = (const ulonglong2 *)c->head_ptr;
        a.hsh = c->head_sh;
        a.ent = c->ent_ptr;
    }

    // ---- 1. events
    if ((e = pgrc_buf_ensure(c, m->d_cursor, 8))) { m->err = c->err; return e; }
    // room for the events (32 bytes each, in four arrays): one per probe for a short text -- a low-quality text against the high-quality
    // one makes 0.6 events per probe --, one per four probes for a long one (a text against itself, forward: 0.18; against its reverse
    // complement, the encoder's call: 0.004); a probe that finds more is run again with what it counted (rounds 1-5a started at one
    // event per 64 probes: the event-rich calls probed twice the first time)
    uint64_t cap = std::max<uint64_t>((nprobes <= (64ull << 20) ? nprobes : nprobes / 4) + 65536, m->d_evk[0].bytes / 8);
    if (c->opt.mem_event_cap) cap = c->opt.mem_event_cap;   // (PGRC_MEM_EVENT_CAP, test knob: start tiny to exercise the regrow-and-rerun path)
    unsigned long long nev = 0;
    (void)hipEventRecord(ev[0], c->stream);
    for (int attempt = 0; attempt < 2; attempt++) {
        for (int k = 0; k < 2; k++)
            if ((e = pgrc_buf_ensure(c, m->d_evk[k], cap * 8)) || (e = pgrc_buf_ensure(c, m->d_evv[k], cap * 8))) { m->err = c->err; return e; }
        (void)hipMemsetAsync(m->d_cursor.p, 0, 8, c->stream);
        hipLaunchKernelGGL(k_mem_probe, dim3((uint32_t)((nprobes + MEM_TPB - 1) / MEM_TPB)), dim3(MEM_TPB), 0, c->stream, a,
                           (unsigned long long *)m->d_cursor.p, (uint64_t *)m->d_evk[0].p, (uint64_t *)m->d_evv[0].p, cap);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipMemcpyAsync(&nev, m->d_cursor.p, 8, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { m->err = std::string("probe kernel: ") + hipGetErrorString(he); return PGRC_E_DEVICE; }
        if (nev <= cap) break;
        cap = nev + nev / 4;                                // the guess was too small: once more, with headroom for the next call
    }
    (void)hipEventRecord(ev[1], c->stream);
    m->ctr.events = nev;
    m->ctr.replay_rounds = m->ctr.event_blocks = 0;
    m->ctr.ms_sort = m->ctr.ms_extend = m->ctr.ms_replay = m->ctr.ms_host = 0.f;
    (void)hipEventSynchronize(ev[1]);
    (void)hipEventElapsedTime(&m->ctr.ms_probe, ev[0], ev[1]);
    if (!nev) return PGRC_OK;
    if (nev >= 0xFFFFFFF0ull) { m->err = "more than 2^32 events"; return PGRC_E_PARAM; }

    // ---- 2. the order in which the reference meets them: by window, then by bucket order
    int tb = 1;
    while ((1ull << tb) < nprobes) tb++;
    // (a stable LSD sort of (key, value) pairs: radix.hip, hand-written; the library's pair sort until round 5)
    uint64_t *ksorted = nullptr, *vsorted = nullptr;
    if ((e = pgrc_radix_sort_pairs_u64(c, (uint64_t *)m->d_evk[0].p, (uint64_t *)m->d_evk[1].p, (uint64_t *)m->d_evv[0].p, (uint64_t *)m->d_evv[1].p, nev, 0,
                                       4 + tb, m->d_tmp, &ksorted, &vsorted))) { m->err = c->err; return e; }
    (void)hipEventRecord(ev[2], c->stream);
    const uint64_t *ek = ksorted, *ep = vsorted;

    // ---- 3. side contexts; extents per run of connected events on a diagonal
    const uint64_t ecap = nev + nev / 8 + 4096;               // per-event arrays: by the events there are (with room for a similar call), not by the first guess
    if ((e = pgrc_buf_ensure(c, m->d_orun, ecap * 4)) || (e = pgrc_buf_ensure(c, m->d_oflag, ecap)) || (e = pgrc_buf_ensure(c, m->d_first, ecap * 4)) ||
        (e = pgrc_buf_ensure(c, m->d_runid, ecap * 4)) || (e = pgrc_buf_ensure(c, m->d_rstart, ecap * 8)) || (e = pgrc_buf_ensure(c, m->d_rend, ecap * 8)) ||
        (e = pgrc_buf_ensure(c, m->d_rdend, ecap * 8)) || (e = pgrc_buf_ensure(c, m->d_outc, ecap)) || (e = pgrc_buf_ensure(c, m->d_small, 64))) { m->err = c->err; return e; }
    for (int k = 0; k < 2; k++)
        if ((e = pgrc_buf_ensure(c, m->d_skey[k], ecap * 8)) || (e = pgrc_buf_ensure(c, m->d_sidx[k], ecap * 8))) { m->err = c->err; return e; }
    // d_small: [0] "a block was replayed" flag, [1] first stale event, [2] events with a context outside a text, [4..7] two u64 of a range
    uint32_t *d_changed = (uint32_t *)m->d_small.p, *d_first_stale = d_changed + 1, *d_nstale = d_changed + 2;
    uint64_t *d_range = (uint64_t *)m->d_small.p + 2;
    MEM_TRY(m, hipMemsetAsync(m->d_small.p, 0, 64, c->stream));
    const uint32_t g = (uint32_t)((nev + 255) / 256);
    // scratch of the scans below (scanops.h): block folds of nev values
    if ((e = pgrc_buf_ensure(c, m->d_scan, sco_scratch_words(nev) * sizeof(uint32_t)))) { m->err = c->err; return e; }
    uint32_t *d_bsum = (uint32_t *)m->d_scan.p;
    {
        uint64_t *sk0 = (uint64_t *)m->d_skey[0].p, *si0 = (uint64_t *)m->d_sidx[0].p;
        hipLaunchKernelGGL(k_mem_flags, dim3(g), dim3(256), 0, c->stream, a, ek, ep, (uint64_t)nev, (uint8_t *)m->d_oflag.p, sk0, si0, d_nstale);
        int db = 1;
        while ((1ull << db) < N2 + m->N) db++;
        // (diagonal, window) order: the events ARE in window order (step 2), so one stable sort by diagonal does it (until round 4 a
        // sort by window came first: a third of the sorting time of an event-rich call, for nothing)
        hipLaunchKernelGGL(k_mem_diag, dim3(g), dim3(256), 0, c->stream, a, ek, ep, (const uint64_t *)si0, (uint64_t)nev, sk0);
        uint64_t *sks = nullptr, *sis = nullptr;
        if ((e = pgrc_radix_sort_pairs_u64(c, sk0, (uint64_t *)m->d_skey[1].p, si0, (uint64_t *)m->d_sidx[1].p, nev, 0, (uint32_t)db, m->d_tmp, &sks, &sis))) { m->err = c->err; return e; }
        const uint64_t *sidx = sis, *skey = sks;
        hipLaunchKernelGGL(k_mem_connect, dim3(g), dim3(256), 0, c->stream, a, ek, ep, sidx, skey, (uint64_t)nev, (uint32_t *)m->d_first.p);
        MEM_TRY(m, (sco_scan<true>(c->stream, (const uint32_t *)m->d_first.p, (uint32_t *)m->d_runid.p, (uint64_t)nev, ScoIdentity(), ScoPlus(), 0u, d_bsum)));
        hipLaunchKernelGGL(k_mem_run_ends, dim3(g), dim3(256), 0, c->stream, a, ek, ep, sidx, (const uint32_t *)m->d_first.p,
                           (const uint32_t *)m->d_runid.p, (uint64_t)nev, (uint64_t *)m->d_rstart.p, (uint64_t *)m->d_rend.p, (uint64_t *)m->d_rdend.p);
        hipLaunchKernelGGL(k_mem_apply, dim3(g), dim3(256), 0, c->stream, sidx, (const uint32_t *)m->d_runid.p, (uint64_t)nev,
                           (const uint64_t *)m->d_rstart.p, (const uint64_t *)m->d_rend.p, min_len, (uint32_t *)m->d_orun.p, (uint8_t *)m->d_oflag.p);
        MEM_TRY(m, hipGetLastError());
    }
    (void)hipEventRecord(ev[3], c->stream);

    // ---- 4. the sequential rules on the device (k_mem_replay)
    uint64_t nmain;
    {   // main loop: blocks of 256 windows while i1 + K + 256 * k2 < N2 + 1 (:365)
        const uint64_t block = 256 * k2;
        const uint64_t lim = N2 + 1;                                               // i1 + K + block < lim
        uint64_t nb = 0;
        if (lim > K + block) nb = (lim - K - block + block - 1) / block;
        nmain = nb * 256;
    }
    uint32_t *d_lead = (uint32_t *)m->d_first.p, *d_lid = (uint32_t *)m->d_runid.p;      // (free again: the runs are numbered)
    hipLaunchKernelGGL(k_mem_lead, dim3(g), dim3(256), 0, c->stream, ek, (uint64_t)nev, nmain, d_lead);
    MEM_TRY(m, (sco_scan<true>(c->stream, (const uint32_t *)d_lead, d_lid, (uint64_t)nev, ScoIdentity(), ScoPlus(), 0u, d_bsum)));
    uint32_t neb = 0, nstale = 0;
    MEM_TRY(m, hipMemcpyAsync(&neb, d_lid + (nev - 1), 4, hipMemcpyDeviceToHost, c->stream));
    MEM_TRY(m, hipMemcpyAsync(&nstale, d_nstale, 4, hipMemcpyDeviceToHost, c->stream));
    MEM_TRY(m, hipStreamSynchronize(c->stream));
    m->ctr.event_blocks = neb;
    if ((e = pgrc_buf_ensure(c, m->d_ebstart, ((size_t)neb + 1) * 4)) || (e = pgrc_buf_ensure(c, m->d_ebin, (size_t)neb * 4)) ||
        (e = pgrc_buf_ensure(c, m->d_ebout, (size_t)neb * 4)) || (e = pgrc_buf_ensure(c, m->d_ebinc, (size_t)neb * 4))) { m->err = c->err; return e; }
    hipLaunchKernelGGL(k_mem_eb_start, dim3(g), dim3(256), 0, c->stream, (const uint32_t *)d_lead, (const uint32_t *)d_lid, (uint64_t)nev, (uint32_t *)m->d_ebstart.p);
    MemReplayArgs ra;
    ra.ek = ek;
    ra.orun = (const uint32_t *)m->d_orun.p;
    ra.oflag = (const uint8_t *)m->d_oflag.p;
    ra.outc = (uint8_t *)m->d_outc.p;
    ra.run_dend = (const uint64_t *)m->d_rdend.p;
    ra.eb_start = (const uint32_t *)m->d_ebstart.p;
    ra.eb_in = (uint32_t *)m->d_ebin.p;
    ra.eb_out = (uint32_t *)m->d_ebout.p;
    ra.eb_inc = (const uint32_t *)m->d_ebinc.p;
    ra.neb = neb;
    ra.K = (uint32_t)K;
    ra.k2 = (uint32_t)k2;
    ra.skip = (uint32_t)(m->K / m->k1 - 1);                                            // :352
    ra.changed = d_changed;
    ra.first = 1;
    const uint32_t gb = (neb + MEM_RP_WAVES - 1) / MEM_RP_WAVES;
    hipLaunchKernelGGL(k_mem_replay, dim3(gb), dim3(64 * MEM_RP_WAVES), 0, c->stream, ra);
    m->ctr.replay_rounds = 1;
    ra.first = 0;
    // rounds until no block's incoming match differs from what it assumed
    auto settle = [&]() -> int {
        for (;;) {
            uint32_t changed = 0;
            MEM_TRY(m, (sco_scan<false>(c->stream, (const uint32_t *)m->d_ebout.p, (uint32_t *)m->d_ebinc.p, (uint64_t)neb, ScoIdentity(), MemLastValid(), MR_NONE, d_bsum)));
            MEM_TRY(m, hipMemsetAsync(d_changed, 0, 4, c->stream));
            hipLaunchKernelGGL(k_mem_replay, dim3(gb), dim3(64 * MEM_RP_WAVES), 0, c->stream, ra);
            MEM_TRY(m, hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, c->stream));
            MEM_TRY(m, hipStreamSynchronize(c->stream));
            if (!changed) return PGRC_OK;
            m->ctr.replay_rounds++;
        }
    };
    if ((e = settle())) return e;
    // events whose registers are stale, in order: the first one reached is decided on the host, then the replay goes on
    while (nstale) {
        uint32_t x = MR_NONE;
        MEM_TRY(m, hipMemsetAsync(d_first_stale, 0xFF, 4, c->stream));
        hipLaunchKernelGGL(k_mem_first_stale, dim3((uint32_t)std::min<uint64_t>((nev + 255) / 256, 4096)), dim3(256), 0, c->stream,
                           (const uint8_t *)m->d_outc.p, (uint64_t)nev, d_first_stale);
        MEM_TRY(m, hipMemcpyAsync(&x, d_first_stale, 4, hipMemcpyDeviceToHost, c->stream));
        MEM_TRY(m, hipStreamSynchronize(c->stream));
        if (x == MR_NONE) break;
        uint64_t key = 0, p = 0;
        uint8_t fl8 = 0;
        MEM_TRY(m, hipMemcpy(&key, ek + x, 8, hipMemcpyDeviceToHost));
        MEM_TRY(m, hipMemcpy(&p, ep + x, 8, hipMemcpyDeviceToHost));
        MEM_TRY(m, hipMemcpy(&fl8, (const uint8_t *)m->d_oflag.p + x, 1, hipMemcpyDeviceToHost));
        StaleWalk sw;
        sw.m = m; sw.dest = dest; sw.N2 = N2; sw.dest_is_src = dest_is_src != 0; sw.rev_compl = rev_compl != 0;
        sw.skip = ra.skip; sw.nmain = nmain; sw.d_ek = ek; sw.d_outc = (const uint8_t *)m->d_outc.p; sw.nev = nev; sw.d_range = d_range;
        const uint64_t t = key >> 4, q = t * k2;
        const uint32_t order = (uint32_t)(key & 15u), fl = fl8;
        m->ctr.stale_lookups++;
        // (c) with the registers as the reference holds them (:401-404)
        const uint32_t l1 = (fl & MF_L1_OK) ? le32(m->src + p - m->LK2) : sw.stale_src_reg(true, t, order);
        const uint32_t r1 = (fl & MF_R1_OK) ? le32(m->src + p + m->KLK24) : sw.stale_src_reg(false, t, order);
        const uint32_t l2 = (fl & MF_L2_OK) ? le32(dest + q - m->LK2) : 0u;   // windows below LK2 come first: still the initial 0
        const uint32_t r2 = (fl & MF_R2_OK) ? le32(dest + q + m->KLK24) : sw.stale_r2(t);
        if (sw.lookup_err) { m->err = "bucket lookup failed"; return PGRC_E_DEVICE; }
        hipLaunchKernelGGL(k_mem_resolve, dim3(1), dim3(1), 0, c->stream, x, (r1 == r2 || l1 == l2) ? 1 : 0, (uint8_t *)m->d_oflag.p,
                           (const uint32_t *)d_lid, (uint32_t *)m->d_ebin.p);
        if ((e = settle())) return e;
    }
    (void)hipEventRecord(ev[4], c->stream);

    // ---- 5. the matches, in discovery order
    const auto th0 = std::chrono::steady_clock::now();
    uint32_t *d_slot = (uint32_t *)m->d_first.p;                                       // (the block leaders are not needed any more)
    MEM_TRY(m, (sco_scan<true>(c->stream, (const uint8_t *)m->d_outc.p, d_slot, (uint64_t)nev, MemIsAccept(), ScoPlus(), 0u, d_bsum)));
    uint32_t nmatch = 0;
    MEM_TRY(m, hipMemcpyAsync(&nmatch, d_slot + (nev - 1), 4, hipMemcpyDeviceToHost, c->stream));
    MEM_TRY(m, hipStreamSynchronize(c->stream));
    if (nmatch) {
        if ((e = pgrc_buf_ensure(c, m->d_match, (size_t)nmatch * sizeof(pgrc_text_match)))) { m->err = c->err; return e; }
        hipLaunchKernelGGL(k_mem_emit, dim3(g), dim3(256), 0, c->stream, ek, ep, (const uint32_t *)m->d_orun.p, (const uint8_t *)m->d_outc.p,
                           (const uint32_t *)d_slot, (uint64_t)nev, (uint32_t)k2, (const uint64_t *)m->d_rstart.p, (const uint64_t *)m->d_rend.p,
                           (pgrc_text_match *)m->d_match.p);
        pgrc_text_match *outp = (pgrc_text_match *)malloc((size_t)nmatch * sizeof(pgrc_text_match));
        if (!outp) { m->err = "out of host memory"; return PGRC_E_ALLOC; }
        hipError_t he = hipMemcpyAsync(outp, m->d_match.p, (size_t)nmatch * sizeof(pgrc_text_match), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { free(outp); m->err = std::string("match download: ") + hipGetErrorString(he); return PGRC_E_DEVICE; }
        *matches = outp;
    }
    *count = nmatch;
    (void)hipEventElapsedTime(&m->ctr.ms_sort, ev[1], ev[2]);
    (void)hipEventElapsedTime(&m->ctr.ms_extend, ev[2], ev[3]);
    (void)hipEventElapsedTime(&m->ctr.ms_replay, ev[3], ev[4]);                       // (with the host's part between the rounds)
    m->ctr.ms_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - th0).count();
    return PGRC_OK;
}

} // extern "C"
