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
//   4. host          : one pass over the events replays (b), the jumps (which never leave a block of 256 windows in
//                      the main loop, :365-424, but carry through in the tail loop, :428-476) and the acceptance.
// The side-context registers l1/r1/l2/r2 are refreshed only when their 4 bytes lie inside the text (:381-382,
// :401-402): for the handful of events at the text ends the host re-creates the stale value the reference would
// hold by walking back through the windows actually examined (bucket lookups from the device on demand).
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

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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
    DevBuf d_dest, d_nmap, d_stage, d_flag, d_cursor, d_evk[2], d_evv[2], d_tmp, d_orun, d_oflag;
    DevBuf d_skey[2], d_sidx[2], d_first, d_runid, d_rstart, d_rend;   // events by (diagonal, window): sort ping-pong, runs
    // pinned, grow-only host mirrors of the event arrays (a std::vector would zero-fill gigabytes per call)
    struct HostBuf { void *p = nullptr; size_t bytes = 0; bool pinned = false; } h_key, h_pos, h_run, h_rstart, h_rend, h_flag;
    hipEvent_t ev[4]{};               // phase timing (created on first use)
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

static void host_release(pgrc_mem_ctx::HostBuf &b) {
    if (b.p) {
        if (b.pinned) (void)hipHostFree(b.p);
        else free(b.p);
    }
    b.p = nullptr;
    b.bytes = 0;
}

static int host_ensure(pgrc_mem_ctx *m, pgrc_mem_ctx::HostBuf &b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return PGRC_OK;
    host_release(b);
    if (hipHostMalloc(&b.p, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess) {
        b.pinned = true;
    } else {                              // no pinned memory left: pageable memory works too, the copies are just slower
        (void)hipGetLastError();
        b.p = malloc(bytes ? bytes : 16);
        b.pinned = false;
        if (!b.p) {
            m->err = "host allocation of " + std::to_string(bytes) + " bytes failed";
            return PGRC_E_ALLOC;
        }
    }
    b.bytes = bytes;
    return PGRC_OK;
}

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
    const ulonglong2 *head;
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
#define MEM_LCAP 2048u

// 1. one thread per destination window
__global__ void __launch_bounds__(MEM_TPB)
k_mem_probe(const MemArgs a, unsigned long long *cursor, uint64_t *__restrict__ evk, uint64_t *__restrict__ evv, uint64_t cap) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t tile[MEM_TILE_WORDS];
    __shared__ uint64_t lk[MEM_LCAP], lv[MEM_LCAP];
    __shared__ uint32_t lcount;
    __shared__ unsigned long long gbase;
    hash_lut_init(lut);
    if (threadIdx.x == 0) lcount = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * MEM_TPB;
    const uint64_t q0 = t0 * a.k2;
    const uint64_t w0 = q0 >> 4;
    const uint32_t need = (uint32_t)((((q0 & 15) + (uint64_t)(MEM_TPB - 1) * a.k2 + a.K + 15) >> 4) + 5);
    for (uint32_t w = threadIdx.x; w < need; w += MEM_TPB) tile[w] = (w0 + w < a.dest_words_alloc) ? a.dest[w0 + w] : 0u;
    __syncthreads();
    const uint64_t t = t0 + threadIdx.x;
    if (t < a.nprobes) {
        const uint64_t q = t * a.k2;
        const uint32_t x = (uint32_t)((q >> 4) - w0);
        const uint32_t sh = ((uint32_t)q & 15u) * 2u;
        const uint32_t d0 = funnel_r(tile[x], tile[x + 1], sh), d1 = funnel_r(tile[x + 1], tile[x + 2], sh),
                       d2 = funnel_r(tile[x + 2], tile[x + 3], sh), d3 = funnel_r(tile[x + 3], tile[x + 4], sh);
        bool has_n = false;
        if (a.nmap)
            for (uint32_t k = 0; k < a.K; k += 16) {
                const uint32_t nb = nbits16(a.nmap, q + k);
                has_n |= (nb & (a.K - k >= 16 ? 0xFFFFu : ((1u << (a.K - k)) - 1u))) != 0;
            }
        if (!has_n) {       // a window with an 'N' equals no source K-mer: it can produce no event
            uint32_t fp;
            const uint32_t h = copmem_hash32_fp(d0, d1, d2, d3, a.K, lut, &fp) & a.mask;
            const ulonglong2 hd = a.head[h];
            const uint32_t cnt = head_count(hd);
            const uint64_t base = hd.y & W1_BASE_MASK;
            const uint32_t dw[4] = {d0, d1, d2, d3};
            for (uint32_t j = 0; j < cnt; j++) {
                const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (cnt == 2 ? hd.y : a.ent[base + j - 1]);
                const uint64_t p = e >> PGRC_FP_BITS;
                if (a.dest_is_src && (a.rev_compl ? a.N2 - p < q : q >= p)) continue;              // :389-392
                if (((uint32_t)e ^ fp) & ((1u << PGRC_FP_BITS) - 1u)) continue;                     // K-mers differ
                bool equal = true;
                for (uint32_t k = 0; k < a.K; k += 16) {
                    uint32_t dx = sym16(a.src, p + k) ^ dw[k >> 4];
                    if (a.K - k < 16) dx &= (1u << (2 * (a.K - k))) - 1u;
                    equal &= dx == 0;
                }
                if (!equal) continue;
                const uint32_t li = atomicAdd(&lcount, 1u);
                if (li < MEM_LCAP) {
                    lk[li] = (t << 4) | j;
                    lv[li] = p;
                } else {                                  // buffer full (low-complexity text): straight to HBM
                    const unsigned long long idx = atomicAdd(cursor, 1ull);
                    if (idx < cap) { evk[idx] = (t << 4) | j; evv[idx] = p; }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nl = min(lcount, MEM_LCAP);
    if (threadIdx.x == 0 && nl) gbase = atomicAdd(cursor, (unsigned long long)nl);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nl; i += MEM_TPB)
        if (gbase + i < cap) { evk[gbase + i] = lk[i]; evv[gbase + i] = lv[i]; }
}

// flags of an event
#define MF_L1_OK 1u    // the source side context lies inside the source text (else the register is stale, :401-402)
#define MF_R1_OK 2u
#define MF_L2_OK 4u    // the same for the destination (:381-382)
#define MF_R2_OK 8u
#define MF_L_EQ 16u    // both left contexts fresh and equal
#define MF_R_EQ 32u

// 3. side contexts and extensions.  Every event of one maximal match (same diagonal, connected by equal symbols) has the
// same extents, and a long match carries one event per lcm(k1, k2) symbols: extending each of them separately would be
// quadratic in the match length (a 10 Mbp duplicate: 10^12 symbol compares).  So the events are ordered by (diagonal,
// window) once, neighbours on a diagonal are tested for being CONNECTED (only the gap between their K-mers is
// compared, usually nothing: the K-mers overlap), a prefix sum numbers the runs, the first event of a run extends to
// the left, the last one to the right, and everybody takes its run's extents: linear in the text.
__global__ void __launch_bounds__(256)
k_mem_flags(const MemArgs a, const uint64_t *__restrict__ evk, const uint64_t *__restrict__ evv, uint64_t nev,
            uint8_t *__restrict__ oflag, uint64_t *__restrict__ qkey, uint64_t *__restrict__ idx) {
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
               uint64_t *__restrict__ run_start, uint64_t *__restrict__ run_end) {
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

// every event learns its run (the host reads the extents per run: 4 bytes per event instead of 16)
__global__ void __launch_bounds__(256)
k_mem_apply(const uint64_t *__restrict__ idx, const uint32_t *__restrict__ runid, uint64_t nev, uint32_t *__restrict__ orun) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nev) return;
    orun[idx[k]] = runid[k] - 1u;
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

struct Replay {
    pgrc_mem_ctx *m;
    const char *dest;
    uint64_t N2;
    bool dest_is_src, rev_compl;
    uint64_t skip, nmain;                        // probes jumped over after a hit; probes of the main loop (whole blocks of 256)
    // history of the probes that ended with a jump: (probe, order of the entry that caused it, 1 = recorded a match)
    struct Jump { uint64_t t; uint32_t order; uint32_t accepted; };
    // the jumps recorded so far, in probe order: the finished chunks' lists (none empty, ascending) and the list of the
    // chunk being replayed
    std::vector<const std::vector<Jump> *> hist;
    const std::vector<Jump> *cur = nullptr;
    int lookup_err = 0;

    uint64_t block_of(uint64_t t) const { return t < nmain ? t / 256 : nmain / 256 + 1; }   // the tail loop is one block
    static bool jump_lt(const Jump &j, uint64_t v) { return j.t < v; }
    // the list that holds the latest jump at a probe < t (or <= t), if any
    const std::vector<Jump> *list_for(uint64_t t, bool inclusive) const {
        if (cur && !cur->empty() && (inclusive ? cur->front().t <= t : cur->front().t < t)) return cur;
        size_t lo = 0, hi = hist.size();                 // first list whose first jump lies beyond t
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const uint64_t f = hist[mid]->front().t;
            if (inclusive ? f <= t : f < t) lo = mid + 1;
            else hi = mid;
        }
        return lo ? hist[lo - 1] : nullptr;
    }
    // was probe t visited, given the jumps recorded so far?
    bool examined(uint64_t t) const {
        // only the latest jump before t can still cover it (jumps are skip probes long, and recorded in order)
        const std::vector<Jump> *v = list_for(t, false);
        if (!v) return true;
        auto it = std::lower_bound(v->begin(), v->end(), t, jump_lt);     // > begin: the list's first jump is before t
        --it;
        return !(block_of(it->t) == block_of(t) && t <= it->t + skip);
    }
    const Jump *jump_at(uint64_t t) const {
        const std::vector<Jump> *v = list_for(t, true);
        if (!v) return nullptr;
        auto it = std::lower_bound(v->begin(), v->end(), t, jump_lt);
        return (it != v->end() && it->t == t) ? &*it : nullptr;
    }
    // the bucket of a destination window, from the device (rare path)
    int bucket(uint64_t q, uint64_t *pos) {
        pgrc_match_ctx *c = m->base;
        const uint32_t h = host_hash(m->K, dest + q) & (uint32_t)(c->cp.hash_size - 1);
        unsigned long long hd[2];
        if (hipMemcpy(hd, (const char *)c->d_head.p + (size_t)h * 16, 16, hipMemcpyDeviceToHost) != hipSuccess) { lookup_err = 1; return 0; }
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
                if (tt == 0) return 0u;              // nothing before: the register still holds its initial 0
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
        while (tt > 0) {
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
    DevBuf *bufs[] = {&m->d_dest, &m->d_nmap, &m->d_stage, &m->d_flag, &m->d_cursor, &m->d_evk[0], &m->d_evk[1], &m->d_evv[0],
                      &m->d_evv[1], &m->d_tmp, &m->d_orun, &m->d_oflag, &m->d_skey[0], &m->d_skey[1], &m->d_sidx[0],
                      &m->d_sidx[1], &m->d_first, &m->d_runid, &m->d_rstart, &m->d_rend};
    for (DevBuf *b : bufs) pgrc_buf_free(*b);
    if (m->have_ev)
        for (auto &x : m->ev) (void)hipEventDestroy(x);
    for (pgrc_mem_ctx::HostBuf *h : {&m->h_key, &m->h_pos, &m->h_run, &m->h_rstart, &m->h_rend, &m->h_flag}) host_release(*h);
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
    a.head = (const ulonglong2 *)c->d_head.p;
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
        a.head = (const ulonglong2 *)c->d_head.p;
        a.ent = c->ent_ptr;
    }

    // ---- 1. events
    if ((e = pgrc_buf_ensure(c, m->d_cursor, 8))) { m->err = c->err; return e; }
    uint64_t cap = std::max<uint64_t>(nprobes / 64 + 65536, m->d_evk[0].bytes / 8);
    if (const char *ev_cap = getenv("PGRC_MEM_EVENT_CAP"))   // test knob: start tiny to exercise the regrow-and-rerun path
        cap = std::max<uint64_t>(1, (uint64_t)atoll(ev_cap));
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
    if ((e = host_ensure(m, m->h_key, nev * 8)) || (e = host_ensure(m, m->h_pos, nev * 8)) || (e = host_ensure(m, m->h_run, nev * 4)) ||
        (e = host_ensure(m, m->h_flag, nev))) { return e; }
    const uint64_t *hk = (const uint64_t *)m->h_key.p, *hp = (const uint64_t *)m->h_pos.p;
    const uint32_t *hr = (const uint32_t *)m->h_run.p;          // run of an event; the extents are per run
    const uint64_t *hrs = nullptr, *hre = nullptr;
    const uint8_t *hf = (const uint8_t *)m->h_flag.p;
    if (nev) {
        // ---- 2. the order in which the reference meets them: by window, then by bucket order
        int tb = 1;
        while ((1ull << tb) < nprobes) tb++;
        rocprim::double_buffer<uint64_t> keys((uint64_t *)m->d_evk[0].p, (uint64_t *)m->d_evk[1].p);
        rocprim::double_buffer<uint64_t> vals((uint64_t *)m->d_evv[0].p, (uint64_t *)m->d_evv[1].p);
        size_t tbytes = 0;
        hipError_t he = rocprim::radix_sort_pairs(nullptr, tbytes, keys, vals, (size_t)nev, 0, 4 + tb, c->stream);
        if (he == hipSuccess && (e = pgrc_buf_ensure(c, m->d_tmp, tbytes + 16))) { m->err = c->err; return e; }
        if (he == hipSuccess) he = rocprim::radix_sort_pairs(m->d_tmp.p, tbytes, keys, vals, (size_t)nev, 0, 4 + tb, c->stream);
        (void)hipEventRecord(ev[2], c->stream);
        // ---- 3. side contexts; extents per run of connected events on a diagonal
        if (nev >= (1ull << 32)) { m->err = "more than 2^32 events"; return PGRC_E_PARAM; }
        if (he == hipSuccess && ((e = pgrc_buf_ensure(c, m->d_orun, cap * 4)) ||
                                 (e = pgrc_buf_ensure(c, m->d_oflag, cap)) || (e = pgrc_buf_ensure(c, m->d_first, cap * 4)) ||
                                 (e = pgrc_buf_ensure(c, m->d_runid, cap * 4)) || (e = pgrc_buf_ensure(c, m->d_rstart, cap * 8)) ||
                                 (e = pgrc_buf_ensure(c, m->d_rend, cap * 8)))) { m->err = c->err; return e; }
        for (int k = 0; k < 2 && he == hipSuccess; k++)
            if ((e = pgrc_buf_ensure(c, m->d_skey[k], cap * 8)) || (e = pgrc_buf_ensure(c, m->d_sidx[k], cap * 8))) { m->err = c->err; return e; }
        if (he == hipSuccess) {
            const uint32_t g = (uint32_t)((nev + 255) / 256);
            const uint64_t *ek = (const uint64_t *)keys.current(), *ep = (const uint64_t *)vals.current();
            rocprim::double_buffer<uint64_t> sk((uint64_t *)m->d_skey[0].p, (uint64_t *)m->d_skey[1].p);
            rocprim::double_buffer<uint64_t> si((uint64_t *)m->d_sidx[0].p, (uint64_t *)m->d_sidx[1].p);
            hipLaunchKernelGGL(k_mem_flags, dim3(g), dim3(256), 0, c->stream, a, ek, ep, (uint64_t)nev, (uint8_t *)m->d_oflag.p, sk.current(), si.current());
            int qb = 1, db = 1;
            while ((1ull << qb) < N2) qb++;
            while ((1ull << db) < N2 + m->N) db++;
            size_t t1 = 0, t2 = 0, t3 = 0;
            he = rocprim::radix_sort_pairs(nullptr, t1, sk, si, (size_t)nev, 0, qb, c->stream);
            if (he == hipSuccess) he = rocprim::radix_sort_pairs(nullptr, t2, sk, si, (size_t)nev, 0, db, c->stream);
            if (he == hipSuccess) he = rocprim::inclusive_scan(nullptr, t3, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)nev, rocprim::plus<uint32_t>(), c->stream);
            if (he == hipSuccess && (e = pgrc_buf_ensure(c, m->d_tmp, std::max(std::max(t1, t2), std::max(t3, tbytes)) + 16))) { m->err = c->err; return e; }
            // stable sorts: by window first, then by diagonal => (diagonal, window)
            if (he == hipSuccess) he = rocprim::radix_sort_pairs(m->d_tmp.p, t1, sk, si, (size_t)nev, 0, qb, c->stream);
            if (he == hipSuccess) {
                hipLaunchKernelGGL(k_mem_diag, dim3(g), dim3(256), 0, c->stream, a, ek, ep, (const uint64_t *)si.current(), (uint64_t)nev, sk.current());
                he = rocprim::radix_sort_pairs(m->d_tmp.p, t2, sk, si, (size_t)nev, 0, db, c->stream);
            }
            if (he == hipSuccess) {
                const uint64_t *sidx = si.current(), *skey = sk.current();
                hipLaunchKernelGGL(k_mem_connect, dim3(g), dim3(256), 0, c->stream, a, ek, ep, sidx, skey, (uint64_t)nev, (uint32_t *)m->d_first.p);
                he = rocprim::inclusive_scan(m->d_tmp.p, t3, (uint32_t *)m->d_first.p, (uint32_t *)m->d_runid.p, (size_t)nev, rocprim::plus<uint32_t>(), c->stream);
                if (he == hipSuccess) {
                    hipLaunchKernelGGL(k_mem_run_ends, dim3(g), dim3(256), 0, c->stream, a, ek, ep, sidx, (const uint32_t *)m->d_first.p,
                                       (const uint32_t *)m->d_runid.p, (uint64_t)nev, (uint64_t *)m->d_rstart.p, (uint64_t *)m->d_rend.p);
                    hipLaunchKernelGGL(k_mem_apply, dim3(g), dim3(256), 0, c->stream, sidx, (const uint32_t *)m->d_runid.p, (uint64_t)nev,
                                       (uint32_t *)m->d_orun.p);
                    he = hipGetLastError();
                }
            }
        }
        (void)hipEventRecord(ev[3], c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_key.p, keys.current(), nev * 8, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_pos.p, vals.current(), nev * 8, hipMemcpyDeviceToHost, c->stream);
        uint32_t nruns = 0;
        if (he == hipSuccess) he = hipMemcpyAsync(&nruns, (const uint32_t *)m->d_runid.p + (nev - 1), 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_run.p, m->d_orun.p, nev * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he == hipSuccess && ((e = host_ensure(m, m->h_rstart, (size_t)nruns * 8)) || (e = host_ensure(m, m->h_rend, (size_t)nruns * 8)))) return e;
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_rstart.p, m->d_rstart.p, (size_t)nruns * 8, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_rend.p, m->d_rend.p, (size_t)nruns * 8, hipMemcpyDeviceToHost, c->stream);
        hrs = (const uint64_t *)m->h_rstart.p;
        hre = (const uint64_t *)m->h_rend.p;
        if (he == hipSuccess) he = hipMemcpyAsync(m->h_flag.p, m->d_oflag.p, nev, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { m->err = std::string("event passes: ") + hipGetErrorString(he); return PGRC_E_DEVICE; }
        (void)hipEventElapsedTime(&m->ctr.ms_sort, ev[1], ev[2]);
        (void)hipEventElapsedTime(&m->ctr.ms_extend, ev[2], ev[3]);
    } else {
        (void)hipEventSynchronize(ev[1]);
    }
    (void)hipEventElapsedTime(&m->ctr.ms_probe, ev[0], ev[1]);

    // ---- 4. the sequential rules, over the events only.
    // The scan is sequential for two reasons: a jump skips the next windows -- but never beyond its block of 256 windows
    // (main loop), so blocks are independent there -- and rule (b) looks at the LAST match recorded, which carries from
    // block to block.  The events are therefore cut into chunks at block boundaries and replayed by several host
    // threads speculatively (incoming last match: none); then the chunks are resolved in order: a chunk whose true
    // incoming match cannot reach into it (it ends before the chunk's first window) keeps its speculative result, the
    // others are replayed again with the right incoming state, round after round until nothing changes (one round when
    // no match straddles a chunk boundary; the chain is at worst sequential).  Chunks holding an event whose side
    // context lies outside a text (the stale-register walk needs the whole jump history) run in order on one thread.
    const auto th0 = std::chrono::steady_clock::now();
    Replay rp;
    rp.m = m; rp.dest = dest; rp.N2 = N2; rp.dest_is_src = dest_is_src != 0; rp.rev_compl = rev_compl != 0;
    rp.skip = (uint64_t)(m->K / m->k1 - 1);                                        // :352
    {   // main loop: blocks of 256 windows while i1 + K + 256 * k2 < N2 + 1 (:365)
        const uint64_t block = 256 * k2;
        const uint64_t lim = N2 + 1;                                               // i1 + K + block < lim
        uint64_t nb = 0;
        if (lim > K + block) nb = (lim - K - block + block - 1) / block;
        rp.nmain = nb * 256;
    }
    struct ChunkRun {
        uint64_t i0 = 0, i1 = 0;
        bool barrier = false;                    // holds an event that may need the stale-register walk: sequential only
        std::vector<pgrc_text_match> res;
        std::vector<Replay::Jump> jumps;
        bool in_have = false, out_have = false;  // incoming match this run assumed / last match recorded inside the chunk
        pgrc_text_match in_last{0, 0, 0}, out_last{0, 0, 0};
    };
    // events [i0, i1) with the incoming last match `in`; seq: the run may meet stale events and then walks the jump
    // history of everything before the chunk (rp.hist) and its own jumps so far (rp.cur)
    auto replay_chunk = [&](ChunkRun &ck, bool in_have, const pgrc_text_match &in, bool seq) -> int {
        ck.res.clear();
        ck.jumps.clear();
        ck.in_have = in_have;
        ck.in_last = in;
        ck.out_have = false;
        std::vector<Replay::Jump> &jumps = ck.jumps;
        if (seq) rp.cur = &ck.jumps;
        bool have_last = in_have;
        pgrc_text_match last = in;
        uint64_t i = ck.i0;
        while (i < ck.i1) {
            const uint64_t t = hk[i] >> 4;
            uint64_t jend = i;
            while (jend < ck.i1 && (hk[jend] >> 4) == t) jend++;
            // probes come in ascending order and a chunk starts with a new block: only the chunk's own latest jump can
            // still cover this one (Replay::examined, the general form, is for the walk-back of the stale registers)
            const bool visited = jumps.empty() || !(rp.block_of(jumps.back().t) == rp.block_of(t) && t <= jumps.back().t + rp.skip);
            if (visited) {
                const uint64_t q = t * k2;
                for (uint64_t x = i; x < jend; x++) {
                    const uint64_t p = hp[x];
                    const uint32_t order = (uint32_t)(hk[x] & 15u);
                    // (b) the window lies inside the previous match, on its diagonal (:393-399)
                    if (have_last && q - p == last.pos_dest - last.pos_src && q + K < last.pos_dest + last.length) {
                        jumps.push_back({t, order, 0u});
                        break;
                    }
                    // (c) side contexts (:401-404); registers whose 4 bytes lie outside a text keep their previous value
                    const uint32_t fl = hf[x];
                    bool pass;
                    if ((fl & 15u) == 15u) {
                        pass = (fl & (MF_L_EQ | MF_R_EQ)) != 0;
                    } else {
                        if (!seq) return -1;             // (cannot happen: such chunks are barriers)
                        m->ctr.stale_lookups++;
                        const uint32_t l1 = (fl & MF_L1_OK) ? le32(m->src + p - m->LK2) : rp.stale_src_reg(true, t, order);
                        const uint32_t r1 = (fl & MF_R1_OK) ? le32(m->src + p + m->KLK24) : rp.stale_src_reg(false, t, order);
                        const uint32_t l2 = (fl & MF_L2_OK) ? le32(dest + q - m->LK2) : 0u;   // windows below LK2 come first: still the initial 0
                        const uint32_t r2 = (fl & MF_R2_OK) ? le32(dest + q + m->KLK24) : rp.stale_r2(t);
                        pass = r1 == r2 || l1 == l2;
                        if (rp.lookup_err) return -2;
                    }
                    if (!pass) continue;
                    // (d) long enough?  right - p1 > minMatchLength with right - p1 = length + 1 (:413)
                    const uint64_t mstart = hrs[hr[x]], mlen = hre[hr[x]] - mstart;
                    if (mlen + 1 > (uint64_t)min_len) {
                        last.pos_src = mstart;
                        last.length = mlen;
                        last.pos_dest = q - (p - mstart);
                        have_last = true;
                        ck.out_have = true;
                        ck.out_last = last;
                        ck.res.push_back(last);
                        jumps.push_back({t, order, 1u});
                        break;
                    }
                }
            }
            i = jend;
        }
        return 0;
    };
    // chunks: cut where the block of 256 windows changes
    std::vector<ChunkRun> chunks;
    {
        unsigned hw = std::thread::hardware_concurrency();
        const char *knob = getenv("PGRC_MEM_REPLAY_THREADS");                         // test / A-B knob (1 = one sequential chunk)
        const uint64_t nthreads = knob ? (uint64_t)std::max(1, atoi(knob)) : std::min<uint64_t>(16, hw ? hw : 1);
        const char *cknob = getenv("PGRC_MEM_REPLAY_CHUNK");                          // test knob: events per chunk
        const uint64_t target = cknob ? (uint64_t)std::max(1, atoi(cknob))
                                      : (nthreads <= 1 || nev < 200000 ? nev + 1 : std::max<uint64_t>(50000, nev / (4 * nthreads)));
        uint64_t i0 = 0;
        while (i0 < nev) {
            uint64_t i1 = std::min<uint64_t>(nev, i0 + target);
            while (i1 < nev && rp.block_of(hk[i1] >> 4) == rp.block_of(hk[i1 - 1] >> 4)) i1++;
            ChunkRun ck;
            ck.i0 = i0;
            ck.i1 = i1;
            chunks.push_back(std::move(ck));
            i0 = i1;
        }
        const bool single = chunks.size() <= 1;
        // run `ids` on the thread pool; a chunk is first scanned for stale-capable events (those make it a barrier)
        auto run_parallel = [&](const std::vector<size_t> &ids, const std::vector<std::pair<bool, pgrc_text_match>> &inc) {
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (;;) {
                    const size_t w = next.fetch_add(1);
                    if (w >= ids.size()) break;
                    ChunkRun &ck = chunks[ids[w]];
                    (void)replay_chunk(ck, inc[w].first, inc[w].second, false);
                }
            };
            const size_t nt = std::min<size_t>(nthreads, ids.size());
            std::vector<std::thread> th;
            for (size_t k = 1; k < nt; k++) th.emplace_back(worker);
            worker();
            for (auto &t : th) t.join();
        };
        if (single) {
            for (ChunkRun &ck : chunks) ck.barrier = true;
        } else {
            // barriers: any event whose four side contexts are not all inside the texts
            std::atomic<size_t> next{0};
            auto scan = [&]() {
                for (;;) {
                    const size_t w = next.fetch_add(1);
                    if (w >= chunks.size()) break;
                    ChunkRun &ck = chunks[w];
                    for (uint64_t x = ck.i0; x < ck.i1; x++)
                        if ((hf[x] & 15u) != 15u) { ck.barrier = true; break; }
                }
            };
            {
                std::vector<std::thread> th;
                for (size_t k = 1; k < std::min<size_t>(nthreads, chunks.size()); k++) th.emplace_back(scan);
                scan();
                for (auto &t : th) t.join();
            }
            std::vector<size_t> ids;
            for (size_t k = 0; k < chunks.size(); k++)
                if (!chunks[k].barrier) ids.push_back(k);
            run_parallel(ids, std::vector<std::pair<bool, pgrc_text_match>>(ids.size(), {false, pgrc_text_match{0, 0, 0}}));
        }
        // resolve in order
        auto reaches = [&](bool have, const pgrc_text_match &l, const ChunkRun &ck) {
            return have && (hk[ck.i0] >> 4) * k2 + K < l.pos_dest + l.length;   // rule (b) could fire inside the chunk
        };
        auto same = [](bool ha, const pgrc_text_match &a, bool hb, const pgrc_text_match &b) {
            return ha == hb && (!ha || (a.pos_src == b.pos_src && a.length == b.length && a.pos_dest == b.pos_dest));
        };
        bool cur_have = false;
        pgrc_text_match cur{0, 0, 0};
        size_t k = 0;
        while (k < chunks.size()) {
            if (chunks[k].barrier) {
                const int rcx = replay_chunk(chunks[k], cur_have, cur, true);
                rp.cur = nullptr;
                if (rcx) { m->err = "bucket lookup failed"; return PGRC_E_DEVICE; }
                if (!chunks[k].jumps.empty()) rp.hist.push_back(&chunks[k].jumps);
                if (chunks[k].out_have) { cur_have = true; cur = chunks[k].out_last; }
                k++;
                continue;
            }
            size_t k2e = k;
            while (k2e < chunks.size() && !chunks[k2e].barrier) k2e++;
            for (;;) {      // rounds over [k, k2e)
                std::vector<size_t> ids;
                std::vector<std::pair<bool, pgrc_text_match>> inc;
                bool h = cur_have;
                pgrc_text_match l = cur;
                for (size_t x = k; x < k2e; x++) {
                    const bool eh = reaches(h, l, chunks[x]);
                    if (!same(eh, l, chunks[x].in_have, chunks[x].in_last)) {
                        ids.push_back(x);
                        inc.push_back({eh, eh ? l : pgrc_text_match{0, 0, 0}});
                    }
                    if (chunks[x].out_have) { h = true; l = chunks[x].out_last; }
                }
                if (ids.empty()) {
                    cur_have = h;
                    cur = l;
                    break;
                }
                run_parallel(ids, inc);
            }
            for (size_t x = k; x < k2e; x++)
                if (!chunks[x].jumps.empty()) rp.hist.push_back(&chunks[x].jumps);
            k = k2e;
        }
    }
    std::vector<pgrc_text_match> res;
    {
        size_t total = 0;
        for (const ChunkRun &ck : chunks) total += ck.res.size();
        res.reserve(total);
        for (const ChunkRun &ck : chunks) res.insert(res.end(), ck.res.begin(), ck.res.end());
    }
    m->ctr.ms_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - th0).count();
    if (!res.empty()) {
        pgrc_text_match *outp = (pgrc_text_match *)malloc(res.size() * sizeof(pgrc_text_match));
        if (!outp) { m->err = "out of host memory"; return PGRC_E_ALLOC; }
        memcpy(outp, res.data(), res.size() * sizeof(pgrc_text_match));
        *matches = outp;
    }
    *count = res.size();
    return PGRC_OK;
}

} // extern "C"
