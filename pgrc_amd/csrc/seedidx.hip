// seedidx.hip -- modes 'd', 'i' and the exact matcher 'e': the reference indexes the READS' seed parts and
// streams the pseudogenome past that index (serial code, node-based unordered_multimap).
//
// Reference behaviour restated:
//   index   : Default/InterleavedConstantLengthPatternsOnTextHashMatcher::addReadsSetOfPatterns / addPackedPatterns,
//             matching/ConstantLengthPatternsOnTextHashMatcher.cpp:23-42, :76-93 (part j of read i -> pattern i*P+j)
//   scan    : iterateOver / moveNext, ConstantLengthPatternsOnTextHashMatcher.h:42-68, :105-137
//   key     : CyclicHash<uint32>, rollinghash/cyclichash.h:100-123 -- randomly keyed per run, so only the
//             EQUIVALENCE it induces is canonical: XOR of per-symbol words rotated by (n-1-i) mod 32, i.e. seeds
//             longer than 32 collide when every rotation class has equal symbol parities.  We evaluate the same
//             cyclic polynomial with two fixed tables (64 key bits).
//   verify  : DefaultReadsExactMatcher::executeMatching ReadsMatchers.cpp:198-230 (first exact hit in scan order),
//             DefaultReadsApproxMatcher :297-341, InterleavedReadsApproxMatcher :364-409 (strict improvement, the
//             stored-position skip, limit = count-1), countSequenceMismatchesVsUnpacked SymbolsPackingFacility.cpp:344-374
//
// MI355X design: the read-part keys go into an open-addressing table in HBM (atomicCAS insert, chained duplicates);
// one thread per text position computes its window key from the 2-bit text and emits a HIT record
// (read << 36 | text position << 4 | 15 - part) for every chained pattern; one thread per hit takes its Hamming count
// (popcount on 2-bit words) and the reference's sequential rule over a read's candidates -- walked in scan order: ascending
// text position, equal positions in descending part index -- is taken as the lexicographic minimum it amounts to, one
// atomicMin per acceptable hit on a 64-bit key per read (section 3).  No sort, no library kernel.
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "ctx.h"
#include "devutil.h"

#define SX_EMPTY 0xFFFFFFFFFFFFFFFFull
#define SX_NIL 0xFFFFFFFFu

// two fixed 32-bit tables over the symbol values A0 C1 G2 T3 N4
__device__ __forceinline__ uint32_t cyc_t0(uint32_t v) {
    return v == 0 ? 0x9E3779B9u : v == 1 ? 0x7F4A7C15u : v == 2 ? 0xF39CC060u : v == 3 ? 0x5CEDC834u : 0x1082276Bu;
}
__device__ __forceinline__ uint32_t cyc_t1(uint32_t v) {
    return v == 0 ? 0xBF58476Du : v == 1 ? 0x1CE4E5B9u : v == 2 ? 0x94D049BBu : v == 3 ? 0x133111EBu : 0x2545F491u;
}
__device__ __forceinline__ uint32_t rotl1(uint32_t x) { return (x << 1) | (x >> 31); }
__device__ __forceinline__ uint64_t mix64d(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t key_fix(uint32_t h0, uint32_t h1) {
    uint64_t k = ((uint64_t)h0 << 32) | h1;
    return k == SX_EMPTY ? SX_EMPTY - 1 : k;
}

struct SeedArgs {
    const uint32_t *pg;
    uint64_t G;
    const uint32_t *reads;
    uint64_t n, stride;
    const uint8_t *nflag;
    const uint32_t *nidx;
    const uint8_t *nascii;
    const uint16_t *nmask;     // [nn][nwr]: bit k of entry w = symbol 16 w + k of that read is an N
    uint32_t nwr;
    uint64_t nn;
    uint32_t L, m, P, cstride; // m = pattern length, P = parts, cstride = symbol stride inside a part (P for mode i)
    uint32_t mode;             // 'd', 'i', 'e'
    uint32_t kmax, kmin, strand;
    uint64_t tbase;            // first window start of the text segment being scanned (hit records keep 32-bit offsets from it)
    uint64_t ibase;            // first read of the batch (reads / result pointers are already offset; nidx holds set-wide indexes)
    uint64_t *tkeys;
    uint32_t *theads;
    uint64_t tmask;
    uint32_t *filter;          // one bit per 2^-fbits of the key space: set iff some indexed key falls there
    uint32_t fshift;           // 64 - fbits: the filter takes the TOP bits of the mixed key, the table its low bits
    uint32_t *next;
    uint64_t *pos;
    uint8_t *rc, *mism;
};

__device__ __forceinline__ uint32_t read_code(const SeedArgs &a, uint64_t i, uint32_t x) {
    return (a.reads[(uint64_t)(x >> 4) * a.stride + i] >> (2u * (x & 15u))) & 3u;
}
__device__ __forceinline__ uint32_t ascii_val(uint8_t ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u; }
__device__ __forceinline__ uint32_t part_offset(const SeedArgs &a, uint32_t j) { return a.mode == 'i' ? j : j * a.m; }

// ---- 1. insert every (read, part) key
__device__ __forceinline__ void table_insert(const SeedArgs &a, uint64_t key, uint32_t e) {
    const uint64_t mixed = mix64d(key);
    if (a.filter) {
        const uint64_t fb = mixed >> a.fshift;
        atomicOr(&a.filter[fb >> 5], 1u << (fb & 31u));
    }
    uint64_t slot = mixed & a.tmask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long *)&a.tkeys[slot], (unsigned long long)SX_EMPTY, (unsigned long long)key);
        if (prev == SX_EMPTY || prev == key) {
            a.next[e] = atomicExch(&a.theads[slot], e);
            return;
        }
        slot = (slot + 1) & a.tmask;
    }
}

__global__ void __launch_bounds__(256) k_seed_insert(const SeedArgs a) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n * a.P) return;
    const uint64_t i = e / a.P;
    const uint32_t j = (uint32_t)(e % a.P);
    if (a.nflag && a.nflag[i]) return; // byte-path reads are inserted by k_seed_insert_ascii
    const uint32_t off = part_offset(a, j);
    uint32_t h0 = 0, h1 = 0;
    for (uint32_t k = 0; k < a.m; k++) {
        const uint32_t c = read_code(a, i, off + k * a.cstride);
        h0 = rotl1(h0) ^ cyc_t0(c);
        h1 = rotl1(h1) ^ cyc_t1(c);
    }
    table_insert(a, key_fix(h0, h1), (uint32_t)e);
}

__global__ void __launch_bounds__(256) k_seed_insert_ascii(const SeedArgs a) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.nn * a.P) return;
    const uint64_t t = x / a.P;
    const uint32_t j = (uint32_t)(x % a.P);
    const uint64_t i = a.nidx[t] - a.ibase;
    const uint8_t *row = a.nascii + t * a.L;
    const uint32_t off = part_offset(a, j);
    uint32_t h0 = 0, h1 = 0;
    for (uint32_t k = 0; k < a.m; k++) {
        const uint32_t c = ascii_val(row[off + k * a.cstride]);
        h0 = rotl1(h0) ^ cyc_t0(c);
        h1 = rotl1(h1) ^ cyc_t1(c);
    }
    table_insert(a, key_fix(h0, h1), (uint32_t)(i * a.P + j));
}

// ---- 2. stream the text past the table.  A block stages its stretch of the text in LDS; a thread walks SCAN_R window
// starts that are cstride apart with the O(1) rolling update of the cyclic polynomial (the reference's own scan,
// cyclichash.h:110-118) and probes the table for each; hits are appended with ONE atomic per wave and iteration
// into per-wave LDS buffers (the fill count is wave-uniform: a register, no LDS atomic) that are flushed with one
// global atomic per block at the end -- or per wave and 512 hits when a buffer fills up (tandem repeats).  Same-address
// returning atomics serialise at ~90 M/s: one per hit, or even one per wave and iteration (4 M at C2), made the
// cursor the whole cost of this kernel, and a spill path that did so turned C3-size runs into seconds.  Single pass: the hit buffer is sized by a guess, the cursor keeps counting past its end, and
// the host reruns the pass with the exact size if it overflowed.
#define SCAN_TPB 256
#define SCAN_R 16
#define SCAN_TILE_WORDS (SCAN_TPB * SCAN_R / 16 + 24)
#define SCAN_WCAP 512u     // hit records per wave buffer
#define SCAN_B 8
__global__ void __launch_bounds__(SCAN_TPB)
k_seed_scan(const SeedArgs a, uint64_t nwin, uint64_t pg_words_alloc, unsigned long long *cursor, uint64_t *hits, uint64_t cap) {
    // windows [a.tbase, nwin) of the text: one segment of fewer than 2^32 window starts
    __shared__ uint32_t tile[SCAN_TILE_WORDS];
    __shared__ uint64_t lbuf[SCAN_TPB / 64][SCAN_WCAP];
    __shared__ uint32_t wtotal[SCAN_TPB / 64];
    __shared__ unsigned long long gbase;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t wcount = 0;                                  // records in this wave's buffer (wave-uniform)
    const uint32_t cs = a.cstride, m = a.m;
    const uint32_t groups = SCAN_TPB / cs;               // runs of SCAN_R * cs consecutive starts, one thread per phase
    const uint32_t per_block = groups * cs * SCAN_R;
    const uint64_t b0 = a.tbase + (uint64_t)blockIdx.x * per_block;
    const uint64_t w0 = b0 >> 4;
    const uint32_t need = (uint32_t)(((b0 & 15) + per_block + (uint64_t)m * cs + 15) >> 4) + 1;   // <= SCAN_TILE_WORDS
    for (uint32_t w = threadIdx.x; w < need; w += SCAN_TPB) tile[w] = (w0 + w < pg_words_alloc) ? a.pg[w0 + w] : 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const bool worker = threadIdx.x < groups * cs;
    const uint32_t g = threadIdx.x / cs, phase = threadIdx.x % cs;
    const uint64_t s0 = b0 + (uint64_t)g * cs * SCAN_R + phase;
    const uint32_t x0 = (uint32_t)(s0 - (w0 << 4));       // tile-relative symbol index of s0
    auto sym = [&](uint32_t x) -> uint32_t { return (tile[x >> 4] >> (2u * (x & 15u))) & 3u; };
    uint32_t h0 = 0, h1 = 0;
    if (worker)
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t c = sym(x0 + k * cs);
            h0 = rotl1(h0) ^ cyc_t0(c);
            h1 = rotl1(h1) ^ cyc_t1(c);
        }
    const uint32_t mr = m & 31u;
    // batches of SCAN_B window starts: all keys of a batch first (rolling), then their table probes as independent
    // loads, then the heads of the keys found -- a dependent round trip per batch instead of one per start
    for (uint32_t r0 = 0; r0 < SCAN_R; r0 += SCAN_B) {
        uint64_t keyv[SCAN_B], slotv[SCAN_B], kv[SCAN_B];
        uint32_t ev[SCAN_B], fw[SCAN_B], fbit[SCAN_B];
#pragma unroll
        for (int b = 0; b < SCAN_B; b++) {
            keyv[b] = key_fix(h0, h1);
            const uint64_t mixed = mix64d(keyv[b]);
            slotv[b] = mixed & a.tmask;
            fbit[b] = (uint32_t)((mixed >> a.fshift) & 31u);
            fw[b] = (uint32_t)(mixed >> a.fshift >> 5);
            if (worker) {                                     // roll to the next start (cyclichash.h:110-118)
                const uint32_t xo = x0 + (r0 + b) * cs;
                const uint32_t co = sym(xo), cn = sym(xo + m * cs);
                const uint32_t o0 = cyc_t0(co), o1 = cyc_t1(co);
                h0 = rotl1(h0) ^ ((o0 << mr) | (mr ? o0 >> (32u - mr) : 0u)) ^ cyc_t0(cn);
                h1 = rotl1(h1) ^ ((o1 << mr) | (mr ? o1 >> (32u - mr) : 0u)) ^ cyc_t1(cn);
            }
        }
        // the filter first: a window whose bit is clear equals no indexed key -- one 4-byte gather in a bitmap of a thirty-second
        // of the table's bytes; only the windows that pass (the hits and ~3 % more) go on to the table, whose probes each
        // cost a random line of their own, collision steps included (the line of the previous slot is long gone from L1 / L2)
#pragma unroll
        for (int b = 0; b < SCAN_B; b++) {
            const uint64_t t = s0 + (uint64_t)(r0 + b) * cs;
            fw[b] = (worker && t < nwin) ? (a.filter ? a.filter[fw[b]] : 0xFFFFFFFFu) : 0u;
        }
#pragma unroll
        for (int b = 0; b < SCAN_B; b++) kv[b] = ((fw[b] >> fbit[b]) & 1u) ? a.tkeys[slotv[b]] : SX_EMPTY;
        // collisions (the slot holds another key) of the whole batch are resolved together: every round issues the
        // next-slot loads of all starts still searching before any of them is looked at
        uint32_t srch = 0, fnd = 0;
#pragma unroll
        for (int b = 0; b < SCAN_B; b++) {
            if (kv[b] == keyv[b]) fnd |= 1u << b;
            else if (kv[b] != SX_EMPTY) srch |= 1u << b;
        }
        while (__any(srch != 0)) {
#pragma unroll
            for (int b = 0; b < SCAN_B; b++)
                if (srch & (1u << b)) {
                    slotv[b] = (slotv[b] + 1) & a.tmask;
                    kv[b] = a.tkeys[slotv[b]];
                }
#pragma unroll
            for (int b = 0; b < SCAN_B; b++)
                if (srch & (1u << b)) {
                    if (kv[b] == SX_EMPTY) srch &= ~(1u << b);
                    else if (kv[b] == keyv[b]) { fnd |= 1u << b; srch &= ~(1u << b); }
                }
        }
#pragma unroll
        for (int b = 0; b < SCAN_B; b++) ev[b] = (fnd & (1u << b)) ? a.theads[slotv[b]] : SX_NIL;
        // the chains of the batch are walked together as well (one round = the next link of every live chain)
        for (;;) {
            uint32_t live = 0;
#pragma unroll
            for (int b = 0; b < SCAN_B; b++) live |= (ev[b] != SX_NIL) ? 1u << b : 0u;
            if (!__any(live != 0)) break;
            uint32_t nx[SCAN_B];
#pragma unroll
            for (int b = 0; b < SCAN_B; b++) nx[b] = (live & (1u << b)) ? a.next[ev[b]] : SX_NIL;
#pragma unroll
            for (int b = 0; b < SCAN_B; b++) {
                const uint64_t t = s0 + (uint64_t)(r0 + b) * cs;
                bool emit = false;
                uint64_t rec = 0;
                if (live & (1u << b)) {
                    const uint32_t e = ev[b];
                    const uint64_t i = e / a.P;
                    const uint32_t j = e % a.P;
                    const uint64_t shift = part_offset(a, j);
                    // ReadsMatchers.cpp:308-309 / :375-376 and :311-312 / :378-379
                    if (shift <= t && t - shift + a.L <= a.G) { emit = true; rec = (i << 36) | ((t - a.tbase) << 4) | (15u - j); }
                }
                const unsigned long long mk = __ballot(emit);
                if (mk) {
                    const uint32_t cnt = (uint32_t)__popcll(mk);
                    if (wcount + cnt > SCAN_WCAP) {              // buffer full: this wave flushes it (one atomic per 512 hits)
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)wcount);
                        base = __shfl(base, 0, 64);
                        for (uint32_t x = lane; x < wcount; x += 64)
                            if (base + x < cap) hits[base + x] = lbuf[wave][x];
                        wcount = 0;
                    }
                    if (emit) lbuf[wave][wcount + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = rec;
                    wcount += cnt;
                }
                ev[b] = nx[b];
            }
        }
    }
    // what is left in the wave buffers leaves with one global atomic per block
    if (lane == 0) wtotal[wave] = wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < SCAN_TPB / 64; w++) tot += wtotal[w];
        gbase = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long off = gbase;
    for (uint32_t w = 0; w < wave; w++) off += wtotal[w];
    for (uint32_t x = lane; x < wcount; x += 64)
        if (off + x < cap) hits[off + x] = lbuf[wave][x];
}

// ---- 3. the hits' Hamming counts and what the reference's rule leaves of them
// reads with N: their packed words hold code 0 at the N positions; a 16-bit mask per word forces those symbols to count
// as mismatches (an N equals no text symbol)
__global__ void __launch_bounds__(256)
k_seed_nmask(const uint8_t *__restrict__ nascii, uint64_t nn, uint32_t L, uint32_t nwr, uint16_t *__restrict__ nmask) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nn * nwr) return;
    const uint64_t t = x / nwr;
    const uint32_t w = (uint32_t)(x % nwr);
    const uint8_t *row = nascii + t * L + 16 * w;
    uint32_t m = 0;
    for (uint32_t k = 0; k < 16 && 16 * w + k < L; k++) m |= (row[k] == 'N') ? 1u << k : 0u;
    nmask[x] = (uint16_t)m;
}

__device__ __forceinline__ uint32_t hamming_vs_text_n(const SeedArgs &a, uint64_t i, uint64_t trow, uint64_t p) {
    uint32_t mm = 0;
    const uint32_t *src = a.pg + (p >> 4);
    const uint32_t b = ((uint32_t)p & 15u) * 2u;
    uint32_t lo = src[0];
    for (uint32_t w = 0; w < a.nwr; w++) {
        const uint32_t hi = src[w + 1];
        const uint32_t x = funnel_r(lo, hi, b) ^ a.reads[(uint64_t)w * a.stride + i];
        uint32_t nb = a.nmask[trow * a.nwr + w];          // bit k -> bit 2 k
        nb = (nb | (nb << 8)) & 0x00FF00FFu;
        nb = (nb | (nb << 4)) & 0x0F0F0F0Fu;
        nb = (nb | (nb << 2)) & 0x33333333u;
        nb = (nb | (nb << 1)) & 0x55555555u;
        mm += (uint32_t)__popc((x | (x >> 1) | nb) & sym_mask((int)w, 0, (int)a.L));
        lo = hi;
    }
    return mm;
}

__device__ __forceinline__ uint64_t lower_bound_u32(const uint32_t *v, uint64_t n, uint32_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- 3b. what the reference's sequential rule leaves of a read's hits (ReadsMatchers.cpp:297-341, :198-230).  It walks the
// hits in scan order -- ascending text position, at one position the parts in descending order: the low 36 bits of a hit
// record, ascending -- and accepts a hit whose count is BELOW the read's current one (<= kmax for a read not matched yet),
// until the count is <= kmin.  So it ends at the first hit with a count <= kmin if there is one, else at the first hit that
// attains the smallest count: a lexicographic MINIMUM over the hits of (count <= kmin ? 0 : count, scan order), taken over the
// hits with count <= the limit the read starts with.  The exact matcher (:198-230): the first hit in scan order that equals the
// read.  (`stored == candidate -> skip`, :313-314, never changes the outcome: such a hit has the stored alignment's own count,
// which is not below it.)  A minimum does not need a read's hits together, let alone in order (rounds 1-3 sorted all 63 bits of
// the hit records with the library and replayed the rule; round 4 first sorted the read bits only, then nothing at all):
// it is one 64-bit atomicMin per acceptable hit on a key per read,
//     count' (8 bits) | strand (1) | window start in the strand's text (40) | 15 - part (4) | unused (3) | count (8)
// (count' = 0 for a count <= kmin; the count itself rides in the low bits, below everything that orders).  The strand bit makes
// the two passes one minimum as well: the RC pass of the reference accepts a hit only with a count BELOW the forward result's
// (ReadsMatchers.cpp:315-316 with the count the first pass left), i.e. at equal count' the forward hit wins, and a forward count
// <= kmin (count' 0) is beaten by nothing.  A read's state before the run (a second-phase run, :304-305) is its start key:
// all ones = not matched, count' << 56 with nothing below = matched with that count -- no hit has a smaller key at the same count',
// a hit's low bits are never all zero (15 - part >= 1).  So: start keys, every strand and segment of the text scanned with the
// hits' Hamming counts taken in whatever order the scan left them, one pass over the reads at the end.  The hits arrive in text
// order, their reads are random: the Hamming kernel takes a read's words from a ROW-major copy of the batch (one or two lines per
// hit instead of one per word).  The hits of one read alignment (one per part that matches) sit next to each other in the hit
// buffer: a hit does not go to memory when a neighbouring lane holds a smaller key of the same read, nor when the key in
// memory is already no larger (a plain load: the key only ever falls) -- random 64-bit atomics run at a sixth of the rate of
// random loads.  C3, mode d (830 M hits per strand): sort + counts + replay 75 ms per strand -> 22 ms, 372 -> 260 ms per run
// (profiles/r04_modes_c3.txt).
#define BK_NONE 0xFFFFFFFFFFFFFFFFull
#define BK_HITBITS ((1ull << 55) - 1ull)

__global__ void __launch_bounds__(256) k_seed_best_init(const SeedArgs a, uint64_t *__restrict__ best) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t c = a.mism[i];
    uint64_t k;
    if (a.mode == 'e') k = a.pos[i] != PGRC_NOT_MATCHED_POS ? 0ull : BK_NONE;      // :198-230: a read matched before is left alone
    else k = c == PGRC_NOT_MATCHED_CNT ? BK_NONE : (uint64_t)(c <= a.kmin ? 0u : c) << 56;
    best[i] = k;
}

// word-major words of the batch -> rows of rw words (rw = words of a read rounded up to 4: 16-byte loads)
#define ROWS_TPB 256
__global__ void __launch_bounds__(ROWS_TPB) k_seed_rows(const SeedArgs a, uint32_t rw, uint32_t *__restrict__ rows) {
    extern __shared__ uint32_t rt[];                       // [rw][ROWS_TPB + 1]
    const uint64_t i0 = (uint64_t)blockIdx.x * ROWS_TPB;
    const uint64_t i = i0 + threadIdx.x;
    for (uint32_t w = 0; w < rw; w++) rt[w * (ROWS_TPB + 1) + threadIdx.x] = (w < a.nwr && i < a.n) ? a.reads[(uint64_t)w * a.stride + i] : 0u;
    __syncthreads();
    const uint64_t nrows = a.n - i0 < ROWS_TPB ? a.n - i0 : ROWS_TPB;
    for (uint64_t k = threadIdx.x; k < nrows * rw; k += ROWS_TPB) rows[i0 * rw + k] = rt[(k % rw) * (ROWS_TPB + 1) + k / rw];
}

template <int RW4>   // rw / 4
__device__ __forceinline__ uint32_t hamming_row_vs_text(const SeedArgs &a, const uint32_t *__restrict__ rows, uint64_t i, uint64_t p) {
    const uint4 *row = (const uint4 *)(rows + i * (uint64_t)(RW4 * 4));
    uint32_t r[RW4 * 4];
#pragma unroll
    for (int q = 0; q < RW4; q++) {
        const uint4 v = row[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
    const uint32_t *src = a.pg + (p >> 4);
    const uint32_t b = ((uint32_t)p & 15u) * 2u;
    uint32_t mm = 0, lo = src[0];
#pragma unroll
    for (int w = 0; w < RW4 * 4; w++) {
        if ((uint32_t)w < a.nwr) {
            const uint32_t hi = src[w + 1];
            mm += mism2(funnel_r(lo, hi, b), r[w], sym_mask(w, 0, (int)a.L));
            lo = hi;
        }
    }
    return mm;
}

template <int RW4>
__global__ void __launch_bounds__(256)
k_seed_hamming_min(const SeedArgs a, const uint64_t *__restrict__ hits, uint64_t nhits, const uint32_t *__restrict__ rows,
                   uint64_t *__restrict__ best) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i = ~0ull, key = BK_NONE;                      // key == BK_NONE: nothing to offer
    if (x < nhits) {
        const uint64_t hkey = hits[x];
        i = hkey >> 36;
        const uint64_t tp = a.tbase + ((hkey >> 4) & 0xFFFFFFFFull);
        const uint32_t j = 15u - (uint32_t)(hkey & 15u);
        const uint64_t p = tp - part_offset(a, j);
        uint32_t mm;
        if (a.nflag && a.nflag[i]) mm = hamming_vs_text_n(a, i, lower_bound_u32(a.nidx, a.nn, (uint32_t)(i + a.ibase)), p);   // a read with N
        else mm = hamming_row_vs_text<RW4>(a, rows, i, p);
        if (a.mode == 'e' ? mm == 0u : mm <= a.kmax)        // ReadsMatchers.cpp:315-319 / :214: no limit a read can have lets the others in
            key = ((uint64_t)(mm <= a.kmin ? 0u : mm) << 56) | ((uint64_t)a.strand << 55) | (tp << 15) | ((hkey & 15ull) << 11) | mm;
    }
    // a neighbour (two lanes either way) with a smaller key of the same read takes this hit's place; keys of one read differ
    bool mine = key != BK_NONE;
#pragma unroll
    for (int d = 1; d <= 2; d++) {
        const uint64_t iu = __shfl_up(i, d, 64), ku = __shfl_up(key, d, 64), id = __shfl_down(i, d, 64), kd = __shfl_down(key, d, 64);
        const uint32_t lane = threadIdx.x & 63u;
        if (lane >= (uint32_t)d && iu == i && ku < key) mine = false;
        if (lane + (uint32_t)d < 64u && id == i && kd < key) mine = false;
    }
    if (!mine || best[i] <= key) return;
    atomicMin((unsigned long long *)&best[i], (unsigned long long)key);
}

__global__ void __launch_bounds__(256) k_seed_best_store(const SeedArgs a, const uint64_t *__restrict__ best) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t k = best[i];
    if (k == BK_NONE || (k & BK_HITBITS) == 0ull) return;             // no acceptable hit: the read keeps what it had
    const uint32_t strand = (uint32_t)(k >> 55) & 1u;
    const uint64_t tp = (k >> 15) & ((1ull << 40) - 1ull);
    const uint32_t j = 15u - (uint32_t)((k >> 11) & 15u);
    const uint64_t p = tp - part_offset(a, j);
    a.pos[i] = strand ? a.G - (p + a.L) : p;
    a.rc[i] = (uint8_t)strand;
    a.mism[i] = (uint8_t)(k & 0xFFu);
}


__global__ void __launch_bounds__(256) k_seed_table_init(uint64_t *keys, uint32_t *heads, uint64_t n) {
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x) {
        keys[s] = SX_EMPTY;
        heads[s] = SX_NIL;
    }
}

// One batch of reads (fewer than 2^28: the hit records keep 28 bits for the read) against both strands, the text
// scanned in segments of fewer than 2^32 window starts (32 bits for the position inside the segment).  A read's
// candidates come in ascending text position, so segment after segment with the per-read state carried in the result
// arrays IS the reference's sequential scan; reads are independent, so batch after batch is its loop over the reads.
static int seedidx_batch(pgrc_match_ctx *c, SeedArgs a, uint64_t seg_windows, int first_strand, int last_strand) {
    const uint64_t span = (uint64_t)a.m * a.cstride;  // extent of a text window
    const uint32_t L = a.L;
    // read-part table
    const uint64_t nent = a.n * a.P;
    uint64_t tsize = 1024;
    while (tsize < 2 * nent) tsize <<= 1;   // (4 / 8 * nent: the exact matcher at C3 12 / 16 % faster, modes d / i unchanged; the filter below does better)
    int e;
    if ((e = pgrc_buf_ensure(c, c->s_keys, tsize * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_vals, tsize * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_tab, nent * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_tmp, 64))) return e;
    // the filter: 32 bits per indexed key (3 % of the windows of a random text pass it by chance), at most 2^36 bits.  It
    // pays when most windows of the text equal no key -- the exact matcher at C3: 100 M keys against 1.9 G windows, 0.20 ->
    // 0.16 s -- and costs a dependent round trip where many do (modes d / i with four parts per read: 0.42 -> 0.43 s): used
    // when there is at most one key per eight text positions (PGRC_SEED_FILTER=0 / 1: never / always, tests and A/B runs)
    bool use_filter = nent * 8 <= c->G;
    if (const char *v = getenv("PGRC_SEED_FILTER")) use_filter = v[0] != '0';
    a.filter = nullptr;
    a.fshift = 0;
    if (use_filter) {
        int fbits = 10;
        while (fbits < 36 && (1ull << fbits) < 32 * nent) fbits++;
        if ((e = pgrc_buf_ensure(c, c->s_filter, (size_t)(1ull << fbits) / 8))) return e;
        HIP_TRY(c, hipMemsetAsync(c->s_filter.p, 0, (size_t)(1ull << fbits) / 8, c->stream));
        a.filter = (uint32_t *)c->s_filter.p;
        a.fshift = 64u - (uint32_t)fbits;
    }
    a.tkeys = (uint64_t *)c->s_keys.p;
    a.theads = (uint32_t *)c->s_vals.p;
    a.tmask = tsize - 1;
    a.next = (uint32_t *)c->s_tab.p;
    if (a.nn) {
        if ((e = pgrc_buf_ensure(c, c->s_nmask, a.nn * a.nwr * sizeof(uint16_t)))) return e;
        a.nmask = (const uint16_t *)c->s_nmask.p;
        hipLaunchKernelGGL(k_seed_nmask, dim3((uint32_t)((a.nn * a.nwr + 255) / 256)), dim3(256), 0, c->stream, a.nascii, a.nn, L,
                           a.nwr, (uint16_t *)c->s_nmask.p);
    }
    hipLaunchKernelGGL(k_seed_table_init, dim3(4096), dim3(256), 0, c->stream, a.tkeys, a.theads, tsize);
    hipLaunchKernelGGL(k_seed_insert, dim3((uint32_t)((nent + 255) / 256)), dim3(256), 0, c->stream, a);
    if (a.nn)
        hipLaunchKernelGGL(k_seed_insert_ascii, dim3((uint32_t)((a.nn * a.P + 255) / 256)), dim3(256), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());

    // a read's hits become its result by the atomic minimum of section 3b: a start key per read, the batch's reads row by row
    const uint32_t rw = (a.nwr + 3u) & ~3u;                 // <= 16: reads have at most 255 symbols (pgrc_match_create)
    if (c->G >= (1ull << 40)) { c->err = "modes d/i/e: texts below 2^40 symbols"; return PGRC_E_PARAM; }
    if ((e = pgrc_buf_ensure(c, c->s_best, a.n * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_rows, a.n * rw * sizeof(uint32_t)))) return e;
    hipLaunchKernelGGL(k_seed_best_init, dim3((uint32_t)((a.n + 255) / 256)), dim3(256), 0, c->stream, a, (uint64_t *)c->s_best.p);
    hipLaunchKernelGGL(k_seed_rows, dim3((uint32_t)((a.n + ROWS_TPB - 1) / ROWS_TPB)), dim3(ROWS_TPB), rw * (ROWS_TPB + 1) * sizeof(uint32_t),
                       c->stream, a, rw, (uint32_t *)c->s_rows.p);
    HIP_TRY(c, hipGetLastError());

    unsigned long long *cursor = (unsigned long long *)c->s_tmp.p;
    for (int pass = first_strand; pass <= last_strand; pass++) {
        a.pg = (const uint32_t *)c->pg2[pass].p;
        a.strand = (uint32_t)pass;
        if (c->G < span) continue; // no window fits (the reference's scan loops are empty / undefined there)
        const uint64_t nwin_all = c->G - span + 1;
        const uint32_t per_block = (SCAN_TPB / a.cstride) * a.cstride * SCAN_R;
        for (uint64_t w0 = 0; w0 < nwin_all; w0 += seg_windows) {
        a.tbase = w0;
        const uint64_t nwin = std::min(nwin_all, w0 + seg_windows);      // this segment: window starts [w0, nwin)
        const uint32_t grid = (uint32_t)((nwin - w0 + per_block - 1) / per_block);
        // hit buffer: a guess (two hits per indexed part, or whatever an earlier pass needed); exact size on overflow
        uint64_t cap = std::max<uint64_t>(2 * nent + 4096, c->s_hits.bytes / sizeof(uint64_t));
        unsigned long long nhits = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            if ((e = pgrc_buf_ensure(c, c->s_hits, cap * sizeof(uint64_t)))) return e;
            HIP_TRY(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL(k_seed_scan, dim3(grid), dim3(SCAN_TPB), 0, c->stream, a, nwin, c->pg_words + PGRC_PG_PAD_WORDS,
                               cursor, (uint64_t *)c->s_hits.p, cap);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&nhits, cursor, sizeof nhits, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (nhits <= cap) break;
            cap = nhits + nhits / 4;                        // the guess was too small: once more, with headroom for the other strand
                                                            // (growing these multi-GB buffers again costs more than the pass itself)
        }
        if (nhits == 0) continue;
        // counts in scan order, one atomicMin per acceptable hit (the host does not wait for it: the next scan follows on the stream)
        // (launches of at most 2^30 hits.  In an experiment of round 4 ONE launch of this kernel over 3.7 G hits gave wrong results
        //  where the same hits in launches of 2^30 were right -- profiles/r04_seed_scan_experiments.txt; the cause was not found:
        //  plain grids of up to 2^32 - 256 threads execute correctly, tools/ubench/biggrid.hip.  Kept as the tested shape.)
        const uint32_t *rows = (const uint32_t *)c->s_rows.p;
        uint64_t *best = (uint64_t *)c->s_best.p;
        for (uint64_t h0 = 0; h0 < nhits; h0 += 1ull << 30) {
            const uint64_t hn = std::min<uint64_t>(1ull << 30, nhits - h0);
            const dim3 hg((uint32_t)((hn + 255) / 256));
            const uint64_t *h = (const uint64_t *)c->s_hits.p + h0;
            switch (rw / 4u) {
            case 1: hipLaunchKernelGGL(k_seed_hamming_min<1>, hg, dim3(256), 0, c->stream, a, h, hn, rows, best); break;
            case 2: hipLaunchKernelGGL(k_seed_hamming_min<2>, hg, dim3(256), 0, c->stream, a, h, hn, rows, best); break;
            case 3: hipLaunchKernelGGL(k_seed_hamming_min<3>, hg, dim3(256), 0, c->stream, a, h, hn, rows, best); break;
            default: hipLaunchKernelGGL(k_seed_hamming_min<4>, hg, dim3(256), 0, c->stream, a, h, hn, rows, best); break;
            }
        }
        HIP_TRY(c, hipGetLastError());
        c->ctr.candidates[pass] += nhits;
        }
        c->ctr.searched[pass] += a.n;
    }
    // the keys that a hit has lowered become the reads' results
    hipLaunchKernelGGL(k_seed_best_store, dim3((uint32_t)((a.n + 255) / 256)), dim3(256), 0, c->stream, a, (const uint64_t *)c->s_best.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PGRC_OK;
}

int pgrc_seedidx_run(pgrc_match_ctx *c, int first_strand, int last_strand) {
    const uint32_t L = c->prm.read_len;
    const char mode = c->prm.mode;
    SeedArgs a;
    a.L = L;
    a.mode = (uint32_t)mode;
    a.P = (mode == 'e') ? 1u : L / c->prm.seed_len;   // targetMismatches + 1 (ReadsMatchers.cpp:236)
    a.m = (mode == 'e') ? L : c->prm.seed_len;
    a.cstride = (mode == 'i') ? a.P : 1u;
    if (a.P == 0 || a.P > 15) { c->err = "modes d/i: 1..15 seed parts supported"; return PGRC_E_PARAM; }
    a.pg = nullptr;
    a.G = c->G;
    a.stride = c->stride;
    a.nwr = (L + 15) / 16;
    a.nmask = nullptr;
    a.kmax = c->prm.max_mismatches;
    a.kmin = c->prm.min_mismatches;
    a.tbase = 0;
    if (c->n == 0) return PGRC_OK;
    int e;
    if (last_strand >= 1) {
        if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) return e;
        c->have_rc = true;
    }
    // limits of the 64-bit hit record (read 28 bits | position in the segment 32 bits | part 4 bits); the knobs force
    // small batches / segments so that tests cover the loops on small inputs
    uint64_t batch = std::min<uint64_t>((1ull << 28) - 1, (1ull << 30) / a.P), seg = (1ull << 32) - 65536;   // (at most 2^30 (read, part) entries per batch: the sizes the tests and the C3 runs cover)
    if (const char *k = getenv("PGRC_SEED_READ_BATCH")) batch = std::max<uint64_t>(1, strtoull(k, nullptr, 10));
    if (const char *k = getenv("PGRC_SEED_SEGMENT")) seg = std::max<uint64_t>(4096, strtoull(k, nullptr, 10));
    const uint32_t *d_nidx = (const uint32_t *)c->nread_idx.p;
    for (uint64_t r0 = 0; r0 < c->n; r0 += batch) {
        const uint64_t r1 = std::min(c->n, r0 + batch);
        a.ibase = r0;
        a.n = r1 - r0;
        a.reads = c->reads2 + r0;                      // word-major: word w of read i at reads[w * stride + i]
        a.pos = (uint64_t *)c->d_pos.p + r0;
        a.rc = (uint8_t *)c->d_rc.p + r0;
        a.mism = (uint8_t *)c->d_mism.p + r0;
        // the batch's part of the side list of reads with N (indexes ascending)
        const auto nlo = std::lower_bound(c->h_nidx.begin(), c->h_nidx.end(), (uint32_t)r0) - c->h_nidx.begin();
        const auto nhi = std::lower_bound(c->h_nidx.begin(), c->h_nidx.end(), (uint32_t)std::min<uint64_t>(r1, 0xFFFFFFFFull)) - c->h_nidx.begin();
        a.nn = (uint64_t)(nhi - nlo);
        a.nflag = a.nn ? (const uint8_t *)c->nread_flag.p + r0 : nullptr;
        a.nidx = d_nidx + nlo;
        a.nascii = (const uint8_t *)c->nread_ascii.p + (uint64_t)nlo * L;
        if ((e = seedidx_batch(c, a, seg, first_strand, last_strand))) return e;
    }
    return PGRC_OK;
}
