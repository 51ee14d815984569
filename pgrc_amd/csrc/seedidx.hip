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
// MI355X design (round 4, its fourth form; the measurements that led here: profiles/r04_seed_scan_experiments.txt).
//   1. The read-part keys go into an open-addressing table in HBM whose slots own a contiguous RANGE of an entry array -- no chains:
//      what a text window matches is read as a stream, by any lane.  Round 5: the table is built by SORTING, not by claiming slots
//      (section 1 below): the (key, entry) pairs are sorted by the key's table order (radix.hip), the sorted entry column IS the entry
//      array, and the distinct keys take their slots by a prefix maximum -- no atomicCAS, no rank atomics, no scan over the table, and
//      a slot is ONE 16-byte word {key, first entry | entries << 32}: a window's key test and its range are one gather.
//   2. ONE scan of the forward text serves both strands (canonical keys, SeedArgs), in two kernels: k_seed_probe finds every
//      window's range (one 8-byte word per window start); k_seed_expand shares the entries of the windows of its stretch among its
//      lanes (prefix sums over the range lengths, a binary search per entry), and the windows with many entries -- repeats, tandem
//      tracts: whole runs of them in one stretch -- go to a list that is sorted by key and taken in units by k_seed_heavy_grouped's
//      blocks from all over the chip (round 5; a wave per window, k_seed_heavy, until then).
//   3. Every (window, entry) pair is a HIT: its Hamming count (popcount on 2-bit words, the read taken from a row-major copy) is
//      computed on the spot, and the reference's sequential rule over a read's hits -- walked in scan order: ascending text
//      position, equal positions in descending part index -- is taken as the lexicographic minimum it amounts to, one atomicMin
//      per acceptable hit on a 64-bit key per read (section 3b).
// No hit records, no sort of hits, no guess of a buffer size (the first run of a context is as fast as the next), no library kernel, and
// the host waits once per batch of reads.
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <vector>

#include "ctx.h"
#include "devutil.h"
#include "scanops.h"

#define SX_EMPTY 0xFFFFFFFFFFFFFFFFull

// two fixed 32-bit tables over the symbol values A0 C1 G2 T3 N4, with table(complement of v) = BIT REVERSAL of table(v) (N: a word
// that is its own reversal).  Bit reversal turns rotl into rotr, so the key of the REVERSE COMPLEMENT of a sequence s[0 .. m) --
// XOR_k rotl(table(comp(s[k])), k), the cyclic polynomial of the reversed, complemented sequence -- is a fixed bijection of the
// key of s itself, half by half:   key_rc = rotl(bit_reverse(key), m - 1)   (key_rc32 below).  Two sequences collide under the one
// iff they collide under the other -- all collisions, the canonical ones of the header and the table-dependent ones on periodic runs
// (32 equal symbols, period-2 and period-4 runs: the XOR over the rotations of one residue class depends on sub-parities of the
// word) alike --, which is what lets ONE scan of the forward text find exactly the candidates of the reference's two (SeedArgs).
__device__ __forceinline__ uint32_t cyc_t0(uint32_t v) {
    return v == 0 ? 0x9E3779B9u : v == 1 ? 0x7F4A7C15u : v == 2 ? 0xA83E52FEu : v == 3 ? 0x9D9EEC79u : 0x10824108u;
}
__device__ __forceinline__ uint32_t cyc_t1(uint32_t v) {
    return v == 0 ? 0xBF58476Du : v == 1 ? 0x1CE4E5B9u : v == 2 ? 0x9DA72738u : v == 3 ? 0xB6E21AFDu : 0x2545A2A4u;
}
__device__ __forceinline__ uint32_t rotl1(uint32_t x) { return (x << 1) | (x >> 31); }
__device__ __forceinline__ uint32_t rotlk(uint32_t x, uint32_t k) { k &= 31u; return k ? (x << k) | (x >> (32u - k)) : x; }
// one half of the key of the reverse complement from the same half of the key (m symbols)
__device__ __forceinline__ uint32_t key_rc32(uint32_t h, uint32_t m) { return rotlk(__brev(h), m - 1u); }
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
    const uint32_t *pg;        // the forward text
    const uint32_t *pg_rc;     // the RC text (Hamming counts of the RC strand's hits)
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
    uint32_t kmax, kmin;
    uint64_t ibase;            // first read of the batch (reads / result pointers are already offset; nidx holds set-wide indexes)
    // the table: slot s = {key in table order (seed_table_key), first entry | entries << 32} owns ent[first .. first + entries);
    // a key's home slot is the low tbits bits of its mixed form; a lookup walks upwards from there (the table has SX_PAD slots behind
    // its 2^tbits: no wrap) until it meets the key or an empty slot
    ulonglong2 *tab;
    uint64_t tmask;
    uint32_t tbits;
    uint32_t *ent;             // entry index | flag << 31, grouped by key
    uint32_t *filter;          // one bit per 2^-fbits of the key space: set iff some indexed key falls there
    uint32_t fshift;           // 64 - fbits: the filter takes the TOP bits of the mixed key, the table its low bits
    // Both strands in ONE scan of the forward text.  A part is indexed under its CANONICAL key: the smaller of the key of the
    // part and the key of its reverse complement, with a flag saying which.  A window of the forward text has both keys as well
    // (the second follows from the first: key_rc32), probes once with the smaller one and carries the same flag: equal flags = the
    // window's key equals the part's, a hit of the forward strand at its start t; different flags = it equals the key of the part's
    // reverse complement, i.e. the part's key equals that of the RC text's window that starts at rc_top - t, rc_top = G - 1 -
    // (m - 1) * cstride (the window's last symbol is the RC window's first): a hit of the RC strand.  The key of the reverse
    // complement is a bijection of the key, so these are exactly the pairs the reference's two scans find with this table --
    // every collision of the rolling hash included.  A window whose two keys are equal is a hit of BOTH strands for the parts
    // under that key, which have equal keys themselves -- flag 0 (a part with flag 1 there has a larger own key: no candidate of
    // either strand).  Each strand has the window starts [0, nwin_all); with cstride > 1 the extent the reference gives a window
    // (m * cstride) is cstride - 1 more than it covers, so the scan walks the starts [0, rc_top] and each strand takes its own range.
    uint32_t want;             // bit 0: hits of the forward strand wanted, bit 1: of the RC strand
    uint64_t nwin_all, rc_top;
    uint64_t *pos;
    uint8_t *rc, *mism;
};

#define SX_FLAG 0x80000000u

__device__ __forceinline__ uint32_t read_code(const SeedArgs &a, uint64_t i, uint32_t x) {
    return (a.reads[(uint64_t)(x >> 4) * a.stride + i] >> (2u * (x & 15u))) & 3u;
}
__device__ __forceinline__ uint32_t ascii_val(uint8_t ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u; }
__device__ __forceinline__ uint32_t part_offset(const SeedArgs &a, uint32_t j) { return a.mode == 'i' ? j : j * a.m; }

// ---- 1. the table, built by sorting (round 5; rounds 3-4 claimed slots with atomicCAS and ranks with atomicAdd: two returning
// random atomics per entry, then a scan over the whole table and a placement pass: 63 of a C3 run's 185 ms in mode d).
//   keys     every (read, part) entry -> the pair (table key of its canonical key, entry index | flag << 31), in entry order;
//   sort     the pairs by the key, all 64 bits (radix.hip): equal keys are adjacent, home slots ascend, and the sorted value column
//            IS the entry array, grouped by key.  A large batch is sorted in three trips through HBM instead of eight: two global
//            passes over the top 16 key bits, then every one of the 65 536 segments in one block's LDS (seedidx_batch);
//   place    the distinct keys d = 0, 1, ... in sorted order take the slots  final(d) = max(home(d), final(d - 1) + 1)
//            = d + max_{j <= d} (home(j) - j):  a prefix MAXIMUM (scanops.h) -- exactly the table that inserting the keys in that
//            order with linear probing would leave, so a lookup that walks upwards from the home slot until it meets the key or an
//            empty slot finds every key: all slots between a key's home and its slot are taken by keys with homes no larger.
// Nothing random is written: the table goes out tile by tile, every line whole (k_seed_place_tiles).
#define SX_PAD 65536ull                  // slots behind the 2^tbits home slots (a cluster at the very end of the table grows into them)

// the key as the table holds it: the mixed key rotated so that its home slot (its low tbits bits) comes first -- ascending table
// keys = ascending home slots; all ones is "empty" (the one key that would map there shares the neighbouring value: 2^-64)
__device__ __forceinline__ uint64_t seed_table_key(uint64_t mixed, uint32_t tbits) {
    const uint64_t k = (mixed << (64u - tbits)) | (mixed >> tbits);
    return k == SX_EMPTY ? SX_EMPTY - 1 : k;
}

__device__ __forceinline__ void seed_pair(const SeedArgs &a, uint64_t kf, uint64_t kr, uint32_t e, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
    const uint64_t key = kr < kf ? kr : kf;                       // canonical (SeedArgs)
    const uint32_t flag = kr < kf ? SX_FLAG : 0u;
    const uint64_t mixed = mix64d(key);
    if (a.filter) {
        const uint64_t fb = mixed >> a.fshift;
        atomicOr(&a.filter[fb >> 5], 1u << (fb & 31u));
    }
    keys[e] = seed_table_key(mixed, a.tbits);
    vals[e] = (uint64_t)(e | flag);
}

__global__ void __launch_bounds__(256) k_seed_keys(const SeedArgs a, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
    // (the two tables side by side in LDS: one 8-byte read per symbol instead of two chains of selects -- the kernel was bound by
    //  its VALU instructions, 21 per symbol; and a word of the read is loaded when the part's next symbol lies in another one,
    //  not once per symbol: 12.9 -> 8.9 -> ... ms at C3)
    __shared__ uint2 tabs[4];
    if (threadIdx.x < 4) tabs[threadIdx.x] = make_uint2(cyc_t0(threadIdx.x), cyc_t1(threadIdx.x));
    __syncthreads();
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n * a.P) return;
    const uint64_t i = e / a.P;
    const uint32_t j = (uint32_t)(e % a.P);
    if (a.nflag && a.nflag[i]) return; // the entries of the reads with N come from their ASCII rows (k_seed_keys_ascii)
    uint32_t x = part_offset(a, j);
    uint32_t h0 = 0, h1 = 0;
    uint32_t cw = 0xFFFFFFFFu, cur = 0;
    for (uint32_t k = 0; k < a.m; k++, x += a.cstride) {
        if ((x >> 4) != cw) {
            cw = x >> 4;
            cur = a.reads[(uint64_t)cw * a.stride + i];
        }
        const uint2 t = tabs[(cur >> (2u * (x & 15u))) & 3u];
        h0 = rotl1(h0) ^ t.x;
        h1 = rotl1(h1) ^ t.y;
    }
    seed_pair(a, key_fix(h0, h1), key_fix(key_rc32(h0, a.m), key_rc32(h1, a.m)), (uint32_t)e, keys, vals);
}

__global__ void __launch_bounds__(256) k_seed_keys_ascii(const SeedArgs a, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
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
    seed_pair(a, key_fix(h0, h1), key_fix(key_rc32(h0, a.m), key_rc32(h1, a.m)), (uint32_t)(i * a.P + j), keys, vals);
}

// From the sorted pairs to what the placement needs -- dstart[d] = where key d's entries start (dstart[nd] = nent), pm[d] = the
// prefix maximum of home(j) - j + SX_BIAS over the keys j <= d, ent[] = the entry words -- segment by segment (any cut of the sorted
// array into consecutive pieces: the sort's 65 536 segments, or pieces of SG_EVEN pairs after the global sort), in two kernels around a
// scan over the SEGMENTS' summaries.  (First form: a flag per pair, a scan over the pairs, a compaction, a second scan over the pairs:
// 7.2 ms at C3 for what is one pass over the keys and one over keys and values.)
//   k_seed_seg_summary   per segment: the keys that start in it, and the maximum of home - rank + SX_BIAS over them, rank counted
//                        from the segment's first key;
//   k_seed_seg_scan      per segment: the number of its first key (exclusive sum) and the prefix maximum that reaches it --
//                        a key's global value is its local one minus the number of its segment's first key;
//   k_seed_seg_write     per segment: dstart, pm, ent.
// (0 is the maximum's identity: a real value is at least SX_BIAS - d > 0.)
#define SX_BIAS (1u << 30)
#define SG_TPB 256
#define SG_EVEN 4096u
__device__ __forceinline__ bool seed_key_starts(const uint64_t *__restrict__ ks, uint32_t i) { return i == 0u || ks[i] != ks[i - 1u]; }

// a block's exclusive sum / inclusive maximum over its threads in thread order (SG_TPB threads), and the block's total / maximum
__device__ __forceinline__ uint32_t sg_block_excl_sum(uint32_t v, uint32_t *smem, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63u) smem[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < SG_TPB / 64; k++) {
        const uint32_t x = smem[k];
        if (k < wv) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}
__device__ __forceinline__ uint32_t sg_block_incl_max(uint32_t v, uint32_t *smem, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o && u > inc) inc = u;
    }
    if (lane == 63u) smem[wv] = inc;
    __syncthreads();
    uint32_t wmax = 0, tot = 0;
    for (uint32_t k = 0; k < SG_TPB / 64; k++) {
        const uint32_t x = smem[k];
        if (k < wv && x > wmax) wmax = x;
        if (x > tot) tot = x;
    }
    __syncthreads();
    *total = tot;
    return inc > wmax ? inc : wmax;
}

__global__ void __launch_bounds__(256) k_seed_even_bounds(uint32_t nent, uint32_t nseg, uint32_t *__restrict__ seg) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nseg) return;
    const uint64_t x = (uint64_t)p * SG_EVEN;
    seg[p] = x < nent ? (uint32_t)x : nent;
}

__global__ void __launch_bounds__(SG_TPB)
k_seed_seg_summary(const uint64_t *__restrict__ ks, const uint32_t *__restrict__ seg, uint32_t tbits, uint32_t *__restrict__ seg_nd, uint32_t *__restrict__ seg_m) {
    __shared__ uint32_t smem[SG_TPB / 64];
    const uint32_t s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
    uint32_t c = 0, m = 0;
    for (uint32_t base = s0; base < s1; base += SG_TPB) {
        const uint32_t i = base + threadIdx.x;
        const bool f = i < s1 && seed_key_starts(ks, i);
        uint32_t tot;
        const uint32_t r = c + sg_block_excl_sum(f ? 1u : 0u, smem, &tot);
        if (f) {
            const uint32_t v = (uint32_t)(ks[i] >> (64u - tbits)) - r + SX_BIAS;
            if (v > m) m = v;
        }
        c += tot;
    }
    uint32_t mm;
    (void)sg_block_incl_max(m, smem, &mm);
    if (threadIdx.x == 0) {
        seg_nd[blockIdx.x] = c;
        seg_m[blockIdx.x] = mm;
    }
}

// one block walks the segments: base_d[p] = keys before segment p, pin[p] = the maximum over the segments q < p that hold a key of
// (seg_m[q] - base_d[q]); *nd_out = all keys
__global__ void __launch_bounds__(SG_TPB)
k_seed_seg_scan(const uint32_t *__restrict__ seg_nd, const uint32_t *__restrict__ seg_m, uint32_t nseg, uint32_t *__restrict__ base_d, uint32_t *__restrict__ pin,
                uint32_t *__restrict__ nd_out) {
    __shared__ uint32_t smem[SG_TPB / 64];
    uint32_t c = 0, carry = 0;
    for (uint32_t base = 0; base < nseg; base += SG_TPB) {
        const uint32_t p = base + threadIdx.x;
        const uint32_t n = p < nseg ? seg_nd[p] : 0u;
        uint32_t tot;
        const uint32_t b = c + sg_block_excl_sum(n, smem, &tot);
        const uint32_t v = (p < nseg && n) ? seg_m[p] - b : 0u;           // the segment's maximum in global numbering
        uint32_t vmax;
        const uint32_t inc = sg_block_incl_max(v, smem, &vmax);
        // exclusive: what reaches segment p is the maximum over the segments before it
        uint32_t prev = __shfl_up(inc, 1, 64);
        __shared__ uint32_t wlast[SG_TPB / 64];
        if ((threadIdx.x & 63u) == 63u) wlast[threadIdx.x >> 6] = inc;
        __syncthreads();
        if ((threadIdx.x & 63u) == 0u) prev = threadIdx.x ? wlast[(threadIdx.x >> 6) - 1u] : 0u;
        __syncthreads();
        if (p < nseg) {
            base_d[p] = b;
            pin[p] = prev > carry ? prev : carry;
        }
        c += tot;
        if (vmax > carry) carry = vmax;
    }
    if (threadIdx.x == 0) *nd_out = c;
}

__global__ void __launch_bounds__(SG_TPB)
k_seed_seg_write(const uint64_t *__restrict__ ks, const uint64_t *__restrict__ vs, const uint32_t *__restrict__ seg, uint32_t nseg, uint32_t nent, uint32_t tbits,
                 const uint32_t *__restrict__ base_d, const uint32_t *__restrict__ pin, uint32_t *__restrict__ dstart, uint32_t *__restrict__ pm,
                 uint32_t *__restrict__ ent) {
    __shared__ uint32_t smem[SG_TPB / 64];
    const uint32_t s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
    const uint32_t d0 = base_d[blockIdx.x];
    uint32_t c = 0, carry = pin[blockIdx.x];
    for (uint32_t base = s0; base < s1; base += SG_TPB) {
        const uint32_t i = base + threadIdx.x;
        const bool f = i < s1 && seed_key_starts(ks, i);
        uint32_t tot;
        const uint32_t r = c + sg_block_excl_sum(f ? 1u : 0u, smem, &tot);
        const uint32_t v = f ? (uint32_t)(ks[i] >> (64u - tbits)) - (d0 + r) + SX_BIAS : 0u;      // home - d + bias (home < 2^31, d < 2^30: no wrap)
        uint32_t vmax;
        const uint32_t inc = sg_block_incl_max(v, smem, &vmax);
        if (f) {
            dstart[d0 + r] = i;
            pm[d0 + r] = inc > carry ? inc : carry;
        }
        if (i < s1) ent[i] = (uint32_t)vs[i];
        c += tot;
        if (vmax > carry) carry = vmax;
    }
    if (blockIdx.x == nseg - 1u && threadIdx.x == 0) dstart[d0 + c] = nent;     // the end of the last key's range
}

// key d goes to slot final(d) = d + (prefix maximum of home - j over j <= d).  The table is WRITTEN tile by tile, every line whole: a
// block owns PL_TILE consecutive slots, finds the keys that land there (final() ascends: two 64-ary searches, one per wave), puts
// them into an LDS image of its slots that starts out empty, and stores the image.  (First form: one kernel filled the 17 GB table
// with "empty", another stored each key's 16 bytes into a line of its own -- 3.2 + 7.5 ms at C3 for what is 17 GB of stores.)
#define PL_TILE 1024u
__global__ void __launch_bounds__(256)
k_seed_place_tiles(const uint64_t *__restrict__ ks, const uint32_t *__restrict__ dstart, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nd_p,
                   uint64_t tslots, ulonglong2 *__restrict__ tab, uint32_t *__restrict__ ovf) {
    __shared__ ulonglong2 tile[PL_TILE];
    __shared__ uint32_t bnd[2];
    const uint32_t nd = *nd_p;
    const uint64_t S = (uint64_t)blockIdx.x * PL_TILE, E = S + PL_TILE < tslots ? S + PL_TILE : tslots;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (wv < 2u) {                                            // the first key whose slot is >= X (nd: none), X = S / E
        const uint64_t X = wv ? E : S;
        uint32_t lo = 0, hi = nd;                             // the answer lies in [lo, hi]
        while (hi > lo) {
            const uint32_t step = (hi - lo + 63u) / 64u;
            const uint64_t q = (uint64_t)lo + (uint64_t)lane * step;
            const bool below = q < hi && q + (uint64_t)pm[q] - SX_BIAS < X;       // final(q) < X: the answer lies behind q
            const uint32_t c = (uint32_t)__popcll(__ballot(below));               // (final() ascends: the lanes that say so are a prefix)
            if (c == 0u) {
                hi = lo;
            } else {
                const uint64_t nhi = (uint64_t)lo + (uint64_t)c * step;
                lo = lo + (c - 1u) * step + 1u;
                if (nhi < hi) hi = (uint32_t)nhi;
            }
        }
        if (lane == 0) bnd[wv] = lo;
    }
    for (uint32_t x = threadIdx.x; x < PL_TILE; x += 256u) tile[x] = make_ulonglong2(SX_EMPTY, 0ull);
    __syncthreads();
    for (uint32_t d = bnd[0] + threadIdx.x; d < bnd[1]; d += 256u) {
        const uint32_t s0 = dstart[d], s1 = dstart[d + 1];
        tile[(uint32_t)((uint64_t)d + (uint64_t)pm[d] - SX_BIAS - S)] = make_ulonglong2(ks[s0], (uint64_t)s0 | ((uint64_t)(s1 - s0) << 32));
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; S + x < E; x += 256u) tab[S + x] = tile[x];
    if (blockIdx.x == 0 && threadIdx.x == 0 && nd && (uint64_t)(nd - 1u) + (uint64_t)pm[nd - 1u] - SX_BIAS >= tslots) *ovf = 1u;   // the last cluster ran over the pad slots
}

// the pairs are sorted by their keys' top SX_SEG_BITS bits: seg[p] = the first pair whose top bits are >= p (seg[2^bits] = nent)
#define SX_SEG_BITS 16u
__global__ void __launch_bounds__(256) k_seed_bounds(const uint64_t *__restrict__ ks, uint32_t nent, uint32_t *__restrict__ seg) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > (1u << SX_SEG_BITS)) return;
    uint32_t lo = 0, hi = nent;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((ks[mid] >> (64u - SX_SEG_BITS)) < (uint64_t)p) lo = mid + 1u; else hi = mid;
    }
    seg[p] = lo;
}

struct SeedMaxOp { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

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

__device__ __forceinline__ uint32_t hamming_vs_text_n(const SeedArgs &a, const uint32_t *text, uint64_t text_w0, uint64_t i, uint64_t trow, uint64_t p) {
    uint32_t mm = 0;
    const uint32_t *src = text + ((p >> 4) - text_w0);
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
// a hit's low bits are never all zero (15 - part >= 1).  So: start keys, the text scanned in any number of launches, in any order,
// every hit's Hamming count taken where the hit is found, one pass over the reads at the end.  A hit's read is random: its words
// come from a ROW-major copy of the batch (one or two lines per hit instead of one per word).  A hit does not go to memory when
// the key in memory is already no larger (a plain load: the key only ever falls) -- random 64-bit atomics run at a sixth of the
// rate of random loads, and the parts of one alignment bring the same key up to P times.
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

template <int RW4>   // rw / 4: the read's row -> registers
__device__ __forceinline__ void load_row(const uint32_t *__restrict__ rows, uint64_t i, uint32_t (&r)[RW4 * 4]) {
    const uint4 *row = (const uint4 *)(rows + i * (uint64_t)(RW4 * 4));
#pragma unroll
    for (int q = 0; q < RW4; q++) {
        const uint4 v = row[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}
template <int RW4>
__device__ __forceinline__ uint32_t hamming_regs_vs_text(const SeedArgs &a, const uint32_t *text, uint64_t text_w0, const uint32_t (&r)[RW4 * 4], uint64_t p) {
    const uint32_t *src = text + ((p >> 4) - text_w0);      // (text: a stretch of the packed text staged in LDS, first word text_w0)
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

// ---- 2. stream the text past the table, in two kernels.
// k_seed_probe: a block stages its stretch of the text in LDS; a thread walks SCAN_R window starts that are cstride apart with
// the O(1) rolling update of the cyclic polynomial (the reference's own scan, cyclichash.h:110-118) -- both keys of a window --
// and probes the table for each: all keys of the thread first, then their probes as independent loads, collisions of all of them
// resolved together.  What a window found goes out as ONE 8-byte word per window start, found or not:
//     first entry (30 bits) | entries (31) | the RC key is the canonical one (bit 62) | equal keys (bit 63);   0 = no key found.
// k_seed_expand: a block takes the words of EXP_WIN consecutive window starts, lists the windows that found a key in LDS, and its
// threads take the list's (window, entry) pairs in turn, whoever found them: pair o belongs to the window f with
// prefix[f] <= o < prefix[f + 1] (binary search in LDS), entry o - prefix[f] of its range -- a read's index and part, a strand, an
// alignment, its Hamming count, its key, one atomicMin.  Why two kernels: the probing needs its registers for loads in flight
// (8 windows per lane), a hit needs few and thrives on resident threads -- fused in one kernel (110 registers, 1024 threads per CU)
// the hits took 118 ms at C3 where they take half of that with 2048 threads per CU; and why ranges instead of the chains of a
// node-based table: a chain costs a dependent gather per entry and leaves a wave waiting for its longest one (the planted repeats
// of the synthetic Pg: 0.9 entries per window on average, dozens for some); ranges are read as streams, by any lane.
#define SCAN_TPB 256
#define SCAN_R 8
#define SCAN_TILE_WORDS (SCAN_TPB * SCAN_R / 16 + 24)
#define WREC_CNT_SH 30u
#define WREC_OFF_MASK ((1ull << WREC_CNT_SH) - 1ull)
#define WREC_CNT_MASK ((1ull << 31) - 1ull)
template <bool FILTER>
__global__ void __launch_bounds__(SCAN_TPB)
k_seed_probe(const SeedArgs a, uint64_t wbase, uint64_t nwin, uint64_t pg_words_alloc, uint64_t *__restrict__ wrec) {
    // window starts [wbase, nwin) of the FORWARD text; wrec[t - wbase] for every one of them
    __shared__ uint32_t tile[SCAN_TILE_WORDS];
    __shared__ uint64_t wst[SCAN_TPB * SCAN_R];
    const uint32_t cs = a.cstride, m = a.m;
    const uint32_t groups = SCAN_TPB / cs;               // runs of SCAN_R * cs consecutive starts, one thread per phase
    const uint32_t per_block = groups * cs * SCAN_R;
    const uint64_t b0 = wbase + (uint64_t)blockIdx.x * per_block;
    const uint64_t w0 = b0 >> 4;
    const uint32_t need = (uint32_t)(((b0 & 15) + per_block + (uint64_t)m * cs + 15) >> 4) + 1;   // <= SCAN_TILE_WORDS
    for (uint32_t w = threadIdx.x; w < need; w += SCAN_TPB) tile[w] = (w0 + w < pg_words_alloc) ? a.pg[w0 + w] : 0u;
    __syncthreads();
    const bool worker = threadIdx.x < groups * cs;
    const uint32_t g = threadIdx.x / cs, phase = threadIdx.x % cs;
    const uint32_t so = g * cs * SCAN_R + phase;          // this thread's first start, relative to b0
    const uint32_t x0 = (uint32_t)(b0 + so - (w0 << 4));  // its tile-relative symbol index
    auto sym = [&](uint32_t x) -> uint32_t { return (tile[x >> 4] >> (2u * (x & 15u))) & 3u; };
    uint32_t h0 = 0, h1 = 0;                              // key of the window (the key of its reverse complement follows from it)
    if (worker)
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t c = sym(x0 + k * cs);
            h0 = rotl1(h0) ^ cyc_t0(c);
            h1 = rotl1(h1) ^ cyc_t1(c);
        }
    const uint32_t mr = m & 31u;
    uint64_t keyv[SCAN_R];                                // the window's canonical key in table order (seed_table_key)
    ulonglong2 kv[SCAN_R];                                // the slot looked at: {key, first entry | entries << 32}
    uint32_t slotv[SCAN_R], fw[SCAN_R], fbit[SCAN_R];      // (slots: the table has at most 2^31, seedidx_batch)
    uint32_t wflag = 0, pal = 0;                          // per start: the RC key is the canonical one; both keys are equal
#pragma unroll
    for (int b = 0; b < SCAN_R; b++) {
        const uint64_t kf = key_fix(h0, h1), kr = key_fix(key_rc32(h0, m), key_rc32(h1, m));
        wflag |= kr < kf ? 1u << b : 0u;
        pal |= kr == kf ? 1u << b : 0u;
        const uint64_t mixed = mix64d(kr < kf ? kr : kf);
        keyv[b] = seed_table_key(mixed, a.tbits);
        slotv[b] = (uint32_t)(mixed & a.tmask);
        fbit[b] = FILTER ? (uint32_t)((mixed >> a.fshift) & 31u) : 0u;
        fw[b] = FILTER ? (uint32_t)(mixed >> a.fshift >> 5) : 0u;
        if (worker) {                                     // roll to the next start (cyclichash.h:110-118)
            const uint32_t xo = x0 + (uint32_t)b * cs;
            const uint32_t co = sym(xo), cn = sym(xo + m * cs);
            const uint32_t o0 = cyc_t0(co), o1 = cyc_t1(co);
            h0 = rotl1(h0) ^ ((o0 << mr) | (mr ? o0 >> (32u - mr) : 0u)) ^ cyc_t0(cn);
            h1 = rotl1(h1) ^ ((o1 << mr) | (mr ? o1 >> (32u - mr) : 0u)) ^ cyc_t1(cn);
        }
    }
    // the filter first: a window whose bit is clear equals no indexed key -- one 4-byte gather in a bitmap of a thirty-second
    // of the table's bytes; only the windows that pass (the hits and ~3 % more) go on to the table, whose probes each
    // cost a random line of their own
#pragma unroll
    for (int b = 0; b < SCAN_R; b++) {
        const uint64_t t = b0 + so + (uint64_t)b * cs;
        fw[b] = (worker && t < nwin) ? (FILTER ? a.filter[fw[b]] : 1u) : 0u;
    }
#pragma unroll
    for (int b = 0; b < SCAN_R; b++) kv[b] = ((fw[b] >> fbit[b]) & 1u) ? a.tab[slotv[b]] : make_ulonglong2(SX_EMPTY, 0ull);   // (the non-temporal hint buys nothing here and costs 13 % on the hits' read rows: profiles/r05_seed_nt_ab.txt)
    // collisions (the slot holds another key) are resolved together: every round issues the next-slot loads of all starts
    // still searching before any of them is looked at
    uint32_t srch = 0, fnd = 0;
#pragma unroll
    for (int b = 0; b < SCAN_R; b++) {
        if (kv[b].x == keyv[b]) fnd |= 1u << b;
        else if (kv[b].x != SX_EMPTY && kv[b].x < keyv[b]) srch |= 1u << b;      // (keys ascend along a cluster: a larger one ends the search too)
    }
    while (__any(srch != 0)) {
#pragma unroll
        for (int b = 0; b < SCAN_R; b++)
            if (srch & (1u << b)) {
                slotv[b] = slotv[b] + 1u;                 // (no wrap: SX_PAD slots lie behind the last home slot, and one of them is empty)
                kv[b] = a.tab[slotv[b]];
            }
#pragma unroll
        for (int b = 0; b < SCAN_R; b++)
            if (srch & (1u << b)) {
                if (kv[b].x == keyv[b]) { fnd |= 1u << b; srch &= ~(1u << b); }
                else if (kv[b].x == SX_EMPTY || kv[b].x > keyv[b]) srch &= ~(1u << b);
            }
    }
    // the range of a key found rides in its slot: one word per window start
    uint32_t lo[SCAN_R], hi[SCAN_R];
#pragma unroll
    for (int b = 0; b < SCAN_R; b++) {
        lo[b] = hi[b] = 0;
        if (fnd & (1u << b)) { lo[b] = (uint32_t)kv[b].y; hi[b] = lo[b] + (uint32_t)(kv[b].y >> 32); }
    }
    // (through LDS: a thread's starts are cstride apart -- written directly, a wave's store would touch 64 lines 8 bytes at a time)
#pragma unroll
    for (int b = 0; b < SCAN_R; b++)
        if (worker)
            wst[so + (uint32_t)b * cs] = hi[b] > lo[b] ? (uint64_t)lo[b] | ((uint64_t)(hi[b] - lo[b]) << WREC_CNT_SH) | ((uint64_t)((wflag >> b) & 1u) << 62) | ((uint64_t)((pal >> b) & 1u) << 63)
                                                        : 0ull;
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < per_block && b0 + x < nwin; x += SCAN_TPB) wrec[b0 + x - wbase] = wst[x];
}

// The words of a strand's text that the alignments of the windows [t_lo, t_hi] of that strand can touch: an alignment starts at most L
// symbols below its window (the part's offset in the read) and reads L symbols + one word of slack.
__device__ __forceinline__ void seed_tile_span(uint64_t t_lo, uint64_t t_hi, uint32_t L, uint64_t *w0, uint32_t *nwords) {
    const uint64_t lo = t_lo > L ? t_lo - L : 0ull;
    *w0 = lo >> 4;
    *nwords = (uint32_t)(((t_hi + L + 32u) >> 4) + 1u - *w0);
}
__device__ __forceinline__ void seed_tile_load(uint32_t *tile, const uint32_t *__restrict__ text, uint64_t w0, uint32_t nwords, uint64_t words_alloc, uint32_t tid, uint32_t nthreads) {
    for (uint32_t w = tid; w < nwords; w += nthreads) tile[w] = (w0 + w < words_alloc) ? text[w0 + w] : 0u;
}

// One hit: entry word `ew` of a window of the forward text at start tf (wflag: the window's RC key is the canonical one; pal: its
// two keys are equal; rep: the second turn of such a window, for the other strand).  tile_fw / tile_rc: the stretch of the forward / the
// RC text that every alignment of the caller's windows lies in, in LDS, from word w0_fw / w0_rc on (seed_tile_span).
// ... in two halves, so that a caller can have the NEXT hit's loads in flight while it counts this one's mismatches:
//   seed_hit_a   decides whether the pair is a hit at all and asks for what the count needs: the read's row, the read's key;
//   seed_hit_b   the count, the key, the minimum.
template <int RW4>
struct SeedHit {
    uint32_t r[RW4 * 4];       // the read's row
    uint64_t bestv;            // the read's key when the hit was set up (a key only ever falls: a stale value can cost an atomic, never a result)
    uint64_t t;                // the window's start in its strand's text
    uint32_t i;                // the read (of the batch)
    uint32_t js;               // part | strand << 8 | live << 16 | the read holds an N << 17
};
template <int RW4>
__device__ __forceinline__ void seed_hit_a(const SeedArgs &a, uint32_t ew, uint32_t wflag, uint32_t pal, uint32_t rep, uint64_t tf,
                                           const uint32_t *__restrict__ rows, const uint64_t *__restrict__ best, uint32_t &cnt0, uint32_t &cnt1, SeedHit<RW4> &h) {
    h.js = 0;
    const uint32_t e = ew & ~SX_FLAG, ef = ew >> 31;
    if (pal && ef) return;                                // (not a candidate: SeedArgs)
    const uint32_t i = e / a.P;
    const uint32_t j = e % a.P;
    const uint64_t shift = part_offset(a, j);
    const uint32_t strand = wflag ^ ef ^ rep;             // the flags of part and window agree = forward
    const uint64_t t = strand ? a.rc_top - tf : tf;       // (tf <= rc_top: the scanned range; wraps otherwise)
    // ReadsMatchers.cpp:308-309 / :375-376 and :311-312 / :378-379: the alignment the hit stands for lies inside the text
    if (!((a.want >> strand) & 1u) || t >= a.nwin_all || shift > t || t - shift + a.L > a.G) return;
    if (strand) cnt1++; else cnt0++;
    const uint32_t nread = (a.nflag && a.nflag[i]) ? 1u : 0u;
    if (!nread) load_row<RW4>(rows, i, h.r);
    h.bestv = best[i];
    h.t = t;
    h.i = i;
    h.js = j | (strand << 8) | (1u << 16) | (nread << 17);
}
template <int RW4>
__device__ __forceinline__ void seed_hit_b(const SeedArgs &a, const SeedHit<RW4> &h, uint64_t *__restrict__ best,
                                           const uint32_t *tile_fw, uint64_t w0_fw, const uint32_t *tile_rc, uint64_t w0_rc) {
    if (!((h.js >> 16) & 1u)) return;
    const uint32_t j = h.js & 0xFFu, strand = (h.js >> 8) & 1u;
    const uint64_t i = h.i, t = h.t, p = t - part_offset(a, j);
    // the text of the strand around the window: staged in LDS by the caller (a hit's eleven text words were eleven gathers of a
    // kernel whose vector-memory instructions, not its bytes, are the limit)
    const uint32_t *text = strand ? tile_rc : tile_fw;
    const uint64_t text_w0 = strand ? w0_rc : w0_fw;
    uint32_t mm;
    if ((h.js >> 17) & 1u) mm = hamming_vs_text_n(a, text, text_w0, i, lower_bound_u32(a.nidx, a.nn, (uint32_t)(i + a.ibase)), p);   // a read with N
    else mm = hamming_regs_vs_text<RW4>(a, text, text_w0, h.r, p);
    if (a.mode == 'e' ? mm != 0u : mm > a.kmax) return;   // ReadsMatchers.cpp:315-319 / :214: no limit a read can have lets it in
    const uint64_t key = ((uint64_t)(mm <= a.kmin ? 0u : mm) << 56) | ((uint64_t)strand << 55) | (t << 15) | ((uint64_t)(15u - j) << 11) | mm;
    if (h.bestv <= key) return;                           // (a plain load: the key only ever falls; random 64-bit atomics run at a sixth of the rate of loads)
    atomicMin((unsigned long long *)&best[i], (unsigned long long)key);
}
template <int RW4>
__device__ __forceinline__ void seed_hit(const SeedArgs &a, uint32_t ew, uint32_t wflag, uint32_t pal, uint32_t rep, uint64_t tf,
                                         const uint32_t *__restrict__ rows, uint64_t *__restrict__ best, uint32_t &cnt0, uint32_t &cnt1,
                                         const uint32_t *tile_fw, uint64_t w0_fw, const uint32_t *tile_rc, uint64_t w0_rc) {
    SeedHit<RW4> h;
    seed_hit_a<RW4>(a, ew, wflag, pal, rep, tf, rows, best, cnt0, cnt1, h);
    seed_hit_b<RW4>(a, h, best, tile_fw, w0_fw, tile_rc, w0_rc);
}

#define EXP_TPB 256
#define EXP_R 8
#define EXP_WIN (EXP_TPB * EXP_R)
#define EXP_TILE_WORDS ((EXP_WIN + 2 * 256 + 64) / 16 + 4)   // words of a text around EXP_WIN windows (reads of at most 255 symbols)
#define HV_TILE_WORDS ((2 * 256 + 64) / 16 + 4)               // ... around one window
#define EXP_HEAVY 32u                    // entries of a window above which it goes to k_seed_heavy
template <int RW4>
__global__ void __launch_bounds__(EXP_TPB)
k_seed_expand(const SeedArgs a, uint64_t wbase, uint64_t nwin, const uint64_t *__restrict__ wrec, const uint32_t *__restrict__ rows,
              uint64_t *__restrict__ best, unsigned long long *__restrict__ counters, uint32_t *__restrict__ hlist, uint32_t heavy_thr, uint64_t pg_words_alloc) {
    // the windows [wbase, nwin) that k_seed_probe described; counters[0 / 1]: hits of the forward / the RC strand
    __shared__ uint32_t f_pre[EXP_WIN + 1];               // entries of a found window (twice that for a window with equal keys; at most 2 * 4096), then their exclusive prefix sums
    __shared__ uint32_t f_off[EXP_WIN];                   // its first entry
    __shared__ uint16_t f_meta[EXP_WIN];                  // window start - the block's first (11 bits) | bit 14: the RC key is the canonical one | bit 15: equal keys
    __shared__ uint32_t scan_tmp[EXP_TPB / 64];
    __shared__ uint16_t h_win[EXP_WIN];                   // the block's heavy windows (below), as offsets from its first window
    __shared__ uint32_t t_fw[EXP_TILE_WORDS], t_rc[EXP_TILE_WORDS];   // the two texts around the block's windows
    __shared__ uint32_t nfound, nheavy, hbase, hitcnt[2];
    if (threadIdx.x == 0) { nfound = 0; nheavy = 0; hitcnt[0] = 0; hitcnt[1] = 0; }
    __syncthreads();
    const uint64_t b0 = wbase + (uint64_t)blockIdx.x * EXP_WIN;
    const uint32_t lane = threadIdx.x & 63u;
    // forward windows [b0, b0 + EXP_WIN) are the RC strand's windows [rc_top - (b0 + EXP_WIN - 1), rc_top - b0]
    uint64_t w0_fw, w0_rc;
    {
        uint32_t nw;
        const uint64_t b1 = b0 + EXP_WIN - 1u;
        seed_tile_span(b0, b1, a.L, &w0_fw, &nw);
        seed_tile_load(t_fw, a.pg, w0_fw, nw, pg_words_alloc, threadIdx.x, EXP_TPB);
        seed_tile_span(a.rc_top > b1 ? a.rc_top - b1 : 0ull, a.rc_top >= b0 ? a.rc_top - b0 : 0ull, a.L, &w0_rc, &nw);
        if (a.want & 2u) seed_tile_load(t_rc, a.pg_rc, w0_rc, nw, pg_words_alloc, threadIdx.x, EXP_TPB);
    }
    uint64_t rv[EXP_R];
#pragma unroll
    for (int q = 0; q < EXP_R; q++) {                                     // (coalesced; all loads first)
        const uint32_t wo = (uint32_t)q * EXP_TPB + threadIdx.x;
        rv[q] = b0 + wo < nwin ? wrec[b0 + wo - wbase] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < EXP_R; q++) {
        const uint32_t wo = (uint32_t)q * EXP_TPB + threadIdx.x;
        const uint64_t r = rv[q];
        // A window with many entries (repeats, tandem tracts: the windows of one tract share a handful of keys with thousands of
        // entries each, and they all lie in ONE block's stretch) is not this block's to expand: it goes on the list of heavy windows
        // that k_seed_heavy's blocks take one at a time, from all over the chip.
        if (r && ((r >> WREC_CNT_SH) & WREC_CNT_MASK) > heavy_thr) {
            h_win[atomicAdd(&nheavy, 1u)] = (uint16_t)wo;
        } else if (r) {
            const uint32_t f = atomicAdd(&nfound, 1u);
            const uint32_t p2 = (uint32_t)(r >> 63);
            f_off[f] = (uint32_t)(r & WREC_OFF_MASK);
            f_pre[f] = (uint32_t)((r >> WREC_CNT_SH) & WREC_CNT_MASK) << p2;
            f_meta[f] = (uint16_t)(wo | ((uint32_t)((r >> 62) & 1u) << 14) | (p2 << 15));
        }
    }
    __syncthreads();
    if (nheavy) {                                         // (one global atomic per block that has any: counters[2] = heavy windows listed)
        if (threadIdx.x == 0) hbase = (uint32_t)atomicAdd(counters + 2, (unsigned long long)nheavy);
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nheavy; x += EXP_TPB) hlist[hbase + x] = (uint32_t)(b0 - wbase) + h_win[x];
    }
    // exclusive prefix sums of the list's lengths (EXP_R consecutive entries per thread)
    const uint32_t nf = nfound;
    uint32_t total;
    {
        uint32_t v[EXP_R], s = 0;
#pragma unroll
        for (int q = 0; q < EXP_R; q++) {
            const uint32_t f = threadIdx.x * EXP_R + q;
            v[q] = f < nf ? f_pre[f] : 0u;
            s += v[q];
        }
        uint32_t inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if (lane >= (uint32_t)o) inc += u;
        }
        if (lane == 63) scan_tmp[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (uint32_t k = 0; k < EXP_TPB / 64; k++) {
            const uint32_t x = scan_tmp[k];
            if (k < (threadIdx.x >> 6)) woff += x;
            tot += x;
        }
        total = tot;
        uint32_t run = woff + inc - s;
#pragma unroll
        for (int q = 0; q < EXP_R; q++) {
            const uint32_t f = threadIdx.x * EXP_R + q;
            if (f < nf) f_pre[f] = run;
            run += v[q];
        }
        if (threadIdx.x == 0) f_pre[nf] = tot;            // the end of the last range
    }
    __syncthreads();
    // the hits: pair o of the block = entry (o - prefix[f]) of the window f that holds it
    uint32_t cnt0 = 0, cnt1 = 0;
    // (one pair ahead: the entry word of the NEXT pair is on its way while this one's read, text and key are worked on)
    auto locate = [&](uint32_t o, uint32_t *meta, uint32_t *rep) -> uint32_t {
        uint32_t flo = 0, fhi = nf;                       // f_pre[flo] <= o < f_pre[fhi]
        while (fhi - flo > 1u) {
            const uint32_t mid = (flo + fhi) >> 1;
            if (f_pre[mid] <= o) flo = mid;
            else fhi = mid;
        }
        *meta = f_meta[flo];
        uint32_t k = o - f_pre[flo];
        *rep = 0;                                         // a window with equal keys: its entries once more, for the other strand
        if (*meta >> 15) {
            const uint32_t half = (f_pre[flo + 1] - f_pre[flo]) >> 1;
            if (k >= half) { k -= half; *rep = 1; }
        }
        return a.ent[f_off[flo] + k];
    };
    // two hits per lane and turn: the row and the key of both are asked for before either's mismatches are counted (the kernel
    // waits for memory two thirds of its time: a hit is entry word -> row and key -> count, three loads in a row)
    uint32_t n_meta[2] = {0, 0}, n_rep[2] = {0, 0}, n_ew[2] = {0, 0};
    for (int u = 0; u < 2; u++)
        if (threadIdx.x + (uint32_t)u * EXP_TPB < total) n_ew[u] = locate(threadIdx.x + (uint32_t)u * EXP_TPB, &n_meta[u], &n_rep[u]);
    for (uint32_t o = threadIdx.x; o < total; o += 2u * EXP_TPB) {
        SeedHit<RW4> h[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t meta = n_meta[u], rep = n_rep[u], ew = n_ew[u];
            h[u].js = 0;
            if (o + (uint32_t)u * EXP_TPB < total)
                seed_hit_a<RW4>(a, ew, (meta >> 14) & 1u, meta >> 15, rep, b0 + (meta & 0x3FFFu), rows, best, cnt0, cnt1, h[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (o + (2u + (uint32_t)u) * EXP_TPB < total) n_ew[u] = locate(o + (2u + (uint32_t)u) * EXP_TPB, &n_meta[u], &n_rep[u]);
#pragma unroll
        for (int u = 0; u < 2; u++) seed_hit_b<RW4>(a, h[u], best, t_fw, w0_fw, t_rc, w0_rc);
    }
    if (cnt0) atomicAdd(&hitcnt[0], cnt0);
    if (cnt1) atomicAdd(&hitcnt[1], cnt1);
    __syncthreads();
    if (threadIdx.x < 2 && hitcnt[threadIdx.x]) atomicAdd(counters + threadIdx.x, (unsigned long long)hitcnt[threadIdx.x]);
}

// The heavy windows of a launch (k_seed_expand): a persistent grid whose WAVES take them one at a time (counters[3] = the next
// one) and walk a window's entries with their 64 lanes.
template <int RW4>
__global__ void __launch_bounds__(EXP_TPB)
k_seed_heavy(const SeedArgs a, uint64_t wbase, const uint64_t *__restrict__ wrec, const uint32_t *__restrict__ hlist, const uint32_t *__restrict__ rows,
             uint64_t *__restrict__ best, unsigned long long *__restrict__ counters, uint64_t pg_words_alloc) {
    __shared__ uint32_t hitcnt[2];
    __shared__ uint32_t t_fw[EXP_TPB / 64][HV_TILE_WORDS], t_rc[EXP_TPB / 64][HV_TILE_WORDS];   // per wave: the two texts around its window
    if (threadIdx.x == 0) { hitcnt[0] = 0; hitcnt[1] = 0; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long nh = counters[2];
    uint32_t cnt0 = 0, cnt1 = 0;
    for (;;) {
        unsigned long long it = 0;
        if (lane == 0) it = atomicAdd(counters + 3, 1ull);
        it = __shfl(it, 0, 64);
        if (it >= nh) break;
        const uint32_t w = hlist[it];
        const uint64_t r = wrec[w];
        const uint32_t off = (uint32_t)(r & WREC_OFF_MASK), pal = (uint32_t)(r >> 63), wflag = (uint32_t)(r >> 62) & 1u;
        const uint32_t cnt = (uint32_t)((r >> WREC_CNT_SH) & WREC_CNT_MASK);
        const uint64_t tf = wbase + w, trc = a.rc_top >= tf ? a.rc_top - tf : 0ull;
        const uint32_t wv = threadIdx.x >> 6;
        uint64_t w0_fw, w0_rc;
        uint32_t nw;
        __builtin_amdgcn_wave_barrier();                  // (the lanes are done with the previous window's tiles)
        seed_tile_span(tf, tf, a.L, &w0_fw, &nw);
        seed_tile_load(t_fw[wv], a.pg, w0_fw, nw, pg_words_alloc, lane, 64u);
        seed_tile_span(trc, trc, a.L, &w0_rc, &nw);
        if (a.want & 2u) seed_tile_load(t_rc[wv], a.pg_rc, w0_rc, nw, pg_words_alloc, lane, 64u);
        __threadfence_block();                            // the wave's LDS writes before its lanes' reads
        __builtin_amdgcn_wave_barrier();
        for (uint32_t rep = 0; rep <= pal; rep++) {
            // (measured and dropped: the NEXT hit's row and key requested before this hit's mismatches are counted -- seed_hit_a /
            //  seed_hit_b one hit apart: 80 registers instead of 62, mode d's heavy windows 25.5 -> 23.1 ms, mode i's 51.7 -> 54.2)
            uint32_t n_ew = lane < cnt ? a.ent[off + lane] : 0u;           // (one entry ahead)
            for (uint32_t k = lane; k < cnt; k += 64u) {
                const uint32_t ew = n_ew;
                if (k + 64u < cnt) n_ew = a.ent[off + k + 64u];
                seed_hit<RW4>(a, ew, wflag, pal, rep, tf, rows, best, cnt0, cnt1, t_fw[wv], w0_fw, t_rc[wv], w0_rc);
            }
        }
    }
    if (cnt0) atomicAdd(&hitcnt[0], cnt0);
    if (cnt1) atomicAdd(&hitcnt[1], cnt1);
    __syncthreads();
    if (threadIdx.x < 2 && hitcnt[threadIdx.x]) atomicAdd(counters + threadIdx.x, (unsigned long long)hitcnt[threadIdx.x]);
}

// ---- the heavy windows, grouped by their KEY (round 5).  The windows of a launch that found a key with many entries are few
// KEYS met many times -- 563 keys at C3 in mode d, 0.7 M windows of the text on them: repeats of the text --, and going window by
// window (k_seed_heavy) every (window, entry) pair loads the entry's read again and looks at the read's key in memory: a kernel that
// waits.  Here the list of heavy windows is sorted by key, a block takes up to HVG_WC windows of ONE key, and a thread keeps ITS entry
// -- the read's row in registers, the smallest key of its hits so far -- while the windows pass: per pair only the text words come
// from memory (staged in LDS, HVG_SB windows at a time), and a read's key in memory is touched once per (entry, unit).
//   k_seed_hv_records   heavy window x -> (first entry of its key) << 32 | window        (sorted by the key with radix.hip)
//   k_seed_hv_units     runs of one key, cut into units of at most HVG_WC windows: first | count << 32
//   k_seed_heavy_grouped  a persistent grid takes the units (counters[3]; counters[4] = their number)
#define HVG_TPB 256
#define HVG_WC 256u
#define HVG_SB 32u
#define HVG_EC 256u       // entries of a key per unit
__global__ void __launch_bounds__(256)
k_seed_hv_records(const uint64_t *__restrict__ wrec, const uint32_t *__restrict__ hlist, uint32_t nh, uint64_t *__restrict__ recs) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nh) return;
    const uint32_t w = hlist[x];
    recs[x] = ((wrec[w] & WREC_OFF_MASK) << 32) | w;
}
__global__ void __launch_bounds__(256)
k_seed_hv_units(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ wrec, uint32_t nh, uint64_t *__restrict__ units, unsigned long long cap,
                unsigned long long *__restrict__ nunits) {
    // units == nullptr: only counted.  A unit: first record (32 bits) | windows (9) << 32 | which HVG_EC entries of the key (23) << 41
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nh) return;
    const uint32_t key = (uint32_t)(recs[x] >> 32);
    if (x && (uint32_t)(recs[x - 1] >> 32) == key) return;            // not the first window of its key
    uint32_t lo = x + 1u, hi = nh;                                    // the first record of the next key
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint32_t)(recs[mid] >> 32) == key) lo = mid + 1u; else hi = mid;
    }
    const uint32_t glen = lo - x;
    const uint32_t cnt = (uint32_t)((wrec[(uint32_t)recs[x]] >> WREC_CNT_SH) & WREC_CNT_MASK);
    const uint32_t nec = (cnt + HVG_EC - 1u) / HVG_EC, nwc = (glen + HVG_WC - 1u) / HVG_WC;
    if (!units) {
        atomicAdd(nunits, (unsigned long long)nec * nwc);
        return;
    }
    unsigned long long at = atomicAdd(nunits, (unsigned long long)nec * nwc);
    for (uint32_t wc = 0; wc < glen; wc += HVG_WC) {
        const uint32_t n = glen - wc < HVG_WC ? glen - wc : HVG_WC;
        for (uint32_t k = 0; k < nec; k++, at++)
            if (at < cap) units[at] = (uint64_t)(x + wc) | ((uint64_t)n << 32) | ((uint64_t)k << 41);
    }
}
template <int RW4>
__global__ void __launch_bounds__(HVG_TPB)
k_seed_heavy_grouped(const SeedArgs a, uint64_t wbase, const uint64_t *__restrict__ wrec, const uint64_t *__restrict__ recs, const uint64_t *__restrict__ units,
                     const uint32_t *__restrict__ rows, uint64_t *__restrict__ best, unsigned long long *__restrict__ counters, uint64_t pg_words_alloc) {
    __shared__ uint32_t t_fw[HVG_SB][HV_TILE_WORDS], t_rc[HVG_SB][HV_TILE_WORDS];      // the two texts around each of the staged windows
    __shared__ uint64_t s_tf[HVG_SB], s_w0f[HVG_SB], s_w0r[HVG_SB];
    __shared__ uint32_t s_fl[HVG_SB];                                                // wflag | pal << 1
    __shared__ uint32_t hitcnt[2];
    __shared__ unsigned long long s_unit;
    if (threadIdx.x == 0) { hitcnt[0] = 0; hitcnt[1] = 0; }
    const unsigned long long nunits = counters[4];
    uint32_t cnt0 = 0, cnt1 = 0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_unit = atomicAdd(counters + 3, 1ull);
        __syncthreads();
        const unsigned long long it = s_unit;
        if (it >= nunits) break;
        const uint64_t u = units[it];
        const uint32_t x0 = (uint32_t)u, nw = (uint32_t)(u >> 32) & 0x1FFu, ek = (uint32_t)(u >> 41);
        const uint64_t r0 = wrec[(uint32_t)recs[x0]];
        const uint32_t cnt_all = (uint32_t)((r0 >> WREC_CNT_SH) & WREC_CNT_MASK);
        const uint32_t off = (uint32_t)(r0 & WREC_OFF_MASK) + ek * HVG_EC, cnt = cnt_all - ek * HVG_EC < HVG_EC ? cnt_all - ek * HVG_EC : HVG_EC;   // this unit's entries
        // the entries of the key, ew at a time side by side; with fewer than HVG_TPB of them several threads share an entry and
        // split the windows between them
        uint32_t ew_ = 1u;
        while (ew_ < cnt && ew_ < (uint32_t)HVG_TPB) ew_ <<= 1;
        const uint32_t streams = (uint32_t)HVG_TPB / ew_, stream = threadIdx.x / ew_;
        for (uint32_t ec = 0; ec < cnt; ec += ew_) {
            const uint32_t eidx = ec + (threadIdx.x & (ew_ - 1u));
            const bool live = eidx < cnt;
            const uint32_t ew = live ? a.ent[off + eidx] : 0u;
            const uint32_t e = ew & ~SX_FLAG, ef = ew >> 31;
            const uint32_t i = e / a.P, j = e % a.P;
            const uint64_t shift = part_offset(a, j);
            const bool nread = live && a.nflag && a.nflag[i];
            const uint64_t trow = nread ? lower_bound_u32(a.nidx, a.nn, (uint32_t)(i + a.ibase)) : 0ull;
            uint32_t row[RW4 * 4];
            if (live && !nread) load_row<RW4>(rows, i, row);
            uint64_t mine = BK_NONE;                                  // the smallest key of this entry's hits in this unit
            for (uint32_t sb = 0; sb < nw; sb += HVG_SB) {
                const uint32_t nsb = nw - sb < HVG_SB ? nw - sb : HVG_SB;
                __syncthreads();                                      // (the previous windows' tiles are done with)
                {
                    const uint32_t q = threadIdx.x / (HVG_TPB / HVG_SB), sub = threadIdx.x % (HVG_TPB / HVG_SB);   // 16 threads per window
                    if (q < nsb) {
                        const uint32_t w = (uint32_t)recs[x0 + sb + q];
                        const uint64_t r = wrec[w];
                        const uint64_t tf = wbase + w, trc = a.rc_top >= tf ? a.rc_top - tf : 0ull;
                        uint64_t w0f, w0r;
                        uint32_t nwf, nwr2;
                        seed_tile_span(tf, tf, a.L, &w0f, &nwf);
                        seed_tile_span(trc, trc, a.L, &w0r, &nwr2);
                        seed_tile_load(t_fw[q], a.pg, w0f, nwf, pg_words_alloc, sub, HVG_TPB / HVG_SB);
                        if (a.want & 2u) seed_tile_load(t_rc[q], a.pg_rc, w0r, nwr2, pg_words_alloc, sub, HVG_TPB / HVG_SB);
                        if (sub == 0) {
                            s_tf[q] = tf;
                            s_w0f[q] = w0f;
                            s_w0r[q] = w0r;
                            s_fl[q] = ((uint32_t)(r >> 62) & 1u) | ((uint32_t)(r >> 63) << 1);
                        }
                    }
                }
                __syncthreads();
                if (live) {
                    for (uint32_t q = stream; q < nsb; q += streams) {
                        const uint64_t tf = s_tf[q];
                        const uint32_t wflag = s_fl[q] & 1u, pal = s_fl[q] >> 1;
                        if (pal && ef) continue;                      // (not a candidate: SeedArgs)
                        for (uint32_t rep = 0; rep <= pal; rep++) {
                            const uint32_t strand = wflag ^ ef ^ rep;
                            const uint64_t t = strand ? a.rc_top - tf : tf;
                            if (!((a.want >> strand) & 1u) || t >= a.nwin_all || shift > t || t - shift + a.L > a.G) continue;   // (seed_hit_a)
                            if (strand) cnt1++; else cnt0++;
                            const uint64_t p = t - shift;
                            const uint32_t *text = strand ? t_rc[q] : t_fw[q];
                            const uint64_t text_w0 = strand ? s_w0r[q] : s_w0f[q];
                            const uint32_t mm = nread ? hamming_vs_text_n(a, text, text_w0, i, trow, p) : hamming_regs_vs_text<RW4>(a, text, text_w0, row, p);
                            if (a.mode == 'e' ? mm != 0u : mm > a.kmax) continue;
                            const uint64_t key = ((uint64_t)(mm <= a.kmin ? 0u : mm) << 56) | ((uint64_t)strand << 55) | (t << 15) | ((uint64_t)(15u - j) << 11) | mm;
                            if (key < mine) mine = key;
                        }
                    }
                }
            }
            if (live && mine != BK_NONE && mine < best[i]) atomicMin((unsigned long long *)&best[i], (unsigned long long)mine);
        }
    }
    if (cnt0) atomicAdd(&hitcnt[0], cnt0);
    if (cnt1) atomicAdd(&hitcnt[1], cnt1);
    __syncthreads();
    if (threadIdx.x < 2 && hitcnt[threadIdx.x]) atomicAdd(counters + threadIdx.x, (unsigned long long)hitcnt[threadIdx.x]);
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




// One batch of reads (at most 2^30 entries: an entry index and a slot number leave a bit for the flag) against both strands.
// Reads are independent, so batch after batch is the reference's loop over the reads; the reduction is a minimum over all
// hits of a read (section 3b), so the text is scanned in launches of at most seg_windows window starts (8 + 4 bytes of scratch
// per start), in any order.
static int seedidx_batch(pgrc_match_ctx *c, SeedArgs a, uint64_t seg_windows, int first_strand, int last_strand) {
    const uint64_t span = (uint64_t)a.m * a.cstride;  // extent of a text window
    const uint32_t L = a.L;
    const uint64_t nent = a.n * a.P;
    uint64_t tsize = 1024;
    uint32_t tbits = 10;
    while (tsize < 2 * nent) { tsize <<= 1; tbits++; }   // (4 / 8 * nent: the exact matcher at C3 12 / 16 % faster, modes d / i unchanged; the filter below does better)
    const uint64_t tslots = tsize + SX_PAD;
    int e;
    // the table; the (key, entry) pairs twice (the sort's ping-pong); the entry array; per distinct key (at most nent): start of its
    // range, prefix maximum of home - number; per segment of the sorted pairs: bounds, keys, maximum, first key's number, maximum before it
    const bool by_segments = c->opt.seed_sort >= 0 ? c->opt.seed_sort != 0 : nent >= (1ull << 20);
    const uint32_t nseg = by_segments ? 1u << SX_SEG_BITS : (uint32_t)((nent + SG_EVEN - 1) / SG_EVEN), OVL_CAP = 255u;
    if ((e = pgrc_buf_ensure(c, c->s_keys, tslots * sizeof(ulonglong2)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_vals, 4 * nent * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_tab, (3 * nent + 8) * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_seg, (5 * ((size_t)nseg + 1) + OVL_CAP + 1) * sizeof(uint32_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->s_tmp, 128))) return e;        // counters, flags
    // the filter: 32 bits per indexed key (3 % of the windows of a random text pass it by chance), at most 2^36 bits.  It
    // pays when most windows of the text equal no key -- the exact matcher at C3: 100 M keys against 1.9 G windows, 0.20 ->
    // 0.16 s -- and costs a dependent round trip where many do (modes d / i with four parts per read: 0.42 -> 0.43 s): used
    // when there is at most one key per eight text positions (PGRC_SEED_FILTER=0 / 1: never / always, tests and A/B runs)
    bool use_filter = nent * 8 <= c->G;
    if (c->opt.seed_filter >= 0) use_filter = c->opt.seed_filter != 0;
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
    a.tab = (ulonglong2 *)c->s_keys.p;
    a.tmask = tsize - 1;
    a.tbits = tbits;
    uint64_t *kA = (uint64_t *)c->s_vals.p, *kB = kA + nent, *vA = kB + nent, *vB = vA + nent;
    uint32_t *dstart = (uint32_t *)c->s_tab.p, *pm = dstart + nent + 1;
    a.ent = pm + nent;
    uint32_t *seg = (uint32_t *)c->s_seg.p, *seg_nd = seg + nseg + 1, *seg_m = seg_nd + nseg + 1, *seg_base = seg_m + nseg + 1, *seg_pin = seg_base + nseg + 1,
             *ovl = seg_pin + nseg + 1;
    if (a.nn) {
        if ((e = pgrc_buf_ensure(c, c->s_nmask, a.nn * a.nwr * sizeof(uint16_t)))) return e;
        a.nmask = (const uint16_t *)c->s_nmask.p;
        hipLaunchKernelGGL(k_seed_nmask, dim3((uint32_t)((a.nn * a.nwr + 255) / 256)), dim3(256), 0, c->stream, a.nascii, a.nn, L,
                           a.nwr, (uint16_t *)c->s_nmask.p);
    }
    unsigned long long *counters = (unsigned long long *)c->s_tmp.p;   // [0 / 1] hits of the forward / the RC strand
    uint32_t *nd_dev = (uint32_t *)((char *)c->s_tmp.p + 64), *ovf_dev = nd_dev + 1;   // distinct keys; "the table ran over"
    HIP_TRY(c, hipMemsetAsync(c->s_tmp.p, 0, 128, c->stream));
    // 1. the table (section 1): pairs, sort, distinct keys, prefix maximum, placement
    hipLaunchKernelGGL(k_seed_keys, dim3((uint32_t)((nent + 255) / 256)), dim3(256), 0, c->stream, a, kA, vA);
    if (a.nn)
        hipLaunchKernelGGL(k_seed_keys_ascii, dim3((uint32_t)((a.nn * a.P + 255) / 256)), dim3(256), 0, c->stream, a, kA, vA);
    HIP_TRY(c, hipGetLastError());
    uint64_t *ks = nullptr, *vs = nullptr;
    // the sort.  A large batch: two global passes over the keys' top 16 bits cut the pairs into 65 536 segments (~4 600 pairs at
    // C3: the keys are hash values), and every segment is sorted by the other 48 bits inside ONE block's LDS (radix.hip,
    // k_rx_segments): three trips through HBM instead of eight (31 -> ~12 ms at C3).  The few segments that hold more than a block
    // takes -- a key with thousands of entries sits in them -- are sorted as ranges of their own with the global passes.
    // PGRC_SEED_SORT=full / segments forces the one or the other (tests).
    if (by_segments) {
        if ((e = pgrc_radix_sort_pairs_u64(c, kA, kB, vA, vB, nent, 64u - SX_SEG_BITS, 64, c->s_sort, &ks, &vs))) return e;
        uint64_t *ko = ks == kA ? kB : kA, *vo = vs == vA ? vB : vA;     // the other halves of the ping-pong: scratch from here on
        HIP_TRY(c, hipMemsetAsync(ovl, 0, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_seed_bounds, dim3((nseg + 1 + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)ks, (uint32_t)nent, seg);
        HIP_TRY(c, hipGetLastError());
        if ((e = pgrc_radix_sort_segments_pairs_u64(c, ks, vs, seg, nseg, 0, 64u - SX_SEG_BITS, ovl, OVL_CAP))) return e;
        uint32_t h_ovl[OVL_CAP + 1];
        HIP_TRY(c, hipMemcpyAsync(h_ovl, ovl, sizeof h_ovl, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (h_ovl[0] > OVL_CAP) {
            // (more large segments than the list holds: the whole array once more, by all its bits)
            if ((e = pgrc_radix_sort_pairs_u64(c, ks, ko, vs, vo, nent, 0, 64, c->s_sort, &ks, &vs))) return e;
        } else if (h_ovl[0]) {
            std::vector<uint32_t> bounds(2 * (size_t)h_ovl[0]);
            for (uint32_t k = 0; k < h_ovl[0]; k++)
                HIP_TRY(c, hipMemcpyAsync(&bounds[2 * k], seg + h_ovl[1 + k], 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (uint32_t k = 0; k < h_ovl[0]; k++) {
                const uint64_t s0 = bounds[2 * k], n = bounds[2 * k + 1] - s0;
                uint64_t *kr = nullptr, *vr = nullptr;
                if ((e = pgrc_radix_sort_pairs_u64(c, ks + s0, ko + s0, vs + s0, vo + s0, n, 0, 64u - SX_SEG_BITS, c->s_sort, &kr, &vr))) return e;
                if (kr != ks + s0) {
                    HIP_TRY(c, hipMemcpyAsync(ks + s0, kr, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
                    HIP_TRY(c, hipMemcpyAsync(vs + s0, vr, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
                }
            }
        }
    } else {
        if ((e = pgrc_radix_sort_pairs_u64(c, kA, kB, vA, vB, nent, 0, 64, c->s_sort, &ks, &vs))) return e;
        hipLaunchKernelGGL(k_seed_even_bounds, dim3((nseg + 1 + 255) / 256), dim3(256), 0, c->stream, (uint32_t)nent, nseg, seg);   // (pieces of SG_EVEN pairs)
    }
    // dstart, pm, ent (and the number of keys) from the sorted pairs, segment by segment
    hipLaunchKernelGGL(k_seed_seg_summary, dim3(nseg), dim3(SG_TPB), 0, c->stream, (const uint64_t *)ks, (const uint32_t *)seg, tbits, seg_nd, seg_m);
    hipLaunchKernelGGL(k_seed_seg_scan, dim3(1), dim3(SG_TPB), 0, c->stream, (const uint32_t *)seg_nd, (const uint32_t *)seg_m, nseg, seg_base, seg_pin, nd_dev);
    hipLaunchKernelGGL(k_seed_seg_write, dim3(nseg), dim3(SG_TPB), 0, c->stream, (const uint64_t *)ks, (const uint64_t *)vs, (const uint32_t *)seg, nseg, (uint32_t)nent, tbits,
                       (const uint32_t *)seg_base, (const uint32_t *)seg_pin, dstart, pm, a.ent);
    hipLaunchKernelGGL(k_seed_place_tiles, dim3((uint32_t)((tslots + PL_TILE - 1) / PL_TILE)), dim3(256), 0, c->stream, (const uint64_t *)ks, (const uint32_t *)dstart,
                       (const uint32_t *)pm, (const uint32_t *)nd_dev, tslots, a.tab, ovf_dev);
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

    // ONE scan of the forward text finds the hits of both strands (SeedArgs), in launches of at most seg_windows window starts
    a.pg = (const uint32_t *)c->pg2[0].p;
    a.pg_rc = (const uint32_t *)c->pg2[1].p;
    a.want = (first_strand == 0 ? 1u : 0u) | (last_strand == 1 ? 2u : 0u);
    const uint64_t nwin_all = c->G >= span ? c->G - span + 1 : 0;       // window starts of either strand (none: the reference's scan loops are empty / undefined there)
    a.nwin_all = nwin_all;
    a.rc_top = nwin_all ? nwin_all - 1u + (a.cstride - 1u) : 0;
    const uint64_t nscan = !nwin_all ? 0 : (a.want & 2u) ? a.rc_top + 1u : nwin_all;
    const uint32_t per_block = (SCAN_TPB / a.cstride) * a.cstride * SCAN_R;
    const uint32_t *rows = (const uint32_t *)c->s_rows.p;
    uint64_t *best = (uint64_t *)c->s_best.p;
    const uint64_t pgw = c->pg_words + PGRC_PG_PAD_WORDS;
    const uint64_t wseg = std::min<uint64_t>(seg_windows, nscan);
    if ((e = pgrc_buf_ensure(c, c->s_hits, std::max<uint64_t>(wseg, 1) * (sizeof(uint64_t) + sizeof(uint32_t))))) return e;     // one word per window start of a launch + the list of the heavy ones
    uint64_t *wrec = (uint64_t *)c->s_hits.p;
    uint32_t *hlist = (uint32_t *)(wrec + std::max<uint64_t>(wseg, 1));
    for (uint64_t w0 = 0; w0 < nscan; w0 += wseg) {
        const uint64_t nwin = std::min(nscan, w0 + wseg);
        const dim3 pgrid((uint32_t)((nwin - w0 + per_block - 1) / per_block)), egrid((uint32_t)((nwin - w0 + EXP_WIN - 1) / EXP_WIN));
        if (a.filter) hipLaunchKernelGGL(k_seed_probe<true>, pgrid, dim3(SCAN_TPB), 0, c->stream, a, w0, nwin, pgw, wrec);
        else hipLaunchKernelGGL(k_seed_probe<false>, pgrid, dim3(SCAN_TPB), 0, c->stream, a, w0, nwin, pgw, wrec);
        HIP_TRY(c, hipMemsetAsync(counters + 2, 0, 3 * sizeof(unsigned long long), c->stream));       // heavy windows listed / units taken / units
        const dim3 hgrid((uint32_t)c->num_cus * 8u);
        const uint32_t heavy_thr = c->opt.seed_heavy ? c->opt.seed_heavy : EXP_HEAVY;
        const bool grouped = c->opt.seed_heavy_form != 0;            // PGRC_SEED_HEAVY_FORM=window: a wave per heavy window (rounds 4-5a)
#define EXP_LAUNCH(R)                                                                                                                                       \
        hipLaunchKernelGGL((k_seed_expand<R>), egrid, dim3(EXP_TPB), 0, c->stream, a, w0, nwin, (const uint64_t *)wrec, rows, best, counters, hlist, heavy_thr, pgw); \
        if (!grouped) hipLaunchKernelGGL((k_seed_heavy<R>), hgrid, dim3(EXP_TPB), 0, c->stream, a, w0, (const uint64_t *)wrec, (const uint32_t *)hlist, rows, best, counters, pgw)
        switch (rw / 4u) {
        case 1: EXP_LAUNCH(1); break;
        case 2: EXP_LAUNCH(2); break;
        case 3: EXP_LAUNCH(3); break;
        default: EXP_LAUNCH(4); break;
        }
#undef EXP_LAUNCH
        HIP_TRY(c, hipGetLastError());
        if (grouped) {
            // the heavy windows by key (k_seed_heavy_grouped): their number decides the sort's launches -- the one wait per launch
            unsigned long long nheavy = 0;
            HIP_TRY(c, hipMemcpyAsync(&nheavy, counters + 2, sizeof nheavy, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (nheavy) {
                const uint32_t nhv = (uint32_t)nheavy;
                if ((e = pgrc_buf_ensure(c, c->s_hv, 2 * (size_t)nhv * sizeof(uint64_t)))) return e;
                uint64_t *recA = (uint64_t *)c->s_hv.p, *recB = recA + nhv, *recs = nullptr;
                hipLaunchKernelGGL(k_seed_hv_records, dim3((nhv + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)wrec, (const uint32_t *)hlist, nhv, recA);
                if ((e = pgrc_radix_sort_u64(c, recA, recB, nhv, 32, 32 + WREC_CNT_SH, c->s_sort, &recs))) return e;
                // the units: counted, then listed (a key's windows x its entries, in pieces of HVG_WC x HVG_EC)
                hipLaunchKernelGGL(k_seed_hv_units, dim3((nhv + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)recs, (const uint64_t *)wrec, nhv, (uint64_t *)nullptr, 0ull, counters + 4);
                unsigned long long nunits = 0;
                HIP_TRY(c, hipMemcpyAsync(&nunits, counters + 4, sizeof nunits, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if ((e = pgrc_buf_ensure(c, c->s_hvu, (size_t)nunits * sizeof(uint64_t)))) return e;
                uint64_t *units = (uint64_t *)c->s_hvu.p;
                HIP_TRY(c, hipMemsetAsync(counters + 4, 0, sizeof(unsigned long long), c->stream));
                hipLaunchKernelGGL(k_seed_hv_units, dim3((nhv + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)recs, (const uint64_t *)wrec, nhv, units, nunits, counters + 4);
#define HVG_LAUNCH(R) hipLaunchKernelGGL((k_seed_heavy_grouped<R>), hgrid, dim3(HVG_TPB), 0, c->stream, a, w0, (const uint64_t *)wrec, (const uint64_t *)recs, (const uint64_t *)units, rows, best, counters, pgw)
                switch (rw / 4u) {
                case 1: HVG_LAUNCH(1); break;
                case 2: HVG_LAUNCH(2); break;
                case 3: HVG_LAUNCH(3); break;
                default: HVG_LAUNCH(4); break;
                }
#undef HVG_LAUNCH
                HIP_TRY(c, hipGetLastError());
            }
        }
    }
    // the keys that a hit has lowered become the reads' results
    hipLaunchKernelGGL(k_seed_best_store, dim3((uint32_t)((a.n + 255) / 256)), dim3(256), 0, c->stream, a, (const uint64_t *)c->s_best.p);
    HIP_TRY(c, hipGetLastError());
    unsigned long long nh[2] = {0, 0};
    uint32_t tinfo[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(nh, counters, sizeof nh, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(tinfo, nd_dev, sizeof tinfo, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (tinfo[1]) { c->err = "modes d/i/e: a cluster of keys ran over the end of the table"; return PGRC_E_DEVICE; }
    c->ctr.candidates[0] += nh[0];
    c->ctr.candidates[1] += nh[1];
    for (int pass = first_strand; pass <= last_strand; pass++) c->ctr.searched[pass] += a.n;
    return PGRC_OK;
}

int pgrc_seedidx_run(pgrc_match_ctx *c, int first_strand, int last_strand) {
    const uint32_t L = c->prm.read_len;
    const char mode = c->prm.mode;
    SeedArgs a;
    memset(&a, 0, sizeof a);
    a.L = L;
    a.mode = (uint32_t)mode;
    a.P = (mode == 'e') ? 1u : L / c->prm.seed_len;   // targetMismatches + 1 (ReadsMatchers.cpp:236)
    a.m = (mode == 'e') ? L : c->prm.seed_len;
    a.cstride = (mode == 'i') ? a.P : 1u;
    if (a.P == 0 || a.P > 15) { c->err = "modes d/i: 1..15 seed parts supported"; return PGRC_E_PARAM; }
    a.G = c->G;
    a.stride = c->stride;
    a.nwr = (L + 15) / 16;
    a.kmax = c->prm.max_mismatches;
    a.kmin = c->prm.min_mismatches;
    if (c->n == 0) return PGRC_OK;
    int e;
    if (last_strand >= 1) {
        if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) return e;
        c->have_rc = true;
    }
    // a batch holds at most 2^30 entries (SX_FLAG); a launch of the scan 2^29 window starts (6 GB of scratch).  The knobs force
    // small batches / launches so that tests cover the loops on small inputs
    uint64_t batch = (1ull << 30) / a.P, seg = 1ull << 29;
    if (c->opt.seed_read_batch) batch = std::max<uint64_t>(1, std::min<uint64_t>(batch, c->opt.seed_read_batch));
    if (c->opt.seed_segment) seg = std::max<uint64_t>(4096, std::min<uint64_t>(seg, c->opt.seed_segment));
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
