// export.hip -- row f1: the export of the matches as the SoA streams of the output reads list, on the device.
//
// Reference behaviour restated (not translated):
//   DefaultReadsMatcher::exportMatchesInPgOrder        matching/ReadsMatchers.cpp:563-595
//   DefaultReadsMatcher::exportMatchesInOriginalOrder  matching/ReadsMatchers.cpp:597-675
//   SeparatedPseudoGenomeOutputBuilder::writeReadsFromIterator / writeExtraReadEntry / writeReadEntry
//                                                      pseudogenome/persistence/SeparatedPseudoGenomePersistence.cpp:961-1019
//   AbstractReadsApproxMatcher::updateEntry            matching/ReadsMatchers.cpp:548-559
//
// In Pg order the reference walks the matched reads sorted by position, copies the entries of the reads list that is
// already on the pseudogenome in front of each of them (every old entry with a SMALLER position: an old entry at the
// same position follows the new one), and appends one record per entry to seven byte streams: offset delta to the
// previously written entry, original index, RC flag, mismatch count, mismatch codes, and the mismatch offsets coded
// backwards from the read end.  All of it is a merge of two sorted lists plus per-entry independent work:
//   positions of the old list  = prefix sum of its offset deltas                     (k_scan_*)
//   rank of every entry        = its index + a binary search in the OTHER list       (k_export_place_*)
//   offset deltas              = difference of neighbours in the merged order        (k_export_offsets)
//   mismatch lists             = prefix sum of the counts, then one thread per entry (k_export_mismatches)
// What stays on the host is only the ORDER of the matched reads: the reference sorts with std::sort / __gnu_parallel::sort
// under a comparator that looks at the position alone, so the order of reads matched at one position is an artefact of
// that algorithm and reaches the archive bytes; the adapter reproduces it with the same algorithm on (position, index)
// pairs and hands the permutation in.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <initializer_list>
#include <thread>
#include <utility>
#include <vector>

#include "ctx.h"
#include "devutil.h"

#define EX_NONE 0xFFFFFFFFu

// ---------------------------------------------------------------- exclusive scan: u8 or u32 values -> u64
#define SC_TPB 256
#define SC_EPT 16
#define SC_EPB (SC_TPB * SC_EPT)

template <typename T>
__device__ __forceinline__ uint64_t sc_block_scan(uint64_t v, uint64_t *smem, uint64_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63) smem[wv] = inc;
    __syncthreads();
    uint64_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < SC_TPB / 64; k++) {
        const uint64_t s = smem[k];
        if (k < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

template <typename T>
__global__ void __launch_bounds__(SC_TPB) k_scan_sums(const T *__restrict__ in, uint64_t n, uint64_t *bsum) {
    __shared__ uint64_t smem[SC_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SC_EPB + (uint64_t)threadIdx.x * SC_EPT;
    uint64_t s = 0;
    for (int k = 0; k < SC_EPT; k++)
        if (base + k < n) s += in[base + k];
    uint64_t tot;
    sc_block_scan<T>(s, smem, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// block sums -> exclusive block offsets (one block: nb = n / 4096 values, a few passes of 256)
__global__ void __launch_bounds__(SC_TPB) k_scan_bsums(uint64_t *bsum, uint64_t nb) {
    __shared__ uint64_t smem[SC_TPB / 64 + 1];
    uint64_t run = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += SC_TPB) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? bsum[i] : 0;
        uint64_t tot;
        const uint64_t ex = sc_block_scan<uint64_t>(v, smem, &tot);
        if (i < nb) bsum[i] = run + ex;
        run += tot;
    }
    if (threadIdx.x == 0) bsum[nb] = run;
}

// out[i] = (INCLUSIVE ? in[0..i] : in[0..i-1]) summed; out has n (+1 for the exclusive form: out[n] = total) entries
template <typename T, bool INCLUSIVE>
__global__ void __launch_bounds__(SC_TPB)
k_scan_write(const T *__restrict__ in, uint64_t n, const uint64_t *__restrict__ bsum, uint64_t nb, uint64_t *out) {
    __shared__ uint64_t smem[SC_TPB / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SC_EPB + (uint64_t)threadIdx.x * SC_EPT;
    uint64_t v[SC_EPT], s = 0;
#pragma unroll
    for (int k = 0; k < SC_EPT; k++) {
        v[k] = (base + k < n) ? (uint64_t)in[base + k] : 0;
        s += v[k];
    }
    uint64_t tot;
    uint64_t off = sc_block_scan<T>(s, smem, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_EPT; k++) {
        if (base + k < n) out[base + k] = INCLUSIVE ? off + v[k] : off;
        off += v[k];
    }
    if (!INCLUSIVE && blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}

template <typename T, bool INCLUSIVE>
static int device_scan(pgrc_match_ctx *c, const T *d_in, uint64_t n, uint64_t *d_out, DevBuf &bs) {
    const uint64_t nb = (n + SC_EPB - 1) / SC_EPB;
    int e;
    if ((e = pgrc_buf_ensure(c, bs, (nb + 2) * sizeof(uint64_t)))) return e;
    if (!n) {
        if (!INCLUSIVE) HIP_TRY(c, hipMemsetAsync(d_out, 0, sizeof(uint64_t), c->stream));
        return PGRC_OK;
    }
    hipLaunchKernelGGL((k_scan_sums<T>), dim3((uint32_t)nb), dim3(SC_TPB), 0, c->stream, d_in, n, (uint64_t *)bs.p);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SC_TPB), 0, c->stream, (uint64_t *)bs.p, nb);
    hipLaunchKernelGGL((k_scan_write<T, INCLUSIVE>), dim3((uint32_t)nb), dim3(SC_TPB), 0, c->stream, d_in, n, (const uint64_t *)bs.p, nb, d_out);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// ---------------------------------------------------------------- merge of the two sorted lists

struct ExportArgs {
    // results and inputs of the matcher
    const uint64_t *pos;
    const uint8_t *rc, *mism;
    // the matched reads in export order, their original indexes
    const uint32_t *order;
    uint64_t m;
    const uint32_t *read_org;     // per read, nullptr = identity
    // the reads list already on the pseudogenome
    const uint64_t *lpos;         // positions (inclusive scan of the offset deltas)
    const uint32_t *lorg;
    const uint8_t *lrc;           // nullptr = all forward
    uint64_t h;
    // merged entries
    uint64_t *epos;               // position; bit 63 marks a new entry written after the old list ran out
    uint32_t *eread;              // the matched read an entry describes, EX_NONE for an old entry / a filler
    uint32_t *eorg;
    uint8_t *erc, *emc;
    uint32_t *errflag;            // set when `order` names a read without a match
};

// first index in a[0, n) with a[i] >= x (UPPER: a[i] > x)
template <bool UPPER>
__device__ __forceinline__ uint64_t bound(const uint64_t *__restrict__ a, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t v = a[mid];
        if (UPPER ? v <= x : v < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// new entry j: every old entry with a smaller position precedes it (writeReadsFromIterator stops at pos >= stopPos,
// SeparatedPseudoGenomePersistence.cpp:1004-1019)
__global__ void __launch_bounds__(256) k_export_place_new(const ExportArgs a) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = a.order[j];
        if (a.mism[i] == PGRC_NOT_MATCHED_CNT) atomicOr(a.errflag, 1u);     // (its position is all ones: nothing below dereferences it)
        const uint64_t p = a.pos[i];
        const uint64_t before = bound<false>(a.lpos, a.h, p);
        const uint64_t r = j + before;
        // once the old list is exhausted writeReadsFromIterator returns -1 instead of the last written position (:1018)
        a.epos[r] = p | (before == a.h ? (1ull << 63) : 0ull);
        a.eread[r] = i;
        a.eorg[r] = a.read_org ? a.read_org[i] : i;
        a.erc[r] = a.rc[i];
        a.emc[r] = a.mism[i];
    }
}

// old entry k: the new entries at positions <= its own precede it.  pm(j) = pos[order[j]] is sorted ascending.
__global__ void __launch_bounds__(256) k_export_place_old(const ExportArgs a) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.h; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = a.lpos[k];
        uint64_t lo = 0, hi = a.m;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (a.pos[a.order[mid]] <= p) lo = mid + 1;
            else hi = mid;
        }
        const uint64_t r = k + lo;
        a.epos[r] = p;
        a.eread[r] = EX_NONE;
        a.eorg[r] = a.lorg[k];
        a.erc[r] = a.lrc ? a.lrc[k] : 0;
        a.emc[r] = 0;
    }
}

// offset of every entry = its position minus the position of the entry written before it, in the 16-bit arithmetic of
// ReadsListEntry::offset (ExtendedReadsListIteratorInterface.h:31-37), stored as 1 or 2 bytes (writeReadLengthValue,
// utils/helper.cpp:198-203)
template <typename OFF>
__global__ void __launch_bounds__(256) k_export_offsets(const uint64_t *__restrict__ epos, uint64_t ne, OFF *__restrict__ off) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ne; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = epos[r];
        const uint64_t p = e & ~(1ull << 63);
        const uint64_t prev = (e >> 63) ? ~0ull : (r ? (epos[r - 1] & ~(1ull << 63)) : 0ull);
        off[r] = (OFF)(uint16_t)(p - prev);
    }
}

// entries given by the caller (original-order export): field arrays from the per-read results
__global__ void __launch_bounds__(256)
k_export_fill_entries(const uint32_t *__restrict__ eread, uint64_t ne, const uint64_t *__restrict__ pos, const uint8_t *__restrict__ rc,
                      const uint8_t *__restrict__ mism, uint64_t *__restrict__ epos, uint8_t *__restrict__ erc, uint8_t *__restrict__ emc) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ne; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = eread[r];
        const bool real = i != EX_NONE && mism[i] != PGRC_NOT_MATCHED_CNT;
        // DefaultReadsListEntry entry(0); entry.advanceEntryByPosition(pos, ...): offset = pos - 0 (ReadsMatchers.cpp:655-667)
        epos[r] = real ? pos[i] : 0ull;
        erc[r] = real ? rc[i] : 0;
        emc[r] = real ? mism[i] : 0;
    }
}

// ---- original-order export: the entry list made on the device ----
// exportMatchesInOriginalOrder writes one entry per ORIGINAL read index that is not an unmatched read of this matcher:
// a matched read's entry, or a filler (position 0) for an index the matcher does not hold -- all even indexes first,
// then all odd ones, when the two files of a pair are interleaved (pairFileMode).  Here that is a table over the
// original indexes: every read of the matcher claims its index (k_oo_claim), every slot of the class-major order looks
// its index up (k_oo_keep), an exclusive scan of the kept slots numbers the entries (k_oo_entries).
#define EX_SKIP 0xFFFFFFFEu        // an original index held by an UNMATCHED read: no entry at all

__global__ void __launch_bounds__(256)
k_oo_claim(const uint32_t *__restrict__ read_org, uint64_t n, const uint8_t *__restrict__ mism, uint64_t total,
           uint32_t *__restrict__ owner, uint32_t *__restrict__ errflag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = read_org[i];
        if (o >= total) { atomicOr(errflag, 1u); continue; }
        const uint32_t was = atomicExch(&owner[o], mism[i] != PGRC_NOT_MATCHED_CNT ? (uint32_t)i : EX_SKIP);
        if (was != EX_NONE) atomicOr(errflag, 2u);          // two reads with one original index
    }
}

// slot q of the output order -> original index: class c = indexes congruent c modulo `parts`, classes in turn
__device__ __forceinline__ uint64_t oo_slot_org(uint64_t q, uint64_t total, uint32_t parts) {
    if (parts == 1) return q;
    const uint64_t evens = (total + 1) >> 1;
    return q < evens ? 2 * q : 2 * (q - evens) + 1;
}

__global__ void __launch_bounds__(256)
k_oo_keep(const uint32_t *__restrict__ owner, uint64_t total, uint32_t parts, uint8_t *__restrict__ keep) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x)
        keep[q] = owner[oo_slot_org(q, total, parts)] != EX_SKIP;
}

__global__ void __launch_bounds__(256)
k_oo_entries(const uint32_t *__restrict__ owner, const uint8_t *__restrict__ keep, const uint64_t *__restrict__ rank, uint64_t total,
             uint32_t parts, uint32_t *__restrict__ eread, uint32_t *__restrict__ eorg) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
        if (!keep[q]) continue;
        const uint64_t o = oo_slot_org(q, total, parts);
        eread[rank[q]] = owner[o];                            // EX_NONE = filler
        eorg[rank[q]] = (uint32_t)o;
    }
}

template <typename OFF>
__global__ void __launch_bounds__(256) k_export_offsets_abs(const uint64_t *__restrict__ epos, uint64_t ne, OFF *__restrict__ off) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ne; r += (uint64_t)gridDim.x * blockDim.x)
        off[r] = (OFF)(uint16_t)epos[r];
}

// ---------------------------------------------------------------- mismatch streams

struct MisArgs {
    const uint32_t *pg;
    const uint32_t *reads;
    uint64_t stride;
    const uint8_t *nflag;
    const uint32_t *nidx;
    const uint8_t *nascii;
    uint64_t nn;
    const uint64_t *pos;
    const uint8_t *rc;
    const uint32_t *eread, *eorg;
    const uint8_t *emc;
    const uint64_t *mbase;        // exclusive scan of emc
    uint64_t ne;
    uint8_t *sym;
    void *revoff;
    uint32_t L, pair_file;
};

__device__ __forceinline__ uint32_t ex_compl(uint32_t v) { return v < 4u ? 3u - v : 4u; } // N <-> N

// One thread per entry.  updateEntry (ReadsMatchers.cpp:548-559): the read is reverse-complemented if it matched the RC
// strand; the mismatch list is taken in the ORIGINAL read's orientation ("reversed": scan from the end, complemented
// symbols, fillEntryWithReversedMismatches :53-66) iff rc (SE) resp. rc != (orgIdx odd) (revComplPairFile), else in the
// pseudogenome's (fillEntryWithMismatches :40-51).  writeReadEntry then emits the codes in list order and the offsets
// coded backwards: L-1 - off[last], off[last]-1 - off[last-1], ... (SeparatedPseudoGenomePersistence.cpp:975-981).
template <typename OFF>
__global__ void __launch_bounds__(256) k_export_mismatches(const MisArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.ne) return;
    const uint32_t cnt = a.emc[r];
    if (!cnt) return;
    const uint32_t i = a.eread[r];
    const uint32_t L = a.L;
    // reads with N keep their symbols as ASCII rows in a side list ordered by read index
    const uint8_t *row = nullptr;
    if (a.nflag && a.nflag[i]) {
        uint64_t lo = 0, hi = a.nn;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (a.nidx[mid] < i) lo = mid + 1;
            else hi = mid;
        }
        row = a.nascii + lo * L;
    }
    auto rval = [&](uint32_t x) -> uint32_t {
        if (row) {
            const uint8_t ch = row[x];
            return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
        }
        return (a.reads[(uint64_t)(x >> 4) * a.stride + i] >> (2u * (x & 15u))) & 3u;
    };
    const uint64_t p = a.pos[i];
    const bool rc = a.rc[i] != 0;
    const bool reversed = a.pair_file ? (rc != (bool)(a.eorg[r] & 1u)) : rc;
    const uint64_t o = a.mbase[r];
    OFF *ro = (OFF *)a.revoff;
    uint32_t emitted = 0, prev_off = 0;
    for (uint32_t step = 0; step < L && emitted < cnt; step++) {
        const uint32_t x = reversed ? L - 1u - step : step;     // index into the (possibly RC'd) read / Pg window
        uint32_t rv = rc ? ex_compl(rval(L - 1u - x)) : rval(x);
        const uint64_t g = p + x;
        uint32_t pv = (a.pg[g >> 4] >> (2u * ((uint32_t)g & 15u))) & 3u;
        if (rv != pv) {
            if (reversed) { pv = ex_compl(pv); rv = ex_compl(rv); }
            a.sym[o + emitted] = (uint8_t)((pv << 4) + rv);
            // list offset of this mismatch = step; the value for the PREVIOUS one is known now
            if (emitted) ro[o + cnt - emitted] = (OFF)(step - 1u - prev_off);
            prev_off = step;
            emitted++;
        }
    }
    if (emitted) ro[o] = (OFF)(L - 1u - prev_off);
}

// ---- the same from per-read mismatch LISTS (round 4: a matcher over several devices).  Every shard makes the lists of its own
// reads on its own device (results.hip, k_extract: codes + list offsets in the orientation updateEntry chooses) and only those
// travel to the exporting device -- 8 bytes per read and 3 per mismatch instead of the packed reads and the N side lists.  An
// entry then copies its read's codes and codes the offsets backwards, as above.
__global__ void __launch_bounds__(256)
k_ex_parity(const uint32_t *__restrict__ eread, const uint32_t *__restrict__ eorg, uint64_t ne, uint8_t *__restrict__ par) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ne; r += (uint64_t)gridDim.x * blockDim.x)
        if (eread[r] != EX_NONE) par[eread[r]] = (uint8_t)(eorg[r] & 1u);
}
__global__ void __launch_bounds__(256)
k_ex_revflags(const uint8_t *__restrict__ rc, const uint8_t *__restrict__ par, uint64_t n, uint32_t pair_file, uint8_t *__restrict__ rev) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        rev[i] = (uint8_t)(pair_file ? ((rc[i] != 0) != (par[i] != 0)) : (rc[i] != 0));     // ReadsMatchers.cpp:553
}
__global__ void __launch_bounds__(256) k_ex_rebase(uint64_t *__restrict__ cum, uint64_t count, uint64_t base) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) cum[i] += base;
}
template <typename OFF>
__global__ void __launch_bounds__(256)
k_export_mismatches_lists(const uint32_t *__restrict__ eread, const uint8_t *__restrict__ emc, const uint64_t *__restrict__ mbase, uint64_t ne,
                          const uint64_t *__restrict__ cum, const uint8_t *__restrict__ codes, const uint16_t *__restrict__ offs, uint32_t L,
                          uint8_t *__restrict__ sym, OFF *__restrict__ ro) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ne) return;
    const uint32_t cnt = emc[r];
    if (!cnt) return;
    const uint64_t src = cum[eread[r]], o = mbase[r];
    uint32_t prev = 0;
    for (uint32_t k = 0; k < cnt; k++) {
        sym[o + k] = codes[src + k];
        const uint32_t step = offs[src + k];
        if (k) ro[o + cnt - k] = (OFF)(step - 1u - prev);
        prev = step;
    }
    ro[o] = (OFF)(L - 1u - prev);
}

// ---------------------------------------------------------------- host side

static uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 65536ull * 4); }

extern "C" void pgrc_match_free_export(pgrc_export_streams *s) {
    if (!s) return;
    free(s->off); free(s->org_idx); free(s->rev_comp); free(s->mis_cnt); free(s->mis_sym); free(s->mis_rev_off);
    memset(s, 0, sizeof *s);
}

namespace {
struct Bufs {
    DevBuf order, rorg, loff, lorg, lrc, lpos, epos, eread, eorg, erc, emc, off, mbase, sym, roff, bs, flag;
    void release() {
        DevBuf *all[] = {&order, &rorg, &loff, &lorg, &lrc, &lpos, &epos, &eread, &eorg, &erc, &emc, &off, &mbase, &sym, &roff, &bs, &flag};
        pgrc_buf_free_all(all, sizeof all / sizeof all[0]);         // (one wait for the device, not one per pooled buffer)
    }
};
}

// view = the gathered results of front f on its first device; b.eread / b.eorg / b.emc / b.mbase are there.  -> b.sym, b.roff
static int mismatch_streams_from_shards(pgrc_match_ctx *view, Bufs &b, uint64_t ne, int pair_file, uint32_t width, uint64_t total) {
    pgrc_match_ctx *f = view->export_front;
    const std::vector<PgrcShardView> sh = pgrc_multi_shards(f);
    const uint64_t n = view->n;
    DevBuf par, rev, cumAll, codesAll, offsAll;
    auto done = [&](int e) { DevBuf *all[] = {&par, &rev, &cumAll, &codesAll, &offsAll}; pgrc_buf_free_all(all, 5); return e; };
    int e;
    if ((e = pgrc_buf_ensure(view, par, n ? n : 1)) || (e = pgrc_buf_ensure(view, rev, n ? n : 1)) || (e = pgrc_buf_ensure(view, cumAll, (n + 1) * sizeof(uint64_t))) ||
        (e = pgrc_buf_ensure(view, codesAll, total)) || (e = pgrc_buf_ensure(view, offsAll, total * sizeof(uint16_t))))
        return done(e);
    // the orientation of every read's list: rc, or rc != (its original index is odd) -- the parity comes from the entries
    hipError_t he = hipMemsetAsync(par.p, 0, n ? n : 1, view->stream);
    if (he == hipSuccess && ne) {
        hipLaunchKernelGGL(k_ex_parity, dim3(grid_for(ne)), dim3(256), 0, view->stream, (const uint32_t *)b.eread.p, (const uint32_t *)b.eorg.p, ne, (uint8_t *)par.p);
        if (n) hipLaunchKernelGGL(k_ex_revflags, dim3(grid_for(n)), dim3(256), 0, view->stream, (const uint8_t *)view->d_rc.p, (const uint8_t *)par.p, n,
                                  pair_file ? 1u : 0u, (uint8_t *)rev.p);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(view->stream);
    if (he != hipSuccess) { view->err = std::string("export: orientation flags: ") + hipGetErrorString(he); return done(pgrc_hip_code(he)); }
    uint64_t base = 0;
    for (const PgrcShardView &s : sh) {
        const uint64_t cnt = s.hi - s.lo;
        if (!cnt) continue;
        pgrc_match_ctx *c = s.ctx;
        DevBuf d_rev, d_cum, d_codes, d_offs;
        uint64_t tot_s = 0;
        int se;
        {
            PgrcDeviceScope cs(c->device);
            if (!cs.ok) { view->err = "hipSetDevice failed"; return done(PGRC_E_NO_DEVICE); }
            se = pgrc_buf_ensure(c, d_rev, cnt);
            if (!se && hipMemcpyAsync(d_rev.p, (const uint8_t *)rev.p + s.lo, cnt, hipMemcpyDefault, c->stream) != hipSuccess) se = PGRC_E_DEVICE;
            if (!se) se = pgrc_extract_lists_device(c, (const uint8_t *)d_rev.p, true, d_cum, d_codes, d_offs, &tot_s);
            if (!se && hipStreamSynchronize(c->stream) != hipSuccess) se = PGRC_E_DEVICE;
            if (se) view->err = c->err.empty() ? "export: a shard's mismatch lists" : c->err;
        }
        if (!se) {
            // the shard's lists into the whole set's: offsets of its reads behind those of the shards before it
            he = hipMemcpyAsync((uint64_t *)cumAll.p + s.lo, d_cum.p, cnt * sizeof(uint64_t), hipMemcpyDefault, view->stream);
            if (he == hipSuccess && base) hipLaunchKernelGGL(k_ex_rebase, dim3(grid_for(cnt)), dim3(256), 0, view->stream, (uint64_t *)cumAll.p + s.lo, cnt, base);
            if (he == hipSuccess && tot_s) he = hipMemcpyAsync((uint8_t *)codesAll.p + base, d_codes.p, tot_s, hipMemcpyDefault, view->stream);
            if (he == hipSuccess && tot_s) he = hipMemcpyAsync((uint16_t *)offsAll.p + base, d_offs.p, tot_s * sizeof(uint16_t), hipMemcpyDefault, view->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(view->stream);
            if (he != hipSuccess) { se = pgrc_hip_code(he); view->err = std::string("export: gathering a shard's mismatch lists: ") + hipGetErrorString(he); }
        }
        {
            PgrcDeviceScope cs(c->device);
            DevBuf *all[] = {&d_rev, &d_cum, &d_codes, &d_offs};
            pgrc_buf_free_all(all, 4);
        }
        if (se) return done(se);
        base += tot_s;
    }
    if (base != total) { view->err = "export: the shards' mismatch lists do not add up to the entries' counts"; return done(PGRC_E_STATE); }
    const uint32_t grid = (uint32_t)((ne + 255) / 256);
    if (width == 1)
        hipLaunchKernelGGL(k_export_mismatches_lists<uint8_t>, dim3(grid), dim3(256), 0, view->stream, (const uint32_t *)b.eread.p, (const uint8_t *)b.emc.p,
                           (const uint64_t *)b.mbase.p, ne, (const uint64_t *)cumAll.p, (const uint8_t *)codesAll.p, (const uint16_t *)offsAll.p, view->prm.read_len,
                           (uint8_t *)b.sym.p, (uint8_t *)b.roff.p);
    else
        hipLaunchKernelGGL(k_export_mismatches_lists<uint16_t>, dim3(grid), dim3(256), 0, view->stream, (const uint32_t *)b.eread.p, (const uint8_t *)b.emc.p,
                           (const uint64_t *)b.mbase.p, ne, (const uint64_t *)cumAll.p, (const uint8_t *)codesAll.p, (const uint16_t *)offsAll.p, view->prm.read_len,
                           (uint8_t *)b.sym.p, (uint16_t *)b.roff.p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(view->stream) != hipSuccess) { view->err = "export: mismatch streams from the lists"; return done(PGRC_E_DEVICE); }
    return done(PGRC_OK);
}

// emc / eread / eorg (device) are filled: scan the counts, extract the mismatch streams, bring everything to the host
// First touch of fresh host allocations by several threads: a device-to-host copy into untouched pages runs at the
// page-fault rate of ONE thread (13 GB/s against the link's 56, profiles/r03_ubench_pcie.txt).
static void touch_pages(std::initializer_list<std::pair<void *, size_t>> bufs) {
    size_t total = 0;
    for (const auto &b : bufs) total += b.second;
    if (total < (64u << 20)) return;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = std::max(1u, std::min(8u, hw ? hw : 1u));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (const auto &b : bufs) {
                volatile uint8_t *p = (volatile uint8_t *)b.first;
                const size_t lo = b.second * t / T, hi = b.second * (t + 1) / T;
                for (size_t x = (lo + 4095) & ~(size_t)4095; x < hi; x += 4096) p[x] = 0;
                if (t == 0 && b.second) p[0] = 0;
            }
        });
    for (auto &x : th) x.join();
}

static int finish_export(pgrc_match_ctx *c, Bufs &b, uint64_t ne, int pair_file, uint32_t width, pgrc_export_streams *out) {
    int e;
    if ((e = pgrc_buf_ensure(c, b.mbase, (ne + 1) * sizeof(uint64_t)))) return e;
    if ((e = device_scan<uint8_t, false>(c, (const uint8_t *)b.emc.p, ne, (uint64_t *)b.mbase.p, b.bs))) return e;
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, (uint64_t *)b.mbase.p + ne, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (total && c->export_front) {                       // a matcher over several devices: the lists come from the shards
        if ((e = pgrc_buf_ensure(c, b.sym, total)) || (e = pgrc_buf_ensure(c, b.roff, total * width))) return e;
        if ((e = mismatch_streams_from_shards(c, b, ne, pair_file, width, total))) return e;
    } else if (total) {
        if ((e = pgrc_buf_ensure(c, b.sym, total)) || (e = pgrc_buf_ensure(c, b.roff, total * width))) return e;
        MisArgs a;
        a.pg = (const uint32_t *)c->pg2[0].p;
        a.reads = c->reads2;
        a.stride = c->stride;
        a.nflag = c->n_nreads ? (const uint8_t *)c->nread_flag.p : nullptr;
        a.nidx = (const uint32_t *)c->nread_idx.p;
        a.nascii = (const uint8_t *)c->nread_ascii.p;
        a.nn = c->n_nreads;
        a.pos = (const uint64_t *)c->d_pos.p;
        a.rc = (const uint8_t *)c->d_rc.p;
        a.eread = (const uint32_t *)b.eread.p;
        a.eorg = (const uint32_t *)b.eorg.p;
        a.emc = (const uint8_t *)b.emc.p;
        a.mbase = (const uint64_t *)b.mbase.p;
        a.ne = ne;
        a.sym = (uint8_t *)b.sym.p;
        a.revoff = b.roff.p;
        a.L = c->prm.read_len;
        a.pair_file = pair_file ? 1u : 0u;
        const uint32_t grid = (uint32_t)((ne + 255) / 256);
        if (width == 1) hipLaunchKernelGGL(k_export_mismatches<uint8_t>, dim3(grid), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL(k_export_mismatches<uint16_t>, dim3(grid), dim3(256), 0, c->stream, a);
        HIP_TRY(c, hipGetLastError());
    }
    out->n_entries = ne;
    out->n_mismatches = total;
    out->off_width = width;
    out->off = (uint8_t *)malloc(std::max<uint64_t>(ne * width, 1));
    out->org_idx = (uint32_t *)malloc(std::max<uint64_t>(ne * sizeof(uint32_t), 1));
    out->rev_comp = (uint8_t *)malloc(std::max<uint64_t>(ne, 1));
    out->mis_cnt = (uint8_t *)malloc(std::max<uint64_t>(ne, 1));
    out->mis_sym = (uint8_t *)malloc(std::max<uint64_t>(total, 1));
    out->mis_rev_off = (uint8_t *)malloc(std::max<uint64_t>(total * width, 1));
    if (!out->off || !out->org_idx || !out->rev_comp || !out->mis_cnt || !out->mis_sym || !out->mis_rev_off) {
        c->err = "export: host allocation failed";
        return PGRC_E_ALLOC;
    }
    touch_pages({{out->off, ne * width}, {out->org_idx, ne * sizeof(uint32_t)}, {out->rev_comp, ne}, {out->mis_cnt, ne}, {out->mis_sym, total},
                 {out->mis_rev_off, total * width}});            // (while the mismatch kernel runs)
    if (ne) {
        HIP_TRY(c, hipMemcpyAsync(out->off, b.off.p, ne * width, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out->org_idx, b.eorg.p, ne * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out->rev_comp, b.erc.p, ne, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out->mis_cnt, b.emc.p, ne, hipMemcpyDeviceToHost, c->stream));
    }
    if (total) {
        HIP_TRY(c, hipMemcpyAsync(out->mis_sym, b.sym.p, total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out->mis_rev_off, b.roff.p, total * width, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PGRC_OK;
}

static int upload(pgrc_match_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    int e = pgrc_buf_ensure(c, b, bytes);
    if (e) return e;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return PGRC_OK;
}

// ---- the order of the matched reads made on the device (round 4): ascending match position, reads matched at one position
// in ascending read index (the reference's order is its sort algorithm's -- std::sort / __gnu_parallel::sort leave equal
// positions in an order of their own, ReadsMatchers.cpp:573-574 -- and only reproducible at -t 1; a caller that needs those
// bytes passes its own order[]).  Records position << 32 | read of the matched reads, compacted in read order, then a stable
// radix sort over the position bits (radix.hip): 4 passes for a text below 2^32.
__global__ void __launch_bounds__(256) k_ord_flags(const uint64_t *__restrict__ pos, uint64_t n, uint8_t *__restrict__ flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        flag[i] = pos[i] != PGRC_NOT_MATCHED_POS ? 1 : 0;
}
__global__ void __launch_bounds__(256)
k_ord_records(const uint64_t *__restrict__ pos, const uint64_t *__restrict__ slot, uint64_t n, uint64_t *__restrict__ rec) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (pos[i] != PGRC_NOT_MATCHED_POS) rec[slot[i]] = (pos[i] << 32) | i;
}
__global__ void __launch_bounds__(256) k_ord_reads(const uint64_t *__restrict__ rec, uint64_t m, uint32_t *__restrict__ order) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (uint64_t)gridDim.x * blockDim.x)
        order[k] = (uint32_t)rec[k];
}

int pgrc_radix_sort_u64(pgrc_match_ctx *c, uint64_t *d_a, uint64_t *d_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi, DevBuf &scratch,
                        uint64_t **sorted);

// -> b.order (device), *m_out = matched reads
static int device_position_order(pgrc_match_ctx *c, Bufs &b, uint64_t *m_out) {
    const uint64_t n = c->n;
    if (c->G >= (1ull << 32)) { c->err = "export_pg_order: the device-made order needs a text below 2^32 symbols (pass order[])"; return PGRC_E_PARAM; }
    DevBuf flag, slot, ra, rb, scratch;
    auto done = [&](int e) { DevBuf *all[] = {&flag, &slot, &ra, &rb, &scratch}; pgrc_buf_free_all(all, 5); return e; };
    int e;
    if ((e = pgrc_buf_ensure(c, flag, n)) || (e = pgrc_buf_ensure(c, slot, (n + 1) * sizeof(uint64_t)))) return done(e);
    if (n) hipLaunchKernelGGL(k_ord_flags, dim3(grid_for(n)), dim3(256), 0, c->stream, (const uint64_t *)c->d_pos.p, n, (uint8_t *)flag.p);
    if ((e = device_scan<uint8_t, false>(c, (const uint8_t *)flag.p, n, (uint64_t *)slot.p, b.bs))) return done(e);
    uint64_t m = 0;
    if (hipMemcpyAsync(&m, (const uint64_t *)slot.p + n, sizeof m, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "export_pg_order: HIP error"; return done(PGRC_E_DEVICE); }
    *m_out = m;
    if ((e = pgrc_buf_ensure(c, b.order, m * sizeof(uint32_t)))) return done(e);
    if (!m) return done(PGRC_OK);
    if ((e = pgrc_buf_ensure(c, ra, m * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, rb, m * sizeof(uint64_t)))) return done(e);
    hipLaunchKernelGGL(k_ord_records, dim3(grid_for(n)), dim3(256), 0, c->stream, (const uint64_t *)c->d_pos.p, (const uint64_t *)slot.p, n, (uint64_t *)ra.p);
    uint32_t pbits = 1;
    while (pbits < 32 && (c->G >> pbits)) pbits++;
    uint64_t *sorted = nullptr;
    if ((e = pgrc_radix_sort_u64(c, (uint64_t *)ra.p, (uint64_t *)rb.p, m, 32u, 32u + pbits, scratch, &sorted))) return done(e);
    hipLaunchKernelGGL(k_ord_reads, dim3(grid_for(m)), dim3(256), 0, c->stream, (const uint64_t *)sorted, m, (uint32_t *)b.order.p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "export_pg_order: HIP error"; return done(PGRC_E_DEVICE); }
    return done(PGRC_OK);
}

static int export_pg_order(pgrc_match_ctx *c, const pgrc_export_pg_order_args *x, Bufs &b, pgrc_export_streams *out) {
    uint64_t m = x->n_matched;
    const uint64_t h = x->list_count;
    const uint32_t width = x->byte_per_read_length ? 1u : 2u;
    int e;
    if (x->order_on_device) {
        if ((e = device_position_order(c, b, &m))) return e;
    } else if ((e = upload(c, b.order, x->order, m * sizeof(uint32_t)))) return e;    // (order == NULL: m == 0, checked by the caller)
    const uint64_t ne = m + h;
    if (x->read_org_idx && (e = upload(c, b.rorg, x->read_org_idx, c->n * sizeof(uint32_t)))) return e;
    if ((e = upload(c, b.loff, x->list_off, h)) || (e = upload(c, b.lorg, x->list_org_idx, h * sizeof(uint32_t)))) return e;
    if (x->list_rev_comp && (e = upload(c, b.lrc, x->list_rev_comp, h))) return e;
    if ((e = pgrc_buf_ensure(c, b.lpos, (h + 1) * sizeof(uint64_t)))) return e;
    if ((e = device_scan<uint8_t, true>(c, (const uint8_t *)b.loff.p, h, (uint64_t *)b.lpos.p, b.bs))) return e;
    if ((e = pgrc_buf_ensure(c, b.epos, ne * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, b.eread, ne * sizeof(uint32_t))) ||
        (e = pgrc_buf_ensure(c, b.eorg, ne * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, b.erc, ne)) ||
        (e = pgrc_buf_ensure(c, b.emc, ne)) || (e = pgrc_buf_ensure(c, b.off, ne * width)))
        return e;
    ExportArgs a;
    a.pos = (const uint64_t *)c->d_pos.p;
    a.rc = (const uint8_t *)c->d_rc.p;
    a.mism = (const uint8_t *)c->d_mism.p;
    a.order = (const uint32_t *)b.order.p;
    a.m = m;
    a.read_org = x->read_org_idx ? (const uint32_t *)b.rorg.p : nullptr;
    a.lpos = (const uint64_t *)b.lpos.p;
    a.lorg = (const uint32_t *)b.lorg.p;
    a.lrc = x->list_rev_comp ? (const uint8_t *)b.lrc.p : nullptr;
    a.h = h;
    a.epos = (uint64_t *)b.epos.p;
    a.eread = (uint32_t *)b.eread.p;
    a.eorg = (uint32_t *)b.eorg.p;
    a.erc = (uint8_t *)b.erc.p;
    a.emc = (uint8_t *)b.emc.p;
    if ((e = pgrc_buf_ensure(c, b.flag, sizeof(uint32_t)))) return e;
    HIP_TRY(c, hipMemsetAsync(b.flag.p, 0, sizeof(uint32_t), c->stream));
    a.errflag = (uint32_t *)b.flag.p;
    if (m) hipLaunchKernelGGL(k_export_place_new, dim3(grid_for(m)), dim3(256), 0, c->stream, a);
    if (h) hipLaunchKernelGGL(k_export_place_old, dim3(grid_for(h)), dim3(256), 0, c->stream, a);
    if (ne) {
        if (width == 1) hipLaunchKernelGGL(k_export_offsets<uint8_t>, dim3(grid_for(ne)), dim3(256), 0, c->stream, (const uint64_t *)b.epos.p, ne, (uint8_t *)b.off.p);
        else hipLaunchKernelGGL(k_export_offsets<uint16_t>, dim3(grid_for(ne)), dim3(256), 0, c->stream, (const uint64_t *)b.epos.p, ne, (uint16_t *)b.off.p);
    }
    HIP_TRY(c, hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(c, hipMemcpyAsync(&bad, b.flag.p, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) { c->err = "export_pg_order: order[] names a read without a match"; return PGRC_E_PARAM; }
    if ((e = finish_export(c, b, ne, x->rev_compl_pair_file, width, out))) return e;
    // the builder's lastWrittenPos: the position of the last entry written
    out->last_pos = 0;
    if (ne) {
        uint64_t last = 0;
        HIP_TRY(c, hipMemcpy(&last, (const uint64_t *)b.epos.p + ne - 1, sizeof last, hipMemcpyDeviceToHost));
        out->last_pos = last & ~(1ull << 63);
    }
    return PGRC_OK;
}

// A multi-device context exports from ONE device: the per-read results of every shard are gathered (peer copies: 10 bytes per
// read) into the layout of a single-device context on the first shard's device; the entry lists, offsets and counts are made
// there.  The packed reads, the N side lists and the text stay where they are: the mismatch streams come from per-read lists
// that every shard extracts on its own device (mismatch_streams_from_shards above; round 3 gathered the reads too: 3.5 GB of
// them + 0.9 GB of results for C3 over 8 devices, now 0.9 GB + ~8 bytes per read and 3 per mismatch of lists).
namespace {
struct GatheredView {
    pgrc_match_ctx view;
    DevBuf pos, rc, mism;
    ~GatheredView() {
        DevBuf *all[] = {&pos, &rc, &mism};
        pgrc_buf_free_all(all, 3);
    }
    int build(pgrc_match_ctx *f) {
        const std::vector<PgrcShardView> sh = pgrc_multi_shards(f);
        pgrc_match_ctx *c0 = sh[0].ctx;
        PgrcDeviceScope scope(c0->device);
        if (!scope.ok) { f->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
        const uint64_t n = f->n;
        int e;
        auto fail = [&](int code) { f->err = c0->err; return code; };
        if ((e = pgrc_buf_ensure(c0, pos, std::max<uint64_t>(n, 1) * 8)) || (e = pgrc_buf_ensure(c0, rc, std::max<uint64_t>(n, 1))) ||
            (e = pgrc_buf_ensure(c0, mism, std::max<uint64_t>(n, 1))))
            return fail(e);
        hipError_t he = hipSuccess;
        for (const PgrcShardView &s : sh) {
            const uint64_t cnt = s.hi - s.lo;
            if (!cnt || he != hipSuccess) continue;
            pgrc_match_ctx *c = s.ctx;
            {   // whatever the shard still has in flight
                PgrcDeviceScope cs(c->device);
                he = hipStreamSynchronize(c->stream);
            }
            if (he == hipSuccess) he = hipMemcpyAsync((uint64_t *)pos.p + s.lo, c->d_pos.p, cnt * 8, hipMemcpyDefault, c0->stream);
            if (he == hipSuccess) he = hipMemcpyAsync((uint8_t *)rc.p + s.lo, c->d_rc.p, cnt, hipMemcpyDefault, c0->stream);
            if (he == hipSuccess) he = hipMemcpyAsync((uint8_t *)mism.p + s.lo, c->d_mism.p, cnt, hipMemcpyDefault, c0->stream);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(c0->stream);
        if (he != hipSuccess) { f->err = std::string("export: gathering the shards: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
        view.prm = c0->prm;
        view.device = c0->device;
        view.stream = c0->stream;
        view.num_cus = c0->num_cus;
        view.nw = c0->nw;
        view.n = n;
        view.stride = 0;
        view.reads2 = nullptr;            // (the reads stay on their shards)
        view.n_nreads = 0;
        view.export_front = f;
        view.d_pos = pos;                 // (DevBuf copies: the view never frees anything, this object does)
        view.d_rc = rc;
        view.d_mism = mism;
        view.pg2[0] = c0->pg2[0];
        view.G = c0->G;
        view.pg_words = c0->pg_words;
        view.have_pg = view.have_reads = view.have_results = true;
        return PGRC_OK;
    }
};
}

// The gathered view of a multi-device context is kept for the next export call (the adapter asks for the mismatch lists and
// for one or two exports of the same results: gathering 3.5 + 0.9 GB at C3 size each time would be the larger part of them);
// whatever changes the text, the reads or the results drops it (pgrc_export_drop_view, called from multi.hip).
static int view_for(pgrc_match_ctx *c, pgrc_match_ctx **w) {
    *w = c;
    if (!c->multi) return PGRC_OK;
    GatheredView *gv = (GatheredView *)c->export_view;
    if (!gv) {
        gv = new GatheredView();
        const int e = gv->build(c);
        if (e) { delete gv; return e; }
        c->export_view = gv;
    }
    *w = &gv->view;
    return PGRC_OK;
}

void pgrc_export_drop_view(pgrc_match_ctx *c) {
    if (!c || !c->export_view) return;
    GatheredView *gv = (GatheredView *)c->export_view;
    c->export_view = nullptr;
    if (!c->multi) { delete gv; return; }
    PgrcDeviceScope scope(gv->view.device);
    delete gv;
}

extern "C" int pgrc_match_export_pg_order(pgrc_match_ctx *c, const pgrc_export_pg_order_args *x, pgrc_export_streams *out) {
    if (!c || !x || !out || (x->list_count && (!x->list_off || !x->list_org_idx))) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    if (!c->have_results || !c->have_pg || !c->have_reads) { c->err = "export: run first"; return PGRC_E_STATE; }
    if (!x->order_on_device) {
        if (x->n_matched > c->n) { c->err = "export: more matched reads than reads"; return PGRC_E_PARAM; }
        if (!x->order && x->n_matched) { c->err = "export_pg_order: order == NULL with n_matched > 0 (set order_on_device to let the library make the order)"; return PGRC_E_PARAM; }
    }
    for (uint64_t j = 0; !x->order_on_device && x->order && j < x->n_matched; j++)
        if (x->order[j] >= c->n) { c->err = "export_pg_order: read index out of range"; return PGRC_E_PARAM; }
    pgrc_match_ctx *w = c;
    int ge = view_for(c, &w);
    if (ge) return ge;
    PgrcDeviceScope scope(w->device);
    Bufs b;
    int e = export_pg_order(w, x, b, out);
    b.release();
    if (e && w != c) c->err = w->err;
    if (e) pgrc_match_free_export(out);
    return e;
}

// b.eread / b.eorg hold the entry list on the device: field arrays, offsets, mismatch streams
static int export_entries_device(pgrc_match_ctx *c, uint64_t ne, int pair_file, uint32_t width, Bufs &b, pgrc_export_streams *out) {
    int e;
    if ((e = pgrc_buf_ensure(c, b.epos, ne * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, b.erc, ne)) || (e = pgrc_buf_ensure(c, b.emc, ne)) ||
        (e = pgrc_buf_ensure(c, b.off, ne * width)))
        return e;
    if (ne) {
        hipLaunchKernelGGL(k_export_fill_entries, dim3(grid_for(ne)), dim3(256), 0, c->stream, (const uint32_t *)b.eread.p, ne,
                           (const uint64_t *)c->d_pos.p, (const uint8_t *)c->d_rc.p, (const uint8_t *)c->d_mism.p, (uint64_t *)b.epos.p,
                           (uint8_t *)b.erc.p, (uint8_t *)b.emc.p);
        if (width == 1) hipLaunchKernelGGL(k_export_offsets_abs<uint8_t>, dim3(grid_for(ne)), dim3(256), 0, c->stream, (const uint64_t *)b.epos.p, ne, (uint8_t *)b.off.p);
        else hipLaunchKernelGGL(k_export_offsets_abs<uint16_t>, dim3(grid_for(ne)), dim3(256), 0, c->stream, (const uint64_t *)b.epos.p, ne, (uint16_t *)b.off.p);
        HIP_TRY(c, hipGetLastError());
    }
    return finish_export(c, b, ne, pair_file, width, out);
}

static int export_entries(pgrc_match_ctx *c, const uint32_t *entry_read, const uint32_t *entry_org_idx, uint64_t ne, int pair_file,
                          uint32_t width, Bufs &b, pgrc_export_streams *out) {
    int e;
    if ((e = upload(c, b.eread, entry_read, ne * sizeof(uint32_t))) || (e = upload(c, b.eorg, entry_org_idx, ne * sizeof(uint32_t)))) return e;
    return export_entries_device(c, ne, pair_file, width, b, out);
}

// exportMatchesInOriginalOrder with the entry list made here (see k_oo_claim)
static int export_original_order(pgrc_match_ctx *c, const pgrc_export_original_order_args *x, Bufs &b, pgrc_export_streams *out) {
    const uint64_t total = x->reads_total_count;
    const uint32_t parts = x->pair_file_mode ? 2u : 1u, width = x->byte_per_read_length ? 1u : 2u;
    DevBuf owner, keep, rank, flag;
    auto done = [&](int code) { DevBuf *all[] = {&owner, &keep, &rank, &flag}; pgrc_buf_free_all(all, 4); return code; };
    int e;
    if ((e = upload(c, b.rorg, x->read_org_idx, c->n * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, owner, std::max<uint64_t>(total, 1) * sizeof(uint32_t))) ||
        (e = pgrc_buf_ensure(c, keep, std::max<uint64_t>(total, 1))) || (e = pgrc_buf_ensure(c, rank, (total + 1) * sizeof(uint64_t))) ||
        (e = pgrc_buf_ensure(c, flag, sizeof(uint32_t))))
        return done(e);
    hipError_t he = hipMemsetAsync(owner.p, 0xFF, std::max<uint64_t>(total, 1) * sizeof(uint32_t), c->stream);   // EX_NONE everywhere: fillers
    if (he == hipSuccess) he = hipMemsetAsync(flag.p, 0, sizeof(uint32_t), c->stream);
    if (he != hipSuccess) { c->err = std::string("export: ") + hipGetErrorString(he); return done(pgrc_hip_code(he)); }
    if (c->n)
        hipLaunchKernelGGL(k_oo_claim, dim3(grid_for(c->n)), dim3(256), 0, c->stream, (const uint32_t *)b.rorg.p, c->n, (const uint8_t *)c->d_mism.p,
                           total, (uint32_t *)owner.p, (uint32_t *)flag.p);
    if (total) hipLaunchKernelGGL(k_oo_keep, dim3(grid_for(total)), dim3(256), 0, c->stream, (const uint32_t *)owner.p, total, parts, (uint8_t *)keep.p);
    if ((e = device_scan<uint8_t, false>(c, (const uint8_t *)keep.p, total, (uint64_t *)rank.p, b.bs))) return done(e);
    uint64_t ne = 0;
    uint32_t bad = 0;
    he = hipMemcpyAsync(&ne, (const uint64_t *)rank.p + total, sizeof ne, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(&bad, flag.p, sizeof bad, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) { c->err = std::string("export: ") + hipGetErrorString(he); return done(pgrc_hip_code(he)); }
    if (bad) {
        c->err = (bad & 1u) ? "export_original_order: an original index is >= reads_total_count" : "export_original_order: two reads share one original index";
        return done(PGRC_E_PARAM);
    }
    if ((e = pgrc_buf_ensure(c, b.eread, std::max<uint64_t>(ne, 1) * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, b.eorg, std::max<uint64_t>(ne, 1) * sizeof(uint32_t))))
        return done(e);
    if (total) {
        hipLaunchKernelGGL(k_oo_entries, dim3(grid_for(total)), dim3(256), 0, c->stream, (const uint32_t *)owner.p, (const uint8_t *)keep.p,
                           (const uint64_t *)rank.p, total, parts, (uint32_t *)b.eread.p, (uint32_t *)b.eorg.p);
        he = hipGetLastError();
        if (he != hipSuccess) { c->err = std::string("export: ") + hipGetErrorString(he); return done(pgrc_hip_code(he)); }
    }
    return done(export_entries_device(c, ne, x->rev_compl_pair_file, width, b, out));
}

extern "C" int pgrc_match_export_entries(pgrc_match_ctx *c, const uint32_t *entry_read, const uint32_t *entry_org_idx, uint64_t n_entries,
                                         int32_t rev_compl_pair_file, int32_t byte_per_read_length, pgrc_export_streams *out) {
    if (!c || !out || (n_entries && (!entry_read || !entry_org_idx))) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    if (!c->have_results || !c->have_pg || !c->have_reads) { c->err = "export: run first"; return PGRC_E_STATE; }
    for (uint64_t k = 0; k < n_entries; k++)
        if (entry_read[k] != EX_NONE && entry_read[k] >= c->n) { c->err = "export_entries: read index out of range"; return PGRC_E_PARAM; }
    pgrc_match_ctx *w = c;
    int ge = view_for(c, &w);
    if (ge) return ge;
    PgrcDeviceScope scope(w->device);
    Bufs b;
    int e = export_entries(w, entry_read, entry_org_idx, n_entries, rev_compl_pair_file, byte_per_read_length ? 1u : 2u, b, out);
    b.release();
    if (e && w != c) c->err = w->err;
    if (e) pgrc_match_free_export(out);
    return e;
}

extern "C" int pgrc_match_export_original_order(pgrc_match_ctx *c, const pgrc_export_original_order_args *x, pgrc_export_streams *out) {
    if (!c || !x || !out || (c->n && !x->read_org_idx)) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    if (!c->have_results || !c->have_pg || !c->have_reads) { c->err = "export: run first"; return PGRC_E_STATE; }
    if (x->reads_total_count >= EX_SKIP) { c->err = "export_original_order: too many original indexes"; return PGRC_E_PARAM; }
    pgrc_match_ctx *w = c;
    int ge = view_for(c, &w);
    if (ge) return ge;
    PgrcDeviceScope scope(w->device);
    Bufs b;
    int e = export_original_order(w, x, b, out);
    b.release();
    if (e && w != c) c->err = w->err;
    if (e) pgrc_match_free_export(out);
    return e;
}
