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
//   * what bounds hash probing on gfx950 is the number of random lane-addresses per second (~51 G/s,
//     the same for 4-, 8- and 16-byte accesses: tools/ubench/gather2.hip), not HBM bytes.  The index
//     is therefore laid out so that a probe is ONE 16-byte gather: a bucket head holding the bucket's
//     two smallest entries; every entry carries a 22-bit fingerprint of the window symbols the
//     sparsified hash ignores, so a false candidate is rejected -- with exactly the reference's
//     head-reject accounting -- without fetching its text window.
//   * index build without atomics: (bucket, entry) records in position order, one STABLE rocPRIM radix sort by
//     bucket, one streaming pass that writes the heads; the sorted entries are ent[].  Stability makes the result
//     the reference's SERIAL index (ascending positions, the 13 smallest kept), bit for bit.
//   * match = one read per lane, as a per-lane state machine; the reference's sequential per-read
//     semantics (limit tightening, false-candidate budget, early exit) are kept exactly.
//   * all of it is random-access bound integer work: no MFMA.
#include <stdlib.h>

#include <cstring>
#include <type_traits>

#include "ctx.h"
#include "devutil.h"
#include "headfmt.h"
#include "matchdev.h"
#include "dualkern.h"

// ----------------------------------------------------------------------------- index build

// The build itself lives in idxsweep.hip (the default front end: two scatter passes that hash the text themselves) and
// idxsort.hip (the partition finish; the stable front end "own").  History: round 1 generated (bucket, entry) records and
// sorted them with the library's radix sort (16.3 ms per strand at C3), round 2 kept the library for the two top passes
// only; both forms left the product in round 5 -- no library kernel is on the path.

// a text shorter than K: every head of this strand's table is empty
__global__ void __launch_bounds__(256) k_heads_empty(ulonglong2 *__restrict__ head, uint64_t hs, uint32_t hsh) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < hs; h += (uint64_t)gridDim.x * blockDim.x)
        head[head_slot(h, hsh)] = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
}

int pgrc_copmem_build_index(pgrc_match_ctx *c, int strand) {
    const uint32_t K = (uint32_t)c->cp.K, k1 = (uint32_t)c->cp.k1;
    const uint64_t hs = c->cp.hash_size;
    c->npos = (c->G >= K) ? (c->G - K) / k1 + 1 : 0;
    const uint64_t npos = c->npos;
    int e;
    if (c->pair_build) {             // both strands' heads in one table of 32-byte slots (ctx.h)
        if ((e = pgrc_buf_ensure(c, c->d_headpair, hs * 4 * sizeof(uint64_t)))) return e;
        c->head_ptr = (ulonglong2 *)c->d_headpair.p + (strand ? c->pair_gm + 1u : 0u);
        c->head_sh = 1u | (c->pair_gm << 8);
    } else {
        if ((e = pgrc_buf_ensure(c, c->d_head, hs * 2 * sizeof(uint64_t)))) return e;
        c->head_ptr = (ulonglong2 *)c->d_head.p;
        c->head_sh = 0;
    }
    c->ent_ptr = nullptr;
    c->index_strand = strand;
    if (!npos) {
        hipLaunchKernelGGL(k_heads_empty, dim3((uint32_t)std::min<uint64_t>((hs + 255) / 256, 65536ull)), dim3(256), 0, c->stream, c->head_ptr, hs, c->head_sh);
        HIP_TRY(c, hipGetLastError());
        if ((e = pgrc_buf_ensure(c, c->d_sval[0], 64))) return e;
        c->ent_ptr = (const uint64_t *)c->d_sval[0].p;
        return PGRC_OK;
    }
    int hbits = 0;
    while ((1ull << hbits) < hs) hbits++;
    // How the records get grouped by bucket (both give the serial reference index, byte for byte):
    //   "sweep" (default): two hand-written scatter passes over the TOP bucket bits that hash the text themselves and
    //             shrink the records to 8 bytes (idxsweep.hip), then one block per partition finishes counting, cap, order
    //             and heads in LDS (idxsort.hip);
    //   "own":    the same finish behind STABLE scatter passes of (u32 bucket, u64 entry) records (idxsort.hip): what
    //             texts with more than 2^30 sampled positions take (the 8-byte records of "sweep" have no room for them).
    // The match kernels address ent[] with 32-bit indices unless they run their 64-bit-position variant; both front ends
    // stop at 2^32 samples (a text of k1 * 2^32 > 17 G symbols).
    if (c->opt.index_front == 0 && pgrc_os_applicable(c, (uint32_t)hbits)) return pgrc_os_build_index(c, strand, (uint32_t)hbits);
    if (!pgrc_ps_applicable(c, (uint32_t)hbits)) {
        c->err = "index build: texts with 2^32 or more sampled positions are not supported";
        return PGRC_E_PARAM;
    }
    const uint32_t cb = pgrc_ps_partition_bits((uint32_t)hbits);
    if ((e = pgrc_ps_scatter_front(c, strand, (uint32_t)hbits, cb))) return e;
    return pgrc_ps_finish(c, (const uint32_t *)c->d_skey[1].p, (const uint64_t *)c->d_sval[1].p, (uint32_t)hbits, cb, (uint64_t *)c->d_sval[0].p);
}

// ---- canonical export for tests: the reference's (cumm, sampledPositions) from heads + ent ----
#define SCAN_TPB 256
#define SCAN_EPT 16
#define SCAN_EPB (SCAN_TPB * SCAN_EPT)

__global__ void __launch_bounds__(256) k_export_counts(const ulonglong2 *__restrict__ head, uint32_t hsh, uint64_t hs, uint32_t *__restrict__ cnt) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < hs; h += (uint64_t)gridDim.x * blockDim.x)
        cnt[h] = head_count(head[head_slot(h, hsh)]);
}

__global__ void __launch_bounds__(256)
k_export_positions(const ulonglong2 *__restrict__ head, uint32_t hsh, const uint64_t *__restrict__ ent, const uint32_t *__restrict__ cumm,
                   uint64_t hs, uint32_t *__restrict__ positions) {
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < hs; h += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 hd = head[head_slot(h, hsh)];
        const uint32_t c = head_count(hd), lo = cumm[h];
        for (uint32_t j = 0; j < c; j++) {
            const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (c == 2 ? hd.y : ent[(hd.y & W1_BASE_MASK) + j - 1]);
            positions[lo + j] = (uint32_t)(e >> PGRC_FP_BITS);
        }
    }
}

int pgrc_copmem_export_index(pgrc_match_ctx *c, uint32_t *h_cumm, uint32_t *h_positions, uint64_t *count) {
    if (c->G >= (1ull << 32)) { c->err = "export_index: 32-bit positions only (test helper)"; return PGRC_E_PARAM; }
    const uint64_t hs = c->cp.hash_size;
    DevBuf cnt, cumm, pos, temp;
    auto cleanup = [&]() { pgrc_buf_free(cnt); pgrc_buf_free(cumm); pgrc_buf_free(pos); pgrc_buf_free(temp); };
    int e;
    if ((e = pgrc_buf_ensure(c, cnt, (hs + 2) * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, cumm, (hs + 2) * sizeof(uint32_t)))) { cleanup(); return e; }
    HIP_TRY(c, hipMemsetAsync(cnt.p, 0, (hs + 2) * sizeof(uint32_t), c->stream));
    const uint32_t sgrid = (uint32_t)((hs + 255) / 256 < 65536u * 8u ? (hs + 255) / 256 : 65536u * 8u);
    hipLaunchKernelGGL(k_export_counts, dim3(sgrid), dim3(256), 0, c->stream, (const ulonglong2 *)c->head_ptr, c->head_sh, hs, (uint32_t *)cnt.p);
    // cumm = exclusive scan of the counts (the build's own scan kernels; in place in `cnt`, then copied)
    if ((e = pgrc_buf_ensure(c, temp, (pgrc_ps_scan_blocks(hs + 2) + 2) * sizeof(uint32_t)))) { cleanup(); return e; }
    if ((e = pgrc_ps_scan_u32(c, (uint32_t *)cnt.p, hs + 2, (uint32_t *)temp.p))) { cleanup(); return e; }
    hipError_t he = hipMemcpyAsync(cumm.p, cnt.p, (hs + 2) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
    uint32_t total = 0;
    if (he == hipSuccess) he = hipMemcpyAsync(&total, (uint32_t *)cumm.p + hs, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) { cleanup(); c->err = std::string("export_index: ") + hipGetErrorString(he); return PGRC_E_DEVICE; }
    if (count) *count = total;
    if (h_cumm && hipMemcpy(h_cumm, cumm.p, (hs + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return PGRC_E_DEVICE; }
    if (h_positions && total) {
        if ((e = pgrc_buf_ensure(c, pos, (size_t)total * sizeof(uint32_t)))) { cleanup(); return e; }
        hipLaunchKernelGGL(k_export_positions, dim3(sgrid), dim3(256), 0, c->stream, (const ulonglong2 *)c->head_ptr, c->head_sh,
                           c->ent_ptr, (const uint32_t *)cumm.p, hs, (uint32_t *)pos.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(h_positions, pos.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            cleanup();
            c->err = "export_index: positions";
            return PGRC_E_DEVICE;
        }
    }
    cleanup();
    return PGRC_OK;
}

// ----------------------------------------------------------------------------- matching

struct MatchArgs {
    const uint32_t *pg;
    uint64_t G;
    const uint32_t *reads;
    uint64_t n, stride;
    const uint8_t *nflag;
    const ulonglong2 *head;       // head of bucket h at head[head_slot(h, hsh)] (headfmt.h: one table per strand, or the pair table)
    uint32_t hsh;
    const uint64_t *ent;
    uint64_t *pos;
    uint8_t *rc, *mism;
    unsigned long long *counters; // [0] searched [1] candidates [2] probes
    unsigned long long *work;     // global read cursor of the persistent match kernel
    uint32_t L, K, k2, mask, kmax, kmin, strand;
    uint32_t k1, early;           // sampling step of the index; early = 1: stop a read once nothing can be accepted any more
    uint32_t chunk;               // reads a wave reserves per visit to the work counter (pgrc_match_chunk)
    // the screened schedule of a two-pass run (see "Exact-match screen" at the kernel): phase 0 = a plain pass,
    // 1 = screen (RC text, exact alignments only, flags and positions to scr_*), 2 = forward pass that honours the flags
    uint32_t phase;
    uint64_t *scr_pos;
    uint8_t *scr_flag;
};

// The query as a per-lane state machine.  A plain loop over seeds makes a whole wave wait out up to
// three dependent memory latencies per seed (bucket head -> bucket entry -> text window) whenever ANY
// of its 64 reads needs them.  Here every lane advances its own read by one memory access per
// iteration -- a head (M_PROBE), the next entries of a long bucket (M_ENTRY) or a text window to verify
// (M_VERIFY) -- and all lanes' loads of an iteration are issued together, so an iteration costs one
// latency.  The per-read order of events is exactly the reference's.
//
// Every random access costs one of the chip's ~50 G line requests per second whatever its width, so:
//   * bucket entries beyond the two in the head are fetched two at a time (one 16-B gather);
//   * verified alignments are remembered per read in a small LDS cache (8 direct-mapped slots): a read
//     accepted with m > 0 mismatches -- or lying in a repeat -- meets the same alignments again at every
//     later seed sampled there; their head/tail counts cannot change, only the limit they are judged
//     against does, so the text window is fetched once.
#ifndef PROBE_AHEAD
#define PROBE_AHEAD 1       // probing lanes fetch the next seed's bucket head together with their own
#endif
#ifdef MATCH_WAVES_PER_EU     // experiments only (tools/variants.sh): the compiler's own choice is 6 for L <= 160
#define MATCH_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(MATCH_WAVES_PER_EU)))
#else
#define MATCH_OCCUPANCY_ATTR
#endif
// reads a wave reserves per visit to the global work counter: 1024 for a whole read set (256 / 512 / 1024: step +0 / -0.2 /
// -0.4 % at C3), less for a short launch -- a block of a streamed run: 7 M reads over ~5000 resident waves are 1.35 chunks of
// 1024 per wave, i.e. half the waves do two chunks while the others wait (10 ms per block instead of 6)
// (Round 4, measured and dropped: "guided" reservation -- a wave that comes for reads near the end of the set takes
//  (reads left) / (resident waves) of them, at least 128, so that the end is spread over all waves.  Whatever the parameters
//  (profiles/r04_guided_ab.txt) the dual kernel ran 10 ms SLOWER at C3, 65.4 -> 75.4 ms.)
static uint32_t pgrc_match_chunk(const pgrc_match_ctx *c, uint64_t n) {
    const uint64_t waves = (uint64_t)c->num_cus * 20u;           // (about what is resident)
    uint32_t chunk = MATCH_CHUNK;
    while (chunk > 64u && n / chunk < waves * 8u) chunk >>= 1;
    return chunk;
}

// Persistent, self-refilling lanes.  With one read per lane for the lifetime of a wave, ~35 % of the
// lane-iterations are idle: reads that match exactly leave after a few seeds (30 % of the reads in the forward
// pass, 43 % in the RC pass) while their wave runs on for the full seed list -- and the memory system is only
// kept busy by lanes that have a gather in flight.  Here a lane that finishes its read immediately takes the
// next one of its wave's chunk (ranks by ballot/popcount, no atomics; a wave reserves MATCH_CHUNK reads at a time
// from one global counter), so every resident lane always has a gather in flight until the read set is
// exhausted.  The per-read sequence of events is untouched.
// Early stop (round 2).  The reference probes every seed of a read even when nothing can be accepted any more; the
// result is the same if the read stops there, and when that is can be proven from what the probes have seen.  The index
// samples every k1-th text position and the read is probed every k2-th symbol (coprime), so an alignment at text
// position p is found through exactly the seeds s with (p + s) % k1 == 0, and k1 consecutive seeds hold one seed of
// every such class.  Call k1 consecutive probes a ROUND, start one every rper = ceil(K / (k1 k2)) k1 seeds (the windows
// of one class in two rounds are then >= K symbols apart: disjoint), and call a round CLEAN if every one of its buckets
// was complete: fewer than 13 entries (nothing dropped by the cap at build time, CopMEMMatcher.cpp:156-159) and not cut
// to 4 by the falses budget (:510-514).  An alignment with m mismatches leaves at least one of any m + 1 disjoint
// windows of its class untouched; that window hashes to the bucket holding p + s, so in a clean round the alignment is
// LOOKED AT -- and accepted if m <= limit at that time (the fingerprint filter and the verify cache are exact).  Limits
// only fall, and an accepted alignment with m mismatches lowers the limit to m - 1.  Hence: after rclean clean rounds
// no alignment with <= rclean - 1 mismatches exists that has not been accepted, and once rclean > limit no later
// candidate can pass `m <= limit`: (cur, best) are final.  Unmatched 150-bp reads at k <= 3 stop after seed 50 of 62,
// a read waiting for an exact match (limit 0) after 5 seeds.  PGRC_EARLY_STOP=0 restores the full loops.
// Exact-match screen (round 2).  In a two-pass run a read that matches the OTHER strand exactly still pays a whole
// forward query to learn that the forward strand has nothing better.  With both indexes alive the run is scheduled as
//   phase 1  every read not yet matched exactly looks on the RC text for an EXACT alignment only (limit 0: one clean
//            round); found, and twice its false-candidate count within the falses budget -> flagged, position kept.
//            (A query at a higher limit counts at most twice as many falses over the same candidates, so no run of the
//            real RC query could have cut a bucket before reaching that alignment: it would accept the same one.)
//   phase 2  the forward pass; a flagged read again looks for an exact alignment only.  Found within the budget: final
//            (the real forward query reaches the same alignment, and the RC pass skips exactly matched reads).  Proven
//            absent by the early-stop rule: whatever the real forward query finds has >= 1 mismatch, the real RC query
//            then returns phase 1's alignment: written as the final result here.  Otherwise (budget exceeded, no clean
//            round) the lane simply starts the read again as a real query.  Unflagged reads: the real query.
//   phase 0  (RC pass) unchanged: reads matched exactly are skipped.
// Only with min_mismatches == 0.  oracle/pgrc_oracle.c restates this schedule (pgrc_or_match_copmem_screened) and
// tests/test_early_stop_rule.py expects it to equal the reference's two passes on every input.
// POS64: text positions need more than 32 bits (Pg >= 4 Gi symbols: the reference's u64 index branch,
// CopMEMMatcher.cpp:579-586); otherwise positions are kept in one register.
template <int NW, int KQ, bool POS64, int STAGE = 0>
__global__ void __launch_bounds__(MATCH_TPB) MATCH_OCCUPANCY_ATTR k_copmem_match_sm(const MatchArgs a) {
    typedef typename std::conditional<POS64, uint64_t, uint32_t>::type pos_t;
    constexpr pos_t POS_NONE = (pos_t)~(pos_t)0;
    constexpr uint32_t EPOCH_BITS = POS64 ? 13u : 16u; // POS64 keeps position bits 32..39 next to the counts
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t fpm_tab[SM_MAX_SEEDS];
    __shared__ uint2 vcache[VC_SLOTS][MATCH_TPB]; // {position, head count | tail count << 8 | epoch << 16}
    __shared__ uint32_t rd_lds[NW][MATCH_TPB];    // the read itself: only needed when a window is verified
    // STAGE > 0 (experiment): the next STAGE reads of the wave's chunk are fetched with full-line loads into LDS and
    // refilling lanes take their read from there instead of loading it themselves
    constexpr int SW = STAGE > 0 ? STAGE : 1;
    __shared__ uint32_t stg[MATCH_TPB / 64][NW][SW];
    __shared__ uint8_t stg_c[MATCH_TPB / 64][SW], stg_f[MATCH_TPB / 64][SW];
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t wbeg = 0, wend = 0, wnext = 0;      // the staged window and the next read of it to hand out (wave-uniform)
    hash_lut_init(lut);
    const int H = ((int)a.L / 8) * 8;
    const uint32_t nseeds = (a.L - a.K) / a.k2 + 1; // seeds s = 0, k2, ... with s + K <= L
    for (uint32_t t = threadIdx.x; t < nseeds && t < SM_MAX_SEEDS; t += blockDim.x)
        fpm_tab[t] = fp_head_mask(a.K, t * a.k2, (uint32_t)H);
#pragma unroll
    for (int k = 0; k < VC_SLOTS; k++) vcache[k][threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_search = 0, n_cand = 0, n_probe = 0, n_ent = 0, n_ver = 0; // wave-uniform (ballot popcounts): SGPRs
    uint32_t sh[NW];              // the read, shifted so that the current seed window starts at bit 0
#pragma unroll
    for (int k = 0; k < NW; k++) sh[k] = 0u;

    ReadState<pos_t> st;
    st.limit = 0; st.falses = 0; st.cur = 0; st.best = POS_NONE; st.done = false;
    const uint32_t budget = (a.L + 1u - a.K) / a.k2;
    const uint32_t sbits = 2u * a.k2;

    enum { M_PROBE = 0, M_ENTRY = 1, M_VERIFY = 2, M_NEED = 3, M_ADV = 4, M_DEAD = 5 };
    uint32_t mode = M_NEED;
    uint32_t idx = 0;             // the read this lane works on (reads are counted in 32 bits, pg-config.h:21-22)
    uint32_t cin = 0, epoch = 0;
    uint32_t cnext = 0, cend = 0; // this wave's reserved range of reads (wave-uniform: kept in SGPRs)
    uint32_t si = 0;              // seed index: s = si * k2
    // Early stop (exact, see the comment at the kernel): rounds of k1 consecutive seeds, a round every rper seeds
    const uint32_t rper = (a.K + a.k1 * a.k2 - 1u) / (a.k1 * a.k2) * a.k1;
    uint32_t rq = 0, rclean = 0;  // seed index inside the current period; rounds whose probes all saw complete buckets
    bool rdirty = false;          // the current round had a capped or truncated bucket
    bool spec = false;            // this read only looks for an exact alignment (phases 1 and 2 of the screened schedule)
    pos_t lo = 0;          // index into ent[] (as many entries as sampled positions: 32 bits unless POS64)
    uint32_t nb = 0, j = 0, fp_read = 0;
    pos_t cand_p = 0;
    uint64_t pend_e = 0;          // an entry already in registers (entry 1 of the head / second half of a pair)
    bool has_pend = false;
#if PROBE_AHEAD
    ulonglong2 hdn = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);   // the head of this lane's NEXT seed, fetched ahead
    uint32_t fpn = 0;
    bool have_n = false;
#endif
    constexpr int PWN = ((NW + 1 + 3) / 4) * 4;

    // judge a verified alignment (head count mh, tail count mt) exactly as CopMEMMatcher.cpp:536-560
    auto judge = [&](uint32_t mh, uint32_t mt, pos_t p) {
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

    for (;;) {
        // ---- refill: lanes without a read take the next ones of the wave's chunk
        const unsigned long long need = __ballot(mode == M_NEED);
        if (need) {
            if (cnext == cend) { // reserve another chunk (one atomic per wave and MATCH_CHUNK reads)
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(a.work, (unsigned long long)a.chunk);
                base = __shfl(base, 0, 64);
                cnext = __builtin_amdgcn_readfirstlane((uint32_t)min((uint64_t)base, a.n));
                cend = __builtin_amdgcn_readfirstlane((uint32_t)min((uint64_t)base + a.chunk, a.n));
            }
            if (STAGE > 0 && wnext == wend && cnext != cend) {    // the window is used up: stage the chunk's next reads
                const uint32_t nst = min((uint32_t)SW, cend - cnext);
                wbeg = wnext = cnext;
                wend = cnext = __builtin_amdgcn_readfirstlane(cnext + nst);
                if (lane < nst) {
#pragma unroll
                    for (int k = 0; k < NW; k++) stg[wv][k][lane] = a.reads[(uint64_t)k * a.stride + wbeg + lane];
                    stg_c[wv][lane] = a.mism[wbeg + lane];
                    stg_f[wv][lane] = (uint8_t)((a.nflag ? (a.nflag[wbeg + lane] & 1u) : 0u) | (a.phase == 2u ? (a.scr_flag[wbeg + lane] & 1u) << 1 : 0u));
                }
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // a wave's LDS accesses are served in order
            }
            const uint32_t avail = STAGE > 0 ? wend - wnext : cend - cnext;
            if (avail == 0) {
                if (mode == M_NEED) mode = M_DEAD;               // the read set is exhausted
            } else {
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                const uint32_t take = min((uint32_t)__popcll(need), avail);
                bool started = false;
                if (mode == M_NEED && rank < take) {
                    const uint32_t sj = wnext - wbeg + rank;
                    idx = STAGE > 0 ? wbeg + sj : cnext + rank;
                    cin = STAGE > 0 ? stg_c[wv][sj] : a.mism[idx];
                    const uint32_t rflags = STAGE > 0 ? (uint32_t)stg_f[wv][sj]
                                                      : (uint32_t)((a.nflag ? (a.nflag[idx] & 1u) : 0u) | (a.phase == 2u ? (a.scr_flag[idx] & 1u) << 1 : 0u));
                    const bool skip = (rflags & 1u) || cin <= a.kmin;   // ReadsMatchers.cpp:430; 'N' reads: byte path
                    if (!skip) {
#pragma unroll
                        for (int k = 0; k < NW; k++)
                            rd_lds[k][threadIdx.x] = sh[k] = STAGE > 0 ? stg[wv][k][sj] : a.reads[(uint64_t)k * a.stride + idx];
                        spec = a.phase == 1u || (a.phase == 2u && (rflags & 2u));   // exact alignments only (the screened schedule)
                        st.limit = spec ? 0u : (cin < a.kmax) ? cin - 1u : a.kmax;   // :488-489
                        st.falses = 0;
                        st.cur = cin;
                        st.best = POS_NONE;
                        st.done = false;
                        si = 0;
                        rq = 0;
                        rclean = 0;
                        rdirty = false;
                        has_pend = false;
#if PROBE_AHEAD
                        have_n = false;
#endif
                        epoch = (epoch + 1u) & ((1u << EPOCH_BITS) - 1u); // invalidates this lane's verify-cache entries
                        if (epoch == 0) {                            // wrapped: really clear them
#pragma unroll
                            for (int k = 0; k < VC_SLOTS; k++) vcache[k][threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                            epoch = 1;
                        }
                        started = true;
                        mode = M_PROBE;
                    }
                }
                if (STAGE > 0) wnext = __builtin_amdgcn_readfirstlane(wnext + take);
                else cnext = __builtin_amdgcn_readfirstlane(cnext + take);
                n_search += (uint32_t)__popcll(__ballot(started));
            }
        }
        if (!__any(mode != M_DEAD)) break;

        const uint32_t m0 = mode;
        // ---- issue this iteration's loads
        ulonglong2 hd = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
        uint64_t v = 0;
        bool counted_ent = false;
        uint32_t ncand_it = 0;        // candidates tested in this iteration (a probing lane may consume two heads)
#if PROBE_AHEAD
        // A probing lane also fetches the head of the NEXT seed: if its own bucket turns out empty (every second one) it
        // consumes that head in the same iteration, otherwise it keeps it for later.  Two gathers in flight per probing
        // lane instead of one; nothing is fetched twice and nothing beyond the read's last seed.
        ulonglong2 hd2 = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
        uint32_t fp2 = 0;
        bool got2 = false;
#endif
        if (m0 == M_PROBE) {
#if PROBE_AHEAD
            if (have_n) {
                hd = hdn;
                fp_read = fpn;
            } else
#endif
            {
                const uint32_t h = hash_fp_window<KQ>(sh[0], NW > 1 ? sh[1 % NW] : 0u, NW > 2 ? sh[2 % NW] : 0u,
                                                      NW > 3 ? sh[3 % NW] : 0u, a.K, lut, &fp_read) & a.mask;
                hd = a.head[head_slot(h, a.hsh)];
            }
#if PROBE_AHEAD
            if (si + 1 < nseeds) {
                // the window of the next seed: the read shifted by one more seed step (sbits <= 30)
                const uint32_t n0 = funnel_r(sh[0], NW > 1 ? sh[1 % NW] : 0u, sbits);
                const uint32_t n1 = NW > 1 ? funnel_r(sh[1 % NW], NW > 2 ? sh[2 % NW] : 0u, sbits) : 0u;
                const uint32_t n2 = NW > 2 ? funnel_r(sh[2 % NW], NW > 3 ? sh[3 % NW] : 0u, sbits) : 0u;
                const uint32_t n3 = NW > 3 ? funnel_r(sh[3 % NW], NW > 4 ? sh[4 % NW] : 0u, sbits) : 0u;
                const uint32_t h2 = hash_fp_window<KQ>(n0, n1, n2, n3, a.K, lut, &fp2) & a.mask;
                hd2 = a.head[head_slot(h2, a.hsh)];
                got2 = true;
            }
            have_n = false;
#endif
        } else if (m0 == M_ENTRY) {
            if (has_pend) {
                v = pend_e;
                has_pend = false;
            } else {
                // entries j, j+1 of the bucket in one gather (the second is kept only if the bucket has it)
                const U64x2A8 q = *reinterpret_cast<const U64x2A8 *>(a.ent + lo + j - 1);
                v = q.x;
                pend_e = q.y;
                has_pend = j + 1 < nb;
                counted_ent = true;
            }
        }
        // ---- consume
        uint32_t next = m0;
        // one index entry against the current seed (CopMEMMatcher.cpp:517-560); sets the lane's next state
        auto take_entry = [&](const uint64_t e) {
            next = (j < nb) ? M_ENTRY : M_ADV;
            const uint32_t s = si * a.k2;
            const uint64_t sp = e >> PGRC_FP_BITS;
            if ((uint64_t)s <= sp && sp - s + a.L <= a.G) {      // :517-520
                ncand_it++;
                const pos_t p = (pos_t)(sp - s);
                const uint32_t x = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
                if ((uint32_t)__popc((x | (x >> 1)) & fpm_tab[si]) > st.limit) {
                    st.falses += 1;                              // certain head reject
                } else {
                    const uint2 cv = vcache[((uint32_t)p * 0x9E3779B1u) >> (32 - VC_BITS)][threadIdx.x];
                    const bool hit = POS64 ? (cv.x == (uint32_t)p && (cv.y >> 19) == epoch &&
                                              ((cv.y >> 11) & 0xFFu) == (uint32_t)((uint64_t)p >> 32))
                                           : (cv.x == (uint32_t)p && (cv.y >> 16) == epoch);
                    if (hit) {
                        judge(cv.y & 0xFFu, (cv.y >> 8) & (POS64 ? 0x7u : 0xFFu), p);
                        if (st.done) next = M_NEED;
                    } else {
                        cand_p = p;
                        next = M_VERIFY;
                    }
                }
            }
        };
        // a bucket head for the current seed: empty -> M_ADV, else its first entry is tested
        auto take_head = [&](const ulonglong2 hx) {
            const uint32_t cnt = head_count(hx);
            if (!cnt) {
                next = M_ADV;
                return;
            }
            nb = cnt;
            if (st.falses > budget) nb = min(nb, PGRC_TRUNC_BUCKET); // :510-514
            // not every sampled position with this hash is looked at: the bucket was capped at build time or is cut here
            if (rq < a.k1 && (cnt >= PGRC_BUCKET_CAP || nb < cnt)) rdirty = true;
            has_pend = cnt == 2 && nb > 1; // entry 1 of a two-entry bucket sits in the head
            pend_e = hx.y;
            lo = (pos_t)(hx.y & W1_BASE_MASK); // count >= 3: entries 1.. live at ent[lo + j - 1]
            j = 1;
            take_entry(hx.x & ENT_MASK);
        };
        auto advance = [&]() {                                       // to the next seed of this read
            si++;
            has_pend = false;
            if (rq == a.k1 - 1u) {                                   // a round is behind this read
                rclean += rdirty ? 0u : 1u;
                rdirty = false;
            }
            rq = (rq + 1u == rper) ? 0u : rq + 1u;
#pragma unroll
            for (int k = 0; k < NW - 1; k++) sh[k] = funnel_r(sh[k], sh[k + 1], sbits);
            sh[NW - 1] >>= sbits;
        };
        bool second_probe = false;
        bool ended_rule = false;      // the early-stop rule (not the end of the seed list) ends this read in this iteration
        if (m0 == M_VERIFY) {
            uint32_t pw[PWN];
            const uint32_t b = ((uint32_t)cand_p & 15u) * 2u;
            const uint32_t *src = a.pg + (cand_p >> 4); // the text is padded: PWN words are always in bounds
#pragma unroll
            for (int k = 0; k < PWN; k += 4) {
                const u32x4 q = reinterpret_cast<const U32x4A4 *>(src + k)->v;
                pw[k] = q.x; pw[k + 1] = q.y; pw[k + 2] = q.z; pw[k + 3] = q.w;
            }
            uint32_t mh = 0, mt = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const uint32_t tw = funnel_r(pw[k], pw[k + 1], b);
                const uint32_t rw = rd_lds[k][threadIdx.x];
                mh += mism2(tw, rw, sym_mask(k, 0, H));
                mt += mism2(tw, rw, sym_mask(k, H, (int)a.L));
            }
            vcache[((uint32_t)cand_p * 0x9E3779B1u) >> (32 - VC_BITS)][threadIdx.x] =
                make_uint2((uint32_t)cand_p, POS64 ? (mh | (mt << 8) | ((uint32_t)((uint64_t)cand_p >> 32) << 11) | (epoch << 19))
                                                   : (mh | (mt << 8) | (epoch << 16)));
            judge(mh, mt, cand_p);
            next = st.done ? M_NEED : (j < nb ? M_ENTRY : M_ADV);
        } else if (m0 == M_ENTRY) {
            j++;
            take_entry(v);
        } else if (m0 == M_PROBE) {
            take_head(hd);
#if PROBE_AHEAD
            if (got2) {
                if (next == M_ADV) {                                 // empty bucket: the next seed's head is here already
                    advance();
                    if (a.early && rclean > st.limit) {              // ... but nothing can be accepted any more
                        next = M_NEED;
                        ended_rule = true;
                    } else {
                        fp_read = fp2;
                        second_probe = true;
                        take_head(hd2);
                    }
                } else {                                             // keep it for when this lane reaches that seed
                    hdn = hd2;
                    fpn = fp2;
                    have_n = true;
                }
            }
#endif
        }
        if (next == M_ADV) {
            advance();
            ended_rule = a.early && rclean > st.limit;
            next = (si < nseeds && !ended_rule) ? M_PROBE : M_NEED;
        }
        // work counters, wave-wide (scalar popcounts instead of five per-lane registers)
        n_probe += (uint32_t)__popcll(__ballot(m0 == M_PROBE)) + (uint32_t)__popcll(__ballot(second_probe));
        n_ent += (uint32_t)__popcll(__ballot(counted_ent));
        n_ver += (uint32_t)__popcll(__ballot(m0 == M_VERIFY));
        n_cand += (uint32_t)__popcll(__ballot(ncand_it >= 1)) + (uint32_t)__popcll(__ballot(ncand_it >= 2));
        if (next == M_NEED && m0 <= M_VERIFY) {
            if (spec) {
                // the screened schedule: an exact alignment, found with so few false candidates that no real query
                // could have cut a bucket on its way to it?
                const bool exact = st.done && st.cur == 0u && 2u * st.falses <= budget;
                if (a.phase == 1u) {
                    a.scr_flag[idx] = exact ? (uint8_t)1 : (uint8_t)0;
                    if (exact) a.scr_pos[idx] = (uint64_t)st.best;
                } else if (exact) {                                  // forward exact: final
                    a.pos[idx] = (uint64_t)st.best;
                    a.rc[idx] = 0;
                    a.mism[idx] = 0;
                } else if (!st.done && ended_rule) {                 // no forward exact exists: phase 1's RC alignment is final
                    a.pos[idx] = a.G - (a.scr_pos[idx] + a.L);
                    a.rc[idx] = 1;
                    a.mism[idx] = 0;
                } else {                                             // undecided: the real query, from the first seed
                    spec = false;
                    st.limit = (cin < a.kmax) ? cin - 1u : a.kmax;
                    st.falses = 0;
                    st.cur = cin;
                    st.best = POS_NONE;
                    st.done = false;
                    si = 0;
                    rq = 0;
                    rclean = 0;
                    rdirty = false;
                    has_pend = false;
#if PROBE_AHEAD
                    have_n = false;
#endif
#pragma unroll
                    for (int k = 0; k < NW; k++) sh[k] = rd_lds[k][threadIdx.x];
                    next = M_PROBE;
                }
            } else if (st.best != POS_NONE && st.cur < cin) {
                // this read is finished (ReadsMatchers.cpp:437-447)
                a.pos[idx] = a.strand ? a.G - ((uint64_t)st.best + a.L) : (uint64_t)st.best;
                a.rc[idx] = (uint8_t)a.strand;
                a.mism[idx] = (uint8_t)st.cur;
            }
        }
        mode = next;
    }
    if (a.counters) {
        if (lane == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)n_search);
            atomicAdd(&a.counters[1], (unsigned long long)n_cand);
            atomicAdd(&a.counters[2], (unsigned long long)n_probe);
            atomicAdd(&a.counters[3], (unsigned long long)n_ent);
            atomicAdd(&a.counters[4], (unsigned long long)n_ver);
        }
    }
}

// Reads containing 'N' (the reference's N read set, ACGNT-packed; an N never equals a Pg symbol and is hashed as the
// byte 0x4E -- SURVEY.md Appendix A notes).  One read per lane, plain loops (they are a percent or two of the reads),
// but on the same packed representation as the main kernel: the read as 2-bit words (N packs as code 0) plus a 16-bit
// mask of its N positions per word, both in LDS.  Windows without an N hash through the LUT and use the fingerprint
// shortcut exactly like k_copmem_match_sm; windows with an N rebuild the ASCII bytes for the hash and always verify.
// Round 4: a PERSISTENT grid walks the side list in tiles of NREAD_TPB reads, and a lane takes its read through the
// strands first .. last one after the other (the RC query of a read only needs that read's forward result:
// ReadsMatchers.cpp:430-447) -- so a two-pass run needs ONE launch, small enough (a few waves per CU) to sit beside the
// dual kernel from its start instead of running behind it.  The early-stop rule of k_copmem_match_sm applies unchanged:
// an N of the read is a mismatch of every alignment, so a window that an alignment's mismatches leave untouched holds no N
// and hashes like the text.
struct NStrandArgs {
    const uint32_t *pg;
    const ulonglong2 *head;       // headfmt.h
    uint32_t hsh;
    const uint64_t *ent;
    unsigned long long *counters; // [0] searched [1] candidates [2] probes (added to the strand's pass counters)
};
struct NReadArgs {
    NStrandArgs st[2];
    uint64_t G;
    uint64_t *pos;
    uint8_t *rc, *mism;
    const uint32_t *nidx;         // the side list: read index, ASCII row
    const uint8_t *nascii;
    const uint8_t *skip_flag;     // not null: reads flagged 3 there were taken by the dual kernel
    uint64_t nn;
    uint32_t L, K, k1, k2, mask, kmax, kmin, early;
    uint32_t first, last;         // strands to go through
};

#define NREAD_TPB 128
__global__ void __launch_bounds__(NREAD_TPB) k_copmem_match_n(const NReadArgs a) {
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    // dynamic LDS, sized for this read length: rdw[NWr + 1][NREAD_TPB] = the packed read (+ one zero word: windows read
    // one word ahead), nmw[NWr + 1][NREAD_TPB] = bit k of word w: symbol 16 w + k is an N.  A lane only ever touches its
    // own column.
    extern __shared__ uint32_t nread_lds[];
    uint32_t (*rdw)[NREAD_TPB] = reinterpret_cast<uint32_t (*)[NREAD_TPB]>(nread_lds);
    uint32_t (*nmw)[NREAD_TPB] = reinterpret_cast<uint32_t (*)[NREAD_TPB]>(nread_lds + ((a.L + 15) / 16 + 1) * NREAD_TPB);
    hash_lut_init(lut);
    __syncthreads();
    const uint32_t NWr = (a.L + 15) / 16;
    const int H = ((int)a.L / 8) * 8;
    const uint32_t budget = (a.L + 1u - a.K) / a.k2;
    const uint32_t rper = (a.K + a.k1 * a.k2 - 1u) / (a.k1 * a.k2) * a.k1;    // early stop: a round every rper seeds (k_copmem_match_sm)
    uint64_t n_srch[2] = {0, 0}, n_cand[2] = {0, 0}, n_probe[2] = {0, 0};
    for (uint64_t tile = blockIdx.x; tile * NREAD_TPB < a.nn; tile += gridDim.x) {
        const uint64_t t = tile * NREAD_TPB + threadIdx.x;
        if (t >= a.nn) continue;
        const uint64_t i = a.nidx[t];
        if (a.skip_flag && a.skip_flag[i] == 3) continue;
        uint32_t cin = a.mism[i];
        if (cin <= a.kmin) continue;                             // ReadsMatchers.cpp:430 (and nothing a later strand could improve)
        for (uint32_t w = 0; w <= NWr; w++) {
            uint32_t pw = 0, nm = 0;
            if (w < NWr) {
                const uint8_t *row = a.nascii + t * a.L + 16 * w;
                for (uint32_t k = 0; k < 16 && 16 * w + k < a.L; k++) {
                    const uint32_t ch = row[k];
                    uint32_t x = (ch >> 1) & 3u;
                    x ^= x >> 1;                                 // A0 C1 G2 T3
                    if (ch == 'N') { nm |= 1u << k; x = 0; }
                    pw |= x << (2 * k);
                }
            }
            rdw[w][threadIdx.x] = pw;
            nmw[w][threadIdx.x] = nm;
        }
        for (uint32_t strand = a.first; strand <= a.last && cin > a.kmin; strand++) {
            const NStrandArgs &sa = a.st[strand];
            n_srch[strand]++;
            uint32_t limit = (cin < a.kmax) ? cin - 1u : a.kmax;  // :488-489
            uint32_t falses = 0, cur = cin;
            uint64_t best = PGRC_NOT_MATCHED_POS;
            bool done = false;
            uint32_t rq = 0, rclean = 0;
            bool rdirty = false;
            for (uint32_t s = 0; s + a.K <= a.L && !done; s += a.k2) {
                // the K-symbol window at s: four words of symbols and their N flags
                const uint32_t q = s >> 4, sh2 = (s & 15u) * 2u, sh1 = s & 15u;
                uint32_t ww[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t lo0 = (q + k <= NWr) ? rdw[q + k][threadIdx.x] : 0u, hi0 = (q + k + 1 <= NWr) ? rdw[q + k + 1][threadIdx.x] : 0u;
                    ww[k] = funnel_r(lo0, hi0, sh2);
                }
                // N flags of symbols s .. s+K-1 (K <= 56): bit x = symbol s + x
                uint64_t nbits = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (q + k <= NWr) nbits |= (uint64_t)(nmw[q + k][threadIdx.x] & 0xFFFFu) << (16 * k);
                if (sh1) {
                    const uint64_t top = (q + 4 <= NWr) ? (uint64_t)(nmw[q + 4][threadIdx.x] & 0xFFFFu) : 0ull;
                    nbits = (nbits >> sh1) | (top << (64 - sh1));
                }
                nbits &= (1ull << a.K) - 1ull;
                const bool wn = nbits != 0;                        // the window holds an N
                uint32_t fp_read = 0, h;
                if (!wn) {
                    h = copmem_hash32_fp(ww[0], ww[1], ww[2], ww[3], a.K, lut, &fp_read);
                } else {
                    // ASCII bytes of the window for maRushPrime1HashSparsified (Hashes.h:54-76): an N is the byte 0x4E
                    h = a.K;
                    for (uint32_t j = 0; j < a.K / 4; j++) {
                        uint32_t w = 0;
                        const uint32_t nby = (j < 3) ? 3u : 2u;
                        for (uint32_t b = 0; b < nby; b++) {
                            const uint32_t x = 4 * j + b;
                            const uint32_t wsel = x < 16 ? ww[0] : x < 32 ? ww[1] : x < 48 ? ww[2] : ww[3];
                            const uint32_t code = (wsel >> (2u * (x & 15u))) & 3u;
                            w |= (((nbits >> x) & 1ull) ? (uint32_t)'N' : code2ascii(code)) << (8 * b);
                        }
                        h = (h ^ (w + j)) * 171717u;
                    }
                }
                h &= a.mask;
                n_probe[strand]++;
                const ulonglong2 hd = sa.head[head_slot(h, sa.hsh)];
                const uint32_t cnt = head_count(hd);
                if (cnt) {
                    uint32_t nb = cnt;
                    if (falses > budget) nb = min(nb, PGRC_TRUNC_BUCKET);          // :510-514
                    if (rq < a.k1 && (cnt >= PGRC_BUCKET_CAP || nb < cnt)) rdirty = true;   // not every position with this hash is looked at
                    const uint32_t fpm = wn ? 0u : fp_head_mask(a.K, s, (uint32_t)H);
                    for (uint32_t j = 0; j < nb; j++) {
                        const uint64_t e = (j == 0) ? (hd.x & ENT_MASK) : (cnt == 2 ? hd.y : sa.ent[(hd.y & W1_BASE_MASK) + j - 1]);
                        const uint64_t sp = e >> PGRC_FP_BITS;
                        if ((uint64_t)s > sp) continue;                                 // :517-520
                        const uint64_t p = sp - s;
                        if (p + a.L > a.G) continue;
                        n_cand[strand]++;
                        if (!wn) {                                                      // certain head reject, as in k_copmem_match_sm
                            const uint32_t x = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
                            if ((uint32_t)__popc((x | (x >> 1)) & fpm) > limit) { falses += 1; continue; }
                        }
                        uint32_t mh = 0, mt = 0;
                        const uint32_t *src = sa.pg + (p >> 4);
                        const uint32_t b = ((uint32_t)p & 15u) * 2u;
                        uint32_t lo = src[0];
                        for (uint32_t k = 0; k < NWr; k++) {
                            const uint32_t hi = src[k + 1];
                            const uint32_t x = funnel_r(lo, hi, b) ^ rdw[k][threadIdx.x];
                            uint32_t nmb = nmw[k][threadIdx.x];                         // bit i -> bit 2 i
                            nmb = (nmb | (nmb << 8)) & 0x00FF00FFu;
                            nmb = (nmb | (nmb << 4)) & 0x0F0F0F0Fu;
                            nmb = (nmb | (nmb << 2)) & 0x33333333u;
                            nmb = (nmb | (nmb << 1)) & 0x55555555u;
                            const uint32_t d = (x | (x >> 1) | nmb);
                            mh += (uint32_t)__popc(d & sym_mask((int)k, 0, H));
                            mt += (uint32_t)__popc(d & sym_mask((int)k, H, (int)a.L));
                            lo = hi;
                        }
                        if (mh > limit) { falses += 1; continue; }                      // :536-539
                        const uint32_t m = mh + mt;
                        if (m > limit) { falses += 2; continue; }                       // :542-551 (counted twice)
                        cur = m;
                        best = p;
                        if (m <= a.kmin) { done = true; break; }
                        limit = m - 1u;
                    }
                }
                // a seed is behind this read: the early-stop rule, as in k_copmem_match_sm
                if (rq == a.k1 - 1u) {
                    rclean += rdirty ? 0u : 1u;
                    rdirty = false;
                }
                rq = (rq + 1u == rper) ? 0u : rq + 1u;
                if (a.early && rclean > limit) break;
            }
            if (best != PGRC_NOT_MATCHED_POS && cur < cin) {
                a.pos[i] = strand ? a.G - (best + a.L) : best;
                a.rc[i] = (uint8_t)strand;
                a.mism[i] = (uint8_t)cur;
                cin = cur;                                       // what the next strand's query has to beat
            }
        }
    }
    for (uint32_t strand = a.first; strand <= a.last; strand++) {
        if (!a.st[strand].counters) continue;
        const uint64_t s0 = wave_sum_u64(n_srch[strand]), s1 = wave_sum_u64(n_cand[strand]), s2 = wave_sum_u64(n_probe[strand]);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.st[strand].counters[0], (unsigned long long)s0);
            atomicAdd(&a.st[strand].counters[1], (unsigned long long)s1);
            atomicAdd(&a.st[strand].counters[2], (unsigned long long)s2);
        }
    }
}

// ----------------------------------------------------------------------------- one query over both strands
// (the kernel: dualkern.h)

#ifdef PGRC_AB_DUAL
int pgrc_copmem_match_dual_r04(pgrc_match_ctx *c);    // tools/variants/dual_r04.hip: round 4's kernel, for in-context A/B runs
#include "../../tools/variants/dualkern_r05a.h"       // round 5's kernel before the VALU diet (PGRC_DUAL_VARIANT=5)
#endif
#ifndef DUAL_WAVES
#define DUAL_WAVES 6             // waves per SIMD the dual kernel is built for at read lengths up to 160 (NW <= 10)
#endif

template <int NW, int WAVES>
static void launch_dual_w(pgrc_match_ctx *c, const DualArgs &a) {
    const uint64_t want = (a.n + MATCH_TPB - 1) / MATCH_TPB;
    // A persistent grid: more blocks than fit (six per CU at 150 bp) simply queue.
    const uint32_t per_cu = 8u;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)c->num_cus * per_cu);
    const bool pos64 = c->G + 256 >= (1ull << 32) || c->opt.force_pos64;
    const bool k28 = a.K == 28;
    if (pos64) {
        if (k28) hipLaunchKernelGGL((k_copmem_match_dual<NW, 7, true, WAVES>), dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
        else hipLaunchKernelGGL((k_copmem_match_dual<NW, 0, true, WAVES>), dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
    } else {
        if (k28) hipLaunchKernelGGL((k_copmem_match_dual<NW, 7, false, WAVES>), dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
        else hipLaunchKernelGGL((k_copmem_match_dual<NW, 0, false, WAVES>), dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
    }
}

template <int NW>
static void launch_dual(pgrc_match_ctx *c, const DualArgs &a) {
#ifdef PGRC_AB_DUAL
    // A/B builds carry round 5's first kernel too (PGRC_DUAL_VARIANT=5) -- 100 / 150 bp reads, K = 28, 32-bit positions only
    if (c->opt.dual_variant == 5 && (NW == 7 || NW == 10) && a.K == 28 && !(c->G + 256 >= (1ull << 32) || c->opt.force_pos64)) {
        const uint64_t want = (a.n + MATCH_TPB - 1) / MATCH_TPB;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)c->num_cus * 8u);
        hipLaunchKernelGGL((k_copmem_match_dual_r05a<(NW == 7 || NW == 10) ? NW : 10, 7, false, 6>), dim3(grid), dim3(MATCH_TPB), 0, c->stream, a);
        return;
    }
#endif
    launch_dual_w<NW, (NW <= 10 ? DUAL_WAVES : 4)>(c, a);
}

// The dual kernel over all reads without N: the ACTIVE index set must describe the RC strand, the alternate set the
// forward strand (api.hip builds them in that order).  The reads with N follow in two ordinary passes (phase 4).
int pgrc_copmem_match_dual(pgrc_match_ctx *c) {
#ifdef PGRC_AB_DUAL
    if (c->opt.dual_variant == 4) return pgrc_copmem_match_dual_r04(c);
#endif
    const uint64_t lo = std::min<uint64_t>(c->range_lo, c->n), rn = std::min<uint64_t>(c->n - lo, c->range_n);   // (a block of a streamed run, or everything)
    if (rn == 0) return PGRC_OK;
    if (c->index_strand != 1 || c->alt_index_strand != 0 || !c->ent_ptr || !c->alt_ent_ptr || !c->d_scr_pos.p || c->head_sh != c->alt_head_sh) {
        c->err = "dual kernel without both indexes";
        return PGRC_E_STATE;
    }
    DualArgs a;
    a.pg[0] = (const uint32_t *)c->pg2[0].p;
    a.pg[1] = (const uint32_t *)c->pg2[1].p;
    a.G = c->G;
    a.reads = c->reads2 + lo;
    a.n = rn;
    a.stride = c->stride;
    a.nflag = (c->n_nreads || c->up_open) ? (const uint8_t *)c->nread_flag.p + lo : nullptr;   // (during an upload the side list is not final yet: the flags are)
    // PGRC_NREAD_INLINE=0: every read with an N goes the byte path (A/B runs, tests)
    a.npos = (a.nflag && c->opt.nread_inline && c->nread_npos.p) ? (const uint32_t *)c->nread_npos.p + lo : nullptr;
    a.head[0] = (const ulonglong2 *)c->alt_head_ptr;
    a.head[1] = (const ulonglong2 *)c->head_ptr;
    a.hsh = c->head_sh;
    a.ent[0] = c->alt_ent_ptr;
    a.ent[1] = c->ent_ptr;
    a.pos = (uint64_t *)c->d_pos.p + lo;
    a.rc = (uint8_t *)c->d_rc.p + lo;
    a.mism = (uint8_t *)c->d_mism.p + lo;
    a.counters = (unsigned long long *)c->d_counters.p + 24;
    a.work = (unsigned long long *)c->d_counters.p + 18;
    a.redo_flag = (uint8_t *)c->d_scr_flag.p + lo;     // (zeroed by the caller; the screen's own use of it is another schedule)
    a.chunk = pgrc_match_chunk(c, rn);
    a.L = c->prm.read_len;
    a.K = (uint32_t)c->cp.K;
    a.k1 = (uint32_t)c->cp.k1;
    a.k2 = (uint32_t)c->cp.k2;
    a.mask = c->cp.hash_size - 1;
    a.kmax = c->prm.max_mismatches;
    switch (c->nw) {
#define CASE_NW(N) case N: launch_dual<N>(c, a); break;
        CASE_NW(2) CASE_NW(3) CASE_NW(4) CASE_NW(5) CASE_NW(6) CASE_NW(7) CASE_NW(8) CASE_NW(9)
        CASE_NW(10) CASE_NW(11) CASE_NW(12) CASE_NW(13) CASE_NW(14) CASE_NW(15) CASE_NW(16)
#undef CASE_NW
    default:
        c->err = "unsupported read length for mode c";
        return PGRC_E_PARAM;
    }
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

template <int NW>
static void launch_match(pgrc_match_ctx *c, const MatchArgs &a) {
    // persistent grid: what the chip can hold (8 blocks of 4 waves per CU is the register/LDS limit at most)
    const uint64_t want = (a.n + MATCH_TPB - 1) / MATCH_TPB;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)c->num_cus * 8u);
    const uint32_t dyn_lds = 0u;
    const bool pos64 = c->G + 256 >= (1ull << 32) || c->opt.force_pos64;   // (PGRC_FORCE_POS64: the 64-bit-position kernel on a small text)
    const bool k28 = a.K == 28;                   // the default seed: compile-time hash loop
    // PGRC_MATCH_STAGE=0: every refilling lane loads its own read (A/B knob; staged is 1.8 % faster on C3, same context)
    const bool staged = c->opt.match_stage;
#define PGRC_LAUNCH_MATCH(KQ_, P64_)                                                                                              \
    do {                                                                                                                          \
        if (staged) hipLaunchKernelGGL((k_copmem_match_sm<NW, KQ_, P64_, MATCH_STAGE>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a); \
        else hipLaunchKernelGGL((k_copmem_match_sm<NW, KQ_, P64_, 0>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a);       \
    } while (0)
    if (pos64) {
        if (k28) PGRC_LAUNCH_MATCH(7, true);
        else PGRC_LAUNCH_MATCH(0, true);
    } else {
        if (k28) PGRC_LAUNCH_MATCH(7, false);
        else PGRC_LAUNCH_MATCH(0, false);
    }
#undef PGRC_LAUNCH_MATCH
}

// The reads with N through the strands first .. last (k_copmem_match_n), on the side stream: it starts when what the main
// stream holds so far is done (the indexes; a streamed run: when c->n_after says so) and the caller joins it later
// (pgrc_copmem_join_nreads).  Every strand asked for must be one of the two index sets of the context.  (Launched BEFORE the
// dual kernel as a small persistent grid it still only starts when that kernel's blocks leave: profiles/r04_nread_beside_ab.txt.)
int pgrc_copmem_match_nreads(pgrc_match_ctx *c, int first, int last, bool only_many) {
    if (!c->n_nreads) return PGRC_OK;
    if (!c->opt.nread_inline) only_many = false;
    if (only_many && !c->n_many) return PGRC_OK;                        // (the dual kernel takes every read with at most 4 N's)
    NReadArgs a;
    a.skip_flag = only_many ? (const uint8_t *)c->nread_flag.p : nullptr;
    for (int s = first; s <= last; s++) {
        const bool act = c->index_strand == s && c->ent_ptr, alt = !act && c->alt_index_strand == s && c->alt_ent_ptr;
        if (!act && !alt) { c->err = "reads with N: no index of that strand"; return PGRC_E_STATE; }
        a.st[s].pg = (const uint32_t *)c->pg2[s].p;
        a.st[s].head = act ? c->head_ptr : c->alt_head_ptr;
        a.st[s].hsh = act ? c->head_sh : c->alt_head_sh;
        a.st[s].ent = act ? c->ent_ptr : c->alt_ent_ptr;
        a.st[s].counters = (unsigned long long *)c->d_counters.p + 8 * s;
    }
    a.G = c->G;
    a.pos = (uint64_t *)c->d_pos.p;
    a.rc = (uint8_t *)c->d_rc.p;
    a.mism = (uint8_t *)c->d_mism.p;
    a.nidx = (const uint32_t *)c->nread_idx.p;
    a.nascii = (const uint8_t *)c->nread_ascii.p;
    a.nn = c->n_nreads;
    a.L = c->prm.read_len;
    a.K = (uint32_t)c->cp.K;
    a.k1 = (uint32_t)c->cp.k1;
    a.k2 = (uint32_t)c->cp.k2;
    a.mask = c->cp.hash_size - 1;
    a.kmax = c->prm.max_mismatches;
    a.kmin = c->prm.min_mismatches;
    a.early = c->opt.early_stop ? 1u : 0u;
    a.first = (uint32_t)first;
    a.last = (uint32_t)last;
    if (!c->side_stream) {   // stream and both events, or nothing
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e0, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
        if (he != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            if (st) (void)hipStreamDestroy(st);
            c->err = std::string("side stream: ") + hipGetErrorString(he);
            return pgrc_hip_code(he);
        }
        c->side_stream = st;
        c->side_ev[0] = e0;
        c->side_ev[1] = e1;
    }
    if (c->n_after) {                                                   // (a streamed run: not behind the blocks still queued)
        HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->n_after, 0));
    } else {
        HIP_TRY(c, hipEventRecord(c->side_ev[0], c->stream));           // the index (and the previous pass) are complete
        HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->side_ev[0], 0));
    }
    const uint64_t tiles = (c->n_nreads + NREAD_TPB - 1) / NREAD_TPB;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(tiles, (uint64_t)c->num_cus * 64u);   // whatever fits
    const uint32_t lds = 2u * ((uint32_t)c->nw + 1u) * NREAD_TPB * (uint32_t)sizeof(uint32_t);
    hipLaunchKernelGGL(k_copmem_match_n, dim3(grid), dim3(NREAD_TPB), lds, c->side_stream, a);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->side_ev[1], c->side_stream));
    return PGRC_OK;
}

// the main stream goes on when the reads with N are done
int pgrc_copmem_join_nreads(pgrc_match_ctx *c) {
    if (c->n_nreads && c->side_stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_ev[1], 0));
    return PGRC_OK;
}

int pgrc_copmem_match_pass(pgrc_match_ctx *c, int strand) { return pgrc_copmem_match_phase(c, strand, 0); }

// phase: 0 = a plain pass; 1 / 2 = the screen and the flag-honouring forward pass of the screened schedule (kernel comment);
// 4 = after the dual kernel: only the reads with N (the byte-path kernel), in the strand order of the two passes
int pgrc_copmem_match_phase(pgrc_match_ctx *c, int strand, int phase) {
    const uint64_t lo = std::min<uint64_t>(c->range_lo, c->n), rn = std::min<uint64_t>(c->n - lo, c->range_n);   // (a block of a streamed run, or everything)
    if (rn == 0) return PGRC_OK;
    MatchArgs a;
    a.pg = (const uint32_t *)c->pg2[strand].p;
    a.G = c->G;
    a.reads = c->reads2 + lo;
    a.n = rn;
    a.stride = c->stride;
    a.nflag = (c->n_nreads || c->up_open) ? (const uint8_t *)c->nread_flag.p + lo : nullptr;   // (during an upload the side list is not final yet: the flags are)
    a.head = (const ulonglong2 *)c->head_ptr;
    a.hsh = c->head_sh;
    a.ent = c->ent_ptr;
    a.pos = (uint64_t *)c->d_pos.p + lo;
    a.rc = (uint8_t *)c->d_rc.p + lo;
    a.mism = (uint8_t *)c->d_mism.p + lo;
    a.counters = (unsigned long long *)c->d_counters.p + (phase == 1 ? 24 : 8 * strand);   // the screen counts apart
    a.work = (unsigned long long *)c->d_counters.p + (phase == 1 ? 18 : 16 + strand);
    a.phase = (uint32_t)phase;
    a.scr_pos = c->d_scr_pos.p ? (uint64_t *)c->d_scr_pos.p + lo : nullptr;
    a.scr_flag = c->d_scr_flag.p ? (uint8_t *)c->d_scr_flag.p + lo : nullptr;
    if (phase && (!a.scr_flag || !a.scr_pos)) { c->err = "screened schedule without its buffers"; return PGRC_E_STATE; }

    a.L = c->prm.read_len;
    a.K = (uint32_t)c->cp.K;
    a.k2 = (uint32_t)c->cp.k2;
    a.mask = c->cp.hash_size - 1;
    a.kmax = c->prm.max_mismatches;
    a.kmin = c->prm.min_mismatches;
    a.strand = (uint32_t)strand;
    a.k1 = (uint32_t)c->cp.k1;
    a.chunk = pgrc_match_chunk(c, rn);
    a.early = c->opt.early_stop ? 1u : 0u;               // PGRC_EARLY_STOP=0: every read probes all its seeds (A/B runs, tests)
    // The reads with N (a percent or two, one lane each, latency-bound) start first on a side stream and run beside the
    // main kernel, whose persistent blocks simply take the remaining slots; the two kernels write disjoint reads.
    // (the screen only flags reads; reads with N are never flagged.  The N kernel walks the side list, whose indexes
    //  count from read 0: it runs on whole-set launches only, never on a block of a streamed run)
    const bool with_n = c->n_nreads && phase != 1 && !c->range_skip_n && lo == 0 && rn == c->n;
    if (with_n) {
        const int ne = pgrc_copmem_match_nreads(c, strand, strand, false);
        if (ne) return ne;
    }
    if (phase != 4)                  // (phase 4: the dual kernel has done every read without N)
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
    if (with_n) return pgrc_copmem_join_nreads(c);   // the pass ends when both kernels have
    return PGRC_OK;
}
