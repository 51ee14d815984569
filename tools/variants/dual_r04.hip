// dual_r04.hip -- A/B builds only (make -C pgrc_amd/csrc AB_DUAL=1): round 4's dual kernel as it was at commit 68e78cd, so that
// the round-5 kernel can be measured against it IN ONE CONTEXT (PGRC_DUAL_VARIANT=4 selects it; tools/ab_match.py).  Not part of
// the product library.  Restates CopMEMMatcher::processApproxMatchQueryTight (matching/copmem/CopMEMMatcher.cpp:483-566) as
// copmem.hip does.
#include <algorithm>

#include "ctx.h"
#include "devutil.h"
#include "headfmt.h"
#include "matchdev.h"

static uint32_t pgrc_match_chunk_r04(const pgrc_match_ctx *c, uint64_t n) {
    const uint64_t waves = (uint64_t)c->num_cus * 20u;
    uint32_t chunk = MATCH_CHUNK;
    while (chunk > 64u && n / chunk < waves * 8u) chunk >>= 1;
    return chunk;
}

struct DualArgsR04 {
    const uint32_t *pg[2];        // packed text, forward and reverse complement
    uint64_t G;
    const uint32_t *reads;
    uint64_t n, stride;
    const uint8_t *nflag;         // reads with N: 1 = the byte path of the ordinary passes, 3 = taken here, its N positions in npos
    const uint32_t *npos;         // (ctx.h nread_npos; nullptr: every flagged read goes the byte path)
    const ulonglong2 *head[2];    // head of bucket h of strand x at head[x][head_slot(h, hsh)] (headfmt.h); the pair table: the two
    uint32_t hsh;                 // heads of a bucket number share a line (one line request and one translation for both gathers)
    const uint64_t *ent[2];
    uint64_t *pos;
    uint8_t *rc;
    uint8_t *mism;
    unsigned long long *counters; // [0] searched [1] candidates [2] heads probed [3] entry fetches [4] verifies [5] redo [6] seeds probed
    unsigned long long *work;
    uint8_t *redo_flag;           // per read: 2 = done again in the reference's order (F_SEQ); introspection only
    uint32_t L, K, k1, k2, mask, kmax;
    uint32_t spec;                // 0, or the small limit + 1 every read first tries (speculative first attempt)
    uint32_t redo_above;          // a bucket of more entries than this, opened while U > budget, sends the read back (4; 0 = round 3's rule: any bucket)
    uint32_t from_end;            // 1: chunks of reads are handed out from the end of the read set
    uint32_t chunk;               // reads a wave reserves per visit to the work counter (pgrc_match_chunk)
};

template <int NW, int KQ, bool POS64>
#ifndef DUAL_WAVES_PER_EU
#define DUAL_WAVES_PER_EU 5      // (experiments: tools/variants.sh)
#endif
__global__ void __launch_bounds__(MATCH_TPB) __attribute__((amdgpu_waves_per_eu(NW <= 10 ? DUAL_WAVES_PER_EU : 4)))   // (97 registers without the hint: 4 waves)
k_copmem_match_dual_r04(const DualArgsR04 a) {
    typedef typename std::conditional<POS64, uint64_t, uint32_t>::type pos_t;
    constexpr pos_t POS_NONE = (pos_t)~(pos_t)0;
    constexpr uint32_t EPOCH_BITS = POS64 ? 12u : 15u;   // one bit less than the single-strand kernel: the strand is part of the tag
    constexpr int SW = 32;           // (round 4 staged 32 reads per wave)
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t fpm_tab[SM_MAX_SEEDS];
    __shared__ uint2 vcache[VC_SLOTS][MATCH_TPB];
    __shared__ uint32_t rd_lds[NW][MATCH_TPB];
    __shared__ uint32_t stg[MATCH_TPB / 64][NW][SW];
    __shared__ uint8_t stg_c[MATCH_TPB / 64][SW], stg_f[MATCH_TPB / 64][SW];
    __shared__ uint32_t stg_n[MATCH_TPB / 64][SW];
    __shared__ ulonglong2 hdR_lds[MATCH_TPB];    // the RC head of a lane's current seed, waiting for the forward bucket to finish
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t wbeg = 0, wend = 0, wnext = 0;
    hash_lut_init(lut);
    const int H = ((int)a.L / 8) * 8;
    const uint32_t nseeds = (a.L - a.K) / a.k2 + 1;
    for (uint32_t t = threadIdx.x; t < nseeds && t < SM_MAX_SEEDS; t += blockDim.x)
        fpm_tab[t] = fp_head_mask(a.K, t * a.k2, (uint32_t)H);
#pragma unroll
    for (int k = 0; k < VC_SLOTS; k++) vcache[k][threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_search = 0, n_cand = 0, n_probe = 0, n_ent = 0, n_ver = 0, n_redo = 0, n_seed = 0;
    uint32_t sh[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) sh[k] = 0u;
    const uint32_t budget = (a.L + 1u - a.K) / a.k2;
    const uint32_t sbits = 2u * a.k2;
    const uint32_t rper = (a.K + a.k1 * a.k2 - 1u) / (a.k1 * a.k2) * a.k1;

    enum { M_PROBE = 0, M_ENTRY = 1, M_VERIFY = 2, M_NEED = 3, M_ADV = 4, M_DEAD = 5 };
    // F_SEQ: the falses bound ran out: the lane does this read again in the reference's order, right here (forward query
    // with its real falses count and bucket truncation, then -- F_SEQ1 -- the RC query from the forward result)
    // F_SPEC (round 4): the read's first attempt, with both strands' limits cut to a.spec - 1 (see the launcher)
    enum { F_ACT0 = 1, F_ACT1 = 2, F_FOUND0 = 4, F_FOUND1 = 8, F_DIRTY0 = 16, F_DIRTY1 = 32, F_REDO = 64, F_FWDEXACT = 128,
           F_SEQ = 256, F_SEQ1 = 512, F_SPEC = 1024 };
    uint32_t mode = M_NEED;
    uint32_t idx = 0, cin = 0, epoch = 0;
    uint32_t npw = 0xFFFFFFFFu;   // the read's N positions, one per byte (0xFF = none): a read with 1-4 N's (hash_fp_window_n)
    uint32_t cnext = 0, cend = 0;
    uint32_t si = 0, rq = 0;
    // per strand (0 forward, 1 RC): own limit (-1 once an exact alignment is accepted), best count and position,
    // bound on the falses of any run, clean rounds
    int lim0 = 0, lim1 = 0, L0 = 0;
    uint32_t cur0 = 0, cur1 = 0, U0 = 0, U1 = 0, rcl0 = 0, rcl1 = 0, fl = 0;
    pos_t best0 = POS_NONE, best1 = POS_NONE;
    uint32_t x = 0;               // the strand whose bucket is being gone through
    pos_t lo = 0;
    uint32_t nb = 0, j = 0, fp_read = 0;
    pos_t cand_p = 0;
    uint64_t pend_e = 0;
    bool has_pend = false;
    constexpr int PWN = ((NW + 1 + 3) / 4) * 4;

    // the limit a candidate of strand s is judged against: its own, capped by what the other strand has found
    auto eff = [&](uint32_t s) -> int {
        if (fl & F_SEQ) return s == 0u ? lim0 : lim1;                // the reference's order: no coupling
        if (s == 0u) return (fl & F_FOUND1) ? min(lim0, (int)cur1) : lim0;
        return (fl & F_FOUND0) ? min(lim1, (int)cur0 - 1) : lim1;
    };

    for (;;) {
        // ---- refill (as in k_copmem_match_sm, staged)
        const unsigned long long need = __ballot(mode == M_NEED);
        if (need) {
            if (cnext == cend) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(a.work, (unsigned long long)a.chunk);
                base = __shfl(base, 0, 64);
                const uint32_t lo_ = (uint32_t)min((uint64_t)base, a.n), hi_ = (uint32_t)min((uint64_t)base + a.chunk, a.n);
                // a.from_end: the chunks are handed out from the END of the read set (each still walked upwards): PgRC's sum set
                // ends with the N set, whose reads rarely match exactly and probe five times the buckets of an average read --
                // taken last they are what the last waves still work on when the others have run dry
                cnext = __builtin_amdgcn_readfirstlane(a.from_end ? (uint32_t)a.n - hi_ : lo_);
                cend = __builtin_amdgcn_readfirstlane(a.from_end ? (uint32_t)a.n - lo_ : hi_);
            }
            if (wnext == wend && cnext != cend) {
                const uint32_t nst = min((uint32_t)SW, cend - cnext);
                wbeg = wnext = cnext;
                wend = cnext = __builtin_amdgcn_readfirstlane(cnext + nst);
                if (lane < nst) {
#pragma unroll
                    for (int k = 0; k < NW; k++) stg[wv][k][lane] = a.reads[(uint64_t)k * a.stride + wbeg + lane];
                    stg_c[wv][lane] = a.mism[wbeg + lane];
                    stg_f[wv][lane] = a.nflag ? a.nflag[wbeg + lane] : (uint8_t)0;
                    stg_n[wv][lane] = a.npos ? a.npos[wbeg + lane] : 0xFFFFFFFFu;
                }
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            const uint32_t avail = wend - wnext;
            if (avail == 0) {
                if (mode == M_NEED) mode = M_DEAD;
            } else {
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                const uint32_t take = min((uint32_t)__popcll(need), avail);
                bool started = false;
                if (mode == M_NEED && rank < take) {
                    const uint32_t sj = wnext - wbeg + rank;
                    idx = wbeg + sj;
                    cin = stg_c[wv][sj];
                    const uint32_t nfl = stg_f[wv][sj];
                    if ((nfl == 0u || (nfl == 3u && a.npos)) && cin != 0u) {   // ReadsMatchers.cpp:430 with min_mismatches == 0
                        npw = nfl ? stg_n[wv][sj] : 0xFFFFFFFFu;
#pragma unroll
                        for (int k = 0; k < NW; k++) rd_lds[k][threadIdx.x] = sh[k] = stg[wv][k][sj];
                        L0 = (cin < a.kmax) ? (int)cin - 1 : (int)a.kmax;   // :488-489
                        const bool spec = a.spec && L0 >= (int)a.spec;       // first with the small limit a.spec - 1
                        lim0 = lim1 = spec ? (int)a.spec - 1 : L0;
                        cur0 = cur1 = cin;
                        best0 = best1 = POS_NONE;
                        U0 = U1 = 0;
                        rcl0 = rcl1 = 0;
                        fl = F_ACT0 | F_ACT1 | (spec ? (uint32_t)F_SPEC : 0u);
                        si = 0;
                        rq = 0;
                        has_pend = false;
                        epoch = (epoch + 1u) & ((1u << EPOCH_BITS) - 1u);
                        if (epoch == 0) {
#pragma unroll
                            for (int k = 0; k < VC_SLOTS; k++) vcache[k][threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                            epoch = 1;
                        }
                        started = true;
                        mode = M_PROBE;
                    }
                }
                wnext = __builtin_amdgcn_readfirstlane(wnext + take);
                n_search += (uint32_t)__popcll(__ballot(started));
            }
        }
        if (!__any(mode != M_DEAD)) break;

        const uint32_t m0 = mode;
        // ---- this iteration's loads
        ulonglong2 hdF = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
        uint64_t v = 0;
        bool counted_ent = false;
        uint32_t ncand_it = 0, nprobe_it = 0;
        bool n_redo_it = false;
        if (m0 == M_PROBE) {
            uint32_t h;
            if (__any(npw != 0xFFFFFFFFu))                           // (wave-uniform: only waves that hold a read with N's take the patched hash)
                h = hash_fp_window_n<KQ>(sh[0], NW > 1 ? sh[1 % NW] : 0u, NW > 2 ? sh[2 % NW] : 0u, NW > 3 ? sh[3 % NW] : 0u, a.K, lut, &fp_read,
                                         npw, si * a.k2) & a.mask;
            else
                h = hash_fp_window<KQ>(sh[0], NW > 1 ? sh[1 % NW] : 0u, NW > 2 ? sh[2 % NW] : 0u,
                                       NW > 3 ? sh[3 % NW] : 0u, a.K, lut, &fp_read) & a.mask;
            ulonglong2 hr = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
            if (fl & F_ACT0) hdF = a.head[0][head_slot(h, a.hsh)];
            if (fl & F_ACT1) hr = a.head[1][head_slot(h, a.hsh)];
            hdR_lds[threadIdx.x] = hr;
            nprobe_it = ((fl & F_ACT0) ? 1u : 0u) + ((fl & F_ACT1) ? 1u : 0u);
        } else if (m0 == M_ENTRY) {
            if (has_pend) {
                v = pend_e;
                has_pend = false;
            } else {
                const U64x2A8 q = *reinterpret_cast<const U64x2A8 *>((x ? a.ent[1] : a.ent[0]) + lo + j - 1);
                v = q.x;
                pend_e = q.y;
                has_pend = j + 1 < nb;
                counted_ent = true;
            }
        }
        // ---- consume
        uint32_t next = m0;
        bool bdone = false;           // the current strand's bucket is finished
        // a verified alignment of the current strand (head count mh, tail count mt)
        auto judge = [&](uint32_t mh, uint32_t mt, pos_t p) {
            const int m = (int)(mh + mt);
            if (fl & F_SEQ) {                                        // the real count of CopMEMMatcher.cpp:536-551
                const int lm = eff(x);
                const uint32_t u = ((int)mh > lm) ? 1u : (m > lm) ? 2u : 0u;
                if (x == 0u) U0 += u; else U1 += u;
            } else {
                // what any run can count for this candidate: 1 if the head alone exceeds every limit a run can have, or if
                // the tail is clean (then it is a head reject or an acceptance), else 2 (a tail reject is counted twice)
                const uint32_t u = ((int)mh > L0 || mt == 0u) ? 1u : 2u;
                if (x == 0u) U0 += u; else U1 += u;
            }
            if (m > eff(x)) return;
            if (x == 0u) { cur0 = (uint32_t)m; best0 = p; lim0 = m - 1; fl |= F_FOUND0; }
            else { cur1 = (uint32_t)m; best1 = p; lim1 = m - 1; fl |= F_FOUND1; }
            if (m == 0) {                                            // m <= min_mismatches: this strand's query returns
                fl &= ~(x == 0u ? (uint32_t)F_ACT0 : (uint32_t)F_ACT1);
                if (x == 0u) fl |= F_FWDEXACT;                       // ... and the RC pass would skip the read
            }
        };
        // what follows an examined entry
        auto after_entry = [&]() {
            if (fl & F_FWDEXACT) next = M_NEED;
            else if (!(fl & (x == 0u ? (uint32_t)F_ACT0 : (uint32_t)F_ACT1))) bdone = true;   // an exact RC alignment: that query is over
            else if (j < nb) next = M_ENTRY;
            else bdone = true;
        };
        auto take_entry = [&](const uint64_t e) {
            const uint32_t s = si * a.k2;
            const uint64_t sp = e >> PGRC_FP_BITS;
            if ((uint64_t)s <= sp && sp - s + a.L <= a.G) {          // :517-520
                ncand_it++;
                const pos_t p = (pos_t)(sp - s);
                const uint32_t xr = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
                const int fpc = __popc((xr | (xr >> 1)) & fpm_tab[si]);   // a lower bound of the head count
                if (fpc > eff(x)) {
                    const uint32_t u = ((fl & F_SEQ) || fpc > L0) ? 1u : 2u;   // (sequential: a certain head reject)
                    if (x == 0u) U0 += u; else U1 += u;
                } else {
                    const uint2 cv = vcache[((uint32_t)p * 0x9E3779B1u + x) >> (32 - VC_BITS)][threadIdx.x];
                    const bool hit = POS64 ? (cv.x == (uint32_t)p && (cv.y >> 20) == epoch && ((cv.y >> 19) & 1u) == x &&
                                              ((cv.y >> 11) & 0xFFu) == (uint32_t)((uint64_t)p >> 32))
                                           : (cv.x == (uint32_t)p && (cv.y >> 17) == epoch && ((cv.y >> 16) & 1u) == x);
                    if (hit) judge(cv.y & 0xFFu, (cv.y >> 8) & (POS64 ? 0x7u : 0xFFu), p);
                    else {
                        cand_p = p;
                        next = M_VERIFY;
                        return;
                    }
                }
            }
            after_entry();
        };
        // open the bucket of strand x for the current seed
        auto open_bucket = [&](const ulonglong2 hx) {
            const uint32_t cnt = head_count(hx);
            if (!cnt) {
                bdone = true;
                return;
            }
            // some run could have cut THIS bucket to its first 4 entries by now (:510-514 -- all the budget ever does; a bucket
            // of at most 4 is the same bucket whatever the falses count): not decidable this way -> the read again, in the
            // reference's order
            if (!(fl & F_SEQ) && (x == 0u ? U0 : U1) > budget && cnt > a.redo_above) {
                fl |= F_REDO;
                next = M_NEED;
                return;
            }
            nb = cnt;
            if ((fl & F_SEQ) && (x == 0u ? U0 : U1) > budget) nb = min(nb, PGRC_TRUNC_BUCKET);   // :510-514
            if (rq < a.k1 && (cnt >= PGRC_BUCKET_CAP || nb < cnt)) fl |= (x == 0u ? (uint32_t)F_DIRTY0 : (uint32_t)F_DIRTY1);
            has_pend = cnt == 2 && nb > 1;
            pend_e = hx.y;
            lo = (pos_t)(hx.y & W1_BASE_MASK);
            j = 1;
            take_entry(hx.x & ENT_MASK);
        };
        if (m0 == M_VERIFY) {
            uint32_t pw[PWN];
            const uint32_t *src = (x ? a.pg[1] : a.pg[0]) + (cand_p >> 4);   // the text is padded: PWN words are always in bounds
#pragma unroll
            for (int k = 0; k < PWN; k += 4) {
                const u32x4 q = reinterpret_cast<const U32x4A4 *>(src + k)->v;
                pw[k] = q.x; pw[k + 1] = q.y; pw[k + 2] = q.z; pw[k + 3] = q.w;
            }
            const uint32_t b = ((uint32_t)cand_p & 15u) * 2u;
            uint32_t mh = 0, mt = 0;
            if (__any(npw != 0xFFFFFFFFu)) {                         // (wave-uniform) an N of the read equals no text symbol
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    const uint32_t xr = funnel_r(pw[k], pw[k + 1], b) ^ rd_lds[k][threadIdx.x];
                    uint32_t d = (xr | (xr >> 1)) & 0x55555555u;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t q = (npw >> (8 * i)) & 0xFFu;
                        d |= ((q >> 4) == (uint32_t)k) ? 1u << (2u * (q & 15u)) : 0u;   // (none = 0xFF: symbol 255 lies beyond every read)
                    }
                    mh += (uint32_t)__popc(d & sym_mask(k, 0, H));
                    mt += (uint32_t)__popc(d & sym_mask(k, H, (int)a.L));
                }
            } else {
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    const uint32_t tw = funnel_r(pw[k], pw[k + 1], b);
                    const uint32_t rw = rd_lds[k][threadIdx.x];
                    mh += mism2(tw, rw, sym_mask(k, 0, H));
                    mt += mism2(tw, rw, sym_mask(k, H, (int)a.L));
                }
            }
            vcache[((uint32_t)cand_p * 0x9E3779B1u + x) >> (32 - VC_BITS)][threadIdx.x] =
                make_uint2((uint32_t)cand_p, POS64 ? (mh | (mt << 8) | ((uint32_t)((uint64_t)cand_p >> 32) << 11) | (x << 19) | (epoch << 20))
                                                   : (mh | (mt << 8) | (x << 16) | (epoch << 17)));
            judge(mh, mt, cand_p);
            after_entry();
        } else if (m0 == M_ENTRY) {
            j++;
            take_entry(v);
        } else if (m0 == M_PROBE) {
            x = (fl & F_ACT0) ? 0u : 1u;
            open_bucket(x == 0u ? hdF : hdR_lds[threadIdx.x]);
        }
        // the forward bucket is done: the RC head of the same seed waits in LDS
        if (bdone && x == 0u && (fl & F_ACT1)) {
            bdone = false;
            x = 1u;
            open_bucket(hdR_lds[threadIdx.x]);
        }
        if (bdone) next = M_ADV;
        if (next == M_ADV) {                                         // to the next seed
            si++;
            has_pend = false;
            if (rq == a.k1 - 1u) {                                   // a round is behind this read
                rcl0 += (fl & F_DIRTY0) ? 0u : 1u;
                rcl1 += (fl & F_DIRTY1) ? 0u : 1u;
                fl &= ~(uint32_t)(F_DIRTY0 | F_DIRTY1);
            }
            rq = (rq + 1u == rper) ? 0u : rq + 1u;
#pragma unroll
            for (int k = 0; k < NW - 1; k++) sh[k] = funnel_r(sh[k], sh[k + 1], sbits);
            sh[NW - 1] >>= sbits;
            if ((fl & F_ACT0) && (int)rcl0 > eff(0u)) fl &= ~(uint32_t)F_ACT0;   // nothing acceptable is left on that strand
            if ((fl & F_ACT1) && (int)rcl1 > eff(1u)) fl &= ~(uint32_t)F_ACT1;
            next = (si < nseeds && (fl & (F_ACT0 | F_ACT1))) ? M_PROBE : M_NEED;
        }
        // ---- the reference's order for a read whose falses bound ran out: restart it as a forward query, then an RC query
        if (next == M_NEED && m0 <= M_VERIFY && (((fl & F_REDO) != 0u) || ((fl & (F_SEQ | F_SEQ1 | F_FWDEXACT)) == F_SEQ))) {
            const bool second = (fl & F_SEQ) != 0u;                  // the forward query just ended: now the RC query
            const uint32_t c1 = (second && (fl & F_FOUND0)) ? cur0 : cin;   // what the RC query has to beat (:488-489)
            if (!second) {
                lim0 = L0; cur0 = cin; best0 = POS_NONE; U0 = 0; rcl0 = 0;
                fl = F_SEQ | F_ACT0;
                n_redo_it = true;
                a.redo_flag[idx] = 2;
            } else {
                lim1 = (c1 < a.kmax) ? (int)c1 - 1 : (int)a.kmax; cur1 = c1; best1 = POS_NONE; U1 = 0; rcl1 = 0;
                fl = (fl & (F_FOUND0 | F_SEQ)) | F_SEQ1 | F_ACT1;
            }
            si = 0;
            rq = 0;
            has_pend = false;
#pragma unroll
            for (int kk = 0; kk < NW; kk++) sh[kk] = rd_lds[kk][threadIdx.x];
            next = M_PROBE;
        }
        // ---- the first attempt with the small limit found nothing on either strand: the read again with its real limit
        // (what the verify cache holds stays valid: counts do not depend on the limit)
        if (next == M_NEED && m0 <= M_VERIFY && (fl & (F_SPEC | F_FOUND0 | F_FOUND1 | F_SEQ)) == F_SPEC) {
            lim0 = lim1 = L0;
            cur0 = cur1 = cin;
            best0 = best1 = POS_NONE;
            U0 = U1 = 0;
            rcl0 = rcl1 = 0;
            fl = F_ACT0 | F_ACT1;
            si = 0;
            rq = 0;
            has_pend = false;
#pragma unroll
            for (int kk = 0; kk < NW; kk++) sh[kk] = rd_lds[kk][threadIdx.x];
            next = M_PROBE;
        }
        {
            const uint32_t seeds_it = (uint32_t)__popcll(__ballot(nprobe_it >= 1));
            n_seed += seeds_it;
            n_probe += seeds_it + (uint32_t)__popcll(__ballot(nprobe_it >= 2));
        }
        n_ent += (uint32_t)__popcll(__ballot(counted_ent));
        n_ver += (uint32_t)__popcll(__ballot(m0 == M_VERIFY));
        n_cand += (uint32_t)__popcll(__ballot(ncand_it >= 1)) + (uint32_t)__popcll(__ballot(ncand_it >= 2));
        const bool fin = next == M_NEED && m0 <= M_VERIFY;
        if (fin) {
            // the read is finished: forward wins ties, RC must be strictly better (ReadsMatchers.cpp:437-447, both passes)
            if ((fl & F_FOUND1) && !((fl & F_FOUND0) && cur0 <= cur1)) {
                a.pos[idx] = a.G - ((uint64_t)best1 + a.L);
                a.rc[idx] = 1;
                a.mism[idx] = (uint8_t)cur1;
            } else if (fl & F_FOUND0) {
                a.pos[idx] = (uint64_t)best0;
                a.rc[idx] = 0;
                a.mism[idx] = (uint8_t)cur0;
            }
        }
        n_redo += (uint32_t)__popcll(__ballot(n_redo_it));
        mode = next;
    }
    if (a.counters && lane == 0) {
        atomicAdd(&a.counters[0], (unsigned long long)n_search);
        atomicAdd(&a.counters[1], (unsigned long long)n_cand);
        atomicAdd(&a.counters[2], (unsigned long long)n_probe);
        atomicAdd(&a.counters[3], (unsigned long long)n_ent);
        atomicAdd(&a.counters[4], (unsigned long long)n_ver);
        atomicAdd(&a.counters[5], (unsigned long long)n_redo);
        atomicAdd(&a.counters[6], (unsigned long long)n_seed);
    }
}

template <int NW>
static void launch_dual_r04(pgrc_match_ctx *c, const DualArgsR04 &a) {
    const uint64_t want = (a.n + MATCH_TPB - 1) / MATCH_TPB;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)c->num_cus * 8u);
    const bool pos64 = c->G + 256 >= (1ull << 32) || c->opt.force_pos64;
    const bool k28 = a.K == 28;
    const uint32_t dyn_lds = 0u;
    if (pos64) {
        if (k28) hipLaunchKernelGGL((k_copmem_match_dual_r04<NW, 7, true>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a);
        else hipLaunchKernelGGL((k_copmem_match_dual_r04<NW, 0, true>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a);
    } else {
        if (k28) hipLaunchKernelGGL((k_copmem_match_dual_r04<NW, 7, false>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a);
        else hipLaunchKernelGGL((k_copmem_match_dual_r04<NW, 0, false>), dim3(grid), dim3(MATCH_TPB), dyn_lds, c->stream, a);
    }
}

// The dual kernel over all reads without N: the ACTIVE index set must describe the RC strand, the alternate set the
// forward strand (api.hip builds them in that order).  The reads with N follow in two ordinary passes (phase 4).
int pgrc_copmem_match_dual_r04(pgrc_match_ctx *c) {
    const uint64_t lo = std::min<uint64_t>(c->range_lo, c->n), rn = std::min<uint64_t>(c->n - lo, c->range_n);   // (a block of a streamed run, or everything)
    if (rn == 0) return PGRC_OK;
    if (c->index_strand != 1 || c->alt_index_strand != 0 || !c->ent_ptr || !c->alt_ent_ptr || !c->d_scr_pos.p || c->head_sh != c->alt_head_sh) {
        c->err = "dual kernel without both indexes";
        return PGRC_E_STATE;
    }
    DualArgsR04 a;
    a.pg[0] = (const uint32_t *)c->pg2[0].p;
    a.pg[1] = (const uint32_t *)c->pg2[1].p;
    a.G = c->G;
    a.reads = c->reads2 + lo;
    a.n = rn;
    a.stride = c->stride;
    a.nflag = (c->n_nreads || c->up_open) ? (const uint8_t *)c->nread_flag.p + lo : nullptr;   // (during an upload the side list is not final yet: the flags are)
    {
        a.npos = (a.nflag && c->opt.nread_inline && c->nread_npos.p) ? (const uint32_t *)c->nread_npos.p + lo : nullptr;
    }
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
    a.chunk = pgrc_match_chunk_r04(c, rn);
    a.L = c->prm.read_len;
    a.K = (uint32_t)c->cp.K;
    a.k1 = (uint32_t)c->cp.k1;
    a.k2 = (uint32_t)c->cp.k2;
    a.mask = c->cp.hash_size - 1;
    a.kmax = c->prm.max_mismatches;
    {
        a.redo_above = PGRC_TRUNC_BUCKET;
        a.spec = 0u;
        a.from_end = 1u;
    }
    switch (c->nw) {
#define CASE_NW(N) case N: launch_dual_r04<N>(c, a); break;
        CASE_NW(7) CASE_NW(10) CASE_NW(16)
#undef CASE_NW
    default:
        c->err = "the round-4 dual kernel of the A/B build takes 100 / 150 / 250 bp reads only";
        return PGRC_E_PARAM;
    }
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

