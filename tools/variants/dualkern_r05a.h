// dualkern_r05a.h -- A/B builds only (make -C pgrc_amd/csrc AB_DUAL=1): the dual kernel as it was at commit 143e9af (round 5,
// six waves, before the VALU diet), so that the current kernel can be measured against it IN ONE CONTEXT
// (PGRC_DUAL_VARIANT=5, 100 / 150 bp reads, K = 28, 32-bit positions; tools/ab_match.py).  Not part of the product library.
// Restates CopMEMMatcher::processApproxMatchQueryTight (matching/copmem/CopMEMMatcher.cpp:483-566) as dualkern.h does.
#pragma once
#include "dualkern.h"

template <int NW, int KQ, bool POS64, int WAVES>
__global__ void __launch_bounds__(MATCH_TPB) __attribute__((amdgpu_waves_per_eu(WAVES)))
k_copmem_match_dual_r05a(const DualArgs a) {
    typedef typename std::conditional<POS64, uint64_t, uint32_t>::type pos_t;
    constexpr pos_t POS_NONE = (pos_t)~(pos_t)0;
    constexpr uint32_t EPOCH_BITS = POS64 ? 12u : 14u;   // the verify cache's tag: epoch | strand | (POS64: position bits 32..39)
    constexpr int SW = DualStage<WAVES>::SW;
    // window words a seed's hash looks at (K = 28: 56 bits = 2 words), and the read words they are cut from
    constexpr int NWIN = KQ ? (KQ * 8 + 31) / 32 : 4, NSRC = NWIN + 1;
    __shared__ uint32_t lut[PGRC_HASH_LUT_WORDS];
    __shared__ uint32_t fpm_tab[SM_MAX_SEEDS];
    __shared__ uint2 vcache[VC_SLOTS][MATCH_TPB];
    __shared__ uint32_t rd_lds[NW][MATCH_TPB];   // the read: its windows are cut from here, and it is what a verified text window is compared with
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
    uint32_t n_search = 0, n_cand = 0, n_probe = 0, n_ent = 0, n_ver = 0, n_redo = 0, n_seed = 0;   // wave-uniform: SGPRs
    const uint32_t budget = (a.L + 1u - a.K) / a.k2;
    const uint32_t rper = (a.K + a.k1 * a.k2 - 1u) / (a.k1 * a.k2) * a.k1;

    enum { M_PROBE = 0, M_ENTRY = 1, M_VERIFY = 2, M_NEED = 3, M_ADV = 4, M_DEAD = 5 };
    // F_SEQ: the falses bound ran out: the lane does this read again in the reference's order, right here (forward query
    // with its real falses count and bucket truncation, then -- F_SEQ1 -- the RC query from the forward result)
    enum { F_ACT0 = 1, F_ACT1 = 2, F_FOUND0 = 4, F_FOUND1 = 8, F_DIRTY0 = 16, F_DIRTY1 = 32, F_REDO = 64, F_FWDEXACT = 128,
           F_SEQ = 256, F_SEQ1 = 512 };
    // ---- the lane's read.  Five registers hold what round 4 kept in twenty:
    //   sr = si | rq << 8 | rcl0 << 16 | rcl1 << 24      seed index (< 240), seed inside the round period, clean rounds per strand
    //   cc = cur0 | cur1 << 8 | cin << 16 | L0 << 24     best count per strand, the read's count before the run, its starting limit
    //   uu = U0 | U1 << 16                               bound on the falses of any run, per strand (<= 2 x 13 x 240)
    //   ll = (lim0 + 1) | (lim1 + 1) << 9 | epoch << 18  own limit per strand (-1 once an exact alignment is accepted)
    //   jn = j | nb << 4 | x << 8 | has_pend << 9        entry index and size of the bucket being gone through, its strand
    uint32_t mode = M_NEED, idx = 0, fl = 0;
    uint32_t sr = 0, cc = 0, uu = 0, ll = 0, jn = 0;
    uint32_t npw = 0xFFFFFFFFu;   // the read's N positions, one per byte (0xFF = none): a read with 1-4 N's (hash_fp_window_n)
    uint32_t cnext = 0, cend = 0;
    pos_t best0 = POS_NONE, best1 = POS_NONE;
    pos_t lo = 0;
    uint32_t fp_read = 0;
    pos_t cand_p = 0;
    uint64_t pend_e = 0;
    constexpr int PWN = ((NW + 1 + 3) / 4) * 4;
#define DK_SI() (sr & 0xFFu)
#define DK_RQ() ((sr >> 8) & 0xFFu)
#define DK_RCL(s) ((sr >> (16u + 8u * (s))) & 0xFFu)
#define DK_CUR(s) ((cc >> (8u * (s))) & 0xFFu)
#define DK_CIN() ((cc >> 16) & 0xFFu)
#define DK_L0() ((int)(cc >> 24))
#define DK_U(s) ((uu >> (16u * (s))) & 0xFFFFu)
#define DK_LIM(s) ((int)((ll >> (9u * (s))) & 0x1FFu) - 1)
#define DK_EPOCH() (ll >> 18)
#define DK_J() (jn & 15u)
#define DK_NB() ((jn >> 4) & 15u)
#define DK_X() ((jn >> 8) & 1u)
#define DK_PEND() ((jn >> 9) & 1u)

    // the limit a candidate of strand s is judged against: its own, capped by what the other strand has found
    auto eff = [&](uint32_t s) -> int {
        const int own = DK_LIM(s);
        if (fl & F_SEQ) return own;                                  // the reference's order: no coupling
        if (s == 0u) return (fl & F_FOUND1) ? min(own, (int)DK_CUR(1u)) : own;
        return (fl & F_FOUND0) ? min(own, (int)DK_CUR(0u) - 1) : own;
    };
    auto set_lim = [&](uint32_t s, int v) { ll = (ll & ~(0x1FFu << (9u * s))) | ((uint32_t)(v + 1) << (9u * s)); };
    auto set_cur = [&](uint32_t s, uint32_t v) { cc = (cc & ~(0xFFu << (8u * s))) | (v << (8u * s)); };

    for (;;) {
        // ---- refill (as in k_copmem_match_sm, staged)
        const unsigned long long need = __ballot(mode == M_NEED);
        if (need) {
            if (cnext == cend) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(a.work, (unsigned long long)a.chunk);
                base = __shfl(base, 0, 64);
                const uint32_t lo_ = (uint32_t)min((uint64_t)base, a.n), hi_ = (uint32_t)min((uint64_t)base + a.chunk, a.n);
                // the chunks are handed out from the END of the read set (each still walked upwards): PgRC's sum set ends with
                // the N set, whose reads rarely match exactly and probe five times the buckets of an average read -- taken
                // last they are what the last waves still work on when the others have run dry (profiles/r04_from_end_ab.txt)
                cnext = __builtin_amdgcn_readfirstlane((uint32_t)a.n - hi_);
                cend = __builtin_amdgcn_readfirstlane((uint32_t)a.n - lo_);
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
                    const uint32_t cin = stg_c[wv][sj];
                    const uint32_t nfl = stg_f[wv][sj];
                    if ((nfl == 0u || (nfl == 3u && a.npos)) && cin != 0u) {   // ReadsMatchers.cpp:430 with min_mismatches == 0
                        npw = nfl ? stg_n[wv][sj] : 0xFFFFFFFFu;
#pragma unroll
                        for (int k = 0; k < NW; k++) rd_lds[k][threadIdx.x] = stg[wv][k][sj];
                        const uint32_t L0 = (cin < a.kmax) ? cin - 1u : a.kmax;   // :488-489
                        cc = cin | (cin << 8) | (cin << 16) | (L0 << 24);
                        uint32_t epoch = (DK_EPOCH() + 1u) & ((1u << EPOCH_BITS) - 1u);
                        if (epoch == 0) {
#pragma unroll
                            for (int k = 0; k < VC_SLOTS; k++) vcache[k][threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                            epoch = 1;
                        }
                        ll = (L0 + 1u) | ((L0 + 1u) << 9) | (epoch << 18);
                        best0 = best1 = POS_NONE;
                        uu = 0;
                        sr = 0;
                        jn = 0;
                        fl = F_ACT0 | F_ACT1;
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
            // the seed's window, cut out of the read's LDS copy: symbols s .. s + K - 1
            const uint32_t s = DK_SI() * a.k2, q = s >> 4, shb = (s & 15u) * 2u;
            uint32_t r[NSRC], w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NSRC; k++) {
                const uint32_t row = min(q + (uint32_t)k, (uint32_t)NW - 1u);
                const uint32_t val = rd_lds[row][threadIdx.x];
                r[k] = (q + (uint32_t)k < (uint32_t)NW) ? val : 0u;
            }
#pragma unroll
            for (int k = 0; k < NWIN; k++) w[k] = funnel_r(r[k], r[k + 1], shb);
            uint32_t h;
            if (__any(npw != 0xFFFFFFFFu))                           // (wave-uniform: only waves that hold a read with N's take the patched hash)
                h = hash_fp_window_n<KQ>(w[0], w[1], w[2], w[3], a.K, lut, &fp_read, npw, s) & a.mask;
            else
                h = hash_fp_window<KQ>(w[0], w[1], w[2], w[3], a.K, lut, &fp_read) & a.mask;
            ulonglong2 hr = make_ulonglong2(HEAD_EMPTY, HEAD_EMPTY);
            if (fl & F_ACT0) hdF = a.head[0][head_slot(h, a.hsh)];
            if (fl & F_ACT1) hr = a.head[1][head_slot(h, a.hsh)];
            hdR_lds[threadIdx.x] = hr;
            nprobe_it = ((fl & F_ACT0) ? 1u : 0u) + ((fl & F_ACT1) ? 1u : 0u);
        } else if (m0 == M_ENTRY) {
            if (DK_PEND()) {
                v = pend_e;
                jn &= ~(1u << 9);
            } else {
                const uint32_t j = DK_J(), nb = DK_NB();
                const U64x2A8 q = *reinterpret_cast<const U64x2A8 *>((DK_X() ? a.ent[1] : a.ent[0]) + lo + j - 1);
                v = q.x;
                pend_e = q.y;
                jn = (jn & ~(1u << 9)) | ((j + 1 < nb ? 1u : 0u) << 9);
                counted_ent = true;
            }
        }
        // ---- consume
        uint32_t next = m0;
        bool bdone = false;           // the current strand's bucket is finished
        // a verified alignment of the current strand (head count mh, tail count mt)
        auto judge = [&](uint32_t mh, uint32_t mt, pos_t p) {
            const uint32_t x = DK_X();
            const int m = (int)(mh + mt);
            uint32_t u;
            if (fl & F_SEQ) {                                        // the real count of CopMEMMatcher.cpp:536-551
                const int lm = eff(x);
                u = ((int)mh > lm) ? 1u : (m > lm) ? 2u : 0u;
            } else {
                // what any run can count for this candidate: 1 if the head alone exceeds every limit a run can have, or if
                // the tail is clean (then it is a head reject or an acceptance), else 2 (a tail reject is counted twice)
                u = ((int)mh > DK_L0() || mt == 0u) ? 1u : 2u;
            }
            uu += u << (16u * x);
            if (m > eff(x)) return;
            set_cur(x, (uint32_t)m);
            set_lim(x, m - 1);
            if (x == 0u) { best0 = p; fl |= F_FOUND0; }
            else { best1 = p; fl |= F_FOUND1; }
            if (m == 0) {                                            // m <= min_mismatches: this strand's query returns
                fl &= ~(x == 0u ? (uint32_t)F_ACT0 : (uint32_t)F_ACT1);
                if (x == 0u) fl |= F_FWDEXACT;                       // ... and the RC pass would skip the read
            }
        };
        // what follows an examined entry
        auto after_entry = [&]() {
            if (fl & F_FWDEXACT) next = M_NEED;
            else if (!(fl & (DK_X() == 0u ? (uint32_t)F_ACT0 : (uint32_t)F_ACT1))) bdone = true;   // an exact RC alignment: that query is over
            else if (DK_J() < DK_NB()) next = M_ENTRY;
            else bdone = true;
        };
        auto take_entry = [&](const uint64_t e) {
            const uint32_t x = DK_X(), si = DK_SI();
            const uint32_t s = si * a.k2;
            const uint64_t sp = e >> PGRC_FP_BITS;
            if ((uint64_t)s <= sp && sp - s + a.L <= a.G) {          // :517-520
                ncand_it++;
                const pos_t p = (pos_t)(sp - s);
                const uint32_t xr = ((uint32_t)e ^ fp_read) & ((1u << PGRC_FP_BITS) - 1u);
                const int fpc = __popc((xr | (xr >> 1)) & fpm_tab[si]);   // a lower bound of the head count
                if (fpc > eff(x)) {
                    const uint32_t u = ((fl & F_SEQ) || fpc > DK_L0()) ? 1u : 2u;   // (sequential: a certain head reject)
                    uu += u << (16u * x);
                } else {
                    const uint32_t epoch = DK_EPOCH();
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
            const uint32_t x = DK_X();
            const uint32_t cnt = head_count(hx);
            if (!cnt) {
                bdone = true;
                return;
            }
            // some run could have cut THIS bucket to its first 4 entries by now (:510-514 -- all the budget ever does; a bucket
            // of at most 4 is the same bucket whatever the falses count): not decidable this way -> the read again, in the
            // reference's order
            if (!(fl & F_SEQ) && DK_U(x) > budget && cnt > PGRC_TRUNC_BUCKET) {
                fl |= F_REDO;
                next = M_NEED;
                return;
            }
            uint32_t nb = cnt;
            if ((fl & F_SEQ) && DK_U(x) > budget) nb = min(nb, PGRC_TRUNC_BUCKET);   // :510-514
            if (DK_RQ() < a.k1 && (cnt >= PGRC_BUCKET_CAP || nb < cnt)) fl |= (x == 0u ? (uint32_t)F_DIRTY0 : (uint32_t)F_DIRTY1);
            pend_e = hx.y;
            lo = (pos_t)(hx.y & W1_BASE_MASK);
            jn = 1u | (nb << 4) | (x << 8) | ((cnt == 2 && nb > 1) ? 1u << 9 : 0u);    // j = 1; entry 1 of a two-entry bucket sits in the head
            take_entry(hx.x & ENT_MASK);
        };
        if (m0 == M_VERIFY) {
            const uint32_t x = DK_X();
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
            const uint32_t epoch = DK_EPOCH();
            vcache[((uint32_t)cand_p * 0x9E3779B1u + x) >> (32 - VC_BITS)][threadIdx.x] =
                make_uint2((uint32_t)cand_p, POS64 ? (mh | (mt << 8) | ((uint32_t)((uint64_t)cand_p >> 32) << 11) | (x << 19) | (epoch << 20))
                                                   : (mh | (mt << 8) | (x << 16) | (epoch << 17)));
            judge(mh, mt, cand_p);
            after_entry();
        } else if (m0 == M_ENTRY) {
            jn++;                                                    // j++ (j <= 13: no carry into nb)
            take_entry(v);
        } else if (m0 == M_PROBE) {
            const uint32_t x = (fl & F_ACT0) ? 0u : 1u;
            jn = x << 8;
            open_bucket(x == 0u ? hdF : hdR_lds[threadIdx.x]);
        }
        // the forward bucket is done: the RC head of the same seed waits in LDS
        if (bdone && DK_X() == 0u && (fl & F_ACT1)) {
            bdone = false;
            jn = 1u << 8;
            open_bucket(hdR_lds[threadIdx.x]);
        }
        if (bdone) next = M_ADV;
        if (next == M_ADV) {                                         // to the next seed
            const uint32_t rq = DK_RQ();
            uint32_t rcl0 = DK_RCL(0u), rcl1 = DK_RCL(1u);
            if (rq == a.k1 - 1u) {                                   // a round is behind this read
                rcl0 += (fl & F_DIRTY0) ? 0u : 1u;
                rcl1 += (fl & F_DIRTY1) ? 0u : 1u;
                fl &= ~(uint32_t)(F_DIRTY0 | F_DIRTY1);
            }
            const uint32_t si = DK_SI() + 1u;
            sr = si | ((rq + 1u == rper ? 0u : rq + 1u) << 8) | (rcl0 << 16) | (rcl1 << 24);
            jn &= ~(1u << 9);
            if ((fl & F_ACT0) && (int)rcl0 > eff(0u)) fl &= ~(uint32_t)F_ACT0;   // nothing acceptable is left on that strand
            if ((fl & F_ACT1) && (int)rcl1 > eff(1u)) fl &= ~(uint32_t)F_ACT1;
            next = (si < nseeds && (fl & (F_ACT0 | F_ACT1))) ? M_PROBE : M_NEED;
        }
        // ---- the reference's order for a read whose falses bound ran out: restart it as a forward query, then an RC query
        if (next == M_NEED && m0 <= M_VERIFY && (((fl & F_REDO) != 0u) || ((fl & (F_SEQ | F_SEQ1 | F_FWDEXACT)) == F_SEQ))) {
            const bool second = (fl & F_SEQ) != 0u;                  // the forward query just ended: now the RC query
            const uint32_t cin = DK_CIN();
            if (!second) {
                set_lim(0u, DK_L0());
                set_cur(0u, cin);
                best0 = POS_NONE;
                uu &= 0xFFFF0000u;
                fl = F_SEQ | F_ACT0;
                n_redo_it = true;
                a.redo_flag[idx] = 2;
            } else {
                const uint32_t c1 = (fl & F_FOUND0) ? DK_CUR(0u) : cin;   // what the RC query has to beat (:488-489)
                set_lim(1u, (c1 < a.kmax) ? (int)c1 - 1 : (int)a.kmax);
                set_cur(1u, c1);
                best1 = POS_NONE;
                uu &= 0x0000FFFFu;
                fl = (fl & (F_FOUND0 | F_SEQ)) | F_SEQ1 | F_ACT1;
            }
            sr = 0;                                                  // first seed again; clean rounds of both strands reset (only the active one is read)
            jn &= ~(1u << 9);
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
            const uint32_t cur0 = DK_CUR(0u), cur1 = DK_CUR(1u);
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
#undef DK_SI
#undef DK_RQ
#undef DK_RCL
#undef DK_CUR
#undef DK_CIN
#undef DK_L0
#undef DK_U
#undef DK_LIM
#undef DK_EPOCH
#undef DK_J
#undef DK_NB
#undef DK_X
#undef DK_PEND
}
