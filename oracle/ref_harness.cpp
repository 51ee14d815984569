// ref_harness.cpp -- drives the REAL reference (compiled from /root/reference by
// oracle/Makefile into oracle/_ref/libpgrc_ref.so) behind a tiny C ABI so that
// tests/ can pin the CPU restatement (oracle/pgrc_oracle.c) and the HIP path
// against it, and so that tests/golden/make_golden.py can generate fixtures.
//
// TEST INFRASTRUCTURE ONLY.  No reference source is copied or modified: the
// matchers' protected result fields are reached by subclassing
// (ReadsMatchers.h:32-43,115-116) and CopMEMMatcher's private index by the
// access-specifier trick in THIS translation unit only.
#include <bits/stdc++.h>
#include <omp.h>

#define private public
#define protected public
#include "matching/copmem/CopMEMMatcher.h"
#undef private
#undef protected

#include "matching/ReadsMatchers.h"
#include "readsset/PackedConstantLengthReadsSet.h"
#include "utils/helper.h"
#include "readsset/DividedPCLReadsSets.h"
#include "readsset/persistance/ReadsSetPersistence.h"
#ifdef PGRC_WITH_HIP_ADAPTER
#include "HipDividedReadsSets.h"
#include "HipReadsMatcher.h"
#include "HipTextMatcher.h"
#include "matching/SimplePgMatcher.h"
#include "pgrc/pgrc-decoder.h"
#include "pgrc/pgrc-encoder.h"
#endif

using namespace PgTools;
using namespace PgReadsSet;

namespace {

template <class M>
struct Probe : M {
    using M::M;
    using M::matchedReadsCount;
    using M::readMatchPos;
    using M::readMatchRC;
};

template <class M>
struct ApproxProbe : M {
    using M::M;
    using M::matchedCountPerMismatches;
    using M::matchedReadsCount;
    using M::readMatchPos;
    using M::readMatchRC;
    using M::readMismatchesCount;
    using M::updateEntry;
    using M::betterMatchCount;
    using M::falseMatchCount;
};

struct Silence {
    std::ostream *old_log;
    std::ios::iostate old_state;
    Silence() : old_log(PgHelpers::logout), old_state(std::cout.rdstate()) {
        if (getenv("PGRC_REF_VERBOSE")) return;   // debugging aid: keep the reference's own log
        PgHelpers::logout = &null_stream;
        std::cout.setstate(std::ios::failbit);
    }
    ~Silence() {
        PgHelpers::logout = old_log;
        std::cout.clear(old_state);
    }
};

struct ReadsHolder {
    PackedConstantLengthReadsSet *lq = nullptr, *nset = nullptr;
    SumOfConstantLengthReadsSets *sum = nullptr;
    ConstantLengthReadsSetInterface *iface = nullptr;
    ReadsHolder(const char *reads, uint64_t n_lq, uint64_t n_n, uint32_t L) {
        lq = new PackedConstantLengthReadsSet(L, "ACGT", 4);
        for (uint64_t i = 0; i < n_lq; i++) lq->addRead(reads + i * L, L);
        iface = lq;
        if (n_n) {
            // the N set is packed over ACGNT (readsset/DividedPCLReadsSets.cpp:16-19)
            nset = new PackedConstantLengthReadsSet(L, "ACGNT", 5);
            for (uint64_t i = 0; i < n_n; i++) nset->addRead(reads + (n_lq + i) * L, L);
            if (n_lq) {
                sum = new SumOfConstantLengthReadsSets(lq, nset); // pgrc-encoder.cpp:349-352
                iface = sum;
            } else {
                iface = nset; // one ACGNT-packed set holding every read (the nReadsLQ configuration)
            }
        }
    }
    ~ReadsHolder() {
        delete sum;
        delete nset;
        delete lq;
    }
};

template <class P>
void dump_common(P &m, uint64_t n, uint64_t *pos, uint8_t *rc, uint64_t *matched) {
    for (uint64_t i = 0; i < n; i++) {
        pos[i] = m.readMatchPos[i];
        rc[i] = m.readMatchRC[i] ? 1 : 0;
    }
    *matched = m.matchedReadsCount;
}

template <class P>
void dump_approx(P &m, uint64_t n, uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t *hist,
                 uint64_t *matched, uint64_t *stats) {
    dump_common(m, n, pos, rc, matched);
    for (uint64_t i = 0; i < n; i++) mism[i] = m.readMismatchesCount[i];
    for (int k = 0; k < 256; k++) hist[k] = m.matchedCountPerMismatches[k];
    if (stats) {
        stats[0] = m.betterMatchCount;
        stats[1] = m.falseMatchCount;
    }
}

} // namespace

extern "C" {

// mode: 'e' exact (DefaultReadsExactMatcher), 'd', 'i', 'c'.  pg is copied (the
// reference reverse-complements it in place, ReadsMatchers.cpp:167-171).
// index_threads sets PgHelpers::numberOfThreads (1 => canonical serial copMEM
// index, CopMEMMatcher.cpp:179-180); omp_threads the width of the per-read loop.
int pgrc_ref_match(char mode, const char *pg, uint64_t G, const char *reads, uint64_t n_lq,
                   uint64_t n_n, uint32_t L, uint32_t seed, uint8_t kmax, uint8_t kmin,
                   int rev_compl, int index_threads, int omp_threads, uint64_t *pos, uint8_t *rc,
                   uint8_t *mism, uint64_t *hist, uint64_t *matched, uint64_t *stats) {
    Silence quiet;
    PgHelpers::numberOfThreads = index_threads;
    omp_set_num_threads(omp_threads);
    std::string text(pg, G);
    ReadsHolder rh(reads, n_lq, n_n, L);
    const uint64_t n = n_lq + n_n;
    const uint32_t prefix = DefaultReadsMatcher::DISABLED_PREFIX_MODE;
    switch (mode) {
    case 'e': {
        Probe<DefaultReadsExactMatcher> m((char *)text.data(), G, rev_compl, rh.iface, prefix);
        m.matchConstantLengthReads();
        dump_common(m, n, pos, rc, matched);
        for (uint64_t i = 0; i < n; i++) mism[i] = pos[i] == UINT64_MAX ? 255 : 0;
        memset(hist, 0, 256 * sizeof(uint64_t));
        hist[0] = *matched;
        hist[255] = n - *matched;
        break;
    }
    case 'd': {
        ApproxProbe<DefaultReadsApproxMatcher> m((char *)text.data(), G, rev_compl, rh.iface,
                                                 prefix, seed, kmax, kmin);
        m.matchConstantLengthReads();
        dump_approx(m, n, pos, rc, mism, hist, matched, stats);
        break;
    }
    case 'i': {
        ApproxProbe<InterleavedReadsApproxMatcher> m((char *)text.data(), G, rev_compl, rh.iface,
                                                     prefix, seed, kmax, kmin);
        m.matchConstantLengthReads();
        dump_approx(m, n, pos, rc, mism, hist, matched, stats);
        break;
    }
    case 'c': {
        ApproxProbe<CopMEMReadsApproxMatcher> m((char *)text.data(), G, rev_compl, rh.iface,
                                                prefix, seed, kmax, kmin);
        m.matchConstantLengthReads();
        dump_approx(m, n, pos, rc, mism, hist, matched, stats);
        break;
    }
    default:
        return 2;
    }
    return 0;
}

// Pg-vs-Pg exact matching: CopMEMMatcher(src, N, target_len, ctor_min_len).matchTexts(...).  out receives count
// triples (posSrcText, length, posDestText) in discovery order; free with pgrc_ref_free.
int pgrc_ref_mem_match(const char *src, uint64_t N, const char *dest, uint64_t N2, int dest_is_src, int rev_compl,
                       uint32_t target_len, uint32_t ctor_min_len, uint32_t min_match_len, int threads,
                       uint64_t **out, uint64_t *count) {
    Silence quiet;
    PgHelpers::numberOfThreads = threads;
    omp_set_num_threads(threads);
    CopMEMMatcher matcher(src, (size_t) N, target_len, ctor_min_len);
    std::vector<TextMatch> res;
    const std::string d(dest, (size_t) N2);
    matcher.matchTexts(res, d, dest_is_src != 0, rev_compl != 0, min_match_len);
    *count = res.size();
    *out = (uint64_t *) malloc((res.size() * 3 + 1) * sizeof(uint64_t));
    for (size_t i = 0; i < res.size(); i++) {
        (*out)[3 * i] = res[i].posSrcText;
        (*out)[3 * i + 1] = res[i].length;
        (*out)[3 * i + 2] = res[i].posDestText;
    }
    return 0;
}

#ifdef PGRC_WITH_HIP_ADAPTER
} // extern "C" (the C++ definition of PgTools::mapReadsIntoPg below cannot live in a C-linkage block)
// ---------------------------------------------------------------------------------------------------------------
// End-to-end drop-in.  The reference's encoder (pgrc-encoder.cpp:359-366) calls PgTools::mapReadsIntoPg.  The
// definition below REPLACES the reference's (whose object code is kept under the name
// pgrc_ref_mapReadsIntoPg_original by oracle/Makefile): with g_gpu_matching off it forwards to the untouched
// original, with it on it is the patched function of INTEGRATION.md section 1 -- same parameter derivation
// (ReadsMatchers.cpp:699-713), HipReadsMatcher in the matcher seam, the reference's own export afterwards
// (:783-792).  New code written against the reference's public interfaces.
static bool g_gpu_matching = false;
static int g_gpu_calls = 0;
// wall time spent in the two accelerated stages, whichever implementation ran (tests/e2e_timing.py)
static double g_map_reads_s = 0, g_text_match_s = 0;
struct StageTimer {
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageTimer(double &a) : acc(a) {}
    ~StageTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

extern "C" const std::vector<bool> pgrc_ref_mapReadsIntoPg_original(
        SeparatedPseudoGenome *sPg, bool revComplPg, bool preserveOrderMode, ConstantLengthReadsSetInterface *readsSet,
        bool pairFileMode, bool revComplPairFile, uint_read_len_max matchPrefixLength, uint16_t preReadsExactMatchingChars,
        uint16_t readsExactMatchingChars, uint16_t minCharsPerMismatch, char preMatchingMode, char matchingMode,
        bool dumpInfo, ostream &pgrcOut, uint8_t compressionLevel, const string &pgDestFilePrefix,
        IndexesMapping *orgIndexesMapping);

namespace PgTools {
    const vector<bool> mapReadsIntoPg(SeparatedPseudoGenome *sPg, bool revComplPg, bool preserveOrderMode,
                                      ConstantLengthReadsSetInterface *readsSet, bool pairFileMode, bool revComplPairFile,
                                      uint_read_len_max matchPrefixLength, uint16_t preReadsExactMatchingChars,
                                      uint16_t readsExactMatchingChars, uint16_t minCharsPerMismatch, char preMatchingMode,
                                      char matchingMode, bool dumpInfo, ostream &pgrcOut, uint8_t compressionLevel,
                                      const string &pgDestFilePrefix, IndexesMapping *orgIndexesMapping) {
        StageTimer timer(g_map_reads_s);
        if (!g_gpu_matching)
            return pgrc_ref_mapReadsIntoPg_original(sPg, revComplPg, preserveOrderMode, readsSet, pairFileMode,
                                                    revComplPairFile, matchPrefixLength, preReadsExactMatchingChars,
                                                    readsExactMatchingChars, minCharsPerMismatch, preMatchingMode,
                                                    matchingMode, dumpInfo, pgrcOut, compressionLevel, pgDestFilePrefix,
                                                    orgIndexesMapping);
        g_gpu_calls++;
        const uint_read_len_max readLength = readsSet->maxReadLength();
        const uint8_t maxMismatches = readLength / minCharsPerMismatch;
        if (readsExactMatchingChars > readLength) readsExactMatchingChars = readLength;
        if (preReadsExactMatchingChars > readLength) preReadsExactMatchingChars = readLength;
        auto kindOf = [&](char mode, uint16_t seed) -> char {
            const char low = (char) tolower(mode);
            if (readLength == seed) return low == 'c' ? 'c' : 'e';
            if (low != 'c' && low != 'd' && low != 'i') {
                fprintf(stderr, "Unknown matching mode: %c.\n", mode);
                exit(EXIT_FAILURE);
            }
            return low;
        };
        const uint16_t firstSeed = preReadsExactMatchingChars > 0 ? preReadsExactMatchingChars : readsExactMatchingChars;
        const char firstMode = preReadsExactMatchingChars > 0 ? preMatchingMode : matchingMode;
        const uint8_t firstMin = (toupper(firstMode) == firstMode) ? maxMismatches : 0;
        char *pg = (char *) sPg->getPgSequence().data();
        const uint_pg_len_max pgLen = sPg->getPgSequence().length();
        HipReadsMatcher *matcher = new HipReadsMatcher(pg, pgLen, revComplPg, readsSet, matchPrefixLength, firstSeed,
                                                       maxMismatches, firstMin, kindOf(firstMode, firstSeed));
        matcher->matchConstantLengthReadsOnDevice();
        if (preReadsExactMatchingChars > 0) {
            // :713 -- the value the reference uses at :752 is still the FIRST phase's (it is recomputed only at :770)
            const uint8_t targetMismatches = readLength / firstSeed - 1;
            const uint8_t secondMin = (toupper(matchingMode) == matchingMode) ? maxMismatches : targetMismatches + 1;
            HipReadsMatcher *second = new HipReadsMatcher(pg, pgLen, revComplPg, readsSet, matchPrefixLength,
                                                          readsExactMatchingChars, maxMismatches, secondMin,
                                                          kindOf(matchingMode, readsExactMatchingChars));
            second->continueMatchingConstantLengthReadsOnDevice(matcher);
            delete matcher;
            matcher = second;
        }
        const vector<bool> res = matcher->getMatchedReadsBitmap();
        if (matchPrefixLength == DefaultReadsMatcher::DISABLED_PREFIX_MODE) {
            if (preserveOrderMode)
                matcher->exportMatchesInOriginalOrderOnDevice(sPg, pgrcOut, compressionLevel, pgDestFilePrefix,
                                                              orgIndexesMapping, pairFileMode, revComplPairFile);
            else
                matcher->exportMatchesInPgOrderOnDevice(sPg, pgrcOut, compressionLevel, pgDestFilePrefix,
                                                        orgIndexesMapping, pairFileMode, revComplPairFile);
        }
        delete matcher;
        return res;
    }
}

// SimplePgMatcher's constructor (matching/SimplePgMatcher.cpp:12-19) with the matcher choice a maintainer would add
// (INTEGRATION.md section 5); the reference's own definition is linked weak (oracle/Makefile).
static bool g_gpu_text_matching = false;
namespace PgTools {
    // times whatever TextMatcher SimplePgMatcher holds (construction = index build included)
    class TimedTextMatcher : public TextMatcher {
        TextMatcher *inner = nullptr;
    public:
        TimedTextMatcher(bool gpu, const char *src, size_t n, uint32_t targetMatchLength, uint32_t minMatchLength) {
            StageTimer timer(g_text_match_s);
            if (gpu) inner = new HipTextMatcher(src, n, targetMatchLength, minMatchLength);
            else inner = new CopMEMMatcher(src, n, targetMatchLength, minMatchLength);
        }
        ~TimedTextMatcher() override { delete inner; }
        void matchTexts(vector<TextMatch> &resMatches, const string &destText, bool destIsSrc, bool revComplMatching,
                        uint32_t minMatchLength) override {
            StageTimer timer(g_text_match_s);
            inner->matchTexts(resMatches, destText, destIsSrc, revComplMatching, minMatchLength);
        }
    };

    SimplePgMatcher::SimplePgMatcher(const string &srcPg, uint32_t targetMatchLength, uint32_t minMatchLength)
            : targetMatchLength(targetMatchLength), srcPg(srcPg) {
        cout << "Source pseudogenome length: " << srcPg.length() << endl;
        if (srcPg.size() >= targetMatchLength)
            matcher = new TimedTextMatcher(g_gpu_text_matching, srcPg.data(), srcPg.length(), targetMatchLength, minMatchLength);
    }
}

// The two factories of DividedPCLReadsSets the encoder calls (pgrc-encoder.cpp:258, :279).  oracle/Makefile keeps the
// reference's own under the names below; these definitions forward to them, or -- g_gpu_division -- to
// integration/HipDividedReadsSets (the call-site change of INTEGRATION.md).
static bool g_gpu_division = false;
static int g_division_calls = 0;
static double g_division_s = 0;                  // wall time inside the two factories, whichever implementation ran
// the FASTQ files of the running encode (pgrc_ref_encode): the GPU division reads them itself unless PGRC_DIVIDE_FROM_ROWS is
// set, in which case it takes the records from the reference's iterator
static std::string g_src_fastq, g_pair_fastq;
static const PgRCParams *g_params = nullptr;     // (revComplPairFile is decided inside the encoder, pgrc-encoder.cpp:49-52: read when needed)
extern "C" DividedPCLReadsSets *pgrc_ref_divide_quality_original(ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt,
                                                                  uint_read_len_max readLength, double error_limit,
                                                                  bool simplified_suffix_mode, bool separateNReadsSet, bool nReadsLQ);
extern "C" DividedPCLReadsSets *pgrc_ref_divide_simple_original(ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt,
                                                                 uint_read_len_max readLength, bool separateNReadsSet, bool nReadsLQ);
namespace PgTools {
    DividedPCLReadsSets *DividedPCLReadsSets::getQualityDivisionBasedReadsSets(
            ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength, double error_limit,
            bool simplified_suffix_mode, bool separateNReadsSet, bool nReadsLQ) {
        StageTimer timer(g_division_s);
        if (!g_gpu_division)
            return pgrc_ref_divide_quality_original(readsIt, readLength, error_limit, simplified_suffix_mode, separateNReadsSet, nReadsLQ);
        g_division_calls++;
        if (!getenv("PGRC_DIVIDE_FROM_ROWS") && !g_src_fastq.empty())
            return HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq(g_src_fastq, g_pair_fastq, g_params && g_params->revComplPairFile, readLength,
                                                                                  error_limit, simplified_suffix_mode, separateNReadsSet, nReadsLQ);
        return HipDividedReadsSets::getQualityDivisionBasedReadsSets(readsIt, readLength, error_limit, simplified_suffix_mode,
                                                                     separateNReadsSet, nReadsLQ);
    }
    DividedPCLReadsSets *DividedPCLReadsSets::getSimpleDividedPCLReadsSets(
            ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength, bool separateNReadsSet, bool nReadsLQ) {
        StageTimer timer(g_division_s);
        if (!g_gpu_division) return pgrc_ref_divide_simple_original(readsIt, readLength, separateNReadsSet, nReadsLQ);
        g_division_calls++;
        if (!getenv("PGRC_DIVIDE_FROM_ROWS") && !g_src_fastq.empty())
            return HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq(g_src_fastq, g_pair_fastq, g_params && g_params->revComplPairFile, readLength, 1,
                                                                                  false, separateNReadsSet, nReadsLQ);
        return HipDividedReadsSets::getSimpleDividedPCLReadsSets(readsIt, readLength, separateNReadsSet, nReadsLQ);
    }
}
extern "C" int pgrc_ref_division_calls() { return g_division_calls; }
extern "C" double pgrc_ref_division_seconds() { return g_division_s; }

// Runs the reference's whole encoder (PgRC.cpp:244-262) on a FASTQ file with the CPU or the GPU matcher.
// Returns the number of times the GPU mapReadsIntoPg ran (>= 0), or a negative error.
extern "C" int pgrc_ref_encode(const char *fastq, const char *pair_fastq, const char *archive, int threads, int use_gpu,
                               int preserve_order, char mode, int seed_len, int min_chars_per_mismatch, char pre_mode,
                               int pre_seed_len) {
    Silence quiet;
    PgHelpers::numberOfThreads = threads;
    omp_set_num_threads(threads);
    g_gpu_matching = (use_gpu & 1) != 0;          // bit 0: reads -> Pg matching (stage 4) on the GPU
    g_gpu_text_matching = (use_gpu & 2) != 0;     // bit 1: Pg -> Pg matching (stage 7) on the GPU
    g_gpu_division = (use_gpu & 4) != 0;          // bit 2: read-set division + packing (stage 1) on the GPU
    g_gpu_calls = 0;
    g_map_reads_s = g_text_match_s = g_division_s = 0;
    PgRCParams *params = new PgRCParams();
    params->setSrcFastqFile(fastq);
    if (pair_fastq && pair_fastq[0]) params->setPairFastqFile(pair_fastq);
    if (preserve_order) params->setPreserveOrderMode();
    if (mode) params->setMatchingMode(mode);
    if (seed_len > 0) params->setReadSeedLength((uint16_t) seed_len);
    if (min_chars_per_mismatch > 0) params->setMinCharsPerMismatch((uint16_t) min_chars_per_mismatch);
    if (pre_mode) params->setPreMatchingMode(pre_mode);
    if (pre_seed_len > 0) params->setPreReadsExactMatchingChars((uint16_t) pre_seed_len);
    params->setPgRCFileName(archive);
    // quality-based division (PgRC's -q): PGRC_REF_Q_PROMILS = error limit in promils, PGRC_REF_Q_FULL=1 = the arithmetic-mean
    // test instead of the one-position one (PgRC.cpp's -q options; kept out of the signature: test knobs)
    if (const char *v = getenv("PGRC_REF_Q_PROMILS")) params->setQualityBasedDivisionErrorLimitInPromils((uint16_t) atoi(v));
    if (getenv("PGRC_REF_Q_FULL")) params->disableSimplifiedSuffixMode4QualityBasedDivision();
    if (getenv("PGRC_REF_N_READS_LQ")) params->setNReadsLQ();
    g_src_fastq = fastq;
    g_pair_fastq = (pair_fastq && pair_fastq[0]) ? pair_fastq : "";
    g_params = params;
    {
        PgRCEncoder encoder(params);
        encoder.executePgRCChain();
    }
    delete params;
    g_src_fastq.clear();
    g_params = nullptr;
    g_gpu_matching = false;
    g_gpu_text_matching = false;
    g_gpu_division = false;
    return g_gpu_calls;
}

extern "C" uint64_t pgrc_ref_bulk_updates() { return HipReadsMatcher::bulkUpdatesServed; }
extern "C" uint64_t pgrc_ref_packed_handovers() { return HipReadsMatcher::packedHandOvers; }
extern "C" uint64_t pgrc_ref_device_exports() { return HipReadsMatcher::deviceExports; }
extern "C" uint64_t pgrc_ref_dual_runs() { return HipReadsMatcher::dualRuns; }
extern "C" uint64_t pgrc_ref_streamed_runs() { return HipReadsMatcher::streamedRuns; }
extern "C" double pgrc_ref_phase_seconds(const char *name) {
    auto it = HipReadsMatcher::phaseSeconds.find(name);
    return it == HipReadsMatcher::phaseSeconds.end() ? 0.0 : it->second;
}

// Stage 4 of the encoder at full size (tools/stage4_c3.py): PgTools::mapReadsIntoPg's matcher on the encoder's LQ + N sum set,
// given as the reference's own packed rows (no FASTQ, no division: the sets are filled with copyPackedRead), either the
// reference's CopMEMReadsApproxMatcher (`index_threads` = PgHelpers::numberOfThreads, `omp_threads`) or HipReadsMatcher;
// with_export: the adapter's Pg-order export up to the builder's own compression (position sort + streams from the
// device) against a synthetic reads list of `list_count` entries already on the pseudogenome.
// secs[0] = read sets built, [1] = matcher constructed + matchConstantLengthReads, [2] = export part; fnv = checksum of
// the result vectors.
namespace {
struct Stage4Probe : HipReadsMatcher {
    using HipReadsMatcher::HipReadsMatcher;
    using HipReadsMatcher::createSeparatedPseudoGenomeOutputBuilder;
    using HipReadsMatcher::readMatchPos;
    using HipReadsMatcher::readMismatchesCount;
    using HipReadsMatcher::readMatchRC;
    using HipReadsMatcher::matchedReadsCount;
};
struct Stage4Cpu : CopMEMReadsApproxMatcher {
    using CopMEMReadsApproxMatcher::CopMEMReadsApproxMatcher;
    using CopMEMReadsApproxMatcher::readMatchPos;
    using CopMEMReadsApproxMatcher::readMismatchesCount;
    using CopMEMReadsApproxMatcher::readMatchRC;
    using CopMEMReadsApproxMatcher::matchedReadsCount;
};
template <class M>
uint64_t stage4_fnv(M &m, uint64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; i++) {
        h = (h ^ m.readMatchPos[i]) * 1099511628211ull;
        h = (h ^ (uint64_t)(m.readMismatchesCount[i] | (m.readMatchRC[i] ? 256u : 0u))) * 1099511628211ull;
    }
    return h;
}
}
extern "C" int pgrc_ref_stage4(int use_adapter, char *pg, uint64_t G, const uint8_t *lq_rows, uint64_t n_lq, const uint8_t *n_rows,
                               uint64_t n_n, uint32_t L, uint32_t seed, uint8_t kmax, int index_threads, int omp_threads, int with_export,
                               uint64_t list_count, double *secs, uint64_t *matched, uint64_t *fnv) {
    Silence quiet;
    PgHelpers::numberOfThreads = index_threads;
    omp_set_num_threads(omp_threads);
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t = now();
    PackedConstantLengthReadsSet lq(L, "ACGT", 4), nset(L, "ACGNT", 5);
    lq.resize(n_lq);
    if (n_lq) lq.copyPackedRead(lq_rows, 0, n_lq);
    nset.resize(n_n);
    if (n_n) nset.copyPackedRead(n_rows, 0, n_n);
    SumOfConstantLengthReadsSets sum(&lq, &nset);                 // pgrc-encoder.cpp:349-352
    secs[0] = now() - t;
    const uint64_t n = n_lq + n_n;
    const uint32_t pm = DefaultReadsMatcher::DISABLED_PREFIX_MODE;
    secs[2] = 0;
    if (!use_adapter) {
        t = now();
        Stage4Cpu m(pg, G, true, &sum, pm, seed, kmax, 0);
        m.matchConstantLengthReads();
        secs[1] = now() - t;
        *matched = m.matchedReadsCount;
        *fnv = stage4_fnv(m, n);
        return 0;
    }
    t = now();
    Stage4Probe m(pg, G, true, &sum, pm, seed, kmax, 0, 'c');
    m.matchConstantLengthReadsOnDevice();
    secs[1] = now() - t;
    *matched = m.matchedReadsCount;
    *fnv = stage4_fnv(m, n);
    if (with_export) {
        ReadsSetProperties props;
        props.readsCount = list_count;
        props.allReadsLength = list_count * L;
        props.constantReadLength = true;
        props.minReadLength = props.maxReadLength = L;
        props.symbolsCount = 4;
        strcpy(props.symbolsList, "ACGT");
        props.generateSymbolOrder();
        auto *rl = new ExtendedReadsListWithConstantAccessOption(L);
        rl->off.resize(list_count);
        rl->orgIdx.resize(list_count);
        rl->revComp.resize(list_count);
        const uint64_t step = list_count ? std::max<uint64_t>(1, std::min<uint64_t>(255, (G - L) / list_count)) : 1;
        for (uint64_t k = 0; k < list_count; k++) {
            rl->off[k] = (uint8_t) step;
            rl->orgIdx[k] = (uint32_t) (n + k);
            rl->revComp[k] = (uint8_t) ((k * 2654435761u >> 13) & 1u);
        }
        SeparatedPseudoGenome sPg(std::string(), rl, &props);
        DirectMapping mapping((uint_reads_cnt_max) n);
        t = now();
        SeparatedPseudoGenomeOutputBuilder *builder = m.createSeparatedPseudoGenomeOutputBuilder(&sPg);
        const std::shared_ptr<void> streams = m.makePgOrderStreams(&sPg, &mapping, false, builder);   // (outlives the builder)
        secs[2] = now() - t;
        delete builder;
    }
    return 0;
}
// the adapter's position order (the reference's sort on (position, index) pairs) of a given result state
extern "C" uint64_t pgrc_ref_position_order(const uint64_t *pos, uint64_t n, int omp_threads, uint32_t *order) {
    omp_set_num_threads(omp_threads);
    std::vector<uint64_t> p(pos, pos + n);
    uint64_t matched = 0;
    for (uint64_t i = 0; i < n; i++) matched += pos[i] != UINT64_MAX;
    std::vector<uint32_t> o;
    HipReadsMatcher::positionOrder(p, (uint_reads_cnt_max) matched, o);
    memcpy(order, o.data(), o.size() * sizeof(uint32_t));
    return o.size();
}
extern "C" uint64_t pgrc_ref_text_match_calls() { return HipTextMatcher::callsServed; }
// seconds the last pgrc_ref_encode spent in mapReadsIntoPg (stage 4) and in SimplePgMatcher's TextMatcher (stage 7)
extern "C" void pgrc_ref_stage_seconds(double *map_reads_s, double *text_match_s) {
    *map_reads_s = g_map_reads_s;
    *text_match_s = g_text_match_s;
}

// CopMEMMatcher::matchTexts through the adapter class (what SimplePgMatcher would call)
extern "C" int pgrc_ref_mem_match_via_adapter(const char *src, uint64_t N, const char *dest, uint64_t N2, int dest_is_src,
                                              int rev_compl, uint32_t target_len, uint32_t min_match_len, uint64_t **out,
                                              uint64_t *count) {
    Silence quiet;
    HipTextMatcher matcher(src, (size_t) N, target_len);
    std::vector<TextMatch> res;
    const std::string d(dest, (size_t) N2);
    static_cast<TextMatcher &>(matcher).matchTexts(res, d, dest_is_src != 0, rev_compl != 0, min_match_len);
    *count = res.size();
    *out = (uint64_t *) malloc((res.size() * 3 + 1) * sizeof(uint64_t));
    for (size_t i = 0; i < res.size(); i++) {
        (*out)[3 * i] = res[i].posSrcText;
        (*out)[3 * i + 1] = res[i].length;
        (*out)[3 * i + 2] = res[i].posDestText;
    }
    return 0;
}

extern "C" int pgrc_ref_decode(const char *archive, int threads) {
    Silence quiet;
    PgHelpers::numberOfThreads = threads;
    omp_set_num_threads(threads);
    PgRCParams *params = new PgRCParams();
    params->setPgRCFileName(archive);
    {
        PgRCDecoder decoder(params);
        decoder.decompressPgRC();
    }
    delete params;
    return 0;
}

// Matching + export of a synthetic case with an explicit reads list on the pseudogenome, stream files written by
// SeparatedPseudoGenomeOutputBuilder::build(prefix): <prefix>_rl_{off,idx,rc,mis_cnt,mis_sym,mis_roff}.pg.
// use_adapter 0: the reference's matcher and its exportMatchesInPgOrder / exportMatchesInOriginalOrder
// (ReadsMatchers.cpp:563-675); 1: HipReadsMatcher and its device export.  archive receives the bytes written to pgrcOut.
extern "C" int pgrc_ref_export_run(int use_adapter, const char *pg, uint64_t G, const char *reads, uint64_t n_lq, uint64_t n_n,
                                   uint32_t L, uint32_t seed, uint8_t kmax, uint8_t kmin, const uint8_t *list_off,
                                   const uint32_t *list_org, const uint8_t *list_rc, uint64_t list_count,
                                   const uint32_t *read_org, uint32_t reads_total, int preserve_order, int pair_file_mode,
                                   int rev_compl_pair_file, int omp_threads, const char *prefix, const char *archive) {
    Silence quiet;
    PgHelpers::numberOfThreads = 1;           // serial (canonical) copMEM index
    omp_set_num_threads(omp_threads);
    const uint64_t n = n_lq + n_n;
    ReadsHolder rh(reads, n_lq, n_n, L);
    ReadsSetProperties props;
    props.readsCount = list_count;
    props.allReadsLength = list_count * L;
    props.constantReadLength = true;
    props.minReadLength = props.maxReadLength = L;
    props.symbolsCount = 4;
    strcpy(props.symbolsList, "ACGT");
    props.generateSymbolOrder();
    auto *rl = new ExtendedReadsListWithConstantAccessOption(L);
    rl->off.assign(list_off, list_off + list_count);
    rl->orgIdx.assign(list_org, list_org + list_count);
    if (list_rc) rl->revComp.assign(list_rc, list_rc + list_count);
    SeparatedPseudoGenome sPg(std::string(pg, G), rl, &props);
    IndexesMapping *mapping;
    if (read_org) {
        std::vector<uint_reads_cnt_max> mv(read_org, read_org + n);
        mapping = new VectorMapping(std::move(mv), reads_total);
    } else
        mapping = new DirectMapping((uint_reads_cnt_max) n);
    std::ostringstream out;
    const uint32_t pm = DefaultReadsMatcher::DISABLED_PREFIX_MODE;
    char *text = (char *) sPg.getPgSequence().data();
    if (use_adapter) {
        HipReadsMatcher m(text, G, true, rh.iface, pm, seed, kmax, kmin, 'c');
        m.matchConstantLengthReadsOnDevice();
        if (preserve_order)
            m.exportMatchesInOriginalOrderOnDevice(&sPg, out, CODER_LEVEL_NORMAL, prefix, mapping, pair_file_mode, rev_compl_pair_file);
        else
            m.exportMatchesInPgOrderOnDevice(&sPg, out, CODER_LEVEL_NORMAL, prefix, mapping, pair_file_mode, rev_compl_pair_file);
    } else {
        CopMEMReadsApproxMatcher m(text, G, true, rh.iface, pm, seed, kmax, kmin);
        m.matchConstantLengthReads();
        if (preserve_order)
            m.exportMatchesInOriginalOrder(&sPg, out, CODER_LEVEL_NORMAL, prefix, mapping, pair_file_mode, rev_compl_pair_file);
        else
            m.exportMatchesInPgOrder(&sPg, out, CODER_LEVEL_NORMAL, prefix, mapping, pair_file_mode, rev_compl_pair_file);
    }
    delete mapping;
    PgHelpers::writeStringToFile(archive, out.str());
    return 0;
}

extern "C" {
// The drop-in itself: the reference's own flow with integration/HipReadsMatcher in the matcher seam.
// entry 0 = DefaultReadsMatcher::matchConstantLengthReads() (base-class driver: initMatching, executeMatching(false),
// in-place RC of the host text, executeMatching(true), RC back); entry 1 = matchConstantLengthReadsOnDevice().
int pgrc_ref_match_via_adapter(char mode, const char *pg, uint64_t G, const char *reads, uint64_t n_lq,
                               uint64_t n_n, uint32_t L, uint32_t seed, uint8_t kmax, uint8_t kmin, int rev_compl,
                               int entry, uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t *hist,
                               uint64_t *matched) {
    Silence quiet;
    std::string text(pg, G);
    ReadsHolder rh(reads, n_lq, n_n, L);
    ApproxProbe<HipReadsMatcher> m((char *)text.data(), G, rev_compl, rh.iface,
                                   DefaultReadsMatcher::DISABLED_PREFIX_MODE, seed, kmax, kmin, mode);
    if (entry == 0) m.matchConstantLengthReads();
    else m.matchConstantLengthReadsOnDevice();
    if (memcmp(text.data(), pg, G) != 0) return 9; // the caller's text must come back unchanged
    dump_approx(m, n_lq + n_n, pos, rc, mism, hist, matched, nullptr);
    return 0;
}
#endif

// copMEM index internals (CopMEMMatcher.cpp:69-231).  cumm/positions are
// malloc'ed copies; free with pgrc_ref_free.
int pgrc_ref_copmem_index(const char *pg, uint64_t G, uint32_t seed, int index_threads, int32_t *K,
                          int32_t *k1, int32_t *k2, uint32_t *hash_size, uint32_t **cumm,
                          uint32_t **positions, uint64_t *count) {
    Silence quiet;
    PgHelpers::numberOfThreads = index_threads;
    std::string text(pg, G);
    CopMEMMatcher m(text.data(), G, seed);
    if (m.bigRef != 0) return 3;
    *K = m.K; *k1 = m.k1; *k2 = m.k2; *hash_size = m.hash_size;
    uint64_t total = m.buffer0.second[m.hash_size];
    *count = total;
    *cumm = (uint32_t *)malloc(((size_t)m.hash_size + 2) * sizeof(uint32_t));
    memcpy(*cumm, m.buffer0.second, ((size_t)m.hash_size + 2) * sizeof(uint32_t));
    *positions = (uint32_t *)malloc((size_t)(total + 1) * sizeof(uint32_t));
    memcpy(*positions, m.buffer0.first, (size_t)total * sizeof(uint32_t));
    return 0;
}

// raw (unmasked) maRushPrime1HashSparsified<K> through the matcher's own
// function-pointer table (CopMEMMatcher.cpp:52-62, :84).
uint32_t pgrc_ref_copmem_hash(uint32_t seed, const char *str, int32_t *K_out) {
    Silence quiet;
    PgHelpers::numberOfThreads = 1;
    static const std::string dummy(4096, 'A');
    CopMEMMatcher m(dummy.data(), dummy.size(), seed);
    if (K_out) *K_out = m.K;
    return m.hashFunc32(str);
}

// One read against a fresh serial index; exposes the per-read false count
// (CopMEMMatcher.cpp:483-566) for the known-answer values of SURVEY.md Appendix C.
uint64_t pgrc_ref_copmem_match_read(const char *pg, uint64_t G, uint32_t seed, const char *read,
                                    uint32_t L, uint8_t kmax, uint8_t kmin, uint8_t *cnt,
                                    uint64_t *falses) {
    Silence quiet;
    PgHelpers::numberOfThreads = 1;
    std::string text(pg, G);
    CopMEMMatcher m(text.data(), G, seed);
    uint64_t better = 0, f = 0;
    uint64_t p = m.approxMatchPattern(read, (uint_read_len_max)L, kmax, kmin, *cnt, better, f);
    *falses = f;
    return p;
}

// Mismatch extraction through AbstractReadsApproxMatcher::updateEntry
// (ReadsMatchers.cpp:548-559).  Returns the number of mismatches written.
int pgrc_ref_extract(const char *pg, uint64_t G, const char *read, uint32_t L, uint64_t pos, int rc,
                     uint8_t cnt, uint32_t org_idx, int rev_compl_pair_file, uint8_t *codes,
                     uint16_t *offsets) {
    Silence quiet;
    std::string text(pg, G);
    bool has_n = memchr(read, 'N', L) != nullptr;
    ReadsHolder rh(read, has_n ? 0 : 1, has_n ? 1 : 0, L);
    ApproxProbe<CopMEMReadsApproxMatcher> m((char *)text.data(), G, true, rh.iface,
                                            DefaultReadsMatcher::DISABLED_PREFIX_MODE, L, 255, 0);
    m.readMatchPos.assign(1, pos);
    m.readMatchRC.assign(1, rc != 0);
    m.readMismatchesCount.assign(1, cnt);
    DefaultReadsListEntry entry(0);
    entry.advanceEntryByPosition(pos, org_idx, rc != 0);
    m.updateEntry(entry, 0, rev_compl_pair_file != 0);
    for (int i = 0; i < entry.mismatchesCount; i++) {
        codes[i] = entry.mismatchCode[i];
        offsets[i] = entry.mismatchOffset[i];
    }
    return entry.mismatchesCount;
}

// SymbolsPackingFacility layout of one read (PackedConstantLengthReadsSet.cpp:36-45).
int pgrc_ref_pack_read(const char *read, uint32_t L, const char *alphabet, uint8_t *dst) {
    Silence quiet;
    PackedConstantLengthReadsSet s(L, alphabet, (uint8_t)strlen(alphabet));
    s.addRead(read, L);
    int spe = SymbolsPackingFacility::maxSymbolsPerElement((uint8_t)strlen(alphabet));
    int bytes = (L + spe - 1) / spe;
    memcpy(dst, s.getPackedRead(0), bytes);
    return bytes;
}

void pgrc_ref_revcomp(char *seq, uint64_t n) { PgHelpers::reverseComplementInPlace(seq, n); }

void pgrc_ref_free(void *p) { free(p); }

int pgrc_ref_max_threads(void) { return omp_get_num_procs(); }

} // extern "C"

// ---- row f3: the reference's read-set division over in-memory FASTQ records -------------------------------------------

namespace {
// n records as two row arrays behind the reference's iterator interface (readsset/iterator/ReadsSetIterator.h:80-96)
struct RowsIterator : ReadsSourceIteratorTemplate<uint_read_len_max> {
    const char *rows, *quals;
    uint64_t n;
    uint32_t L;
    int64_t at = -1;
    std::string read, qual;
    RowsIterator(const char *r, const char *q, uint64_t n_, uint32_t L_) : rows(r), quals(q), n(n_), L(L_) {}
    bool moveNext() override {
        if (at + 1 >= (int64_t) n) return false;
        at++;
        read.assign(rows + (size_t) at * L, L);
        if (quals) qual.assign(quals + (size_t) at * L, L);
        return true;
    }
    std::string &getRead() override { return read; }
    std::string &getQualityInfo() override { return qual; }
    uint_read_len_max getReadLength() override { return (uint_read_len_max) L; }
    void rewind() override { at = -1; }
    IndexesMapping *retainVisitedIndexesMapping() override { return new DirectMapping((uint_reads_cnt_max) n); }
};
}

static int divide_out(DividedPCLReadsSets *sets, uint64_t n, uint32_t L, uint8_t *hq_rows, uint8_t *lq_rows, uint8_t *n_rows,
                      uint32_t *lq_index, uint32_t *n_index, uint64_t counts[3], uint32_t symbols[3]) {
    PackedConstantLengthReadsSet *set[3] = {sets->getHqReadsSet(), sets->getLqReadsSet(), sets->getNReadsSet()};
    uint8_t *dst[3] = {hq_rows, lq_rows, n_rows};
    for (int k = 0; k < 3; k++) {
        counts[k] = set[k] ? set[k]->readsCount() : 0;
        symbols[k] = set[k] ? set[k]->getReadsSetProperties()->symbolsCount : 0;
        if (!set[k] || !counts[k]) continue;
        const uint32_t per = symbols[k] == 4 ? 4 : 3, rb = (L + per - 1) / per;
        memcpy(dst[k], set[k]->getPackedRead(0), (size_t) counts[k] * rb);
    }
    IndexesMapping *lm = sets->getLqReadsIndexesMapping(), *nm = sets->getNReadsIndexesMapping();
    if (lm && lm->getMappedReadsCount() != counts[1]) return 2;
    for (uint64_t i = 0; lm && i < counts[1]; i++) lq_index[i] = (uint32_t) lm->getReadOriginalIndex((uint_reads_cnt_max) i);
    if (nm && nm->getMappedReadsCount() != counts[2]) return 3;
    for (uint64_t i = 0; nm && i < counts[2]; i++) n_index[i] = (uint32_t) nm->getReadOriginalIndex((uint_reads_cnt_max) i);
    if (n != ~0ull && lm && lm->getReadsTotalCount() != n) return 4;
    delete sets;
    return 0;
}

// DividedPCLReadsSets::getQualityDivisionBasedReadsSets (use_adapter = 0) or integration/HipDividedReadsSets (1) over the
// records; the sets' packed rows and the mappings go to the caller's buffers (n * ceil(L / 3) bytes / n entries each)
extern "C" int pgrc_ref_divide(int use_adapter, const char *reads, const char *quals, uint64_t n, uint32_t L, double error_limit,
                               int simplified, int separate_n, int n_reads_lq, uint8_t *hq_rows, uint8_t *lq_rows, uint8_t *n_rows,
                               uint32_t *lq_index, uint32_t *n_index, uint64_t counts[3], uint32_t symbols[3]) {
    Silence quiet;
    RowsIterator it(reads, quals, n, L);
    DividedPCLReadsSets *sets = nullptr;
    if (use_adapter) {
#ifdef PGRC_WITH_HIP_ADAPTER
        sets = HipDividedReadsSets::getQualityDivisionBasedReadsSets(&it, (uint_read_len_max) L, error_limit, simplified != 0,
                                                                      separate_n != 0, n_reads_lq != 0);
#else
        return -1;
#endif
    } else {
#ifdef PGRC_WITH_HIP_ADAPTER
        sets = pgrc_ref_divide_quality_original(&it, (uint_read_len_max) L, error_limit, simplified != 0, separate_n != 0, n_reads_lq != 0);
#else
        sets = DividedPCLReadsSets::getQualityDivisionBasedReadsSets(&it, (uint_read_len_max) L, error_limit, simplified != 0,
                                                                     separate_n != 0, n_reads_lq != 0);
#endif
    }
    return divide_out(sets, n, L, hq_rows, lq_rows, n_rows, lq_index, n_index, counts, symbols);
}

// ... the same over FASTQ FILES: the reference's managed iterator (ReadsSetPersistence.cpp:20-56) feeding its factory, or
// HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq, which parses the text on the device (buffers for max_reads)
extern "C" int pgrc_ref_divide_files(int use_adapter, const char *src, const char *pair, int rev_compl_pair, uint32_t L, double error_limit,
                                     int simplified, int separate_n, int n_reads_lq, uint8_t *hq_rows, uint8_t *lq_rows, uint8_t *n_rows,
                                     uint32_t *lq_index, uint32_t *n_index, uint64_t counts[3], uint32_t symbols[3]) {
    Silence quiet;
    DividedPCLReadsSets *sets = nullptr;
    const std::string pairFile = (pair && pair[0]) ? pair : "";
    if (use_adapter) {
#ifdef PGRC_WITH_HIP_ADAPTER
        sets = HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq(src, pairFile, rev_compl_pair != 0, (uint_read_len_max) L, error_limit,
                                                                               simplified != 0, separate_n != 0, n_reads_lq != 0);
#else
        return -1;
#endif
    } else {
        ReadsSourceIteratorTemplate<uint_read_len_max> *it = ReadsSetPersistence::createManagedReadsIterator(src, pairFile, rev_compl_pair != 0);
#ifdef PGRC_WITH_HIP_ADAPTER
        sets = pgrc_ref_divide_quality_original(it, (uint_read_len_max) L, error_limit, simplified != 0, separate_n != 0, n_reads_lq != 0);
#else
        sets = DividedPCLReadsSets::getQualityDivisionBasedReadsSets(it, (uint_read_len_max) L, error_limit, simplified != 0, separate_n != 0,
                                                                     n_reads_lq != 0);
#endif
        delete it;
    }
    return divide_out(sets, ~0ull, L, hq_rows, lq_rows, n_rows, lq_index, n_index, counts, symbols);
}

// the reference's table of per-quality probabilities (utils/helper.cpp:284-327, a global of that file)
extern float *qualityLut;
extern "C" float pgrc_ref_quality_lut(int c) { return qualityLut[c]; }
#ifdef PGRC_WITH_HIP_ADAPTER
extern "C" uint64_t pgrc_ref_divide_batches() { return HipDividedReadsSets::batchesServed; }
#endif
