// SumOfConstantLengthReadsSets (readsset/ReadsSetInterface.h:45-58) -- the LQ + N set the encoder hands to
// mapReadsIntoPg (pgrc-encoder.cpp:349-352) -- keeps its two halves private and has no accessor.  In a PgRC tree the
// maintainer adds `friend class PgTools::HipReadsMatcher;` to that class.  Built against
// an UNPATCHED tree (-DPGRC_UNPATCHED_TREE; oracle/Makefile compiles the reference's sources where they lie and
// must not edit them) this one translation unit opens the class up instead; layout and ABI are unaffected.
#ifdef PGRC_UNPATCHED_TREE
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#define private public
#include "readsset/ReadsSetInterface.h"
// ... and SeparatedPseudoGenomeOutputBuilder keeps its stream destinations private; the device export appends whole
// streams to them (in a PgRC tree: `friend class HipReadsMatcher;` in that class as well)
#include "pseudogenome/persistence/SeparatedPseudoGenomePersistence.h"
#undef private
#endif

#include "HipReadsMatcher.h"

#include <numeric>

#include "pgrc_match.h"
#include "readsset/PackedConstantLengthReadsSet.h"

#include <chrono>
#include <omp.h>
#include <thread>
#include <parallel/algorithm>

// what an export call of the library returned, freed with the object
struct pgrc_export_streams_view {
    pgrc_export_streams s;
    pgrc_export_streams_view() { memset(&s, 0, sizeof s); }
    ~pgrc_export_streams_view() { pgrc_match_free_export(&s); }
    pgrc_export_streams_view(const pgrc_export_streams_view &) = delete;
    pgrc_export_streams_view &operator=(const pgrc_export_streams_view &) = delete;
};

namespace {
    // PGRC_HIP_TIMING=1: phase times of the adapter on stderr (hand-over, device run, result fetch)
    struct PhaseLog {
        const char *what;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        explicit PhaseLog(const char *w) : what(w) {}
        ~PhaseLog() {
            static const bool on = getenv("PGRC_HIP_TIMING") != nullptr;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            PgTools::HipReadsMatcher::phaseSeconds[what] += s;
            if (on) fprintf(stderr, "HipReadsMatcher: %s %.3f s\n", what, s);
        }
    };
}

namespace PgTools {

    uint64_t HipReadsMatcher::bulkUpdatesServed = 0;
    uint64_t HipReadsMatcher::packedHandOvers = 0;
    uint64_t HipReadsMatcher::deviceExports = 0;
    uint64_t HipReadsMatcher::dualRuns = 0;
    uint64_t HipReadsMatcher::streamedRuns = 0;
    std::map<std::string, double> HipReadsMatcher::phaseSeconds;

    HipReadsMatcher::HipReadsMatcher(char *pgPtr, const uint_pg_len_max pgLength, bool revComplPg,
                                     ConstantLengthReadsSetInterface *readsSet, uint32_t matchPrefixLength,
                                     uint16_t readsExactMatchingChars, uint8_t maxMismatches, uint8_t minMismatches,
                                     char hipMode)
            : AbstractReadsApproxMatcher(pgPtr, pgLength, revComplPg, readsSet, matchPrefixLength,
                                         readsExactMatchingChars, maxMismatches, minMismatches),
              hipMode(hipMode), seedChars(readsExactMatchingChars) {
        if (matchPrefixLength != DISABLED_PREFIX_MODE) {
            fprintf(stderr, "HipReadsMatcher: prefix matching mode is not supported.\n");
            exit(EXIT_FAILURE);
        }
        pgrc_match_params prm;
        prm.read_len = readLength;
        prm.seed_len = readsExactMatchingChars > readLength ? readLength : readsExactMatchingChars;
        prm.max_mismatches = maxMismatches;
        prm.min_mismatches = minMismatches;
        prm.mode = hipMode;
        prm.device = -1;
        // PGRC_DEVICES = "all" or a comma-separated list of HIP device ordinals: ONE matcher over several GPUs (reads
        // sharded, text all-gathered; include/pgrc_match.h, pgrc_match_create_multi).  Unset: the current device.
        int e;
        std::vector<int32_t> devices;
        if (const char *dv = getenv("PGRC_DEVICES")) {
            if (strcmp(dv, "all") == 0) {
                int32_t cnt = 0;
                pgrc_match_device_count(&cnt);
                for (int32_t d = 0; d < cnt; d++) devices.push_back(d);
            } else {
                for (const char *p = dv; *p;) {
                    char *end = nullptr;
                    const long d = strtol(p, &end, 10);
                    if (end == p) break;
                    devices.push_back((int32_t) d);
                    p = (*end == ',') ? end + 1 : end;
                }
            }
        }
        if (!devices.empty())
            e = pgrc_match_create_multi(&prm, (int32_t) devices.size(), devices.data(), &ctx);
        else
            e = pgrc_match_create(&prm, &ctx);
        if (e) {
            fprintf(stderr, "HipReadsMatcher: %s (error %d)\n", pgrc_match_last_error(nullptr), e);
            exit(EXIT_FAILURE); // the reference's error convention (ReadsMatchers.cpp:738-739)
        }
    }

    HipReadsMatcher::~HipReadsMatcher() {
        pgrc_match_destroy(ctx);
    }

    void HipReadsMatcher::failOn(int code, const char *what) {
        if (!code) return;
        fprintf(stderr, "HipReadsMatcher: %s failed: %s (error %d)\n", what, pgrc_match_last_error(ctx), code);
        exit(EXIT_FAILURE);
    }

    // The reference's packed sets go over as they are (f3): an "ACGT" set (4 symbols per byte), an "ACGNT" set (3 per
    // byte), or the LQ + N sum of two such sets; the device unpacks them (pgrc_match_append_reads_packed).
    bool HipReadsMatcher::packedHalves(PackedConstantLengthReadsSet *half[2], int32_t sym[2]) const {
        auto packedSymbols = [](PackedConstantLengthReadsSet *p) -> int32_t {
            if (!p) return 0;
            const ReadsSetProperties *pr = p->getReadsSetProperties();
            if (pr->symbolsCount == 4 && strncmp(pr->symbolsList, "ACGT", 4) == 0) return 4;
            if (pr->symbolsCount == 5 && strncmp(pr->symbolsList, "ACGNT", 5) == 0) return 5;
            return 0;
        };
        half[0] = dynamic_cast<PackedConstantLengthReadsSet *>(readsSet);
        half[1] = nullptr;
        if (auto *sum = dynamic_cast<SumOfConstantLengthReadsSets *>(readsSet)) {
            half[0] = dynamic_cast<PackedConstantLengthReadsSet *>(sum->clrs1);
            half[1] = dynamic_cast<PackedConstantLengthReadsSet *>(sum->clrs2);
            if (!half[1]) half[0] = nullptr;
        }
        sym[0] = packedSymbols(half[0]);
        sym[1] = packedSymbols(half[1]);
        return sym[0] && (!half[1] || sym[1]);
    }

    void HipReadsMatcher::appendPackedHalves(PackedConstantLengthReadsSet *half[2], const int32_t sym[2], uint_reads_cnt_max count) {
        uint_reads_cnt_max left = count;
        for (int h = 0; h < 2 && half[h]; h++) {
            const uint_reads_cnt_max cnt = std::min<uint64_t>(left, half[h]->readsCount());
            failOn(pgrc_match_append_reads_packed(ctx, cnt ? half[h]->getPackedRead(0) : nullptr, cnt, sym[h]), "append_reads_packed");
            left -= cnt;
        }
    }

    void HipReadsMatcher::upload() {
        if (uploaded) return;
        uploaded = true;
        // Modes d/i/e index the reads the way DefaultConstantLengthPatternsOnTextHashMatcher::addReadsSetOfPatterns does
        // (matching/ConstantLengthPatternsOnTextHashMatcher.cpp:30): the first getReadsSetProperties()->readsCount
        // reads.  For a PackedConstantLengthReadsSet that is every read; the LQ + N SumOfConstantLengthReadsSets of
        // pgrc-encoder.cpp:349-352 never fills its properties, so the reference indexes -- and matches -- nothing there.
        // Kept bit for bit: only the indexed reads go to the device, the others keep their state.
        deviceReads = readsCount;
        if (hipMode != 'c') {
            const uint_reads_cnt_max indexed = readsSet->getReadsSetProperties()->readsCount;
            if (indexed < deviceReads) deviceReads = indexed;
        }
        if (deviceReads == 0) return;
        PhaseLog log("hand-over of the pseudogenome and the reads");
        const uint_reads_cnt_max readsCount = deviceReads;   // shadows the member for the rest of this function
        failOn(pgrc_match_set_pg_ascii(ctx, pgPtr, pgLength), "set_pg_ascii");
        PackedConstantLengthReadsSet *half[2];
        int32_t sym[2];
        if (packedHalves(half, sym)) {
            failOn(pgrc_match_begin_reads(ctx, readsCount), "begin_reads");
            appendPackedHalves(half, sym, readsCount);
            failOn(pgrc_match_end_reads(ctx), "end_reads");
            packedHandOver = true;
            packedHandOvers++;
        } else {
            // any other ConstantLengthReadsSetInterface: stream the rows through getRead(i, buf) in bounded blocks
            const uint_reads_cnt_max block = 1u << 20;
            std::vector<char> buf((size_t) std::min<uint64_t>(block, readsCount ? readsCount : 1) * readLength);
            failOn(pgrc_match_begin_reads(ctx, readsCount), "begin_reads");
            for (uint_reads_cnt_max first = 0; first < readsCount; first += block) {
                const uint_reads_cnt_max cnt = std::min<uint64_t>(block, (uint64_t) readsCount - first);
                #pragma omp parallel for
                for (uint_reads_cnt_max k = 0; k < cnt; k++)
                    readsSet->getRead(first + k, buf.data() + (size_t) k * readLength);
                failOn(pgrc_match_append_reads_ascii(ctx, buf.data(), cnt), "append_reads_ascii");
            }
            failOn(pgrc_match_end_reads(ctx), "end_reads");
        }
    }

    void HipReadsMatcher::fetchResults() {
        PhaseLog log("result fetch");
        readMatchPos.resize(readsCount);
        readMismatchesCount.resize(readsCount);
        readMatchRC.resize(readsCount, false);
        uint64_t hist[NOT_MATCHED_COUNT + 1] = {0};
        uint64_t matched = 0;
        if (deviceReads) {
            std::vector<uint8_t> rc(deviceReads);
            failOn(pgrc_match_get_results(ctx, readMatchPos.data(), rc.data(), readMismatchesCount.data(), hist,
                                          &matched), "get_results");
            pgrc_match_counters ctr;
            if (pgrc_match_get_counters(ctx, &ctr) == PGRC_OK && ctr.screened == 2) dualRuns++;
            // vector<bool> is bit-packed: threads may only share it along 64-bit word boundaries (the reference's own
            // parallel loop does not respect that, ReadsMatchers.cpp:426-446)
            const uint64_t chunk = 64u * 4096u, total = deviceReads;
            #pragma omp parallel for schedule(static)
            for (uint64_t c0 = 0; c0 < total; c0 += chunk) {
                const uint64_t c1 = std::min<uint64_t>(c0 + chunk, total);
                for (uint64_t i = c0; i < c1; i++)
                    readMatchRC[i] = rc[i] != 0;
            }
        }
        for (uint_reads_cnt_max i = deviceReads; i < readsCount; i++) {   // reads that never went to the device
            hist[readMismatchesCount[i]]++;
            matched += readMismatchesCount[i] != NOT_MATCHED_COUNT;
        }
        matchedReadsCount = matched;
        for (int k = 0; k <= NOT_MATCHED_COUNT; k++)
            matchedCountPerMismatches[k] = hist[k];
    }

    void HipReadsMatcher::initMatching() {
        DefaultReadsMatcher::initMatching();
        readMismatchesCount.clear();
        readMismatchesCount.insert(readMismatchesCount.end(), readsCount, NOT_MATCHED_COUNT);
        upload();
        if (deviceReads) failOn(pgrc_match_init_results(ctx), "init_results");
    }

    void HipReadsMatcher::initMatchingContinuation(DefaultReadsMatcher *pMatcher) {
        AbstractReadsApproxMatcher::initMatchingContinuation(pMatcher); // takes the result vectors over (:111-133)
        upload();
        if (!deviceReads) return;
        std::vector<uint8_t> rc(deviceReads);
        for (uint_reads_cnt_max i = 0; i < deviceReads; i++)
            rc[i] = readMatchRC[i] ? 1 : 0;
        failOn(pgrc_match_set_results(ctx, readMatchPos.data(), rc.data(), readMismatchesCount.data()), "set_results");
    }

    // ---- export support: mismatch lists of all matched reads in one device pass (replaces the per-read
    //      getRead + reverseComplementInPlace + fillEntryWith(Reversed)Mismatches of ReadsMatchers.cpp:548-559) ----
    void HipReadsMatcher::initEntryUpdating() {
        PhaseLog log("bulk mismatch extraction");
        bulkMismatches = false;
        if (!uploaded || deviceReads != readsCount || readsCount == 0) return;   // inherited per-read path
        mmCum.resize((size_t) readsCount + 1);
        failOn(pgrc_match_extract_mismatches(ctx, nullptr, mmCum.data(), nullptr, nullptr), "extract_mismatches");
        const uint64_t total = mmCum[readsCount];
        mmCodes.resize(total);
        mmOffsets.resize(total);
        if (total)
            failOn(pgrc_match_extract_mismatches(ctx, nullptr, mmCum.data(), mmCodes.data(), mmOffsets.data()),
                   "extract_mismatches");
        bulkMismatches = true;
    }

    static inline uint8_t complementValue(uint8_t v) { return v < 4 ? 3 - v : v; }   // A<->T, C<->G, N stays

    void HipReadsMatcher::updateEntry(DefaultReadsListEntry &entry, uint_reads_cnt_max matchIdx, bool revComplPairFile) {
        if (!bulkMismatches) {
            AbstractReadsApproxMatcher::updateEntry(entry, matchIdx, revComplPairFile);
            return;
        }
        // the device lists are in the SE form: "reversed" (original read orientation) iff the read matched the RC
        // strand.  With revComplPairFile the reference wants the reversed form iff rc != (orgIdx odd) (:553); the
        // other form of the same list is its mirror image: reverse order, offset -> L-1-offset, symbols complemented.
        bulkUpdatesServed++;
        const bool rc = readMatchRC[matchIdx];
        const bool wantReversed = revComplPairFile ? (rc != (bool) (entry.idx % 2)) : rc;
        const uint64_t first = mmCum[matchIdx], last = mmCum[matchIdx + 1];
        if (wantReversed == rc) {
            for (uint64_t k = first; k < last; k++)
                entry.addMismatch(mmCodes[k], mmOffsets[k]);
        } else {
            for (uint64_t k = last; k-- > first;) {
                const uint8_t code = mmCodes[k];
                entry.addMismatch((uint8_t) ((complementValue(code >> 4) << 4) + complementValue(code & 15)),
                                  (uint_read_len_max) (readLength - 1 - mmOffsets[k]));
            }
        }
    }

    void HipReadsMatcher::closeEntryUpdating() {
        std::vector<uint64_t>().swap(mmCum);
        std::vector<uint8_t>().swap(mmCodes);
        std::vector<uint16_t>().swap(mmOffsets);
        bulkMismatches = false;
    }

    SeparatedPseudoGenomeOutputBuilder *HipReadsMatcher::createSeparatedPseudoGenomeOutputBuilder(
            SeparatedPseudoGenome *sPg, bool allStreams) {
        if (hipMode != 'e')
            return AbstractReadsApproxMatcher::createSeparatedPseudoGenomeOutputBuilder(sPg, allStreams);
        DefaultReadsListIteratorInterface *rlIt = sPg->getReadsList();
        const bool isRevCompEnabled = allStreams || sPg->getReadsList()->isRevCompEnabled();
        const bool areMismatchesEnabled = allStreams || sPg->getReadsList()->areMismatchesEnabled();
        auto *builder = new SeparatedPseudoGenomeOutputBuilder(!isRevCompEnabled && !this->revComplPg,
                                                               !areMismatchesEnabled);
        builder->setReadsSourceIterator(rlIt);
        builder->copyPseudoGenomeProperties(sPg);
        return builder;
    }

    // ---- export on the device (row f1) ----

    bool HipReadsMatcher::deviceExportPossible(SeparatedPseudoGenome *sPg) const {
        if (getenv("PGRC_HOST_EXPORT")) return false;                    // A/B knob: the inherited export
        if (!uploaded || deviceReads != readsCount || readsCount == 0) return false;
        if (SeparatedPseudoGenomePersistence::enableReadPositionRepresentation ||
            !SeparatedPseudoGenomePersistence::enableRevOffsetMismatchesRepresentation)
            return false;
        ExtendedReadsListWithConstantAccessOption *rl = sPg->getReadsList();
        if (!rl || !rl->misCnt.empty()) return false;                    // old entries with mismatches: inherited path
        if (rl->off.size() < rl->readsCount || rl->orgIdx.size() < rl->readsCount) return false;
        if (!rl->revComp.empty() && rl->revComp.size() < rl->readsCount) return false;
        return true;
    }

    // Makes [src, src + bytes) the CONTENTS of a string stream of the builder without copying it: libstdc++'s
    // basic_stringbuf::setbuf(s, n) adopts the external array as the buffer, n characters long (its str() returns them and
    // a later write grows into a string of its own).  GCC's documented extension -- PgRC builds with GCC only, it sorts
    // with __gnu_parallel.  The array must outlive the stream.  A stream that already holds something, or is not a string
    // stream, gets an ordinary write.  (Written entry by entry, or as one block, a stream of a C3-size export grows by
    // doubling and copies itself as it goes: 0.5 s for the 1.4 GB of the six streams.)
    // Guarded (round 4): only under libstdc++ (__GLIBCXX__), never with PGRC_NO_ADOPT set (tests take the fallback that
    // way), and checked afterwards -- a stream whose put position is not `bytes` after the adoption (another library's
    // setbuf is a no-op) gets the ordinary write after all.
    static void adoptAsContents(std::ostream *dest, const void *src, uint64_t bytes) {
        auto *oss = dynamic_cast<std::ostringstream *>(dest);
        bool adopt = oss && bytes != 0 && oss->tellp() == std::streampos(0) && !getenv("PGRC_NO_ADOPT");
#if !defined(__GLIBCXX__)
        adopt = false;
#endif
        if (adopt) {
            oss->rdbuf()->pubsetbuf(const_cast<char *>((const char *) src), (std::streamsize) bytes);
            oss->seekp(0, std::ios_base::end);
            if (oss->good() && oss->tellp() == std::streampos((std::streamoff) bytes)) return;
            oss->clear();                                                // not adopted: start over with a stream of its own
            oss->str(std::string());
        }
        dest->write((const char *) src, (std::streamsize) bytes);
    }

    // what writeReadEntry (SeparatedPseudoGenomePersistence.cpp:961-989) appends entry by entry, as whole streams; the
    // builder's streams READ FROM `v` until the builder is deleted
    void HipReadsMatcher::appendStreams(SeparatedPseudoGenomeOutputBuilder *builder, const pgrc_export_streams_view &v) {
        const pgrc_export_streams &s = v.s;
        adoptAsContents(builder->rlOffDest, s.off, s.n_entries * s.off_width);
        adoptAsContents(builder->rlOrgIdxDest, s.org_idx, s.n_entries * sizeof(uint_reads_cnt_std));
        if (!builder->disableRevComp)
            adoptAsContents(builder->rlRevCompDest, s.rev_comp, s.n_entries);
        if (!builder->disableMismatches) {
            adoptAsContents(builder->rlMisCntDest, s.mis_cnt, s.n_entries);
            adoptAsContents(builder->rlMisSymDest, s.mis_sym, s.n_mismatches);
            adoptAsContents(builder->rlMisRevOffDest, s.mis_rev_off, s.n_mismatches * s.off_width);
        }
        builder->readsCounter += s.n_entries;
        builder->lastWrittenPos = s.last_pos;
    }

    // The reference sorts the matched reads' INDEXES with a comparator that looks their positions up
    // (`readMatchPos[idx1] < readMatchPos[idx2]`, ReadsMatchers.cpp:573-574).  The same algorithm
    // (parallel_algorithm::sort = __gnu_parallel::sort, sequential std::sort below its size / thread thresholds) on
    // (position, index) pairs, compared by position alone, sees the same outcome for every comparison it makes and so
    // moves its elements the same way: the same permutation, ties included, without the comparator's random accesses.
    // (The size of an element does not enter the algorithm either -- introsort's and the multiway merge's thresholds count
    // elements --, so positions below 2^32 are sorted as 8-byte (position, index) records: less memory to move.)
    namespace {
        template<typename P>
        void sortedByPosition(const vector<uint64_t> &readMatchPos, const std::vector<uint64_t> &first, uint64_t m,
                              std::vector<uint32_t> &order) {
            struct PosIdx { P pos; uint32_t idx; };                  // (no constructor: the array is not filled twice)
            const uint64_t n = readMatchPos.size();
            const int T = (int) first.size() - 1;
            std::unique_ptr<PosIdx[]> byPos(new PosIdx[m ? m : 1]);
            {
                PhaseLog log("  sort: (position, read) pairs");
                // (one iteration per slice: every slice is written whatever the size of the team OpenMP really delivers)
                #pragma omp parallel for schedule(static, 1) num_threads(T)
                for (int t = 0; t < T; t++) {
                    const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
                    uint64_t at = first[t];
                    for (uint64_t i = lo; i < hi; i++)
                        if (readMatchPos[i] != DefaultReadsMatcher::NOT_MATCHED_POSITION) {
                            byPos[at].pos = (P) readMatchPos[i];
                            byPos[at].idx = (uint32_t) i;
                            at++;
                        }
                }
            }
            {
                PhaseLog log("  sort: the reference's sort");
                __gnu_parallel::sort(byPos.get(), byPos.get() + m,
                                     [](const PosIdx &a, const PosIdx &b) -> bool { return a.pos < b.pos; });
            }
            PhaseLog log("  sort: order array");
            order.resize(m);
            #pragma omp parallel for schedule(static)
            for (uint64_t k = 0; k < m; k++) order[k] = byPos[k].idx;
        }
    }

    void HipReadsMatcher::positionOrder(const vector<uint64_t> &readMatchPos, uint_reads_cnt_max matchedReadsCount,
                                        std::vector<uint32_t> &order) {
        // the matched reads in index order -- what the reference's serial loop collects (:567-571) -- made by all threads:
        // matched reads per slice (and the largest position), slice offsets, every slice written at its offset
        const uint64_t n = readMatchPos.size();
        const int T = std::max(1, omp_get_max_threads());
        std::vector<uint64_t> first((size_t) T + 1, 0), top((size_t) T, 0);
        // (one iteration per slice, not one slice per thread number: OpenMP may deliver fewer threads than asked for)
        #pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int t = 0; t < T; t++) {
            const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
            uint64_t cnt = 0, mx = 0;
            for (uint64_t i = lo; i < hi; i++)
                if (readMatchPos[i] != NOT_MATCHED_POSITION) {
                    cnt++;
                    mx = std::max<uint64_t>(mx, readMatchPos[i]);
                }
            first[t + 1] = cnt;
            top[t] = mx;
        }
        for (int k = 0; k < T; k++) first[k + 1] += first[k];
        (void) matchedReadsCount;
        if (*std::max_element(top.begin(), top.end()) <= UINT32_MAX)
            sortedByPosition<uint32_t>(readMatchPos, first, first[T], order);
        else
            sortedByPosition<uint64_t>(readMatchPos, first, first[T], order);
    }

    // everything of the Pg-order export that precedes the builder's own build / compressedBuild: the order of the matched
    // reads, then the merged streams from the device, appended to `builder`
    std::shared_ptr<void> HipReadsMatcher::makePgOrderStreams(SeparatedPseudoGenome *sPg, IndexesMapping *orgIndexesMapping,
                                                              bool revComplPairFile, SeparatedPseudoGenomeOutputBuilder *builder) {
        // The order of the matched reads.  At -t 1 it is the reference's own: its sort leaves reads matched at one position in
        // an order of its own that reaches the archive bytes, so the adapter repeats that sort (positionOrder).  At -t > 1 the
        // reference's archive is not reproducible anyway (racy index build, parallel Pg generator: SURVEY 8c), so nothing
        // depends on that tie order and the library makes the order on the device -- ascending position, ties by read index
        // (round 4; C3 size: 0.67 s of host sort -> a few ms).  PGRC_DEVICE_SORT=0 / 1 forces the one or the other.
        std::vector<uint32_t> order;
        const char *ds = getenv("PGRC_DEVICE_SORT");
        const bool deviceSort = (ds ? ds[0] == '1' : PgHelpers::numberOfThreads > 1) && pgLength < (1ull << 32);
        if (!deviceSort) {
            PhaseLog log("export: position sort");
            positionOrder(readMatchPos, matchedReadsCount, order);
        }
        auto st = std::make_shared<pgrc_export_streams_view>();
        PhaseLog log("export: streams from the device");
        std::vector<uint32_t> readOrg(readsCount);
        {
            PhaseLog log2("  streams: original indexes");
            #pragma omp parallel for
            for (uint_reads_cnt_max i = 0; i < readsCount; i++)
                readOrg[i] = orgIndexesMapping->getReadOriginalIndex(i);
        }
        ExtendedReadsListWithConstantAccessOption *rl = sPg->getReadsList();
        pgrc_export_pg_order_args a;
        a.order = deviceSort ? nullptr : order.data();          // (an empty vector's data() may be NULL: fine with n_matched == 0)
        a.n_matched = deviceSort ? 0 : order.size();
        a.order_on_device = deviceSort ? 1 : 0;
        a.read_org_idx = readOrg.data();
        a.list_off = rl->off.data();
        a.list_org_idx = rl->orgIdx.data();
        a.list_rev_comp = rl->revComp.empty() ? nullptr : rl->revComp.data();
        a.list_count = rl->readsCount;
        a.rev_compl_pair_file = revComplPairFile ? 1 : 0;
        a.byte_per_read_length = PgHelpers::bytePerReadLengthMode ? 1 : 0;
        {
            PhaseLog log2("  streams: library call");
            failOn(pgrc_match_export_pg_order(ctx, &a, &st->s), "export_pg_order");
        }
        {
            PhaseLog log2("  streams: append to the builder");
            appendStreams(builder, *st);
        }
        return st;
    }

    void HipReadsMatcher::exportMatchesInPgOrderOnDevice(SeparatedPseudoGenome *sPg, ostream &pgrcOut,
                                                         uint8_t compressionLevel, const string &outPgPrefix,
                                                         IndexesMapping *orgIndexesMapping, bool pairFileMode,
                                                         bool revComplPairFile) {
        if (!deviceExportPossible(sPg)) {
            exportMatchesInPgOrder(sPg, pgrcOut, compressionLevel, outPgPrefix, orgIndexesMapping, pairFileMode,
                                   revComplPairFile);
            return;
        }
        deviceExports++;
        SeparatedPseudoGenomeOutputBuilder *builder = this->createSeparatedPseudoGenomeOutputBuilder(sPg);
        const std::shared_ptr<void> streams = makePgOrderStreams(sPg, orgIndexesMapping, revComplPairFile, builder);   // (outlives the builder)
        PhaseLog log("export: the reference's stream compression");
        builder->build(outPgPrefix);
        builder->compressedBuild(pgrcOut, compressionLevel);
        if (pairFileMode)
            builder->updateOriginalIndexesIn(sPg);
        delete (builder);
    }

    void HipReadsMatcher::exportMatchesInOriginalOrderOnDevice(SeparatedPseudoGenome *sPg, ostream &pgrcOut,
                                                               uint8_t compressionLevel, const string &outPgPrefix,
                                                               IndexesMapping *orgIndexesMapping, bool pairFileMode,
                                                               bool revComplPairFile) {
        if (!deviceExportPossible(sPg)) {
            exportMatchesInOriginalOrder(sPg, pgrcOut, compressionLevel, outPgPrefix, orgIndexesMapping, pairFileMode,
                                         revComplPairFile);
            return;
        }
        deviceExports++;
        // What the reference's walk over the original read order (ReadsMatchers.cpp:600-667) produces, without the walk:
        // the entry list is a table over the original indexes and is made on the device
        // (pgrc_match_export_original_order); the host keeps only the caller-visible side effect, the reads list's
        // `pos` by original index (:603-610, :663, :673).
        ExtendedReadsListWithConstantAccessOption *const list = sPg->getReadsList();
        const uint64_t allReads = orgIndexesMapping->getReadsTotalCount();
        std::vector<uint_pg_len_max> pgPosOfOrg(allReads, (uint_pg_len_max) -1);
        {
            // entries already on the pseudogenome: position = running sum of the offset deltas
            const size_t listed = list->readsCount;
            std::vector<uint_pg_len_max> listPos(listed);
            std::inclusive_scan(list->off.begin(), list->off.begin() + listed, listPos.begin(), std::plus<uint_pg_len_max>(),
                                (uint_pg_len_max) 0);
            #pragma omp parallel for
            for (size_t k = 0; k < listed; k++)
                pgPosOfOrg[list->orgIdx[k]] = listPos[k];
        }
        std::vector<uint32_t> readOrg(readsCount);
        #pragma omp parallel for
        for (uint_reads_cnt_max i = 0; i < readsCount; i++) {
            readOrg[i] = orgIndexesMapping->getReadOriginalIndex(i);
            if (readMatchPos[i] != NOT_MATCHED_POSITION)
                pgPosOfOrg[readOrg[i]] = readMatchPos[i];
        }
        // the builder must not see the old list's offsets / indexes (the reference drops them before it creates one)
        list->orgIdx.clear();
        list->off.clear();
        SeparatedPseudoGenomeOutputBuilder *builder = this->createSeparatedPseudoGenomeOutputBuilder(sPg);
        pgrc_export_streams_view st;                                     // (outlives the builder, whose streams read from it)
        {
            PhaseLog log("export: entry list and streams from the device");
            pgrc_export_original_order_args a;
            a.read_org_idx = readOrg.data();
            a.reads_total_count = allReads;
            a.pair_file_mode = pairFileMode ? 1 : 0;
            a.rev_compl_pair_file = revComplPairFile ? 1 : 0;
            a.byte_per_read_length = PgHelpers::bytePerReadLengthMode ? 1 : 0;
            failOn(pgrc_match_export_original_order(ctx, &a, &st.s), "export_original_order");
            appendStreams(builder, st);
        }
        PhaseLog log("export: the reference's stream compression");
        builder->build(outPgPrefix);
        builder->compressedBuild(pgrcOut, compressionLevel, true);
        delete (builder);
        list->pos = std::move(pgPosOfOrg);
    }

    void HipReadsMatcher::executeMatching(bool revCompMode) {
        if (deviceReads) failOn(pgrc_match_run_pass(ctx, revCompMode ? 1 : 0), "run_pass");
        fetchResults();
    }

    // matchConstantLengthReads (ReadsMatchers.cpp:162-172) with its steps overlapped instead of in turn -- the library's
    // pipelined hand-over (include/pgrc_match.h): both index builds start as soon as the text is on the device, every
    // block of packed rows is matched while the next one is copied, results come back block by block.  Taken where it
    // applies (mode c, both strands, first phase, one device, packed sets); PGRC_NO_STREAM=1 keeps the steps in turn.
    bool HipReadsMatcher::matchStreamed() {
        PackedConstantLengthReadsSet *half[2];
        int32_t sym[2];
        if (hipMode != 'c' || !revComplPg || minMismatches != 0 || uploaded || readsCount == 0 || getenv("PGRC_DEVICES") ||
            getenv("PGRC_NO_STREAM") || !packedHalves(half, sym))
            return false;
        PhaseLog log("streamed hand-over + matching + results");
        std::vector<uint8_t> rc;
        // the result vectors of the reference (initMatching: 0.9 GB filled by one thread at C3 size) are made on two helper
        // threads while this one hands the pseudogenome over and starts the index builds
        // (their pages are touched by all threads first -- the fills below then run at memory speed, not at one thread's
        //  page-fault rate; clear() + insert() of initMatching keep the reserved storage)
        readMatchPos.clear();
        readMatchPos.reserve(readsCount);
        readMismatchesCount.clear();
        readMismatchesCount.reserve(readsCount);
        rc.reserve(readsCount);
        // (Touching reserved-but-unsized vector storage is outside what the standard promises; libstdc++'s vectors keep the
        //  storage through clear() / assign() within capacity, which is all this relies on.  Guarded: libstdc++ only, and
        //  PGRC_NO_PRETOUCH skips it -- it is a page-fault optimisation, nothing depends on it.)
#if defined(__GLIBCXX__)
        if (!getenv("PGRC_NO_PRETOUCH"))
#else
        if (false)
#endif
        {
            struct Area { volatile char *p; size_t bytes; } areas[3] = {{(volatile char *) readMatchPos.data(), (size_t) readsCount * sizeof(uint64_t)},
                                                                       {(volatile char *) readMismatchesCount.data(), (size_t) readsCount},
                                                                       {(volatile char *) rc.data(), (size_t) readsCount}};
            for (const Area &ar : areas) {
                const int64_t pages = (int64_t) ((ar.bytes + 4095) / 4096);
                #pragma omp parallel for schedule(static)
                for (int64_t pg = 0; pg < pages; pg++) ar.p[(size_t) pg * 4096] = 0;
            }
        }
        std::thread fillA([&]() { DefaultReadsMatcher::initMatching(); });
        std::thread fillB([&]() { readMismatchesCount.assign(readsCount, NOT_MATCHED_COUNT); rc.resize(readsCount); });
        uploaded = true;
        deviceReads = readsCount;
        bool ahead = false;
        int pgErr;
        {
            PhaseLog log2("  library: pseudogenome up, index builds started");
            pgErr = pgrc_match_set_pg_ascii(ctx, pgPtr, pgLength);
            if (!pgErr) ahead = pgrc_match_prepare_index(ctx, 1) == PGRC_OK;   // (no room for both indexes: the plain steps below)
        }
        {
            PhaseLog log2("  host: result vectors (the wait for them)");
            fillA.join();
            fillB.join();
        }
        failOn(pgErr, "set_pg_ascii");
        PhaseLog log3("  library: reads up, matching, results down (+ strand flags)");
        failOn(pgrc_match_begin_reads(ctx, readsCount), "begin_reads");
        const bool streamed = ahead && pgrc_match_stream_begin(ctx, readMatchPos.data(), rc.data(), readMismatchesCount.data()) == PGRC_OK;
        appendPackedHalves(half, sym, readsCount);
        failOn(pgrc_match_end_reads(ctx), "end_reads");
        packedHandOver = true;
        packedHandOvers++;
        if (!streamed) {
            failOn(pgrc_match_init_results(ctx), "init_results");
            failOn(pgrc_match_run(ctx, 1), "run");
            fetchResults();
            return true;
        }
        uint64_t hist[NOT_MATCHED_COUNT + 1] = {0}, matched = 0;
        failOn(pgrc_match_stream_end(ctx, hist, &matched), "stream_end");
        streamedRuns++;
        pgrc_match_counters ctr;
        if (pgrc_match_get_counters(ctx, &ctr) == PGRC_OK && ctr.screened == 2) dualRuns++;
        const uint64_t chunk = 64u * 4096u, total = readsCount;     // (vector<bool>: threads share it along word boundaries only)
        #pragma omp parallel for schedule(static)
        for (uint64_t c0 = 0; c0 < total; c0 += chunk) {
            const uint64_t c1 = std::min<uint64_t>(c0 + chunk, total);
            for (uint64_t i = c0; i < c1; i++)
                readMatchRC[i] = rc[i] != 0;
        }
        matchedReadsCount = matched;
        for (int k = 0; k <= NOT_MATCHED_COUNT; k++)
            matchedCountPerMismatches[k] = hist[k];
        return true;
    }

    void HipReadsMatcher::matchConstantLengthReadsOnDevice() {
        if (matchStreamed()) return;
        initMatching();
        {
            PhaseLog log("device run (both strands)");
            if (deviceReads) failOn(pgrc_match_run(ctx, revComplPg ? 1 : 0), "run");
        }
        fetchResults();
    }

    void HipReadsMatcher::continueMatchingConstantLengthReadsOnDevice(DefaultReadsMatcher *pMatcher) {
        initMatchingContinuation(pMatcher);
        if (deviceReads) failOn(pgrc_match_run(ctx, revComplPg ? 1 : 0), "run");
        fetchResults();
    }
}
