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
#include <parallel/algorithm>

struct pgrc_export_streams_view {
    const pgrc_export_streams *s;
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

    // what writeReadEntry (SeparatedPseudoGenomePersistence.cpp:961-989) appends entry by entry, as whole streams
    void HipReadsMatcher::appendStreams(SeparatedPseudoGenomeOutputBuilder *builder, const pgrc_export_streams_view &v) {
        const pgrc_export_streams &s = *v.s;
        builder->rlOffDest->write((const char *) s.off, (std::streamsize) (s.n_entries * s.off_width));
        builder->rlOrgIdxDest->write((const char *) s.org_idx, (std::streamsize) (s.n_entries * sizeof(uint_reads_cnt_std)));
        if (!builder->disableRevComp)
            builder->rlRevCompDest->write((const char *) s.rev_comp, (std::streamsize) s.n_entries);
        if (!builder->disableMismatches) {
            builder->rlMisCntDest->write((const char *) s.mis_cnt, (std::streamsize) s.n_entries);
            builder->rlMisSymDest->write((const char *) s.mis_sym, (std::streamsize) s.n_mismatches);
            builder->rlMisRevOffDest->write((const char *) s.mis_rev_off, (std::streamsize) (s.n_mismatches * s.off_width));
        }
        builder->readsCounter += s.n_entries;
        builder->lastWrittenPos = s.last_pos;
    }

    // The reference sorts the matched reads' INDEXES with a comparator that looks their positions up
    // (`readMatchPos[idx1] < readMatchPos[idx2]`, ReadsMatchers.cpp:573-574).  The same algorithm
    // (parallel_algorithm::sort = __gnu_parallel::sort, sequential std::sort below its size / thread thresholds) on
    // (position, index) pairs, compared by position alone, sees the same outcome for every comparison it makes and so
    // moves its elements the same way: the same permutation, ties included, without the comparator's random accesses.
    void HipReadsMatcher::positionOrder(const vector<uint64_t> &readMatchPos, uint_reads_cnt_max matchedReadsCount,
                                        std::vector<uint32_t> &order) {
        typedef std::pair<uint64_t, uint_reads_cnt_max> PosIdx;
        std::vector<PosIdx> byPos;
        byPos.reserve(matchedReadsCount);
        const uint_reads_cnt_max n = readMatchPos.size();
        for (uint_reads_cnt_max i = 0; i < n; i++)
            if (readMatchPos[i] != NOT_MATCHED_POSITION)
                byPos.emplace_back(readMatchPos[i], i);
        __gnu_parallel::sort(byPos.begin(), byPos.end(),
                             [](const PosIdx &a, const PosIdx &b) -> bool { return a.first < b.first; });
        order.resize(byPos.size());
        for (size_t k = 0; k < byPos.size(); k++) order[k] = byPos[k].second;
    }

    // everything of the Pg-order export that precedes the builder's own build / compressedBuild: the order of the matched
    // reads, then the merged streams from the device, appended to `builder`
    void HipReadsMatcher::makePgOrderStreams(SeparatedPseudoGenome *sPg, IndexesMapping *orgIndexesMapping, bool revComplPairFile,
                                             SeparatedPseudoGenomeOutputBuilder *builder) {
        std::vector<uint32_t> order;
        {
            PhaseLog log("export: position sort");
            positionOrder(readMatchPos, matchedReadsCount, order);
        }
        std::vector<uint32_t> readOrg(readsCount);
        #pragma omp parallel for
        for (uint_reads_cnt_max i = 0; i < readsCount; i++)
            readOrg[i] = orgIndexesMapping->getReadOriginalIndex(i);
        pgrc_export_streams st;
        PhaseLog log("export: streams from the device");
        ExtendedReadsListWithConstantAccessOption *rl = sPg->getReadsList();
        pgrc_export_pg_order_args a;
        a.order = order.data();
        a.n_matched = order.size();
        a.read_org_idx = readOrg.data();
        a.list_off = rl->off.data();
        a.list_org_idx = rl->orgIdx.data();
        a.list_rev_comp = rl->revComp.empty() ? nullptr : rl->revComp.data();
        a.list_count = rl->readsCount;
        a.rev_compl_pair_file = revComplPairFile ? 1 : 0;
        a.byte_per_read_length = PgHelpers::bytePerReadLengthMode ? 1 : 0;
        failOn(pgrc_match_export_pg_order(ctx, &a, &st), "export_pg_order");
        pgrc_export_streams_view v{&st};
        appendStreams(builder, v);
        pgrc_match_free_export(&st);
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
        makePgOrderStreams(sPg, orgIndexesMapping, revComplPairFile, builder);
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
        {
            PhaseLog log("export: entry list and streams from the device");
            pgrc_export_original_order_args a;
            a.read_org_idx = readOrg.data();
            a.reads_total_count = allReads;
            a.pair_file_mode = pairFileMode ? 1 : 0;
            a.rev_compl_pair_file = revComplPairFile ? 1 : 0;
            a.byte_per_read_length = PgHelpers::bytePerReadLengthMode ? 1 : 0;
            pgrc_export_streams st;
            failOn(pgrc_match_export_original_order(ctx, &a, &st), "export_original_order");
            pgrc_export_streams_view v{&st};
            appendStreams(builder, v);
            pgrc_match_free_export(&st);
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
        DefaultReadsMatcher::initMatching();
        readMismatchesCount.assign(readsCount, NOT_MATCHED_COUNT);
        uploaded = true;
        deviceReads = readsCount;
        failOn(pgrc_match_set_pg_ascii(ctx, pgPtr, pgLength), "set_pg_ascii");
        const bool ahead = pgrc_match_prepare_index(ctx, 1) == PGRC_OK;   // (no room for both indexes: the plain steps below)
        failOn(pgrc_match_begin_reads(ctx, readsCount), "begin_reads");
        std::vector<uint8_t> rc(readsCount);
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
