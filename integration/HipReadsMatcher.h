// HipReadsMatcher -- the reference-side binding of libpgrc_match.so: a matcher class that plugs into
// PgRC's DefaultReadsMatcher seam (matching/ReadsMatchers.h:24-83, :109-143) and forwards the matching to
// the MI355X library through the C ABI of include/pgrc_match.h.
//
// This file is NEW code for the PgRC tree (it is not part of the reference); it is compiled against the
// reference's headers.  Everything downstream of the matcher -- getMatchedReadsBitmap, exportMatchesInPgOrder /
// exportMatchesInOriginalOrder, updateEntry -- is inherited unchanged and consumes the result fields this class
// fills (readMatchPos, readMatchRC, readMismatchesCount, matchedReadsCount, matchedCountPerMismatches).
#ifndef PGTOOLS_HIPREADSMATCHER_H
#define PGTOOLS_HIPREADSMATCHER_H

#include <map>
#include <memory>
#include <string>

#include "matching/ReadsMatchers.h"

struct pgrc_match_ctx;
struct pgrc_export_streams_view;

namespace PgTools {

    class HipReadsMatcher : public AbstractReadsApproxMatcher {
    private:
        pgrc_match_ctx *ctx = nullptr;
        const char hipMode;                  // 'c', 'd', 'i' or 'e' (the matcher the reference would have built)
        const uint16_t seedChars;
        bool uploaded = false;
        bool packedHandOver = false;         // the reads went over in the reference's own packed layout (no getRead)
        uint_reads_cnt_max deviceReads = 0;  // reads that take part (all in mode c; see upload() for modes d/i/e)

        // mismatch lists of all matched reads, filled by initEntryUpdating() (CSR: mmCum[i] .. mmCum[i+1])
        bool bulkMismatches = false;
        std::vector<uint64_t> mmCum;
        std::vector<uint8_t> mmCodes;
        std::vector<uint16_t> mmOffsets;

        void failOn(int code, const char *what);
        void upload();
        bool packedHalves(PgReadsSet::PackedConstantLengthReadsSet *half[2], int32_t sym[2]) const;
        void appendPackedHalves(PgReadsSet::PackedConstantLengthReadsSet *half[2], const int32_t sym[2], uint_reads_cnt_max count);
        bool matchStreamed();
        void fetchResults();
        bool deviceExportPossible(SeparatedPseudoGenome *sPg) const;
        void appendStreams(SeparatedPseudoGenomeOutputBuilder *builder, const struct pgrc_export_streams_view &s);

    protected:
        void initMatching() override;
        void initMatchingContinuation(DefaultReadsMatcher *pMatcher) override;
        // one pass on the device; the text handed to the library at initMatching() is the forward one, the
        // reverse complement is derived on the GPU (the in-place RC of pgPtr by the caller is not needed).
        void executeMatching(bool revCompMode = false) override;

        // export hooks (ReadsMatchers.h:52-54): the mismatch lists come from one device pass over all matched reads
        // (pgrc_match_extract_mismatches) instead of one getRead + compare per read on the host
        void initEntryUpdating() override;
        void updateEntry(DefaultReadsListEntry &entry, uint_reads_cnt_max matchIdx, bool revComplPairFile) override;
        void closeEntryUpdating() override;

        // kind 'e' stands in for DefaultReadsExactMatcher, whose builder drops the mismatch streams on
        // `!areMismatchesEnabled` alone (ReadsMatchers.cpp:534-544); the inherited approx rule (:521-532) also asks
        // for maxMismatches == 0
        SeparatedPseudoGenomeOutputBuilder *createSeparatedPseudoGenomeOutputBuilder(
                SeparatedPseudoGenome *sPg, bool allStreams = true) override;

    public:
        HipReadsMatcher(char *pgPtr, const uint_pg_len_max pgLength, bool revComplPg,
                        ConstantLengthReadsSetInterface *readsSet, uint32_t matchPrefixLength,
                        uint16_t readsExactMatchingChars, uint8_t maxMismatches, uint8_t minMismatches,
                        char hipMode);

        ~HipReadsMatcher() override;

        // Same contract as DefaultReadsMatcher::matchConstantLengthReads() (ReadsMatchers.cpp:162-172) without
        // its two host-side sweeps PgHelpers::reverseComplementInPlace(pgPtr, pgLength).
        void matchConstantLengthReadsOnDevice();

        // Same contract as continueMatchingConstantLengthReads (ReadsMatchers.cpp:174-184).
        void continueMatchingConstantLengthReadsOnDevice(DefaultReadsMatcher *pMatcher);

        // Same contracts -- and the same archive bytes -- as exportMatchesInPgOrder / exportMatchesInOriginalOrder
        // (ReadsMatchers.cpp:563-675).  The host keeps the reference's sort of the matched reads (its tie order is
        // an artefact of std::sort / __gnu_parallel::sort that reaches the archive) but runs it on compact
        // (position, index) pairs; the merge with the reads list already on the pseudogenome, the offset deltas,
        // the mismatch lists and their reverse-offset coding come from the device as whole streams
        // (pgrc_match_export_pg_order / pgrc_match_export_entries) and go into the builder's destinations in one
        // write each, instead of seven stream writes per entry.  Whatever the device path does not cover (a reads
        // list that carries mismatches, the position / plain-offset representations of PgRC.cpp:158-162) takes the
        // inherited export with the bulk mismatch lists.  A matcher over several devices exports from its first
        // device (the library gathers the shards' results and reads there).
        void exportMatchesInPgOrderOnDevice(SeparatedPseudoGenome *sPg, ostream &pgrcOut, uint8_t compressionLevel,
                                            const string &outPgPrefix, IndexesMapping *orgIndexesMapping,
                                            bool pairFileMode, bool revComplPairFile);
        void exportMatchesInOriginalOrderOnDevice(SeparatedPseudoGenome *sPg, ostream &pgrcOut, uint8_t compressionLevel,
                                                  const string &outPgPrefix, IndexesMapping *orgIndexesMapping,
                                                  bool pairFileMode, bool revComplPairFile);
        // (returns the keeper of the library's stream buffers: the builder's streams read from them, keep it until the
        //  builder is deleted)
        std::shared_ptr<void> makePgOrderStreams(SeparatedPseudoGenome *sPg, IndexesMapping *orgIndexesMapping,
                                                 bool revComplPairFile, SeparatedPseudoGenomeOutputBuilder *builder);
        // exports that took the device path (diagnostics / tests)
        static uint64_t deviceExports;
        // seconds spent per phase of the adapter, summed over calls (diagnostics; PGRC_HIP_TIMING=1 also logs them)
        static std::map<std::string, double> phaseSeconds;
        // the matched reads in the order exportMatchesInPgOrder walks them (ReadsMatchers.cpp:567-574): the reference's
        // sort algorithm with the reference's comparator outcome, run on (position, index) pairs
        static void positionOrder(const vector<uint64_t> &readMatchPos, uint_reads_cnt_max matchedReadsCount,
                                  std::vector<uint32_t> &order);

        // entries whose mismatch list was served from the device extraction (diagnostics / tests)
        static uint64_t bulkUpdatesServed;
        // uploads that took the reference's packed rows as they are (no getRead)
        static uint64_t packedHandOvers;
        // device runs that took the dual kernel (one query per read over both strands; diagnostics / tests)
        static uint64_t dualRuns;
        // ... that overlapped hand-over, matching and result fetch (matchStreamed)
        static uint64_t streamedRuns;
    };
}

#endif //PGTOOLS_HIPREADSMATCHER_H
