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

#include "matching/ReadsMatchers.h"

struct pgrc_match_ctx;

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
        void fetchResults();

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

        // entries whose mismatch list was served from the device extraction (diagnostics / tests)
        static uint64_t bulkUpdatesServed;
        // uploads that took the reference's packed rows as they are (no getRead)
        static uint64_t packedHandOvers;
    };
}

#endif //PGTOOLS_HIPREADSMATCHER_H
