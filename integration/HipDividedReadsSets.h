// HipDividedReadsSets -- the reference-side binding of libpgrc_match.so for MAKING the packed read sets (SURVEY.md
// section 8 row f3): what DividedPCLReadsSets::getQualityDivisionBasedReadsSets / getSimpleDividedPCLReadsSets
// (readsset/DividedPCLReadsSets.cpp:59-114) do record by record -- which set a read goes to, its packed row, its entry
// in the LQ / N mapping -- done per batch of records on the MI355X through the C ABI of include/pgrc_reads.h.
// The FASTQ parsing stays the reference's iterator; the results are the reference's own objects.
//
// NEW code for the PgRC tree (not part of the reference), compiled against the reference's headers.  In a PgRC tree the
// maintainer adds `friend class PgTools::HipDividedReadsSets;` to DividedPCLReadsSets (its two mappings have no setter);
// INTEGRATION.md shows the call-site change in pgrc/pgrc-encoder.cpp.
#ifndef PGTOOLS_HIPDIVIDEDREADSSETS_H
#define PGTOOLS_HIPDIVIDEDREADSSETS_H

#include "readsset/DividedPCLReadsSets.h"

namespace PgTools {

    class HipDividedReadsSets {
    public:
        // same arguments and result as DividedPCLReadsSets::getQualityDivisionBasedReadsSets (DividedPCLReadsSets.h:38-40)
        static DividedPCLReadsSets *getQualityDivisionBasedReadsSets(
                ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength,
                double error_limit, bool simplified_suffix_mode, bool separateNReadsSet = false, bool nReadsLQ = false);

        // ... as DividedPCLReadsSets::getSimpleDividedPCLReadsSets (DividedPCLReadsSets.h:42-44)
        static DividedPCLReadsSets *getSimpleDividedPCLReadsSets(
                ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength,
                bool separateNReadsSet, bool nReadsLQ);

        // The same sets straight from the FASTQ file(s): what the two factories above make of
        // ReadsSetPersistence::createManagedReadsIterator(srcFastqFile, pairFastqFile, revComplPairFile)
        // (pgrc-encoder.cpp:255-256, ReadsSetPersistence.cpp:20-56), with the lines of the text found on the device too: the
        // files are read in pieces and go up as they are.  error_limit = 1 gives getSimpleDividedPCLReadsSets' result.
        static DividedPCLReadsSets *getQualityDivisionBasedReadsSetsFromFastq(
                const string &srcFastqFile, const string &pairFastqFile, bool revComplPairFile, uint_read_len_max readLength,
                double error_limit, bool simplified_suffix_mode, bool separateNReadsSet = false, bool nReadsLQ = false);

        static uint64_t batchesServed;     // diagnostics / tests
    };
}

#endif //PGTOOLS_HIPDIVIDEDREADSSETS_H
