// DividedPCLReadsSets keeps its LQ / N mappings private and has no setter: only its own static factories fill them
// (readsset/DividedPCLReadsSets.cpp:94-96).  In a PgRC tree the maintainer adds
// `friend class PgTools::HipDividedReadsSets;` to that class.  Built against an UNPATCHED tree
// (-DPGRC_UNPATCHED_TREE; oracle/Makefile compiles the reference's sources where they lie and must not edit them) this
// one translation unit opens the class up instead; layout and ABI are unaffected.
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
#include "readsset/DividedPCLReadsSets.h"
#undef private
#endif

#include "HipDividedReadsSets.h"

#include <cstdio>
#include <cstdlib>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "pgrc_reads.h"
#include "readsset/persistance/ReadsSetPersistence.h"

namespace PgTools {

    uint64_t HipDividedReadsSets::batchesServed = 0;

    namespace {
        void failOn(int code, const pgrc_divider *d, const char *what) {
            if (code == PGRC_OK) return;
            fprintf(stderr, "HipDividedReadsSets: %s failed (%d): %s\n", what, code, pgrc_divider_last_error(d));
            exit(EXIT_FAILURE);
        }

        // the packed rows of one set's part of a batch, behind the reads the set already holds
        void appendRows(PackedConstantLengthReadsSet *set, const uint8_t *rows, uint64_t count) {
            if (!count) return;
            const uint_reads_cnt_max had = set->readsCount();
            set->resize(had + (uint_reads_cnt_max) count);
            set->copyPackedRead(rows, had, (uint_reads_cnt_max) count);
        }
    }

    DividedPCLReadsSets *HipDividedReadsSets::getQualityDivisionBasedReadsSets(
            ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength, double error_limit,
            bool simplified_suffix_mode, bool separateNReadsSet, bool nReadsLQ) {
        DividedPCLReadsSets *readsSets = new DividedPCLReadsSets(readLength, separateNReadsSet, nReadsLQ);
        pgrc_divide_params prm;
        prm.read_len = readLength;
        prm.error_limit = error_limit;
        prm.simplified_suffix_mode = simplified_suffix_mode ? 1 : 0;
        prm.separate_n_reads_set = separateNReadsSet ? 1 : 0;
        prm.n_reads_lq = nReadsLQ ? 1 : 0;
        prm.device = -1;
        pgrc_divider *divider = nullptr;
        failOn(pgrc_divider_create(&prm, &divider), nullptr, "divider_create");
        const bool byQuality = error_limit < 1;
        // records per batch: two row arrays of this many rows on the host and in HBM (PGRC_DIVIDE_BATCH: tests use tiny ones)
        uint64_t batch = 4u << 20;
        if (const char *v = getenv("PGRC_DIVIDE_BATCH")) batch = std::max<uint64_t>(1, strtoull(v, nullptr, 10));
        std::vector<char> rows(batch * readLength), quals(byQuality ? batch * readLength : 0);
        vector<uint_reads_cnt_max> lqMapping, nMapping;
        uint_reads_cnt_max seen = 0;
        bool more = true;
        while (more) {
            uint64_t cnt = 0;
            while (cnt < batch && (more = readsIt->moveNext())) {
                if (readsIt->getReadLength() != readLength) {               // addRead, PackedConstantLengthReadsSet.cpp:37-40
                    fprintf(stderr, "Unsupported variable length reads.\n");
                    exit(EXIT_FAILURE);
                }
                memcpy(rows.data() + cnt * readLength, readsIt->getRead().data(), readLength);
                if (byQuality) memcpy(quals.data() + cnt * readLength, readsIt->getQualityInfo().data(), readLength);
                cnt++;
            }
            if (!cnt) break;
            pgrc_divided_reads part;                        // (the divider's arrays: valid until its next run)
            failOn(pgrc_divider_run(divider, rows.data(), byQuality ? quals.data() : nullptr, cnt, &part), divider, "divider_run");
            batchesServed++;
            appendRows(readsSets->getHqReadsSet(), part.hq_rows, part.n_hq);
            appendRows(readsSets->getLqReadsSet(), part.lq_rows, part.n_lq);
            if (separateNReadsSet) appendRows(readsSets->getNReadsSet(), part.n_rows, part.n_n);
            for (uint64_t k = 0; k < part.n_lq; k++) lqMapping.push_back(seen + part.lq_index[k]);
            for (uint64_t k = 0; k < part.n_n; k++) nMapping.push_back(seen + part.n_index[k]);
            seen += (uint_reads_cnt_max) cnt;
        }
        pgrc_divider_destroy(divider);
        cout << "Filtered " << (lqMapping.size() + nMapping.size());
        if (separateNReadsSet)
            cout << " (including " << nMapping.size() << " containing N)";
        cout << " reads (out of " << seen << ") on the device." << endl;
        readsSets->lqMapping = new VectorMapping(std::move(lqMapping), seen);
        if (separateNReadsSet)
            readsSets->nMapping = new VectorMapping(std::move(nMapping), seen);
        return readsSets;
    }

    DividedPCLReadsSets *HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq(
            const string &srcFastqFile, const string &pairFastqFile, bool revComplPairFile, uint_read_len_max readLength,
            double error_limit, bool simplified_suffix_mode, bool separateNReadsSet, bool nReadsLQ) {
        {   // FASTQ only (ManagedReadsSetIterator picks the format by the first byte, ReadsSetPersistence.cpp:36-46): FASTA and
            // plain concatenated reads take the reference's iterator and the record-by-record form above
            FILE *probe = fopen(srcFastqFile.c_str(), "rb");
            const int first = probe ? fgetc(probe) : EOF;
            if (probe) fclose(probe);
            if (first != '@') {
                ReadsSourceIteratorTemplate<uint_read_len_max> *it =
                        ReadsSetPersistence::createManagedReadsIterator(srcFastqFile, pairFastqFile, revComplPairFile);
                DividedPCLReadsSets *sets = getQualityDivisionBasedReadsSets(it, readLength, error_limit, simplified_suffix_mode,
                                                                            separateNReadsSet, nReadsLQ);
                delete it;
                return sets;
            }
        }
        DividedPCLReadsSets *readsSets = new DividedPCLReadsSets(readLength, separateNReadsSet, nReadsLQ);
        pgrc_divide_params prm;
        prm.read_len = readLength;
        prm.error_limit = error_limit;
        prm.simplified_suffix_mode = simplified_suffix_mode ? 1 : 0;
        prm.separate_n_reads_set = separateNReadsSet ? 1 : 0;
        prm.n_reads_lq = nReadsLQ ? 1 : 0;
        prm.device = -1;
        pgrc_divider *divider = nullptr;
        failOn(pgrc_divider_create(&prm, &divider), nullptr, "divider_create");
        const bool paired = !pairFastqFile.empty();
        // the files are mapped and handed over window by window: no read() into a buffer of ours, the copy to the device is
        // the only pass over the text (PGRC_FASTQ_PIECE bytes of each file per call; tests use tiny ones).  A window starts
        // where the records taken so far end: what a piece ends with -- a record cut in two, records whose mates have not
        // come yet -- is simply seen again.
        struct Mapped { const char *p = nullptr; size_t bytes = 0, at = 0; } in[2];
        const string *names[2] = {&srcFastqFile, &pairFastqFile};
        for (int f = 0; f < (paired ? 2 : 1); f++) {
            const int fd = open(names[f]->c_str(), O_RDONLY);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0) {
                fprintf(stderr, "cannot open reads file %s\n", names[f]->c_str());
                exit(EXIT_FAILURE);
            }
            in[f].bytes = (size_t) st.st_size;
            if (in[f].bytes) {
                void *m = mmap(nullptr, in[f].bytes, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) {
                    fprintf(stderr, "cannot map reads file %s\n", names[f]->c_str());
                    exit(EXIT_FAILURE);
                }
                (void) madvise(m, in[f].bytes, MADV_SEQUENTIAL);
                in[f].p = (const char *) m;
            }
            close(fd);
        }
        size_t piece = 256u << 20;
        if (const char *v = getenv("PGRC_FASTQ_PIECE")) piece = std::max<size_t>(64, strtoull(v, nullptr, 10));
        vector<uint_reads_cnt_max> lqMapping, nMapping;
        uint_reads_cnt_max seen = 0;
        size_t window[2] = {piece, piece};                  // grows when a window holds no whole record (or no mate for one)
        for (;;) {
            size_t len[2];
            int finalBits = 0;                              // bit 1 / bit 2: the first / the second file ends in this window
            for (int f = 0; f < (paired ? 2 : 1); f++) {
                len[f] = std::min(window[f], in[f].bytes - in[f].at);
                if (in[f].at + len[f] == in[f].bytes) finalBits |= 2 << f;
                if (len[f] >= (1ull << 31)) {
                    fprintf(stderr, "HipDividedReadsSets: a FASTQ record or the lag between the two files exceeds 2 GiB\n");
                    exit(EXIT_FAILURE);
                }
            }
            pgrc_divided_reads part;                        // (the divider's arrays: valid until its next run)
            uint64_t used[2] = {0, 0}, cnt = 0;
            failOn(pgrc_divider_run_fastq(divider, in[0].p + in[0].at, len[0], paired ? (in[1].p ? in[1].p + in[1].at : "") : nullptr,
                                          paired ? len[1] : 0, revComplPairFile ? 1 : 0, finalBits, &used[0], paired ? &used[1] : nullptr, &cnt,
                                          &part), divider, "divider_run_fastq");
            // the reference stops at the first exhausted file (pair files of different lengths included): so does this,
            // without growing the other file's window to its end
            const bool final = pgrc_divider_last_was_terminal(divider) != 0;
            batchesServed++;
            appendRows(readsSets->getHqReadsSet(), part.hq_rows, part.n_hq);
            appendRows(readsSets->getLqReadsSet(), part.lq_rows, part.n_lq);
            if (separateNReadsSet) appendRows(readsSets->getNReadsSet(), part.n_rows, part.n_n);
            for (uint64_t k = 0; k < part.n_lq; k++) lqMapping.push_back(seen + part.lq_index[k]);
            for (uint64_t k = 0; k < part.n_n; k++) nMapping.push_back(seen + part.n_index[k]);
            seen += (uint_reads_cnt_max) cnt;
            if (final) break;
            for (int f = 0; f < (paired ? 2 : 1); f++) {
                in[f].at += used[f];
                // nothing taken from a window that did not reach the end of its file: it has to show more next time
                window[f] = (used[f] == 0 && in[f].at + len[f] < in[f].bytes) ? window[f] * 2 : piece;
            }
        }
        for (int f = 0; f < (paired ? 2 : 1); f++)
            if (in[f].p) munmap((void *) in[f].p, in[f].bytes);
        pgrc_divider_destroy(divider);
        cout << "Filtered " << (lqMapping.size() + nMapping.size());
        if (separateNReadsSet)
            cout << " (including " << nMapping.size() << " containing N)";
        cout << " reads (out of " << seen << ") on the device, from the FASTQ text." << endl;
        readsSets->lqMapping = new VectorMapping(std::move(lqMapping), seen);
        if (separateNReadsSet)
            readsSets->nMapping = new VectorMapping(std::move(nMapping), seen);
        return readsSets;
    }

    DividedPCLReadsSets *HipDividedReadsSets::getSimpleDividedPCLReadsSets(
            ReadsSourceIteratorTemplate<uint_read_len_max> *readsIt, uint_read_len_max readLength, bool separateNReadsSet,
            bool nReadsLQ) {
        // (the reference's own shortcut for "no N division" only skips the N test: error_limit = 1 does the same here)
        return getQualityDivisionBasedReadsSets(readsIt, readLength, 1, false, separateNReadsSet, nReadsLQ);
    }
}
