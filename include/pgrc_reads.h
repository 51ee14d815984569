/*
 * pgrc_reads.h -- C ABI of libpgrc_match.so, part 3: making the packed read sets on
 * MI355X -- the step BEFORE the matching path.
 *
 * Drop-in boundary: DividedPCLReadsSets::getQualityDivisionBasedReadsSets (readsset/DividedPCLReadsSets.cpp:59-100;
 * getSimpleDividedPCLReadsSets, :102-114, is the same with error_limit = 1), which the encoder runs over the FASTQ
 * iterator (pgrc/pgrc-encoder.cpp:254-262).  Per read, in input order (:68-87):
 *     the read holds an 'N' and N reads are set apart (separateNReadsSet || nReadsLQ)   -> N set (or the LQ set)
 *     else error_limit < 1 and !QualityDividingReadsSetIterator::isQualityHigh()         -> LQ set
 *     else                                                                               -> HQ set
 * and every read is appended to its set with PackedConstantLengthReadsSet::addRead (readsset/
 * PackedConstantLengthReadsSet.cpp:36-45) = SymbolsPackingFacility::packSequence (coders/SymbolsPackingFacility.cpp:
 * 147-162): big-endian base-|alphabet| digits, 4 symbols per byte over "ACGT", 3 per byte over "ACGNT", a last partial
 * byte padded with zero digits.  Which set uses which alphabet follows the constructor (DividedPCLReadsSets.cpp:10-21).
 *
 * The caller hands over a batch of FASTQ records as two row arrays (symbols and quality characters, read_len bytes per
 * row -- what the reference's iterator yields record by record) and gets back the packed rows of the three sets and the
 * batch-local indexes of the LQ and N reads (lqMapping / nMapping minus the index of the batch's first read).
 * integration/HipDividedReadsSets.{h,cpp} is the reference-side caller; INTEGRATION.md shows the one-line change.
 *
 * Same conventions as pgrc_match.h: 0 = success, PGRC_E_* otherwise; host buffers stay the caller's; no CPU fallback.
 */
#ifndef PGRC_READS_H
#define PGRC_READS_H

#include <stddef.h>
#include <stdint.h>

#include "pgrc_match.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgrc_divider pgrc_divider;

typedef struct {
    uint32_t read_len;               /* constant read length, 1..255 */
    double error_limit;              /* QualityDividingReadsSetIterator's error_level (pgrc-encoder.cpp:259: promils / 1000.0);
                                        >= 1: no quality division, the quality rows are not looked at */
    int32_t simplified_suffix_mode;  /* isQualityHigh() = quality[(int) (read_len * (1 - error_limit))] > '#'
                                        (DivisionReadsSetDecorators.cpp:14, :31-33); else the arithmetic mean of the
                                        correct-base probabilities of all positions (utils/helper.cpp:452-475) */
    int32_t separate_n_reads_set;    /* DividedPCLReadsSets' constructor arguments */
    int32_t n_reads_lq;
    int32_t device;                  /* HIP device, -1 = the current one */
} pgrc_divide_params;

typedef struct {
    uint64_t n_hq, n_lq, n_n;                    /* reads of the batch that went to each set */
    uint32_t hq_symbols, lq_symbols, n_symbols;  /* alphabet of the set: 4 = "ACGT", 5 = "ACGNT", 0 = the set does not exist */
    uint32_t hq_row_bytes, lq_row_bytes, n_row_bytes;   /* PackedConstantLengthReadsSet::packedLength */
    const uint8_t *hq_rows, *lq_rows, *n_rows;   /* packedReads of the set's part of this batch, reads in input order */
    const uint32_t *lq_index, *n_index;          /* index in the batch of every LQ / N read, ascending */
} pgrc_divided_reads;

int pgrc_divider_create(const pgrc_divide_params *params, pgrc_divider **out);
void pgrc_divider_destroy(pgrc_divider *d);
const char *pgrc_divider_last_error(const pgrc_divider *d);   /* NULL: why the last pgrc_divider_create failed */

/* One batch: `reads` and `quals` are n rows of read_len bytes (quals may be NULL when error_limit >= 1).  A symbol
 * outside ACGNT is PGRC_E_SYMBOL (the reference's validateSymbol exits there).
 * The arrays in *out are the divider's (pinned host memory, reused): they stay valid until its next run or its destruction
 * -- the caller appends them to its sets right away (PackedConstantLengthReadsSet::copyPackedRead). */
int pgrc_divider_run(pgrc_divider *d, const char *reads, const char *quals, uint64_t n, pgrc_divided_reads *out);

/* The same straight from FASTQ text (FASTQReadsSourceIterator, readsset/iterator/ReadsSetIterator.cpp:189-224: four lines per
 * record read with std::getline -- identifier, symbols, '+' line, qualities; the read is the leading run of letters of the
 * symbol line and must be read_len long, the quality string is cut or zero-padded to that length).  `text` is a piece of
 * the file, in file order; the call takes every COMPLETE record of it, divides them as pgrc_divider_run does and reports in
 * *consumed where the next piece has to start (the caller keeps the rest and appends what it reads next).  final_piece:
 * bit 0 (1) = nothing follows in either text; bit 1 (2) / bit 2 (4) = the first / the second text ends here while the other
 * may go on (round 4: pair files of different lengths -- the reference stops at the first exhausted file, and so does this:
 * pgrc_divider_last_was_terminal tells the caller when).  Where a text ends,
 * a last line without a newline counts; text that ends inside a record is PGRC_E_PARAM (what the
 * reference makes of a cut-off last record depends on strings left over from the record before: not reproduced).
 * pair_text != NULL: the second file of a pair (ManagedReadsSetIterator, readsset/persistance/ReadsSetPersistence.cpp:20-56):
 * records are taken from the two texts in turn, the first file first, as many whole pairs as both pieces hold (at the final
 * piece: until one file has no more, like the reference); rev_compl_pair != 0 reverse-complements the second file's reads
 * (RevComplPairReadsSetIterator, ReadsSetIterator.cpp:256-274; qualities are not reversed).  *n_records = records taken,
 * the batch indexes in *out count them in that order.  Pieces must be shorter than 2 GiB.  A record whose read is not
 * read_len letters long is PGRC_E_PARAM (the reference exits with "Unsupported variable length reads"). */
int pgrc_divider_run_fastq(pgrc_divider *d, const char *text, uint64_t bytes, const char *pair_text, uint64_t pair_bytes,
                           int32_t rev_compl_pair, int32_t final_piece, uint64_t *consumed, uint64_t *pair_consumed,
                           uint64_t *n_records, pgrc_divided_reads *out);

/* 1 if the last pgrc_divider_run_fastq took the last records the reference's iteration would take (its source is through
 * when its turn comes: one text ended and its records are used up) -- the caller stops; 0: hand over the next pieces */
int pgrc_divider_last_was_terminal(const pgrc_divider *d);

/* timing of the last run in milliseconds (upload [+ parsing], kernels, download) */
int pgrc_divider_last_ms(const pgrc_divider *d, float ms[3]);

#ifdef __cplusplus
}
#endif
#endif /* PGRC_READS_H */
