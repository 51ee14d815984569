/*
 * pgrc_oracle.h -- CPU restatement of PgRC's read-to-pseudogenome matching path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and there only as the checker /
 * the timed CPU baseline, never as the thing measured or shipped.  The HIP
 * library (pgrc_amd/csrc -> libpgrc_match.so) never links or calls it.
 *
 * Parity pin: every function here is checked against outputs of the real
 * reference compiled in the build container (oracle/_ref/libpgrc_ref.so, see
 * oracle/Makefile + oracle/ref_harness.cpp) and against the committed golden
 * fixtures generated from it (tests/golden/, tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  No reference source is copied: the code is
 * written from the behavioural specification in SURVEY.md Appendix A.
 */
#ifndef PGRC_ORACLE_H
#define PGRC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGRC_OR_NOT_MATCHED_POS UINT64_MAX /* ReadsMatchers.cpp:69 */
#define PGRC_OR_NOT_MATCHED_CNT 255        /* ReadsMatchers.h:17 */
#define PGRC_OR_BUCKET_CAP 13              /* CopMEMMatcher.h:11 (limit 12 => 13 kept) */
#define PGRC_OR_TRUNC_BUCKET 4             /* CopMEMMatcher.h:13 */

/* copMEM parameters derived from the seed length and the Pg length.
 * matching/copmem/CopMEMMatcher.cpp:69-96 (initParams), :111-137 (calcCoprimes). */
typedef struct {
    int32_t L;          /* seed ("target match") length */
    int32_t K;          /* hashed k-mer length */
    int32_t k1;         /* Pg sampling step */
    int32_t k2;         /* read probing step */
    uint32_t hash_size; /* power of two in [2^24, 2^31] */
} pgrc_or_copmem_params;

/* Returns 0 on success, non-zero when the reference would exit(EXIT_FAILURE)
 * (seed < 24: CopMEMMatcher.cpp:77-80; L/K mismatch: :115-118). */
int pgrc_or_copmem_derive(uint32_t seed_len, uint64_t pg_len, pgrc_or_copmem_params *out);

/* maRushPrime1HashSparsified<K> over ASCII bytes, raw u32 (not masked).
 * matching/copmem/Hashes.h:54-76. */
uint32_t pgrc_or_copmem_hash(int K, const char *str);

/* Canonical (serial-build) copMEM seed index: CopMEMMatcher.cpp:140-231 with
 * PgHelpers::numberOfThreads == 1.  cumm has hash_size+2 entries laid out as
 * the reference leaves them after the fill pass (bucket h = [cumm[h], cumm[h+1])),
 * positions has cumm[hash_size] entries.  Both are malloc'ed; free with
 * pgrc_or_index_free. */
typedef struct {
    pgrc_or_copmem_params p;
    uint64_t pg_len;
    uint32_t *cumm;
    uint32_t *positions;
    uint64_t count;
} pgrc_or_index;

int pgrc_or_index_build(const char *pg, uint64_t pg_len, uint32_t seed_len, pgrc_or_index *out);
void pgrc_or_index_free(pgrc_or_index *idx);

/* One read against the index: CopMEMMatcher.cpp:483-566
 * (processApproxMatchQueryTight).  *cnt is in/out (current mismatch count,
 * 255 = unmatched).  Returns the match position in the indexed text or
 * PGRC_OR_NOT_MATCHED_POS.  *falses receives this read's false-candidate
 * count (currentFalseMatchCount, incl. the double-counted tail rejects). */
/* test switch (default 0 = the reference's loops): stop a read once the HIP kernel's early-stop rule holds; and the
 * number of seed probes executed since the last reset */
void pgrc_or_set_early_stop(int on);
void pgrc_or_set_dual_spec(int small_limit_plus_1);   /* test switch: the dual scheme's speculative first attempt */
uint64_t pgrc_or_probe_count(int reset);
uint64_t pgrc_or_copmem_match_read(const pgrc_or_index *idx, const char *pg, const char *read,
                                   uint32_t read_len, uint8_t kmax, uint8_t kmin, uint8_t *cnt,
                                   uint64_t *falses, uint64_t *candidates);

/* Result block shared by all modes (ReadsMatchers.h:32-43,115-116). */
typedef struct {
    uint64_t *pos;      /* [n] UINT64_MAX = unmatched */
    uint8_t *rc;        /* [n] 0/1 */
    uint8_t *mism;      /* [n] 255 = unmatched */
    uint64_t hist[256]; /* matchedCountPerMismatches */
    uint64_t matched;   /* matchedReadsCount */
    /* work counters (not parity targets; feed the roofline's algorithmic bytes) */
    uint64_t searched[2];   /* reads not skipped, per pass */
    uint64_t candidates[2]; /* verified candidates, per pass */
    uint64_t falses[2];
} pgrc_or_result;

/* Mode 'c': two-pass driver, ReadsMatchers.cpp:162-172 + :421-451.
 * reads: n*read_len ASCII bytes, row-major.  pg is NOT modified (the oracle
 * works on a private reverse-complemented copy for pass 2).  threads is the
 * OpenMP width of the per-read loop (the index build is always the serial
 * canonical one).  Arrays in res must be caller-allocated with n entries.
 * If init != 0 the result arrays are initialised (initMatching :97-105,
 * :411-415); with init == 0 they are taken as the state handed over by a
 * previous phase (continueMatchingConstantLengthReads :174-184). */
int pgrc_or_match_copmem(const char *pg, uint64_t pg_len, const char *reads, uint64_t n,
                         uint32_t read_len, uint32_t seed_len, uint8_t kmax, uint8_t kmin,
                         int rev_compl_pg, int threads, int init, pgrc_or_result *res);
/* one query over both strands at once (the HIP path's dual kernel), restated: must equal
 * pgrc_or_match_copmem(..., rev_compl_pg = 1, ...) on every input; kmin must be 0; *aborted = reads redone in the
 * reference's order because the falses bound exceeded the budget */
int pgrc_or_match_copmem_dual(const char *pg, uint64_t pg_len, const char *reads, uint64_t n,
                              uint32_t read_len, uint32_t seed_len, uint8_t kmax, uint8_t kmin,
                              int threads, int init, pgrc_or_result *res, uint64_t *aborted);
/* the HIP path's schedule for a two-pass run (exact-match screen on the RC text first), restated: must equal
 * pgrc_or_match_copmem(..., rev_compl_pg = 1, ...) on every input */
int pgrc_or_match_copmem_screened(const char *pg, uint64_t pg_len, const char *reads, uint64_t n,
                                  uint32_t read_len, uint32_t seed_len, uint8_t kmax, uint8_t kmin,
                                  int threads, int init, pgrc_or_result *res);

/* Modes 'e' (exact, ReadsMatchers.cpp:198-230), 'd' (:297-341), 'i' (:364-409)
 * over the read-side seed index (ConstantLengthPatternsOnTextHashMatcher.{h,cpp})
 * with the canonical exact-seed semantics of SURVEY.md section 8a: candidates in
 * ascending text position, equal-seed patterns in descending pattern index. */
int pgrc_or_match_seedindex(char mode, const char *pg, uint64_t pg_len, const char *reads,
                            uint64_t n, uint32_t read_len, uint32_t seed_len, uint8_t kmax,
                            uint8_t kmin, int rev_compl_pg, pgrc_or_result *res);

/* mapReadsIntoPg parameter derivation, ReadsMatchers.cpp:699-713. */
typedef struct {
    uint8_t kmax;     /* maxMismatches = readLength / minCharsPerMismatch */
    uint8_t kmin;     /* shortcut (upper-case mode) ? kmax : 0 */
    uint32_t seed_len; /* clipped to read_len */
    uint8_t parts;    /* targetMismatches + 1 */
    char matcher;     /* 'c', 'd', 'i' or 'e' (exact matcher selected) */
} pgrc_or_map_params;
int pgrc_or_map_derive(uint32_t read_len, uint32_t seed_len, uint32_t min_chars_per_mismatch,
                       char mode, pgrc_or_map_params *out);

/* Mismatch extraction for one matched read: ReadsMatchers.cpp:40-66, :548-559,
 * utils/helper.cpp:358-362.  read is the ORIGINAL read (it is reverse
 * complemented here when rc != 0).  reversed selects
 * fillEntryWithReversedMismatches.  Writes cnt (code, offset) pairs. */
void pgrc_or_extract_mismatches(const char *pg, uint64_t pos, const char *read, uint32_t read_len,
                                int rc, int reversed, uint8_t cnt, uint8_t *codes,
                                uint16_t *offsets);

/* Pg-vs-Pg exact matching: CopMEMMatcher::matchTexts (matching/copmem/CopMEMMatcher.cpp:604-622 ->
 * processExactMatchQueryTight :333-481) with a matcher built by CopMEMMatcher(src, N, target_len) (:571-591).
 * dest is the text as handed to matchTexts (the caller reverse-complements it, SimplePgMatcher.cpp:31-41).
 * Matches come out in discovery order.  Returns non-zero where the reference would exit (min_match_len < K). */
typedef struct {
    uint64_t pos_src, length, pos_dest;   /* TextMatch, matching/TextMatchers.h:10-16 */
} pgrc_or_text_match;
int pgrc_or_mem_match(const char *src, uint64_t N, const char *dest, uint64_t N2, int dest_is_src, int rev_compl,
                      uint32_t target_len, uint32_t min_match_len, pgrc_or_text_match **out, uint64_t *count);
void pgrc_or_mem_free(pgrc_or_text_match *m);

/* ---- export of the matches (row f1): the streams SeparatedPseudoGenomeOutputBuilder collects
 *      (ReadsMatchers.cpp:563-675, SeparatedPseudoGenomePersistence.cpp:961-1019).  The caller provides the arrays:
 *      off / mis_rev_off 2 bytes per value are always enough. */
typedef struct {
    uint64_t n_entries, n_mismatches;
    uint32_t off_width;
    uint8_t *off;
    uint32_t *org_idx;
    uint8_t *rev_comp, *mis_cnt, *mis_sym, *mis_rev_off;
    uint64_t last_pos;
} pgrc_or_export_streams;
int pgrc_or_export_pg_order(const char *pg, const char *reads, uint32_t L, const uint64_t *pos, const uint8_t *rc,
                            const uint8_t *mism, const uint32_t *order, uint64_t m, const uint32_t *read_org,
                            const uint8_t *list_off, const uint32_t *list_org, const uint8_t *list_rc, uint64_t h,
                            int pair_file, int byte_per_read_length, pgrc_or_export_streams *s);
int pgrc_or_export_entries(const char *pg, const char *reads, uint32_t L, const uint64_t *pos, const uint8_t *rc,
                           const uint8_t *mism, const uint32_t *entry_read, const uint32_t *entry_org, uint64_t ne,
                           int pair_file, int byte_per_read_length, pgrc_or_export_streams *s);


/* helpers */
void pgrc_or_revcomp(char *seq, uint64_t n);                 /* helper.cpp:383-393 */
uint8_t pgrc_or_sym2val(char c);                             /* helper.cpp:277-283: A0 C1 G2 T3 N4 */
/* SymbolsPackingFacility layout (coders/SymbolsPackingFacility.cpp:133-178):
 * sigma symbols, spe symbols per byte, first symbol most significant. */
void pgrc_or_pack_read(const char *read, uint32_t read_len, const char *alphabet, uint8_t *dst);
void pgrc_or_unpack_read(const uint8_t *src, uint32_t read_len, const char *alphabet, char *dst);


/* Row f3: DividedPCLReadsSets::getQualityDivisionBasedReadsSets (readsset/DividedPCLReadsSets.cpp:59-100) over n records
 * given as symbol rows and quality rows (read_len bytes each; quals may be NULL when error_limit >= 1).  The packed rows
 * of the three sets (caller's buffers, n * ceil(read_len / 3) bytes are enough for each), the batch indexes of the LQ / N
 * reads (n entries each), counts[3] = reads per set (HQ, LQ, N), symbols[3] = alphabet size of each set (0: no such set). */
int pgrc_or_divide_reads(const char *reads, const char *quals, uint64_t n, uint32_t read_len, double error_limit,
                         int simplified_suffix_mode, int separate_n, int n_reads_lq, uint8_t *hq_rows, uint8_t *lq_rows,
                         uint8_t *n_rows, uint32_t *lq_index, uint32_t *n_index, uint64_t counts[3], uint32_t symbols[3]);
/* FASTQReadsSourceIterator + RevComplPairReadsSetIterator over whole files held in memory (readsset/iterator/ReadsSetIterator.cpp:
 * 189-224, :256-284): the records' symbol rows and quality rows (read_len bytes each; the caller's buffers), records taken
 * from the two texts in turn when pair_text != NULL.  Returns the number of records, -1 when a read is not read_len long,
 * -3 when the text ends inside a record. */
int64_t pgrc_or_fastq_records(const char *text, uint64_t bytes, const char *pair_text, uint64_t pair_bytes, int rev_compl_pair,
                              uint32_t read_len, char *reads, char *quals, uint64_t max_records);
/* qualityLut[c] (utils/helper.cpp:284-327) as this file restates it */
float pgrc_or_quality_lut(int c);

#ifdef __cplusplus
}
#endif
#endif
