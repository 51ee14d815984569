/*
 * pgrc_match.h -- C ABI of libpgrc_match.so: PgRC's read-to-pseudogenome matching
 * path as hand-written HIP kernels for MI355X (gfx950).
 *
 * This is the drop-in boundary for the reference's DefaultReadsMatcher seam
 * (matching/ReadsMatchers.h:24-83, :109-143).  The reference has no FFI; the
 * entry points below are what a `HipReadsApproxMatcher : AbstractReadsApproxMatcher`
 * adapter binds (INTEGRATION.md shows that adapter and the 6-line change to
 * mapReadsIntoPg, matching/ReadsMatchers.cpp:716-740).  Plain pointers and
 * sizes only; no torch / STL types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a PGRC_E_* code otherwise;
 *     pgrc_match_last_error(ctx) gives the message.  The reference's convention
 *     is fprintf(stderr)+exit(EXIT_FAILURE) (ReadsMatchers.cpp:738-739,
 *     CopMEMMatcher.cpp:77-80): the adapter maps non-zero to that.
 *   - the caller keeps ownership of every host buffer; the context owns all
 *     device memory (the reference matcher likewise borrows pgPtr / readsSet,
 *     ReadsMatchers.cpp:71-95).
 *   - single caller, blocking unless stated; work is issued on the context's
 *     HIP stream (pgrc_match_set_stream).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with PGRC_E_NO_DEVICE.
 *
 * Symbol values: A0 C1 G2 T3 (utils/helper.cpp:277-283); N (4) exists only in reads.
 */
#ifndef PGRC_MATCH_H
#define PGRC_MATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGRC_NOT_MATCHED_POS UINT64_MAX /* DefaultReadsMatcher::NOT_MATCHED_POSITION, ReadsMatchers.cpp:69 */
#define PGRC_NOT_MATCHED_CNT 255        /* NOT_MATCHED_COUNT, ReadsMatchers.h:17 */

enum {
    PGRC_OK = 0,
    PGRC_E_PARAM = 1,       /* bad argument / unsupported configuration */
    PGRC_E_SEED_SHORT = 2,  /* copMEM needs seed >= 24 (CopMEMMatcher.cpp:77-80) */
    PGRC_E_NO_DEVICE = 3,   /* no usable HIP device */
    PGRC_E_ALLOC = 4,
    PGRC_E_SYMBOL = 5,      /* symbol outside ACGT (Pg) / ACGNT (reads) */
    PGRC_E_STATE = 6,       /* call order (e.g. run before set_pg) */
    PGRC_E_MODE = 7,        /* "Unknown matching mode" (ReadsMatchers.cpp:737-739) */
    PGRC_E_DEVICE = 8       /* HIP runtime failure other than out-of-memory (launch failure, invalid value, ...) */
};

typedef struct pgrc_match_ctx pgrc_match_ctx;

/* Matcher configuration = the constructor arguments of the reference matchers
 * (ReadsMatchers.h:131-132, :159-160, :178-179, :196-197). */
typedef struct {
    uint32_t read_len;      /* readsSet->maxReadLength(), <= 255 */
    uint32_t seed_len;      /* readsExactMatchingChars (already clipped to read_len) */
    uint8_t max_mismatches; /* maxMismatches */
    uint8_t min_mismatches; /* minMismatches (0, or max_mismatches in "shortcut" modes) */
    char mode;              /* 'c' copMEM (default), 'd', 'i', 'e' exact matcher */
    int32_t device;         /* HIP device ordinal, -1 = current */
} pgrc_match_params;

/* mapReadsIntoPg's parameter derivation (ReadsMatchers.cpp:699-723): fills *out from
 * the CLI-level values (-s [mode]len, -M minCharsPerMismatch). */
int pgrc_match_derive_params(uint32_t read_len, uint32_t seed_len, uint32_t min_chars_per_mismatch,
                             char mode_char, pgrc_match_params *out);

int pgrc_match_create(const pgrc_match_params *params, pgrc_match_ctx **out);
/* One matcher over SEVERAL GPUs of this node, in one process (the reference is one process with one matcher object,
 * pgrc-encoder.cpp:342-374, and shards its per-read loop over threads, ReadsMatchers.cpp:426-428; here the same
 * loop shards over devices).  The returned context answers every entry point of this header: reads are split into
 * contiguous, even-aligned ranges (PE mates 2q, 2q+1 stay on one device, ReadsMatchers.cpp:553), every device packs
 * 1/n of the text handed to pgrc_match_set_pg_ascii and ONE all-gather (RCCL over xGMI; peer copies when a device is
 * listed twice, which only makes sense for rehearsals on a smaller box) replicates the packed text, every device
 * builds the index and matches its reads, results land in the caller's arrays at the shard offsets and the
 * histograms are summed; the export streams are made on the first device (the shards' results and reads are
 * gathered there).  params->device is ignored; n_devices in [1, 32].  pgrc_match_set_stream,
 * pgrc_match_set_reads_device and pgrc_match_get_results_device need a single-device context. */
int pgrc_match_create_multi(const pgrc_match_params *params, int32_t n_devices, const int32_t *devices,
                            pgrc_match_ctx **out);
/* visible HIP devices (what "all" means for an adapter that honours PGRC_DEVICES) */
int pgrc_match_device_count(int32_t *count);
/* shards of a context: 1 for pgrc_match_create; per shard its device and (after the reads were set) read range */
int32_t pgrc_match_shard_count(const pgrc_match_ctx *ctx);
int pgrc_match_shard_info(const pgrc_match_ctx *ctx, int32_t shard, int32_t *device, uint64_t *first_read,
                          uint64_t *n_reads);
void pgrc_match_destroy(pgrc_match_ctx *ctx);
const char *pgrc_match_last_error(const pgrc_match_ctx *ctx);
/* hipStream_t to issue all work on (0 = the null stream). */
int pgrc_match_set_stream(pgrc_match_ctx *ctx, void *hip_stream);

/* ---- pseudogenome (replaces `char* pgPtr, pgLength`, ReadsMatchers.h:26-27) ---- */
/* ASCII ACGT text on the host; packed to 2 bits/symbol on the device. */
int pgrc_match_set_pg_ascii(pgrc_match_ctx *ctx, const char *pg, uint64_t pg_len);
/* 2-bit packed text already in HBM (16 symbols per little-endian u32, symbol i at
 * bits 2*(i%16)); e.g. the result of the RCCL all-gather.  Copied into the context. */
int pgrc_match_set_pg_packed_device(pgrc_match_ctx *ctx, const void *d_words, uint64_t pg_len);
/* Packs host ASCII pg[first .. first+count) into d_words_out (device), count/16 rounded
 * up words; `first` must be a multiple of 16.  Used per rank before the all-gather. */
int pgrc_match_pack_pg_slice(pgrc_match_ctx *ctx, const char *pg_slice, uint64_t count,
                             void *d_words_out);

/* ---- reads (replaces ConstantLengthReadsSetInterface*, readsset/ReadsSetInterface.h:28-43) ---- */
/* n rows of read_len ASCII symbols (what readsSet->getRead(i, buf) yields,
 * ReadsMatchers.cpp:432).  Rows containing 'N' take the byte-compare kernel. */
int pgrc_match_set_reads_ascii(pgrc_match_ctx *ctx, const char *reads, uint64_t n);
/* The same, streamed: begin(n), then append consecutive blocks of rows, then end. */
int pgrc_match_begin_reads(pgrc_match_ctx *ctx, uint64_t n);
int pgrc_match_append_reads_ascii(pgrc_match_ctx *ctx, const char *reads, uint64_t count);
int pgrc_match_end_reads(pgrc_match_ctx *ctx);
/* The reference's own packed read sets, taken as they are (PackedConstantLengthReadsSet::getPackedRead,
 * readsset/PackedConstantLengthReadsSet.h:40; layout: coders/SymbolsPackingFacility.cpp:133-178, base-sigma digits,
 * first symbol most significant, a partial last byte padded with digit 0):
 *   symbols = 4: an "ACGT" set, ceil(read_len/4) bytes per read, 4 symbols per byte;
 *   symbols = 5: an "ACGNT" set (A0 C1 G2 N3 T4), ceil(read_len/3) bytes per read, 3 symbols per byte
 * (the two set kinds of readsset/DividedPCLReadsSets.cpp:10-21).  Streamed like the ASCII rows; the LQ + N sum set
 * of pgrc-encoder.cpp:349-352 is begin(n_lq + n_n), append_packed(lq rows, n_lq, 4), append_packed(n rows, n_n, 5),
 * end.  Unpacking to the library's 2-bit words (+ the side list of reads with N) happens on the device. */
int pgrc_match_append_reads_packed(pgrc_match_ctx *ctx, const uint8_t *packed, uint64_t count, int32_t symbols);
/* one ACGT set in one call (= begin, append_packed(..., 4), end) */
int pgrc_match_set_reads_packed(pgrc_match_ctx *ctx, const uint8_t *packed, uint64_t n);
/* Reads already in HBM in the library's layout: word-major u32 [words_per_read][stride]
 * (word w of read i at d_words[w*stride + i]), 16 symbols per word.  Borrowed, not copied. */
int pgrc_match_set_reads_device(pgrc_match_ctx *ctx, const void *d_words, uint64_t n, uint64_t stride);
uint32_t pgrc_match_words_per_read(uint32_t read_len);

/* ---- matching (DefaultReadsMatcher::matchConstantLengthReads, ReadsMatchers.cpp:162-172) ---- */
/* initMatching (:97-105, :411-415): pos = NOT_MATCHED, rc = 0, count = 255. */
int pgrc_match_init_results(pgrc_match_ctx *ctx);
/* hand-over from a previous phase (transferMatchingResults, :111-133). */
int pgrc_match_set_results(pgrc_match_ctx *ctx, const uint64_t *pos, const uint8_t *rc,
                           const uint8_t *mism);
/* forward pass, then (rev_compl_pg != 0) the pass over the reverse-complemented Pg. */
int pgrc_match_run(pgrc_match_ctx *ctx, int rev_compl_pg);
/* a single executeMatching(revCompMode) (ReadsMatchers.h:46): strand 0 = the text as given, 1 = its
 * reverse complement (built on the device; the host text is never modified). */
int pgrc_match_run_pass(pgrc_match_ctx *ctx, int strand);
/* readMatchPos / readMatchRC / readMismatchesCount / matchedCountPerMismatches /
 * matchedReadsCount (ReadsMatchers.h:32-35, :115-116).  Any pointer may be NULL. */
int pgrc_match_get_results(pgrc_match_ctx *ctx, uint64_t *pos, uint8_t *rc, uint8_t *mism,
                           uint64_t hist[256], uint64_t *matched);
/* device-resident result arrays (u64[n], u8[n], u8[n]); valid until the next set_reads/destroy. */
int pgrc_match_get_results_device(pgrc_match_ctx *ctx, void **d_pos, void **d_rc, void **d_mism);

/* ---- mismatch extraction (AbstractReadsApproxMatcher::updateEntry, ReadsMatchers.cpp:548-559,
 *      fillEntryWith(Reversed)Mismatches :40-66) ---- */
/* For every read: cum[i+1]-cum[i] = its mismatch count (0 for unmatched reads); codes[] =
 * (val(pg)<<4)+val(read), offsets[] as the reference emits them.  reversed_flags[i] != 0
 * selects the "reversed" form; NULL means reversed = rc (the SE rule; for
 * revComplPairFile the caller passes rc != (orgIdx & 1), :553).  cum has n+1 entries;
 * codes/offsets need cum[n] entries: call once with codes == NULL to get cum only. */
int pgrc_match_extract_mismatches(pgrc_match_ctx *ctx, const uint8_t *reversed_flags, uint64_t *cum,
                                  uint8_t *codes, uint16_t *offsets);

/* Large device buffers of destroyed contexts are kept for the next context of the process (re-allocating tens of GB right
 * after freeing them can stall for seconds; at most PGRC_DEVICE_POOL_GB, default 96, 0 = keep nothing).  This returns them
 * to the driver; the result is the number of bytes freed. */
uint64_t pgrc_match_trim_device_memory(void);

/* ---- pipelined hand-over of a whole job (mode c, both strands, first phase; single-device contexts): the steps of
 *      set_pg / set_reads / run / get_results overlap instead of following each other (pgrc_amd/csrc/stream.hip).
 *   pgrc_match_set_pg_ascii(ctx, ...);
 *   pgrc_match_prepare_index(ctx, 1);                 both strands' index builds start now, beneath the upload of the reads
 *   pgrc_match_begin_reads(ctx, n);
 *   pgrc_match_stream_begin(ctx, pos, rc, mism);      the caller's result arrays (n entries each)
 *   pgrc_match_append_reads_*(ctx, rows, count) ...   every block is matched as soon as it is on the device, while the
 *                                                     caller copies the next; a worker thread brings its results back
 *   pgrc_match_end_reads(ctx);
 *   pgrc_match_stream_end(ctx, hist, &matched);       reads with N, histogram, last downloads: the arrays are complete
 * The results are those of pgrc_match_init_results + pgrc_match_run(ctx, 1) (DefaultReadsMatcher::
 * matchConstantLengthReads, ReadsMatchers.cpp:162-172).  Afterwards the context is in the state a plain run leaves
 * (exports, extract_mismatches, get_results work).  PGRC_E_STATE where streaming does not apply (other modes,
 * min_mismatches > 0, multi-device contexts): take the plain calls. ---- */
int pgrc_match_prepare_index(pgrc_match_ctx *ctx, int32_t both_strands);
int pgrc_match_stream_begin(pgrc_match_ctx *ctx, uint64_t *pos, uint8_t *rc, uint8_t *mism);
int pgrc_match_stream_end(pgrc_match_ctx *ctx, uint64_t hist[256], uint64_t *matched);

/* ---- export of the matches as reads-list streams (DefaultReadsMatcher::exportMatchesInPgOrder /
 *      exportMatchesInOriginalOrder, ReadsMatchers.cpp:563-675; SeparatedPseudoGenomeOutputBuilder::writeReadEntry /
 *      writeReadsFromIterator, pseudogenome/persistence/SeparatedPseudoGenomePersistence.cpp:961-1019) ---- */
/* The byte streams the builder would have appended, ready to be written into its destinations in one piece each.
 * Read-length values (offsets) take off_width bytes: 1 with PgHelpers::bytePerReadLengthMode, else 2
 * (writeReadLengthValue, utils/helper.cpp:198-203).  Arrays are malloc'ed by the library: pgrc_match_free_export. */
typedef struct {
    uint64_t n_entries;     /* entries of the output reads list (old list entries + matched reads, or the given entries) */
    uint64_t n_mismatches;  /* mismatches over all entries */
    uint32_t off_width;
    uint8_t *off;           /* rlOff:       n_entries offsets to the previously written entry (:966) */
    uint32_t *org_idx;      /* rlOrgIdx:    n_entries original read indexes (:967) */
    uint8_t *rev_comp;      /* rlRevComp:   n_entries flags (:969) */
    uint8_t *mis_cnt;       /* rlMisCnt:    n_entries counts (:971) */
    uint8_t *mis_sym;       /* rlMisSym:    n_mismatches codes (actual << 4) + mismatch (:973-974) */
    uint8_t *mis_rev_off;   /* rlMisRevOff: n_mismatches offsets coded backwards from the read end (:975-981) */
    uint64_t last_pos;      /* position of the last entry (the builder's lastWrittenPos) */
} pgrc_export_streams;

/* exportMatchesInPgOrder: the matched reads, in the order `order` gives them (ascending match position; the order among
 * reads matched at ONE position is whatever the caller's sort made it -- the reference's std::sort /
 * __gnu_parallel::sort order is an artefact of that algorithm, so the adapter reproduces it on (position, index)
 * pairs and passes the permutation in), merged with the reads list already on the pseudogenome (offset deltas,
 * original indexes, RC flags of SeparatedPseudoGenome::getReadsList(); it must carry no mismatches): in front of a new
 * entry go all old entries at SMALLER positions, an old entry at the same position follows it (:1004-1019). */
typedef struct {
    const uint32_t *order;        /* n_matched read indexes.  NULL is only accepted with n_matched == 0 (no matched read) or with
                                   * order_on_device set: anything else is PGRC_E_PARAM (a forgotten order must not silently
                                   * become another tie order) */
    uint64_t n_matched;
    const uint32_t *read_org_idx; /* original index of every READ (IndexesMapping::getReadOriginalIndex), NULL = identity */
    const uint8_t *list_off;      /* the old list: list_count offset deltas */
    const uint32_t *list_org_idx;
    const uint8_t *list_rev_comp; /* NULL = all forward */
    uint64_t list_count;
    int32_t rev_compl_pair_file;  /* mismatch lists in the original read's orientation iff rc != (orgIdx odd) (:553) */
    int32_t byte_per_read_length; /* PgHelpers::bytePerReadLengthMode */
    int32_t order_on_device;      /* != 0: the library makes the order on the device -- ascending match position, reads matched at
                                   * one position in ascending read index (a stable radix sort of (position, read) records; text
                                   * below 2^32 symbols) -- and ignores order / n_matched */
} pgrc_export_pg_order_args;
int pgrc_match_export_pg_order(pgrc_match_ctx *ctx, const pgrc_export_pg_order_args *args, pgrc_export_streams *out);
/* exportMatchesInOriginalOrder for a caller-made entry list (pgrc_match_export_original_order makes the list itself):
 * entry_read[k] = the read entry k describes, or UINT32_MAX for a filler entry (position 0, no mismatches);
 * entry_org_idx[k] = its original index.  Offsets are the entries' own positions (every entry starts from a fresh
 * DefaultReadsListEntry(0), :655-667). */
int pgrc_match_export_entries(pgrc_match_ctx *ctx, const uint32_t *entry_read, const uint32_t *entry_org_idx,
                              uint64_t n_entries, int32_t rev_compl_pair_file, int32_t byte_per_read_length,
                              pgrc_export_streams *out);
/* exportMatchesInOriginalOrder with the entry list made by the library (ReadsMatchers.cpp:597-675): one entry per
 * original read index in [0, reads_total_count) -- the matched read that carries it, or a filler (position 0, no
 * mismatches) for an index none of the matcher's reads carries; an index carried by an UNMATCHED read gets no entry.
 * With pair_file_mode all even indexes come first, then all odd ones (:625-667).  read_org_idx[i] =
 * IndexesMapping::getReadOriginalIndex(i) for every read of the matcher; the indexes must be distinct and below
 * reads_total_count (PGRC_E_PARAM otherwise). */
typedef struct {
    const uint32_t *read_org_idx;
    uint64_t reads_total_count;   /* IndexesMapping::getReadsTotalCount() */
    int32_t pair_file_mode;
    int32_t rev_compl_pair_file;
    int32_t byte_per_read_length;
} pgrc_export_original_order_args;
int pgrc_match_export_original_order(pgrc_match_ctx *ctx, const pgrc_export_original_order_args *args, pgrc_export_streams *out);
void pgrc_match_free_export(pgrc_export_streams *streams);

/* ---- introspection (tests, bench) ---- */
typedef struct {
    int32_t K, k1, k2;
    uint32_t hash_size;
} pgrc_copmem_params;
int pgrc_match_copmem_params(uint32_t seed_len, uint64_t pg_len, pgrc_copmem_params *out);
/* canonical copMEM index of the forward (strand 0) or reverse-complemented (1) Pg in the
 * reference's layout: cumm[hash_size+2], positions[count].  NULL pointers = query count only. */
int pgrc_match_export_index(pgrc_match_ctx *ctx, int strand, uint32_t *cumm, uint32_t *positions,
                            uint64_t *count);
/* 2-bit packed Pg of either strand, copied to the host (ceil(pg_len/16) words). */
int pgrc_match_export_pg(pgrc_match_ctx *ctx, int strand, uint32_t *words);

typedef struct {
    uint64_t searched[2];   /* reads not skipped (count > min_mismatches), per pass */
    uint64_t candidates[2]; /* verified candidates, per pass */
    uint64_t probes[2];     /* seed lookups, per pass */
    uint64_t entry_fetches[2]; /* index entries fetched beyond the inline ones (mode c) */
    uint64_t verifies[2];   /* text windows fetched and compared (mode c) */
    uint64_t index_entries[2];
    /* device time in ms of the last run, by kernel class (HIP events on the ctx stream) */
    float ms_index[2];
    float ms_match[2];
    float ms_other;
    float ms_total;
    float ms_allgather;     /* multi-device contexts: the last all-gather of the packed text (host clock) */
    /* two-pass runs of mode c with min_mismatches == 0: 1 = the run took the screened schedule (both indexes first, an
     * exact-match screen on the RC text, then the two passes); ms_screen = the screen launch, its probes / candidates /
     * fetches are counted with strand 1 */
    uint32_t screened;      /* 2 = one query per read over both strands (the dual kernel): ms_screen = that launch, its
                             * work is in dual[], ms_match[] = the two ordinary passes over what it left undecided
                             * (redo_reads) and the reads with N */
    float ms_screen;
    uint64_t redo_reads;
    uint64_t dual[5];       /* screened == 2: the dual kernel's own searched / candidates / heads probed / entry fetches /
                             * verifies (the per-strand counters above then describe the two ordinary passes after it) */
    uint32_t schedule_downgraded; /* 1 = the second index set (or the screen's arrays) did not fit in device memory: this
                             * context runs the two passes in turn until another text / read set is handed over */
    uint64_t dual_seed_probes; /* screened == 2: seeds the dual kernel probed -- a seed's forward and RC head (dual[2] counts
                             * both) lie in ONE 128-byte line of the pair table: this is the number of line requests for heads */
} pgrc_match_counters;
/* flags[i] != 0: read i was one of `redo_reads` -- the dual kernel met a bucket it could not judge without the
 * reference's own falses count and did the read again in the reference's order (tests and bench.py draw their parity
 * samples from these reads).  n bytes; all zero when the last run did not take the dual kernel. */
int pgrc_match_get_redo_flags(pgrc_match_ctx *ctx, uint8_t *flags);
/* ---- run-time options.  A context reads the environment ONCE, in pgrc_match_create / pgrc_match_create_multi /
 *      pgrc_mem_create -- never at a launch.  None of the variables changes a result; unset = the library's choice.
 *   schedule of a two-strand run of mode c:
 *     PGRC_DUAL=0|1          never / whenever it applies: one query per read over both strands (the dual kernel)
 *     PGRC_SCREEN=0|1        never / whenever it applies: the screened schedule (exact-match screen, forward pass, RC pass)
 *     PGRC_EARLY_STOP=0      every read probes all its seeds, as the reference does (and the two passes in turn)
 *     PGRC_BUILD_STREAMS=1   the two index builds of such a run on one stream instead of two
 *     PGRC_HEAD_PAIR=0|1..4  no pair table (a head table per strand) / groups of 1, 2, 4, 8 buckets (default 3 = groups of 4)
 *     PGRC_NREAD_INLINE=0    every read with an N takes the byte-path kernel (default: the dual kernel takes those with <= 4 N)
 *     PGRC_MATCH_STAGE=0     the per-strand match kernel without staged refills
 *   index build:  PGRC_INDEX_SORT=sweep|own (front end), PGRC_INDEX_FINISH=general (the general finish kernel for every
 *                 partition), PGRC_INDEX_CFG=0 (the passes without the XCD-aware tile order: A/B runs)
 *   modes d/i/e:  PGRC_SEED_FILTER=0|1, PGRC_SEED_HEAVY=n, PGRC_SEED_READ_BATCH=n, PGRC_SEED_SEGMENT=n,
 *                 PGRC_SEED_SORT=full|segments (the table's pairs sorted by global passes / by segments in LDS; default by size),
 *                 PGRC_SEED_HEAVY_FORM=window|grouped (windows on keys with many entries: a wave per window / grouped by key, default)
 *   hand-over:    PGRC_UPLOAD_CHUNK_MB=n (staging chunk of append_reads_*), PGRC_STREAM_TIMING (milestones on stderr),
 *                 PGRC_HOST_PACK=0 (an ASCII text goes up as bytes and a kernel packs it; default: host threads pack it
 *                 into pinned buffers), PGRC_HOST_THREADS=n (those threads, default up to 8)
 *   tests:        PGRC_FORCE_POS64=1, PGRC_TEST_NO_SECOND_INDEX, PGRC_TEST_SEGMENT_TOP_BITS=n, PGRC_MEM_EVENT_CAP=n, PGRC_ALLGATHER=rccl|copy
 *   process-wide, read once per process: PGRC_DEVICE_POOL_GB (above), PGRC_DEBUG_ALLOC (log every device allocation)
 * pgrc_match_reload_options reads the environment again for a LIVE context (tests and A/B tools that change a variable
 * between two runs of one context). */
int pgrc_match_reload_options(pgrc_match_ctx *ctx);
/* enable per-kernel HIP-event timing + work counters for subsequent runs */
int pgrc_match_set_profiling(pgrc_match_ctx *ctx, int enabled);
int pgrc_match_get_counters(pgrc_match_ctx *ctx, pgrc_match_counters *out);
/* the same for a caller that may have been built against another version of this header: pgrc_match_counters only ever grows at
 * its END, and this writes exactly out_size bytes -- the first out_size bytes of the current struct, zeros beyond what the
 * library knows -- so `pgrc_match_get_counters_sized(ctx, &c, sizeof c)` never writes past the caller's struct (pgrc_match_get_counters
 * writes the library's sizeof) */
int pgrc_match_get_counters_sized(pgrc_match_ctx *ctx, void *out, size_t out_size);

/* ---- synthetic inputs (include/pgrc_synth.h) ---- */
#include "pgrc_synth.h"
/* host loops */
void pgrc_synth_pg_host(const pgrc_synth_pg *g, char *ascii_out);
void pgrc_synth_reads_host(const pgrc_synth_pg *g, const char *pg_ascii, const pgrc_synth_reads *rs,
                           uint64_t first_read, uint64_t count, char *ascii_out);
/* HIP generators: write straight into HBM in the library's packed layouts */
int pgrc_synth_pg_device(const pgrc_synth_pg *g, void *d_words_out, void *hip_stream);
int pgrc_synth_reads_device(const pgrc_synth_pg *g, const void *d_pg_words, const pgrc_synth_reads *rs,
                            uint64_t first_read, uint64_t count, void *d_words_out, uint64_t stride,
                            void *hip_stream);

const char *pgrc_match_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PGRC_MATCH_H */
