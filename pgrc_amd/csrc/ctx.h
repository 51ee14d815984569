// ctx.h -- shared declarations of libpgrc_match.so (host side of the HIP path).
//
// Data layout in HBM:
//   pg2[strand]   u32[ceil(G/16)+PG_PAD]   2-bit text, symbol i at bits 2*(i%16) of word i/16
//   reads2        u32[NW][stride]          word-major ("transposed") reads: thread i of a wave
//                                          reads word w of read i at reads2[w*stride+i] => every
//                                          wave-level load is one coalesced 256-B line
//   head          2 x u64[hash_size]       the bucket's two smallest entries (pos << 22 | 22-bit fingerprint); for
//                                          buckets with >= 3 entries w0 carries a flag, w1 the count and the index
//                                          of entry 1 in ent[]: one 16-B gather per probe (copmem.hip)
//   ent           u64[sampled positions]   all entries sorted by (bucket, position): the build's radix-sort output
//   pos/rc/mism   u64[n] / u8[n] / u8[n]   per-read results (ReadsMatchers.h:32-35,115)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgrc_match.h"

#define PGRC_PG_PAD_WORDS 32u   // zero words after the text so NW+1-word windows never fault
#define PGRC_MAX_NW 16          // ceil(255/16)
#define PGRC_BUCKET_CAP 13u     // HASH_COLLISIONS_PER_POSITION_LIMIT + 1 (CopMEMMatcher.h:11)
#define PGRC_TRUNC_BUCKET 4u    // UNLIMITED_NUMBER_OF_HASH_COLLISIONS_PER_POSITION (CopMEMMatcher.h:13)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct pgrc_multi;   // multi.hip

// Run-time options of a context.  The environment is read ONCE, when the context is created (pgrc_options_from_env,
// api.hip) -- never at a launch -- so a production matcher does not change schedule because somebody's environment
// changes under it; pgrc_match_reload_options(ctx) reads it again (tests and A/B tools that toggle a knob of a live
// context).  include/pgrc_match.h documents every variable.  None of them changes a result.
struct PgrcOptions {
    int dual = -1;                  // PGRC_DUAL        -1 the library's choice, 0 never, 1 whenever the dual kernel applies
    int screen = -1;                // PGRC_SCREEN      -1 not asked for, 0 never, 1 whenever the screened schedule applies
    bool early_stop = true;         // PGRC_EARLY_STOP=0: every read probes all its seeds, as the reference does
    bool builds_in_turn = false;    // PGRC_BUILD_STREAMS=1: the two index builds of a two-strand run on one stream
    uint32_t head_pair = 4;         // PGRC_HEAD_PAIR   0: a head table per strand; 1..4: pair table with groups of 1, 2, 4, 8 buckets
    int index_front = 0;            // PGRC_INDEX_SORT  0 "sweep" (idxsweep.hip), 1 "own" (idxsort.hip's stable scatter passes)
    bool index_finish_general = false;   // PGRC_INDEX_FINISH=general: the general finish kernel for every partition
    int index_cfg = -1;             // PGRC_INDEX_CFG=0  the passes of the index build without the XCD-aware tile order (A/B runs)
    bool match_stage = true;        // PGRC_MATCH_STAGE=0: refilling lanes load their own read (k_copmem_match_sm)
    bool nread_inline = true;       // PGRC_NREAD_INLINE=0: every read with an N takes the byte path
    bool force_pos64 = false;       // PGRC_FORCE_POS64=1: the 64-bit-position kernels on a small text (tests)
    bool test_no_second_index = false;   // PGRC_TEST_NO_SECOND_INDEX: the second index set "does not fit" (tests)
    bool stream_timing = false;     // PGRC_STREAM_TIMING: milestones of a streamed run on stderr
    bool host_pack = true;          // PGRC_HOST_PACK=0: an ASCII text goes up as bytes and a kernel packs it (rounds 1-4); default: host threads pack it into pinned buffers
    uint32_t host_threads = 0;      // PGRC_HOST_THREADS: host threads that pack the text (0 = up to 8)
    uint64_t upload_chunk_mb = 0;   // PGRC_UPLOAD_CHUNK_MB: staging chunk of append_reads_* (0 = 256, or 1024 for a streamed run)
    int seed_filter = -1;           // PGRC_SEED_FILTER  modes d/i/e: -1 where it pays, 0 never, 1 always
    uint32_t test_segment_top_bits = 0;   // PGRC_TEST_SEGMENT_TOP_BITS: the segment sort's short way over this many top bits (tests: 8 makes it fail and the long way run)
    int seed_heavy_form = 1;        // PGRC_SEED_HEAVY_FORM=window|grouped: heavy windows a wave each / grouped by their key (default)
    int seed_sort = -1;             // PGRC_SEED_SORT=full|segments: how modes d/i/e sort their (key, entry) pairs (-1: by the batch's size)
    uint32_t seed_heavy = 0;        // PGRC_SEED_HEAVY   modes d/i/e: entries of a window above which the persistent grid expands it (0 = default)
    uint64_t seed_read_batch = 0;   // PGRC_SEED_READ_BATCH / PGRC_SEED_SEGMENT: reads per batch / window starts per launch (tests; 0 = default)
    uint64_t seed_segment = 0;
    uint64_t mem_event_cap = 0;     // PGRC_MEM_EVENT_CAP: first guess of the Pg-vs-Pg matcher's event buffer (tests; 0 = default)
    int allgather = 0;              // PGRC_ALLGATHER    multi-device contexts: 0 by device list, 1 "rccl", 2 "copy" (tests)
    int dual_variant = -1;          // PGRC_DUAL_VARIANT which build of the dual kernel runs (A/B builds only; -1 = default)
};
PgrcOptions pgrc_options_from_env();
bool pgrc_pack_ascii_host(const uint8_t *src, uint64_t count, uint32_t *dst, uint32_t threads);   // api.hip: ASCII ACGT -> 2-bit words, false = a symbol outside ACGT

struct pgrc_match_ctx {
    PgrcOptions opt;                    // read from the environment by pgrc_match_create (see above)

    // several devices behind this object (pgrc_match_create_multi): it is then only the front, the work happens in
    // one child context per device
    pgrc_multi *multi = nullptr;
    void *export_view = nullptr;        // multi-device front: the shards' results gathered on the first device (export.hip)
    pgrc_match_ctx *export_front = nullptr;   // set in that gathered view: the front whose shards hold the reads (mismatch lists per shard)

    pgrc_match_params prm{};
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;  // the kernel for reads with N runs beside the main match kernel (created on first use)
    hipEvent_t side_ev[2]{};
    std::string err;

    // pseudogenome
    uint64_t G = 0;
    uint64_t pg_words = 0;
    DevBuf pg2[2];          // [0] forward, [1] reverse complement
    bool have_pg = false, have_rc = false;

    // reads
    uint64_t n = 0, stride = 0;
    uint32_t nw = 0;
    DevBuf reads_own;       // owned copy (set_reads_ascii / _packed)
    const uint32_t *reads2 = nullptr; // active pointer (owned or borrowed)
    bool have_reads = false;
    // reads with 'N': byte path
    uint64_t n_nreads = 0;
    DevBuf nread_idx;       // u32[n_nreads] read index
    DevBuf nread_ascii;     // u8[n_nreads][read_len]
    DevBuf nread_flag;      // u8[n] 0 = no N; 1 = N's, the byte path (k_copmem_match_n); 3 = at most 4 N's: their positions are in
                            // nread_npos and the dual kernel takes the read itself (the other schedules treat 3 like 1)
    DevBuf nread_npos;      // u32[n] four position bytes (0xFF = none) of the reads flagged 3 (pack.hip, k_npos_rows)
    uint64_t n_many = 0;    // reads flagged 1: more N's than the dual kernel takes
    std::vector<uint32_t> h_nidx; // host copy of nread_idx (ascending): modes d/i/e cut it per batch of reads

    // chunked upload state (pgrc_match_begin_reads / _append_ / _end_)
    bool up_open = false;
    uint64_t up_next = 0;
    std::vector<uint32_t> up_nidx;                  // reads with N seen so far (ascending)
    uint64_t up_nmany = 0;                          // ... of them with more than 4 N's
    std::vector<DevBuf> up_nchunks;                 // their ASCII rows, one device buffer per appended block that had some
    std::vector<uint64_t> up_nchunk_rows;

    // results
    DevBuf d_pos, d_rc, d_mism, d_hist, d_counters;
    bool have_results = false;
    uint64_t hist[256]{};
    uint64_t matched = 0;

    // copMEM index (rebuilt per pass, buffers reused)
    pgrc_copmem_params cp{};
    uint64_t npos = 0;
    DevBuf d_head;                      // ulonglong2[hash_size] bucket heads (one strand's table of 16-byte heads)
    // Where the ACTIVE set's heads really are: head h at head_ptr[head_slot(h, head_sh)] (headfmt.h).  head_sh = 0: d_head,
    // one table per strand; otherwise (round 4, runs that keep both strands' indexes) the pair table d_headpair, in which
    // the forward and the RC head of a bucket number share a line -- the dual kernel's two gathers of a seed are then ONE
    // line request and one translation (copmem.hip, "The dual kernel").
    ulonglong2 *head_ptr = nullptr;
    uint32_t head_sh = 0;
    DevBuf d_headpair;                  // ulonglong2[2 * hash_size], shared by both index sets (never swapped)
    uint32_t pair_gm = 3;               // pair table: buckets per group - 1 (a power of two minus one; headfmt.h)
    bool pair_build = false;            // pgrc_copmem_build_index writes into d_headpair (set around the builds of both strands)
    DevBuf d_skey[2], d_sval[2], d_sorttmp; // (bucket, entry) records: radix sort ping-pong + rocPRIM scratch (grow-only)
    const uint64_t *ent_ptr = nullptr;  // the sorted entries (one of d_sval[]): ent[] of the match kernel
    int index_strand = -1;  // which strand the buffers currently describe
    bool os_lds_allowed = false;        // idxsweep.hip's kernels were granted their dynamic LDS on this context's device
    // the screened schedule of a two-pass run (copmem.hip, "Exact-match screen") keeps both strands' indexes: a second
    // set of index buffers that swaps roles with the first, and per read the flag / position the screen found
    DevBuf alt_head, alt_skey[2], alt_sval[2], alt_sorttmp;
    const uint64_t *alt_ent_ptr = nullptr;
    int alt_index_strand = -1;
    ulonglong2 *alt_head_ptr = nullptr;
    uint32_t alt_head_sh = 0;
    DevBuf d_scr_pos, d_scr_flag;
    bool screen_broken = false;         // no room for the second set: the passes run as the reference orders them
    hipStream_t build_stream = nullptr; // the RC index is built beside the forward one (screened schedule)
    hipEvent_t build_ev[2]{};

    // pipelined hand-over (stream.hip): both indexes built ahead of the run; blocks of reads matched as they arrive
    bool idx_prepared = false;          // pgrc_match_prepare_index built (or is building) both strands' indexes of the current text
    uint64_t range_lo = 0, range_n = ~0ull;   // the match launchers work on reads [range_lo, range_lo + range_n) ...
    bool range_skip_n = false;          // ... and leave the reads with N to a later pass
    struct StreamBlock { uint64_t lo, cnt; hipEvent_t done; };
    bool st_on = false, st_dual = false;
    uint64_t *st_pos = nullptr;         // the caller's result arrays: a block's results are copied there as soon as they exist
    uint8_t *st_rc = nullptr, *st_mism = nullptr;
    DevBuf up_flag, up_lidx;            // append_reads_*: bad-symbol flag, positions of a block's N rows (kept: freeing a small buffer
                                        // goes to hipFree, which waits for the whole device -- i.e. for the matching of a streamed run)
    std::vector<DevBuf> st_keep;        // streamed run: small device buffers to give back at its end (see end_reads)
    hipEvent_t st_ready = nullptr;      // streamed run: indexes built and per-read state initialised (what the N passes wait for)
    hipEvent_t n_after = nullptr;       // when set: the N kernel's side stream waits for this event instead of the main stream's tail
    DevBuf up_stage[2];                 // staging areas of append_reads_* (grow-only: no allocation per call), used in turn
    hipStream_t up_stream[2] = {nullptr, nullptr};   // a streamed run uploads and unpacks beside the matching, on two streams in
    hipEvent_t up_ev[2] = {nullptr, nullptr};        // turn: the copy of chunk k+1 must not queue behind the unpacking of chunk k,
    uint64_t up_chunk = 0;                           // which waits for a free CU while the persistent match kernel of chunk k-1 runs
    std::thread st_worker;              // downloads finished blocks while the caller uploads the next ones
    std::mutex st_mu;
    std::condition_variable st_cv;
    std::deque<StreamBlock> st_q;
    bool st_quit = false;
    int st_err = 0;
    std::vector<std::pair<uint64_t, uint64_t>> st_nblocks;   // blocks holding reads with N: downloaded again at the end
    double st_t0 = 0;
    hipEvent_t st_tbase = nullptr;                           // PGRC_STREAM_TIMING: device times of the blocks' match launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> st_tev;

    // read-side seed index (modes d / i / e)
    DevBuf s_keys, s_vals, s_tab, s_hits, s_tmp, s_sort, s_seg, s_hv, s_hvu;
    DevBuf s_filter;        // modes d/i/e: one bit per slice of the key space (seedidx.hip)
    DevBuf s_nmask;                             // N masks of the reads with N (modes d/i/e)
    DevBuf s_best, s_rows;                      // the atomic-minimum reduction: one key per read, the batch's reads row by row (seedidx.hip 3c)

    // profiling
    bool profiling = false;
    pgrc_match_counters ctr{};
    hipEvent_t ev[16]{};
    bool have_events = false;
};

// HIP failure -> ABI error code: out of memory is PGRC_E_ALLOC, a missing / invalid device PGRC_E_NO_DEVICE, anything
// else (launch failure, invalid value, ...) PGRC_E_DEVICE
static inline int pgrc_hip_code(hipError_t e) {
    if (e == hipErrorOutOfMemory) return PGRC_E_ALLOC;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return PGRC_E_NO_DEVICE;
    return PGRC_E_DEVICE;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                 \
            return pgrc_hip_code(e__);                                                       \
        }                                                                                    \
    } while (0)

// Every ABI entry point runs on its context's device and leaves the caller's current device (e.g. torch's) as it
// found it.
struct PgrcDeviceScope {
    int prev = -1;
    bool ok = true;
    explicit PgrcDeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) prev = -1;                       // nothing to switch, nothing to restore
        else ok = hipSetDevice(dev) == hipSuccess;
    }
    ~PgrcDeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define PGRC_ON_DEVICE(ctx)                                                                  \
    PgrcDeviceScope dev_scope__((ctx)->device);                                              \
    if (!dev_scope__.ok) {                                                                   \
        (ctx)->err = "hipSetDevice(" + std::to_string((ctx)->device) + ") failed";           \
        return PGRC_E_NO_DEVICE;                                                             \
    }

int pgrc_buf_ensure(pgrc_match_ctx *c, DevBuf &b, size_t bytes);
void pgrc_buf_free(DevBuf &b);
void pgrc_buf_free_all(DevBuf *const *bufs, size_t count);   // one wait for the device for the whole batch

// api.hip: (re)allocates and clears both strands' text buffers for a text of G symbols
extern "C" int pgrc_pg_alloc(pgrc_match_ctx *c, uint64_t G);

// multi.hip: the front context's side of every entry point
int pgrc_multi_set_pg_ascii(pgrc_match_ctx *f, const char *pg, uint64_t G);
int pgrc_multi_set_pg_packed_device(pgrc_match_ctx *f, const void *d_words, uint64_t G);
int pgrc_multi_pack_pg_slice(pgrc_match_ctx *f, const char *pg, uint64_t count, void *d_words_out);
int pgrc_multi_begin_reads(pgrc_match_ctx *f, uint64_t n);
int pgrc_multi_append_reads(pgrc_match_ctx *f, const void *rows, uint64_t count, int32_t symbols /* 0 = ASCII */);
int pgrc_multi_end_reads(pgrc_match_ctx *f);
int pgrc_multi_set_reads_device(pgrc_match_ctx *f, const void *d_words, uint64_t n, uint64_t stride);
int pgrc_multi_init_results(pgrc_match_ctx *f);
int pgrc_multi_set_results(pgrc_match_ctx *f, const uint64_t *pos, const uint8_t *rc, const uint8_t *mism);
int pgrc_multi_run(pgrc_match_ctx *f, int first, int last);
int pgrc_multi_get_results(pgrc_match_ctx *f, uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t hist[256], uint64_t *matched);
int pgrc_multi_extract_mismatches(pgrc_match_ctx *f, const uint8_t *reversed_flags, uint64_t *cum, uint8_t *codes,
                                  uint16_t *offsets);
int pgrc_multi_export_index(pgrc_match_ctx *f, int strand, uint32_t *cumm, uint32_t *positions, uint64_t *count);
int pgrc_multi_export_pg(pgrc_match_ctx *f, int strand, uint32_t *words);
int pgrc_multi_set_profiling(pgrc_match_ctx *f, int enabled);
int pgrc_multi_get_counters(pgrc_match_ctx *f, pgrc_match_counters *out);
void pgrc_multi_destroy(pgrc_match_ctx *f);
struct PgrcShardView { pgrc_match_ctx *ctx; uint64_t lo, hi; };     // one shard of a multi-device context: reads [lo, hi)
std::vector<PgrcShardView> pgrc_multi_shards(pgrc_match_ctx *f);
void pgrc_export_drop_view(pgrc_match_ctx *front);   // export.hip: forget the gathered view (text, reads or results change)

// pack.hip
int pgrc_launch_pack_ascii(pgrc_match_ctx *c, const uint8_t *d_ascii, uint64_t count, uint32_t *d_words,
                           uint32_t *d_errflag);
int pgrc_launch_revcomp(pgrc_match_ctx *c, const uint32_t *d_fw, uint32_t *d_rc, uint64_t G);
int pgrc_launch_pack_reads_ascii(pgrc_match_ctx *c, const uint8_t *d_ascii, uint64_t first, uint64_t count,
                                 uint32_t L, uint32_t *d_words, uint64_t stride, uint8_t *d_nflag,
                                 uint32_t *d_errflag);
int pgrc_launch_repack_reads_ref(pgrc_match_ctx *c, const uint8_t *d_packed, uint64_t first, uint64_t count,
                                 uint32_t L, uint32_t *d_words, uint64_t stride);
int pgrc_launch_unpack_reads_acgnt(pgrc_match_ctx *c, const uint8_t *d_packed, uint64_t first, uint64_t count,
                                   uint32_t L, uint32_t *d_words, uint64_t stride, uint8_t *d_nflag, uint32_t *d_errflag);
int pgrc_launch_npos_rows(pgrc_match_ctx *c, const uint8_t *d_rows, int symbols, uint64_t first, uint64_t count, uint32_t L,
                          uint8_t *d_nflag, uint32_t *d_npos);
int pgrc_launch_nrows_ascii_acgnt(pgrc_match_ctx *c, const uint8_t *d_packed, const uint32_t *d_local_idx, uint64_t count,
                                  uint32_t L, uint8_t *d_ascii);

// copmem.hip
int pgrc_copmem_build_index(pgrc_match_ctx *c, int strand);
// idxsort.hip: the partition build of the same index (hand-written scatter passes, in-LDS finish of a partition)
uint32_t pgrc_ps_partition_bits(uint32_t hbits);
bool pgrc_ps_applicable(const pgrc_match_ctx *c, uint32_t hbits);
int pgrc_ps_scatter_front(pgrc_match_ctx *c, int strand, uint32_t hbits, uint32_t cb);
int pgrc_ps_finish(pgrc_match_ctx *c, const uint32_t *d_keys, const uint64_t *d_vals, uint32_t hbits, uint32_t cb, uint64_t *d_ent);
int pgrc_ps_finish_packed(pgrc_match_ctx *c, const uint64_t *d_recs, const uint32_t *d_pstart, uint32_t *d_slow, uint32_t np, uint32_t cb,
                          uint32_t rec_sh, uint64_t *d_ent);
uint64_t pgrc_ps_scan_blocks(uint64_t n);
int pgrc_ps_scan_u32(pgrc_match_ctx *c, uint32_t *d_io, uint64_t n, uint32_t *d_bsum);
// idxsweep.hip: the default front end (one-sweep scatter passes that hash the text themselves, 8-byte records)
bool pgrc_os_applicable(const pgrc_match_ctx *c, uint32_t hbits);
int pgrc_os_build_index(pgrc_match_ctx *c, int strand, uint32_t hbits);
int pgrc_copmem_match_pass(pgrc_match_ctx *c, int strand);
// stream.hip: a block of reads just arrived on the device (append_reads_*): match it now if streaming is on
int pgrc_stream_block_arrived(pgrc_match_ctx *c, uint64_t lo, uint64_t cnt, bool may_hold_n, int turn);
void pgrc_stream_abort(pgrc_match_ctx *c);
// api.hip
void pgrc_swap_index_sets(pgrc_match_ctx *c);
int pgrc_prepare_both_indexes(pgrc_match_ctx *c);
bool pgrc_dual_applies(const pgrc_match_ctx *c);
int pgrc_copmem_match_phase(pgrc_match_ctx *c, int strand, int phase);
int pgrc_copmem_match_dual(pgrc_match_ctx *c);
int pgrc_copmem_match_nreads(pgrc_match_ctx *c, int first, int last, bool only_many);
int pgrc_copmem_join_nreads(pgrc_match_ctx *c);
int pgrc_copmem_export_index(pgrc_match_ctx *c, uint32_t *h_cumm, uint32_t *h_positions, uint64_t *count);

// radix.hip: stable LSD radix sort of 64-bit records by the bit field [bit_lo, bit_hi) (hand-written; d_b / k_b / v_b: scratch of n
// records; `scratch` grows as needed; *sorted: wherever the last pass put them); the pairs form carries a 64-bit value per key
int pgrc_radix_sort_u64(pgrc_match_ctx *c, uint64_t *d_a, uint64_t *d_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi, DevBuf &scratch,
                        uint64_t **sorted);
int pgrc_radix_sort_pairs_u64(pgrc_match_ctx *c, uint64_t *k_a, uint64_t *k_b, uint64_t *v_a, uint64_t *v_b, uint64_t n, uint32_t bit_lo, uint32_t bit_hi,
                              DevBuf &scratch, uint64_t **ksorted, uint64_t **vsorted);
#define PGRC_RX_SEGMENT_MAX 8192u      // pairs a segment of pgrc_radix_sort_segments_pairs_u64 may hold (radix.hip RX_TILE)
int pgrc_radix_sort_segments_pairs_u64(pgrc_match_ctx *c, uint64_t *keys, uint64_t *vals, const uint32_t *seg, uint32_t nseg, uint32_t bit_lo, uint32_t bit_hi,
                                       uint32_t *ovl, uint32_t cap);

// seedidx.hip (modes d / i / e)
int pgrc_seedidx_run(pgrc_match_ctx *c, int first_strand, int last_strand);

// results.hip
int pgrc_launch_init_results(pgrc_match_ctx *c);
int pgrc_launch_hist(pgrc_match_ctx *c);
int pgrc_extract_mismatches(pgrc_match_ctx *c, const uint8_t *reversed_flags, uint64_t *cum, uint8_t *codes,
                            uint16_t *offsets);
int pgrc_extract_lists_device(pgrc_match_ctx *c, const uint8_t *d_revflags, bool lists, DevBuf &d_cum, DevBuf &d_codes, DevBuf &d_offs,
                              uint64_t *total_out);
