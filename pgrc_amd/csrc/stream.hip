// stream.hip -- the pipelined hand-over of a whole matching job (include/pgrc_match.h, "pipelined hand-over").
//
// The reference's call site (pgrc/pgrc-encoder.cpp:342-374 -> PgTools::mapReadsIntoPg -> matchConstantLengthReads,
// matching/ReadsMatchers.cpp:162-172) hands over a finished pseudogenome and a packed read set and wants three result
// vectors back.  Done one step after the other -- text up, reads up, two index builds, matching, results down -- the
// device idles during the copies and the PCIe link during the matching: 226 M reads/s for the encoder's LQ + N sum set
// against 950 M reads/s for the matching alone.  Here the steps overlap:
//   * pgrc_match_prepare_index starts both strands' index builds as soon as the text is on the device (their own streams),
//     so they run beneath the upload of the reads;
//   * after pgrc_match_stream_begin every block of rows that pgrc_match_append_reads_* brings is matched as soon as it is
//     unpacked (upload + unpacking on one stream, matching on another), while the caller's thread is already copying the
//     next block: a read's result depends on (read, text, index) only -- the reference's own `omp parallel for` over the
//     reads (ReadsMatchers.cpp:426-428) -- so blocks are independent;
//   * a worker thread copies a block's results into the caller's arrays as soon as its match is done (PCIe is full duplex);
//   * pgrc_match_stream_end runs what needs the whole set (the reads with N: their side list is complete only after
//     pgrc_match_end_reads; the histogram) and waits for the last download.
// Results are those of pgrc_match_init_results + pgrc_match_run(ctx, 1): tests/test_gpu_stream.py compares them bit for bit.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "ctx.h"

// PGRC_STREAM_TIMING=1: host-clock milestones of a streamed run on stderr
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define ST_MARK(c, what) do { if ((c)->opt.stream_timing) fprintf(stderr, "pgrc stream: %8.2f ms  %s\n", (now_s() - (c)->st_t0) * 1e3, what); } while (0)

extern "C" int pgrc_match_prepare_index(pgrc_match_ctx *c, int32_t both_strands) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) { c->err = "prepare_index: single-device contexts only"; return PGRC_E_STATE; }
    if (c->prm.mode != 'c' || !both_strands) { c->err = "prepare_index: two-strand runs of mode c only"; return PGRC_E_STATE; }
    if (!c->have_pg) { c->err = "prepare_index: set the pseudogenome first"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    return pgrc_prepare_both_indexes(c);
}

// the worker: downloads the results of finished blocks into the caller's arrays
static void stream_worker(pgrc_match_ctx *c) {
    hipStream_t down = nullptr;               // (its own stream: a plain hipMemcpy would queue behind the matching of later blocks)
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&down, hipStreamNonBlocking) != hipSuccess) {
        std::lock_guard<std::mutex> g(c->st_mu);
        c->st_err = PGRC_E_NO_DEVICE;
        return;
    }
    for (;;) {
        pgrc_match_ctx::StreamBlock b;
        {
            std::unique_lock<std::mutex> g(c->st_mu);
            c->st_cv.wait(g, [&]() { return c->st_quit || !c->st_q.empty(); });
            if (c->st_q.empty()) break;
            b = c->st_q.front();
            c->st_q.pop_front();
        }
        hipError_t he = b.done ? hipEventSynchronize(b.done) : hipSuccess;
        if (he == hipSuccess && b.cnt) {
            he = hipMemcpyAsync(c->st_pos + b.lo, (const uint64_t *)c->d_pos.p + b.lo, b.cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(c->st_rc + b.lo, (const uint8_t *)c->d_rc.p + b.lo, b.cnt, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(c->st_mism + b.lo, (const uint8_t *)c->d_mism.p + b.lo, b.cnt, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipStreamSynchronize(down);
        }
        if (c->opt.stream_timing) fprintf(stderr, "pgrc stream: %8.2f ms  results of block [%llu, +%llu) on the host\n", (now_s() - c->st_t0) * 1e3, (unsigned long long)b.lo, (unsigned long long)b.cnt);
        if (b.done) (void)hipEventDestroy(b.done);
        if (he != hipSuccess) {
            std::lock_guard<std::mutex> g(c->st_mu);
            if (!c->st_err) c->st_err = pgrc_hip_code(he);
        }
    }
    (void)hipStreamDestroy(down);
}

static void stop_worker(pgrc_match_ctx *c) {
    if (!c->st_worker.joinable()) return;
    {
        std::lock_guard<std::mutex> g(c->st_mu);
        c->st_quit = true;
    }
    c->st_cv.notify_all();
    c->st_worker.join();
    for (pgrc_match_ctx::StreamBlock &b : c->st_q)
        if (b.done) (void)hipEventDestroy(b.done);
    c->st_q.clear();
}

static void release_kept(pgrc_match_ctx *c) {
    for (DevBuf &b : c->st_keep) pgrc_buf_free(b);
    c->st_keep.clear();
}

void pgrc_stream_abort(pgrc_match_ctx *c) {
    release_kept(c);
    if (!c->st_on && !c->st_worker.joinable()) return;
    stop_worker(c);
    c->st_on = false;
    c->range_lo = 0;
    c->range_n = ~0ull;
    c->range_skip_n = false;
}

extern "C" int pgrc_match_stream_begin(pgrc_match_ctx *c, uint64_t *pos, uint8_t *rc, uint8_t *mism) {
    if (!c || !pos || !rc || !mism) return PGRC_E_PARAM;
    if (c->multi) { c->err = "stream_begin: single-device contexts only"; return PGRC_E_STATE; }
    if (c->prm.mode != 'c' || c->prm.min_mismatches != 0) { c->err = "stream_begin: mode c with min_mismatches == 0 only"; return PGRC_E_STATE; }
    if (!c->have_pg || !c->up_open || c->up_next != 0) { c->err = "stream_begin: call it after set_pg and begin_reads, before the first rows"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    pgrc_stream_abort(c);
    int e;
    if (!(c->idx_prepared && c->index_strand == 1 && c->alt_index_strand == 0) && (e = pgrc_prepare_both_indexes(c))) return e;
    // the per-read state of a first phase (DefaultReadsMatcher::initMatching, ReadsMatchers.cpp:97-105); have_reads is not
    // set yet, so not through pgrc_match_init_results
    if ((e = pgrc_launch_init_results(c))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_scr_pos, c->n * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, c->d_scr_flag, c->n ? c->n : 1))) return e;
    HIP_TRY(c, hipMemsetAsync(c->d_scr_flag.p, 0, c->n ? c->n : 1, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, 32 * sizeof(uint64_t), c->stream));
    memset(&c->ctr, 0, sizeof c->ctr);
    if (!c->st_ready) HIP_TRY(c, hipEventCreateWithFlags(&c->st_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->st_ready, c->stream));              // both indexes built, per-read state initialised
    // (Round 5, tried and dropped, profiles/r05_boundary_stream_variants.txt: the blocks' launches on two streams in turn, so that
    //  one block's launch fills the CUs while the previous one drains -- no gain, the two launches slow each other down; one CU per
    //  XCD masked out of the match stream for the upload side's kernels -- 10 % slower; a first block of a quarter of the size -- the
    //  first copy's blit kernels wait for the index builds anyway; five blocks per CU instead of eight -- no difference.)
    // (Tried and dropped: the blocks' match kernels on a CU-masked stream that leaves two CUs per XCD to the small kernels of
    //  the upload side, which otherwise wait for a persistent launch to end -- the whole job was 2 % slower, A/B in one process.)
    for (int k = 0; k < 2; k++) {
        if (c->up_stream[k]) continue;
        hipError_t he = hipStreamCreateWithFlags(&c->up_stream[k], hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&c->up_ev[k], hipEventDisableTiming);
        if (he != hipSuccess) { c->up_stream[k] = nullptr; c->err = std::string("stream_begin: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
    }
    // (begin_reads cleared the N flags with a blocking call when index builds were in flight: the upload streams need not
    //  -- and must not -- wait for the main stream, where those builds are queued)
    c->st_pos = pos;
    c->st_rc = rc;
    c->st_mism = mism;
    c->st_dual = pgrc_dual_applies(c);
    c->st_quit = false;
    c->st_err = 0;
    c->st_nblocks.clear();
    c->st_worker = std::thread(stream_worker, c);
    c->st_on = true;
    c->have_results = false;
    c->st_t0 = now_s();
    if (c->opt.stream_timing) {
        if (!c->st_tbase) HIP_TRY(c, hipEventCreate(&c->st_tbase));
        HIP_TRY(c, hipEventRecord(c->st_tbase, c->stream));
    }
    ST_MARK(c, "stream_begin done");
    return PGRC_OK;
}

// rows [lo, lo + cnt) are unpacked (queued on the upload stream): match them on the main stream, hand them to the worker
int pgrc_stream_block_arrived(pgrc_match_ctx *c, uint64_t lo, uint64_t cnt, bool may_hold_n, int turn) {
    if (!c->st_on || !cnt) return PGRC_OK;
    if (c->opt.stream_timing) fprintf(stderr, "pgrc stream: %8.2f ms  block [%llu, +%llu) copied and queued for unpacking\n", (now_s() - c->st_t0) * 1e3, (unsigned long long)lo, (unsigned long long)cnt);
    HIP_TRY(c, hipEventRecord(c->up_ev[turn], c->up_stream[turn]));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->up_ev[turn], 0));
    // the kernels' read cursors start at zero for every launch
    HIP_TRY(c, hipMemsetAsync((uint64_t *)c->d_counters.p + 16, 0, 3 * sizeof(uint64_t), c->stream));
    c->range_lo = lo;
    c->range_n = cnt;
    c->range_skip_n = true;
    int e;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->opt.stream_timing && c->st_tbase) {
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->stream));
    }
    if (c->st_dual) {
        e = pgrc_copmem_match_dual(c);                               // active set = strand 1, alternate = strand 0
    } else {
        pgrc_swap_index_sets(c);                                     // the two passes in the reference's order
        e = pgrc_copmem_match_phase(c, 0, 0);
        pgrc_swap_index_sets(c);
        if (!e) e = pgrc_copmem_match_phase(c, 1, 0);
    }
    c->range_lo = 0;
    c->range_n = ~0ull;
    c->range_skip_n = false;
    if (t1) {
        (void)hipEventRecord(t1, c->stream);
        c->st_tev.emplace_back(t0, t1);
    }
    if (e) return e;
    hipEvent_t done = nullptr;
    HIP_TRY(c, hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(done, c->stream));
    if (may_hold_n) c->st_nblocks.emplace_back(lo, cnt);
    {
        std::lock_guard<std::mutex> g(c->st_mu);
        c->st_q.push_back({lo, cnt, done});
    }
    c->st_cv.notify_one();
    return PGRC_OK;
}

extern "C" int pgrc_match_stream_end(pgrc_match_ctx *c, uint64_t hist[256], uint64_t *matched) {
    if (!c) return PGRC_E_PARAM;
    if (!c->st_on) { c->err = "stream_end: no streamed run in progress"; return PGRC_E_STATE; }
    if (c->up_open || !c->have_reads) { pgrc_stream_abort(c); c->err = "stream_end: call pgrc_match_end_reads first"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    int e = PGRC_OK;
    ST_MARK(c, "stream_end called");
    // the reads with N: their kernel over the side list, forward strand then RC strand (ReadsMatchers.cpp:162-172), on the
    // side stream and BESIDE the blocks that are still being matched (one lane per read, latency-bound: it fits in; the
    // blocks' kernels skip the reads with N, so the two write disjoint reads); the main stream joins it at its tail
    if (c->n_nreads) {
        c->n_after = c->st_ready;
        e = pgrc_copmem_match_nreads(c, 0, 1, c->st_dual);   // one launch: a read's forward query, then its RC query (the blocks' dual kernel took the reads with at most 4 N)
        if (!e) e = pgrc_copmem_join_nreads(c);
        c->n_after = nullptr;
    }
    if (c->opt.stream_timing) {
        (void)hipStreamSynchronize(c->stream);
        ST_MARK(c, "all blocks matched, reads with N done");
        for (auto &ev : c->st_tev) {                                 // (device clock, from the event recorded in stream_begin)
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, c->st_tbase, ev.first);
            (void)hipEventElapsedTime(&b, c->st_tbase, ev.second);
            fprintf(stderr, "pgrc stream:   a block's match launches on the device: %8.2f .. %8.2f ms after the main stream reached stream_begin's mark\n", a, b);
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        c->st_tev.clear();
    }
    if (!e) e = pgrc_launch_hist(c);                                 // synchronises the main stream: every block is done
    ST_MARK(c, "histogram done");
    if (!e && c->n_nreads) {
        std::lock_guard<std::mutex> g(c->st_mu);
        for (const auto &b : c->st_nblocks) c->st_q.push_back({b.first, b.second, nullptr});
    }
    stop_worker(c);                                                  // (drains the queue first)
    ST_MARK(c, "last download done");
    c->st_on = false;
    release_kept(c);
    if (!e && c->st_err) { e = c->st_err; c->err = "stream_end: a result download failed"; }
    if (e) return e;
    uint64_t scr[8];
    HIP_TRY(c, hipMemcpy(scr, (const uint64_t *)c->d_counters.p + 24, sizeof scr, hipMemcpyDeviceToHost));
    if (c->st_dual) {
        for (int k = 0; k < 5; k++) c->ctr.dual[k] = scr[k];
        c->ctr.redo_reads = scr[5];
        c->ctr.dual_seed_probes = scr[6];
        c->ctr.screened = 2;
    }
    c->ctr.ms_total = (float)((now_s() - c->st_t0) * 1e3);           // (host clock: stream_begin .. here)
    c->have_results = true;
    if (hist) memcpy(hist, c->hist, sizeof c->hist);
    if (matched) *matched = c->matched;
    return PGRC_OK;
}
