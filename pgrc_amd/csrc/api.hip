// api.hip -- the C ABI of libpgrc_match.so (include/pgrc_match.h): context, host<->HBM staging of the
// pseudogenome and the reads, the two-pass driver (DefaultReadsMatcher::matchConstantLengthReads,
// matching/ReadsMatchers.cpp:162-172) and result retrieval.  No CPU fallback exists: every compute
// entry point needs a HIP device.
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "ctx.h"
#include "devutil.h"

// Large device buffers that a context gives back are kept for the next context of the process instead of going back to
// the driver: re-allocating tens of GB right after freeing them can stall for seconds (the driver scrubs freed memory
// before it hands it out again: tools/ubench/realloc.hip, and 1.3-2.2 s seen for the second matcher of a process in
// tools/boundary_c3.py) -- and PgRC creates one matcher per phase.  At most PGRC_DEVICE_POOL_GB (default 96, 0 = off) are held
// per process; pgrc_match_trim_device_memory() returns them.
namespace {
struct PooledBuf { void *p; size_t bytes; int device; };
std::mutex g_pool_mu;
std::vector<PooledBuf> g_pool;
size_t g_pool_bytes = 0;
size_t pool_cap() {
    static const size_t cap = []() {
        const char *v = getenv("PGRC_DEVICE_POOL_GB");
        return (size_t)(v ? std::max(0ll, atoll(v)) : 96ll) << 30;
    }();
    return cap;
}
const size_t POOL_MIN = 64ull << 20;
}

int pgrc_buf_ensure(pgrc_match_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.p && b.bytes >= bytes) return PGRC_OK;
    pgrc_buf_free(b);
    int dev = -1;
    if (bytes >= POOL_MIN && hipGetDevice(&dev) == hipSuccess) {
        std::lock_guard<std::mutex> g(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t k = 0; k < g_pool.size(); k++)        // smallest fitting buffer of this device, not more than twice the size
            if (g_pool[k].device == dev && g_pool[k].bytes >= bytes && g_pool[k].bytes <= 2 * bytes && (best == g_pool.size() || g_pool[k].bytes < g_pool[best].bytes))
                best = k;
        if (best != g_pool.size()) {
            b.p = g_pool[best].p;
            b.bytes = g_pool[best].bytes;
            g_pool_bytes -= b.bytes;
            g_pool.erase(g_pool.begin() + (long)best);
            return PGRC_OK;
        }
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e == hipErrorOutOfMemory && pgrc_match_trim_device_memory() > 0) {   // what the pool holds may be what is missing
        (void)hipGetLastError();
        e = hipMalloc(&b.p, bytes);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        c->err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        return pgrc_hip_code(e);
    }
    b.bytes = bytes;
    static const bool debug_alloc = getenv("PGRC_DEBUG_ALLOC") != nullptr;     // (process-wide, read once)
    if (debug_alloc) fprintf(stderr, "pgrc alloc ctx %p buf %p: %zu bytes at %p\n", (void *)c, (void *)&b, bytes, b.p);
    return PGRC_OK;
}

// quiesced: the caller has waited for the device since the buffer was last used (pgrc_buf_free_all: one wait for a batch)
static void buf_free(DevBuf &b, bool quiesced) {
    if (b.p) {
        int dev = -1;
        bool kept = false;
        if (b.bytes >= POOL_MIN && hipGetDevice(&dev) == hipSuccess) {
            // A pooled buffer can be handed to any context, stream or thread at once: nothing may still be using it.  hipFree
            // used to give that guarantee by waiting for the device; so does this (large buffers are only given back on
            // destroy, on growth and on error paths -- never between the blocks of a streamed run, which keeps what it must free
            // until its end).
            if (!quiesced) (void)hipDeviceSynchronize();
            std::lock_guard<std::mutex> g(g_pool_mu);
            if (g_pool_bytes + b.bytes <= pool_cap()) {
                g_pool.push_back({b.p, b.bytes, dev});
                g_pool_bytes += b.bytes;
                kept = true;
            }
        }
        if (!kept) (void)hipFree(b.p);
    }
    b.p = nullptr;
    b.bytes = 0;
}

void pgrc_buf_free(DevBuf &b) { buf_free(b, false); }

// several buffers at once: ONE wait for the device (if any of them is large enough for the pool), not one per buffer
void pgrc_buf_free_all(DevBuf *const *bufs, size_t count) {
    bool any = false;
    for (size_t k = 0; k < count; k++) any |= bufs[k]->p && bufs[k]->bytes >= POOL_MIN;
    if (any) (void)hipDeviceSynchronize();
    for (size_t k = 0; k < count; k++) buf_free(*bufs[k], true);
}

extern "C" uint64_t pgrc_match_trim_device_memory(void) {
    std::vector<PooledBuf> take;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        take.swap(g_pool);
        g_pool_bytes = 0;
    }
    uint64_t freed = 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (const PooledBuf &pb : take) {
        if (pb.device != cur) (void)hipSetDevice(pb.device);
        (void)hipFree(pb.p);
        freed += pb.bytes;
        if (pb.device != cur && cur >= 0) (void)hipSetDevice(cur);
    }
    return freed;
}

static thread_local std::string g_create_err; // reported by pgrc_match_last_error(NULL)

// The environment, read once per context (ctx.h, PgrcOptions; include/pgrc_match.h lists the variables)
PgrcOptions pgrc_options_from_env() {
    PgrcOptions o;
    auto flag = [](const char *name) -> int { const char *v = getenv(name); return (v && (v[0] == '0' || v[0] == '1')) ? v[0] - '0' : -1; };
    auto num = [](const char *name) -> long long { const char *v = getenv(name); return (v && v[0]) ? atoll(v) : -1; };
    o.dual = flag("PGRC_DUAL");
    o.screen = flag("PGRC_SCREEN");
    o.early_stop = flag("PGRC_EARLY_STOP") != 0;
    o.builds_in_turn = flag("PGRC_BUILD_STREAMS") == 1;
    if (const char *hp = getenv("PGRC_HEAD_PAIR")) o.head_pair = hp[0] == '0' ? 0u : (hp[0] >= '1' && hp[0] <= '4') ? 1u << (hp[0] - '1') : 4u;
    if (const char *is = getenv("PGRC_INDEX_SORT")) o.index_front = !strcmp(is, "own") ? 1 : 0;
    if (const char *fi = getenv("PGRC_INDEX_FINISH")) o.index_finish_general = !strcmp(fi, "general");
    o.index_cfg = (int)num("PGRC_INDEX_CFG");
    o.match_stage = flag("PGRC_MATCH_STAGE") != 0;
    o.nread_inline = flag("PGRC_NREAD_INLINE") != 0;
    o.force_pos64 = flag("PGRC_FORCE_POS64") == 1;
    o.test_no_second_index = getenv("PGRC_TEST_NO_SECOND_INDEX") != nullptr;
    o.stream_timing = getenv("PGRC_STREAM_TIMING") != nullptr;
    o.host_pack = flag("PGRC_HOST_PACK") != 0;
    if (num("PGRC_HOST_THREADS") > 0) o.host_threads = (uint32_t)std::min<long long>(256, num("PGRC_HOST_THREADS"));
    if (num("PGRC_UPLOAD_CHUNK_MB") > 0) o.upload_chunk_mb = (uint64_t)std::min<long long>(4096, num("PGRC_UPLOAD_CHUNK_MB"));
    o.seed_filter = flag("PGRC_SEED_FILTER");
    if (num("PGRC_TEST_SEGMENT_TOP_BITS") > 0) o.test_segment_top_bits = (uint32_t)std::min<long long>(64, num("PGRC_TEST_SEGMENT_TOP_BITS")) & ~7u;
    if (const char *hf = getenv("PGRC_SEED_HEAVY_FORM")) o.seed_heavy_form = !strcmp(hf, "window") ? 0 : 1;
    if (const char *ss = getenv("PGRC_SEED_SORT")) o.seed_sort = !strcmp(ss, "full") ? 0 : !strcmp(ss, "segments") ? 1 : -1;
    if (num("PGRC_SEED_HEAVY") > 0) o.seed_heavy = (uint32_t)std::min<long long>(4096, num("PGRC_SEED_HEAVY"));
    if (num("PGRC_SEED_READ_BATCH") > 0) o.seed_read_batch = (uint64_t)num("PGRC_SEED_READ_BATCH");
    if (num("PGRC_SEED_SEGMENT") > 0) o.seed_segment = (uint64_t)num("PGRC_SEED_SEGMENT");
    if (num("PGRC_MEM_EVENT_CAP") > 0) o.mem_event_cap = (uint64_t)num("PGRC_MEM_EVENT_CAP");
    if (const char *ag = getenv("PGRC_ALLGATHER")) o.allgather = !strcmp(ag, "rccl") ? 1 : !strcmp(ag, "copy") ? 2 : 0;
    o.dual_variant = (int)num("PGRC_DUAL_VARIANT");
    return o;
}

// ---- host-side packing of an ASCII text to 2 bits per symbol (A0 C1 G2 T3; symbol i at bits 2 (i mod 16) of word i / 16: ctx.h)
// returns false when a symbol outside ACGT was met (the words are then garbage there)
static bool pack_ascii_scalar(const uint8_t *src, uint64_t count, uint32_t *dst) {
    bool ok = true;
    for (uint64_t w = 0; w * 16 < count; w++) {
        uint32_t word = 0;
        const uint64_t m = std::min<uint64_t>(16, count - w * 16);
        for (uint64_t k = 0; k < m; k++) {
            const uint32_t ch = src[w * 16 + k];
            uint32_t x = (ch >> 1) & 3u;
            x ^= x >> 1;                                      // A0 C1 G2 T3
            ok &= ch == (uint32_t)"ACGT"[x];
            word |= x << (2 * k);
        }
        dst[w] = word;
    }
    return ok;
}

#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static bool pack_ascii_avx2(const uint8_t *src, uint64_t count, uint32_t *dst) {
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i w1 = _mm256_set1_epi16(0x0401), w2 = _mm256_set1_epi32(0x00100001);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i allok = _mm256_set1_epi8(-1);
    uint64_t i = 0;
    for (; i + 32 <= count; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
        const __m256i code = _mm256_xor_si256(_mm256_and_si256(_mm256_srli_epi16(v, 1), three), _mm256_and_si256(_mm256_srli_epi16(v, 2), one));
        allok = _mm256_and_si256(allok, _mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, code), v));
        const __m256i b4 = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w1), w2);     // one packed byte in the low byte of every dword
        const __m256i sh = _mm256_shuffle_epi8(b4, pick);
        dst[i / 16] = (uint32_t)_mm256_extract_epi32(sh, 0);
        dst[i / 16 + 1] = (uint32_t)_mm256_extract_epi32(sh, 4);
    }
    bool ok = _mm256_movemask_epi8(allok) == -1;
    if (i < count) ok &= pack_ascii_scalar(src + i, count - i, dst + i / 16);
    return ok;
}
#endif

// `threads` host threads (0: up to 16 of the machine's), each a contiguous range of whole words
bool pgrc_pack_ascii_host(const uint8_t *src, uint64_t count, uint32_t *dst, uint32_t threads) {
    auto one = [](const uint8_t *s, uint64_t n, uint32_t *d) -> bool {
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) return pack_ascii_avx2(s, n, d);
#endif
        return pack_ascii_scalar(s, n, d);
    };
    uint32_t T = threads ? threads : std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency()));
    const uint64_t words = (count + 15) / 16;
    if (words < (1u << 16)) T = 1;
    if (T == 1) return one(src, count, dst);
    std::vector<std::thread> th;
    std::vector<char> okv(T, 1);
    const uint64_t per = (words + T - 1) / T;
    for (uint32_t t = 0; t < T; t++) {
        const uint64_t w0 = std::min(words, t * per), w1 = std::min(words, w0 + per);
        if (w0 == w1) continue;
        th.emplace_back([&, t, w0, w1]() { okv[t] = one(src + w0 * 16, std::min(count, w1 * 16) - w0 * 16, dst + w0) ? 1 : 0; });
    }
    for (std::thread &x : th) x.join();
    bool ok = true;
    for (char v : okv) ok &= v != 0;
    return ok;
}

static int isqrt_floor(int v) {
    int r = 0;
    while ((r + 1) * (r + 1) <= v) r++;
    return r;
}

extern "C" {

const char *pgrc_match_version(void) { return "pgrc_match 0.1 (gfx950)"; }

// CopMEMMatcher::initParams / calcCoprimes, matching/copmem/CopMEMMatcher.cpp:69-96, :111-137
int pgrc_match_copmem_params(uint32_t seed_len, uint64_t pg_len, pgrc_copmem_params *out) {
    const int L = (int)seed_len;
    if (L < 24) return PGRC_E_SEED_SHORT;
    int K = L > 110 ? 56 : L > 62 ? 44 : L > 53 ? 40 : L > 46 ? 36 : L > 42 ? 32 : L > 32 ? 28 : (L / 4 - 1) * 4;
    const int kmml = (L / 4 - 1) * 4;
    if (kmml < K) K = kmml;
    const int t = L - K + 1;
    if (t <= 0) return PGRC_E_PARAM;
    int k1, k2;
    if (t >= 20) {
        k1 = isqrt_floor(t) + 1;
        k2 = k1 - 1;
        if (k1 * k2 > t) { --k1; --k2; }
    } else if (t >= 15) { k1 = 5; k2 = 3; }
    else if (t >= 12) { k1 = 4; k2 = 3; }
    else if (t >= 10) { k1 = 5; k2 = 2; }
    else if (t >= 6) { k1 = 3; k2 = 2; }
    else { k1 = t; k2 = 1; }
    uint32_t hs;
    int i = 24;
    do { hs = 1u << (i++); } while (i <= 31 && (uint64_t)hs < pg_len / (uint64_t)k1);
    out->K = K; out->k1 = k1; out->k2 = k2; out->hash_size = hs;
    return PGRC_OK;
}

// mapReadsIntoPg, matching/ReadsMatchers.cpp:699-740
int pgrc_match_derive_params(uint32_t read_len, uint32_t seed_len, uint32_t min_chars_per_mismatch, char mode_char,
                             pgrc_match_params *out) {
    if (!out || read_len == 0 || read_len > 255 || seed_len == 0 || min_chars_per_mismatch == 0) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    out->read_len = read_len;
    out->max_mismatches = (uint8_t)(read_len / min_chars_per_mismatch);
    out->seed_len = std::min(seed_len, read_len);
    const bool upper = mode_char >= 'A' && mode_char <= 'Z';
    const char lower = upper ? (char)(mode_char - 'A' + 'a') : mode_char;
    out->min_mismatches = upper ? out->max_mismatches : 0;
    out->device = -1;
    if (out->seed_len == read_len) out->mode = (lower == 'c') ? 'c' : 'e';
    else if (lower == 'c' || lower == 'd' || lower == 'i') out->mode = lower;
    else return PGRC_E_MODE;
    return PGRC_OK;
}

uint32_t pgrc_match_words_per_read(uint32_t read_len) { return (read_len + 15) / 16; }

int pgrc_match_create(const pgrc_match_params *p, pgrc_match_ctx **out) {
    if (!p || !out) return PGRC_E_PARAM;
    *out = nullptr;
    if (p->read_len == 0 || p->read_len > 255 || p->seed_len == 0 || p->seed_len > p->read_len) return PGRC_E_PARAM;
    if (p->mode != 'c' && p->mode != 'd' && p->mode != 'i' && p->mode != 'e') return PGRC_E_MODE;
    if (p->mode == 'c' && p->seed_len < 24) return PGRC_E_SEED_SHORT;
    // the reference's u8 mismatch accumulator (CopMEMMatcher.cpp:524-534) cannot wrap while kmax <= 247
    if (p->mode == 'c' && p->max_mismatches > 247) return PGRC_E_PARAM;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev == 0) {
        g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(he) + " (devices: " + std::to_string(ndev) + ")";
        return PGRC_E_NO_DEVICE;
    }
    int dev = p->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    PgrcDeviceScope scope(dev < ndev ? dev : 0);   // the caller's current device is restored on return
    if (dev >= ndev || !scope.ok) {
        g_create_err = "hipSetDevice(" + std::to_string(dev) + ") failed (devices: " + std::to_string(ndev) + ")";
        return PGRC_E_NO_DEVICE;
    }
    (void)hipGetLastError(); // start from a clean sticky-error state
    pgrc_match_ctx *c = new pgrc_match_ctx();
    c->prm = *p;
    c->opt = pgrc_options_from_env();
    c->device = dev;
    c->nw = (p->read_len + 15) / 16;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
    int be;
    if ((be = pgrc_buf_ensure(c, c->d_hist, 256 * sizeof(uint64_t))) || (be = pgrc_buf_ensure(c, c->d_counters, 32 * sizeof(uint64_t)))) {
        g_create_err = c->err;
        pgrc_buf_free(c->d_hist);
        delete c;
        return be;
    }
    *out = c;
    return PGRC_OK;
}

void pgrc_match_destroy(pgrc_match_ctx *c) {
    if (!c) return;
    if (c->multi) { pgrc_multi_destroy(c); return; }
    PgrcDeviceScope scope(c->device);
    pgrc_stream_abort(c);
    // nothing of this context may still be running when its buffers go to the pool, from where the next context takes them
    // (index builds started ahead and never used, an abandoned streamed run)
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 2; k++) {
        if (c->up_stream[k]) {
            (void)hipStreamDestroy(c->up_stream[k]);
            (void)hipEventDestroy(c->up_ev[k]);
        }
        pgrc_buf_free(c->up_stage[k]);
    }
    pgrc_buf_free(c->up_flag);
    pgrc_buf_free(c->up_lidx);
    if (c->st_ready) (void)hipEventDestroy(c->st_ready);
    if (c->st_tbase) (void)hipEventDestroy(c->st_tbase);
    if (c->side_stream) {
        (void)hipStreamDestroy(c->side_stream);
        (void)hipEventDestroy(c->side_ev[0]);
        (void)hipEventDestroy(c->side_ev[1]);
    }
    if (c->build_stream) {
        (void)hipStreamDestroy(c->build_stream);
        (void)hipEventDestroy(c->build_ev[0]);
        (void)hipEventDestroy(c->build_ev[1]);
    }
    DevBuf *bufs[] = {&c->pg2[0], &c->pg2[1], &c->reads_own, &c->nread_idx, &c->nread_ascii, &c->nread_flag, &c->nread_npos, &c->d_pos,
                      &c->d_rc, &c->d_mism, &c->d_hist, &c->d_counters, &c->d_head, &c->d_headpair, &c->d_skey[0], &c->d_skey[1], &c->d_sval[0], &c->d_sval[1], &c->d_sorttmp,
                      &c->alt_head, &c->alt_skey[0], &c->alt_skey[1], &c->alt_sval[0], &c->alt_sval[1], &c->alt_sorttmp, &c->d_scr_pos, &c->d_scr_flag,
                      &c->s_keys, &c->s_filter, &c->s_vals, &c->s_tab, &c->s_hits, &c->s_tmp, &c->s_sort, &c->s_seg, &c->s_hv, &c->s_hvu, &c->s_nmask, &c->s_best, &c->s_rows};
    pgrc_buf_free_all(bufs, sizeof bufs / sizeof bufs[0]);       // (the device is idle: waited for above)
    for (DevBuf &b : c->up_nchunks) pgrc_buf_free(b);
    if (c->have_events)
        for (auto &e : c->ev) (void)hipEventDestroy(e);
    delete c;
}

const char *pgrc_match_last_error(const pgrc_match_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int pgrc_match_set_stream(pgrc_match_ctx *c, void *s) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) { c->err = "set_stream: a multi-device context runs on one stream per device"; return PGRC_E_PARAM; }
    c->stream = (hipStream_t)s;
    return PGRC_OK;
}

int pgrc_match_set_profiling(pgrc_match_ctx *c, int enabled) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_set_profiling(c, enabled);
    PGRC_ON_DEVICE(c);
    if (enabled && !c->have_events) {
        for (auto &e : c->ev) HIP_TRY(c, hipEventCreate(&e));
        c->have_events = true;
    }
    c->profiling = enabled != 0;
    return PGRC_OK;
}

int pgrc_match_reload_options(pgrc_match_ctx *c) {
    if (!c) return PGRC_E_PARAM;
    c->opt = pgrc_options_from_env();
    if (c->multi)
        for (const PgrcShardView &sv : pgrc_multi_shards(c)) sv.ctx->opt = c->opt;
    return PGRC_OK;
}

// ------------------------------------------------------------------ pseudogenome

int pgrc_pg_alloc(pgrc_match_ctx *c, uint64_t G) {
    if (G + 256 >= (1ull << 40)) {
        c->err = "pseudogenome of 2^40 symbols or more is not supported (entries keep 40 position bits)";
        return PGRC_E_PARAM;
    }
    if (G < c->prm.read_len) {
        c->err = "pseudogenome shorter than a read";
        return PGRC_E_PARAM;
    }
    if (c->idx_prepared) {                       // index builds started ahead of a run may still be reading the old text
        if (c->build_stream) (void)hipStreamSynchronize(c->build_stream);
        (void)hipStreamSynchronize(c->stream);
    }
    if (G != c->G) c->screen_broken = false;     // another text: the second index set may fit now
    c->G = G;
    c->pg_words = (G + 15) / 16;
    const size_t bytes = (c->pg_words + PGRC_PG_PAD_WORDS) * sizeof(uint32_t);
    int e;
    for (int s = 0; s < 2; s++) {
        if ((e = pgrc_buf_ensure(c, c->pg2[s], bytes))) return e;
        HIP_TRY(c, hipMemsetAsync(c->pg2[s].p, 0, bytes, c->stream));
    }
    c->have_pg = false;
    c->have_rc = false;
    c->index_strand = -1;
    c->alt_index_strand = -1;
    c->idx_prepared = false;
    if (c->prm.mode == 'c') {
        int r = pgrc_match_copmem_params(c->prm.seed_len, G, &c->cp);
        if (r) { c->err = "copMEM parameter derivation failed (seed too short?)"; return r; }
    }
    return PGRC_OK;
}

int pgrc_match_pack_pg_slice(pgrc_match_ctx *c, const char *pg, uint64_t count, void *d_words_out) {
    if (!c || (!pg && count) || !d_words_out) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_pack_pg_slice(c, pg, count, d_words_out);
    PGRC_ON_DEVICE(c);
    // Round 5: the text is packed to 2 bits per symbol ON THE HOST (pgrc_pack_ascii_host below: a few threads, AVX2 where the CPU
    // has it) and a quarter of the bytes cross the link -- 1.9 GB of ASCII took 35 ms of a C3-size job's 150 ms, and the link is that
    // job's bound.  Chunk k is copied while chunk k + 1 is packed.  PGRC_HOST_PACK=0: the bytes go up as they are and
    // a kernel packs them (rounds 1-4).
    if (c->opt.host_pack && count >= (1ull << 22)) {
        // T worker threads (started once per call) pack chunk k into pinned buffer k % 2, each its share of the chunk's words; this
        // thread copies a chunk when all have packed it and frees its buffer when the copy is through -- the workers pack chunk
        // k + 1 meanwhile.  The buffers are pinned (one pair per process: the copies are plain DMA, nothing is pinned on the fly).
        const uint64_t CH = 32ull << 20;                     // symbols per chunk (a multiple of 16)
        static std::mutex pin_mu;
        static uint32_t *pin[2] = {nullptr, nullptr};
        std::unique_lock<std::mutex> pin_lock(pin_mu);       // (one text at a time through the pair of buffers)
        for (int k = 0; k < 2; k++)
            if (!pin[k]) HIP_TRY(c, hipHostMalloc((void **)&pin[k], CH / 4, hipHostMallocDefault));
        const uint32_t T = c->opt.host_threads ? c->opt.host_threads : std::min<uint32_t>(8u, std::max(1u, std::thread::hardware_concurrency()));
        const uint64_t nchunks = (count + CH - 1) / CH;
        std::vector<std::atomic<uint32_t>> packed(nchunks);
        for (auto &x : packed) x.store(0);
        std::atomic<uint64_t> copied{0};                     // chunks whose copy is through
        std::atomic<bool> bad{false};
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                for (uint64_t k = 0; k < nchunks; k++) {
                    while (k >= 2 && copied.load(std::memory_order_acquire) < k - 1) std::this_thread::yield();     // buffer k % 2 is free
                    const uint64_t off = k * CH, len = std::min(CH, count - off), words = (len + 15) / 16;
                    const uint64_t per = (words + T - 1) / T, w0 = std::min(words, t * per), w1 = std::min(words, w0 + per);
                    if (w1 > w0 && !pgrc_pack_ascii_host((const uint8_t *)pg + off + w0 * 16, std::min(len, w1 * 16) - w0 * 16, pin[k & 1] + w0, 1)) bad = true;
                    packed[k].fetch_add(1, std::memory_order_release);
                }
            });
        hipError_t he = hipSuccess;
        for (uint64_t k = 0; k < nchunks; k++) {
            while (packed[k].load(std::memory_order_acquire) < T) std::this_thread::yield();
            const uint64_t off = k * CH, len = std::min(CH, count - off), words = (len + 15) / 16;
            if (he == hipSuccess) he = hipMemcpyAsync((uint32_t *)d_words_out + off / 16, pin[k & 1], words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            copied.store(k + 1, std::memory_order_release);  // (also after an error: the workers must run out)
        }
        for (std::thread &x : th) x.join();
        HIP_TRY(c, he);
        if (bad) {
            c->err = "pseudogenome contains a symbol outside ACGT";
            return PGRC_E_SYMBOL;
        }
        return PGRC_OK;
    }
    const uint64_t CH = 64ull << 20; // 64 Mi symbols per staging chunk (multiple of 16)
    DevBuf stage, flag;
    int e;
    if ((e = pgrc_buf_ensure(c, stage, (size_t)std::min(CH, count ? count : 16)))) return e;
    if ((e = pgrc_buf_ensure(c, flag, sizeof(uint32_t)))) { pgrc_buf_free(stage); return e; }
    (void)hipMemsetAsync(flag.p, 0, sizeof(uint32_t), c->stream);
    int rcode = PGRC_OK;
    for (uint64_t off = 0; off < count && rcode == PGRC_OK; off += CH) {
        const uint64_t len = std::min(CH, count - off);
        if (hipMemcpyAsync(stage.p, pg + off, len, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rcode = PGRC_E_DEVICE; break; }
        rcode = pgrc_launch_pack_ascii(c, (const uint8_t *)stage.p, len, (uint32_t *)d_words_out + off / 16, (uint32_t *)flag.p);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rcode = PGRC_E_DEVICE;
    }
    uint32_t bad = 0;
    if (rcode == PGRC_OK && hipMemcpy(&bad, flag.p, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) rcode = PGRC_E_DEVICE;
    pgrc_buf_free(stage);
    pgrc_buf_free(flag);
    if (rcode == PGRC_OK && bad) {
        c->err = "pseudogenome contains a symbol outside ACGT";
        return PGRC_E_SYMBOL;
    }
    if (rcode != PGRC_OK && c->err.empty()) c->err = "pack_pg_slice: HIP error";
    return rcode;
}

int pgrc_match_set_pg_ascii(pgrc_match_ctx *c, const char *pg, uint64_t G) {
    if (!c || !pg) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_set_pg_ascii(c, pg, G);
    PGRC_ON_DEVICE(c);
    int e = pgrc_pg_alloc(c, G);
    if (e) return e;
    if ((e = pgrc_match_pack_pg_slice(c, pg, G, c->pg2[0].p))) return e;
    c->have_pg = true;
    return PGRC_OK;
}

int pgrc_match_set_pg_packed_device(pgrc_match_ctx *c, const void *d_words, uint64_t G) {
    if (!c || !d_words) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_set_pg_packed_device(c, d_words, G);
    PGRC_ON_DEVICE(c);
    int e = pgrc_pg_alloc(c, G);
    if (e) return e;
    HIP_TRY(c, hipMemcpyAsync(c->pg2[0].p, d_words, c->pg_words * sizeof(uint32_t), hipMemcpyDefault, c->stream));   // (the source may live on another device)
    // clear the bits after symbol G-1 in the last word: padding must read as zero
    if (G % 16) {
        uint32_t last;
        HIP_TRY(c, hipMemcpyAsync(&last, (const uint32_t *)d_words + c->pg_words - 1, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        last &= (1u << (2 * (G % 16))) - 1u;
        HIP_TRY(c, hipMemcpyAsync((uint32_t *)c->pg2[0].p + c->pg_words - 1, &last, 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->have_pg = true;
    return PGRC_OK;
}

int pgrc_match_export_pg(pgrc_match_ctx *c, int strand, uint32_t *words) {
    if (!c || !words || strand < 0 || strand > 1) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_export_pg(c, strand, words);
    if (!c->have_pg) { c->err = "export_pg: no pseudogenome set"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    if (strand == 1 && !c->have_rc) {
        int e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G);
        if (e) return e;
        c->have_rc = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(words, c->pg2[strand].p, c->pg_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PGRC_OK;
}

// ------------------------------------------------------------------ reads

static int alloc_results(pgrc_match_ctx *c, uint64_t n) {
    int e;
    if ((e = pgrc_buf_ensure(c, c->d_pos, n * sizeof(uint64_t)))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_rc, n))) return e;
    if ((e = pgrc_buf_ensure(c, c->d_mism, n))) return e;
    c->have_results = false;
    return PGRC_OK;
}

static int begin_reads(pgrc_match_ctx *c, uint64_t n, bool own) {
    pgrc_stream_abort(c);                        // (a streamed run that was never finished)
    if (n >= (1ull << 32) - 1) { c->err = "reads count must stay below 2^32-1 (uint_reads_cnt_max, pg-config.h:21-22)"; return PGRC_E_PARAM; }
    if (n != c->n) c->screen_broken = false;     // another read set: the screen's per-read arrays may fit now
    c->n = n;
    c->n_nreads = 0;
    c->n_many = 0;
    c->h_nidx.clear();
    c->have_reads = false;
    int e;
    if (own) {
        c->stride = (n + 63) & ~63ull;
        if ((e = pgrc_buf_ensure(c, c->reads_own, (size_t)c->nw * std::max<uint64_t>(c->stride, 64) * sizeof(uint32_t)))) return e;
        c->reads2 = (const uint32_t *)c->reads_own.p;
    }
    return alloc_results(c, n);
}

// Chunked upload of ASCII rows: begin(n) -> append(rows, count)* -> end().  Lets a caller that can only
// produce reads one by one (ConstantLengthReadsSetInterface::getRead, ReadsSetInterface.h:41) stream them
// through a bounded host buffer.
int pgrc_match_begin_reads(pgrc_match_ctx *c, uint64_t n) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_begin_reads(c, n);
    PGRC_ON_DEVICE(c);
    int e = begin_reads(c, n, true);
    if (e) return e;
    if ((e = pgrc_buf_ensure(c, c->nread_flag, n))) return e;
    // (nread_npos, 4 bytes per read, only exists once a block that CAN hold an N arrives: ASCII rows or an ACGNT set -- an ACGT
    //  set cannot, and at C3 the array is 400 MB that a memory-tight job would rather give its second index set)
    // (cleared on a stream of its own and waited for: the main stream may hold index builds started ahead of the run --
    //  pgrc_match_prepare_index --, which neither this nor the uploads that follow should queue behind)
    if (!c->up_stream[0]) {
        hipError_t he = hipStreamCreateWithFlags(&c->up_stream[0], hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&c->up_ev[0], hipEventDisableTiming);
        if (he != hipSuccess) { c->up_stream[0] = nullptr; c->err = std::string("begin_reads: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
    }
    HIP_TRY(c, hipMemsetAsync(c->nread_flag.p, 0, n ? n : 1, c->up_stream[0]));
    HIP_TRY(c, hipStreamSynchronize(c->up_stream[0]));
    c->up_next = 0;
    c->up_chunk = 0;
    c->up_nidx.clear();
    c->up_nmany = 0;
    for (DevBuf &b : c->up_nchunks) pgrc_buf_free(b);
    c->up_nchunks.clear();
    c->up_nchunk_rows.clear();
    c->up_open = true;
    return PGRC_OK;
}

// One block of rows in any of the three host formats: `symbols` 0 = ASCII rows (read_len bytes), 4 = the reference's
// ACGT packing (4 symbols per byte), 5 = its ACGNT packing (3 symbols per byte).  Staged through a bounded device
// buffer and converted to the word-major 2-bit layout there; reads holding an N are flagged, and their ASCII rows
// (made on the device for the packed formats) are kept for the side list of end_reads.
#define UP_MARK(c, what) do { if ((c)->opt.stream_timing && (c)->st_on) fprintf(stderr, "pgrc stream: %8.2f ms    append: %s\n", (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - (c)->st_t0) * 1e3, what); } while (0)
static int append_rows(pgrc_match_ctx *c, const uint8_t *rows, uint64_t count, int32_t symbols) {
    if (!c->up_open || c->up_next + count > c->n) { c->err = "append_reads: outside begin/end or too many rows"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    const uint32_t L = c->prm.read_len;
    const uint32_t rb = symbols == 0 ? L : symbols == 4 ? (L + 3) / 4 : (L + 2) / 3;   // host bytes per row
    DevBuf &flag = c->up_flag, &lidx = c->up_lidx;
    auto cleanup = [&]() {};
    int e;
    // rows per staging chunk: ~256 MiB, a multiple of 1024 rows (a streamed run matches chunk by chunk: whole lines of the
    // word-major read array)
    // (a streamed run matches chunk by chunk, and every launch of the persistent match kernel pays ~4 ms of ramp and drain:
    //  C3 through the boundary 0.29 / 0.23 / 0.22 / 0.20 s with chunks of 128 / 256 / 512 / 1024 MiB, profiles/r03_boundary_c3.json)
    uint64_t chunk_mib = c->st_on ? 1024 : 256;
    if (c->opt.upload_chunk_mb) chunk_mib = c->opt.upload_chunk_mb;   // (experiments)
    const uint64_t CHR = std::max<uint64_t>(1024, ((chunk_mib << 20) / rb) & ~1023ull);
    for (int k = 0; k < 2; k++)
        if ((e = pgrc_buf_ensure(c, c->up_stage[k], (size_t)std::min(CHR, std::max<uint64_t>(count, 1)) * rb))) return e;
    if ((e = pgrc_buf_ensure(c, flag, sizeof(uint32_t)))) { cleanup(); return e; }
    // a streamed run (stream.hip) uploads and unpacks on streams of its own, beside the matching of the blocks before
    // (no call on the null stream and no hipFree in here: both wait for the main stream, where a streamed run's matching is queued)
    const bool streamed = c->st_on;
    hipStream_t main_stream = c->stream;
    hipStream_t ctl = streamed ? c->up_stream[0] : c->stream;
    if (hipMemsetAsync(flag.p, 0, sizeof(uint32_t), ctl) != hipSuccess || hipStreamSynchronize(ctl) != hipSuccess) { cleanup(); c->err = "append_reads: HIP error"; return PGRC_E_DEVICE; }
    int rcode = PGRC_OK;
    std::vector<uint8_t> nf, hrows;
    std::vector<uint32_t> local;
    for (uint64_t off = 0; off < count && rcode == PGRC_OK; off += CHR) {
        const uint64_t cnt = std::min(CHR, count - off);
        const uint64_t first = c->up_next + off;
        const int turn = (int)(c->up_chunk++ & 1u);
        hipStream_t up = streamed ? c->up_stream[turn] : c->stream;
        void *stage = c->up_stage[turn].p;
        if (hipMemcpyAsync(stage, rows + off * rb, cnt * rb, hipMemcpyHostToDevice, up) != hipSuccess) { rcode = PGRC_E_DEVICE; break; }
        c->stream = up;                                      // (the launchers below queue on c->stream)
        if (symbols == 0)
            rcode = pgrc_launch_pack_reads_ascii(c, (const uint8_t *)stage, first, cnt, L, (uint32_t *)c->reads_own.p, c->stride,
                                                 (uint8_t *)c->nread_flag.p, (uint32_t *)flag.p);
        else if (symbols == 4)
            rcode = pgrc_launch_repack_reads_ref(c, (const uint8_t *)stage, first, cnt, L, (uint32_t *)c->reads_own.p, c->stride);
        else
            rcode = pgrc_launch_unpack_reads_acgnt(c, (const uint8_t *)stage, first, cnt, L, (uint32_t *)c->reads_own.p, c->stride,
                                                   (uint8_t *)c->nread_flag.p, (uint32_t *)flag.p);
        // where the N's of the flagged reads are (flag 1 -> 3 for reads with at most 4 of them: the dual kernel's own)
        if (rcode == PGRC_OK && symbols != 4) rcode = pgrc_buf_ensure(c, c->nread_npos, std::max<uint64_t>(c->n, 1) * sizeof(uint32_t));   // (first such block: allocated)
        if (rcode == PGRC_OK && symbols != 4)
            rcode = pgrc_launch_npos_rows(c, (const uint8_t *)stage, symbols, first, cnt, L, (uint8_t *)c->nread_flag.p, (uint32_t *)c->nread_npos.p);
        c->stream = main_stream;
        // (the staging area is reused by the next chunk: its copy is queued behind this chunk's kernel on the same stream)
        if (symbols != 4) UP_MARK(c, "rows copied, unpack queued");
        if (!(streamed && symbols == 4) && hipStreamSynchronize(up) != hipSuccess) rcode = PGRC_E_DEVICE;
        if (symbols != 4) UP_MARK(c, "unpacked");
        if (rcode == PGRC_OK && symbols != 4) {               // (an ACGT set cannot hold an N)
            // reads with 'N' -> side list: keep their ASCII rows (they are a small minority)
            nf.resize(cnt);
            if (hipMemcpyAsync(nf.data(), (const uint8_t *)c->nread_flag.p + first, cnt, hipMemcpyDeviceToHost, up) != hipSuccess ||
                hipStreamSynchronize(up) != hipSuccess) { rcode = PGRC_E_DEVICE; break; }
            local.clear();
            hrows.clear();
            for (uint64_t k = 0; k < cnt; k++)
                if (nf[k]) {
                    c->up_nidx.push_back((uint32_t)(first + k));
                    if (nf[k] == 1) c->up_nmany++;
                    if (symbols == 0) hrows.insert(hrows.end(), rows + (off + k) * rb, rows + (off + k + 1) * rb);
                    else local.push_back((uint32_t)k);
                }
            const size_t nn = symbols == 0 ? hrows.size() / L : local.size();
            UP_MARK(c, "N flags on the host, list made");
            if (nn) {      // their ASCII rows stay in HBM: uploaded (ASCII input) or made there (packed input)
                DevBuf nrows;
                if ((e = pgrc_buf_ensure(c, nrows, nn * L))) { rcode = e; break; }
                UP_MARK(c, "room for the N rows");
                c->up_nchunks.push_back(nrows);
                c->up_nchunk_rows.push_back(nn);
                if (symbols == 0) {
                    if (hipMemcpyAsync(nrows.p, hrows.data(), nn * L, hipMemcpyHostToDevice, up) != hipSuccess || hipStreamSynchronize(up) != hipSuccess) rcode = PGRC_E_DEVICE;
                } else {
                    if ((e = pgrc_buf_ensure(c, lidx, nn * sizeof(uint32_t)))) { rcode = e; break; }
                    if (hipMemcpyAsync(lidx.p, local.data(), nn * sizeof(uint32_t), hipMemcpyHostToDevice, up) != hipSuccess) { rcode = PGRC_E_DEVICE; break; }
                    c->stream = up;
                    rcode = pgrc_launch_nrows_ascii_acgnt(c, (const uint8_t *)stage, (const uint32_t *)lidx.p, nn, L, (uint8_t *)nrows.p);
                    c->stream = main_stream;
                    if (hipStreamSynchronize(up) != hipSuccess) rcode = PGRC_E_DEVICE;    // (stage is reused by the next block)
                    UP_MARK(c, "N rows as ASCII on the device");
                }
            }
        }
        if (rcode == PGRC_OK && streamed) rcode = pgrc_stream_block_arrived(c, first, cnt, symbols != 4, turn);
    }
    uint32_t bad = 0;
    if (rcode == PGRC_OK && streamed && (hipStreamSynchronize(c->up_stream[0]) != hipSuccess || hipStreamSynchronize(c->up_stream[1]) != hipSuccess)) rcode = PGRC_E_DEVICE;
    if (rcode == PGRC_OK && !streamed && hipStreamSynchronize(c->stream) != hipSuccess) rcode = PGRC_E_DEVICE;
    if (rcode == PGRC_OK && (hipMemcpyAsync(&bad, flag.p, sizeof bad, hipMemcpyDeviceToHost, ctl) != hipSuccess || hipStreamSynchronize(ctl) != hipSuccess)) rcode = PGRC_E_DEVICE;
    cleanup();
    if (rcode != PGRC_OK) { if (c->err.empty()) c->err = "append_reads: HIP error"; pgrc_stream_abort(c); return rcode; }
    if (bad) { c->err = symbols == 5 ? "packed reads hold a byte outside the ACGNT code range" : "reads contain a symbol outside ACGNT"; pgrc_stream_abort(c); return PGRC_E_SYMBOL; }
    c->up_next += count;
    return PGRC_OK;
}

int pgrc_match_append_reads_ascii(pgrc_match_ctx *c, const char *reads, uint64_t count) {
    if (!c || (!reads && count)) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_append_reads(c, reads, count, 0);
    return append_rows(c, (const uint8_t *)reads, count, 0);
}

int pgrc_match_append_reads_packed(pgrc_match_ctx *c, const uint8_t *packed, uint64_t count, int32_t symbols) {
    if (!c || (!packed && count) || (symbols != 4 && symbols != 5)) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_append_reads(c, packed, count, symbols);
    return append_rows(c, packed, count, symbols);
}

int pgrc_match_end_reads(pgrc_match_ctx *c) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_end_reads(c);
    if (!c->up_open || c->up_next != c->n) { c->err = "end_reads: fewer rows appended than announced"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    int e;
    c->n_nreads = c->up_nidx.size();
    c->n_many = c->up_nmany;
    c->h_nidx = c->up_nidx;
    if (c->n_nreads) {
        const uint32_t L = c->prm.read_len;
        if ((e = pgrc_buf_ensure(c, c->nread_idx, c->up_nidx.size() * sizeof(uint32_t)))) return e;
        if ((e = pgrc_buf_ensure(c, c->nread_ascii, c->n_nreads * L))) return e;
        // (a streamed run has its matching queued on the main stream: the side list is put together beside it)
        hipStream_t s = (c->st_on && c->up_stream[0]) ? c->up_stream[0] : c->stream;
        HIP_TRY(c, hipMemcpyAsync(c->nread_idx.p, c->up_nidx.data(), c->up_nidx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        uint64_t at = 0;
        for (size_t k = 0; k < c->up_nchunks.size(); k++) {
            HIP_TRY(c, hipMemcpyAsync((uint8_t *)c->nread_ascii.p + at * L, c->up_nchunks[k].p, c->up_nchunk_rows[k] * L, hipMemcpyDeviceToDevice, s));
            at += c->up_nchunk_rows[k];
        }
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    // (chunks below the pool's size go back through hipFree, which waits for the device: a streamed run keeps them until its end)
    if (c->st_on) {
        for (DevBuf &b : c->up_nchunks) c->st_keep.push_back(b);
    } else
    for (DevBuf &b : c->up_nchunks) pgrc_buf_free(b);
    c->up_nchunks.clear();
    c->up_nchunk_rows.clear();
    c->up_nidx.clear();
    c->up_nidx.shrink_to_fit();
    c->up_open = false;
    c->have_reads = true;
    return PGRC_OK;
}

int pgrc_match_set_reads_ascii(pgrc_match_ctx *c, const char *reads, uint64_t n) {
    if (!c || (!reads && n)) return PGRC_E_PARAM;
    int e = pgrc_match_begin_reads(c, n);
    if (e) return e;
    if ((e = pgrc_match_append_reads_ascii(c, reads, n))) return e;
    return pgrc_match_end_reads(c);
}

int pgrc_match_set_reads_packed(pgrc_match_ctx *c, const uint8_t *packed, uint64_t n) {
    if (!c || (!packed && n)) return PGRC_E_PARAM;
    int e = pgrc_match_begin_reads(c, n);
    if (e) return e;
    if ((e = pgrc_match_append_reads_packed(c, packed, n, 4))) return e;
    return pgrc_match_end_reads(c);
}

int pgrc_match_set_reads_device(pgrc_match_ctx *c, const void *d_words, uint64_t n, uint64_t stride) {
    if (!c || (!d_words && n) || stride < n) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_set_reads_device(c, d_words, n, stride);
    PGRC_ON_DEVICE(c);
    int e = begin_reads(c, n, false);
    if (e) return e;
    c->reads2 = (const uint32_t *)d_words;
    c->stride = stride;
    c->have_reads = true;
    return PGRC_OK;
}

// ------------------------------------------------------------------ matching

int pgrc_match_init_results(pgrc_match_ctx *c) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_init_results(c);
    if (!c->have_reads) { c->err = "init_results: no reads set"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    int e = pgrc_launch_init_results(c);
    if (e) return e;
    c->have_results = true;
    return PGRC_OK;
}

int pgrc_match_set_results(pgrc_match_ctx *c, const uint64_t *pos, const uint8_t *rc, const uint8_t *mism) {
    if (!c || !pos || !rc || !mism) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_set_results(c, pos, rc, mism);
    if (!c->have_reads) { c->err = "set_results: no reads set"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    if (c->n) {
        HIP_TRY(c, hipMemcpyAsync(c->d_pos.p, pos, c->n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_rc.p, rc, c->n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_mism.p, mism, c->n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->have_results = true;
    return PGRC_OK;
}

// the two sets of index buffers of the screened schedule swap roles
static void swap_index_sets(pgrc_match_ctx *c) {
    std::swap(c->d_head, c->alt_head);
    std::swap(c->head_ptr, c->alt_head_ptr);
    std::swap(c->head_sh, c->alt_head_sh);
    for (int k = 0; k < 2; k++) {
        std::swap(c->d_skey[k], c->alt_skey[k]);
        std::swap(c->d_sval[k], c->alt_sval[k]);
    }
    std::swap(c->d_sorttmp, c->alt_sorttmp);
    std::swap(c->ent_ptr, c->alt_ent_ptr);
    std::swap(c->index_strand, c->alt_index_strand);
}

} // extern "C"
void pgrc_swap_index_sets(pgrc_match_ctx *c) { swap_index_sets(c); }
extern "C" {

// Two-pass runs of mode c with min_mismatches == 0 take the screened schedule (copmem.hip, "Exact-match screen") unless
// PGRC_SCREEN=0 or the second set of index buffers does not fit.
static bool screen_wanted(const pgrc_match_ctx *c, int first, int last) {
    if (c->prm.mode != 'c' || first != 0 || last != 1 || c->prm.min_mismatches != 0 || !c->n || c->screen_broken) return false;
    if (c->opt.screen >= 0) return c->opt.screen == 1;     // PGRC_SCREEN: 0 never, 1 whenever it applies; unset: where it pays
    if (!c->opt.early_stop) return false;       // (the screen's proofs of absence ARE the early-stop rule)
    // The screen is one more sweep over all reads (~5 probes each) and saves a read that matches the other strand exactly
    // the forward query it would lose: with fewer than ~48 seeds per read (L = 100: 37) the two about cancel
    // (profiles/r02_screen_ab.txt: C2 +2.6 %, C3 -12 %).
    const uint32_t K = (uint32_t)c->cp.K, k2 = (uint32_t)c->cp.k2;
    return k2 && c->prm.read_len >= K && (c->prm.read_len - K) / k2 + 1 >= 48u;
}

// ... and, when they apply, rather ONE query per read over both strands (copmem.hip, "The dual kernel"); PGRC_DUAL=0: never
static bool dual_wanted(const pgrc_match_ctx *c, int first, int last) {
    if (c->prm.mode != 'c' || first != 0 || last != 1 || c->prm.min_mismatches != 0 || !c->n || c->screen_broken) return false;
    if (c->opt.dual == 0) return false;
    if (!c->opt.early_stop || c->opt.screen >= 0) return false;   // the older schedules were asked for
    if (c->opt.dual == 1) return true;
    // Round 2 kept short reads (fewer than 48 seeds) on the two passes: the dual kernel cost C2 (100 bp, 37 seeds) +19 % then.
    // With the pair table and nothing launched behind it, it wins at every length measured (round 4, one context each,
    // profiles/r04_c2_dual_ab.txt): C2 16.3 -> 12.5 ms, 75 bp 13.4 -> 11.0, 50 bp 9.9 -> 8.6, 1 M x 150 bp 7.2 -> 4.6.
    const uint32_t K = (uint32_t)c->cp.K, k2 = (uint32_t)c->cp.k2;
    return k2 && c->prm.read_len >= K;
}

} // extern "C" (a template, and helpers of stream.hip)
// The RC text and both strands' indexes: strand 0 ends up in the ALTERNATE set of index buffers, strand 1 in the active
// one (what the dual kernel and the screened schedule expect).  Both builds at once on two streams (they share nothing
// but the bandwidth; PGRC_BUILD_STREAMS=1: in turn); the main stream is made to wait for the second.  `mark` is called
// after the RC text and after the builds (profiling events).  PGRC_E_ALLOC: no room for the second set -- the caller frees
// what it got.
template <typename F>
static int pgrc_build_both_indexes(pgrc_match_ctx *c, bool *two_streams, F mark) {
    int e;
    if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) return e;
    c->have_rc = true;
    mark(); // 1
    bool two = !c->opt.builds_in_turn;
    if (two && !c->build_stream) {
        hipError_t he = hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking);
        for (int k = 0; k < 2 && he == hipSuccess; k++) he = hipEventCreateWithFlags(&c->build_ev[k], hipEventDisableTiming);
        if (he != hipSuccess) { (void)hipGetLastError(); c->build_stream = nullptr; two = false; }
    }
    hipStream_t main_stream = c->stream;
    if (two) {
        HIP_TRY(c, hipEventRecord(c->build_ev[0], main_stream));             // the RC text is ready
        HIP_TRY(c, hipStreamWaitEvent(c->build_stream, c->build_ev[0], 0));
    }
    // Both strands' heads go into ONE table of 32-byte slots {forward head, RC head} per bucket number (round 4): the dual
    // kernel's two gathers of a seed then hit one 64-byte line and one translation.  PGRC_HEAD_PAIR=0: a table per strand
    // (A/B runs); no room for the pair table: the same.
    {
        bool pair = c->opt.head_pair != 0;
        if (pair) {
            const int pe = pgrc_buf_ensure(c, c->d_headpair, (size_t)c->cp.hash_size * 4 * sizeof(uint64_t));
            if (pe == PGRC_E_ALLOC) { (void)hipGetLastError(); pair = false; }
            else if (pe) return pe;
        }
        c->pair_build = pair;
        // buckets per group (headfmt.h; PGRC_HEAD_PAIR=1..4: 1, 2, 4, 8): 4 -- measured at C3 in one context
        // (profiles/r04_head_interleave_ab.txt): both gathers of a seed cost one request as long as they share a 128-BYTE line
        // (groups of 1, 2, 4: dual kernel 79.5 -> 64.7 ms; groups of 8 = 256 bytes: 79.6), and the builds' head stores cost
        // nothing extra only as whole 64-byte runs (groups of 1 and 2: index pair 16.2 -> 24.9 ms; groups of 4: 17.0)
        c->pair_gm = pair ? c->opt.head_pair - 1u : 3u;
    }
    if ((e = pgrc_copmem_build_index(c, 0))) { c->pair_build = false; return e; }
    if (!two) mark(); // 2
    swap_index_sets(c);
    if (two) c->stream = c->build_stream;
    // (PGRC_TEST_NO_SECOND_INDEX: tests take the out-of-memory road without exhausting the device)
    e = c->opt.test_no_second_index ? PGRC_E_ALLOC : pgrc_copmem_build_index(c, 1);
    c->pair_build = false;
    c->stream = main_stream;
    if (two) {
        if (hipEventRecord(c->build_ev[1], c->build_stream) != hipSuccess || hipStreamWaitEvent(main_stream, c->build_ev[1], 0) != hipSuccess) {
            c->err = "index build streams";
            return PGRC_E_DEVICE;
        }
        mark(); // 2: both indexes (ms_index[0] is then the pair, ms_index[1] ~ 0)
    }
    *two_streams = two;
    return e;
}

bool pgrc_dual_applies(const pgrc_match_ctx *c) {
    pgrc_match_ctx probe;                     // (dual_wanted looks at n only to rule out an empty set)
    probe.prm = c->prm;
    probe.opt = c->opt;
    probe.cp = c->cp;
    probe.n = 1;
    probe.screen_broken = c->screen_broken;
    return dual_wanted(&probe, 0, 1);
}

// pgrc_match_prepare_index / pgrc_match_stream_begin (stream.hip): both strands' indexes now, on their streams
int pgrc_prepare_both_indexes(pgrc_match_ctx *c) {
    bool two = false;
    const int e = pgrc_build_both_indexes(c, &two, []() {});
    if (e == PGRC_E_ALLOC) {
        // no room for the second set: give back what it got; later runs take the two passes with one set
        (void)hipGetLastError();
        if (two) (void)hipStreamSynchronize(c->build_stream);
        DevBuf *part[] = {&c->d_head, &c->d_skey[0], &c->d_skey[1], &c->d_sval[0], &c->d_sval[1], &c->d_sorttmp};
        for (DevBuf *b : part) pgrc_buf_free(*b);
        c->ent_ptr = nullptr;
        c->index_strand = -1;
        swap_index_sets(c);
        if (c->head_sh) {                   // (the forward index that was built sits in the pair table: nothing keeps it)
            (void)hipStreamSynchronize(c->stream);
            pgrc_buf_free(c->d_headpair);
            c->head_ptr = c->alt_head_ptr = nullptr;
            c->head_sh = c->alt_head_sh = 0;
            c->ent_ptr = nullptr;
            c->index_strand = -1;
        }
        c->screen_broken = true;
        c->err = "prepare_index: no room for both strands' indexes";
    }
    c->idx_prepared = e == PGRC_OK;
    return e;
}

extern "C" {
static int run_passes(pgrc_match_ctx *c, int first, int last) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_run(c, first, last);
    if (!c->have_pg || !c->have_reads) { c->err = "run: set the pseudogenome and the reads first"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    int e;
    if (!c->have_results && (e = pgrc_match_init_results(c))) return e;
    HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, 32 * sizeof(uint64_t), c->stream));
    memset(&c->ctr, 0, sizeof c->ctr);
    const bool prof = c->profiling && c->have_events;
    int evi = 0;
    auto mark = [&]() { if (prof) (void)hipEventRecord(c->ev[evi], c->stream); evi++; };
    mark(); // 0
    bool fallback_layout = false;
    const bool dual = dual_wanted(c, first, last);
    bool screened = dual || screen_wanted(c, first, last);
    if (screened && ((e = pgrc_buf_ensure(c, c->d_scr_pos, c->n * sizeof(uint64_t))) || (e = pgrc_buf_ensure(c, c->d_scr_flag, c->n)))) {
        if (e != PGRC_E_ALLOC) return e;
        (void)hipGetLastError();
        c->screen_broken = true;
        screened = false;
    }
    if (screened) {
        bool two = false;
        if (c->idx_prepared && c->index_strand == 1 && c->alt_index_strand == 0 && c->have_rc) {
            // both indexes were built ahead of the run (pgrc_match_prepare_index): the builds may still be in flight on
            // their streams; this run's work is ordered behind them
            c->idx_prepared = false;
            mark(); // 1
            if (c->build_stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->build_ev[1], 0));
            mark(); // 2
            e = PGRC_OK;
            two = true;
        } else {
            e = pgrc_build_both_indexes(c, &two, [&]() { mark(); });
        }
        if (e == PGRC_E_ALLOC) {
            // no room for both indexes: free what the second set got, and run the passes in the reference's order
            (void)hipGetLastError();
            if (two) (void)hipStreamSynchronize(c->build_stream);   // (nothing of a half-started build may still be running)
            DevBuf *part[] = {&c->d_head, &c->d_skey[0], &c->d_skey[1], &c->d_sval[0], &c->d_sval[1], &c->d_sorttmp};
            for (DevBuf *b : part) pgrc_buf_free(*b);
            c->ent_ptr = nullptr;
            c->index_strand = -1;
            swap_index_sets(c);
            c->screen_broken = true;
            screened = false;
            if (c->index_strand != 0 || !c->ent_ptr) {
                // (it was already the FORWARD index that found no room, e.g. for the pair table's share of it: once more on its
                //  own, into a table of its own -- if that does not fit either, the run fails with PGRC_E_ALLOC)
                if (c->head_sh) { (void)hipStreamSynchronize(c->stream); pgrc_buf_free(c->d_headpair); c->head_ptr = c->alt_head_ptr = nullptr; c->head_sh = c->alt_head_sh = 0; }
                if ((e = pgrc_copmem_build_index(c, 0))) return e;
            }
            mark(); // 3
            if ((e = pgrc_copmem_match_pass(c, 0))) return e;
            mark(); // 4
            c->pair_build = c->head_sh != 0;      // (the forward index sits in the pair table: the RC heads take its other half)
            e = pgrc_copmem_build_index(c, 1);
            c->pair_build = false;
            if (e) return e;
            mark(); // 5
            if ((e = pgrc_copmem_match_pass(c, 1))) return e;
            mark(); // 6
            fallback_layout = true;  // (events: see the timing below)
        } else {
            if (e) return e;
            mark(); // 3
            HIP_TRY(c, hipMemsetAsync(c->d_scr_flag.p, 0, c->n, c->stream));
            if (dual) {
                // the reads with N: those with at most 4 N's are the dual kernel's own (its N-aware hash and verification);
                // the others go the byte path on the side stream, a read's forward query and then its RC query in one lane,
                // behind the dual kernel.  (Launched first, as a small persistent grid, that kernel still only starts when the
                //  dual kernel's persistent blocks leave -- profiles/r04_nread_beside_ab.txt)
                if ((e = pgrc_copmem_match_dual(c))) return e;             // one query per read over both strands
                mark(); // 4
                if ((e = pgrc_copmem_match_nreads(c, 0, 1, true))) return e;
                if ((e = pgrc_copmem_join_nreads(c))) return e;
                mark(); // 5: what the reads with N took beyond the dual kernel
                mark(); // 6
            } else {
                if ((e = pgrc_copmem_match_phase(c, 1, 1))) return e;      // screen on the RC text
                mark(); // 4
                swap_index_sets(c);
                if ((e = pgrc_copmem_match_phase(c, 0, 2))) return e;      // forward pass honouring the flags
                mark(); // 5
                swap_index_sets(c);
                if ((e = pgrc_copmem_match_phase(c, 1, 0))) return e;      // RC pass over what is left
                mark(); // 6
            }
        }
    } else if (c->prm.mode == 'c') {
        for (int pass = first; pass <= last; pass++) {
            if (pass == 1) {
                // PgHelpers::reverseComplementInPlace(pgPtr), ReadsMatchers.cpp:168 -- rebuilt every run like the reference
                if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) return e;
                c->have_rc = true;
            }
            mark(); // 1 / 4
            if ((e = pgrc_copmem_build_index(c, pass))) return e;
            mark(); // 2 / 5
            if ((e = pgrc_copmem_match_pass(c, pass))) return e;
            mark(); // 3 / 6
        }
    } else {
        if ((e = pgrc_seedidx_run(c, first, last))) return e;
    }
    if ((e = pgrc_launch_hist(c))) return e; // synchronises the stream
    mark();
    uint64_t ctr[16];
    HIP_TRY(c, hipMemcpy(ctr, c->d_counters.p, sizeof ctr, hipMemcpyDeviceToHost));
    uint64_t scr[8];
    HIP_TRY(c, hipMemcpy(scr, (const uint64_t *)c->d_counters.p + 24, sizeof scr, hipMemcpyDeviceToHost));
    for (int s = 0; s < 2 && c->prm.mode == 'c'; s++) {      // (modes d/i/e: pgrc_seedidx_run filled searched / candidates)
        c->ctr.searched[s] = ctr[8 * s + 0];
        c->ctr.candidates[s] = ctr[8 * s + 1];
        c->ctr.probes[s] = ctr[8 * s + 2];
        c->ctr.entry_fetches[s] = ctr[8 * s + 3];
        c->ctr.verifies[s] = ctr[8 * s + 4];
    }
    if (screened && dual) {
        for (int k = 0; k < 5; k++) c->ctr.dual[k] = scr[k];
        c->ctr.redo_reads = scr[5];
        c->ctr.dual_seed_probes = scr[6];
        c->ctr.screened = 2;
    } else if (screened) {           // the screen ran on the RC text: its work counts with that strand's (not "searched")
        c->ctr.candidates[1] += scr[1];
        c->ctr.probes[1] += scr[2];
        c->ctr.entry_fetches[1] += scr[3];
        c->ctr.verifies[1] += scr[4];
        c->ctr.screened = 1;
    }
    c->ctr.schedule_downgraded = c->screen_broken ? 1u : 0u;
    if (prof) {
        HIP_TRY(c, hipEventSynchronize(c->ev[evi - 1]));
        float ms = 0;
        if (c->prm.mode == 'c' && (screened || fallback_layout)) {
            // events: 1 start of the forward index, 2 its end; then (screened) RC index, screen, forward match, RC match
            //         or (fallback) forward match, RC index, RC match
            auto span = [&](int x, int y) { float t = 0; (void)hipEventElapsedTime(&t, c->ev[x], c->ev[y]); return t; };
            c->ctr.ms_index[0] = span(1, 2);
            if (screened) {
                c->ctr.ms_index[1] = span(2, 3);
                c->ctr.ms_screen = span(3, 4);
                c->ctr.ms_match[0] = span(4, 5);
                c->ctr.ms_match[1] = span(5, 6);
            } else {
                c->ctr.ms_match[0] = span(3, 4);
                c->ctr.ms_index[1] = span(4, 5);
                c->ctr.ms_match[1] = span(5, 6);
            }
        } else if (c->prm.mode == 'c') {
            int b = 1;
            for (int pass = first; pass <= last; pass++, b += 3) {
                (void)hipEventElapsedTime(&ms, c->ev[b], c->ev[b + 1]);
                c->ctr.ms_index[pass] = ms;
                (void)hipEventElapsedTime(&ms, c->ev[b + 1], c->ev[b + 2]);
                c->ctr.ms_match[pass] = ms;
            }
        }
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[evi - 1]);
        c->ctr.ms_total = ms;
        c->ctr.ms_other = ms - c->ctr.ms_index[0] - c->ctr.ms_index[1] - c->ctr.ms_match[0] - c->ctr.ms_match[1] - c->ctr.ms_screen;
    }
    return PGRC_OK;
}

int pgrc_match_run(pgrc_match_ctx *c, int rev_compl_pg) { return run_passes(c, 0, rev_compl_pg ? 1 : 0); }

// One executeMatching(revCompMode) (ReadsMatchers.h:46): strand 0 = the text as given, 1 = its reverse complement.
int pgrc_match_run_pass(pgrc_match_ctx *c, int strand) {
    if (strand < 0 || strand > 1) return PGRC_E_PARAM;
    return run_passes(c, strand, strand);
}

int pgrc_match_get_results(pgrc_match_ctx *c, uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t hist[256], uint64_t *matched) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_get_results(c, pos, rc, mism, hist, matched);
    if (!c->have_results) { c->err = "get_results: nothing computed"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->n) {
        if (pos) HIP_TRY(c, hipMemcpy(pos, c->d_pos.p, c->n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (rc) HIP_TRY(c, hipMemcpy(rc, c->d_rc.p, c->n, hipMemcpyDeviceToHost));
        if (mism) HIP_TRY(c, hipMemcpy(mism, c->d_mism.p, c->n, hipMemcpyDeviceToHost));
    }
    if (hist) memcpy(hist, c->hist, sizeof c->hist);
    if (matched) *matched = c->matched;
    return PGRC_OK;
}

int pgrc_match_get_results_device(pgrc_match_ctx *c, void **d_pos, void **d_rc, void **d_mism) {
    if (!c) return PGRC_E_PARAM;
    if (c->multi) { c->err = "get_results_device: the results of a multi-device context live on several devices"; return PGRC_E_PARAM; }
    if (!c->have_reads) { c->err = "get_results_device: no reads set"; return PGRC_E_STATE; }
    if (d_pos) *d_pos = c->d_pos.p;
    if (d_rc) *d_rc = c->d_rc.p;
    if (d_mism) *d_mism = c->d_mism.p;
    return PGRC_OK;
}

int pgrc_match_get_counters(pgrc_match_ctx *c, pgrc_match_counters *out) {
    if (!c || !out) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_get_counters(c, out);
    *out = c->ctr;
    out->index_entries[0] = out->index_entries[1] = c->npos;
    return PGRC_OK;
}

// the same for a caller built against an older (shorter) or newer (longer) pgrc_match_counters: the struct only ever grows at
// its end, so the first min(size, sizeof) bytes are what that caller knows; what it has beyond is zeroed
int pgrc_match_get_counters_sized(pgrc_match_ctx *c, void *out, size_t out_size) {
    if (!c || !out) return PGRC_E_PARAM;
    pgrc_match_counters full;
    const int e = pgrc_match_get_counters(c, &full);
    if (e) return e;
    memset(out, 0, out_size);
    memcpy(out, &full, std::min(out_size, sizeof full));
    return PGRC_OK;
}

// introspection: which reads the dual kernel of the last run did again in the reference's order (flags[i] != 0)
int pgrc_match_get_redo_flags(pgrc_match_ctx *c, uint8_t *flags) {
    if (!c || !flags) return PGRC_E_PARAM;
    if (c->multi) {
        for (const PgrcShardView &sv : pgrc_multi_shards(c)) {
            int e = pgrc_match_get_redo_flags(sv.ctx, flags + sv.lo);
            if (e) { c->err = sv.ctx->err; return e; }
        }
        return PGRC_OK;
    }
    if (!c->have_results) { c->err = "get_redo_flags: run first"; return PGRC_E_STATE; }
    if (c->ctr.screened != 2) { memset(flags, 0, c->n); return PGRC_OK; }     // no dual kernel in the last run: nothing was redone
    PGRC_ON_DEVICE(c);
    HIP_TRY(c, hipMemcpy(flags, c->d_scr_flag.p, c->n, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < c->n; i++) flags[i] = flags[i] >> 1;
    return PGRC_OK;
}

int pgrc_match_extract_mismatches(pgrc_match_ctx *c, const uint8_t *reversed_flags, uint64_t *cum, uint8_t *codes,
                                  uint16_t *offsets) {
    if (!c || !cum) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_extract_mismatches(c, reversed_flags, cum, codes, offsets);
    if (!c->have_results || !c->have_pg) { c->err = "extract_mismatches: run first"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    return pgrc_extract_mismatches(c, reversed_flags, cum, codes, offsets);
}

int pgrc_match_export_index(pgrc_match_ctx *c, int strand, uint32_t *cumm, uint32_t *positions, uint64_t *count) {
    if (!c || strand < 0 || strand > 1) return PGRC_E_PARAM;
    if (c->multi) return pgrc_multi_export_index(c, strand, cumm, positions, count);
    if (!c->have_pg || c->prm.mode != 'c') { c->err = "export_index: mode c with a pseudogenome only"; return PGRC_E_STATE; }
    PGRC_ON_DEVICE(c);
    int e;
    if (strand == 1 && !c->have_rc) {
        if ((e = pgrc_launch_revcomp(c, (const uint32_t *)c->pg2[0].p, (uint32_t *)c->pg2[1].p, c->G))) return e;
        c->have_rc = true;
    }
    if (c->index_strand != strand && (e = pgrc_copmem_build_index(c, strand))) return e;
    return pgrc_copmem_export_index(c, cumm, positions, count);
}

} // extern "C"
