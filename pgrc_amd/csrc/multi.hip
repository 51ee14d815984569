// multi.hip -- several GPUs behind ONE matcher object, in one process (pgrc_match_create_multi, include/pgrc_match.h).
//
// The reference is a single process holding a single matcher (pgrc-encoder.cpp:342-374) whose per-read loop is an
// `omp parallel for` (ReadsMatchers.cpp:426-428): reads are independent units.  Here that loop shards over devices:
//   * contiguous, even-aligned read ranges (PE mates 2q, 2q+1 stay together: ReadsMatchers.cpp:553 pairs them by
//     orgIdx parity) -- one child context per device, one host thread per child while a call is in flight;
//   * the text is replicated: every device packs 1/n of the host ASCII text into ITS slice of its own text buffer and
//     ONE in-place all-gather (RCCL, ncclCommInitAll communicators of this process; C3: 58.6 MB per rank over xGMI)
//     completes all copies; every device then builds the seed index locally (no further exchange);
//   * results are written straight into the caller's arrays at the shard offsets, histograms are summed.
// A device listed twice (rehearsing the sharded path on a smaller box) cannot join a RCCL communicator twice; those
// contexts exchange their slices with peer / device-to-device copies instead -- the same data movement.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded on first use, never linked

#include "ctx.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

bool rccl_load(RcclApi &r, std::string &err) {
    if (r.lib) return true;
    // RTLD_NOLOAD first: a process that already carries a RCCL (PyTorch bundles one under the same SONAME) keeps it
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char *n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) { err = std::string("RCCL not found: ") + dlerror(); return false; }
#define RCCL_SYM(field, name)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                   \
    if (!r.field) { err = std::string("RCCL symbol missing: ") + name; dlclose(h); return false; }
    RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(AllGather, "ncclAllGather")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    r.lib = h;
    return true;
}

} // namespace

struct pgrc_multi {
    std::vector<pgrc_match_ctx *> child;
    std::vector<int> dev;
    std::vector<uint64_t> lo, hi;      // read range of child r (valid after begin_reads)
    bool distinct = true;              // no device listed twice
    uint64_t up_next = 0;              // streamed upload: rows seen so far
    bool up_open = false;
    RcclApi rccl;
    std::vector<ncclComm_t> comm;      // one communicator per child (created at the first all-gather)
    float ms_allgather = 0;
};

// contiguous read range of shard r; boundaries are even (same arithmetic as pgrc_amd/dist.py:shard_range)
static void shard_range(uint64_t n, size_t r, size_t world, uint64_t *lo, uint64_t *hi) {
    uint64_t per = (n + world - 1) / world;
    per = (per + 1) & ~1ull;
    *lo = std::min<uint64_t>(n, (uint64_t)r * per);
    *hi = std::min<uint64_t>(n, *lo + per);
}

// fn(r) on every child, one host thread per device while the call is in flight
template <class F>
static int on_children(pgrc_match_ctx *f, F fn) {
    pgrc_multi *m = f->multi;
    const size_t k = m->child.size();
    std::vector<int> rc(k, PGRC_OK);
    if (k == 1) rc[0] = fn((size_t)0);
    else {
        std::vector<std::thread> th;
        th.reserve(k);
        for (size_t r = 0; r < k; r++) th.emplace_back([&rc, &fn, r]() { rc[r] = fn(r); });
        for (auto &t : th) t.join();
    }
    for (size_t r = 0; r < k; r++)
        if (rc[r]) {
            f->err = "shard " + std::to_string(r) + " (device " + std::to_string(m->dev[r]) + "): " + m->child[r]->err;
            return rc[r];
        }
    return PGRC_OK;
}

extern "C" {

int pgrc_match_device_count(int32_t *count) {
    if (!count) return PGRC_E_PARAM;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return PGRC_E_NO_DEVICE; }
    *count = n;
    return n > 0 ? PGRC_OK : PGRC_E_NO_DEVICE;
}

int32_t pgrc_match_shard_count(const pgrc_match_ctx *c) {
    if (!c) return 0;
    return c->multi ? (int32_t)c->multi->child.size() : 1;
}

int pgrc_match_shard_info(const pgrc_match_ctx *c, int32_t shard, int32_t *device, uint64_t *first_read, uint64_t *n_reads) {
    if (!c || shard < 0 || shard >= pgrc_match_shard_count(c)) return PGRC_E_PARAM;
    if (!c->multi) {
        if (device) *device = c->device;
        if (first_read) *first_read = 0;
        if (n_reads) *n_reads = c->n;
        return PGRC_OK;
    }
    const pgrc_multi *m = c->multi;
    if (device) *device = m->dev[shard];
    if (first_read) *first_read = m->lo.empty() ? 0 : m->lo[shard];
    if (n_reads) *n_reads = m->lo.empty() ? 0 : m->hi[shard] - m->lo[shard];
    return PGRC_OK;
}

int pgrc_match_create_multi(const pgrc_match_params *p, int32_t n_devices, const int32_t *devices, pgrc_match_ctx **out) {
    if (!p || !out || !devices || n_devices < 1 || n_devices > 32) return PGRC_E_PARAM;
    *out = nullptr;
    pgrc_match_ctx *f = new pgrc_match_ctx();
    pgrc_multi *m = new pgrc_multi();
    f->multi = m;
    f->prm = *p;
    f->opt = pgrc_options_from_env();
    f->device = devices[0];
    f->nw = (p->read_len + 15) / 16;
    int e = PGRC_OK;
    for (int32_t r = 0; r < n_devices && !e; r++) {
        for (int32_t q = 0; q < r; q++)
            if (devices[q] == devices[r]) m->distinct = false;
        pgrc_match_params cp = *p;
        cp.device = devices[r];
        if (cp.device < 0) { e = PGRC_E_PARAM; break; }
        pgrc_match_ctx *c = nullptr;
        e = pgrc_match_create(&cp, &c);       // (a failure leaves its message with pgrc_match_last_error(NULL))
        if (!e) {
            m->child.push_back(c);
            m->dev.push_back(devices[r]);
        }
    }
    if (e) {
        pgrc_multi_destroy(f);
        return e;
    }
    *out = f;
    return PGRC_OK;
}

} // extern "C"

std::vector<PgrcShardView> pgrc_multi_shards(pgrc_match_ctx *f) {
    std::vector<PgrcShardView> v;
    pgrc_multi *m = f->multi;
    for (size_t r = 0; r < m->child.size(); r++)
        v.push_back({m->child[r], m->lo.empty() ? 0 : m->lo[r], m->lo.empty() ? 0 : m->hi[r]});
    return v;
}

void pgrc_multi_destroy(pgrc_match_ctx *f) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    if (m) {
        for (size_t r = 0; r < m->comm.size(); r++)
            if (m->comm[r] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comm[r]);
        for (pgrc_match_ctx *c : m->child) pgrc_match_destroy(c);
        delete m;
    }
    f->multi = nullptr;
    delete f;
}

// ------------------------------------------------------------------ pseudogenome

// In-place all-gather of the packed text: child r holds words [r*sw, (r+1)*sw) of its own pg2[0] and receives the rest.
static int allgather_text(pgrc_match_ctx *f, uint64_t sw) {
    pgrc_multi *m = f->multi;
    const size_t k = m->child.size();
    m->ms_allgather = 0;
    // PGRC_ALLGATHER = "rccl" / "copy": force an engine (tests); default by device list
    const bool force_rccl = f->opt.allgather == 1, force_copy = f->opt.allgather == 2;
    if (k == 1 && !force_rccl) return PGRC_OK;
    const bool use_rccl = force_rccl || (m->distinct && !force_copy);
    const auto t0 = std::chrono::steady_clock::now();
    if (use_rccl) {
        if (!rccl_load(m->rccl, f->err)) return PGRC_E_DEVICE;
        if (m->comm.empty()) {
            m->comm.assign(k, nullptr);
            ncclResult_t nr = m->rccl.CommInitAll(m->comm.data(), (int)k, m->dev.data());
            if (nr != ncclSuccess) {
                m->comm.clear();
                f->err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(nr) +
                         (m->distinct ? "" : " (a device is listed twice: RCCL needs distinct devices)");
                return PGRC_E_DEVICE;
            }
        }
        ncclResult_t nr = m->rccl.GroupStart();
        for (size_t r = 0; r < k && nr == ncclSuccess; r++) {
            pgrc_match_ctx *c = m->child[r];
            PgrcDeviceScope scope(c->device);
            uint32_t *buf = (uint32_t *)c->pg2[0].p;
            nr = m->rccl.AllGather(buf + r * sw, buf, (size_t)sw, ncclUint32, m->comm[r], c->stream);
        }
        const ncclResult_t ge = m->rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ge;
        if (nr != ncclSuccess) { f->err = std::string("ncclAllGather: ") + m->rccl.GetErrorString(nr); return PGRC_E_DEVICE; }
    } else {
        // every device pulls the other slices into its own copy (peer copies over xGMI between distinct devices,
        // plain device-to-device copies between contexts that share a device)
        for (size_t r = 0; r < k; r++) {
            pgrc_match_ctx *c = m->child[r];
            PgrcDeviceScope scope(c->device);
            for (size_t s = 0; s < k; s++) {
                if (s == r) continue;
                const uint32_t *src = (const uint32_t *)m->child[s]->pg2[0].p + s * sw;
                uint32_t *dst = (uint32_t *)c->pg2[0].p + s * sw;
                hipError_t he = m->dev[s] == m->dev[r]
                                    ? hipMemcpyAsync(dst, src, sw * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream)
                                    : hipMemcpyPeerAsync(dst, m->dev[r], src, m->dev[s], sw * sizeof(uint32_t), c->stream);
                if (he != hipSuccess) { f->err = std::string("all-gather copy: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
            }
        }
    }
    for (size_t r = 0; r < k; r++) {
        pgrc_match_ctx *c = m->child[r];
        PgrcDeviceScope scope(c->device);
        hipError_t he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { f->err = std::string("all-gather: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
    }
    m->ms_allgather = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PGRC_OK;
}

int pgrc_multi_set_pg_ascii(pgrc_match_ctx *f, const char *pg, uint64_t G) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    const size_t k = m->child.size();
    const uint64_t words = (G + 15) / 16, sw = (words + k - 1) / k;     // k * sw <= words + k - 1 < words + PGRC_PG_PAD_WORDS
    int e = on_children(f, [&](size_t r) -> int {
        pgrc_match_ctx *c = m->child[r];
        PgrcDeviceScope scope(c->device);
        if (!scope.ok) { c->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
        int ce = pgrc_pg_alloc(c, G);
        if (ce) return ce;
        const uint64_t lo = std::min<uint64_t>(G, (uint64_t)r * sw * 16), hi = std::min<uint64_t>(G, (uint64_t)(r + 1) * sw * 16);
        if (hi > lo && (ce = pgrc_match_pack_pg_slice(c, pg + lo, hi - lo, (uint32_t *)c->pg2[0].p + r * sw))) return ce;
        hipError_t he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { c->err = std::string("pack: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
        return PGRC_OK;
    });
    if (e) return e;
    if ((e = allgather_text(f, sw))) return e;
    for (pgrc_match_ctx *c : m->child) c->have_pg = true;
    f->G = G;
    f->cp = m->child[0]->cp;
    f->have_pg = true;
    return PGRC_OK;
}

int pgrc_multi_set_pg_packed_device(pgrc_match_ctx *f, const void *d_words, uint64_t G) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    int e = on_children(f, [&](size_t r) { return pgrc_match_set_pg_packed_device(m->child[r], d_words, G); });
    if (e) return e;
    f->G = G;
    f->cp = m->child[0]->cp;
    f->have_pg = true;
    return PGRC_OK;
}

// the output buffer lives on one device: the shard on that device does the packing
int pgrc_multi_pack_pg_slice(pgrc_match_ctx *f, const char *pg, uint64_t count, void *d_words_out) {
    pgrc_multi *m = f->multi;
    hipPointerAttribute_t at;
    int owner = -1;
    if (hipPointerGetAttributes(&at, d_words_out) == hipSuccess) owner = at.device;
    else (void)hipGetLastError();
    for (size_t r = 0; r < m->child.size(); r++)
        if (m->dev[r] == owner) {
            int e = pgrc_match_pack_pg_slice(m->child[r], pg, count, d_words_out);
            if (e) f->err = m->child[r]->err;
            return e;
        }
    f->err = "pack_pg_slice: the output buffer is not on one of this context's devices";
    return PGRC_E_PARAM;
}

int pgrc_multi_export_pg(pgrc_match_ctx *f, int strand, uint32_t *words) {
    int e = pgrc_match_export_pg(f->multi->child[0], strand, words);
    if (e) f->err = f->multi->child[0]->err;
    return e;
}

int pgrc_multi_export_index(pgrc_match_ctx *f, int strand, uint32_t *cumm, uint32_t *positions, uint64_t *count) {
    int e = pgrc_match_export_index(f->multi->child[0], strand, cumm, positions, count);   // the index is replicated
    if (e) f->err = f->multi->child[0]->err;
    return e;
}

// ------------------------------------------------------------------ reads

int pgrc_multi_begin_reads(pgrc_match_ctx *f, uint64_t n) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    const size_t k = m->child.size();
    if (n >= (1ull << 32) - 1) { f->err = "reads count must stay below 2^32-1 (uint_reads_cnt_max, pg-config.h:21-22)"; return PGRC_E_PARAM; }
    m->lo.assign(k, 0);
    m->hi.assign(k, 0);
    for (size_t r = 0; r < k; r++) shard_range(n, r, k, &m->lo[r], &m->hi[r]);
    f->n = n;
    f->have_reads = false;
    f->have_results = false;
    m->up_next = 0;
    m->up_open = false;
    int e = on_children(f, [&](size_t r) { return pgrc_match_begin_reads(m->child[r], m->hi[r] - m->lo[r]); });
    if (e) return e;
    m->up_open = true;
    return PGRC_OK;
}

// rows [up_next, up_next + count) of the announced set: every shard takes its part
int pgrc_multi_append_reads(pgrc_match_ctx *f, const void *rows, uint64_t count, int32_t symbols) {
    pgrc_multi *m = f->multi;
    if (!m->up_open || m->up_next + count > f->n) { f->err = "append_reads: outside begin/end or too many rows"; return PGRC_E_STATE; }
    const uint32_t L = f->prm.read_len;
    const uint64_t rb = symbols == 0 ? L : symbols == 4 ? (L + 3) / 4 : (L + 2) / 3;
    const uint64_t a = m->up_next, b = m->up_next + count;
    int e = on_children(f, [&](size_t r) -> int {
        const uint64_t x = std::max(a, m->lo[r]), y = std::min(b, m->hi[r]);
        if (y <= x) return PGRC_OK;
        const uint8_t *src = (const uint8_t *)rows + (x - a) * rb;
        return symbols == 0 ? pgrc_match_append_reads_ascii(m->child[r], (const char *)src, y - x)
                            : pgrc_match_append_reads_packed(m->child[r], src, y - x, symbols);
    });
    if (e) return e;
    m->up_next = b;
    return PGRC_OK;
}

int pgrc_multi_end_reads(pgrc_match_ctx *f) {
    pgrc_multi *m = f->multi;
    if (!m->up_open || m->up_next != f->n) { f->err = "end_reads: fewer rows appended than announced"; return PGRC_E_STATE; }
    int e = on_children(f, [&](size_t r) { return pgrc_match_end_reads(m->child[r]); });
    if (e) return e;
    m->up_open = false;
    f->have_reads = true;
    return PGRC_OK;
}

int pgrc_multi_set_reads_device(pgrc_match_ctx *f, const void *d_words, uint64_t n, uint64_t stride) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    if (m->child.size() != 1) { f->err = "set_reads_device: reads resident on one device cannot feed several"; return PGRC_E_PARAM; }
    int e = pgrc_match_set_reads_device(m->child[0], d_words, n, stride);
    if (e) { f->err = m->child[0]->err; return e; }
    m->lo.assign(1, 0);
    m->hi.assign(1, n);
    f->n = n;
    f->have_reads = true;
    f->have_results = false;
    return PGRC_OK;
}

// ------------------------------------------------------------------ matching

int pgrc_multi_init_results(pgrc_match_ctx *f) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    if (!f->have_reads) { f->err = "init_results: no reads set"; return PGRC_E_STATE; }
    int e = on_children(f, [&](size_t r) { return pgrc_match_init_results(m->child[r]); });
    if (!e) f->have_results = true;
    return e;
}

int pgrc_multi_set_results(pgrc_match_ctx *f, const uint64_t *pos, const uint8_t *rc, const uint8_t *mism) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    if (!f->have_reads) { f->err = "set_results: no reads set"; return PGRC_E_STATE; }
    int e = on_children(f, [&](size_t r) { return pgrc_match_set_results(m->child[r], pos + m->lo[r], rc + m->lo[r], mism + m->lo[r]); });
    if (!e) f->have_results = true;
    return e;
}

int pgrc_multi_run(pgrc_match_ctx *f, int first, int last) {
    pgrc_export_drop_view(f);
    pgrc_multi *m = f->multi;
    if (!f->have_pg || !f->have_reads) { f->err = "run: set the pseudogenome and the reads first"; return PGRC_E_STATE; }
    int e = on_children(f, [&](size_t r) -> int {
        pgrc_match_ctx *c = m->child[r];
        return first == last ? pgrc_match_run_pass(c, first) : pgrc_match_run(c, 1);
    });
    if (e) return e;
    f->have_results = true;
    memset(f->hist, 0, sizeof f->hist);
    f->matched = 0;
    for (pgrc_match_ctx *c : m->child) {
        for (int x = 0; x < 256; x++) f->hist[x] += c->hist[x];
        f->matched += c->matched;
    }
    return PGRC_OK;
}

int pgrc_multi_get_results(pgrc_match_ctx *f, uint64_t *pos, uint8_t *rc, uint8_t *mism, uint64_t hist[256], uint64_t *matched) {
    pgrc_multi *m = f->multi;
    if (!f->have_results) { f->err = "get_results: nothing computed"; return PGRC_E_STATE; }
    const size_t k = m->child.size();
    std::vector<uint64_t> h(k * 256, 0), mt(k, 0);
    int e = on_children(f, [&](size_t r) {
        const uint64_t lo = m->lo[r];
        return pgrc_match_get_results(m->child[r], pos ? pos + lo : nullptr, rc ? rc + lo : nullptr, mism ? mism + lo : nullptr,
                                      &h[r * 256], &mt[r]);
    });
    if (e) return e;
    if (hist) {
        memset(hist, 0, 256 * sizeof(uint64_t));
        for (size_t r = 0; r < k; r++)
            for (int x = 0; x < 256; x++) hist[x] += h[r * 256 + x];
    }
    if (matched) {
        *matched = 0;
        for (size_t r = 0; r < k; r++) *matched += mt[r];
    }
    return PGRC_OK;
}

// cum[] is global (reads in set order); codes / offsets of shard r start at cum[lo_r]
int pgrc_multi_extract_mismatches(pgrc_match_ctx *f, const uint8_t *reversed_flags, uint64_t *cum, uint8_t *codes, uint16_t *offsets) {
    pgrc_multi *m = f->multi;
    if (!f->have_results || !f->have_pg) { f->err = "extract_mismatches: run first"; return PGRC_E_STATE; }
    const size_t k = m->child.size();
    std::vector<std::vector<uint64_t>> lc(k);
    int e = on_children(f, [&](size_t r) {
        lc[r].assign(m->hi[r] - m->lo[r] + 1, 0);
        return pgrc_match_extract_mismatches(m->child[r], reversed_flags ? reversed_flags + m->lo[r] : nullptr, lc[r].data(), nullptr, nullptr);
    });
    if (e) return e;
    std::vector<uint64_t> base(k + 1, 0);
    for (size_t r = 0; r < k; r++) base[r + 1] = base[r] + lc[r].back();
    for (size_t r = 0; r < k; r++)
        for (uint64_t i = 0; i + m->lo[r] < m->hi[r]; i++) cum[m->lo[r] + i] = base[r] + lc[r][i];
    cum[f->n] = base[k];
    if (!codes || !offsets || !base[k]) return PGRC_OK;
    return on_children(f, [&](size_t r) -> int {
        if (base[r + 1] == base[r]) return PGRC_OK;
        return pgrc_match_extract_mismatches(m->child[r], reversed_flags ? reversed_flags + m->lo[r] : nullptr, lc[r].data(),
                                             codes + base[r], offsets + base[r]);
    });
}

int pgrc_multi_set_profiling(pgrc_match_ctx *f, int enabled) {
    pgrc_multi *m = f->multi;
    return on_children(f, [&](size_t r) { return pgrc_match_set_profiling(m->child[r], enabled); });
}

// work counters summed over the shards; device times = the slowest shard's (the shards run side by side)
int pgrc_multi_get_counters(pgrc_match_ctx *f, pgrc_match_counters *out) {
    pgrc_multi *m = f->multi;
    memset(out, 0, sizeof *out);
    for (pgrc_match_ctx *c : m->child) {
        pgrc_match_counters x;
        int e = pgrc_match_get_counters(c, &x);
        if (e) return e;
        for (int s = 0; s < 2; s++) {
            out->searched[s] += x.searched[s];
            out->candidates[s] += x.candidates[s];
            out->probes[s] += x.probes[s];
            out->entry_fetches[s] += x.entry_fetches[s];
            out->verifies[s] += x.verifies[s];
            out->index_entries[s] = x.index_entries[s];
            out->ms_index[s] = std::max(out->ms_index[s], x.ms_index[s]);
            out->ms_match[s] = std::max(out->ms_match[s], x.ms_match[s]);
        }
        out->ms_other = std::max(out->ms_other, x.ms_other);
        out->ms_total = std::max(out->ms_total, x.ms_total);
        out->ms_screen = std::max(out->ms_screen, x.ms_screen);
        out->screened = std::max(out->screened, x.screened);
        out->redo_reads += x.redo_reads;
        out->schedule_downgraded |= x.schedule_downgraded;
        for (int k = 0; k < 5; k++) out->dual[k] += x.dual[k];
        out->dual_seed_probes += x.dual_seed_probes;
    }
    out->ms_allgather = m->ms_allgather;
    return PGRC_OK;
}
