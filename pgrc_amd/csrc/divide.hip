// divide.hip -- making the packed read sets (include/pgrc_reads.h) on gfx950.
//
// Reference behaviour restated (not translated):
//   DividedPCLReadsSets::getQualityDivisionBasedReadsSets, readsset/DividedPCLReadsSets.cpp:59-100 (which set a read
//   goes to), QualityDividingReadsSetIterator::isQualityHigh / containsN, readsset/iterator/DivisionReadsSetDecorators.cpp:
//   14, :30-38, :66-69, PgHelpers::qualityScore2correctProbArithAvg + qualityLut, utils/helper.cpp:452-475, :284-327,
//   PackedConstantLengthReadsSet::addRead -> SymbolsPackingFacility::packSequence, coders/SymbolsPackingFacility.cpp:147-186.
//
// The reference walks the FASTQ records one by one: classify, append to a vector, push an index.  Every step depends on
// the record alone except WHERE it lands in its set -- a prefix count.  So a batch of records (two row arrays: symbols and
// quality characters) becomes
//   1. k_dv_symbols : every byte of the symbol rows once, coalesced: which reads hold an 'N', which a byte outside ACGNT;
//   2. k_dv_quality : the quality test -- the sum of the per-position probabilities in fixed point over coalesced pieces of
//                     the quality rows: it equals the reference's sum of doubles bit for bit (see the kernel), so the
//                     comparison with the error limit sees the same double;
//   3. k_dv_class + three exclusive scans: the read's set and its row in it;
//   4. k_dv_pack    : one thread per OUTPUT byte (3 or 4 symbols as base-5 / base-4 digits), rows written in place.
// All byte work, bound by the host link (2 x read_len bytes per read up, read_len / 4 down): no MFMA.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "ctx.h"
#include "pgrc_reads.h"

#define DV_HQ 0u
#define DV_LQ 1u
#define DV_N 2u
#define DVF_HAS_N 1u
#define DVF_BAD 2u          // a byte outside ACGNT in the symbol row
#define DVF_BAD_Q 4u        // a quality character beyond the reference's table (133 entries)

struct pgrc_divider {
    pgrc_match_ctx base;     // device, stream, error text, scan scratch (only the plumbing of the matcher's context)
    pgrc_divide_params prm{};
    int suffix_pos = 0;
    DevBuf d_reads, d_quals, d_flags, d_high, d_cls, d_cnt[3], d_bsum, d_rows[3], d_idx[2], d_lut, d_err;
    DevBuf d_text[2], d_nl[2], d_ls[2];   // FASTQ text of one or two files, newlines per 16 bytes, line starts
    // the results of a run on the host: pinned, grow-only, the divider's (fresh pageable pages would be touched for the first
    // time by the copy, at one thread's page-fault rate: 11 GB/s against the link's 56)
    struct HostBuf { void *p = nullptr; size_t bytes = 0; } h_rows[3], h_idx[2];
    hipEvent_t ev[4]{};
    bool have_ev = false;
    float ms[3] = {0, 0, 0};
    bool last_terminal = false;   // the last pgrc_divider_run_fastq took what the reference's iteration would take before it ends
};

static int dv_host_ensure(pgrc_divider *d, pgrc_divider::HostBuf &b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return PGRC_OK;
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 4096);
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        d->base.err = "divider: pinned host allocation of " + std::to_string(want) + " bytes failed";
        return PGRC_E_ALLOC;
    }
    b.bytes = want;
    return PGRC_OK;
}

static thread_local std::string g_div_create_err;

// ------------------------------------------------------------------------------------------------ device side

__global__ void __launch_bounds__(256)
k_dv_symbols(const uint8_t *__restrict__ reads, uint64_t nbytes, uint32_t L, uint32_t *__restrict__ flags) {
    const uint64_t nwords = (nbytes + 3) / 4;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        if (w * 4 + 4 <= nbytes) v = ((const uint32_t *)reads)[w];
        else for (uint32_t k = 0; w * 4 + k < nbytes; k++) v |= (uint32_t)reads[w * 4 + k] << (8 * k);
        // (the common case first: four bytes of ACGT)
        bool plain = true;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t ch = (v >> (8 * k)) & 255u;
            plain &= ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        }
        if (plain) continue;
        for (uint32_t k = 0; k < 4 && w * 4 + k < nbytes; k++) {
            const uint32_t ch = (v >> (8 * k)) & 255u;
            if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') continue;
            atomicOr(&flags[(w * 4 + k) / L], ch == 'N' ? DVF_HAS_N : DVF_BAD);
        }
    }
}

// isQualityHigh (DivisionReadsSetDecorators.cpp:30-38).  lut = qualityLut (helper.cpp:284-327).  The reference sums the
// table's FLOAT entries of all positions as doubles (qualityScore2correctProbArithAvg(quality, 1, true), helper.cpp:452-475:
// two accumulators over the even and the odd positions, added at the end), divides by the length and tests
// 1 - q <= error_limit.  The entries are 0 or lie in [0.2, 1]: multiples of 2^-26 below 2^0, so any sum of up to 255 of
// them is a multiple of 2^-26 below 2^8 -- 34 bits -- and EXACT in a double whatever the order of the additions.  The
// sum is therefore taken in fixed point (entry * 2^26 as an integer), by all lanes over coalesced 16-byte pieces of the
// quality rows; converted back it is the reference's val1 + val2 bit for bit, and the division and the subtraction
// that follow are single IEEE operations.
__global__ void __launch_bounds__(256)
k_dv_quality(const uint8_t *__restrict__ quals, uint64_t n, uint32_t L, int simplified, int suffix_pos, double error_limit,
             const float *__restrict__ lut, uint8_t *__restrict__ high, uint32_t *__restrict__ flags) {
    if (simplified) {
        const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        // `quality[suffix_pos] > '#'` on a std::string of (signed) chars (DivisionReadsSetDecorators.cpp:33): a byte of 128 or
        // more is negative there, and suffix_pos == L (error_limit 0) reads the string's terminating 0 -- never high
        if (r < n) high[r] = ((uint32_t)suffix_pos < L && (int8_t)quals[r * L + suffix_pos] > (int8_t)'#') ? 1 : 0;
        return;
    }
    __shared__ uint32_t fx[256];                          // entry * 2^26; characters past the table: flagged, counted as 0
    __shared__ unsigned long long sum[4][64];
    __shared__ uint32_t badq[4][64];
    fx[threadIdx.x] = threadIdx.x < 133 ? (uint32_t)(lut[threadIdx.x] * 67108864.0f) : 0u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    sum[wave][lane] = 0ull;
    badq[wave][lane] = 0u;
    __syncthreads();
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + wave) * 64;                   // this wave's 64 reads
    if (r0 < n) {
        const uint32_t nr = (uint32_t)min((uint64_t)64, n - r0);
        const uint32_t nbytes = nr * L;
        const uint8_t *src = quals + r0 * L;                                       // (64 * L bytes per wave: 16-byte aligned)
        for (uint32_t off = lane * 16; off < nbytes; off += 64 * 16) {
            const uint4 v = *(const uint4 *)(src + off);                           // (the buffer has 16 bytes of slack)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t cur = off / L, left = (cur + 1) * L - off;                    // read of the first byte, bytes left in its row
            unsigned long long acc = 0;
            uint32_t bad = 0;
            for (uint32_t k = 0; k < 16 && off + k < nbytes; k++) {
                if (left == 0) {
                    atomicAdd(&sum[wave][cur], acc);
                    if (bad) atomicOr(&badq[wave][cur], 1u);
                    acc = 0;
                    bad = 0;
                    cur++;
                    left = L;
                }
                const uint32_t ch = (w[k >> 2] >> (8 * (k & 3))) & 255u;
                acc += fx[ch];
                bad |= ch > 132u;
                left--;
            }
            atomicAdd(&sum[wave][cur], acc);
            if (bad) atomicOr(&badq[wave][cur], 1u);
        }
    }
    __syncthreads();
    if (r0 + lane < n) {
        const double total = (double)sum[wave][lane] * (1.0 / 67108864.0);         // = val1 + val2 of the reference, exactly
        const double p = total / (double)(int)L;
        high[r0 + lane] = (1 - p <= error_limit) ? 1 : 0;
        if (badq[wave][lane]) atomicOr(&flags[r0 + lane], DVF_BAD_Q);
    }
}

// the set of a read (DividedPCLReadsSets.cpp:68-87) and its 0/1 entry in the three count arrays (scanned afterwards)
__global__ void __launch_bounds__(256)
k_dv_class(const uint32_t *__restrict__ flags, const uint8_t *__restrict__ high, uint64_t n, int n_apart, int separate_n, int by_quality,
           uint8_t *__restrict__ cls, uint32_t *__restrict__ c_hq, uint32_t *__restrict__ c_lq, uint32_t *__restrict__ c_n, uint32_t *err) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    uint32_t c = 3;                                       // (entry n: the end of the scans)
    if (r < n) {
        const uint32_t fl = flags[r];
        if (fl & (DVF_BAD | DVF_BAD_Q)) atomicOr(err, fl & (DVF_BAD | DVF_BAD_Q));
        if (n_apart && (fl & DVF_HAS_N)) c = separate_n ? DV_N : DV_LQ;
        else if (by_quality && !high[r]) c = DV_LQ;
        else c = DV_HQ;
        cls[r] = (uint8_t)c;
    }
    c_hq[r] = c == DV_HQ;
    c_lq[r] = c == DV_LQ;
    c_n[r] = c == DV_N;
}

struct DvPackArgs {
    const uint8_t *reads;
    const uint8_t *cls;
    const uint32_t *slot[3];      // exclusive counts: row of the read in its set
    uint8_t *rows[3];
    uint32_t *idx[2];             // batch index of every LQ / N read
    uint32_t row_bytes[3], per[3], base[3];
    uint32_t row_bytes_max, L;
    uint64_t n;
};

// symbolOrder of "ACGT" / "ACGNT" (ReadsSetBase.h:76-81): the position in the list
__device__ __forceinline__ uint32_t dv_order(uint32_t ch, uint32_t base) {
    const uint32_t x = (ch >> 1) & 3u;                    // A0 C1 T2 G3
    const uint32_t acgt = x ^ (x >> 1);                   // A0 C1 G2 T3
    if (base == 4) return acgt;
    return ch == 'N' ? 3u : (acgt == 3u ? 4u : acgt);     // A0 C1 G2 N3 T4
}

// packSequence (SymbolsPackingFacility.cpp:147-186): byte b of a row = symbols [b * per, (b + 1) * per) as digits, the first
// one the most significant; digits past the end of the read are 0
__global__ void __launch_bounds__(256)
k_dv_pack(const DvPackArgs a) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = g / a.row_bytes_max;
    const uint32_t b = (uint32_t)(g % a.row_bytes_max);
    if (r >= a.n) return;
    const uint32_t c = a.cls[r];
    if (b >= a.row_bytes[c]) return;
    const uint32_t per = a.per[c], base = a.base[c], slot = a.slot[c][r];
    const uint8_t *s = a.reads + r * a.L;
    uint32_t v = 0;
    for (uint32_t j = 0; j < per; j++) {
        v *= base;
        const uint32_t x = b * per + j;
        if (x < a.L) v += dv_order(s[x], base);
    }
    a.rows[c][(uint64_t)slot * a.row_bytes[c] + b] = (uint8_t)v;
    if (b == 0 && c != DV_HQ) a.idx[c - 1][slot] = (uint32_t)r;
}

// ---- FASTQ text: lines and rows
// newlines per 16 bytes (entry n16: 0, the end of the scan)
__global__ void __launch_bounds__(256)
k_fq_count(const uint8_t *__restrict__ text, uint64_t bytes, uint64_t n16, uint32_t *__restrict__ cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n16) return;
    uint32_t c = 0;
    if (i < n16) {
        const uint4 v = *(const uint4 *)(text + 16 * i);                 // (32 zero bytes follow the text)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t k = 0; k < 16 && 16 * i + k < bytes; k++) c += ((w[k >> 2] >> (8 * (k & 3))) & 255u) == '\n';
    }
    cnt[i] = c;
}
// ls[k] = first byte of line k: ls[0] = 0, ls[j + 1] = position after the j-th newline; ls[nl + 1 ..] = bytes (the end of a last
// line that has no newline)
__global__ void __launch_bounds__(256)
k_fq_starts(const uint8_t *__restrict__ text, uint64_t bytes, uint64_t n16, const uint32_t *__restrict__ before, uint32_t nl,
            uint32_t *__restrict__ ls) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ls[0] = 0;
        for (uint32_t k = 1; k <= 6; k++) ls[nl + k] = (uint32_t)bytes;
    }
    if (i >= n16) return;
    uint32_t j = before[i];
    const uint4 v = *(const uint4 *)(text + 16 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16 && 16 * i + k < bytes; k++)
        if (((w[k >> 2] >> (8 * (k & 3))) & 255u) == '\n') ls[++j] = (uint32_t)(16 * i + k + 1);
}

struct FqRowsArgs {
    const uint8_t *text[2];
    const uint32_t *ls[2];
    uint32_t len[2], nl[2];     // bytes and newlines of each text
    uint32_t paired, rc_second, L;
    uint64_t n;
    uint8_t *reads, *quals;     // quals: null when the division does not ask them
    uint32_t *err;
};
__device__ __forceinline__ bool fq_alpha(uint32_t ch) { return (ch - 'A') < 26u || (ch - 'a') < 26u; }
// What PgHelpers' complement table does to a symbol of the second file's reads (utils/helper.cpp:261-282): both cases of
// A C G T N U and of the IUPAC pairs Y/R K/M B/V D/H map to the UPPER-case complement (so a lower-case `a` comes out as a
// valid `T`), every other byte to 0 -- which k_dv_symbols then reports as a symbol outside ACGNT, as the reference's
// packing would.
__device__ __forceinline__ uint32_t fq_complement(uint32_t ch) {
    switch (ch | 0x20u) {                // (letters only: the cases below are lower-case letters)
    case 'a': return 'T';
    case 'c': return 'G';
    case 'g': return 'C';
    case 't': return 'A';
    case 'n': return 'N';
    case 'u': return 'A';
    case 'y': return 'R';
    case 'r': return 'Y';
    case 'k': return 'M';
    case 'm': return 'K';
    case 'b': return 'V';
    case 'd': return 'H';
    case 'h': return 'D';
    case 'v': return 'B';
    default: return 0u;
    }
}

// record k of the batch = record k (or k / 2 of file k % 2) of the text: column x < L copies symbol and quality character,
// column L checks that the run of letters ends there (FASTQReadsSourceIterator::moveNext, :218-220)
__global__ void __launch_bounds__(256)
k_fq_rows(const FqRowsArgs a) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k = g / (a.L + 1);
    const uint32_t x = (uint32_t)(g % (a.L + 1));
    if (k >= a.n) return;
    const uint32_t f = a.paired ? (uint32_t)(k & 1) : 0u;
    const uint64_t r = a.paired ? k >> 1 : k;
    const uint8_t *t = a.text[f];
    const uint32_t *ls = a.ls[f];
    // line j spans [ls[j], ls[j + 1] - 1) -- up to its newline --, a last line without one [ls[j], ls[j + 1]) = up to the end
    const uint32_t nl = a.nl[f];
    const uint32_t j1 = (uint32_t)(4 * r + 1), j3 = j1 + 2;
    const uint32_t s = ls[j1];
    const uint32_t e = ls[j1 + 1] - (j1 < nl ? 1u : 0u);
    const uint32_t ch = (s + x < e) ? t[s + x] : 0u;
    if (x == a.L) {
        if (fq_alpha(ch)) atomicOr(a.err, 1u);                         // the read is longer than read_len
        return;
    }
    if (!fq_alpha(ch)) atomicOr(a.err, 1u);                            // ... or shorter
    const bool rc = a.rc_second && f == 1;
    a.reads[k * a.L + (rc ? a.L - 1 - x : x)] = (uint8_t)(rc ? fq_complement(ch) : ch);
    if (a.quals) {
        const uint32_t qs = ls[j3];
        const uint32_t qe = ls[j3 + 1] - (j3 < nl ? 1u : 0u);
        a.quals[k * a.L + x] = (qs + x < qe) ? t[qs + x] : (uint8_t)0;  // quality.resize(length): cut or zero-padded, never reversed
    }
}

// ------------------------------------------------------------------------------------------------ host side

#define DIV_TRY(d, expr)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (d)->base.err = std::string(#expr) + ": " + hipGetErrorString(e__);              \
            return pgrc_hip_code(e__);                                                       \
        }                                                                                    \
    } while (0)

extern "C" {

const char *pgrc_divider_last_error(const pgrc_divider *d) { return d ? d->base.err.c_str() : g_div_create_err.c_str(); }

int pgrc_divider_create(const pgrc_divide_params *p, pgrc_divider **out) {
    if (!p || !out) return PGRC_E_PARAM;
    *out = nullptr;
    if (p->read_len == 0 || p->read_len > 255) { g_div_create_err = "read_len must be 1..255"; return PGRC_E_PARAM; }
    if (!(p->error_limit >= 0)) { g_div_create_err = "error_limit must be >= 0"; return PGRC_E_PARAM; }
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev == 0) {
        g_div_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(he) + " (devices: " + std::to_string(ndev) + ")";
        return PGRC_E_NO_DEVICE;
    }
    int dev = p->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    PgrcDeviceScope scope(dev < ndev ? dev : 0);
    if (dev >= ndev || !scope.ok) { g_div_create_err = "hipSetDevice(" + std::to_string(dev) + ") failed"; return PGRC_E_NO_DEVICE; }
    (void)hipGetLastError();
    pgrc_divider *d = new pgrc_divider();
    d->prm = *p;
    d->base.device = dev;
    // suffix_pos = read_length * (1 - error_level), a double truncated to int (DivisionReadsSetDecorators.cpp:14)
    d->suffix_pos = (int)((double)p->read_len * (1 - p->error_limit));
    // (suffix_pos == read_len -- error_limit 0 -- is what the reference accepts too: it tests the quality string's terminator)
    if (p->error_limit < 1 && p->simplified_suffix_mode && (d->suffix_pos < 0 || d->suffix_pos > (int)p->read_len)) {
        g_div_create_err = "simplified suffix mode: the tested position lies outside the read";
        delete d;
        return PGRC_E_PARAM;
    }
    he = hipStreamCreateWithFlags(&d->base.stream, hipStreamNonBlocking);
    // qualityLut (helper.cpp:284-327): the probability that a base call of Phred quality q is right, 1 - 10^(-q / 10), as a
    // float, for the characters '!' + 0 .. '!' + 40; 1 for the next 59; 0 below '!'.  (The reference spells the 41 values
    // out as decimal literals; their float roundings equal those of the formula -- tests/test_divide_oracle.py.)
    float lut[133];
    for (int c = 0; c < 133; c++) lut[c] = c < 33 ? 0.f : (c - 33 <= 40 ? (float)(1.0 - pow(10.0, -(double)(c - 33) / 10.0)) : 1.f);
    int e = PGRC_OK;
    if (he == hipSuccess) e = pgrc_buf_ensure(&d->base, d->d_lut, sizeof lut);
    if (he == hipSuccess && !e) e = pgrc_buf_ensure(&d->base, d->d_err, sizeof(uint32_t));
    if (he == hipSuccess && !e) he = hipMemcpyAsync(d->d_lut.p, lut, sizeof lut, hipMemcpyHostToDevice, d->base.stream);
    if (he == hipSuccess && !e) he = hipStreamSynchronize(d->base.stream);
    if (he != hipSuccess || e) {
        g_div_create_err = he != hipSuccess ? std::string("divider: ") + hipGetErrorString(he) : d->base.err;
        const int code = he != hipSuccess ? pgrc_hip_code(he) : e;
        pgrc_divider_destroy(d);
        return code;
    }
    *out = d;
    return PGRC_OK;
}

void pgrc_divider_destroy(pgrc_divider *d) {
    if (!d) return;
    PgrcDeviceScope scope(d->base.device);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&d->d_reads, &d->d_quals, &d->d_flags, &d->d_high, &d->d_cls, &d->d_cnt[0], &d->d_cnt[1], &d->d_cnt[2], &d->d_bsum,
                      &d->d_rows[0], &d->d_rows[1], &d->d_rows[2], &d->d_idx[0], &d->d_idx[1], &d->d_lut, &d->d_err,
                      &d->d_text[0], &d->d_text[1], &d->d_nl[0], &d->d_nl[1], &d->d_ls[0], &d->d_ls[1]};
    for (DevBuf *b : bufs) pgrc_buf_free(*b);
    for (auto *h : {&d->h_rows[0], &d->h_rows[1], &d->h_rows[2], &d->h_idx[0], &d->h_idx[1]})
        if (h->p) (void)hipHostFree(h->p);
    if (d->have_ev)
        for (auto &x : d->ev) (void)hipEventDestroy(x);
    if (d->base.stream) (void)hipStreamDestroy(d->base.stream);
    delete d;
}

int pgrc_divider_last_was_terminal(const pgrc_divider *d) { return d && d->last_terminal ? 1 : 0; }

int pgrc_divider_last_ms(const pgrc_divider *d, float ms[3]) {
    if (!d || !ms) return PGRC_E_PARAM;
    memcpy(ms, d->ms, sizeof d->ms);
    return PGRC_OK;
}

// the sets of n records whose symbol rows (and quality rows, when the division asks them) lie in d_reads / d_quals;
// ev[1] has been recorded by the caller where its own part (upload or parsing) ended
static int divide_resident(pgrc_divider *d, uint64_t n, pgrc_divided_reads *out) {
    const pgrc_divide_params &p = d->prm;
    const bool by_quality = p.error_limit < 1;
    pgrc_match_ctx *c = &d->base;
    const uint32_t L = p.read_len;
    const bool n_apart = p.separate_n_reads_set || p.n_reads_lq;
    // alphabets (DividedPCLReadsSets.cpp:10-21)
    const uint32_t sym[3] = {n_apart ? 4u : 5u, p.separate_n_reads_set ? 4u : 5u, p.separate_n_reads_set ? 5u : 0u};
    uint32_t per[3], rb[3];
    for (int k = 0; k < 3; k++) {
        per[k] = sym[k] == 4 ? 4u : 3u;                  // SymbolsPackingFacility::maxSymbolsPerElement: 4^4 - 1, 5^3 - 1 <= 255
        rb[k] = sym[k] ? (L + per[k] - 1) / per[k] : 0u;
    }
    out->hq_symbols = sym[0]; out->lq_symbols = sym[1]; out->n_symbols = sym[2];
    out->hq_row_bytes = rb[0]; out->lq_row_bytes = rb[1]; out->n_row_bytes = rb[2];
    if (n == 0) return PGRC_OK;
    int e;
    const size_t bytes = (size_t)n * L;
    if ((e = pgrc_buf_ensure(c, d->d_flags, n * sizeof(uint32_t))) || (e = pgrc_buf_ensure(c, d->d_high, n)) || (e = pgrc_buf_ensure(c, d->d_cls, n)) ||
        (e = pgrc_buf_ensure(c, d->d_bsum, (pgrc_ps_scan_blocks(n + 1) + 1) * sizeof(uint32_t))))
        return e;
    for (int k = 0; k < 3; k++)
        if ((e = pgrc_buf_ensure(c, d->d_cnt[k], (n + 1) * sizeof(uint32_t)))) return e;
    hipStream_t s = c->stream;
    DIV_TRY(d, hipMemsetAsync(d->d_flags.p, 0, n * sizeof(uint32_t), s));
    DIV_TRY(d, hipMemsetAsync(d->d_err.p, 0, sizeof(uint32_t), s));
    const uint32_t g = (uint32_t)((n + 1 + 255) / 256);
    hipLaunchKernelGGL(k_dv_symbols, dim3((uint32_t)std::min<uint64_t>((bytes / 4 + 255) / 256 + 1, 65536)), dim3(256), 0, s,
                       (const uint8_t *)d->d_reads.p, (uint64_t)bytes, L, (uint32_t *)d->d_flags.p);
    if (by_quality)
        hipLaunchKernelGGL(k_dv_quality, dim3(g), dim3(256), 0, s, (const uint8_t *)d->d_quals.p, n, L, p.simplified_suffix_mode ? 1 : 0,
                           d->suffix_pos, p.error_limit, (const float *)d->d_lut.p, (uint8_t *)d->d_high.p, (uint32_t *)d->d_flags.p);
    hipLaunchKernelGGL(k_dv_class, dim3(g), dim3(256), 0, s, (const uint32_t *)d->d_flags.p, (const uint8_t *)d->d_high.p, n, n_apart ? 1 : 0,
                       p.separate_n_reads_set ? 1 : 0, by_quality ? 1 : 0, (uint8_t *)d->d_cls.p, (uint32_t *)d->d_cnt[0].p,
                       (uint32_t *)d->d_cnt[1].p, (uint32_t *)d->d_cnt[2].p, (uint32_t *)d->d_err.p);
    DIV_TRY(d, hipGetLastError());
    for (int k = 0; k < 3; k++)
        if ((e = pgrc_ps_scan_u32(c, (uint32_t *)d->d_cnt[k].p, n + 1, (uint32_t *)d->d_bsum.p))) return e;
    uint32_t cnt[3] = {0, 0, 0}, bad = 0;
    for (int k = 0; k < 3; k++) DIV_TRY(d, hipMemcpyAsync(&cnt[k], (const uint32_t *)d->d_cnt[k].p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    DIV_TRY(d, hipMemcpyAsync(&bad, d->d_err.p, sizeof bad, hipMemcpyDeviceToHost, s));
    DIV_TRY(d, hipStreamSynchronize(s));
    if (bad & DVF_BAD) { c->err = "reads contain a symbol outside ACGNT"; return PGRC_E_SYMBOL; }
    if (bad & DVF_BAD_Q) { c->err = "a quality character lies beyond the reference's table (133 entries)"; return PGRC_E_SYMBOL; }
    for (int k = 0; k < 3; k++)
        if ((e = pgrc_buf_ensure(c, d->d_rows[k], (size_t)cnt[k] * rb[k] + 16))) return e;
    for (int k = 0; k < 2; k++)
        if ((e = pgrc_buf_ensure(c, d->d_idx[k], ((size_t)cnt[k + 1] + 1) * sizeof(uint32_t)))) return e;
    DvPackArgs a;
    a.reads = (const uint8_t *)d->d_reads.p;
    a.cls = (const uint8_t *)d->d_cls.p;
    for (int k = 0; k < 3; k++) {
        a.slot[k] = (const uint32_t *)d->d_cnt[k].p;
        a.rows[k] = (uint8_t *)d->d_rows[k].p;
        a.row_bytes[k] = rb[k];
        a.per[k] = per[k];
        a.base[k] = sym[k];
    }
    a.idx[0] = (uint32_t *)d->d_idx[0].p;
    a.idx[1] = (uint32_t *)d->d_idx[1].p;
    a.row_bytes_max = std::max(rb[0], std::max(rb[1], rb[2]));
    a.L = L;
    a.n = n;
    const uint64_t work = n * a.row_bytes_max;
    hipLaunchKernelGGL(k_dv_pack, dim3((uint32_t)((work + 255) / 256)), dim3(256), 0, s, a);
    DIV_TRY(d, hipGetLastError());
    (void)hipEventRecord(d->ev[2], s);
    out->n_hq = cnt[0]; out->n_lq = cnt[1]; out->n_n = cnt[2];
    for (int k = 0; k < 3; k++)
        if ((e = dv_host_ensure(d, d->h_rows[k], (size_t)cnt[k] * rb[k]))) return e;
    for (int k = 0; k < 2; k++)
        if ((e = dv_host_ensure(d, d->h_idx[k], (size_t)cnt[k + 1] * sizeof(uint32_t)))) return e;
    hipError_t he = hipSuccess;
    for (int k = 0; k < 3 && he == hipSuccess; k++)
        if (cnt[k] && rb[k]) he = hipMemcpyAsync(d->h_rows[k].p, d->d_rows[k].p, (size_t)cnt[k] * rb[k], hipMemcpyDeviceToHost, s);
    for (int k = 0; k < 2 && he == hipSuccess; k++)
        if (cnt[k + 1]) he = hipMemcpyAsync(d->h_idx[k].p, d->d_idx[k].p, (size_t)cnt[k + 1] * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipEventRecord(d->ev[3], s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) { c->err = std::string("divider: ") + hipGetErrorString(he); return pgrc_hip_code(he); }
    out->hq_rows = (const uint8_t *)d->h_rows[0].p;
    out->lq_rows = (const uint8_t *)d->h_rows[1].p;
    out->n_rows = (const uint8_t *)d->h_rows[2].p;
    out->lq_index = (const uint32_t *)d->h_idx[0].p;
    out->n_index = (const uint32_t *)d->h_idx[1].p;
    for (int k = 0; k < 3; k++) (void)hipEventElapsedTime(&d->ms[k], d->ev[k], d->ev[k + 1]);
    return PGRC_OK;
}

static int dv_events(pgrc_divider *d) {
    if (d->have_ev) return PGRC_OK;
    for (auto &x : d->ev) DIV_TRY(d, hipEventCreate(&x));
    d->have_ev = true;
    return PGRC_OK;
}

int pgrc_divider_run(pgrc_divider *d, const char *reads, const char *quals, uint64_t n, pgrc_divided_reads *out) {
    if (!d || !out || (n && !reads)) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    const bool by_quality = d->prm.error_limit < 1;
    if (by_quality && n && !quals) { d->base.err = "divider: quality rows are needed when error_limit < 1"; return PGRC_E_PARAM; }
    if (n >= (1ull << 32) - 1) { d->base.err = "divider: batches of less than 2^32 - 1 reads"; return PGRC_E_PARAM; }
    pgrc_match_ctx *c = &d->base;
    PgrcDeviceScope scope(c->device);
    if (!scope.ok) { c->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
    int e;
    if ((e = dv_events(d))) return e;
    const size_t bytes = (size_t)n * d->prm.read_len;
    if (n) {
        if ((e = pgrc_buf_ensure(c, d->d_reads, bytes + 16)) || (by_quality && (e = pgrc_buf_ensure(c, d->d_quals, bytes + 16)))) return e;
        (void)hipEventRecord(d->ev[0], c->stream);
        DIV_TRY(d, hipMemcpyAsync(d->d_reads.p, reads, bytes, hipMemcpyHostToDevice, c->stream));
        if (by_quality) DIV_TRY(d, hipMemcpyAsync(d->d_quals.p, quals, bytes, hipMemcpyHostToDevice, c->stream));
        (void)hipEventRecord(d->ev[1], c->stream);
    }
    return divide_resident(d, n, out);
}

// ---- FASTQ text in, sets out (FASTQReadsSourceIterator, readsset/iterator/ReadsSetIterator.cpp:189-224; the pairing and the
// reverse complement of the second file's reads, RevComplPairReadsSetIterator, :256-284; ManagedReadsSetIterator,
// readsset/persistance/ReadsSetPersistence.cpp:20-56).  The reference reads four lines per record with std::getline --
// identifier, symbols, '+' line, qualities --, alternately from the two files of a pair; the read is the leading run of
// letters of the symbol line, the quality string is cut or zero-padded to that length.  Here the caller hands over a piece
// of each file's text; the lines are found by all threads (newlines counted per 16 bytes, an exclusive scan, line starts
// written in place), the records' rows are copied out byte-parallel -- every second record of a pair reverse-complemented
// when asked -- and the division above runs on them where they are: the text goes up once, nothing else.
int pgrc_divider_run_fastq(pgrc_divider *d, const char *text, uint64_t bytes, const char *pair_text, uint64_t pair_bytes,
                           int32_t rev_compl_pair, int32_t final_piece, uint64_t *consumed, uint64_t *pair_consumed,
                           uint64_t *n_records, pgrc_divided_reads *out) {
    if (!d || !out || !consumed || !n_records || (bytes && !text) || (pair_bytes && !pair_text) || (pair_text && !pair_consumed)) return PGRC_E_PARAM;
    memset(out, 0, sizeof *out);
    *consumed = 0;
    *n_records = 0;
    if (pair_consumed) *pair_consumed = 0;
    pgrc_match_ctx *c = &d->base;
    if (bytes >= (1ull << 31) || pair_bytes >= (1ull << 31)) { c->err = "divider: pieces of FASTQ text below 2 GiB"; return PGRC_E_PARAM; }
    PgrcDeviceScope scope(c->device);
    if (!scope.ok) { c->err = "hipSetDevice failed"; return PGRC_E_NO_DEVICE; }
    int e;
    if ((e = dv_events(d))) return e;
    const uint32_t L = d->prm.read_len;
    const bool paired = pair_text != nullptr;
    const bool by_quality = d->prm.error_limit < 1;
    hipStream_t s = c->stream;
    const char *src[2] = {text, pair_text};
    const uint64_t len[2] = {bytes, pair_bytes};
    // final_piece: bit 0 = nothing follows in either text; bit 1 / bit 2 = the first / the second text ends here (the other may go on)
    const bool fin[2] = {(final_piece & 3) != 0, (final_piece & 5) != 0};
    uint64_t lines[2] = {0, 0};
    uint32_t nls[2] = {0, 0};
    (void)hipEventRecord(d->ev[0], s);
    for (int f = 0; f < (paired ? 2 : 1); f++) {
        const uint64_t n16 = (len[f] + 15) / 16;
        if ((e = pgrc_buf_ensure(c, d->d_text[f], len[f] + 32)) || (e = pgrc_buf_ensure(c, d->d_nl[f], (n16 + 1) * sizeof(uint32_t))) ||
            (e = pgrc_buf_ensure(c, d->d_bsum, (pgrc_ps_scan_blocks(n16 + 1) + 1) * sizeof(uint32_t))))
            return e;
        if (len[f]) DIV_TRY(d, hipMemcpyAsync(d->d_text[f].p, src[f], len[f], hipMemcpyHostToDevice, s));
        DIV_TRY(d, hipMemsetAsync((uint8_t *)d->d_text[f].p + len[f], 0, 32, s));
        hipLaunchKernelGGL(k_fq_count, dim3((uint32_t)((n16 + 1 + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d->d_text[f].p, len[f], n16,
                           (uint32_t *)d->d_nl[f].p);
        if ((e = pgrc_ps_scan_u32(c, (uint32_t *)d->d_nl[f].p, n16 + 1, (uint32_t *)d->d_bsum.p))) return e;
        uint32_t nl = 0;
        uint8_t last = '\n';
        DIV_TRY(d, hipMemcpyAsync(&nl, (const uint32_t *)d->d_nl[f].p + n16, sizeof nl, hipMemcpyDeviceToHost, s));
        DIV_TRY(d, hipStreamSynchronize(s));
        if (len[f]) last = (uint8_t)src[f][len[f] - 1];
        // std::getline: a last piece without its newline is a line too, once nothing more will come
        lines[f] = nl + ((fin[f] && len[f] && last != '\n') ? 1u : 0u);
        nls[f] = nl;
        if ((e = pgrc_buf_ensure(c, d->d_ls[f], ((size_t)nl + 8) * sizeof(uint32_t)))) return e;
        hipLaunchKernelGGL(k_fq_starts, dim3((uint32_t)((n16 + 255) / 256 + 1)), dim3(256), 0, s, (const uint8_t *)d->d_text[f].p, len[f], n16,
                           (const uint32_t *)d->d_nl[f].p, nl, (uint32_t *)d->d_ls[f].p);
        DIV_TRY(d, hipGetLastError());
    }
    // whole records in what we have: four lines each
    const uint64_t rec[2] = {lines[0] / 4, lines[1] / 4};
    uint64_t n;                       // records taken: the two files in turn, first file first, until one of them has no more
    uint64_t take[2];
    // `terminal`: the reference's iteration ends with this call -- a source is through when its turn comes (it stops at the
    // first exhausted file).  One text may end here while the other goes on (final_piece bits 2 / 4): the pairs its records
    // form are taken as soon as the other piece holds their mates, whatever lies beyond them there.
    bool terminal;
    if (!paired) { n = rec[0]; take[0] = rec[0]; take[1] = 0; terminal = fin[0]; }
    else if ((fin[0] && fin[1]) || (fin[0] && rec[1] >= rec[0]) || (fin[1] && rec[0] >= rec[1] + 1)) {
        n = rec[0] <= rec[1] ? 2 * rec[0] : 2 * rec[1] + 1; take[0] = (n + 1) / 2; take[1] = n / 2; terminal = true;
    } else { const uint64_t m = std::min(rec[0], rec[1]); n = 2 * m; take[0] = take[1] = m; terminal = false; }
    d->last_terminal = terminal;
    if (terminal && fin[paired ? (int)(n & 1) : 0]) {
        // The record the reference would read next: none (its source is through: it stops) -- or the beginning of one.  What it
        // makes of a cut-off last record depends on state left over in its iterator (a std::getline on a stream already at
        // its end leaves the string of the record BEFORE in place): not reproduced, reported instead.
        const int f = paired ? (int)(n & 1) : 0;
        if (lines[f] % 4 && rec[f] == (paired ? n >> 1 : n)) {
            c->err = "FASTQ text ends inside a record";
            return PGRC_E_PARAM;
        }
    }
    if (n >= (1ull << 32) - 1) { c->err = "divider: batches of less than 2^32 - 1 reads"; return PGRC_E_PARAM; }
    const size_t rows = (size_t)n * L;
    if ((e = pgrc_buf_ensure(c, d->d_reads, rows + 16)) || (by_quality && (e = pgrc_buf_ensure(c, d->d_quals, rows + 16)))) return e;
    DIV_TRY(d, hipMemsetAsync(d->d_err.p, 0, sizeof(uint32_t), s));
    uint32_t ends[2] = {0, 0};        // first byte after the last record taken from each piece
    if (n) {
        FqRowsArgs a;
        for (int f = 0; f < 2; f++) {
            a.text[f] = (const uint8_t *)d->d_text[f].p;
            a.ls[f] = (const uint32_t *)d->d_ls[f].p;
            a.len[f] = (uint32_t)len[f];
            a.nl[f] = nls[f];
        }
        a.paired = paired ? 1u : 0u;
        a.rc_second = (paired && rev_compl_pair) ? 1u : 0u;
        a.L = L;
        a.n = n;
        a.reads = (uint8_t *)d->d_reads.p;
        a.quals = by_quality ? (uint8_t *)d->d_quals.p : nullptr;
        a.err = (uint32_t *)d->d_err.p;
        const uint64_t work = n * (L + 1);
        hipLaunchKernelGGL(k_fq_rows, dim3((uint32_t)((work + 255) / 256)), dim3(256), 0, s, a);
        DIV_TRY(d, hipGetLastError());
        uint32_t bad = 0;
        DIV_TRY(d, hipMemcpyAsync(&bad, d->d_err.p, sizeof bad, hipMemcpyDeviceToHost, s));
        for (int f = 0; f < (paired ? 2 : 1); f++)         // line 4 * take starts where the records taken end (ls[nl + 1] = the end of the text)
            if (take[f]) DIV_TRY(d, hipMemcpyAsync(&ends[f], (const uint32_t *)d->d_ls[f].p + 4 * take[f], sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        DIV_TRY(d, hipStreamSynchronize(s));
        if (bad) { c->err = "Unsupported variable length reads (a FASTQ record whose read is not read_len letters long)"; return PGRC_E_PARAM; }
    }
    *consumed = ends[0];
    if (paired) *pair_consumed = ends[1];
    (void)hipEventRecord(d->ev[1], s);
    *n_records = n;
    return divide_resident(d, n, out);
}

} // extern "C"
