// pack.hip -- boundary kernels: ASCII -> 2-bit packing of the pseudogenome and the reads,
// reverse complement of the packed pseudogenome, repack of the reference's own read packing.
//
// Replaces, on the device: PgHelpers::reverseComplementInPlace (utils/helper.cpp:383-393, the
// two in-place RC sweeps of ReadsMatchers.cpp:167-171) and the per-read unpack
// readsSet->getRead() (ReadsMatchers.cpp:432 -> SymbolsPackingFacility::reverseSequence,
// coders/SymbolsPackingFacility.cpp:216-236).  All are single-pass HBM-bound streams.
#include "ctx.h"
#include "devutil.h"

// A0 C1 G2 T3 from ASCII: x = (c>>1)&3 gives A0 C1 T2 G3, x ^= x>>1 swaps the last two.
__device__ __forceinline__ uint32_t ascii2code(uint32_t c) {
    uint32_t x = (c >> 1) & 3u;
    return x ^ (x >> 1);
}
__device__ __forceinline__ bool is_acgt(uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// one thread = one output word = 16 ASCII symbols (one 16-B load)
__global__ void __launch_bounds__(256) k_pack_ascii(const uint8_t *__restrict__ ascii, uint64_t count,
                                                    uint32_t *__restrict__ words, uint32_t *errflag) {
    const uint64_t nwords = (count + 15) / 16;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
         w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = w * 16;
        uint32_t out = 0;
        bool bad = false;
        if (base + 16 <= count) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ascii + base);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t c = (q[k] >> (8 * b)) & 0xFFu;
                    bad |= !is_acgt(c);
                    out |= ascii2code(c) << (2 * (4 * k + b));
                }
        } else {
            for (uint32_t k = 0; base + k < count; k++) {
                uint32_t c = ascii[base + k];
                bad |= !is_acgt(c);
                out |= ascii2code(c) << (2 * k);
            }
        }
        words[w] = out;
        if (bad) atomicOr(errflag, 1u);
    }
}

int pgrc_launch_pack_ascii(pgrc_match_ctx *c, const uint8_t *d_ascii, uint64_t count, uint32_t *d_words,
                           uint32_t *d_errflag) {
    if (count == 0) return PGRC_OK;
    uint64_t nwords = (count + 15) / 16;
    uint32_t grid = (uint32_t)((nwords + 255) / 256 < 65536 * 4 ? (nwords + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_pack_ascii, dim3(grid), dim3(256), 0, c->stream, d_ascii, count, d_words, d_errflag);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// rc[i] = 3 - fw[G-1-i]; one thread per output word
__global__ void __launch_bounds__(256) k_revcomp(const uint32_t *__restrict__ fw, uint32_t *__restrict__ rc,
                                                 uint64_t G, uint64_t nwords) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
         w += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t s = (int64_t)G - 16 * (int64_t)w - 16; // first source symbol of this word
        uint32_t out;
        if (s >= 0) {
            const uint64_t q = (uint64_t)s >> 4;
            const uint32_t sh = ((uint32_t)s & 15u) * 2u;
            out = revcomp_word(funnel_r(fw[q], fw[q + 1], sh));
        } else {
            const uint32_t cnt = (uint32_t)(16 + s); // valid symbols in this (last) word, 1..15
            out = revcomp_word(fw[0] << (2 * (16 - cnt))) & ((1u << (2 * cnt)) - 1u);
        }
        rc[w] = out;
    }
}

int pgrc_launch_revcomp(pgrc_match_ctx *c, const uint32_t *d_fw, uint32_t *d_rc, uint64_t G) {
    uint64_t nwords = (G + 15) / 16;
    if (!nwords) return PGRC_OK;
    uint32_t grid = (uint32_t)((nwords + 255) / 256 < 65536 * 4 ? (nwords + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_revcomp, dim3(grid), dim3(256), 0, c->stream, d_fw, d_rc, G, nwords);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// reads: ASCII rows -> word-major 2-bit layout.  Thread t = w*count + i handles word w of read i:
// stores are coalesced along i.  'N' marks the read for the byte path; any other symbol is an error.
__global__ void __launch_bounds__(256)
k_pack_reads_ascii(const uint8_t *__restrict__ ascii, uint64_t first, uint64_t count, uint32_t L, uint32_t nw,
                   uint32_t *__restrict__ words, uint64_t stride, uint8_t *__restrict__ nflag, uint32_t *errflag) {
    const uint64_t total = count * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t % count;
        const uint32_t w = (uint32_t)(t / count);
        const uint8_t *row = ascii + i * L;
        uint32_t out = 0;
        bool bad = false, hasn = false;
        const uint32_t lim = (16 * w + 16 <= L) ? 16 : L - 16 * w;
        for (uint32_t k = 0; k < lim; k++) {
            uint32_t ch = row[16 * w + k];
            if (ch == 'N') hasn = true;
            else if (!is_acgt(ch)) bad = true;
            else out |= ascii2code(ch) << (2 * k);
        }
        words[(uint64_t)w * stride + first + i] = out;
        if (hasn) nflag[first + i] = 1;
        if (bad) atomicOr(errflag, 1u);
    }
}

int pgrc_launch_pack_reads_ascii(pgrc_match_ctx *c, const uint8_t *d_ascii, uint64_t first, uint64_t count,
                                 uint32_t L, uint32_t *d_words, uint64_t stride, uint8_t *d_nflag,
                                 uint32_t *d_errflag) {
    if (!count) return PGRC_OK;
    uint32_t nw = (L + 15) / 16;
    uint64_t total = count * nw;
    uint32_t grid = (uint32_t)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_pack_reads_ascii, dim3(grid), dim3(256), 0, c->stream, d_ascii, first, count, L, nw,
                       d_words, stride, d_nflag, d_errflag);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// The reference's ACGT packing (4 symbols per byte, first symbol most significant,
// SymbolsPackingFacility.cpp:143-178) -> our little-endian 2-bit words: reverse the bit pairs
// of every byte.
__device__ __forceinline__ uint32_t pair_reverse_bytes(uint32_t v) {
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    return v;
}

__global__ void __launch_bounds__(256)
k_repack_reads_ref(const uint8_t *__restrict__ packed, uint64_t first, uint64_t count, uint32_t L, uint32_t nw,
                   uint32_t pb, uint32_t *__restrict__ words, uint64_t stride) {
    const uint64_t total = count * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t % count;
        const uint32_t w = (uint32_t)(t / count);
        const uint8_t *row = packed + i * pb;
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4 && 4 * w + b < pb; b++) v |= (uint32_t)row[4 * w + b] << (8 * b);
        v = pair_reverse_bytes(v);
        v &= (16 * w + 16 <= L) ? 0xFFFFFFFFu : ((1u << (2 * (L - 16 * w))) - 1u);
        words[(uint64_t)w * stride + first + i] = v;
    }
}

int pgrc_launch_repack_reads_ref(pgrc_match_ctx *c, const uint8_t *d_packed, uint64_t first, uint64_t count,
                                 uint32_t L, uint32_t *d_words, uint64_t stride) {
    if (!count) return PGRC_OK;
    uint32_t nw = (L + 15) / 16, pb = (L + 3) / 4;
    uint64_t total = count * nw;
    uint32_t grid = (uint32_t)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_repack_reads_ref, dim3(grid), dim3(256), 0, c->stream, d_packed, first, count, L, nw, pb,
                       d_words, stride);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// The reference's ACGNT packing (the N read set, readsset/DividedPCLReadsSets.cpp:16-19): 3 symbols per byte as base-5
// digits, first symbol most significant, order A0 C1 G2 N3 T4 (SymbolsPackingFacility.cpp:133-178; a partial last
// byte is padded with digit 0).  lut[v]: bits 0-5 the three 2-bit codes (N packs as 0), bits 8-10 the N flags,
// bit 15 = not a code (v >= 125).
__device__ __forceinline__ void acgnt_lut_init(uint32_t *lut) {
    for (uint32_t v = threadIdx.x; v < 256; v += blockDim.x) {
        uint32_t out = 0;
        if (v >= 125) out = 1u << 15;
        else {
            const uint32_t d[3] = {v / 25u, (v / 5u) % 5u, v % 5u};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (d[k] == 3u) out |= 1u << (8 + k);
                else out |= (d[k] == 4u ? 3u : d[k]) << (2 * k);
            }
        }
        lut[v] = out;
    }
}

// thread t = w * count + i: word w of read i (stores coalesced along i)
__global__ void __launch_bounds__(256)
k_unpack_reads_acgnt(const uint8_t *__restrict__ packed, uint64_t first, uint64_t count, uint32_t L, uint32_t nw, uint32_t pb,
                     uint32_t *__restrict__ words, uint64_t stride, uint8_t *__restrict__ nflag, uint32_t *errflag) {
    __shared__ uint32_t lut[256];
    acgnt_lut_init(lut);
    __syncthreads();
    const uint64_t total = count * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t % count;
        const uint32_t w = (uint32_t)(t / count);
        const uint8_t *row = packed + i * pb;
        const uint32_t x0 = 16 * w, x1 = min(L, x0 + 16);
        uint32_t out = 0;
        bool hasn = false, bad = false;
        for (uint32_t b = x0 / 3; 3 * b < x1; b++) {            // the (at most 7) bytes that overlap symbols [x0, x1)
            const uint32_t e = lut[row[b]];
            bad |= (e >> 15) != 0;
#pragma unroll
            for (uint32_t k = 0; k < 3; k++) {
                const uint32_t x = 3 * b + k;
                if (x >= x0 && x < x1) {
                    out |= ((e >> (2 * k)) & 3u) << (2 * (x - x0));
                    hasn |= ((e >> (8 + k)) & 1u) != 0;
                }
            }
        }
        words[(uint64_t)w * stride + first + i] = out;
        if (hasn) nflag[first + i] = 1;
        if (bad) atomicOr(errflag, 1u);
    }
}

int pgrc_launch_unpack_reads_acgnt(pgrc_match_ctx *c, const uint8_t *d_packed, uint64_t first, uint64_t count, uint32_t L,
                                   uint32_t *d_words, uint64_t stride, uint8_t *d_nflag, uint32_t *d_errflag) {
    if (!count) return PGRC_OK;
    const uint32_t nw = (L + 15) / 16, pb = (L + 2) / 3;
    const uint64_t total = count * nw;
    const uint32_t grid = (uint32_t)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_unpack_reads_acgnt, dim3(grid), dim3(256), 0, c->stream, d_packed, first, count, L, nw, pb, d_words,
                       stride, d_nflag, d_errflag);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// ASCII rows of selected ACGNT-packed reads (the reads with N: the side list keeps them as bytes); one thread per symbol
__global__ void __launch_bounds__(256)
k_nrows_ascii_acgnt(const uint8_t *__restrict__ packed, const uint32_t *__restrict__ local_idx, uint64_t count, uint32_t L,
                    uint32_t pb, uint8_t *__restrict__ ascii) {
    const uint64_t total = count * L;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / L;
        const uint32_t x = (uint32_t)(t % L);
        const uint32_t v = packed[(uint64_t)local_idx[r] * pb + x / 3];
        const uint32_t d = x % 3 == 0 ? v / 25u : x % 3 == 1 ? (v / 5u) % 5u : v % 5u;
        ascii[t] = (uint8_t)"ACGNT"[d < 4u ? d : 4u];   // (a byte outside the code range is reported by the unpack kernel)
    }
}

int pgrc_launch_nrows_ascii_acgnt(pgrc_match_ctx *c, const uint8_t *d_packed, const uint32_t *d_local_idx, uint64_t count,
                                  uint32_t L, uint8_t *d_ascii) {
    if (!count) return PGRC_OK;
    const uint64_t total = count * L;
    const uint32_t grid = (uint32_t)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_nrows_ascii_acgnt, dim3(grid), dim3(256), 0, c->stream, d_packed, d_local_idx, count, L, (L + 2) / 3,
                       d_ascii);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}

// Where the N's of the reads flagged by the kernels above are.  One thread per read of the block; a flagged read with at
// most 4 N's gets their positions as four bytes (0xFF = none; a read has at most 255 symbols) in npos[read] and the flag 3:
// the dual kernel (copmem.hip) takes such a read like any other -- it patches the window hash where an N falls into a
// window (the reference hashes the byte 'N') and counts every N as a mismatch.  More N's: the flag stays 1 and the read
// goes the byte path (k_copmem_match_n).  rows: the block's rows as uploaded, ASCII (symbols = 0) or ACGNT-packed (5).
__global__ void __launch_bounds__(256)
k_npos_rows(const uint8_t *__restrict__ rows, int symbols, uint64_t first, uint64_t count, uint32_t L, uint32_t rb,
            uint8_t *__restrict__ nflag, uint32_t *__restrict__ npos) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!nflag[first + i]) continue;
        const uint8_t *row = rows + i * rb;
        uint32_t w = 0xFFFFFFFFu, cnt = 0;
        for (uint32_t x = 0; x < L; x++) {
            bool isn;
            if (symbols == 0) isn = row[x] == 'N';
            else {
                const uint32_t v = row[x / 3];
                isn = (x % 3 == 0 ? v / 25u : x % 3 == 1 ? (v / 5u) % 5u : v % 5u) == 3u;     // A0 C1 G2 N3 T4
            }
            if (isn) {
                if (cnt < 4) w = (w & ~(0xFFu << (8 * cnt))) | (x << (8 * cnt));
                cnt++;
            }
        }
        if (cnt >= 1 && cnt <= 4) {
            npos[first + i] = w;
            nflag[first + i] = 3;
        }
    }
}

int pgrc_launch_npos_rows(pgrc_match_ctx *c, const uint8_t *d_rows, int symbols, uint64_t first, uint64_t count, uint32_t L,
                          uint8_t *d_nflag, uint32_t *d_npos) {
    if (!count) return PGRC_OK;
    if (L > 255 || !d_npos) { c->err = "npos_rows: a read position must fit one byte (read_len <= 255)"; return PGRC_E_PARAM; }
    const uint32_t rb = symbols == 0 ? L : (L + 2) / 3;
    const uint32_t grid = (uint32_t)((count + 255) / 256 < 65536 ? (count + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_npos_rows, dim3(grid), dim3(256), 0, c->stream, d_rows, symbols, first, count, L, rb, d_nflag, d_npos);
    HIP_TRY(c, hipGetLastError());
    return PGRC_OK;
}
