// synth.hip -- synthetic pseudogenome / read-set generators (include/pgrc_synth.h): the same pure
// functions run as host loops (tests, golden fixtures, CPU baseline) and as HIP kernels that fill
// HBM directly in the library's packed layouts (bench.py: no PCIe traffic in the workload setup).
#include "ctx.h"
#include "devutil.h"

extern "C" void pgrc_synth_pg_host(const pgrc_synth_pg *g, char *out) {
    for (uint64_t i = 0; i < g->pg_len; i++) out[i] = "ACGT"[pgrc_synth_pg_base(g, i)];
}

static uint32_t host_code(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }

extern "C" void pgrc_synth_reads_host(const pgrc_synth_pg *g, const char *pg_ascii, const pgrc_synth_reads *rs,
                                      uint64_t first_read, uint64_t count, char *out) {
    const uint32_t L = rs->read_len;
    for (uint64_t r = 0; r < count; r++) {
        const uint64_t j = first_read + r;
        char *dst = out + r * L;
        uint8_t codes[256];
        pgrc_synth_read_hdr h = pgrc_synth_read_header(g, rs, j);
        if (h.random || !pg_ascii) {
            pgrc_synth_read_codes(g, rs, j, codes);
        } else {
            // same result as pgrc_synth_read_codes, but sourcing the bases from the generated text
            if (!h.rc) for (uint32_t k = 0; k < L; k++) codes[k] = (uint8_t)host_code(pg_ascii[h.start + k]);
            else for (uint32_t k = 0; k < L; k++) codes[k] = (uint8_t)(3u - host_code(pg_ascii[h.start + (L - 1 - k)]));
            for (uint32_t q = 0; q < h.nsub; q++) {
                uint64_t w = pgrc_rnd(rs->seed, PGRC_S_SUB, j * 8 + q);
                uint32_t p = (uint32_t)(w % L);
                uint32_t d = 1 + (uint32_t)((w >> 32) % 3);
                codes[p] = (uint8_t)((codes[p] + d) & 3u);
            }
        }
        for (uint32_t k = 0; k < L; k++) dst[k] = "ACGT"[codes[k]];
        const uint32_t nn = pgrc_synth_read_n_count(rs, j);
        for (uint32_t q = 0; q < nn; q++) dst[pgrc_synth_read_n_pos(rs, j, q)] = 'N';
    }
}

__global__ void __launch_bounds__(256) k_synth_pg(const pgrc_synth_pg g, uint32_t *__restrict__ words, uint64_t nwords) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t out = 0;
        for (uint32_t k = 0; k < 16; k++) {
            const uint64_t i = w * 16 + k;
            if (i < g.pg_len) out |= pgrc_synth_pg_base(&g, i) << (2 * k);
        }
        words[w] = out;
    }
}

extern "C" int pgrc_synth_pg_device(const pgrc_synth_pg *g, void *d_words_out, void *hip_stream) {
    const uint64_t nwords = (g->pg_len + 15) / 16;
    if (!nwords) return PGRC_OK;
    uint32_t grid = (uint32_t)((nwords + 255) / 256 < 65536 * 4 ? (nwords + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(k_synth_pg, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, *g, (uint32_t *)d_words_out, nwords);
    return hipGetLastError() == hipSuccess ? PGRC_OK : PGRC_E_DEVICE;
}

// one read per thread; bases gathered from the packed Pg already in HBM
__global__ void __launch_bounds__(256)
k_synth_reads(const pgrc_synth_pg g, const uint32_t *__restrict__ pg, const pgrc_synth_reads rs, uint64_t first,
              uint64_t count, uint32_t *__restrict__ words, uint64_t stride) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const uint64_t j = first + r;
    const uint32_t L = rs.read_len, nw = (L + 15) / 16;
    uint8_t codes[256];
    pgrc_synth_read_hdr h = pgrc_synth_read_header(&g, &rs, j);
    if (h.random) {
        pgrc_synth_read_codes(&g, &rs, j, codes);
    } else {
        for (uint32_t k = 0; k < L; k++) {
            const uint64_t x = h.rc ? h.start + (L - 1 - k) : h.start + k;
            uint32_t c = (pg[x >> 4] >> (2u * ((uint32_t)x & 15u))) & 3u;
            codes[k] = (uint8_t)(h.rc ? 3u - c : c);
        }
        for (uint32_t q = 0; q < h.nsub; q++) {
            uint64_t w = pgrc_rnd(rs.seed, PGRC_S_SUB, j * 8 + q);
            uint32_t p = (uint32_t)(w % L);
            uint32_t d = 1 + (uint32_t)((w >> 32) % 3);
            codes[p] = (uint8_t)((codes[p] + d) & 3u);
        }
    }
    for (uint32_t w = 0; w < nw; w++) {
        uint32_t out = 0;
        for (uint32_t k = 0; k < 16 && 16 * w + k < L; k++) out |= (uint32_t)codes[16 * w + k] << (2 * k);
        words[(uint64_t)w * stride + r] = out;
    }
}

extern "C" int pgrc_synth_reads_device(const pgrc_synth_pg *g, const void *d_pg_words, const pgrc_synth_reads *rs,
                                       uint64_t first_read, uint64_t count, void *d_words_out, uint64_t stride,
                                       void *hip_stream) {
    if (!count) return PGRC_OK;
    if (rs->read_len > 255 || rs->read_len == 0 || g->pg_len < rs->read_len) return PGRC_E_PARAM;
    const uint32_t grid = (uint32_t)((count + 255) / 256);
    hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, *g, (const uint32_t *)d_pg_words,
                       *rs, first_read, count, (uint32_t *)d_words_out, stride);
    return hipGetLastError() == hipSuccess ? PGRC_OK : PGRC_E_DEVICE;
}
