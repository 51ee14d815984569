// matchdev.h -- device-side pieces shared by the match kernels of mode c (copmem.hip; the A/B builds under tools/variants/).
#pragma once

#include <type_traits>

#include "devutil.h"
#include "headfmt.h"

#define MATCH_TPB 256

// Per-read state of the reference's sequential query (CopMEMMatcher.cpp:483-566).
template <typename pos_t>
struct ReadState {
    uint32_t limit, falses, cur;
    pos_t best; // all ones = none (text positions stay below 2^32 - 256, resp. 2^40 - 256: api.hip alloc_pg)
    bool done;
};

#define SM_MAX_SEEDS 240
#ifndef VC_BITS
#define VC_BITS 2           // verify-cache slots per read = 1 << VC_BITS (direct mapped)
#endif
#define VC_SLOTS (1 << VC_BITS)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U32x4A4 { u32x4 v; }; // 16-B load that only needs 4-B alignment
struct __attribute__((packed, aligned(8))) U64x2A8 { unsigned long long x, y; }; // 16-B load, 8-B aligned
// KQ = K/4 when known at compile time (7 for the default seed 38), 0 = run-time loop.
template <int KQ>
__device__ __forceinline__ uint32_t hash_fp_window(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t w3,
                                                   uint32_t K, const uint32_t *lut, uint32_t *fp_out) {
    if (KQ == 0) return copmem_hash32_fp(w0, w1, w2, w3, K, lut, fp_out);
    const uint32_t w[4] = {w0, w1, w2, w3};
    uint32_t h = 4u * KQ, fp = 0, fb = 0;
#pragma unroll
    for (int j = 0; j < KQ; j++) {
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;   // static register, static shift
        const uint32_t x = (j < 3) ? lut[b & 63u] : lut[64u + (b & 15u)];
        h = (h ^ (x + (uint32_t)j)) * 171717u;
        const uint32_t width = (j < 3) ? 2u : 4u;
        if (fb + width <= PGRC_FP_BITS) {
            fp |= ((j < 3) ? (b >> 6) : (b >> 4)) << fb;
            fb += width;
        }
    }
    *fp_out = fp;
    return h;
}

// The same for a window of a read that holds N's (npw: up to four read positions, one per byte, 0xFF = none; s = the
// window's first read position).  The packed read carries code 0 = 'A' (0x41) where the read has an N, the reference
// hashes the byte 'N' (0x4E): every N that falls on a hashed byte of step j adds 0x0D to that byte of the step's word --
// no carry leaves the byte, so the patched word is exactly the reference's.  The fingerprint keeps the packed codes: a
// difference it counts at an N position is a real mismatch (an N equals nothing), one it misses only weakens the lower
// bound, which stays a lower bound.
template <int KQ>
__device__ __forceinline__ uint32_t hash_fp_window_n(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t w3,
                                                     uint32_t K, const uint32_t *lut, uint32_t *fp_out, uint32_t npw, uint32_t s) {
    uint32_t qj[4], inc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = ((npw >> (8 * i)) & 0xFFu) - s;          // (none = 0xFF: 255 - s >= K, since s + K <= L <= 255)
        const uint32_t j = q >> 2, b = q & 3u;
        const bool hashed = q < K && b < (j < 3u ? 3u : 2u);
        qj[i] = hashed ? j : 0xFFFFFFFFu;
        inc[i] = 0x0Du << (8u * b);
    }
    const uint32_t w[4] = {w0, w1, w2, w3};
    const uint32_t kq = KQ ? (uint32_t)KQ : (K >> 2);
    uint32_t h = 4u * kq, fp = 0, fb = 0;
#pragma unroll
    for (uint32_t j = 0; j < (KQ ? (uint32_t)KQ : 14u); j++) {
        if (!KQ && j >= kq) break;
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        uint32_t x = (j < 3) ? lut[b & 63u] : lut[64u + (b & 15u)];
#pragma unroll
        for (int i = 0; i < 4; i++) x += (qj[i] == j) ? inc[i] : 0u;
        h = (h ^ (x + j)) * 171717u;
        const uint32_t width = (j < 3) ? 2u : 4u;
        if (fb + width <= PGRC_FP_BITS) {
            fp |= ((j < 3) ? (b >> 6) : (b >> 4)) << fb;
            fb += width;
        }
    }
    *fp_out = fp;
    return h;
}

#ifndef MATCH_STAGE
#define MATCH_STAGE 32      // reads a wave stages in LDS per burst of full-line loads (k_copmem_match_sm, STAGE)
#endif
#ifndef MATCH_CHUNK
#define MATCH_CHUNK 1024u // reads a wave reserves per visit to the global work counter (256 / 512 / 1024: step +0 / -0.2 / -0.4 %)
#endif

