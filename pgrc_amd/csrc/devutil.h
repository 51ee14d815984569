// devutil.h -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PGRC_WAVE 64

// (hi:lo) >> sh, low 32 bits; sh in [0,31]  (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel_r(uint32_t lo, uint32_t hi, uint32_t sh) {
    return __funnelshift_r(lo, hi, sh);
}

// ASCII of a 2-bit code: "ACGT" packed in one constant
__device__ __forceinline__ uint32_t code2ascii(uint32_t c) { return (0x54474341u >> (8u * c)) & 0xFFu; }

// LUTs for maRushPrime1HashSparsified over 2-bit text (Hashes.h:54-76 reads 4 ASCII bytes per
// step = exactly one byte of 2-bit text; it keeps 3 symbols for steps 0..2 and 2 afterwards):
//   lut[0..63]  : low 6 bits (3 symbols) -> the 3 ASCII bytes  (mask 0x00FFFFFF)
//   lut[64..79] : low 4 bits (2 symbols) -> the 2 ASCII bytes  (mask 0x0000FFFF)
#define PGRC_HASH_LUT_WORDS 80
__device__ __forceinline__ void hash_lut_init(uint32_t *lut) {
    for (uint32_t t = threadIdx.x; t < PGRC_HASH_LUT_WORDS; t += blockDim.x) {
        uint32_t v;
        if (t < 64) v = code2ascii(t & 3) | (code2ascii((t >> 2) & 3) << 8) | (code2ascii((t >> 4) & 3) << 16);
        else v = code2ascii(t & 3) | (code2ascii((t >> 2) & 3) << 8);
        lut[t] = v;
    }
}

// Low 32 bits of the reference's u64 multiply-xor fold (Hashes.h:54-76): the final cast to u32 only ever sees
// the low word, and xor / multiply keep the low word closed, so 32-bit arithmetic is exact.
// t0..t3 hold the K-symbol window (2 bits/symbol, little endian); kq = K/4 steps.
//
// The fold additionally returns a FINGERPRINT of the window: the symbols the sparsified hash
// ignores (symbol 3 of steps 0..2, symbols 2,3 of the later steps), packed 2 bits each in step
// order, at most 22 bits (exactly what K = 28 yields; entries keep 40 bits for the position).  Stored next to every indexed position, it lets the match kernel reject a
// false candidate without touching the pseudogenome: the fingerprint symbols are ordinary window
// symbols, so their mismatches are a lower bound of the Hamming distance.
#define PGRC_FP_BITS 22u
__device__ __forceinline__ uint32_t copmem_hash32_fp(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3,
                                                     uint32_t K, const uint32_t *lut, uint32_t *fp_out) {
    uint32_t h = K, fp = 0, fb = 0;
    const uint32_t kq = K >> 2;
    for (uint32_t j = 0; j < kq; j++) {
        uint32_t b = t0 & 0xFFu;
        uint32_t w = (j < 3) ? lut[b & 63u] : lut[64u + (b & 15u)];
        h = (h ^ (w + j)) * 171717u;
        const uint32_t width = (j < 3) ? 2u : 4u;
        if (fb + width <= PGRC_FP_BITS) {
            fp |= ((j < 3) ? (b >> 6) : (b >> 4)) << fb;
            fb += width;
        }
        t0 = funnel_r(t0, t1, 8);
        t1 = funnel_r(t1, t2, 8);
        t2 = funnel_r(t2, t3, 8);
        t3 >>= 8;
    }
    *fp_out = fp;
    return h;
}

// Per-symbol mask (bits at even positions) of the fingerprint symbols whose READ offset s+u lies
// below `head` -- i.e. those that belong to the head part of the reference's two-stage Hamming
// count (CopMEMMatcher.cpp:495, :528-551).  Uniform per seed offset s.
__device__ __forceinline__ uint32_t fp_head_mask(uint32_t K, uint32_t s, uint32_t head) {
    uint32_t m = 0, fb = 0;
    const uint32_t kq = K >> 2;
    for (uint32_t j = 0; j < kq; j++) {
        const uint32_t width = (j < 3) ? 2u : 4u;
        if (fb + width > PGRC_FP_BITS) break;
        if (j < 3) {
            if (s + 4 * j + 3 < head) m |= 1u << fb;
        } else {
            if (s + 4 * j + 2 < head) m |= 1u << fb;
            if (s + 4 * j + 3 < head) m |= 1u << (fb + 2);
        }
        fb += width;
    }
    return m;
}

// mismatching symbols of two 2-bit words under a per-symbol mask (mask bits at even positions)
__device__ __forceinline__ uint32_t mism2(uint32_t a, uint32_t b, uint32_t mask) {
    uint32_t x = a ^ b;
    return (uint32_t)__popc((x | (x >> 1)) & mask);
}

// 0x55555555 restricted to symbols [from, to) of word k (symbols 16k .. 16k+15)
__device__ __host__ __forceinline__ uint32_t sym_mask(int k, int from, int to) {
    int lo = from - 16 * k, hi = to - 16 * k;
    if (lo < 0) lo = 0;
    if (hi > 16) hi = 16;
    if (hi <= lo) return 0u;
    uint32_t m = (hi == 16) ? 0xFFFFFFFFu : ((1u << (2 * hi)) - 1u);
    if (lo > 0) m &= ~((1u << (2 * lo)) - 1u);
    return m & 0x55555555u;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// reverse the order of the 16 symbols of a 2-bit word and complement them
__device__ __forceinline__ uint32_t revcomp_word(uint32_t w) {
    w = __brev(w);                                            // reverses bits: symbol order reversed, bit pairs swapped
    w = ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1);  // swap the bits of each pair back
    return ~w;                                                // complement: 3 - code
}
