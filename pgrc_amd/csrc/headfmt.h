// headfmt.h -- layout of the copMEM seed index in HBM (built in copmem.hip; read by the match kernels there and
// by the text-matcher kernels of mem.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define HEAD_EMPTY 0xFFFFFFFFFFFFFFFFull
#define HEAD_OVF (1ull << 63)              // w0 flag: the bucket has >= 3 entries
#define ENT_MASK ((1ull << 62) - 1)        // entry = position << 22 | fingerprint, position < 2^40
#define W1_BASE_MASK ((1ull << 56) - 1)

// 16-B bucket head:
//   w0 = HEAD_EMPTY                                 empty bucket
//   w0 = entry0,            w1 = HEAD_EMPTY         one entry
//   w0 = entry0,            w1 = entry1             two entries
//   w0 = entry0 | HEAD_OVF, w1 = base | count << 56 count = min(n, 13) >= 3: entries 1.. at ent[base + j - 1]
// entry = position << 22 | fingerprint (62 bits), so u64 order = position order.
// Where the head of bucket h is: head[head_slot(h, fmt)] (ctx.h, head_sh).  fmt = 0: a table of its own per strand, slot h.
// fmt = 1 | gm << 8 (round 4): the pair table -- groups of gm + 1 consecutive buckets; a group's forward heads are followed by
// its RC heads (the strand's base pointer is already offset by gm + 1 heads), so both heads of a bucket number lie in one line of
// 32 * (gm + 1) bytes and each strand's build writes whole runs of 16 * (gm + 1) bytes.
__host__ __device__ __forceinline__ uint64_t head_slot(uint64_t h, uint32_t fmt) {
    const uint64_t gm = fmt >> 8;
    return ((h & ~gm) << (fmt & 1u)) | (h & gm);
}

__device__ __forceinline__ uint32_t head_count(const ulonglong2 hd) {
    if (hd.x == HEAD_EMPTY) return 0u;
    if (hd.x & HEAD_OVF) return (uint32_t)(hd.y >> 56) & 15u;
    return hd.y == HEAD_EMPTY ? 1u : 2u;
}

