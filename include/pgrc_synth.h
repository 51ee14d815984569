/*
 * pgrc_synth.h -- deterministic synthetic pseudogenome + read-set generator
 * (the concretisation of BASELINE.json's configs).
 *
 * Everything is a PURE FUNCTION of (seed, index) built on the splitmix64
 * finaliser, so the same bytes come out of the host loops (tests, golden
 * fixtures, the CPU baseline) and of the HIP generator kernels (bench.py fills
 * HBM directly; nothing crosses PCIe).  Plain C99 inline functions; under
 * hipcc they are host+device.
 *
 * Pseudogenome: uniform ACGT; the Pg is cut into regions of `grid` symbols.
 * Every region holds one planted copy of a `plant_len` segment whose source is
 * drawn from a small pool (nreg / pool_div sources), and every `tandem_every`-th
 * region (by hash) holds a short-period tandem tract -- these exercise the
 * 13-entry bucket cap and the tie-breaks of the copMEM index.
 *
 * Reads (percent of all reads): 3 fully random, 60 exact, 15 one substitution,
 * 10 two, 5 three, 4 four-five, 3 six-eight; start uniform in [0, G-L], 50 %
 * reverse-complemented.  paired: odd reads lie 200-500 bp downstream of their
 * even mate on the opposite strand.  The last n_with_n reads additionally get
 * 1-3 'N' symbols (host/ASCII generator only; the N read set of the reference,
 * readsset/DividedPCLReadsSets.cpp:16-19).
 */
#ifndef PGRC_SYNTH_H
#define PGRC_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PGRC_HD __host__ __device__ static inline
#else
#define PGRC_HD static inline
#endif

typedef struct {
    uint64_t seed;
    uint64_t pg_len;
    uint32_t grid;         /* region size, default 20000 */
    uint32_t plant_len;    /* planted segment length, default 3000 */
    uint32_t pool_div;     /* sources = max(1, nreg / pool_div), default 8 */
    uint32_t tandem_every; /* 1 region in tandem_every carries a tandem tract; 0 = none */
} pgrc_synth_pg;

typedef struct {
    uint64_t seed;
    uint64_t n;          /* reads in the whole set (indices are global: shards pass first_read) */
    uint32_t read_len;
    uint32_t paired;     /* bit 0: PE (mates interleaved: 2q, 2q+1); bit 1: experiment only -- start positions ascending
                          * in j (a perfectly position-sorted read set, to measure what locality could buy) */
    uint64_t n_with_n;   /* the last n_with_n reads of the set carry 'N's (ASCII generator only) */
} pgrc_synth_reads;

PGRC_HD uint64_t pgrc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* counter-based stream: independent 64-bit value for every (seed, stream, ctr) */
PGRC_HD uint64_t pgrc_rnd(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return pgrc_mix64(pgrc_mix64(seed + 0x9E3779B97F4A7C15ull * (stream + 1)) ^
                      (ctr * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull));
}

enum {
    PGRC_S_PG = 1, PGRC_S_PLANT_DST = 2, PGRC_S_PLANT_POOL = 3, PGRC_S_PLANT_SRC = 4,
    PGRC_S_TANDEM = 5, PGRC_S_TYPE = 6, PGRC_S_START = 7, PGRC_S_RC = 8, PGRC_S_SUB = 9,
    PGRC_S_RAND = 10, PGRC_S_MATE = 11, PGRC_S_N = 12, PGRC_S_NSUB = 13
};

/* uniform base at i before planting: 32 bases per 64-bit draw */
PGRC_HD uint32_t pgrc_synth_base0(uint64_t seed, uint64_t i) {
    return (uint32_t)(pgrc_rnd(seed, PGRC_S_PG, i >> 5) >> (2 * (i & 31))) & 3u;
}

/* base code (A0 C1 G2 T3) of the synthetic Pg at position i */
PGRC_HD uint32_t pgrc_synth_pg_base(const pgrc_synth_pg *g, uint64_t i) {
    const uint64_t G = g->pg_len, grid = g->grid, pl = g->plant_len;
    if (grid == 0 || G < grid || pl == 0 || 2 * pl + 1024 > grid) return pgrc_synth_base0(g->seed, i);
    const uint64_t nreg = G / grid;
    const uint64_t r = i / grid;
    if (r >= nreg) return pgrc_synth_base0(g->seed, i);
    const uint64_t half = grid / 2;
    const uint64_t off = i - r * grid;
    if (off < half) {
        /* planted copy lives in the first half of the region */
        uint64_t dst = pgrc_rnd(g->seed, PGRC_S_PLANT_DST, r) % (half - pl);
        if (off >= dst && off < dst + pl) {
            uint64_t npool = nreg / (g->pool_div ? g->pool_div : 1);
            if (npool == 0) npool = 1;
            uint64_t q = pgrc_rnd(g->seed, PGRC_S_PLANT_POOL, r) % npool;
            uint64_t src = pgrc_rnd(g->seed, PGRC_S_PLANT_SRC, q) % (G - pl);
            return pgrc_synth_base0(g->seed, src + (off - dst));
        }
    } else if (g->tandem_every) {
        uint64_t t = pgrc_rnd(g->seed, PGRC_S_TANDEM, r);
        if (t % g->tandem_every == 0) {
            uint64_t len = 200 + ((t >> 8) % 800);
            uint64_t unit = 1 + ((t >> 24) % 6);
            uint64_t start = half + ((t >> 32) % (half - 1000));
            if (off >= start && off < start + len) {
                uint64_t u = (off - start) % unit;
                return (uint32_t)(pgrc_mix64(t ^ 0xABCDull) >> (2 * u)) & 3u;
            }
        }
    }
    return pgrc_synth_base0(g->seed, i);
}

typedef struct {
    uint32_t random;  /* 1 = fully random read */
    uint32_t rc;
    uint32_t nsub;
    uint64_t start;
} pgrc_synth_read_hdr;

PGRC_HD uint32_t pgrc_synth_nsub(uint32_t t, uint64_t v) {
    /* t in [3,100): 60/15/10/5/4/3 percent for 0/1/2/3/4-5/6-8 substitutions */
    if (t < 63) return 0;
    if (t < 78) return 1;
    if (t < 88) return 2;
    if (t < 93) return 3;
    if (t < 97) return 4 + (uint32_t)(v & 1);
    return 6 + (uint32_t)(v % 3);
}

PGRC_HD pgrc_synth_read_hdr pgrc_synth_read_header(const pgrc_synth_pg *g, const pgrc_synth_reads *rs,
                                                  uint64_t j) {
    pgrc_synth_read_hdr h;
    const uint64_t L = rs->read_len;
    const uint64_t span = g->pg_len - L + 1;
    uint32_t t = (uint32_t)(pgrc_rnd(rs->seed, PGRC_S_TYPE, j) % 100);
    h.random = t < 3;
    h.nsub = h.random ? 0 : pgrc_synth_nsub(t, pgrc_rnd(rs->seed, PGRC_S_NSUB, j));
    uint64_t anchor = (rs->paired & 1u) ? (j & ~1ull) : j;
    h.start = pgrc_rnd(rs->seed, PGRC_S_START, anchor) % span;
    if (rs->paired & 2u) { /* anchor * span / n without 128-bit arithmetic (anchor, span % n < n < 2^32) */
        const uint64_t nn = rs->n ? rs->n : 1;
        h.start = anchor * (span / nn) + (anchor * (span % nn)) / nn;
    }
    h.rc = (uint32_t)(pgrc_rnd(rs->seed, PGRC_S_RC, anchor) & 1);
    if ((rs->paired & 1u) && (j & 1)) {
        uint64_t d = 200 + pgrc_rnd(rs->seed, PGRC_S_MATE, j) % 301;
        h.start = (h.start + d < span) ? h.start + d : span - 1;
        h.rc ^= 1u;
    }
    return h;
}

/* Fills codes[0..L) with the 2-bit codes of read j (N's are NOT applied here). */
PGRC_HD void pgrc_synth_read_codes(const pgrc_synth_pg *g, const pgrc_synth_reads *rs, uint64_t j,
                                   uint8_t *codes) {
    const uint32_t L = rs->read_len;
    pgrc_synth_read_hdr h = pgrc_synth_read_header(g, rs, j);
    if (h.random) {
        for (uint32_t k = 0; k < L; k++)
            codes[k] = (uint8_t)((pgrc_rnd(rs->seed, PGRC_S_RAND, j * 8 + (k >> 5)) >> (2 * (k & 31))) & 3u);
        return;
    }
    if (!h.rc) {
        for (uint32_t k = 0; k < L; k++) codes[k] = (uint8_t)pgrc_synth_pg_base(g, h.start + k);
    } else {
        for (uint32_t k = 0; k < L; k++)
            codes[k] = (uint8_t)(3u - pgrc_synth_pg_base(g, h.start + (L - 1 - k)));
    }
    for (uint32_t q = 0; q < h.nsub; q++) {
        uint64_t w = pgrc_rnd(rs->seed, PGRC_S_SUB, j * 8 + q);
        uint32_t p = (uint32_t)(w % L);
        uint32_t d = 1 + (uint32_t)((w >> 32) % 3);
        codes[p] = (uint8_t)((codes[p] + d) & 3u);
    }
}

/* number of N symbols of read j and their positions (ASCII generator) */
PGRC_HD uint32_t pgrc_synth_read_n_count(const pgrc_synth_reads *rs, uint64_t j) {
    if (j + rs->n_with_n < rs->n) return 0;
    return 1 + (uint32_t)(pgrc_rnd(rs->seed, PGRC_S_N, j * 4) % 3);
}
PGRC_HD uint32_t pgrc_synth_read_n_pos(const pgrc_synth_reads *rs, uint64_t j, uint32_t q) {
    return (uint32_t)(pgrc_rnd(rs->seed, PGRC_S_N, j * 4 + 1 + q) % rs->read_len);
}

#endif /* PGRC_SYNTH_H */
