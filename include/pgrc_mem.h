/*
 * pgrc_mem.h -- C ABI of libpgrc_match.so, part 2: pseudogenome-vs-pseudogenome exact matching
 *  on MI355X.
 *
 * Drop-in boundary: the reference's TextMatcher seam (matching/TextMatchers.h:53-61).
 * SimplePgMatcher (matching/SimplePgMatcher.cpp:12-19) builds `new CopMEMMatcher(srcPg, len, targetMatchLength,
 * minMatchLength)` and calls `matcher->matchTexts(matches, destText, destIsSrc, revComplMatching, minMatchLength)`
 * (:24-55) once per destination pseudogenome.  The entry points below are what a `HipTextMatcher : TextMatcher`
 * binds (integration/HipTextMatcher.{h,cpp}); INTEGRATION.md shows the one-line change in SimplePgMatcher.
 *
 * Result semantics = CopMEMMatcher::matchTexts -> processExactMatchQueryTight
 * (matching/copmem/CopMEMMatcher.cpp:333-481, :604-622) over the SERIAL seed index (PgHelpers::numberOfThreads
 * == 1): the same matches in the same discovery order, including that scan's sequential skip rules and its
 * stale side-context registers near the text ends.
 *
 * Same conventions as pgrc_match.h: 0 = success, PGRC_E_* otherwise; host buffers stay the caller's; no CPU
 * fallback -- without a HIP device every call fails.
 */
#ifndef PGRC_MEM_H
#define PGRC_MEM_H

#include <stddef.h>
#include <stdint.h>

#include "pgrc_match.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgrc_mem_ctx pgrc_mem_ctx;

/* TextMatch (matching/TextMatchers.h:10-16) */
typedef struct {
    uint64_t pos_src;  /* posSrcText */
    uint64_t length;
    uint64_t pos_dest; /* posDestText, in the coordinates of the text handed to pgrc_mem_match_texts */
} pgrc_text_match;

/* CopMEMMatcher(srcText, srcLength, targetMatchLength, minMatchLength) (CopMEMMatcher.cpp:571-591).
 * ctor_min_match_len: UINT32_MAX or >= target_match_len (what SimplePgMatcher passes; smaller values change K and
 * are not supported).  24 <= target_match_len <= 255. */
int pgrc_mem_create(uint32_t target_match_len, uint32_t ctor_min_match_len, int32_t device, pgrc_mem_ctx **out);
void pgrc_mem_destroy(pgrc_mem_ctx *ctx);
const char *pgrc_mem_last_error(const pgrc_mem_ctx *ctx);

/* The source text (ACGT): packed to HBM and indexed (the constructor's processRef, :176-231, serial semantics).
 * The pointer is borrowed until the context is destroyed or another source is set (like CopMEMMatcher::start1). */
int pgrc_mem_set_src_ascii(pgrc_mem_ctx *ctx, const char *src, uint64_t n);

/* matchTexts (:604-622).  dest is the text as the reference hands it over (SimplePgMatcher reverse-complements it
 * first when revComplMatching, SimplePgMatcher.cpp:31-41); it may contain 'N'.  With dest_is_src the library uses
 * its own copy of the source (or its reverse complement) on the device and dest is only read on the host.
 * *matches is malloc'ed (free with pgrc_mem_free_matches), discovery order. */
int pgrc_mem_match_texts(pgrc_mem_ctx *ctx, const char *dest, uint64_t n2, int dest_is_src, int rev_compl_matching,
                         uint32_t min_match_len, pgrc_text_match **matches, uint64_t *count);
void pgrc_mem_free_matches(pgrc_text_match *matches);

/* introspection (tests, bench) */
typedef struct {
    uint64_t probes;       /* destination windows hashed */
    uint64_t events;       /* (window, index entry) pairs with equal K-mers */
    uint64_t stale_lookups;/* events that needed the stale-register emulation on the host */
    float ms_index, ms_probe, ms_sort, ms_extend;
    float ms_host;         /* compaction + download of the matches */
    float ms_replay;       /* the sequential rules on the device, all rounds (host clock) */
    uint32_t replay_rounds;/* passes over the event blocks until every block had seen the last match before it */
    uint32_t event_blocks; /* blocks of 256 windows that hold events */
} pgrc_mem_counters;
int pgrc_mem_get_counters(pgrc_mem_ctx *ctx, pgrc_mem_counters *out);

#ifdef __cplusplus
}
#endif
#endif /* PGRC_MEM_H */
