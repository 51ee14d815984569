#include "HipTextMatcher.h"

#include "pgrc_mem.h"

namespace PgTools {

    uint64_t HipTextMatcher::callsServed = 0;

    HipTextMatcher::HipTextMatcher(const char *srcText, const size_t srcLength, const uint32_t targetMatchLength,
                                   uint32_t minMatchLength) {
        int e = pgrc_mem_create(targetMatchLength, minMatchLength, -1, &ctx);
        if (e) {
            fprintf(stderr, "HipTextMatcher: %s (error %d)\n", pgrc_mem_last_error(nullptr), e);
            exit(EXIT_FAILURE);      // the reference's error convention (CopMEMMatcher.cpp:77-80)
        }
        e = pgrc_mem_set_src_ascii(ctx, srcText, srcLength);
        if (e) {
            fprintf(stderr, "HipTextMatcher: %s (error %d)\n", pgrc_mem_last_error(ctx), e);
            exit(EXIT_FAILURE);
        }
    }

    HipTextMatcher::~HipTextMatcher() {
        pgrc_mem_destroy(ctx);
    }

    void HipTextMatcher::matchTexts(vector<TextMatch> &resMatches, const string &destText, bool destIsSrc,
                                    bool revComplMatching, uint32_t minMatchLength) {
        resMatches.clear();
        pgrc_text_match *m = nullptr;
        uint64_t n = 0;
        const int e = pgrc_mem_match_texts(ctx, destText.data(), destText.length(), destIsSrc ? 1 : 0,
                                           revComplMatching ? 1 : 0, minMatchLength, &m, &n);
        if (e) {
            fprintf(stderr, "HipTextMatcher: %s (error %d)\n", pgrc_mem_last_error(ctx), e);
            exit(EXIT_FAILURE);      // e.g. "Minimal matching length cannot be smaller than K" (CopMEMMatcher.cpp:606-609)
        }
        resMatches.reserve(n);
        for (uint64_t i = 0; i < n; i++)
            resMatches.emplace_back(m[i].pos_src, m[i].length, m[i].pos_dest);
        pgrc_mem_free_matches(m);
        callsServed++;
    }
}
