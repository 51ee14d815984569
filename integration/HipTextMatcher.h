// HipTextMatcher -- the reference-side binding of libpgrc_match.so for pseudogenome-vs-pseudogenome exact matching:
// a TextMatcher (matching/TextMatchers.h:53-61) that SimplePgMatcher can hold instead of its CopMEMMatcher
// (matching/SimplePgMatcher.cpp:12-19) and that forwards matchTexts to the MI355X library through the C ABI of
// include/pgrc_mem.h.
//
// NEW code for the PgRC tree (not part of the reference), compiled against the reference's headers.
#ifndef PGTOOLS_HIPTEXTMATCHER_H
#define PGTOOLS_HIPTEXTMATCHER_H

#include "matching/TextMatchers.h"

struct pgrc_mem_ctx;

namespace PgTools {

    class HipTextMatcher : public TextMatcher {
    private:
        pgrc_mem_ctx *ctx = nullptr;

    public:
        // same arguments as CopMEMMatcher(srcText, srcLength, targetMatchLength, minMatchLength)
        // (matching/copmem/CopMEMMatcher.h:79); srcText is borrowed, like CopMEMMatcher::start1
        HipTextMatcher(const char *srcText, const size_t srcLength, const uint32_t targetMatchLength,
                       uint32_t minMatchLength = UINT32_MAX);

        ~HipTextMatcher() override;

        void matchTexts(vector<TextMatch> &resMatches, const string &destText, bool destIsSrc, bool revComplMatching,
                        uint32_t minMatchLength) override;

        static uint64_t callsServed;     // diagnostics / tests
    };
}

#endif //PGTOOLS_HIPTEXTMATCHER_H
