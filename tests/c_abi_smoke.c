/* A plain C99 consumer of the C ABI (no HIP, no C++ in this translation unit): what a cgo / FFI binding sees.
 * Built and run by tests/test_abi.py (compile + symbol check, no GPU) and tests/test_gpu_parity.py (run). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgrc_match.h"

int main(int argc, char **argv) {
    const uint64_t G = 200000, n = 4000;
    const uint32_t L = 100;
    pgrc_synth_pg g = {2024, G, 20000, 3000, 8, 4};
    pgrc_synth_reads rs = {77, n, L, 0, 0};
    char *pg = (char *)malloc(G), *reads = (char *)malloc(n * L);
    uint64_t *pos = (uint64_t *)malloc(n * sizeof(uint64_t)), hist[256], matched = 0;
    uint8_t *rc = (uint8_t *)malloc(n), *mism = (uint8_t *)malloc(n);
    pgrc_match_params prm;
    pgrc_match_ctx *ctx = NULL;
    int e, dry = argc > 1 && strcmp(argv[1], "--dry") == 0;
    pgrc_synth_pg_host(&g, pg);
    pgrc_synth_reads_host(&g, pg, &rs, 0, n, reads);
    if ((e = pgrc_match_derive_params(L, 38, 50, 'c', &prm))) { fprintf(stderr, "derive: %d\n", e); return 2; }
    if (dry) { printf("dry ok: mode %c kmax %u\n", prm.mode, (unsigned)prm.max_mismatches); return 0; }
    if ((e = pgrc_match_create(&prm, &ctx))) { fprintf(stderr, "create: %d %s\n", e, pgrc_match_last_error(NULL)); return 3; }
    if ((e = pgrc_match_set_pg_ascii(ctx, pg, G)) || (e = pgrc_match_set_reads_ascii(ctx, reads, n)) ||
        (e = pgrc_match_init_results(ctx)) || (e = pgrc_match_run(ctx, 1)) ||
        (e = pgrc_match_get_results(ctx, pos, rc, mism, hist, &matched))) {
        fprintf(stderr, "error %d: %s\n", e, pgrc_match_last_error(ctx));
        return 4;
    }
    {
        uint64_t fnv = 1469598103934665603ull, i;
        for (i = 0; i < n; i++) fnv = (fnv ^ pos[i] ^ ((uint64_t)mism[i] << 56) ^ ((uint64_t)rc[i] << 48)) * 1099511628211ull;
        printf("matched %llu of %llu, exact %llu, digest %016llx\n", (unsigned long long)matched, (unsigned long long)n,
               (unsigned long long)hist[0], (unsigned long long)fnv);
    }
    /* the round-2 entry points from plain C: one matcher over two shards (both on device 0), reads handed over in the
     * reference's ACGT packing (4 symbols per byte, first symbol most significant), the export streams */
    {
        pgrc_match_ctx *two = NULL;
        const int32_t devs[2] = {0, 0};
        const uint32_t pb = (L + 3) / 4;
        uint8_t *rows = (uint8_t *)calloc(n, pb), *rc2 = (uint8_t *)malloc(n), *mism2 = (uint8_t *)malloc(n);
        uint64_t *pos2 = (uint64_t *)malloc(n * sizeof(uint64_t)), hist2[256], matched2 = 0, i;
        uint32_t *entry_read = (uint32_t *)malloc(n * sizeof(uint32_t)), *entry_org = (uint32_t *)malloc(n * sizeof(uint32_t)), x;
        pgrc_export_streams st;
        uint64_t mm = 0;
        for (i = 0; i < n; i++)
            for (x = 0; x < L; x++) {
                const char ch = reads[i * L + x];
                const uint32_t code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u;
                rows[i * pb + x / 4] |= (uint8_t)(code << (2 * (3 - x % 4)));
            }
        if ((e = pgrc_match_create_multi(&prm, 2, devs, &two))) { fprintf(stderr, "create_multi: %d %s\n", e, pgrc_match_last_error(NULL)); return 5; }
        if ((e = pgrc_match_set_pg_ascii(two, pg, G)) || (e = pgrc_match_begin_reads(two, n)) ||
            (e = pgrc_match_append_reads_packed(two, rows, n / 3, 4)) ||
            (e = pgrc_match_append_reads_packed(two, rows + (n / 3) * pb, n - n / 3, 4)) || (e = pgrc_match_end_reads(two)) ||
            (e = pgrc_match_init_results(two)) || (e = pgrc_match_run(two, 1)) ||
            (e = pgrc_match_get_results(two, pos2, rc2, mism2, hist2, &matched2))) {
            fprintf(stderr, "multi error %d: %s\n", e, pgrc_match_last_error(two));
            return 6;
        }
        if (pgrc_match_shard_count(two) != 2 || matched2 != matched || memcmp(pos, pos2, n * sizeof(uint64_t)) || memcmp(rc, rc2, n) ||
            memcmp(mism, mism2, n) || memcmp(hist, hist2, sizeof hist)) {
            fprintf(stderr, "two shards disagree with one device\n");
            return 7;
        }
        for (i = 0; i < n; i++) { entry_read[i] = (uint32_t)i; entry_org[i] = (uint32_t)i; mm += mism[i] == 255 ? 0 : mism[i]; }
        if ((e = pgrc_match_export_entries(two, entry_read, entry_org, n, 0, 1, &st))) {
            fprintf(stderr, "export error %d: %s\n", e, pgrc_match_last_error(two));
            return 8;
        }
        if (st.n_entries != n || st.n_mismatches != mm || st.off_width != 1) { fprintf(stderr, "export counts\n"); return 9; }
        printf("two shards agree; export: %llu entries, %llu mismatches\n", (unsigned long long)st.n_entries, (unsigned long long)st.n_mismatches);
        pgrc_match_free_export(&st);
        pgrc_match_destroy(two);
        free(rows); free(rc2); free(mism2); free(pos2); free(entry_read); free(entry_org);
    }
    pgrc_match_destroy(ctx);
    free(pg); free(reads); free(pos); free(rc); free(mism);
    return 0;
}
