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
    pgrc_match_destroy(ctx);
    free(pg); free(reads); free(pos); free(rc); free(mism);
    return 0;
}
