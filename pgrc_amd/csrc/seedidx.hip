// seedidx.hip -- modes 'd' / 'i' / 'e' (read-side seed index).  Placeholder until built.
#include "ctx.h"

int pgrc_seedidx_run(pgrc_match_ctx *c, int rev_compl_pg) {
    (void)rev_compl_pg;
    c->err = "modes d/i/e are not built yet";
    return PGRC_E_MODE;
}
