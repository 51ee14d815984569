#!/usr/bin/env python3
"""Mode c at the C3 size with PgRC's shipped default -M 3 (k <= 50 instead of BASELINE's k <= 3): with such a limit the
fingerprint can reject almost nothing, so nearly every candidate is verified against the text."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pgrc_amd import MatchContext, synth
n, L, G, seed_len = 100_000_000, 150, 1_875_000_000, 38
g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345)
nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
torch.cuda.synchronize()
for kmax in (3, 10, 50):
    ctx = MatchContext(L, seed_len, kmax, 0, "c"); ctx.set_pg_packed_device(d_pg.data_ptr(), G); ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ctx.set_profiling(True)
    ts = []
    for _ in range(2):
        ctx.init_results(); torch.cuda.synchronize(); t = time.perf_counter(); ctx.run(True); ts.append(time.perf_counter() - t)
    _, _, _, hist, matched = ctx.get_results(arrays=False)
    c = ctx.counters()
    print(json.dumps({"kmax": kmax, "best_s": min(ts), "reads_per_s": n / min(ts), "matched": matched,
                      "schedule": {0: "two passes", 1: "screened", 2: "dual"}[c["screened"]], "redo_reads": c["redo_reads"], "dual": c["dual"],
                      "ms_index_pair": c["ms_index"][0], "ms_dual_kernel": c["ms_screen"], "ms_match_after": c["ms_match"],
                      "probes": c["probes"], "verifies": c["verifies"], "entry_fetches": c["entry_fetches"]}), flush=True)
    del ctx
