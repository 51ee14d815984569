#!/usr/bin/env python3
"""Cost of the byte path for reads with 'N' (the reference's separate N read set): C2-size run with 0 % and 2 % N reads."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from pgrc_amd import MatchContext, synth
n, L, G = 10_000_000, 100, 125_000_000
g = synth.pg_params(G, seed=12345)
pg = synth.pg_host(g)
modes = sys.argv[1:] or ["c"]
for mode, n_with_n in [(m, k) for m in modes for k in (0, 200_000)]:
    rs = synth.reads_params(n, L, seed=12345, n_with_n=n_with_n)
    reads = synth.reads_host(g, pg, rs)
    ctx = MatchContext(L, 38, 2, 0, mode); ctx.set_pg_ascii(pg); ctx.set_reads_ascii(reads); ctx.set_profiling(True)
    ts = []
    for _ in range(3):
        ctx.init_results(); t = time.perf_counter(); ctx.run(True); ts.append(time.perf_counter() - t)
    c = ctx.counters()
    print(json.dumps({"mode": mode, "n_with_n": n_with_n, "best_s": min(ts), "ms_total": c["ms_total"], "ms_match": c["ms_match"], "ms_index": c["ms_index"], "ms_other": c["ms_other"]}), flush=True)
    del ctx
