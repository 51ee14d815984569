#!/usr/bin/env python3
"""Boundary costs at the C3 size (100 M reads): result fetch and the bulk mismatch extraction the adapter's export uses."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pgrc_amd import MatchContext, synth
n, L, G, seed_len, kmax = 100_000_000, 150, 1_875_000_000, 38, 3
g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345)
nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
torch.cuda.synchronize()
ctx = MatchContext(L, seed_len, kmax, 0, "c"); ctx.set_pg_packed_device(d_pg.data_ptr(), G); ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
ctx.init_results(); ctx.run(True)
t = time.perf_counter(); pos, rc, mism, hist, matched = ctx.get_results(); t_get = time.perf_counter() - t
for rep in range(2):
    t = time.perf_counter(); cum, codes, offs = ctx.extract_mismatches(); t_ext = time.perf_counter() - t
    print(json.dumps({"reads": n, "matched": matched, "get_results_s": t_get, "extract_s": t_ext, "mismatches": int(cum[n])}), flush=True)
