#!/usr/bin/env python3
"""Wall time of modes e / d / i / c through the C ABI (device-resident inputs) at C1 / C2-like sizes."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pgrc_amd import MatchContext, synth
def run(mode, n, L, G, seed_len, kmax):
    g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345)
    nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
    d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    ctx = MatchContext(L, seed_len, kmax, 0, mode); ctx.set_pg_packed_device(d_pg.data_ptr(), G); ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ts = []
    for _ in range(3):
        ctx.init_results(); torch.cuda.synchronize(); t = time.perf_counter(); ctx.run(True); ts.append(time.perf_counter() - t)
    _, _, _, hist, matched = ctx.get_results(arrays=False)
    print(json.dumps({"mode": mode, "n": n, "L": L, "G": G, "seed": seed_len, "kmax": kmax, "best_s": min(ts), "reads_per_s": n / min(ts), "matched": matched}), flush=True)
run("e", 1_000_000, 100, 12_500_000, 100, 0)
run("d", 1_000_000, 100, 12_500_000, 38, 2)
run("i", 1_000_000, 100, 12_500_000, 38, 2)
run("c", 1_000_000, 100, 12_500_000, 38, 2)
run("d", 10_000_000, 100, 125_000_000, 38, 2)
run("i", 10_000_000, 100, 125_000_000, 38, 2)
run("c", 10_000_000, 100, 125_000_000, 38, 2)
