#!/usr/bin/env python3
"""Modes d / i at the C3 size (100 M x 150 bp reads, 1.875 Gbp Pg, seed 38, k <= 3) through the C ABI, inputs resident:
wall time of pgrc_match_run and the matched fraction (the CPU reference is serial here and would need hours)."""
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
for mode in (sys.argv[1:] or ["d", "i", "c"]):
    ctx = MatchContext(L, seed_len, kmax, 0, mode); ctx.set_pg_packed_device(d_pg.data_ptr(), G); ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ts = []
    for _ in range(2):
        ctx.init_results(); torch.cuda.synchronize(); t = time.perf_counter(); ctx.run(True); ts.append(time.perf_counter() - t)
    if os.environ.get("MODES_DIGEST"):       # a digest of the three result vectors: two runs under different knobs are compared by it
        import hashlib, zlib
        pos, rc, mism, hist, matched = ctx.get_results()
        dig = "%08x-%08x-%08x" % (zlib.crc32(pos.data), zlib.crc32(rc.data), zlib.crc32(mism.data))
        del pos, rc, mism
    else:
        dig = None
        _, _, _, hist, matched = ctx.get_results(arrays=False)
    # a roofline line for rows a5-a7 (SURVEY 8d's style of counting: the bytes the algorithm has to touch, whatever the design moves):
    # per window start and strand one 8-byte table key; per hit the entry (4 B), the read's and the window's L symbols at 2 bits,
    # the read's 8-byte key; per read its packed words and its three result fields.  The binding limit is the same as the dual
    # kernel's: random line requests (one per window probe, ~3 per hit), not bytes.
    P = 1 if mode == "e" else L // seed_len
    nwin = G - (L if mode == "e" else seed_len * (P if mode == "i" else 1)) + 1
    cand = ctx.counters()["candidates"]; hits = (cand[0] + cand[1]) // len(ts)      # (the counters add up over the runs)
    alg_bytes = 2 * nwin * 8 + hits * (4 + 2 * ((L + 3) // 4) + 8) + n * (4 * nw + 10)
    roof = {"bound": "hbm", "achieved": alg_bytes / min(ts) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg_bytes / min(ts) / 1e9 / 8000.0,
            "algorithmic_bytes": alg_bytes, "windows_per_strand": nwin, "hits_per_run": hits,
            "random_requests_estimate": 2 * nwin + 3 * hits + n * P, "requests_G_per_s": (2 * nwin + 3 * hits + n * P) / min(ts) / 1e9}
    print(json.dumps({"mode": mode, "digest": dig, "roofline": roof, "n": n, "L": L, "G": G, "seed": seed_len, "kmax": kmax, "best_s": min(ts), "first_s": ts[0],
                      "reads_per_s": n / min(ts), "matched": matched, "candidates": ctx.counters()["candidates"], "free_gb": torch.cuda.mem_get_info()[0] / 2**30}), flush=True)
    del ctx
