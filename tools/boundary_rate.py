#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary: host ASCII / host packed rows -> device -> match -> results on the host.
(Never the bench `value`; recorded in DESIGN.md section 5.)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--pg", type=int, default=250_000_000)
    ap.add_argument("--L", type=int, default=150); a = ap.parse_args()
    import numpy as np, ctypes as C
    from pgrc_amd import MatchContext, synth
    g = synth.pg_params(a.pg, seed=12345); pg = synth.pg_host(g)
    rs = synth.reads_params(a.reads, a.L, seed=12345); reads = synth.reads_host(g, pg, rs)
    out = {"reads": a.reads, "L": a.L, "pg": a.pg}
    for kind in ("ascii", "packed"):
        if kind == "packed":  # the reference's own 4-symbols-per-byte rows (first symbol most significant)
            code = np.zeros(256, dtype=np.uint8); code[list(b"ACGT")] = [0, 1, 2, 3]
            c = code[reads]; pad = (-a.L) % 4
            c = np.concatenate([c, np.zeros((a.reads, pad), dtype=np.uint8)], axis=1).reshape(a.reads, -1, 4)
            rows = (c[:, :, 0] << 6 | c[:, :, 1] << 4 | c[:, :, 2] << 2 | c[:, :, 3]).astype(np.uint8)
        t0 = time.perf_counter()
        ctx = MatchContext(a.L, 38, a.L // 50, 0, "c"); ctx.set_pg_ascii(pg)
        t1 = time.perf_counter()
        if kind == "ascii": ctx.set_reads_ascii(reads)
        else: ctx.set_reads_packed(rows, a.reads)
        t2 = time.perf_counter()
        ctx.init_results(); ctx.run(True)
        t3 = time.perf_counter()
        pos, rc, mism, hist, matched = ctx.get_results()
        t4 = time.perf_counter()
        out[kind] = {"set_pg_s": t1 - t0, "set_reads_s": t2 - t1, "run_s": t3 - t2, "get_results_s": t4 - t3,
                     "total_s": t4 - t0, "reads_per_s_incl_pcie": a.reads / (t4 - t0),
                     "upload_GBps": (reads.nbytes if kind == "ascii" else rows.nbytes) / (t2 - t1) / 1e9}
        ctx.close()
    print(json.dumps(out))
if __name__ == "__main__":
    main()
