#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary: host ASCII / host packed rows -> device -> match -> results on the host.
(Never the bench `value`; recorded in DESIGN.md section 5.)  Legs: ASCII rows; one ACGT-packed set; the LQ + N sum set
of pgrc-encoder.cpp:349-352 in the reference's own packed layouts (ACGT 4/byte + ACGNT 3/byte, 2 % of the reads in
the N set), handed over as ASCII rows (what getRead yields) and as packed sets (pgrc_match_append_reads_packed)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def pack_acgt(reads):
    import numpy as np
    n, L = reads.shape
    code = np.zeros(256, dtype=np.uint8); code[list(b"ACGT")] = [0, 1, 2, 3]
    c = code[reads]; pad = (-L) % 4
    c = np.concatenate([c, np.zeros((n, pad), dtype=np.uint8)], axis=1).reshape(n, -1, 4)
    return (c[:, :, 0] << 6 | c[:, :, 1] << 4 | c[:, :, 2] << 2 | c[:, :, 3]).astype(np.uint8)


def pack_acgnt(reads):
    import numpy as np
    n, L = reads.shape
    code = np.zeros(256, dtype=np.uint8); code[list(b"ACGNT")] = [0, 1, 2, 3, 4]
    c = code[reads]; pad = (-L) % 3
    c = np.concatenate([c, np.zeros((n, pad), dtype=np.uint8)], axis=1).reshape(n, -1, 3)
    return (c[:, :, 0] * 25 + c[:, :, 1] * 5 + c[:, :, 2]).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--pg", type=int, default=250_000_000)
    ap.add_argument("--L", type=int, default=150); a = ap.parse_args()
    import numpy as np
    from pgrc_amd import MatchContext, synth
    g = synth.pg_params(a.pg, seed=12345); pg = synth.pg_host(g)
    n_n = a.reads // 50
    out = {"reads": a.reads, "L": a.L, "pg": a.pg, "n_set_reads": n_n}
    legs = [("ascii", 0), ("packed", 0), ("sum_set_ascii", n_n), ("sum_set_packed", n_n)]
    cache = {}
    for kind, with_n in legs:
        if with_n not in cache:
            rs = synth.reads_params(a.reads, a.L, seed=12345, n_with_n=with_n)
            reads = synth.reads_host(g, pg, rs)
            n_lq = a.reads - with_n
            cache = {with_n: (reads, pack_acgt(reads[:n_lq]), pack_acgnt(reads[n_lq:]) if with_n else None)}
        reads, lq_rows, n_rows = cache[with_n]
        n_lq = a.reads - with_n
        t0 = time.perf_counter()
        ctx = MatchContext(a.L, 38, a.L // 50, 0, "c"); ctx.set_pg_ascii(pg)
        t1 = time.perf_counter()
        if kind in ("ascii", "sum_set_ascii"): ctx.set_reads_ascii(reads); nbytes = reads.nbytes
        elif kind == "packed": ctx.set_reads_packed(lq_rows, a.reads); nbytes = lq_rows.nbytes
        else: ctx.set_reads_packed_sets([(lq_rows, n_lq, 4), (n_rows, with_n, 5)]); nbytes = lq_rows.nbytes + n_rows.nbytes
        t2 = time.perf_counter()
        ctx.init_results(); ctx.run(True)
        t3 = time.perf_counter()
        pos, rc, mism, hist, matched = ctx.get_results()
        t4 = time.perf_counter()
        out[kind] = {"set_pg_s": t1 - t0, "set_reads_s": t2 - t1, "run_s": t3 - t2, "get_results_s": t4 - t3,
                     "total_s": t4 - t0, "reads_per_s_incl_pcie": a.reads / (t4 - t0),
                     "upload_GBps": nbytes / (t2 - t1) / 1e9, "matched": matched}
        ctx.close()
    assert out["sum_set_ascii"]["matched"] == out["sum_set_packed"]["matched"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
