#!/usr/bin/env python3
"""In-process A/B of run-time options (PGRC_* variables, re-read by the context before every run: MatchContext.reload_options) on one device, interleaved rounds
(cdna_hip_programming.md rule 24: never rank builds across processes/devices).
usage: tools/ab_match.py [--workload C3] [--rounds 4] VAR=val,VAR=val  VAR=val ...   (each arg = one variant)"""
import argparse, json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--sorted-reads", action="store_true", help="experiment: position-sorted read set")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch, bench
    from pgrc_amd import MatchContext, synth
    n, L, G, seed_len, M, mode, paired = bench.WORKLOADS[a.workload]
    nfrac = bench.N_FRACTION.get(a.workload, 0.0)
    ctx = MatchContext(L, seed_len, L // M, 0, mode)
    if nfrac:      # the LQ + N sum set through the boundary (tools/boundary_c3.py), once, before anything is timed
        import boundary_c3
        pg_host, lq_rows, n_rows, n_lq, n_n = boundary_c3.make_host_inputs(n, L, G, nfrac)
        ctx.set_pg_ascii(pg_host); ctx.set_reads_packed_sets([(lq_rows, n_lq, 4), (n_rows, n_n, 5)])
        del pg_host, lq_rows, n_rows
    else:
        g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345, paired=paired)
        if a.sorted_reads: rs.paired |= 2
        nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
        d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
        d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
        torch.cuda.synchronize()
        ctx.set_pg_packed_device(d_pg.data_ptr(), G)
        ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ctx.set_profiling(True)
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in a.variants]
    keys = sorted({k for v in variants for k in v})
    res = {i: [] for i in range(len(variants))}
    ref = None
    for r in range(a.rounds + 1):
        for i, v in enumerate(variants):
            for k in keys: os.environ.pop(k, None)
            os.environ.update(v)
            ctx.reload_options()
            ctx.init_results(); ctx.run(True)
            c = ctx.counters()
            _, _, _, hist, matched = ctx.get_results(arrays=False)
            if ref is None: ref = (hist.tolist(), matched)
            assert (hist.tolist(), matched) == ref, "variants disagree on results"
            if r: res[i].append((c["ms_match"][0], c["ms_match"][1], c["ms_index"][0] + c["ms_index"][1], c["ms_total"], c["ms_screen"]))
    for i, v in enumerate(variants):
        med = [statistics.median(x[k] for x in res[i]) for k in range(5)]
        print(f"{a.variants[i]:40s} dual/screen {med[4]:7.2f}  match_fwd {med[0]:7.2f}  match_rc {med[1]:7.2f}  index {med[2]:7.2f}  total {med[3]:7.2f} ms")
if __name__ == "__main__":
    main()
