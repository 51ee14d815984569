#!/usr/bin/env python3
"""In-process A/B of compile-time variants of the match kernel (tools/variants.sh build): every variant library is
loaded into the SAME process (one copy of the package per library), all contexts share one text and one read set in
HBM, and the variants run interleaved.
CAUTION: every context has its own index and result buffers, and the match kernel's time depends on where they were
allocated (up to 13 % between contexts of ONE library: DESIGN.md section 9).  Give every variant several contexts
(x=base y=base u=w7 v=w7) and compare the fastest of each -- or use a run-time knob in one context (tools/ab_match.py).
usage: tools/ab_libs.py [--workload C3] [--rounds 6] base w7 a=base b=base ...   ("name=variant": a second context)"""
import argparse, importlib.util, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_copy(name, lib):
    os.environ["PGRC_MATCH_LIB"] = lib
    pkg = os.path.join(ROOT, "pgrc_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    assert mod.LIB_PATH == lib
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch, bench
    from pgrc_amd import synth
    n, L, G, seed_len, M, mode, paired = bench.WORKLOADS[a.workload]
    g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345, paired=paired)
    nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
    d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    ctxs = []
    for v in a.variants:       # "name" or "name=variant": several contexts of ONE library (same code object) under different names
        m = load_copy("pgrc_amd_" + v.split("=")[0], os.path.join(ROOT, "pgrc_amd", "variants", f"libpgrc_match_{v.split('=')[-1]}.so"))
        ctx = m.MatchContext(L, seed_len, L // M, 0, mode); ctx.set_pg_packed_device(d_pg.data_ptr(), G)
        ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd); ctx.set_profiling(True)
        ctxs.append(ctx)
    res = {i: [] for i in range(len(ctxs))}
    ref = None
    for r in range(a.rounds + 1):
        for i, ctx in enumerate(ctxs):
            ctx.init_results(); ctx.run(True)
            c = ctx.counters()
            _, _, _, hist, matched = ctx.get_results(arrays=False)
            if ref is None: ref = (hist.tolist(), matched)
            assert (hist.tolist(), matched) == ref, "variants disagree on results"
            if r: res[i].append((c["ms_screen"] if c["screened"] == 2 else c["ms_match"][0], c["ms_match"][1], c["ms_total"]))   # (dual schedule: the dual kernel)
    for i, v in enumerate(a.variants):
        med = [statistics.median(x[k] for x in res[i]) for k in range(3)]
        mn = [min(x[k] for x in res[i]) for k in range(3)]
        print(f"{v:12s} median: fwd {med[0]:7.2f} rc {med[1]:7.2f} total {med[2]:7.2f}   min: fwd {mn[0]:7.2f} rc {mn[1]:7.2f} total {mn[2]:7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
