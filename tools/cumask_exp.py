#!/usr/bin/env python3
"""How much does the match kernel lose on fewer CUs?  Runs the C3 passes of ONE context on streams created with
hipExtStreamCreateWithCUMask (bit b of the mask = the (b / 8)-th CU of XCC b % 8 on a 8-XCC part: keeping the first N
bits drops the same number of CUs from every XCC).  The question behind it: could the RC index build run beside the
forward match on CUs set aside for it?  (Tried: no -- profiles/r02_cumask_overlap.txt)  usage: tools/cumask_exp.py [--workload C3] [--rounds 3] 256 240 224 192 128"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("cus", nargs="+", type=int)
    a = ap.parse_args()
    import torch, bench
    from pgrc_amd import MatchContext, synth
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n, L, G, seed_len, M, mode, paired = bench.WORKLOADS[a.workload]
    g = synth.pg_params(G, seed=12345); rs = synth.reads_params(n, L, seed=12345, paired=paired)
    nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
    d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda"); synth.pg_device(g, d_pg.data_ptr())
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda"); synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    ctx = MatchContext(L, seed_len, L // M, 0, mode); ctx.set_pg_packed_device(d_pg.data_ptr(), G)
    ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd); ctx.set_profiling(True)
    streams = {}
    for ncu in a.cus:
        words = (C.c_uint32 * 8)(*[((1 << max(0, min(32, ncu - 32 * w))) - 1) & 0xFFFFFFFF for w in range(8)])
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
        assert rc == 0, f"hipExtStreamCreateWithCUMask({ncu}) = {rc}"
        streams[ncu] = st
    res = {k: [] for k in a.cus}
    ref = None
    for r in range(a.rounds + 1):
        for ncu in a.cus:
            ctx.set_stream(streams[ncu].value)
            ctx.init_results(); ctx.run(True)
            c = ctx.counters()
            _, _, _, hist, matched = ctx.get_results(arrays=False)
            if ref is None: ref = (hist.tolist(), matched)
            assert (hist.tolist(), matched) == ref
            if r: res[ncu].append((c["ms_match"][0], c["ms_match"][1], c["ms_index"][0] + c["ms_index"][1], c["ms_total"]))
    for ncu in a.cus:
        med = [statistics.median(x[k] for x in res[ncu]) for k in range(4)]
        print(f"CUs {ncu:4d}  match_fwd {med[0]:7.2f}  match_rc {med[1]:7.2f}  index {med[2]:7.2f}  total {med[3]:7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
