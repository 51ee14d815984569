#!/usr/bin/env python3
"""The boundary at the size of BASELINE config C3: the encoder's LQ + N sum set (100 M x 150 bp: 98 M reads as the
reference's ACGT rows, 4 symbols per byte, + 2 M reads holding an N as its ACGNT rows, 3 per byte) and a 1.875 Gbp
pseudogenome as ASCII, all in HOST memory; timed from context creation to the three result arrays back on the host
(PCIe included, never the bench `value`).  Two legs:
  plain      set_pg_ascii, set_reads_packed_sets, init_results, run, get_results -- one step after the other;
  pipelined  set_pg_ascii, prepare_index, begin_reads, stream_begin, append_reads_packed ..., end_reads, stream_end
             (pgrc_amd/csrc/stream.hip): index builds beneath the upload, blocks matched while the next ones are copied,
             results downloaded block by block.
Both must produce the same arrays.  usage: python tools/boundary_c3.py [--reads N --pg G] > profiles/rNN_boundary_c3.json"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_host_inputs(n, L, G, nfrac):
    """-> (pg ASCII, LQ rows in the reference's ACGT packing, N-set rows in its ACGNT packing, n_lq, n_n): the workload's
    synthetic inputs (include/pgrc_synth.h) generated on the device and brought to host memory"""
    import numpy as np
    import torch
    import bench
    from pgrc_amd import synth
    n_n = int(n * nfrac) & ~1
    n_lq = n - n_n
    nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
    g = synth.pg_params(G, seed=12345)
    rs = synth.reads_params(n, L, seed=12345, n_with_n=n_n)
    d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda")
    synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    # host text
    pg = bench.unpack_pg_to_ascii(d_pg[:pgw].cpu().numpy().view(np.uint32))[:G].copy()
    # LQ rows in the reference's ACGT packing (SymbolsPackingFacility: first symbol in the high bits), made on the device
    # from the word-major 2-bit read set: byte b of a row = symbols 4b .. 4b+3
    rb = (L + 3) // 4
    rev = torch.tensor([((v & 3) << 6) | (((v >> 2) & 3) << 4) | (((v >> 4) & 3) << 2) | (v >> 6) for v in range(256)], dtype=torch.uint8, device="cuda")
    lq = torch.empty((n_lq, rb), dtype=torch.uint8, device="cuda")
    for w in range(nw):
        words = d_rd[w * stride: w * stride + n_lq]
        for k in range(4):
            b = 4 * w + k
            if b < rb:
                lq[:, b] = rev[((words >> (8 * k)) & 0xFF).long()]
    lq_rows = lq.cpu().numpy()
    del lq, rev
    # the N set: the generator's last n_n reads carry an N; their ASCII rows come from the host generator
    n_ascii = synth.reads_host(g, pg, rs, n_lq, n_n)
    code = np.zeros(256, dtype=np.uint8); code[list(b"ACGNT")] = [0, 1, 2, 3, 4]
    c = code[n_ascii]; pad = (-L) % 3
    c = np.concatenate([c, np.zeros((n_n, pad), dtype=np.uint8)], axis=1).reshape(n_n, -1, 3)
    n_rows = (c[:, :, 0] * 25 + c[:, :, 1] * 5 + c[:, :, 2]).astype(np.uint8)
    del d_rd, d_pg, c
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    return pg, lq_rows, n_rows, n_lq, n_n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--pg", type=int, default=1_875_000_000)
    ap.add_argument("--L", type=int, default=150)
    ap.add_argument("--nfrac", type=float, default=0.02)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--legs", default="plain,pipelined")
    ap.add_argument("--ab-env", default="", help="NAME=v1,v2: the pipelined leg alternates the variable between the values, rep by rep (one process)")
    a = ap.parse_args()
    import numpy as np
    from pgrc_amd import MatchContext
    n, L, G = a.reads, a.L, a.pg
    t0 = time.perf_counter()
    pg, lq_rows, n_rows, n_lq, n_n = make_host_inputs(n, L, G, a.nfrac)
    prep_s = time.perf_counter() - t0
    sets = [(lq_rows, n_lq, 4), (n_rows, n_n, 5)]
    out = {"reads": n, "L": L, "pg": G, "n_set_reads": n_n, "prep_s": prep_s, "host_bytes": {"pg_ascii": int(pg.nbytes), "lq_rows": int(lq_rows.nbytes), "n_rows": int(n_rows.nbytes), "results": n * 10},
           "note": "context creation to results on the host; the result arrays exist and are touched before (the reference's initMatching)"}
    ref = None
    # the caller's result arrays exist (and have been written once) before the job starts, as the reference's matcher's do:
    # DefaultReadsMatcher::initMatching fills readMatchPos / readMatchRC / readMismatchesCount (ReadsMatchers.cpp:97-105)
    res = (np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64), np.zeros(n, dtype=np.uint8), np.full(n, 255, dtype=np.uint8))
    for r_ in res:
        r_ += 0                                                  # (touch every page)
    # one throw-away job first: the process's first large device allocations can stall for seconds while the driver
    # scrubs memory an EARLIER process freed (tools/ubench/realloc.hip: 4 s for 36 GB), and the first launch loads the
    # code object -- neither is the boundary's cost
    warm = MatchContext(L, 38, L // 50, 0, "c"); warm.set_pg_ascii(pg); warm.set_reads_packed_sets(sets); warm.init_results(); warm.run(True)
    warm.get_results(out=res)
    del warm
    for leg in a.legs.split(","):
        runs = []
        for rep in range(a.reps):
            if a.ab_env and leg == "pipelined":
                name, vals = a.ab_env.split("=")
                os.environ[name] = vals.split(",")[rep % len(vals.split(","))]
            t = [time.perf_counter()]
            ctx = MatchContext(L, 38, L // 50, 0, "c"); ctx.set_pg_ascii(pg); t.append(time.perf_counter())
            if leg == "plain":
                ctx.set_reads_packed_sets(sets); t.append(time.perf_counter())
                ctx.init_results(); ctx.run(True); t.append(time.perf_counter())
                pos, rc, mism, hist, matched = ctx.get_results(out=res); t.append(time.perf_counter())
                ph = {"set_pg_s": t[1] - t[0], "set_reads_s": t[2] - t[1], "run_s": t[3] - t[2], "get_results_s": t[4] - t[3]}
            else:
                ctx.prepare_index(True); t.append(time.perf_counter())
                pos, rc, mism, hist, matched = ctx.match_streamed(sets, out=res); t.append(time.perf_counter())
                ph = {"set_pg_s": t[1] - t[0], "prepare_index_call_s": t[2] - t[1], "streamed_upload_match_download_s": t[3] - t[2]}
            total = t[-1] - t[0]
            if a.ab_env and leg == "pipelined":
                ph["env"] = a.ab_env.split("=")[0] + "=" + os.environ[a.ab_env.split("=")[0]]
            ph.update({"total_s": total, "reads_per_s_incl_pcie": n / total, "matched": int(matched), "redo_reads": int(ctx.counters()["redo_reads"])})
            runs.append(ph)
            if ref is None:
                ref = (pos.copy(), rc.copy(), mism.copy())
            else:
                ph["equal_to_first_run"] = bool(np.array_equal(pos, ref[0]) and np.array_equal(rc, ref[1]) and np.array_equal(mism, ref[2]))
            del ctx
        out[leg] = runs
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
