#!/usr/bin/env python3
"""Parity at BASELINE's FULL Pg size: GPU result for the first `--sample` reads of a workload vs the real
reference (oracle/_ref, serial canonical index, PgHelpers::numberOfThreads = 1) or the oracle port, run on the
host cores against the whole pseudogenome.  Test infrastructure (uses oracle/); writes a JSON summary."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this directory


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--sample", type=int, default=200000)
    ap.add_argument("--gpu-reads", type=int, default=0, help="reads matched on the GPU (0 = the sample only)")
    ap.add_argument("--threads", type=int, default=1,
                    help="threads of the reference's per-read loop; >1 exposes the reference's vector<bool> race "
                         "(DESIGN.md, reference quirk 7), so parity runs use 1")
    ap.add_argument("--checker", default="auto", choices=["auto", "reference", "port"])
    ap.add_argument("--out", default="gpurun_out/fullscale_parity.json")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import oracle as orc
    from pgrc_amd import MatchContext, synth

    n_per, L, G, seed_len, M, mode, paired = bench.WORKLOADS[a.workload]
    kmax = L // M
    n_gpu = a.gpu_reads or a.sample
    g = synth.pg_params(G, seed=12345)
    rs = synth.reads_params(n_per, L, seed=12345, paired=paired)
    nw, stride, pg_words = (L + 15) // 16, (n_gpu + 63) & ~63, (G + 15) // 16
    d_pg = torch.zeros(pg_words + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    d_reads = torch.empty(nw * stride, dtype=torch.int32, device="cuda")
    synth.reads_device(g, d_pg.data_ptr(), rs, 0, n_gpu, d_reads.data_ptr(), stride)
    torch.cuda.synchronize()
    ctx = MatchContext(L, seed_len, kmax, 0, mode)
    ctx.set_pg_packed_device(d_pg.data_ptr(), G)
    ctx.set_reads_device(d_reads.data_ptr(), n_gpu, stride, keep=d_reads)
    ctx.init_results()
    t = time.perf_counter()
    ctx.run(True)
    gpu_s = time.perf_counter() - t
    pos, rc, mism, hist, matched = ctx.get_results()
    ns = a.sample
    pg = bench.unpack_pg_to_ascii(ctx.export_pg(0))[:G]
    reads = synth.reads_host(g, pg, rs, 0, ns)
    use_ref = orc.have_ref() if a.checker == "auto" else a.checker == "reference"
    t = time.perf_counter()
    if mode != "c":
        use_ref = use_ref and a.checker == "reference"      # (modes d / i / e: the oracle's serial scan by default -- it also counts the candidates)
    if use_ref:
        r = orc.ref_match(mode, pg, reads, seed_len, kmax, 0, True, 0, 1, a.threads)
    else:
        r = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0, True, a.threads)
    cpu_s = time.perf_counter() - t
    out = {
        "workload": a.workload, "pg_len": G, "read_len": L, "sample_reads": ns, "gpu_reads": n_gpu,
        "checker": "reference (serial index, numberOfThreads=1)" if use_ref else "oracle port",
        "pos_equal": bool(np.array_equal(pos[:ns], r["pos"])),
        "rc_equal": bool(np.array_equal(rc[:ns], r["rc"])),
        "mism_equal": bool(np.array_equal(mism[:ns], r["mism"])),
        "n_pos_diff": int((pos[:ns] != r["pos"]).sum()), "n_mism_diff": int((mism[:ns] != r["mism"]).sum()),
        "matched_in_sample": int((r["mism"] != 255).sum()), "gpu_s": gpu_s, "cpu_s": cpu_s,
    }
    if mode != "c" and not use_ref:
        out["candidates_gpu"] = ctx.counters()["candidates"]
        out["candidates_checker"] = [int(v) for v in r["candidates"]]
        out["candidates_equal"] = out["candidates_gpu"] == out["candidates_checker"]
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    sys.exit(0 if out["pos_equal"] and out["rc_equal"] and out["mism_equal"] and out.get("candidates_equal", True) else 1)


if __name__ == "__main__":
    main()
