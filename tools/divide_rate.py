#!/usr/bin/env python3
"""Row f3: the read-set division + packing (include/pgrc_reads.h) on FASTQ-like records in host memory, batch by batch
through the C ABI (PCIe-inclusive by nature: the records come from a file), the kernels' own time and bytes, and the compiled
reference's loop (DividedPCLReadsSets::getQualityDivisionBasedReadsSets, one thread) on a sample.  Writes one JSON object.
usage: tools/divide_rate.py [--reads N] [--L 150] [--batch B] [--cpu-reads M]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--L", type=int, default=150)
    ap.add_argument("--batch", type=int, default=4 << 20)
    ap.add_argument("--cpu-reads", type=int, default=2_000_000)
    a = ap.parse_args()
    import numpy as np
    from divide_util import make_records, oracle_divide, ref_divide, same
    import oracle as orc
    from pgrc_amd import DividedPCLReadsSets
    n, L = a.reads, a.L
    base_r, base_q = make_records(seed=1, n=1_000_000, L=L)
    reps = -(-n // base_r.shape[0])
    reads, quals = np.tile(base_r, (reps, 1))[:n], np.tile(base_q, (reps, 1))[:n]
    combo = (0.05, False, True, False)                     # quality division by the arithmetic mean, N reads apart
    d = DividedPCLReadsSets(L, *combo)
    d.divide(reads[:1000], quals[:1000])                   # (first launch loads the code object)
    out = {"reads": n, "L": L, "batch": a.batch, "params": {"error_limit": combo[0], "simplified_suffix_mode": combo[1], "separateNReadsSet": combo[2], "nReadsLQ": combo[3]}}
    runs = []
    for rep in range(3):
        t = time.perf_counter()
        ms = {"upload": 0.0, "kernels": 0.0, "download": 0.0}
        counts = [0, 0, 0]
        for lo in range(0, n, a.batch):
            g = d.divide(reads[lo:lo + a.batch], quals[lo:lo + a.batch])
            for k, v in d.last_ms().items():
                ms[k] += v
            counts = [counts[0] + g["n_hq"], counts[1] + g["n_lq"], counts[2] + g["n_n"]]
        dt = time.perf_counter() - t
        runs.append({"wall_s": dt, "reads_per_s": n / dt, "ms": ms, "counts": counts})
    out["gpu"] = runs
    best = min(runs, key=lambda r: r["wall_s"])
    # bytes the kernels must move per read: symbol row + quality row in, symbol row in again for the packing, packed row out
    alg = 3 * L + (L + 3) // 4
    out["kernels"] = {"ms_per_M_reads": best["ms"]["kernels"] / (n / 1e6), "algorithmic_bytes_per_read": alg,
                      "achieved_GBps": alg * n / (best["ms"]["kernels"] * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0}
    out["link"] = {"bytes_per_read_up": 2 * L, "bytes_per_read_down": (L + 3) // 4, "upload_GBps": 2 * L * n / (best["ms"]["upload"] * 1e-3) / 1e9}
    # the same straight from FASTQ text (parsed on the device): 4 M of the records as one text, handed over in pieces of 256 MiB
    from divide_util import make_fastq
    nf = min(n, 4_000_000)
    base_text = make_fastq(base_r[:500_000], base_q[:500_000], seed=3)
    text = base_text * (nf // 500_000)
    nf = 500_000 * (nf // 500_000)
    piece = 256 << 20
    best = None
    for rep in range(3):
        t = time.perf_counter()
        at, taken, ms = 0, 0, {"upload": 0.0, "kernels": 0.0, "download": 0.0}
        while True:
            chunk = text[at: at + piece]
            final = at + len(chunk) >= len(text)
            g, nrec, used, _ = d.divide_fastq(chunk, None, False, final=final)
            for k, v in d.last_ms().items():
                ms[k] += v
            taken += nrec
            at += used
            if final:
                break
        dt = time.perf_counter() - t
        if best is None or dt < best["wall_s"]:
            best = {"wall_s": dt, "records": taken, "records_per_s": taken / dt, "text_bytes": len(text), "text_GBps": len(text) / dt / 1e9,
                    "ms (upload + parsing, kernels, download)": ms}
    assert best["records"] == nf
    out["gpu_from_fastq_text"] = best
    m = min(a.cpu_reads, n)
    if m and orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_divide"):
        t = time.perf_counter()
        r = ref_divide(reads[:m], quals[:m], *combo)
        dt = time.perf_counter() - t
        g = d.divide(reads[:m], quals[:m])
        out["cpu_reference"] = {"reads": m, "wall_s": dt, "reads_per_s": m / dt, "threads": 1, "equal_to_gpu_on_the_sample": same(g, r) is None}
    d.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
