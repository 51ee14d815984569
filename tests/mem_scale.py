#!/usr/bin/env python3
"""Row f2 at BASELINE's pseudogenome size: Pg-vs-Pg exact matching of a synthetic C3-size Pg (1.875 Gbp) against
itself on the GPU (through the C ABI), timed by phase, and -- optionally -- the same call on the compiled reference
(serial code, one core) for bit-parity and as the CPU baseline.  Test infrastructure (uses oracle/_ref); writes JSON.

usage: python tests/mem_scale.py [--pg-len N] [--cases fwd,rc,lq] [--no-reference] [--out FILE]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pg-len", type=int, default=1_875_000_000)
    ap.add_argument("--cases", default="fwd,rc,lq")
    ap.add_argument("--lq-len", type=int, default=60_000_000)
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--out", default="gpurun_out/mem_scale.json")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import oracle as orc
    from pgrc_amd import CopMEMMatcher, synth

    G = a.pg_len
    g = synth.pg_params(G, seed=12345)
    d_pg = torch.zeros((G + 15) // 16 + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    torch.cuda.synchronize()
    src = bench.unpack_pg_to_ascii(d_pg.cpu().numpy().view(np.uint32)[: (G + 15) // 16])[:G].copy()
    del d_pg
    out = {"pg_len": G, "target_match_len": 45, "cases": {}}
    t = time.perf_counter()
    m = CopMEMMatcher(src, 45)
    out["gpu_index_s"] = time.perf_counter() - t
    print(json.dumps({"src ready": G, "gpu_index_s": out["gpu_index_s"]}), flush=True)
    for case in a.cases.split(","):
        if case == "fwd":        # forward self-match: finds the planted forward copies of the synthetic Pg
            dest, dis, rc = src, 1, 0
        elif case == "rc":       # what the encoder runs on the HQ Pg (SimplePgMatcher.cpp:31-34)
            dest, dis, rc = orc.revcomp_ascii(src), 1, 1
        elif case == "long":     # ONE 20 Mbp exact copy of the source: 1.7 M events that all belong to the same match
            dest, dis, rc = src[G // 5: G // 5 + 20_000_000].copy(), 0, 1
        else:                    # an "LQ pseudogenome" that is the reverse complement of a slice of the source with a
            rng = np.random.default_rng(3)   # substitution every ~700 symbols: the text handed over is the slice (:36-38)
            sl = src[G // 3: G // 3 + a.lq_len].copy()
            pos = rng.integers(0, sl.size, size=sl.size // 700)
            sl[pos] = np.frombuffer(b"ACGT", dtype=np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), sl[pos]) + 1) % 4]
            dest, dis, rc = sl, 0, 1
        t = time.perf_counter()
        gm = m.matchTexts(dest, dis, rc)
        gpu_s = time.perf_counter() - t
        ctr = m.counters()
        rec = {"dest_len": int(dest.size), "dest_is_src": dis, "rev_compl": rc, "matches": int(len(gm)),
               "matched_symbols": int(gm[:, 1].sum()) if len(gm) else 0, "gpu_s": gpu_s,
               "gpu_windows_per_s": ctr["probes"] / gpu_s, "counters": ctr,
               "digest": __import__("hashlib").sha256(np.ascontiguousarray(gm).tobytes()).hexdigest()[:16]}
        print(json.dumps({case: rec}), flush=True)
        if not a.no_reference:
            t = time.perf_counter()
            rm = orc.ref_mem_match(src, dest, dis, rc)
            rec["ref_s_incl_index"] = time.perf_counter() - t
            rec["identical"] = bool(np.array_equal(gm, rm))
            rec["ref_matches"] = int(len(rm))
            print(json.dumps({case: {"ref_s_incl_index": rec["ref_s_incl_index"], "identical": rec["identical"]}}), flush=True)
        out["cases"][case] = rec
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    ok = all(c.get("identical", True) for c in out["cases"].values())
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
