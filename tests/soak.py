#!/usr/bin/env python3
"""One-off randomized soak (not part of the suite): HIP path vs oracle on many random small configurations of both the
read matcher (modes c/d/i/e) and the text matcher.  usage: python tests/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from mem_util import mem_sweep_texts  # noqa: E402
from util import gpu_match, make_inputs  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    from pgrc_amd import CopMEMMatcher
    t0 = time.time()
    n_reads_cases = n_mem_cases = 0
    while time.time() - t0 < budget:
        # --- read matcher
        mode = str(rng.choice(["c", "c", "c", "d", "i", "e"]))
        L = int(rng.integers(40, 256))
        if mode == "e":
            seed_len = L
        elif mode == "c":
            seed_len = int(rng.integers(24, min(L, 140) + 1))
        else:
            seed_len = int(rng.integers(max(20, L // 15 + 1), L // 2 + 1))
        M = int(rng.choice([1000, 60, 50, 25, 10, 4, 3]))
        kmax = min(L // M, 247)
        kmin = kmax if rng.random() < 0.25 else 0
        G = int(rng.integers(L + 50, 120000))
        n = int(rng.integers(1, 3000))
        nn = int(rng.integers(0, min(n, 200))) if rng.random() < 0.4 else 0
        rev = bool(rng.random() < 0.8)
        seed = int(rng.integers(0, 1 << 30))
        pg, reads = make_inputs(G, n, L, seed=seed, n_with_n=nn, pool_div=int(rng.choice([8, 64])), tandem_every=int(rng.choice([0, 2, 64])))
        o = orc.oracle_match(mode, pg, reads, seed_len, kmax, kmin, rev)
        g = gpu_match(mode, pg, reads, seed_len, kmax, kmin, rev)
        for k in ("pos", "rc", "mism", "hist"):
            if not np.array_equal(np.asarray(g[k]), np.asarray(o[k])):
                print("READS MISMATCH", dict(mode=mode, L=L, seed_len=seed_len, M=M, kmin=kmin, G=G, n=n, nn=nn, rev=rev, seed=seed), k, flush=True)
                sys.exit(1)
        n_reads_cases += 1
        # --- text matcher
        target = int(rng.choice([24, 28, 33, 38, 45, 45, 45, 50, 64, 90, 130, 255]))
        min_len = target + int(rng.integers(0, 2)) * int(rng.integers(0, 40))
        mseed = int(rng.integers(0, 1 << 30))
        src, other = mem_sweep_texts(mseed, target)
        m = CopMEMMatcher(src, target)
        for dis, rc in ((0, 1), (1, 1), (0, 0), (1, 0)):
            d = orc.mem_dest(src, other, dis, rc)
            if not np.array_equal(m.matchTexts(d, dis, rc, min_len), orc.oracle_mem_match(src, d, dis, rc, target, min_len)):
                print("MEM MISMATCH", dict(target=target, min_len=min_len, seed=mseed, dis=dis, rc=rc), flush=True)
                sys.exit(1)
        m.close()
        n_mem_cases += 1
        if (n_reads_cases % 50) == 0:
            print(f"{n_reads_cases} read-matcher cases, {n_mem_cases} text-matcher cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"soak ok: {n_reads_cases} read-matcher cases, {n_mem_cases} text-matcher cases, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
