#!/usr/bin/env python3
"""One-off randomized soak at medium sizes (not part of the suite): texts of 5-80 Mbp and 0.2-2 M reads, so that the
index build runs with realistic partition and tile counts (every PGRC_INDEX_SORT / PGRC_INDEX_FINISH variant), the match
kernel with many waves in flight, and the sharded matcher with shards of real size.  HIP path vs oracle (mode c; the
index itself is compared for a third of the cases); a quarter of the cases take mode d / i / e instead (texts up to 30 Mbp:
the oracle's scan is serial), results AND the number of (window, part) pairs with equal keys per strand against the oracle's.
usage: python tests/soak_medium.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from util import gpu_match, make_inputs  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    cases = 0
    while time.time() - t0 < budget:
        L = int(rng.choice([100, 100, 150, 150, 250, int(rng.integers(60, 256))]))
        seed_len = int(rng.choice([38, 38, 38, 32, 45, min(L, 64), L]))
        kmax = int(rng.choice([L // 50, L // 50, L // 25, L // 3]))
        kmax = min(kmax, 247)
        kmin = kmax if rng.random() < 0.15 else 0
        G = int(rng.integers(5_000_000, 80_000_000))
        n = int(rng.integers(200_000, 2_000_000))
        nn = int(rng.integers(0, 20_000)) if rng.random() < 0.4 else 0
        seed = int(rng.integers(0, 1 << 30))
        pg, reads = make_inputs(G, n, L, seed=seed, n_with_n=nn, pool_div=int(rng.choice([8, 64])), tandem_every=int(rng.choice([0, 2, 64])),
                                paired=bool(rng.random() < 0.3))
        shards = int(rng.integers(2, 5)) if rng.random() < 0.3 else 0
        variant = str(rng.choice(["", "", "own"]))
        icfg = str(rng.choice(["", "", "0"]))     # round 5: the passes without the XCD-aware tile order
        finish = "general" if rng.random() < 0.25 else ""
        early = "0" if rng.random() < 0.2 else ""        # the match kernel probing every seed of every read
        stage = "0" if rng.random() < 0.2 else ""        # ... and without staged refills
        dual = str(rng.choice(["", "1", "1", "0"]))         # one query per read over both strands: forced / where it pays / never
        screen = str(rng.choice(["", "", "0", "1"]))       # "" = the dual kernel where it applies       # ... and the two passes in the reference's order (no exact-match screen)
        pairk = str(rng.choice(["", "", "0", "1", "3", "4"]))       # round 4: the pair table by group size (0 = a table per strand)
        inline = "0" if rng.random() < 0.3 else ""
        for key, val in (("PGRC_INDEX_SORT", variant), ("PGRC_INDEX_CFG", icfg), ("PGRC_INDEX_FINISH", finish), ("PGRC_EARLY_STOP", early), ("PGRC_MATCH_STAGE", stage), ("PGRC_SCREEN", screen), ("PGRC_DUAL", dual),
                         ("PGRC_HEAD_PAIR", pairk), ("PGRC_NREAD_INLINE", inline)):
            if val:
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        if rng.random() < 0.25:                          # modes d / i / e (seedidx.hip)
            mode = str(rng.choice(["d", "i", "e"]))
            Gs, ns = min(G, 30_000_000), min(n, 1_000_000)
            sl = L if mode == "e" else int(rng.choice([38, 38, 25, 45]))
            if mode != "e" and L // sl < 1:
                sl = L
            km = 0 if mode == "e" else max(L // sl - 1, 0)
            kn = km if (mode != "e" and rng.random() < 0.15) else 0
            heavy = str(rng.choice(["", "", "1", "4096"]))
            seg = str(int(rng.integers(1 << 20, 1 << 24))) if rng.random() < 0.3 else ""
            ssort = str(rng.choice(["", "", "segments", "full"]))
            hform = str(rng.choice(["", "", "window"]))
            for key, val in (("PGRC_SEED_HEAVY", heavy), ("PGRC_SEED_SEGMENT", seg), ("PGRC_SEED_SORT", ssort), ("PGRC_SEED_HEAVY_FORM", hform)):
                if val:
                    os.environ[key] = val
                else:
                    os.environ.pop(key, None)
            what = dict(mode=mode, L=L, seed_len=sl, kmax=km, kmin=kn, G=Gs, n=ns, nn=nn, seed=seed, shards=shards, heavy=heavy, seg=seg, sort=ssort, hform=hform)
            pg, reads = pg[:Gs], reads[:ns]
            o = orc.oracle_match(mode, pg, reads, sl, km, kn)
            g = gpu_match(mode, pg, reads, sl, km, kn, True, devices=[0] * shards if shards else None)
            for k in ("pos", "rc", "mism", "hist"):
                if not np.array_equal(np.asarray(g[k]), np.asarray(o[k])):
                    print("MISMATCH", what, k, int((np.asarray(g[k]) != np.asarray(o[k])).sum()), flush=True)
                    sys.exit(1)
            if not shards and g["ctx"].counters()["candidates"] != o["candidates"]:
                print("CANDIDATES MISMATCH", what, g["ctx"].counters()["candidates"], o["candidates"], flush=True)
                sys.exit(1)
            del g
            cases += 1
            print(f"{cases} ok ({time.time() - t0:.0f} s) {what}", flush=True)
            continue
        what = dict(L=L, seed_len=seed_len, kmax=kmax, kmin=kmin, G=G, n=n, nn=nn, seed=seed, shards=shards, variant=variant, finish=finish, early=early, stage=stage, screen=screen, dual=dual, pairk=pairk, inline=inline, icfg=icfg)
        o = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin, True, 16)
        g = gpu_match("c", pg, reads, seed_len, kmax, kmin, True, devices=[0] * shards if shards else None)
        for k in ("pos", "rc", "mism", "hist"):
            if not np.array_equal(np.asarray(g[k]), np.asarray(o[k])):
                print("MISMATCH", what, k, int((np.asarray(g[k]) != np.asarray(o[k])).sum()), flush=True)
                sys.exit(1)
        if cases % 3 == 0 and not shards:
            _, cumm, positions = orc.oracle_index(pg, seed_len)
            c, p = g["ctx"].export_index(0)
            if not (np.array_equal(c, cumm) and np.array_equal(p, positions)):
                print("INDEX MISMATCH", what, flush=True)
                sys.exit(1)
        del g
        cases += 1
        print(f"{cases} ok ({time.time() - t0:.0f} s) {what}", flush=True)
    print(f"soak ok: {cases} medium cases, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
