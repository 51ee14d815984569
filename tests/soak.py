#!/usr/bin/env python3
"""One-off randomized soak (not part of the suite): HIP path vs oracle on many random small configurations of both the
read matcher (modes c/d/i/e; reads handed over as ASCII rows or as the reference's packed LQ + N sets; one device or a
matcher over 2-5 shards; every index-build variant) with the export streams of every case, of the text matcher, and of
the read-set division (row f3).
usage: python tests/soak.py [seconds] [seed]"""
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
import export_util as xu  # noqa: E402


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
        shards = int(rng.integers(2, 6)) if rng.random() < 0.3 else 0
        packed = bool(rng.random() < 0.5)
        variant = str(rng.choice(["", "", "own"]))
        icfg = str(rng.choice(["", "", "0"]))     # round 5: the passes without the XCD-aware tile order
        finish = "general" if rng.random() < 0.3 else ""
        # modes d / i / e: small text segments and read batches now and then (the loops a >= 4 Gi text / >= 2^28 reads take)
        seg = str(int(rng.integers(4096, 40000))) if mode != "c" and rng.random() < 0.4 else ""
        batch = str(int(rng.integers(1, 1500))) if mode != "c" and rng.random() < 0.4 else ""
        stage = "0" if rng.random() < 0.2 else ""        # the match kernel without staged refills
        early = "0" if rng.random() < 0.25 else ""       # ... probing every seed of every read (no early stop)
        dual = str(rng.choice(["", "1", "1", "0"]))         # one query per read over both strands: forced / where it pays / never
        screen = str(rng.choice(["", "", "0", "1"]))       # "" = the dual kernel where it applies       # ... the two passes in the reference's order (no exact-match screen)
        # round 4: the pair table by group size (0 = a table per strand), reads with few N's inside the dual kernel or not
        pairk = str(rng.choice(["", "", "0", "1", "2", "3", "4"]))
        inline = "0" if rng.random() < 0.3 else ""
        heavy = str(rng.choice(["", "", "1", "4", "4096"]))   # modes d / i / e: entries of a window above which the persistent grid takes it
        ssort = str(rng.choice(["", "", "segments", "full"]))  # round 5: the table's pairs sorted by segments in LDS / by global passes
        hform = str(rng.choice(["", "", "window"]))            # ... heavy windows grouped by key (default) / a wave per window
        for key, val in (("PGRC_INDEX_SORT", variant), ("PGRC_INDEX_CFG", icfg), ("PGRC_INDEX_FINISH", finish), ("PGRC_SEED_SEGMENT", seg), ("PGRC_SEED_READ_BATCH", batch),
                         ("PGRC_MATCH_STAGE", stage), ("PGRC_EARLY_STOP", early), ("PGRC_SCREEN", screen), ("PGRC_DUAL", dual),
                         ("PGRC_HEAD_PAIR", pairk), ("PGRC_NREAD_INLINE", inline), ("PGRC_SEED_HEAVY", heavy), ("PGRC_SEED_SORT", ssort), ("PGRC_SEED_HEAVY_FORM", hform)):
            if val:
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        g = gpu_match(mode, pg, reads, seed_len, kmax, kmin, rev, devices=[0] * shards if shards else None,
                      n_nset=nn if packed else None)
        what = dict(mode=mode, L=L, seed_len=seed_len, M=M, kmin=kmin, G=G, n=n, nn=nn, rev=rev, seed=seed, shards=shards,
                    packed=packed, variant=variant, finish=finish, seg=seg, batch=batch, stage=stage, early=early, screen=screen, dual=dual,
                    pairk=pairk, inline=inline, icfg=icfg, heavy=heavy, ssort=ssort, hform=hform)
        for k in ("pos", "rc", "mism", "hist"):
            if not np.array_equal(np.asarray(g[k]), np.asarray(o[k])):
                print("READS MISMATCH", what, k, flush=True)
                sys.exit(1)
        # --- export streams of this result (single-device contexts), Pg order with a random reads list and original order
        if not shards:
            h = int(rng.integers(0, max(1, (G - L) // 40)))
            off = rng.integers(0, 80, size=h).astype(np.uint8)
            while h and int(off.astype(np.int64).sum()) > G - L:
                off = off[: off.size // 2]
                h = off.size
            total = h + n
            perm = rng.permutation(total).astype(np.uint32)
            case = {"pg": pg, "reads": reads, "n_n": nn, "L": L, "list_off": off, "list_org": perm[:h].copy(),
                    "list_rc": (rng.random(h) < 0.5).astype(np.uint8) if rng.random() < 0.7 else None,
                    "read_org": np.concatenate([np.sort(perm[h:h + n - nn]), np.sort(perm[h + n - nn:])]).astype(np.uint32), "total": total}
            res = {k: g[k] for k in ("pos", "rc", "mism")}
            pair, byte_mode = bool(rng.random() < 0.5), bool(rng.random() < 0.7)
            order = xu.stable_order(res["pos"])
            want = xu.oracle_export_pg_order(case, res, order, pair_file=pair, byte_mode=byte_mode)
            got = g["ctx"].export_pg_order(order if rng.random() < 0.5 else None, off, case["list_org"], case["list_rc"], case["read_org"], pair, byte_mode)   # (None: the order made on the device)
            er, eo = xu.original_order_entries(case["read_org"], res["mism"] != 255, total, pair, n - nn)
            want2 = xu.oracle_export_entries(case, res, er, eo, pair_file=pair, byte_mode=byte_mode)
            got2 = g["ctx"].export_entries(er, eo, pair, byte_mode)
            got3 = g["ctx"].export_original_order(case["read_org"], total, pair, pair, byte_mode)
            for k in xu.STREAMS:
                if not np.array_equal(got[k], want[k]) or not np.array_equal(got2[k], want2[k]) or not np.array_equal(got3[k], want2[k]):
                    print("EXPORT MISMATCH", what, dict(h=h, pair=pair, byte_mode=byte_mode), k, flush=True)
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
        # --- read-set division (row f3)
        from divide_util import make_records, oracle_divide, same
        from pgrc_amd import DividedPCLReadsSets
        dL = int(rng.integers(1, 256))
        dn = int(rng.choice([1, 63, 64, 65, 1000, 5000, 40000]))
        lim = float(rng.choice([1.0, 1.0, 0.001, 0.01, 0.05, 0.2, 0.5, 0.9, 0.999]))
        simp, sepn, nlq = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if lim < 1 and simp and not (0 <= int(dL * (1 - lim)) < dL):
            simp = False
        dseed = int(rng.integers(0, 1 << 30))
        dr, dq = make_records(dseed, dn, dL, n_frac=float(rng.choice([0.0, 0.02, 0.5])), low_quality_frac=float(rng.choice([0.0, 0.3, 1.0])))
        dv = DividedPCLReadsSets(dL, lim, simp, sepn, nlq)
        bad = same(dv.divide(dr, dq), oracle_divide(dr, dq, lim, simp, sepn, nlq))
        dv.close()
        if bad is not None:
            print("DIVIDE MISMATCH", dict(L=dL, n=dn, error_limit=lim, simplified=simp, separate_n=sepn, n_reads_lq=nlq, seed=dseed, field=bad), flush=True)
            sys.exit(1)
        if (n_reads_cases % 50) == 0:
            print(f"{n_reads_cases} read-matcher cases, {n_mem_cases} text-matcher cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"soak ok: {n_reads_cases} read-matcher cases, {n_mem_cases} text-matcher cases, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
