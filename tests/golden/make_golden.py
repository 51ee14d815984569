#!/usr/bin/env python3
"""Generates the committed golden fixtures from the REAL reference compiled in the build container
(oracle/_ref/libpgrc_ref.so, recipe: oracle/Makefile + oracle/ref_harness.cpp).  Fixtures are data only:
generator parameters (inputs are re-derived from include/pgrc_synth.h) + the reference's outputs.

    python tests/golden/make_golden.py        # needs /root/reference (run `make -C oracle ref` first)
    python tests/golden/make_golden.py --only-missing     # keeps the fixtures that exist (a zip file carries its writing time:
                                                          # regenerated data would be equal, the bytes of the files not)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
from util import make_inputs  # noqa: E402

# (name, mode char as given to -s, L, seed_len, M, G, n, n_with_n, rev_compl, gen seed, paired)
CASES = [
    ("c_L100_s38_M50", "c", 100, 38, 50, 200000, 10000, 0, True, 101, False),
    ("c_L100_s38_M3", "c", 100, 38, 3, 200000, 10000, 0, True, 102, False),
    ("C_L100_s38_M50_shortcut", "C", 100, 38, 50, 200000, 10000, 0, True, 103, False),
    ("c_L150_s38_M50", "c", 150, 38, 50, 200000, 10000, 0, True, 104, False),
    ("c_L150_s38_M3_pe", "c", 150, 38, 3, 200000, 10000, 0, True, 105, True),
    ("c_L250_s38_M50", "c", 250, 38, 50, 200000, 6000, 0, True, 106, False),
    ("c_L100_s100_M50_exactish", "c", 100, 100, 50, 200000, 10000, 0, True, 107, False),
    ("c_L150_s150_M3_exactish", "c", 150, 150, 3, 200000, 8000, 0, True, 108, False),
    ("c_L100_s38_M50_nreads", "c", 100, 38, 50, 200000, 10000, 800, True, 109, False),
    ("c_L100_s38_M3_fwdonly", "c", 100, 38, 3, 200000, 10000, 0, False, 110, False),
    ("c_L64_s32_M10", "c", 64, 32, 10, 150000, 8000, 0, True, 111, False),
    ("c_L40_s24_M8", "c", 40, 24, 8, 100000, 8000, 0, True, 112, False),
    ("e_L100", "d", 100, 100, 50, 200000, 10000, 0, True, 113, False),
    ("e_L150", "d", 150, 150, 50, 200000, 8000, 0, True, 114, False),
    ("d_L100_s38_M50", "d", 100, 38, 50, 200000, 10000, 0, True, 115, False),
    ("d_L100_s38_M3", "d", 100, 38, 3, 200000, 10000, 0, True, 116, False),
    ("D_L100_s38_M50_shortcut", "D", 100, 38, 50, 200000, 10000, 0, True, 117, False),
    ("d_L150_s38_M50", "d", 150, 38, 50, 200000, 8000, 0, True, 118, False),
    ("d_L250_s45_M50", "d", 250, 45, 50, 200000, 5000, 0, True, 119, False),
    ("i_L100_s38_M50", "i", 100, 38, 50, 200000, 10000, 0, True, 120, False),
    ("i_L100_s38_M3", "i", 100, 38, 3, 200000, 10000, 0, True, 121, False),
    ("i_L150_s38_M50", "i", 150, 38, 50, 200000, 8000, 0, True, 122, False),
    ("i_L250_s45_M50", "i", 250, 45, 50, 200000, 5000, 0, True, 123, False),
    # round 5: the cells of SURVEY 8(c)'s grid (L x mode x M x seed) that were still empty
    ("e_L250", "d", 250, 250, 50, 200000, 5000, 0, True, 124, False),
    ("C_L150_s38_M50_shortcut", "C", 150, 38, 50, 200000, 8000, 0, True, 125, False),
    ("C_L250_s38_M50_shortcut", "C", 250, 38, 50, 200000, 5000, 0, True, 126, False),
    ("D_L150_s38_M50_shortcut", "D", 150, 38, 50, 200000, 8000, 0, True, 127, False),
    ("D_L250_s45_M50_shortcut", "D", 250, 45, 50, 200000, 5000, 0, True, 128, False),
    ("c_L250_s38_M3", "c", 250, 38, 3, 200000, 5000, 0, True, 129, False),
    ("c_L250_s250_M50_exactish", "c", 250, 250, 50, 200000, 5000, 0, True, 130, False),
    ("d_L150_s38_M3", "d", 150, 38, 3, 200000, 8000, 0, True, 131, False),
    ("i_L150_s38_M3", "i", 150, 38, 3, 200000, 8000, 0, True, 132, False),
    ("d_L250_s45_M3", "d", 250, 45, 3, 200000, 5000, 0, True, 133, False),
    ("i_L250_s45_M3", "i", 250, 45, 3, 200000, 5000, 0, True, 134, False),
    ("I_L150_s38_M50_shortcut", "I", 150, 38, 50, 200000, 8000, 0, True, 135, False),
    ("c_L150_s38_M50_nreads", "c", 150, 38, 50, 200000, 8000, 600, True, 136, False),
]


def spice(reads, pg, L):
    """Edge cases on top of the generator: duplicated reads, reads with identical parts, reads hanging over the
    Pg ends, a poly-A read."""
    reads = reads.copy()
    reads[11] = reads[10]
    reads[12] = reads[10]
    h = L // 2
    reads[20, h:2 * h] = reads[20, :h]
    reads[30, : L - 10] = pg[-(L - 10):]
    reads[31, 10:] = pg[: L - 10]
    reads[40, :] = ord("A")
    return reads


def case_inputs(c):
    name, mode, L, seed_len, M, G, n, n_with_n, rev, gseed, paired = c
    pg, reads = make_inputs(G, n, L, seed=gseed, n_with_n=n_with_n, paired=paired)
    return pg, spice(reads, pg, L)


def derive(mode, L, seed_len, M):
    kmax = L // M
    kmin = kmax if mode.isupper() else 0
    seed_len = min(seed_len, L)
    low = mode.lower()
    kind = ("c" if low == "c" else "e") if seed_len == L else low
    return kind, seed_len, kmax, kmin


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    manifest = {}
    only_missing = "--only-missing" in sys.argv
    if only_missing:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
    for c in CASES:
        name, mode, L, seed_len, M, G, n, n_with_n, rev, gseed, paired = c
        if only_missing and name in manifest and os.path.exists(os.path.join(HERE, name + ".npz")):
            continue
        pg, reads = case_inputs(c)
        kind, sl, kmax, kmin = derive(mode, L, seed_len, M)
        if kind == "c":
            r = orc.ref_match("c", pg, reads, sl, kmax, kmin, rev, n_nset=n_with_n, index_threads=1)
        else:
            # modes d/i/e: one set holding every read (the LQ+N sum set indexes nothing there, see tests)
            r = orc.ref_match(kind, pg, reads, sl, kmax, kmin, rev, n_nset=(n if n_with_n else 0), index_threads=1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pos=r["pos"], rc=r["rc"], mism=r["mism"], hist=r["hist"],
                            matched=np.uint64(r["matched"]))
        manifest[name] = {"mode": mode, "kind": kind, "L": L, "seed_len": seed_len, "M": M, "G": G, "n": n,
                          "n_with_n": n_with_n, "rev_compl": rev, "gen_seed": gseed, "paired": paired,
                          "kmax": kmax, "kmin": kmin,
                          "inputs_sha256": hashlib.sha256(pg.tobytes() + reads.tobytes()).hexdigest(),
                          "matched": r["matched"]}
        print(name, r["matched"], "/", n)
    if only_missing:
        with open(os.path.join(HERE, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    # copMEM index dumps (CopMEMMatcher.cpp:140-231, serial build) for three tiny texts
    for name, G, seed_len, gseed in (("idx_s38", 100000, 38, 201), ("idx_s45", 100000, 45, 202), ("idx_s150", 120000, 150, 203)):
        pg, _ = make_inputs(G, 1, 150, seed=gseed)
        pg[5000:7000] = ord("A")                     # poly-A tract: overflows a bucket (cap 13)
        pg[30000:31200] = np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 1200)
        prm, cumm, positions = orc.ref_index(pg, seed_len, 1)
        counts = np.diff(cumm[:-1].astype(np.int64))
        nz = np.flatnonzero(counts).astype(np.uint32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), positions=positions, buckets=nz,
                            counts=counts[nz].astype(np.uint8))
        manifest[name] = {"index": True, "G": G, "seed_len": seed_len, "gen_seed": gseed, **prm,
                          "entries": int(positions.size), "inputs_sha256": hashlib.sha256(pg.tobytes()).hexdigest()}
        print(name, prm, positions.size)
    # mismatch lists (updateEntry, ReadsMatchers.cpp:548-559) for one case, SE rule and pair-file rule
    pg, reads = case_inputs(CASES[8])
    r = orc.ref_match("c", pg, reads, 38, 33, 0, True, n_nset=800, index_threads=1)
    rows = []
    for i in np.flatnonzero((r["mism"] != 255) & (r["mism"] > 0))[:600]:
        for pair_file in (0, 1):
            codes, offs = orc.ref_extract(pg, r["pos"][i], reads[i], r["rc"][i], r["mism"][i], org_idx=int(i),
                                          rev_compl_pair_file=bool(pair_file))
            rows.append((int(i), pair_file, codes.tolist(), offs.tolist()))
    np.savez_compressed(os.path.join(HERE, "extract_c_nreads_M3.npz"), pos=r["pos"], rc=r["rc"], mism=r["mism"])
    with open(os.path.join(HERE, "extract_c_nreads_M3.json"), "w") as f:
        json.dump(rows, f)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
