#!/usr/bin/env python3
"""Golden fixtures for the inputs on which the HIP path's exact shortcuts (early stop, exact-match screen, one query over
both strands: pgrc_amd/csrc/copmem.hip) have the most to get wrong, at the headline configuration L = 150, seed 38,
-M 50 (k <= 3) -- where the dual kernel is the default schedule -- and at PgRC's shipped -M 3: repeat families and tandem
tracts (capped buckets, falses budgets running out), reverse palindromes (reads matching both strands equally well),
texts of period 5 .. 60 with reads from both strands.  Outputs of the REAL reference (serial index build), like
make_golden.py; inputs are rebuilt from the seeds by `case_inputs` below.

    python tests/golden/make_golden_hard.py        # needs /root/reference (run `make -C oracle ref` first)
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

L, SEED_LEN = 150, 38
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a_, b_ in zip(b"ACGTN", b"TGCAN"):
    COMP[a_] = b_

# name -> (kind, generator seed, M)
CASES = {
    "hard_repeat_families_M50": ("repeats", 301, 50),
    "hard_repeat_families_M3": ("repeats", 302, 3),
    "hard_palindromes_M50": ("palindromes", 303, 50),
    "hard_periods_5_13_M50": ("periods:5,13", 304, 50),
    "hard_periods_23_37_M50": ("periods:23,37", 305, 50),
    "hard_period_60_M50": ("periods:60", 306, 50),
    "hard_periods_7_29_M3": ("periods:7,29", 307, 3),
    "hard_both_strands_pe_M50": ("pe", 308, 50),
}


def revcomp(a):
    return COMP[a[::-1]]


def mutate(rng, w, most):
    w = w.copy()
    for _ in range(int(rng.integers(0, most + 1))):
        w[int(rng.integers(0, w.size))] = rng.choice(ACGT)
    return w


def case_inputs(name):
    from util import make_inputs
    kind, seed, _ = CASES[name]
    rng = np.random.default_rng(seed)
    if kind == "repeats":
        # planted copies drawn from a tiny pool (families of dozens of near-identical regions) + tandem tracts; every
        # second read from the other strand
        pg, reads = make_inputs(300_000, 6000, L, seed=seed, pool_div=64, tandem_every=2)
        for i in range(0, 3000, 2):
            reads[i] = revcomp(reads[i])
        return pg, reads
    if kind == "pe":
        return make_inputs(300_000, 6000, L, seed=seed, paired=True, n_with_n=200)
    if kind == "palindromes":
        pg, reads = make_inputs(250_000, 4000, L, seed=seed)
        for k in range(60):
            half = rng.choice(ACGT, size=L)
            pal = np.concatenate([half, revcomp(half)])              # equals its own reverse complement
            at = 2000 + 4000 * k
            pg[at:at + 2 * L] = pal
            reads[3 * k] = pal[L // 2:L // 2 + L]                    # centred: the same alignment on both strands
            reads[3 * k + 1] = mutate(rng, pal[20:20 + L], 3)        # off centre, a few substitutions
            reads[3 * k + 2] = revcomp(mutate(rng, pal[100:100 + L], 2))
        return pg, reads
    periods = [int(x) for x in kind.split(":")[1].split(",")]
    G, n = 120_000, 3000
    pg = np.empty(G, dtype=np.uint8)
    seg = G // len(periods)
    for j, per in enumerate(periods):
        unit = rng.choice(ACGT, size=per)
        lo, hi = j * seg, (G if j == len(periods) - 1 else (j + 1) * seg)
        pg[lo:hi] = np.tile(unit, (hi - lo) // per + 1)[:hi - lo]
    flips = rng.integers(0, G, size=G // 150)
    pg[flips] = rng.choice(ACGT, size=flips.size)
    reads = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        st = int(rng.integers(0, G - L))
        w = mutate(rng, pg[st:st + L], 4)
        reads[i] = revcomp(w) if rng.random() < 0.5 else w
    reads[-50:] = rng.choice(ACGT, size=(50, L))                     # and some that match nothing
    return pg, reads


def main():
    import oracle as orc
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    manifest = {}
    for name, (kind, seed, M) in CASES.items():
        pg, reads = case_inputs(name)
        n_nset = 200 if kind == "pe" else 0
        r = orc.ref_match("c", pg, reads, SEED_LEN, L // M, 0, True, n_nset=n_nset, index_threads=1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pos=r["pos"], rc=r["rc"], mism=r["mism"], hist=r["hist"],
                            matched=np.uint64(r["matched"]))
        manifest[name] = {"kind": kind, "gen_seed": seed, "M": M, "kmax": L // M, "L": L, "seed_len": SEED_LEN, "n": int(reads.shape[0]),
                          "G": int(pg.size), "n_nset": n_nset, "matched": int(r["matched"]), "rc_matched": int(r["rc"].sum()),
                          "inputs_sha256": hashlib.sha256(pg.tobytes() + reads.tobytes()).hexdigest()}
        print(name, manifest[name]["matched"], "/", reads.shape[0], "rc", manifest[name]["rc_matched"])
    with open(os.path.join(HERE, "manifest_hard.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
