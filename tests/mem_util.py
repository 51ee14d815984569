"""Synthetic texts for the Pg-vs-Pg exact-matching tests."""
import numpy as np

import oracle as orc

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_pg(G, seed, nrep=30, replen=400, rc_frac=0.5):
    """random ACGT text with nrep internal repeats (forward or reverse-complemented copies)"""
    rng = np.random.default_rng(seed)
    pg = rng.choice(ACGT, size=G)
    for _ in range(nrep):
        L = int(rng.integers(30, replen))
        s, d = int(rng.integers(0, G - L)), int(rng.integers(0, G - L))
        seg = pg[s:s + L].copy()
        if rng.random() < rc_frac:
            seg = orc.revcomp_ascii(seg)
        pg[d:d + L] = seg
    return pg


def make_pair(seed, G=200000, G2=60000, with_n=False, low_complexity=False):
    """(src, other): `other` carries copies of src segments on both strands, the src ends at its own ends,
    optionally N runs (the N pseudogenome) and low-complexity tracts in both texts"""
    src = make_pg(G + seed * 1013, seed, nrep=200)
    other = make_pg(G2, 100 + seed, nrep=10)
    rng = np.random.default_rng(seed)
    if low_complexity:
        src[5000:9000] = ord("A")
        src[20000:23000] = np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 3000)
        other[1000:2500] = ord("T")
        other[7000:8200] = np.resize(np.frombuffer(b"CGT", dtype=np.uint8), 1200)
    for _ in range(60):
        L = int(rng.integers(40, 600))
        s, d = int(rng.integers(0, src.size - L)), int(rng.integers(0, other.size - L))
        seg = src[s:s + L].copy()
        if rng.random() < 0.5:
            seg = orc.revcomp_ascii(seg)
        other[d:d + L] = seg
    other[:300] = orc.revcomp_ascii(src[-300:])
    other[-200:] = orc.revcomp_ascii(src[:200])
    if with_n:
        for _ in range(40):
            p = int(rng.integers(0, other.size - 3))
            other[p:p + int(rng.integers(1, 4))] = ord("N")
    return src, other


COMBOS = ((0, 1), (1, 1), (0, 0), (1, 0))   # (dest_is_src, rev_compl); the encoder uses the first two
