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


# ---- differential sweep (tests/test_gpu_fuzz.py::test_mem_sweep, tests/test_mem_oracle_vs_ref.py)

def mem_sweep_cases():
    rng = np.random.default_rng(20260202)
    out = []
    for target in (24, 27, 28, 31, 33, 36, 38, 43, 45, 47, 50, 54, 60, 63, 64, 90, 111, 130, 200, 255):
        out.append((target, target + int(rng.integers(0, 3)) * int(rng.integers(0, 30)), int(rng.integers(0, 1 << 30))))
    for _ in range(12):
        out.append((45, 45, int(rng.integers(0, 1 << 30))))
    return out


def mem_sweep_texts(seed, target):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    G = int(rng.integers(20000, 90000))
    src = rng.choice(acgt, size=G)
    style = int(rng.integers(0, 4))
    if style == 1:      # tandem / periodic stretches
        for _ in range(6):
            unit = rng.choice(acgt, size=int(rng.integers(1, 9)))
            s, ln = int(rng.integers(0, G - 3000)), int(rng.integers(100, 2500))
            src[s:s + ln] = np.resize(unit, ln)
    for _ in range(int(rng.integers(5, 60))):       # internal repeats, both strands
        ln = int(rng.integers(target - 5, 6 * target))
        s, d = int(rng.integers(0, G - ln)), int(rng.integers(0, G - ln))
        seg = src[s:s + ln].copy()
        src[d:d + ln] = orc.revcomp_ascii(seg) if rng.random() < 0.5 else seg
    G2 = int(rng.integers(1, 4)) * 768 + int(rng.integers(-3, 4)) if style == 3 else int(rng.integers(target, 30000))
    other = rng.choice(acgt, size=max(G2, 1))
    for _ in range(int(rng.integers(3, 40))):
        ln = int(rng.integers(target - 5, 8 * target))
        if ln >= other.size:
            continue
        s, d = int(rng.integers(0, G - ln)), int(rng.integers(0, other.size - ln))
        seg = src[s:s + ln].copy()
        other[d:d + ln] = orc.revcomp_ascii(seg) if rng.random() < 0.5 else seg
    if style == 2 and other.size > 10:
        for _ in range(int(rng.integers(1, 30))):
            p = int(rng.integers(0, other.size - 3))
            other[p:p + int(rng.integers(1, 4))] = ord("N")
    if other.size > 2 * target:                     # the source's ends, where the side-context registers go stale
        other[:target + 10] = orc.revcomp_ascii(src[-(target + 10):])
        other[-(target + 10):] = src[:target + 10]
    return src, other
