#!/usr/bin/env python3
"""Golden fixtures for row f2 (Pg-vs-Pg exact matching) from the REAL reference compiled in the build container
(oracle/_ref/libpgrc_ref.so): CopMEMMatcher(src, targetLen).matchTexts(...) outputs in discovery order.  Fixtures are
data only: the generator parameters (texts are re-derived by tests/mem_util.py from numpy's PCG64 streams; a digest
of each text is stored so that a drifting generator is noticed) + the reference's match triples.

    python tests/golden/make_golden_mem.py        # needs /root/reference (run `make -C oracle ref` first)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
from mem_util import COMBOS, make_pair  # noqa: E402

# (name, seed, G, G2, with_n, low_complexity, target_len, min_len)
MEM_CASES = [
    ("mem_plain", 21, 120000, 40000, False, False, 45, 45),
    ("mem_n", 22, 120000, 40000, True, False, 45, 45),
    ("mem_lowcomplexity", 23, 120000, 40000, False, True, 45, 45),
    ("mem_L50_min60", 24, 100000, 30000, True, True, 50, 60),
    ("mem_L36", 25, 100000, 30000, False, False, 36, 36),
]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    out = {}
    for name, seed, G, G2, with_n, lowc, tl, ml in MEM_CASES:
        src, other = make_pair(seed, G=G, G2=G2, with_n=with_n, low_complexity=lowc)
        out[name + "/src_digest"] = np.frombuffer(digest(src).encode(), dtype=np.uint8)
        out[name + "/other_digest"] = np.frombuffer(digest(other).encode(), dtype=np.uint8)
        for dis, rc in COMBOS:
            d = orc.mem_dest(src, other, dis, rc)
            r = orc.ref_mem_match(src, d, dis, rc, tl, ml)
            out[f"{name}/{dis}{rc}"] = r
            print(name, dis, rc, len(r))
    np.savez_compressed(os.path.join(HERE, "mem_cases.npz"), **out)


if __name__ == "__main__":
    main()
