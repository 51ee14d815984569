"""Loads tests/golden/mem_cases.npz (row f2 fixtures, tests/golden/make_golden_mem.py) and re-derives the texts."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_mem import MEM_CASES  # noqa: E402  (parameters only; importing does not need the reference)
from mem_util import COMBOS, make_pair  # noqa: E402

NAMES = [c[0] for c in MEM_CASES]


def load(name):
    z = np.load(os.path.join(HERE, "golden", "mem_cases.npz"))
    _, seed, G, G2, with_n, lowc, tl, ml = next(c for c in MEM_CASES if c[0] == name)
    src, other = make_pair(seed, G=G, G2=G2, with_n=with_n, low_complexity=lowc)
    for arr, key in ((src, "src_digest"), (other, "other_digest")):
        want = bytes(z[f"{name}/{key}"]).decode()
        got = hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()[:16]
        assert got == want, f"text generator drifted for {name}/{key}"
    expected = {(dis, rc): z[f"{name}/{dis}{rc}"] for dis, rc in COMBOS}
    return src, other, tl, ml, expected
