#!/usr/bin/env python3
"""Modes d / i / e at a middle size (L = 150, seed 38, k <= 3) against the oracle: results and the number of (window, part) pairs with
equal keys per strand -- the one scan of the forward text must find exactly the candidates of the oracle's two scans.
usage: python tools/modes_vs_oracle.py <pg length> <reads> <mode> [<mode> ...]   (e.g. 30000000 2000000 d i e: ~1 min of oracle per mode)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
from util import make_inputs, gpu_match
G, n, L = int(sys.argv[1]), int(sys.argv[2]), 150
for mode in sys.argv[3:]:
    pg, reads = make_inputs(G, n, L, seed=777)
    t = time.time(); o = orc.oracle_match(mode, pg, reads, 38, 3, 0); to = time.time() - t
    g = gpu_match(mode, pg, reads, 38, 3, 0)
    bad = np.flatnonzero((g["pos"] != o["pos"]) | (g["rc"] != o["rc"]) | (g["mism"] != o["mism"]))
    print(mode, "G", G, "n", n, "oracle s %.1f" % to, "matched", g["matched"], o["matched"], "bad", bad.size, "candidates gpu", g["ctx"].counters()["candidates"], "oracle", o["candidates"], flush=True)
    for i in bad[:8]:
        print("  read", i, "gpu", g["pos"][i], g["rc"][i], g["mism"][i], "oracle", o["pos"][i], o["rc"][i], o["mism"][i])
    if bad.size:
        print("  oracle rc of bad:", np.bincount(o["rc"][bad], minlength=2), "gpu rc of bad:", np.bincount(g["rc"][bad], minlength=2),
              "oracle mism of bad:", np.bincount(np.minimum(o["mism"][bad], 9), minlength=10))
