#!/usr/bin/env python3
"""One-off randomized sweep on the CPU (not part of the suite): the oracle's restatements of the HIP path's exact
shortcuts -- the early-stop rule, the screened schedule, the dual-strand query -- against the reference order on random
configurations, a third of them low-complexity texts with reads from both strands.
usage: python tests/sweep_rules.py <seed> <seconds>      (round 2: seeds 1 and 2, 500 s each: 398 + 398 cases, no difference)"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import oracle as orc
import test_early_stop_rule as t
from util import make_inputs
rng = np.random.default_rng(int(sys.argv[1]))
t0=time.time(); n_cases=0; ab=0
while time.time()-t0 < float(sys.argv[2]):
    L = int(rng.integers(40, 256))
    seed_len = int(rng.integers(24, min(L, 140) + 1))
    M = int(rng.choice([1000, 60, 50, 25, 10, 4, 3]))
    kmax = min(L // M, 247)
    kmin = kmax if rng.random() < 0.15 else 0
    G = int(rng.integers(L + 50, 200000))
    n = int(rng.integers(1, 3000))
    pg, reads = make_inputs(G, n, L, seed=int(rng.integers(0, 1 << 30)), pool_div=int(rng.choice([8, 64])), tandem_every=int(rng.choice([0, 2, 64])))
    # now and then a low-complexity text (period of a few dozen) with reads from both strands
    if rng.random() < 0.3:
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        unit = rng.choice(acgt, size=int(rng.integers(5, 60)))
        pg = np.tile(unit, G // unit.size + 1)[:G].copy()
        fl = rng.integers(0, G, size=max(1, G // 150)); pg[fl] = rng.choice(acgt, size=fl.size)
        comp = np.zeros(256, dtype=np.uint8)
        for a_, b_ in zip(b"ACGT", b"TGCA"): comp[a_] = b_
        for i in range(min(n, 1500)):
            st = int(rng.integers(0, G - L))
            w = pg[st:st+L].copy()
            if rng.random() < 0.5: w = comp[w[::-1]]
            for _ in range(int(rng.integers(0, 4))): w[int(rng.integers(0, L))] = rng.choice(acgt)
            reads[i] = w
    t._both(pg, reads, seed_len, kmax, kmin, True)
    n_cases += 1
    if kmin == 0: ab += t._both.last_aborted
print("sweep ok", n_cases, "cases; reads done in the reference's order by the dual scheme:", ab, flush=True)
