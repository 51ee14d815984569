"""Randomized sweep on the CPU: the oracle's restatements of the HIP path's exact shortcuts -- the early-stop rule, the
screened schedule, the dual-strand query -- against the reference order on random configurations, a third of them
low-complexity texts with reads from both strands.  `tests/test_sweep_rules.py` runs a fixed-seed slice of it in the CPU
suite; as a script it goes on for as long as asked over the same seed sequence:
    python tests/sweep_rules.py <seconds>      (round 2: 2 x 500 s, 796 cases; round 3: see profiles/)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import test_early_stop_rule as t  # noqa: E402
from util import make_inputs  # noqa: E402

SEEDS = (1, 2, 3, 5, 8, 13, 21, 34)     # the suite takes the first cases of each; the script cycles through them


def one_case(rng):
    """draws one configuration and checks the three restatements against the reference order; returns the number of
    reads the dual scheme had to redo in the reference's order (0 unless kmin == 0)"""
    L = int(rng.integers(40, 256))
    seed_len = int(rng.integers(24, min(L, 140) + 1))
    M = int(rng.choice([1000, 60, 50, 25, 10, 4, 3]))
    kmax = min(L // M, 247)
    kmin = kmax if rng.random() < 0.15 else 0
    G = int(rng.integers(L + 50, 200000))
    n = int(rng.integers(1, 3000))
    pg, reads = make_inputs(G, n, L, seed=int(rng.integers(0, 1 << 30)), pool_div=int(rng.choice([8, 64])),
                            tandem_every=int(rng.choice([0, 2, 64])))
    if rng.random() < 0.3:
        # a low-complexity text (period of a few dozen symbols, a few substitutions) with reads from both strands
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        unit = rng.choice(acgt, size=int(rng.integers(5, 60)))
        pg = np.tile(unit, G // unit.size + 1)[:G].copy()
        fl = rng.integers(0, G, size=max(1, G // 150))
        pg[fl] = rng.choice(acgt, size=fl.size)
        comp = np.zeros(256, dtype=np.uint8)
        for a_, b_ in zip(b"ACGT", b"TGCA"):
            comp[a_] = b_
        for i in range(min(n, 1500)):
            st = int(rng.integers(0, G - L))
            w = pg[st:st + L].copy()
            if rng.random() < 0.5:
                w = comp[w[::-1]]
            for _ in range(int(rng.integers(0, 4))):
                w[int(rng.integers(0, L))] = rng.choice(acgt)
            reads[i] = w
    t._both(pg, reads, seed_len, kmax, kmin, True)
    return t._both.last_aborted if kmin == 0 else 0


def sweep(seed, max_cases=None, seconds=None):
    rng = np.random.default_rng(seed)
    t0, cases, redone = time.time(), 0, 0
    while (max_cases is None or cases < max_cases) and (seconds is None or time.time() - t0 < seconds):
        redone += one_case(rng)
        cases += 1
    return cases, redone


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    t0, total, redone, k = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        c, r = sweep(SEEDS[k % len(SEEDS)] + 1000 * (k // len(SEEDS)), seconds=min(60.0, budget - (time.time() - t0)))
        total, redone, k = total + c, redone + r, k + 1
    print("sweep ok", total, "cases; reads done in the reference's order by the dual scheme:", redone, flush=True)
