"""The early-stop rule of the HIP match kernel (pgrc_amd/csrc/copmem.hip, "Early stop"), checked on the CPU: the
oracle's restatement of the reference's per-read query (CopMEMMatcher.cpp:483-566) is run with its loops as the
reference has them and with the rule switched on (a test switch of the oracle); positions, strands and mismatch counts
must not differ on any input, while the number of executed seed probes must fall."""
import numpy as np
import pytest

import oracle as orc
from util import make_inputs


def _both(pg, reads, seed_len, kmax, kmin, rev=True):
    lib = orc.oracle()
    lib.pgrc_or_set_early_stop(0)
    lib.pgrc_or_probe_count(1)
    full = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin, rev)
    p_full = lib.pgrc_or_probe_count(1)
    lib.pgrc_or_set_early_stop(1)
    try:
        early = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin, rev)
    finally:
        lib.pgrc_or_set_early_stop(0)
    p_early = lib.pgrc_or_probe_count(1)
    for k in ("pos", "rc", "mism", "hist"):
        assert np.array_equal(np.asarray(full[k]), np.asarray(early[k])), k
    assert full["matched"] == early["matched"]
    if rev:
        # ... and the HIP path's schedule of a two-pass run (exact-match screen on the RC text first)
        lib.pgrc_or_probe_count(1)
        scr = orc.oracle_match_screened(pg, reads, seed_len, kmax, kmin)
        p_scr = lib.pgrc_or_probe_count(1)
        for k in ("pos", "rc", "mism", "hist"):
            assert np.array_equal(np.asarray(full[k]), np.asarray(scr[k])), ("screened", k)
        assert full["matched"] == scr["matched"]
        _both.last_screened = p_scr
        if kmin == 0:
            # ... and ONE query over both strands at once (the dual kernel's scheme)
            lib.pgrc_or_probe_count(1)
            du = orc.oracle_match_dual(pg, reads, seed_len, kmax)
            _both.last_dual = lib.pgrc_or_probe_count(1)
            _both.last_aborted = du["aborted"]
            for k in ("pos", "rc", "mism", "hist"):
                assert np.array_equal(np.asarray(full[k]), np.asarray(du[k])), ("dual", k)
            assert full["matched"] == du["matched"]
            # ... with a speculative first attempt at a small limit (round 4; 3 is the kernel's default, 0 = exact only)
            for small in (3, 0, 1):
                lib.pgrc_or_set_dual_spec(small + 1)
                try:
                    ds = orc.oracle_match_dual(pg, reads, seed_len, kmax)
                finally:
                    lib.pgrc_or_set_dual_spec(0)
                for k in ("pos", "rc", "mism", "hist"):
                    assert np.array_equal(np.asarray(full[k]), np.asarray(ds[k])), ("dual, first attempt at", small, k)
    return p_full, p_early


CASES = [  # L, seed_len, M (kmax = L // M), kmin = kmax?, G, n, pool_div, tandem_every
    (150, 38, 50, False, 400000, 6000, 8, 2),
    (100, 38, 50, False, 200000, 6000, 8, 2),
    (250, 38, 50, False, 300000, 3000, 8, 2),
    (100, 38, 3, False, 150000, 3000, 8, 2),      # k <= 33: limits far above the number of rounds
    (150, 45, 50, False, 300000, 4000, 64, 0),
    (64, 32, 10, False, 100000, 4000, 8, 64),
    (100, 24, 25, False, 100000, 4000, 8, 2),
    (150, 38, 50, True, 300000, 4000, 8, 2),      # kmin == kmax: the first acceptable alignment ends the read
    (255, 100, 20, False, 200000, 2000, 8, 2),
    (120, 64, 30, False, 200000, 3000, 64, 2),
]


@pytest.mark.parametrize("L,seed_len,M,kmin_is_kmax,G,n,pool_div,tandem", CASES)
def test_early_stop_rule_changes_nothing(L, seed_len, M, kmin_is_kmax, G, n, pool_div, tandem):
    pg, reads = make_inputs(G, n, L, seed=L * 1000 + seed_len + M, pool_div=pool_div, tandem_every=tandem)
    kmax = L // M
    p_full, p_early = _both(pg, reads, seed_len, kmax, kmax if kmin_is_kmax else 0)
    assert p_early <= p_full
    if kmax <= 3 and not kmin_is_kmax:
        assert p_early < 0.8 * p_full            # the typical configuration saves a lot


def test_early_stop_rule_on_capped_and_truncated_buckets():
    """Low-complexity text: buckets at the 13-entry cap and reads whose falses budget runs out -- the rounds such
    probes fall into do not count, and results stay the same."""
    rng = np.random.default_rng(5)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=37)
    pg = np.tile(unit, 3000)[:100000].copy()
    flips = rng.integers(0, pg.size, size=600)
    pg[flips] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=flips.size)
    _, reads = make_inputs(100000, 3000, 100, seed=3)
    starts = rng.integers(0, pg.size - 100, size=2000)
    for i, st in enumerate(starts):
        reads[i] = pg[st:st + 100]
        for _ in range(int(rng.integers(0, 4))):
            reads[i, int(rng.integers(0, 100))] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8))
    for kmax in (2, 5, 33):
        _both(pg, reads, 38, kmax, 0)


def test_early_stop_rule_random_sweep():
    rng = np.random.default_rng(99)
    for _ in range(25):
        L = int(rng.integers(40, 256))
        seed_len = int(rng.integers(24, min(L, 140) + 1))
        M = int(rng.choice([1000, 60, 50, 25, 10, 4, 3]))
        kmax = min(L // M, 247)
        kmin = kmax if rng.random() < 0.25 else 0
        G = int(rng.integers(L + 50, 150000))
        n = int(rng.integers(1, 2500))
        pg, reads = make_inputs(G, n, L, seed=int(rng.integers(0, 1 << 30)), pool_div=int(rng.choice([8, 64])),
                                tandem_every=int(rng.choice([0, 2, 64])))
        _both(pg, reads, seed_len, kmax, kmin, bool(rng.random() < 0.8))
