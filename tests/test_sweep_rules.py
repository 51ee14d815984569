"""A fixed-seed slice of tests/sweep_rules.py in the CPU suite: the three exact shortcuts of the HIP match path (early
stop, exact-match screen, one query over both strands), as the oracle restates them, against the reference order
(CopMEMMatcher.cpp:483-566 under ReadsMatchers.cpp:421-451) on random configurations.  Every case asserts inside
`test_early_stop_rule._both`; the time guard only bounds the suite (about a minute in all), it never hides a failure."""
import time

import pytest

import sweep_rules


@pytest.mark.parametrize("seed", sweep_rules.SEEDS)
def test_restated_shortcuts_equal_the_reference_order(seed):
    cases, _ = sweep_rules.sweep(seed, max_cases=8, seconds=9.0)
    assert cases >= 1


def test_sweep_reaches_reads_the_dual_scheme_redoes():
    """the slice must include reads that run out of their falses budget (the dual scheme then redoes them in the
    reference's order): low-complexity texts make them"""
    t0, redone, k = time.time(), 0, 0
    while redone == 0 and time.time() - t0 < 30:
        redone += sweep_rules.sweep(100 + k, max_cases=4)[1]
        k += 1
    assert redone > 0
