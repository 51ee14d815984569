"""Pg-vs-Pg exact matching (row f2): the oracle restatement of CopMEMMatcher::matchTexts against the compiled
reference -- same matches, in the same discovery order."""
import numpy as np
import pytest

import oracle as orc
from mem_util import COMBOS, make_pair, mem_sweep_cases, mem_sweep_texts

pytestmark = pytest.mark.skipif(not (orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_mem_match")),
                                reason="needs oracle/_ref with the text-matcher harness")


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("with_n,low_complexity", [(False, False), (True, False), (False, True)])
def test_mem_oracle_equals_reference(seed, with_n, low_complexity):
    src, other = make_pair(seed, with_n=with_n, low_complexity=low_complexity)
    for dest_is_src, rev_compl in COMBOS:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl)
        r = orc.ref_mem_match(src, d, dest_is_src, rev_compl)
        assert len(r) > 20
        assert np.array_equal(o, r), (seed, dest_is_src, rev_compl)


@pytest.mark.parametrize("target_len,min_len", [(45, 45), (45, 60), (50, 50), (36, 36), (64, 64), (120, 120)])
def test_mem_oracle_other_lengths(target_len, min_len):
    src, other = make_pair(7, G=150000, G2=50000)
    for dest_is_src, rev_compl in COMBOS[:2]:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl, target_len, min_len)
        r = orc.ref_mem_match(src, d, dest_is_src, rev_compl, target_len, min_len)
        assert np.array_equal(o, r), (target_len, min_len, dest_is_src)


def test_mem_short_texts():
    src, other = make_pair(3, G=30000, G2=2000)
    for n2 in (31, 32, 33, 45, 100, 400, 767, 768, 769, 800, 2000):
        d = orc.mem_dest(src, other[:n2], 0, 1)
        assert np.array_equal(orc.oracle_mem_match(src, d, 0, 1), orc.ref_mem_match(src, d, 0, 1)), n2


@pytest.mark.parametrize("target,min_len,seed", mem_sweep_cases())
def test_mem_oracle_sweep(target, min_len, seed):
    """the texts of the GPU sweep (tests/test_gpu_fuzz.py::test_mem_sweep): the oracle is pinned on them too"""
    src, other = mem_sweep_texts(seed, target)
    for dest_is_src, rev_compl in COMBOS:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl, target, min_len)
        r = orc.ref_mem_match(src, d, dest_is_src, rev_compl, target, min_len)
        assert np.array_equal(o, r), (target, min_len, seed, dest_is_src, rev_compl)
