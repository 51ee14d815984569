"""Row f3 on the CPU: the oracle's restatement of the read-set division (DividedPCLReadsSets.cpp:59-100) against the
compiled reference run over the same in-memory records."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
from divide_util import COMBOS, make_records, oracle_divide, ref_divide, same

needs_ref = pytest.mark.skipif(not (orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_divide")), reason="needs oracle/_ref with the division harness")


@needs_ref
def test_quality_table_formula_equals_the_reference_table():
    """qualityLut (utils/helper.cpp:284-327) is written there as decimal literals; the oracle and the library derive it
    as (float) (1 - 10^(-q / 10)): the same floats"""
    o, r = orc.oracle(), orc.ref()
    o.pgrc_or_quality_lut.restype = C.c_float
    o.pgrc_or_quality_lut.argtypes = [C.c_int]
    r.pgrc_ref_quality_lut.restype = C.c_float
    r.pgrc_ref_quality_lut.argtypes = [C.c_int]
    for c in range(133):
        assert o.pgrc_or_quality_lut(c) == r.pgrc_ref_quality_lut(c), c


@needs_ref
@pytest.mark.parametrize("L", [37, 100, 150, 151])
@pytest.mark.parametrize("combo", COMBOS)
def test_oracle_division_equals_reference(L, combo):
    error_limit, simplified, separate_n, n_reads_lq = combo
    reads, quals = make_records(seed=L * 31 + int(error_limit * 1000), n=3000, L=L)
    o = oracle_divide(reads, quals, *combo)
    r = ref_divide(reads, quals, *combo)
    assert same(o, r) is None, same(o, r)
    assert o["n_hq"] + o["n_lq"] + o["n_n"] == 3000
    if error_limit < 1:
        assert 0 < o["n_lq"] < 3000                          # the quality test cuts somewhere in the middle
    if separate_n:
        assert o["n_n"] > 0 and o["symbols"] == (4, 4, 5)


@needs_ref
def test_oracle_division_without_quality_rows():
    reads, _ = make_records(seed=5, n=2000, L=100)
    for combo in COMBOS[:4]:
        assert same(oracle_divide(reads, None, *combo), ref_divide(reads, None, *combo)) is None


def test_oracle_division_properties():
    """without the compiled reference: every read lands in exactly one set, rows unpack to the reads, indexes ascend"""
    reads, quals = make_records(seed=11, n=1500, L=150)
    o = oracle_divide(reads, quals, 0.05, False, True, False)
    assert o["n_hq"] + o["n_lq"] + o["n_n"] == 1500
    assert np.all(np.diff(o["lq_index"].astype(np.int64)) > 0) and np.all(np.diff(o["n_index"].astype(np.int64)) > 0)
    has_n = (reads == ord("N")).any(axis=1)
    assert np.array_equal(np.flatnonzero(has_n), o["n_index"])
    lib = orc.oracle()
    lib.pgrc_or_unpack_read.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p]
    back = np.zeros(150, np.uint8)
    rb = o["row_bytes"][2]
    for k, i in enumerate(o["n_index"][:50]):
        row = np.ascontiguousarray(o["n_rows"][k * rb:(k + 1) * rb])
        lib.pgrc_or_unpack_read(row.ctypes.data, 150, b"ACGNT", back.ctypes.data)
        assert np.array_equal(back, reads[i])
