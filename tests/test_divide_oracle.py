"""Row f3 on the CPU: the oracle's restatement of the read-set division (DividedPCLReadsSets.cpp:59-100) against the
compiled reference run over the same in-memory records."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
from divide_util import (COMBOS, make_fastq, make_records, oracle_divide, oracle_divide_fastq, oracle_fastq_rows, ref_divide,
                         ref_divide_files, same)

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


@needs_ref
@pytest.mark.parametrize("paired,rev", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("crlf,trailing", [(False, True), (True, True), (False, False)])
def test_oracle_fastq_division_equals_reference_on_files(tmp_path, paired, rev, crlf, trailing):
    """the oracle's line reader + division against the reference's managed FASTQ iterator feeding its factory, on real files:
    identifier lines of varying length, CR LF line ends, a last line without a newline, two files read in turn with the second
    file's reads reverse-complemented"""
    L, n = 100, 1500
    reads, quals = make_records(seed=3, n=n, L=L)
    a = make_fastq(reads[0::2] if paired else reads, quals[0::2] if paired else quals, seed=1, crlf=crlf, trailing_newline=trailing)
    b = make_fastq(reads[1::2], quals[1::2], seed=2, crlf=crlf, trailing_newline=trailing) if paired else None
    (tmp_path / "a.fq").write_bytes(a)
    if paired:
        (tmp_path / "b.fq").write_bytes(b)
    for combo in (COMBOS[1], COMBOS[5], COMBOS[8]):
        o = oracle_divide_fastq(a, b, rev, L, combo)
        r = ref_divide_files(tmp_path / "a.fq", (tmp_path / "b.fq") if paired else None, rev, L, n, combo)
        assert same(o, r) is None, (combo, same(o, r))
        assert o["n_hq"] + o["n_lq"] + o["n_n"] == n


@needs_ref
def test_oracle_fastq_uneven_pair_and_cut_off_record(tmp_path):
    """what the reference's iterator does at the ends: the second file one record short (it stops after the first file's
    extra read); a cut-off last record is reported"""
    L = 64
    reads, quals = make_records(seed=9, n=41, L=L)
    a, b = make_fastq(reads[0::2], quals[0::2], seed=1), make_fastq(reads[1::2], quals[1::2], seed=2)     # 21 and 20 records
    (tmp_path / "a.fq").write_bytes(a)
    (tmp_path / "b.fq").write_bytes(b)
    combo = (0.2, False, True, False)
    o = oracle_divide_fastq(a, b, True, L, combo)
    assert o["n_hq"] + o["n_lq"] + o["n_n"] == 41
    assert same(o, ref_divide_files(tmp_path / "a.fq", tmp_path / "b.fq", True, L, 50, combo)) is None
    cut = make_fastq(reads[:10], quals[:10], seed=4)
    cut = cut[: cut.rstrip(b"\n").rfind(b"\n+")]              # drop the '+' line and the qualities of the last record
    assert oracle_divide_fastq(cut, None, False, L, combo) is None      # (reported, not guessed: see pgrc_or_fastq_records)
