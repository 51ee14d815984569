"""The CPU oracle against the committed golden vectors (outputs of the real reference, generated in the build
container by tests/golden/make_golden.py).  Runs everywhere, including where /root/reference is absent."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as orc
from golden_util import GOLD, HARD_CASES, INDEX_CASES, MATCH_CASES, cumm_to_sparse, load_case, load_hard_case, load_index_case
from util import assert_same_results


@pytest.mark.parametrize("name", MATCH_CASES)
def test_oracle_reproduces_reference_output(name):
    m, pg, reads, kind, sl, kmax, kmin, gold = load_case(name)
    o = orc.oracle_match(kind, pg, reads, sl, kmax, kmin, m["rev_compl"])
    assert_same_results(o, gold, name)


@pytest.mark.parametrize("name", HARD_CASES)
def test_oracle_and_its_restated_schedules_reproduce_reference_output_on_hard_inputs(name):
    """L = 150, seed 38: repeat families, reverse palindromes, short-period texts, reads from both strands.  The oracle in
    the reference's order, and its restatements of the HIP path's exact shortcuts (early stop; exact-match screen; one
    query over both strands), all against the real reference's output."""
    m, pg, reads, gold = load_hard_case(name)
    lib = orc.oracle()
    lib.pgrc_or_set_early_stop(0)
    assert_same_results(orc.oracle_match("c", pg, reads, m["seed_len"], m["kmax"], 0, True), gold, name)
    lib.pgrc_or_set_early_stop(1)
    try:
        assert_same_results(orc.oracle_match("c", pg, reads, m["seed_len"], m["kmax"], 0, True), gold, name + " early stop")
    finally:
        lib.pgrc_or_set_early_stop(0)
    assert_same_results(orc.oracle_match_screened(pg, reads, m["seed_len"], m["kmax"], 0), gold, name + " screened")
    du = orc.oracle_match_dual(pg, reads, m["seed_len"], m["kmax"])
    assert_same_results(du, gold, name + " dual")
    if "repeat" in name or "period" in name:
        assert du["aborted"] > 0            # reads whose falses budget matters: redone in the reference's order


@pytest.mark.parametrize("name", INDEX_CASES)
def test_oracle_index_reproduces_reference_index(name):
    m, pg, positions, buckets, counts = load_index_case(name)
    prm, cumm, pos = orc.oracle_index(pg, m["seed_len"])
    assert prm == {k: m[k] for k in ("K", "k1", "k2", "hash_size")}
    nz, cnt = cumm_to_sparse(cumm)
    assert np.array_equal(pos, positions) and np.array_equal(nz, buckets) and np.array_equal(cnt, counts)
    assert counts.max() == 13  # the poly-A tract overflows a bucket: cap exercised


def test_oracle_mismatch_lists_reproduce_reference():
    from golden_util import MANIFEST  # noqa: F401
    import make_golden as mg
    pg, reads = mg.case_inputs(mg.CASES[8])
    z = np.load(os.path.join(GOLD, "extract_c_nreads_M3.npz"))
    with open(os.path.join(GOLD, "extract_c_nreads_M3.json")) as f:
        rows = json.load(f)
    assert len(rows) > 500
    for i, pair_file, codes, offs in rows:
        rc = int(z["rc"][i])
        reversed_ = (rc != (i & 1)) if pair_file else bool(rc)
        co, oo = orc.oracle_extract(pg, z["pos"][i], reads[i], rc, reversed_, int(z["mism"][i]))
        assert co.tolist() == codes and oo.tolist() == offs, (i, pair_file)


# ---- known-answer values captured from the compiled reference during the survey

def lcg_pg(n=100000):
    s = 12345
    out = bytearray(n)
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = b"ACGT"[(s >> 33) & 3]
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def test_appendix_c_hash_values():
    h = orc.oracle().pgrc_or_copmem_hash(28, b"ACGTACGTACGTACGTACGTACGTACGT")
    assert h == 0x01F1BD14
    pg = lcg_pg()
    assert pg[:28].tobytes() == b"ATGCAGTGGCTCCACATTACGATTGCCA"
    assert orc.oracle().pgrc_or_copmem_hash(28, pg[0:28].tobytes()) == 0xB8E2FC85
    assert orc.oracle().pgrc_or_copmem_hash(28, pg[5:33].tobytes()) == 0x1F4C265F


def test_appendix_c_index_and_queries():
    pg = lcg_pg()
    prm, cumm, positions = orc.oracle_index(pg, 38)
    assert prm == {"K": 28, "k1": 5, "k2": 2, "hash_size": 1 << 24}
    assert positions.size == 19995
    nz, _ = cumm_to_sparse(cumm)
    assert list(nz[:3]) == [313, 610, 620]
    assert positions[cumm[313]] == 27995 and positions[cumm[610]] == 87645 and positions[cumm[620]] == 80665
    fnv = 1469598103934665603
    for v in positions.tolist():
        fnv = ((fnv ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert fnv == 0x88D100CA3AC03E4E

    idx = orc.Index()
    assert orc.oracle().pgrc_or_index_build(pg.ctypes.data_as(C.c_void_p), pg.size, 38, C.byref(idx)) == 0

    def query(read, kmax):
        cnt = C.c_uint8(255)
        falses, cands = C.c_uint64(0), C.c_uint64(0)
        p = orc.oracle().pgrc_or_copmem_match_read(C.byref(idx), pg.ctypes.data_as(C.c_void_p),
                                                   read.ctypes.data_as(C.c_void_p), read.size, kmax, 0, C.byref(cnt),
                                                   C.byref(falses), C.byref(cands))
        return p, cnt.value, falses.value

    def sub(read, off):
        read[off] = ord("A") if read[off] != ord("A") else ord("C")

    r = pg[777:877].copy(); sub(r, 10); sub(r, 60)
    assert query(r, 33) == (777, 2, 3)
    r = pg[777:877].copy(); sub(r, 10)
    assert query(r, 0) == (orc.NOT_MATCHED_POS, 255, 6)
    r = pg[777:877].copy(); sub(r, 98)
    assert query(r, 0) == (orc.NOT_MATCHED_POS, 255, 14)  # tail rejects are counted twice
    orc.oracle().pgrc_or_index_free(C.byref(idx))


@pytest.mark.parametrize("seed_len,K,k1,k2,entries", [
    (24, 20, 5, 1, 19997), (28, 24, 5, 1, 19996), (32, 28, 5, 1, 19995), (38, 28, 5, 2, 19995), (45, 32, 4, 3, 24993),
    (54, 40, 5, 3, 19993), (64, 44, 5, 4, 19992), (100, 44, 8, 7, 12495), (150, 56, 10, 9, 9995), (250, 56, 14, 13, 7139)])
def test_appendix_b_copmem_parameters(seed_len, K, k1, k2, entries):
    p = orc.CopmemParams()
    assert orc.oracle().pgrc_or_copmem_derive(seed_len, 100000, C.byref(p)) == 0
    assert (p.K, p.k1, p.k2, p.hash_size) == (K, k1, k2, 1 << 24)
    assert (100000 - K) // k1 + 1 == entries  # "sampled positions" column (no bucket overflows on random text)


def test_map_params_derivation():
    mp = orc.MapParams()
    assert orc.oracle().pgrc_or_map_derive(100, 38, 50, b"c", C.byref(mp)) == 0
    assert (mp.kmax, mp.kmin, mp.seed_len, mp.parts, mp.matcher) == (2, 0, 38, 2, b"c")
    assert orc.oracle().pgrc_or_map_derive(150, 38, 3, b"C", C.byref(mp)) == 0
    assert (mp.kmax, mp.kmin, mp.matcher) == (50, 50, b"c")
    assert orc.oracle().pgrc_or_map_derive(100, 100, 50, b"d", C.byref(mp)) == 0
    assert mp.matcher == b"e"
    assert orc.oracle().pgrc_or_map_derive(100, 200, 50, b"c", C.byref(mp)) == 0
    assert (mp.seed_len, mp.matcher) == (100, b"c")
    assert orc.oracle().pgrc_or_map_derive(100, 38, 50, b"x", C.byref(mp)) == 2


# ---- row f2: Pg-vs-Pg exact matching

import mem_golden_util as mg  # noqa: E402


@pytest.mark.parametrize("name", mg.NAMES)
def test_oracle_mem_match_reproduces_reference_output(name):
    src, other, tl, ml, expected = mg.load(name)
    for (dis, rc), want in expected.items():
        got = orc.oracle_mem_match(src, orc.mem_dest(src, other, dis, rc), dis, rc, tl, ml)
        assert len(want) > 5 and np.array_equal(got, want), (name, dis, rc)


# ---- row f1: the export streams of the reference (tests/golden/export_*.npz)

import export_util as xu  # noqa: E402


@pytest.mark.parametrize("name", xu.EXPORT_GOLDEN)
def test_oracle_export_reproduces_reference_streams(name):
    case, pair, kmax, res, order, gold = xu.load_export_golden(name)
    # the committed match results are what the oracle computes for these inputs
    o = orc.oracle_match("c", case["pg"], case["reads"], 38, kmax, 0)
    for k in ("pos", "rc", "mism"):
        assert np.array_equal(o[k], res[k]), k
    got = xu.stream_bytes(xu.oracle_export_pg_order(case, res, order, pair_file=pair))
    for k in xu.STREAMS:
        assert got[k] == gold["pg"][k], (name, k)
    n = case["reads"].shape[0]
    er, eo = xu.original_order_entries(case["read_org"], res["mism"] != 255, case["total"], pair, n - case["n_n"])
    got = xu.stream_bytes(xu.oracle_export_entries(case, res, er, eo, pair_file=pair))
    for k in xu.STREAMS:
        assert got[k] == gold["org"][k], (name, "original order", k)
