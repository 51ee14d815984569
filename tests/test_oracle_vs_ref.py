"""Pins the CPU restatement (oracle/pgrc_oracle.c) to the REAL reference compiled in this container
(oracle/_ref/libpgrc_ref.so, built by oracle/Makefile from /root/reference).  Skipped where the
reference build is absent; tests/test_oracle_golden.py then carries the pin through the fixtures."""
import numpy as np
import pytest

import oracle as orc
from pgrc_amd import synth

pytestmark = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def make_inputs(G, n, L, seed, n_with_n=0, tandem_every=2, paired=False):
    g = synth.pg_params(G, seed=seed, tandem_every=tandem_every)
    pg = synth.pg_host(g)
    rs = synth.reads_params(n, L, seed=seed, n_with_n=n_with_n, paired=paired)
    reads = synth.reads_host(g, pg, rs)
    return pg, reads


def assert_same(a, b, what):
    for k in ("pos", "rc", "mism", "hist"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs at {np.flatnonzero(a[k] != b[k])[:10]}"
    assert a["matched"] == b["matched"], what


@pytest.mark.parametrize("seed_len", [24, 28, 32, 38, 45, 54, 64, 100, 150, 250])
def test_copmem_params_and_index(seed_len):
    pg, _ = make_inputs(100000 if seed_len < 200 else 120000, 1, max(seed_len, 30), seed=7)
    pr, cr, posr = orc.ref_index(pg, seed_len)
    po, co, poso = orc.oracle_index(pg, seed_len)
    assert pr == po
    assert np.array_equal(cr, co)
    assert np.array_equal(posr, poso)


def test_copmem_hash_matches():
    rng = np.random.default_rng(1)
    for seed_len in (24, 28, 32, 38, 45, 50, 54, 64, 120):
        s = bytes(rng.choice(list(b"ACGTN"), size=64).astype(np.uint8))
        import ctypes as C
        K = C.c_int32()
        hr = orc.ref().pgrc_ref_copmem_hash(seed_len, s, C.byref(K))
        ho = orc.oracle().pgrc_or_copmem_hash(K.value, s)
        assert hr == ho


@pytest.mark.parametrize("L,seed_len,M,mode", [
    (100, 38, 50, "c"), (100, 38, 3, "c"), (100, 38, 50, "C"), (150, 38, 50, "c"), (250, 38, 50, "c"),
    (100, 100, 50, "c"), (64, 32, 10, "c"), (100, 24, 3, "c"),
])
def test_copmem_match(L, seed_len, M, mode):
    pg, reads = make_inputs(300000, 6000, L, seed=11 + L)
    kmax = L // M
    kmin = kmax if mode.isupper() else 0
    r = orc.ref_match("c", pg, reads, seed_len, kmax, kmin)
    o = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin)
    assert_same(r, o, f"copmem L={L} seed={seed_len} M={M} mode={mode}")


def test_copmem_match_with_n_reads_and_no_rc():
    pg, reads = make_inputs(200000, 4000, 100, seed=3, n_with_n=500)
    r = orc.ref_match("c", pg, reads, 38, 2, 0, n_nset=500)
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    assert_same(r, o, "copmem N reads")
    r = orc.ref_match("c", pg, reads, 38, 33, 0, rev_compl=False, n_nset=500)
    o = orc.oracle_match("c", pg, reads, 38, 33, 0, rev_compl=False)
    assert_same(r, o, "copmem fwd only")


@pytest.mark.parametrize("mode,L,seed_len,M", [
    ("e", 100, 100, 50), ("d", 100, 38, 50), ("i", 100, 38, 50), ("d", 100, 38, 3), ("i", 100, 38, 3),
    ("d", 150, 38, 50), ("i", 150, 38, 50), ("d", 100, 25, 10), ("i", 100, 25, 10), ("e", 150, 150, 3),
])
def test_seedindex_modes(mode, L, seed_len, M):
    pg, reads = make_inputs(200000, 4000, L, seed=5 + L + seed_len)
    # identical reads / identical parts exercise the equal-key (LIFO) order
    reads[100] = reads[50]
    reads[101] = reads[50]
    reads[200, : L // 2] = reads[200, L // 2: 2 * (L // 2)]
    kmax = L // M
    r = orc.ref_match(mode, pg, reads, seed_len, kmax, 0)
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same(r, o, f"mode {mode} L={L} seed={seed_len} M={M}")


def test_seedindex_shortcut_and_n_reads():
    pg, reads = make_inputs(150000, 3000, 100, seed=9, n_with_n=300)
    for mode in ("d", "i"):
        # one ACGNT-packed set holding every read.  (With the LQ+N SumOfConstantLengthReadsSets the
        # reference's modes d/i/e index ZERO reads: the sum never fills getReadsSetProperties()->readsCount,
        # ConstantLength...HashMatcher.cpp:30 -- see test below and DESIGN.md "reference quirks".)
        r = orc.ref_match(mode, pg, reads, 38, 2, 2, n_nset=3000)
        o = orc.oracle_match(mode, pg, reads, 38, 2, 2)
        assert_same(r, o, f"mode {mode} shortcut")


def test_reference_quirk_sum_set_indexes_nothing_in_modes_d_i():
    pg, reads = make_inputs(100000, 500, 100, seed=10, n_with_n=50)
    for mode in ("d", "i", "e"):
        r = orc.ref_match(mode, pg, reads, 38 if mode != "e" else 100, 2, 0, n_nset=50)
        assert r["matched"] == 0


def test_extract_mismatches():
    pg, reads = make_inputs(100000, 3000, 100, seed=21, n_with_n=200)
    o = orc.oracle_match("c", pg, reads, 38, 33, 0)
    checked = 0
    for i in np.flatnonzero((o["mism"] != 255) & (o["mism"] > 0))[:400]:
        for pair_file in (False, True):
            rc = int(o["rc"][i])
            reversed_ = (rc != (i & 1)) if pair_file else bool(rc)
            cr, offr = orc.ref_extract(pg, o["pos"][i], reads[i], rc, o["mism"][i], org_idx=i, rev_compl_pair_file=pair_file)
            co, offo = orc.oracle_extract(pg, o["pos"][i], reads[i], rc, reversed_, o["mism"][i])
            assert np.array_equal(cr, co) and np.array_equal(offr, offo), (i, pair_file)
            checked += 1
    assert checked > 100


def test_pack_layout_and_revcomp():
    import ctypes as C
    rng = np.random.default_rng(4)
    for L, alpha in ((100, b"ACGT"), (150, b"ACGT"), (101, b"ACGNT"), (37, b"ACGT"), (250, b"ACGNT")):
        read = rng.choice(list(alpha), size=L).astype(np.uint8)
        a = np.zeros(128, dtype=np.uint8)
        b = np.zeros(128, dtype=np.uint8)
        nb = orc.ref().pgrc_ref_pack_read(read.ctypes.data_as(C.c_void_p), L, alpha, a.ctypes.data_as(C.c_void_p))
        orc.oracle().pgrc_or_pack_read(read.ctypes.data_as(C.c_void_p), L, alpha, b.ctypes.data_as(C.c_void_p))
        assert np.array_equal(a[:nb], b[:nb])
        back = np.zeros(L, dtype=np.uint8)
        orc.oracle().pgrc_or_unpack_read(b.ctypes.data_as(C.c_void_p), L, alpha, back.ctypes.data_as(C.c_void_p))
        assert np.array_equal(back, read)
    s = rng.choice(list(b"ACGTN"), size=1001).astype(np.uint8)
    x, y = s.copy(), s.copy()
    orc.ref().pgrc_ref_revcomp(x.ctypes.data_as(C.c_void_p), x.size)
    orc.oracle().pgrc_or_revcomp(y.ctypes.data_as(C.c_void_p), y.size)
    assert np.array_equal(x, y)


def test_cyclic_hash_equivalence_candidates_are_canonical():
    """CyclicHash<uint32> rotates by one bit per symbol, so symbols 32 apart share a rotation: seeds longer than 32
    collide DETERMINISTICALLY when, per rotation class, every symbol has the same parity (cyclichash.h:100-123).
    Such candidates are verified and can be accepted in modes d/i -- the canonical seed key must reproduce them."""
    pg, reads0 = make_inputs(120000, 600, 100, seed=77)
    rng = np.random.default_rng(7)
    starts = rng.integers(0, pg.size - 100, size=600)
    for mode in ("d", "i"):
        reads = reads0.copy()
        for k in range(600):
            r = pg[starts[k]: starts[k] + 100].copy()
            q = k % 6  # seed positions q and q+32 share a rotation (38-symbol seeds)
            if mode == "d":   # parts = read[0:38], read[38:76]
                pairs = [(0 + q, 0 + q + 32), (38 + q, 38 + q + 32)]
            else:             # parts = read[j::2][:38], j = 0, 1
                pairs = [(0 + 2 * q, 0 + 2 * (q + 32)), (1 + 2 * q, 1 + 2 * (q + 32))]
            for a, b in pairs:
                r[a], r[b] = r[b], r[a]
            reads[k] = r
        r = orc.ref_match(mode, pg, reads, 38, 33, 0)
        o = orc.oracle_match(mode, pg, reads, 38, 33, 0)
        assert_same(r, o, f"cyclic equivalence mode {mode}")
        # 9/16 of these reads have both parts broken as strings: only the collision candidates reach them
        assert r["matched"] > 550
