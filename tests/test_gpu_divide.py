"""Row f3 on the GPU: making the packed read sets through the C ABI of include/pgrc_reads.h, against the oracle and the
compiled reference, and through integration/HipDividedReadsSets inside the reference's own classes."""
import numpy as np
import pytest

import oracle as orc
from divide_util import COMBOS, make_records, oracle_divide, ref_divide, same

pytestmark = pytest.mark.gpu

HAVE_REF = orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_divide")


def gpu_divide(reads, quals, error_limit, simplified, separate_n, n_reads_lq):
    from pgrc_amd import DividedPCLReadsSets
    d = DividedPCLReadsSets(reads.shape[1], error_limit, simplified, separate_n, n_reads_lq)
    res = d.divide(reads, quals)
    d.close()
    return res


@pytest.mark.parametrize("L", [37, 100, 150, 151, 255])
@pytest.mark.parametrize("combo", COMBOS)
def test_divide_parity(L, combo):
    reads, quals = make_records(seed=L * 17 + int(combo[0] * 1000), n=20000, L=L)
    g = gpu_divide(reads, quals, *combo)
    o = oracle_divide(reads, quals, *combo)
    assert same(g, o) is None, same(g, o)
    if HAVE_REF:
        r = ref_divide(reads[:4000], quals[:4000], *combo)
        assert same(gpu_divide(reads[:4000], quals[:4000], *combo), r) is None


def test_divide_quality_threshold_stress():
    """means of the correct-base probabilities packed around the limit: reads of ONE repeated quality character and reads
    that mix two neighbouring ones in every proportion -- the comparison `1 - q <= error_limit` must fall as in the
    reference for all of them (sums of floats are exact in doubles here; the division and the subtraction are IEEE)"""
    L = 150
    rows, rng = [], np.random.default_rng(3)
    for q in range(2, 42):
        rows.append(np.full(L, 33 + q, np.uint8))
        for k in range(1, L, 7):
            r = np.full(L, 33 + q, np.uint8)
            r[rng.permutation(L)[:k]] = 33 + min(41, q + 1)
            rows.append(r)
    quals = np.stack(rows)
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=quals.shape)].copy()
    for promils in (1, 2, 5, 10, 16, 20, 25, 32, 50, 63, 100, 126, 200, 251, 316, 369, 500, 602, 794, 999):
        combo = (promils / 1000.0, False, True, False)
        g = gpu_divide(reads, quals, *combo)
        o = oracle_divide(reads, quals, *combo)
        assert same(g, o) is None, (promils, same(g, o))
        if HAVE_REF:
            assert same(g, ref_divide(reads, quals, *combo)) is None, promils


def test_divide_edge_cases():
    from pgrc_amd import DividedPCLReadsSets, PgrcMatchError
    reads, quals = make_records(seed=2, n=100, L=100)
    d = DividedPCLReadsSets(100, 0.05, False, True, False)
    empty = d.divide(reads[:0], quals[:0])
    assert empty["n_hq"] == empty["n_lq"] == empty["n_n"] == 0 and empty["symbols"] == (4, 4, 5)
    one = d.divide(reads[:1], quals[:1])
    assert same(one, oracle_divide(reads[:1], quals[:1], 0.05, False, True, False)) is None
    with pytest.raises(PgrcMatchError):
        d.divide(reads, None)                               # quality rows are needed when error_limit < 1
    bad = reads.copy()
    bad[7, 3] = ord("X")
    with pytest.raises(PgrcMatchError) as e:
        d.divide(bad, quals)
    assert e.value.code == 5                                # PGRC_E_SYMBOL (the reference's validateSymbol exits)
    again = d.divide(reads, quals)                          # the context survives the error
    assert same(again, oracle_divide(reads, quals, 0.05, False, True, False)) is None
    d.close()
    all_n = np.full((64, 100), ord("N"), np.uint8)
    g = gpu_divide(all_n, quals[:64], 1.0, True, True, False)
    assert g["n_n"] == 64 and same(g, oracle_divide(all_n, quals[:64], 1.0, True, True, False)) is None
    with pytest.raises(PgrcMatchError):
        DividedPCLReadsSets(0)
    with pytest.raises(PgrcMatchError):
        DividedPCLReadsSets(100, 0.0, True)                 # simplified mode would test the position past the read


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
@pytest.mark.parametrize("batch", [1, 777, 100000])
def test_divide_reference_adapter_drop_in(monkeypatch, batch):
    """integration/HipDividedReadsSets inside the reference's class hierarchy: the DividedPCLReadsSets object it returns
    (three PackedConstantLengthReadsSets, two VectorMappings) equals the reference factory's, batch after batch"""
    monkeypatch.setenv("PGRC_DIVIDE_BATCH", str(batch))
    r = orc.ref()
    if not hasattr(r, "pgrc_ref_divide_batches"):
        pytest.skip("oracle/_ref was built without the adapter")
    r.pgrc_ref_divide_batches.restype = __import__("ctypes").c_uint64
    reads, quals = make_records(seed=23, n=3000 if batch > 1 else 300, L=150)
    for combo in COMBOS[1:7]:
        before = r.pgrc_ref_divide_batches()
        a = ref_divide(reads, quals, *combo, use_adapter=True)
        assert r.pgrc_ref_divide_batches() - before == -(-reads.shape[0] // batch)
        assert same(a, ref_divide(reads, quals, *combo)) is None, (combo, same(a, ref_divide(reads, quals, *combo)))


def test_divide_larger_batch_and_rate():
    """2 M records of 150 bp in one call (600 MB up): parity with the oracle, and the phases are reported"""
    from pgrc_amd import DividedPCLReadsSets
    reads, quals = make_records(seed=77, n=200000, L=150)
    reads, quals = np.tile(reads, (10, 1)), np.tile(quals, (10, 1))
    d = DividedPCLReadsSets(150, 0.05, False, True, False)
    g = d.divide(reads, quals)
    ms = d.last_ms()
    d.close()
    o = oracle_divide(reads, quals, 0.05, False, True, False)
    assert same(g, o) is None
    assert ms["upload"] > 0 and ms["kernels"] > 0
