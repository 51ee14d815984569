"""Pg-vs-Pg exact matching (row f2) on the GPU through the C ABI of include/pgrc_mem.h, against the oracle and the
compiled reference: the same matches in the same discovery order."""
import numpy as np
import pytest

import oracle as orc
from mem_util import COMBOS, make_pair, make_pg

pytestmark = pytest.mark.gpu

HAVE_REF = orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_mem_match")


def check(src, other, combos=COMBOS, target_len=45, min_len=None, what=""):
    from pgrc_amd import CopMEMMatcher
    m = CopMEMMatcher(src, target_len)
    total = 0
    for dest_is_src, rev_compl in combos:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        g = m.matchTexts(d, dest_is_src, rev_compl, min_len)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl, target_len, min_len)
        assert np.array_equal(g, o), f"{what} vs oracle: destIsSrc={dest_is_src} rc={rev_compl}: {len(g)} / {len(o)} matches"
        if HAVE_REF:
            r = orc.ref_mem_match(src, d, dest_is_src, rev_compl, target_len, min_len)
            assert np.array_equal(g, r), f"{what} vs reference: destIsSrc={dest_is_src} rc={rev_compl}"
        total += len(g)
    m.close()
    return total


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("with_n,low_complexity", [(False, False), (True, False), (False, True), (True, True)])
def test_mem_parity(seed, with_n, low_complexity):
    src, other = make_pair(seed, with_n=with_n, low_complexity=low_complexity)
    assert check(src, other, what=f"seed {seed}") > 100


@pytest.mark.parametrize("target_len,min_len", [(45, 45), (45, 60), (50, 50), (36, 36), (64, 64), (120, 120), (24, 24), (255, 255)])
def test_mem_other_lengths(target_len, min_len):
    src, other = make_pair(7, G=150000, G2=50000)
    check(src, other, COMBOS[:2], target_len, min_len, what=f"L={target_len}")


def test_mem_short_and_ragged_texts():
    src, other = make_pair(3, G=30000, G2=2000)
    from pgrc_amd import CopMEMMatcher
    m = CopMEMMatcher(src, 45)
    for n2 in (1, 31, 32, 33, 45, 100, 400, 767, 768, 769, 770, 800, 1535, 1536, 1537, 2000):
        d = orc.mem_dest(src, other[:n2], 0, 1)
        g = m.matchTexts(d, False, True)
        assert np.array_equal(g, orc.oracle_mem_match(src, d, 0, 1)), n2


def test_mem_matches_touching_the_text_ends_and_stale_registers():
    """Source K-mers at the very start / end of the source (their side contexts lie outside the text, so the
    reference tests them against whatever an earlier entry left in its registers) copied all over the destination."""
    rng = np.random.default_rng(5)
    src = make_pg(120000, 55, nrep=50)
    other = make_pg(40000, 56, nrep=5)
    head, tail = src[:70].copy(), src[-70:].copy()
    for k in range(40):
        seg = (head, tail)[k % 2][: int(rng.integers(46, 70))] if k % 4 < 2 else (head, tail)[k % 2][-int(rng.integers(46, 70)):]
        d = int(rng.integers(100, other.size - 100))
        other[d:d + seg.size] = orc.revcomp_ascii(seg) if k % 3 else seg
    other[:60] = orc.revcomp_ascii(src[-60:])
    other[-60:] = orc.revcomp_ascii(src[:60])
    from pgrc_amd import CopMEMMatcher
    m = CopMEMMatcher(src, 45)
    stale = 0
    for dest_is_src, rev_compl in COMBOS:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        g = m.matchTexts(d, dest_is_src, rev_compl)
        stale += m.counters()["stale_lookups"]
        assert np.array_equal(g, orc.oracle_mem_match(src, d, dest_is_src, rev_compl)), (dest_is_src, rev_compl)
        if HAVE_REF:
            assert np.array_equal(g, orc.ref_mem_match(src, d, dest_is_src, rev_compl)), (dest_is_src, rev_compl)
    assert stale > 0        # the host-side register emulation was exercised (between rounds of the device replay)


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
def test_mem_reference_adapter_drop_in():
    """integration/HipTextMatcher inside the reference's TextMatcher hierarchy against the reference's CopMEMMatcher."""
    if not hasattr(orc.ref(), "pgrc_ref_mem_match_via_adapter"):
        pytest.skip("oracle/_ref was built without the adapter")
    src, other = make_pair(11, with_n=True)
    for dest_is_src, rev_compl in COMBOS:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        a = orc.ref_mem_match_via_adapter(src, d, dest_is_src, rev_compl)
        r = orc.ref_mem_match(src, d, dest_is_src, rev_compl)
        assert len(r) > 20 and np.array_equal(a, r), (dest_is_src, rev_compl)


def test_mem_long_matches():
    """Matches of 50-150 kbp (thousands of events each, all with the same extents): the run-based extension must give
    what the reference's per-event extension gives."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    src = rng.choice(acgt, size=500000)
    src[300000:360000] = src[20000:80000]                      # a 60 kbp forward duplicate inside the source
    src[400000:450000] = orc.revcomp_ascii(src[100000:150000])  # and a 50 kbp reverse-complement one
    other = rng.choice(acgt, size=260000)
    other[5000:155000] = src[200000:350000]                    # a 150 kbp copy (spans the internal duplicate)
    other[160000:250000] = orc.revcomp_ascii(src[10000:100000])
    other[100000] = ord("N") if other[100000] != ord("N") else ord("A")   # an N splits the long copy
    from pgrc_amd import CopMEMMatcher
    m = CopMEMMatcher(src, 45)
    for dest_is_src, rev_compl in COMBOS:
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        g = m.matchTexts(d, dest_is_src, rev_compl)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl)
        assert np.array_equal(g, o), (dest_is_src, rev_compl, len(g), len(o))
        assert int(g[:, 1].max()) >= 49000
        if HAVE_REF:
            assert np.array_equal(g, orc.ref_mem_match(src, d, dest_is_src, rev_compl))


def test_mem_replay_rounds_on_the_device():
    """The sequential rules run on the device, one thread per block of 256 windows, in rounds until every block has seen the
    last match recorded before it (mem.hip, step 4).  One long copy makes every block after the first record the match
    again in round 0 and take rule (b) in round 1; matches that straddle a block boundary cost their successor a replay."""
    from pgrc_amd import CopMEMMatcher
    rng = np.random.default_rng(91)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    src = rng.choice(acgt, size=400000)
    other = rng.choice(acgt, size=300000)
    other[1000:201000] = src[100000:300000]                    # a 200 kbp copy: 260 blocks of 768 symbols
    for k in range(300):                                        # and 300 short ones, many across a block boundary
        d = 201500 + k * 320
        other[d:d + 300] = src[k * 300: k * 300 + 300]
    m = CopMEMMatcher(src, 45)
    g = m.matchTexts(other, False, False)
    c = m.counters()
    assert np.array_equal(g, orc.oracle_mem_match(src, other, 0, 0))
    if HAVE_REF:
        assert np.array_equal(g, orc.ref_mem_match(src, other, 0, 0))
    assert c["event_blocks"] > 300 and c["replay_rounds"] >= 2, c
    assert len(g) >= 300 and int(g[:, 1].max()) >= 199000
    m.close()
    for seed, with_n, lowc in ((0, False, False), (1, True, True), (2, False, True)):
        src, other = make_pair(seed, with_n=with_n, low_complexity=lowc)
        assert check(src, other, what=f"seed {seed}") > 100


def test_mem_event_buffer_regrows(monkeypatch):
    """more events than the first guess of the event buffer: the probe pass is rerun with the exact size"""
    monkeypatch.setenv("PGRC_MEM_EVENT_CAP", "7")
    src, other = make_pair(2, low_complexity=True)
    assert check(src, other, COMBOS[:2], what="tiny event cap") > 50


def test_mem_errors():
    from pgrc_amd import CopMEMMatcher, PgrcMatchError
    src, other = make_pair(1, G=20000, G2=3000)
    with pytest.raises(PgrcMatchError):
        CopMEMMatcher(src, 20)                       # "Minimal matching length too short" (CopMEMMatcher.cpp:77-80)
    m = CopMEMMatcher(src, 45)
    with pytest.raises(PgrcMatchError):
        m.matchTexts(other, False, False, 20)        # minMatchLength < K (:606-609)
    with pytest.raises(PgrcMatchError):
        m.matchTexts(other, True, True)              # dest_is_src with a text of another length
    with pytest.raises(PgrcMatchError) as e:
        CopMEMMatcher(np.frombuffer(b"ACGTNACGT" * 20, dtype=np.uint8), 45)   # the source must be over ACGT
    assert e.value.code == 5
    assert len(m.matchTexts(other[:10], False, True)) == 0                    # shorter than K: no window, no match
    bad = other.copy()
    bad[100] = ord("%")
    with pytest.raises(PgrcMatchError) as e:
        m.matchTexts(bad, False, False)
    assert e.value.code == 5
