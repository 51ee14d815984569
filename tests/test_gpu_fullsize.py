"""BASELINE.json's full sizes (C3: 100 M x 150 bp against a 1.875 Gbp Pg; C1: exact matcher) through
size-independent properties, plus a sample checked against the reference / oracle on the whole pseudogenome."""
import numpy as np
import pytest
import torch

import oracle as orc
from pgrc_amd import MatchContext, synth
from util import revcomp

pytestmark = pytest.mark.gpu


def _unpack(words, G):
    lut = np.zeros((256, 4), dtype=np.uint8)
    for b in range(256):
        for k in range(4):
            lut[b, k] = b"ACGT"[(b >> (2 * k)) & 3]
    return lut[words.view(np.uint8)].reshape(-1)[:G]


def test_c3_full_size_properties_and_sample():
    n, L, G, seed_len, kmax = 100_000_000, 150, 1_875_000_000, 38, 3
    g = synth.pg_params(G, seed=12345)
    rs = synth.reads_params(n, L, seed=12345)
    nw, stride, pgw = (L + 15) // 16, (n + 63) & ~63, (G + 15) // 16
    d_pg = torch.zeros(pgw + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda")
    synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    ctx = MatchContext(L, seed_len, kmax, 0, "c")
    ctx.set_pg_packed_device(d_pg.data_ptr(), G)
    ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ctx.init_results()
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    # 1. bookkeeping: histogram = histogram of the per-read counts; matched = reads with a position
    assert int(hist.sum()) == n and matched == n - int(hist[255])
    assert np.array_equal(np.bincount(mism, minlength=256).astype(np.uint64), hist)
    assert np.array_equal(pos == np.uint64(2**64 - 1), mism == 255)
    assert int(mism[mism != 255].max()) <= kmax and not rc[mism == 255].any()
    assert int(pos[mism != 255].max()) <= G - L
    # 2. idempotence: a second run over the finished state changes nothing (every pass must strictly improve)
    ctx.run(True)
    pos2, rc2, mism2, hist2, _ = ctx.get_results()
    assert np.array_equal(pos, pos2) and np.array_equal(rc, rc2) and np.array_equal(mism, mism2) and np.array_equal(hist, hist2)
    # 3. planted truth + reported alignments are real: on a sample, recompute the Hamming distance at the reported place
    pg = _unpack(ctx.export_pg(0), G)
    ns = 300_000
    reads = synth.reads_host(g, pg, rs, 0, ns)
    idx = np.flatnonzero(mism[:ns] != 255)
    win = pg[pos[idx, None].astype(np.int64) + np.arange(L)[None, :]]
    rd = reads[idx]
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    rd_rc = comp[rd[:, ::-1]]
    ham = np.where(rc[idx, None] != 0, rd_rc != win, rd != win).sum(axis=1)
    assert np.array_equal(ham.astype(np.uint8), mism[idx])
    # 4. bit-identity on a sample against the reference (serial canonical index) or the oracle, whole Pg
    m = 100_000
    if orc.have_ref():
        r = orc.ref_match("c", pg, reads[:m], seed_len, kmax, 0, True, 0, 1, 1)   # one thread: the reference's RC-flag race
    else:
        r = orc.oracle_match("c", pg, reads[:m], seed_len, kmax, 0, True, 16)
    assert np.array_equal(pos[:m], r["pos"]) and np.array_equal(rc[:m], r["rc"]) and np.array_equal(mism[:m], r["mism"])
    # the generator plants 60 % exact reads (3 % of all reads are random): at least that many must match exactly
    assert hist[0] >= 0.55 * n and matched >= 0.85 * n
    # 5. the read-side seed-index modes on the same inputs: reported alignments are real, the exact matches agree with mode c
    exact_c = mism == 0
    for mode in ("d", "i"):
        cx = MatchContext(L, seed_len, kmax, 0, mode)
        cx.set_pg_packed_device(d_pg.data_ptr(), G)
        cx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
        cx.init_results()
        cx.run(True)
        p2, r2, m2, h2, mt2 = cx.get_results()
        assert int(h2.sum()) == n and np.array_equal(np.bincount(m2, minlength=256).astype(np.uint64), h2)
        assert mt2 >= 0.85 * n and int(m2[m2 != 255].max()) <= kmax
        # every part of a read with an exact occurrence hits there, so modes d / i find all of them; mode c can miss some
        # (its index is sampled and its buckets are capped)
        assert (m2 == 0)[exact_c].all() and int((m2 == 0).sum()) >= int(exact_c.sum())
        idx2 = np.flatnonzero(m2[:ns] != 255)
        win2 = pg[p2[idx2, None].astype(np.int64) + np.arange(L)[None, :]]
        ham2 = np.where(r2[idx2, None] != 0, comp[reads[idx2][:, ::-1]] != win2, reads[idx2] != win2).sum(axis=1)
        assert np.array_equal(ham2.astype(np.uint8), m2[idx2]), mode
        del cx


def test_c1_full_size_exact_matcher():
    """configs[0]: 1 M x 100 bp, exact match against a 12.5 Mbp Pg (DefaultReadsExactMatcher)."""
    n, L, G = 1_000_000, 100, 12_500_000
    g = synth.pg_params(G, seed=12345)
    pg = synth.pg_host(g)
    rs = synth.reads_params(n, L, seed=12345)
    reads = synth.reads_host(g, pg, rs)
    ctx = MatchContext(L, L, 0, 0, "e")
    ctx.set_pg_ascii(pg)
    ctx.set_reads_ascii(reads)
    ctx.init_results()
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    o = orc.oracle_match("e", pg, reads, L, 0, 0, True)
    assert np.array_equal(pos, o["pos"]) and np.array_equal(rc, o["rc"]) and matched == o["matched"]
    # every reported position is an exact occurrence
    idx = np.flatnonzero(mism == 0)[:200_000]
    win = pg[pos[idx, None].astype(np.int64) + np.arange(L)[None, :]]
    rd = np.where(rc[idx, None] != 0, revcomp(reads[idx].reshape(-1)).reshape(idx.size, L)[::-1], reads[idx])
    assert np.array_equal(win, rd)


def test_pg_vs_pg_matching_full_size_properties():
    """Row f2 at BASELINE's pseudogenome size (1.875 Gbp against itself, forward and reverse-complement): every
    reported match is a real exact match, long enough, not extensible to the right, extensible to the left by at most
    the reference's one symbol at a text start; the self-match filter holds.
    (Identity with the reference at this size: tests/mem_scale.py, profiles/r01_mem_scale_C3.json.)"""
    from pgrc_amd import CopMEMMatcher
    G = 1_875_000_000
    g = synth.pg_params(G, seed=12345)
    d_pg = torch.zeros((G + 15) // 16 + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    torch.cuda.synchronize()
    src = _unpack(d_pg.cpu().numpy().view(np.uint32)[: (G + 15) // 16], G).copy()
    del d_pg
    m = CopMEMMatcher(src, 45)
    rng = np.random.default_rng(1)
    for rev_compl in (False, True):
        dest = orc.revcomp_ascii(src) if rev_compl else src
        mt = m.matchTexts(dest, True, rev_compl)
        ctr = m.counters()
        assert ctr["probes"] == (G - 32) // 3 + 1
        assert len(mt) > (100_000 if not rev_compl else 10_000)
        ps, ln, pd = mt[:, 0].astype(np.int64), mt[:, 1].astype(np.int64), mt[:, 2].astype(np.int64)
        assert ln.min() >= 45 and (ps + ln).max() <= G and (pd + ln).max() <= G
        # self matches are dropped (CopMEMMatcher.cpp:389-392): the anchor lies before the source position (forward)
        if not rev_compl:
            assert (pd < ps).all()
        # (the same match CAN be reported twice: a window near its end is no longer "inside" it, :393-399 -- the
        #  reference removes such duplicates afterwards with sort + unique, SimplePgMatcher.cpp:95-96)
        for k in rng.choice(len(mt), size=3000, replace=False):
            a, b, n = ps[k], pd[k], ln[k]
            assert np.array_equal(src[a:a + n], dest[b:b + n])
            assert a + n == G or b + n == G or src[a + n] != dest[b + n]                  # right-maximal
            if a > 0 and b > 0 and src[a - 1] == dest[b - 1]:                            # the off-by-one at a text start
                assert a == 1 or b == 1
