"""Every BASELINE.json config at its FULL size, through size-independent properties plus a sample checked bit for bit
against the reference (serial canonical index, whole pseudogenome) -- or the oracle port where oracle/_ref is absent:
  C1  1 M x 100 bp, exact matcher                     test_c1_full_size_exact_matcher
  C2  10 M x 100 bp, k <= 2, Pg 125 Mbp               test_c2_full_size
  C3  100 M x 150 bp SE, k <= 3, Pg 1.875 Gbp         test_c3_full_size_properties_and_sample
  C4  100 M x 150 bp PE, reads in 8 shards, the packed Pg assembled from 8 slices    test_c4_full_size_eight_shards
  C5  one GPU's 1/8 of 500 M x 250 bp, k <= 5, Pg 3.1 Gbp (hash 2^30)               test_c5_shard_full_size
  P64 Pg of 4.4 Gbp: the 64-bit-position kernels at a real >= 4 Gi text             test_p64_full_size
and row f2 (Pg-vs-Pg) at the C3 Pg size."""
import numpy as np
import pytest
import torch

import oracle as orc
from pgrc_amd import MatchContext, synth
from util import revcomp

pytestmark = pytest.mark.gpu

_LUT = np.zeros((256, 4), dtype=np.uint8)
for _b in range(256):
    for _k in range(4):
        _LUT[_b, _k] = b"ACGT"[(_b >> (2 * _k)) & 3]
_COMP = np.zeros(256, dtype=np.uint8)
for _x, _y in zip(b"ACGTN", b"TGCAN"):
    _COMP[_x] = _y


def _unpack(words, G):
    return _LUT[words.view(np.uint8)].reshape(-1)[:G]


def _check(mode, pg, reads, seed_len, kmax, threads=16):
    """the checker: the compiled reference with its serial index and one thread in the per-read loop (its RC-flag
    race), else the oracle port"""
    if orc.have_ref():
        return orc.ref_match(mode, pg, reads, seed_len, kmax, 0, True, 0, 1, 1)
    return orc.oracle_match(mode, pg, reads, seed_len, kmax, 0, True, threads)


def _sample_over_all_reads(ctx, n, n_stride, n_redo, first=0):
    """read indexes for a parity sample that is not just the head of the set: the first `first` reads, a stride over ALL
    reads, and a stride over the reads the dual kernel of the last run did again in the reference's order (the ones in
    repeat families, where the falses budget matters: pgrc_match_get_redo_flags)"""
    redo = np.flatnonzero(ctx.redo_flags())
    parts = [np.arange(min(first, n), dtype=np.int64), np.linspace(0, n - 1, n_stride).astype(np.int64)]
    if redo.size:
        parts.append(redo[np.linspace(0, redo.size - 1, min(n_redo, redo.size)).astype(np.int64)].astype(np.int64))
    idx = np.unique(np.concatenate(parts))
    return idx, int(np.isin(idx, redo).sum())


def _rows_from_hbm(d_rd, idx, L, stride):
    """ASCII rows of reads `idx` from the word-major 2-bit read set in HBM: exactly what the GPU matched"""
    nw = (L + 15) // 16
    ix = torch.as_tensor(idx, device=d_rd.device)
    rows = torch.stack([d_rd[w * stride + ix] for w in range(nw)], dim=1).cpu().numpy().view(np.uint32)
    sym = (rows[:, :, None] >> (2 * np.arange(16, dtype=np.uint32))[None, None, :]) & 3
    return np.frombuffer(b"ACGT", dtype=np.uint8)[sym.reshape(idx.size, nw * 16)[:, :L]]


def _bookkeeping(n, G, L, kmax, pos, rc, mism, hist, matched):
    assert int(hist.sum()) == n and matched == n - int(hist[255])
    assert np.array_equal(np.bincount(mism, minlength=256).astype(np.uint64), hist)
    assert np.array_equal(pos == np.uint64(2**64 - 1), mism == 255)
    assert int(mism[mism != 255].max()) <= kmax and not rc[mism == 255].any()
    assert int(pos[mism != 255].max()) <= G - L


def _alignments_are_real(pg, reads, pos, rc, mism):
    """recompute the Hamming distance at every reported place of a sample"""
    L = reads.shape[1]
    idx = np.flatnonzero(mism != 255)
    win = pg[pos[idx, None].astype(np.int64) + np.arange(L)[None, :]]
    rd = reads[idx]
    ham = np.where(rc[idx, None] != 0, _COMP[rd[:, ::-1]] != win, rd != win).sum(axis=1)
    assert np.array_equal(ham.astype(np.uint8), mism[idx])


class _C3World:
    """The C3 / C4 pseudogenome (1.875 Gbp, seed 12345) in HBM and on the host, and ONE reference run over it for the
    samples of both tests (the reference's index build over the whole text is what costs: ~15 s)."""
    n, L, G, seed_len, kmax, ns = 100_000_000, 150, 1_875_000_000, 38, 3, 100_000

    def __init__(self):
        self.g = synth.pg_params(self.G, seed=12345)
        self.pgw = (self.G + 15) // 16
        self.d_pg = torch.zeros(self.pgw + 64, dtype=torch.int32, device="cuda")
        synth.pg_device(self.g, self.d_pg.data_ptr())
        torch.cuda.synchronize()
        self.pg = _unpack(self.d_pg.cpu().numpy().view(np.uint32)[: self.pgw], self.G)
        self.rs_se = synth.reads_params(self.n, self.L, seed=12345)
        self.rs_pe = synth.reads_params(self.n, self.L, seed=12345, paired=True)
        self.reads_se = synth.reads_host(self.g, self.pg, self.rs_se, 0, 3 * self.ns)
        self.reads_pe = synth.reads_host(self.g, self.pg, self.rs_pe, 0, self.ns)
        both = np.concatenate([self.reads_se[: self.ns], self.reads_pe])
        r = _check("c", self.pg, both, self.seed_len, self.kmax)
        self.ref_se = {k: r[k][: self.ns] for k in ("pos", "rc", "mism")}
        self.ref_pe = {k: r[k][self.ns:] for k in ("pos", "rc", "mism")}

    def device_reads(self, rs):
        nw, stride = (self.L + 15) // 16, (self.n + 63) & ~63
        d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda")
        synth.reads_device(self.g, self.d_pg.data_ptr(), rs, 0, self.n, d_rd.data_ptr(), stride)
        torch.cuda.synchronize()
        return d_rd, stride


@pytest.fixture(scope="module")
def c3world():
    w = _C3World()
    yield w
    del w.d_pg
    torch.cuda.empty_cache()


def test_c3_full_size_properties_and_sample(c3world):
    w = c3world
    n, L, G, seed_len, kmax = w.n, w.L, w.G, w.seed_len, w.kmax
    d_pg, pg, g, rs = w.d_pg, w.pg, w.g, w.rs_se
    d_rd, stride = w.device_reads(rs)
    ctx = MatchContext(L, seed_len, kmax, 0, "c")
    ctx.set_pg_packed_device(d_pg.data_ptr(), G)
    ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ctx.init_results()
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    idx, n_redo = _sample_over_all_reads(ctx, n, 30_000, 30_000)          # (the redo flags are those of THIS run)
    assert ctx.counters()["screened"] == 2 and n_redo >= 10_000
    # 1. bookkeeping: histogram = histogram of the per-read counts; matched = reads with a position
    _bookkeeping(n, G, L, kmax, pos, rc, mism, hist, matched)
    # 2. idempotence: a second run over the finished state changes nothing (every pass must strictly improve)
    ctx.run(True)
    pos2, rc2, mism2, hist2, _ = ctx.get_results()
    assert np.array_equal(pos, pos2) and np.array_equal(rc, rc2) and np.array_equal(mism, mism2) and np.array_equal(hist, hist2)
    # 3. planted truth + reported alignments are real: on a sample, recompute the Hamming distance at the reported place
    ns = 300_000
    reads = w.reads_se
    _alignments_are_real(pg, reads, pos[:ns], rc[:ns], mism[:ns])
    comp = _COMP
    # 4. bit-identity on a sample against the reference (serial canonical index) or the oracle, whole Pg
    m = w.ns
    r = w.ref_se
    assert np.array_equal(pos[:m], r["pos"]) and np.array_equal(rc[:m], r["rc"]) and np.array_equal(mism[:m], r["mism"])
    # ... and on a stride over ALL reads plus the reads the dual kernel redid in the reference's order (rows read back from HBM)
    rows = _rows_from_hbm(d_rd, idx, L, stride)
    assert np.array_equal(rows[:50], reads[idx[:50]]) if idx[49] < reads.shape[0] else True
    r2 = _check("c", pg, rows, seed_len, kmax)
    assert np.array_equal(pos[idx], r2["pos"]) and np.array_equal(rc[idx], r2["rc"]) and np.array_equal(mism[idx], r2["mism"])
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


def test_c3_full_size_seed_modes_bit_parity(c3world):
    """Rows a5-a7 bit for bit at the FULL C3 text (until round 5 only the builder-run script
    tests/fullscale_parity.py did this): a sample of the workload's reads matched against the whole 1.875 Gbp Pg in modes d, i
    and e -- positions, strands, counts AND the number of (window, part) pairs with equal keys per strand -- against the
    oracle's serial scans of the whole text (DefaultReadsApproxMatcher / InterleavedReadsApproxMatcher /
    DefaultReadsExactMatcher::executeMatching, matching/ReadsMatchers.cpp:198-230, :297-409).  The three scans (about 80 s
    each on one core) run side by side in threads: the checker's C code releases the interpreter lock."""
    from concurrent.futures import ThreadPoolExecutor
    w = c3world
    L, G, kmax = w.L, w.G, w.kmax
    ns = 60_000
    reads = w.reads_se[:ns]
    nw, stride = (L + 15) // 16, (ns + 63) & ~63
    d_rd = torch.empty(nw * stride, dtype=torch.int32, device="cuda")
    synth.reads_device(w.g, w.d_pg.data_ptr(), w.rs_se, 0, ns, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    legs = [("d", w.seed_len, kmax), ("i", w.seed_len, kmax), ("e", L, 0)]
    with ThreadPoolExecutor(max_workers=3) as pool:
        futs = [pool.submit(orc.oracle_match, mode, w.pg, reads, sl, km, 0, True, 1) for mode, sl, km in legs]
        got = []
        for mode, sl, km in legs:                 # (the GPU legs while the scans run)
            cx = MatchContext(L, sl, km, 0, mode)
            cx.set_pg_packed_device(w.d_pg.data_ptr(), G)
            cx.set_reads_device(d_rd.data_ptr(), ns, stride, keep=d_rd)
            cx.init_results()
            cx.run(True)
            got.append(cx.get_results()[:3] + (cx.counters()["candidates"],))
            del cx
        want = [f.result() for f in futs]
    for (mode, sl, km), (pos, rc, mism, cand), o in zip(legs, got, want):
        assert np.array_equal(pos, o["pos"]) and np.array_equal(rc, o["rc"]) and np.array_equal(mism, o["mism"]), mode
        assert [int(v) for v in cand] == [int(v) for v in o["candidates"]], (mode, cand, o["candidates"])
        assert int((mism != 255).sum()) >= (0.55 if mode == "e" else 0.85) * ns, mode


def test_c4_full_size_eight_shards(c3world):
    """configs[3]: 100 M x 150 bp PE, reads sharded 8 ways, the packed Pg assembled from 8 slices.  The 8 ranks' work
    runs back to back on this device: every rank's slice of the host text is packed by pgrc_match_pack_pg_slice into
    its place of the gathered buffer (what the all-gather assembles), that buffer becomes the text
    (pgrc_match_set_pg_packed_device), and every shard's reads -- even-aligned ranges, PE mates together -- are matched
    against it.  The concatenation must equal the one-shot run bit for bit, and a sample the reference."""
    from pgrc_amd import dist as pdist
    w = c3world
    n, L, G, seed_len, kmax = w.n, w.L, w.G, w.seed_len, w.kmax
    world = 8
    d_rd, stride = w.device_reads(w.rs_pe)
    one = MatchContext(L, seed_len, kmax, 0, "c")
    one.set_pg_packed_device(w.d_pg.data_ptr(), G)
    one.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    one.init_results()
    one.run(True)
    pos, rc, mism, hist, matched = one.get_results()
    _bookkeeping(n, G, L, kmax, pos, rc, mism, hist, matched)
    assert hist[0] >= 0.55 * n and matched >= 0.85 * n
    m = w.ns
    _alignments_are_real(w.pg, w.reads_pe, pos[:m], rc[:m], mism[:m])
    for k in ("pos", "rc", "mism"):
        assert np.array_equal({"pos": pos, "rc": rc, "mism": mism}[k][:m], w.ref_pe[k]), k
    del one
    # the gathered text, slice by slice
    sw = pdist.pg_slice(G, 0, world)[2]
    d_full = torch.zeros(sw * world + 64, dtype=torch.int32, device="cuda")
    ctx = MatchContext(L, seed_len, kmax, 0, "c")
    for r in range(world):
        lo, hi, _ = pdist.pg_slice(G, r, world)
        ctx.pack_pg_slice(w.pg[lo:hi], d_full.data_ptr() + 4 * r * sw)
    torch.cuda.synchronize()
    assert torch.equal(d_full[: w.pgw], w.d_pg[: w.pgw])
    ctx.set_pg_packed_device(d_full.data_ptr(), G)
    total_hist = np.zeros(256, dtype=np.uint64)
    end = 0
    for r in range(world):
        lo, hi = pdist.shard_range(n, r, world)
        assert lo == end and lo % 2 == 0
        end = hi
        # word w of read i lives at words[w * stride + i]: a shard is the same array seen from read `lo` on
        ctx.set_reads_device(d_rd.data_ptr() + 4 * lo, hi - lo, stride, keep=d_rd)
        ctx.init_results()
        ctx.run(True)
        p, c, mm, h, _ = ctx.get_results()
        assert np.array_equal(p, pos[lo:hi]) and np.array_equal(c, rc[lo:hi]) and np.array_equal(mm, mism[lo:hi]), r
        total_hist += h
    assert end == n and np.array_equal(total_hist, hist)


def _full_size_run(n, L, G, seed_len, kmax, sample, also_mode=None):
    """inputs generated in HBM, one run over all n reads, properties, and the first `sample` reads against the checker
    on the whole text"""
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
    _bookkeeping(n, G, L, kmax, pos, rc, mism, hist, matched)
    assert hist[0] >= 0.55 * n and matched >= 0.85 * n
    pg = _unpack(d_pg.cpu().numpy().view(np.uint32)[:pgw], G)
    reads = synth.reads_host(g, pg, rs, 0, sample)
    _alignments_are_real(pg, reads, pos[:sample], rc[:sample], mism[:sample])
    # the parity sample: the first reads, a stride over all of them, and a stride over the reads the dual kernel redid in
    # the reference's order -- one checker run over the whole text for the lot (rows read back from HBM)
    idx, n_redo = _sample_over_all_reads(ctx, n, sample // 4, sample // 4, first=sample // 2)
    assert n_redo >= min(1000, int(ctx.counters()["redo_reads"]))
    rows = _rows_from_hbm(d_rd, idx, L, stride)
    assert np.array_equal(rows[: sample // 2], reads[: sample // 2])
    r = _check("c", pg, rows, seed_len, kmax)
    for k, v in (("pos", pos), ("rc", rc), ("mism", mism)):
        assert np.array_equal(v[idx], r[k]), k
    if also_mode:
        # a read-side seed-index mode on the same inputs: reported alignments are real, every read mode c matched exactly
        # is matched exactly (every part of such a read hits at its occurrence)
        cx = MatchContext(L, seed_len, kmax, 0, also_mode)
        cx.set_pg_packed_device(d_pg.data_ptr(), G)
        cx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
        cx.init_results()
        cx.run(True)
        p2, r2, m2, h2, mt2 = cx.get_results()
        _bookkeeping(n, G, L, kmax, p2, r2, m2, h2, mt2)
        assert (m2 == 0)[mism == 0].all() and mt2 >= 0.85 * n
        _alignments_are_real(pg, reads, p2[:sample], r2[:sample], m2[:sample])
        assert int((p2[m2 != 255] >= np.uint64(2**32)).sum()) > 100_000
    return ctx, pos, mism


def test_c2_full_size():
    """configs[1]: 10 M x 100 bp SE, k <= 2 (-M 50), Pg 125 Mbp"""
    _full_size_run(10_000_000, 100, 125_000_000, 38, 2, 200_000)


def test_c5_shard_full_size():
    """configs[4]: one GPU's share (1/8) of 500 M x 250 bp, k <= 5, against the 3.1 Gbp Pg (hash table 2^30, positions
    still 32-bit)."""
    ctx, pos, mism = _full_size_run(62_500_000, 250, 3_100_000_000, 38, 5, 50_000)
    assert ctx.counters()["index_entries"][0] > 600_000_000


def test_p64_full_size():
    """A text of 4.4 Gbp (>= 4 Gi symbols): the reference's u64 index branch (CopMEMMatcher.cpp:579-586), here the
    64-bit-position kernels, at a size where positions really leave 32 bits; mode d on the same inputs scans the text in
    two segments."""
    ctx, pos, mism = _full_size_run(50_000_000, 150, 4_400_000_000, 38, 3, 50_000, also_mode="d")
    assert int((pos[mism != 255] >= np.uint64(2**32)).sum()) > 100_000


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


def test_pg_vs_pg_matching_full_size_properties(c3world):
    """Row f2 at BASELINE's pseudogenome size (1.875 Gbp against itself, forward and reverse-complement): every
    reported match is a real exact match, long enough, not extensible to the right, extensible to the left by at most
    the reference's one symbol at a text start; the self-match filter holds.
    (Identity with the reference at this size: tests/mem_scale.py, profiles/r01_mem_scale_C3.json.)"""
    from pgrc_amd import CopMEMMatcher
    G = c3world.G
    src = c3world.pg
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
