"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle -- and against the
real reference where oracle/_ref is present -- on the same seeded inputs.  Bit-exact: positions,
RC flags, mismatch counts, histogram, matched count, index arrays, mismatch lists."""
import os

import numpy as np
import pytest

import oracle as orc
from util import assert_same_results, gpu_match, make_inputs, pack2, revcomp

pytestmark = pytest.mark.gpu


def test_extension_is_the_loaded_native_code():
    import pgrc_amd
    assert pgrc_amd.LIB_PATH.endswith("pgrc_amd/libpgrc_match.so")
    with open("/proc/self/maps") as f:
        assert "libpgrc_match.so" in f.read()


@pytest.mark.parametrize("G", [100000, 100003, 65536 + 7, 1000])
def test_pack_and_revcomp(G):
    from pgrc_amd import MatchContext
    pg, _ = make_inputs(max(G, 1000), 1, 100, seed=G)
    pg = pg[:G]
    ctx = MatchContext(100, 38, 2, 0, "c")
    ctx.set_pg_ascii(pg)
    assert np.array_equal(ctx.export_pg(0), pack2(pg))
    assert np.array_equal(ctx.export_pg(1), pack2(revcomp(pg)))


@pytest.mark.parametrize("seed_len", [24, 28, 38, 45, 64, 100, 150, 250])
def test_index_matches_oracle(seed_len):
    from pgrc_amd import MatchContext, copmem_params
    L = max(seed_len, 100)
    pg, _ = make_inputs(300000, 1, L, seed=seed_len)
    prm, cumm, positions = orc.oracle_index(pg, seed_len)
    assert copmem_params(seed_len, pg.size) == prm
    ctx = MatchContext(L, seed_len, 2, 0, "c")
    ctx.set_pg_ascii(pg)
    c, p = ctx.export_index(0)
    assert np.array_equal(c, cumm)
    assert np.array_equal(p, positions)
    pgr = revcomp(pg)
    _, cumm_r, positions_r = orc.oracle_index(pgr, seed_len)
    c, p = ctx.export_index(1)
    assert np.array_equal(c, cumm_r)
    assert np.array_equal(p, positions_r)


def test_index_bucket_cap_on_low_complexity_text():
    """poly-A / tandem tracts overflow buckets: the 13 smallest positions must survive, ascending."""
    from pgrc_amd import MatchContext
    rng = np.random.default_rng(5)
    pg = rng.choice(list(b"ACGT"), size=200000).astype(np.uint8)
    pg[1000:9000] = ord("A")
    pg[50000:54000] = np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 4000)
    pg[120000:121500] = np.resize(np.frombuffer(b"TTGCA", dtype=np.uint8), 1500)
    _, cumm, positions = orc.oracle_index(pg, 38)
    assert (np.diff(cumm[:-1].astype(np.int64)) == 13).any()
    ctx = MatchContext(100, 38, 2, 0, "c")
    ctx.set_pg_ascii(pg)
    c, p = ctx.export_index(0)
    assert np.array_equal(c, cumm) and np.array_equal(p, positions)


@pytest.mark.parametrize("variant", ["sweep", "sweep+general", "sweep/0", "own", "own+general"])
def test_index_build_variants(monkeypatch, variant):
    """The ways the records get grouped by bucket (copmem.hip: the scatter passes of idxsweep.hip that hash the text
    themselves, the default -- PGRC_INDEX_CFG=0: without the XCD-aware tile order --; the stable scatter passes of
    idxsort.hip) and both finish kernels give the serial reference index -- on a
    uniform text with realistic partition sizes, and on repeats / low-complexity tracts whose buckets overflow the 13-entry
    cap and whose partitions overflow the fast kernel."""
    from pgrc_amd import MatchContext
    sort, _, fin = variant.partition("+")
    sort, _, cfg = sort.partition("/")
    monkeypatch.setenv("PGRC_INDEX_SORT", sort)
    if cfg:
        monkeypatch.setenv("PGRC_INDEX_CFG", cfg)
    if fin:
        monkeypatch.setenv("PGRC_INDEX_FINISH", fin)
    rng = np.random.default_rng(11)
    texts = []
    pg, _ = make_inputs(40_000_000, 1, 100, seed=3)           # 8 M samples over 2048 partitions
    texts.append((pg, 38))
    lowc = rng.choice(list(b"ACGT"), size=3_000_000).astype(np.uint8)
    lowc[100_000:700_000] = ord("A")                          # one bucket with 120 k records
    lowc[1_000_000:1_400_000] = np.resize(np.frombuffer(b"ACGTTG", dtype=np.uint8), 400_000)
    lowc[2_000_000:2_600_000] = lowc[1_000_000:1_600_000]     # a long repeat: many buckets with 2 entries
    texts.append((lowc, 38))
    texts.append((lowc[:1_500_000], 45))
    for pg, seed_len in texts:
        _, cumm, positions = orc.oracle_index(pg, seed_len)
        ctx = MatchContext(100, seed_len, 2, 0, "c")
        ctx.set_pg_ascii(pg)
        for strand in (0, 1):
            if strand:
                _, cumm, positions = orc.oracle_index(revcomp(pg), seed_len)
            c, p = ctx.export_index(strand)
            assert np.array_equal(c, cumm) and np.array_equal(p, positions), (variant, pg.size, seed_len, strand)


@pytest.mark.parametrize("mode", ["d", "i", "e"])
def test_seedindex_text_segments_and_read_batches(monkeypatch, mode):
    """Modes d / i / e scan a text of 2^32 or more symbols in segments of fewer than 2^32 window starts and take 2^28 or
    more reads in batches (the 64-bit hit record keeps 32 + 28 bits); the knobs force both loops on a small input:
    segment after segment, batch after batch must be the reference's one sequential scan."""
    L = 100
    pg, reads = make_inputs(150_000, 5000, L, seed=77, n_with_n=300, pool_div=64, tandem_every=2)
    seed_len, kmax = (L, 0) if mode == "e" else (38, 2)
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    for seg, batch in ((4096, None), (None, 700), (10_000, 1300), (65_536, 4700)):
        if seg:
            monkeypatch.setenv("PGRC_SEED_SEGMENT", str(seg))
        else:
            monkeypatch.delenv("PGRC_SEED_SEGMENT", raising=False)
        if batch:
            monkeypatch.setenv("PGRC_SEED_READ_BATCH", str(batch))
        else:
            monkeypatch.delenv("PGRC_SEED_READ_BATCH", raising=False)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), o, f"mode {mode} segment {seg} batch {batch}")
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0, devices=[0, 0]), o, f"mode {mode}, 2 shards, segment {seg} batch {batch}")


@pytest.mark.parametrize("mode", ["d", "i", "e"])
@pytest.mark.parametrize("filt", ["0", "1"])
def test_seedindex_scan_filter_on_and_off(monkeypatch, mode, filt):
    """The scan asks a one-bit-per-key-slice filter before it probes the table when keys are sparse against the text
    (seedidx.hip); PGRC_SEED_FILTER forces it on / off: same results, also with reads that hold an N and with tandem repeats."""
    monkeypatch.setenv("PGRC_SEED_FILTER", filt)
    L = 100
    pg, reads = make_inputs(200_000, 6000, L, seed=78, n_with_n=200, pool_div=64, tandem_every=3)
    seed_len, kmax = (L, 0) if mode == "e" else (38, 2)
    assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), orc.oracle_match(mode, pg, reads, seed_len, kmax, 0), f"mode {mode} filter {filt}")


@pytest.mark.parametrize("mode", ["d", "i", "e"])
@pytest.mark.parametrize("sort", ["full", "segments", "segments-long"])
def test_seedindex_table_sort_forms(monkeypatch, mode, sort):
    """The (key, entry) pairs of the table are sorted by all 64 key bits with global passes (small batches) or by two global passes
    over the top 16 bits and one in-LDS sort per segment (large batches; radix.hip k_rx_segments); PGRC_SEED_SORT forces either.
    Same results and hits per strand -- with few reads (most segments empty or of one pair), with tandem repeats, and with a pool
    of repeated reads large enough that the segments holding their keys exceed what a block takes (sorted as ranges of their own)."""
    monkeypatch.setenv("PGRC_SEED_SORT", sort.split("-")[0])
    if sort == "segments-long":    # a segment is first sorted by its keys' top 32 bits and checked: with 8 the check fails and all the passes run
        monkeypatch.setenv("PGRC_TEST_SEGMENT_TOP_BITS", "8")
    else:
        monkeypatch.delenv("PGRC_TEST_SEGMENT_TOP_BITS", raising=False)
    L = 100
    seed_len, kmax = (L, 0) if mode == "e" else (38, 2)
    pg, reads = make_inputs(200_000, 6000, L, seed=79, n_with_n=100, pool_div=64, tandem_every=3)
    assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), orc.oracle_match(mode, pg, reads, seed_len, kmax, 0), f"mode {mode} sort {sort}")
    # 40 000 reads that are copies of 4: every key is held by 10 000 entries
    pg, _ = make_inputs(100_000, 16, L, seed=80)
    reads = np.ascontiguousarray(np.stack([pg[s:s + L] for s in (1000, 20_000, 50_000, 90_000)] * 10_000))
    assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), orc.oracle_match(mode, pg, reads, seed_len, kmax, 0), f"mode {mode} sort {sort}, repeated reads")
    if sort == "segments" and mode == "e":   # more over-full segments than the list of them holds (255): the whole array is sorted by all its bits instead
        pg, _ = make_inputs(400_000, 16, L, seed=82)
        reads = np.ascontiguousarray(np.stack([pg[s:s + L] for s in range(1000, 1000 + 300 * 1200, 1200)] * 9000).reshape(9000, 300, L).transpose(1, 0, 2).reshape(-1, L))
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), orc.oracle_match(mode, pg, reads, seed_len, kmax, 0), f"mode {mode} sort {sort}, 300 keys of 9000 entries")
    if sort != "full" and mode == "d":   # enough entries for segments of ~14 keys each (65 536 segments)
        pg, reads = make_inputs(2_000_000, 300_000, L, seed=81)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0), orc.oracle_match(mode, pg, reads, seed_len, kmax, 0), f"mode {mode} sort {sort}, 900 k entries")


@pytest.mark.parametrize("host_pack", ["1", "0"])
def test_text_upload_both_packers(monkeypatch, host_pack):
    """An ASCII text is packed on the host (AVX2 / scalar, several threads, pinned buffers: the default since round 5) or goes up as
    bytes and is packed by a kernel (PGRC_HOST_PACK=0): the same packed text -- texts that are no multiple of 16 or of a packing
    chunk's share of a thread, and one above 4 Mi symbols (below it the host path is not taken) -- and a symbol outside ACGT is
    refused by both with PGRC_E_SYMBOL, wherever it stands."""
    from pgrc_amd import MatchContext
    monkeypatch.setenv("PGRC_HOST_PACK", host_pack)
    L = 100
    for G in (300_001, 5_000_019):
        pg, reads = make_inputs(G, 4000, L, seed=900 + G % 7)
        assert_same_results(gpu_match("c", pg, reads, 38, 2, 0), orc.oracle_match("c", pg, reads, 38, 2, 0), f"host_pack={host_pack} G={G}")
        for at in (0, G // 3 + 5, G - 1):
            bad = pg.copy()
            bad[at] = ord("N")
            ctx = MatchContext(L, 38, 2, 0, "c")
            with pytest.raises(Exception) as e:
                ctx.set_pg_ascii(bad)
            assert "ACGT" in str(e.value) or getattr(e.value, "code", None) == 5, str(e.value)


CASES = [
    # L, seed, M, mode, G, n
    (100, 38, 50, "c", 400000, 20000),
    (100, 38, 3, "c", 400000, 20000),
    (100, 38, 50, "C", 400000, 20000),
    (150, 38, 50, "c", 400000, 20000),
    (250, 38, 50, "c", 400000, 12000),
    (100, 100, 50, "c", 300000, 10000),
    (150, 150, 3, "c", 300000, 10000),
    (64, 32, 10, "c", 200000, 10000),
    (40, 24, 8, "c", 100000, 8000),
    (255, 45, 20, "c", 300000, 6000),
]


@pytest.mark.parametrize("L,seed_len,M,mode,G,n", CASES)
def test_copmem_parity(L, seed_len, M, mode, G, n):
    pg, reads = make_inputs(G, n, L, seed=1000 + L + seed_len + M)
    kmax = L // M
    kmin = kmax if mode.isupper() else 0
    o = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin)
    g = gpu_match("c", pg, reads, seed_len, kmax, kmin)
    assert_same_results(g, o, f"L={L} seed={seed_len} M={M} {mode}")
    ctr = g["ctx"].counters()
    # The kernel stops a read once nothing can be accepted any more, and a two-pass run with kmin == 0 screens the reads
    # for exact RC alignments first: its work counters are those of the oracle's restatement of that rule / schedule
    # (tests/test_early_stop_rule.py shows on the CPU that neither changes a result) ...
    oe = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin, early_stop=True)
    assert_same_results(oe, o, "early-stop rule")
    legs = [({"PGRC_SCREEN": "0"}, oe, 0), ({"PGRC_SCREEN": "0", "PGRC_EARLY_STOP": "0"}, o, 0)]
    if kmin == 0:
        legs.append(({"PGRC_DUAL": "1"}, None, 2))
    if kmin == 0:
        osc = orc.oracle_match_screened(pg, reads, seed_len, kmax, kmin)
        assert_same_results(osc, o, "screened schedule")
        legs.append(({"PGRC_SCREEN": "1"}, osc, 1))
    else:
        osc = None
        assert ctr["screened"] == 0
    if ctr["screened"] == 2:                         # the default for kmin == 0: ONE query per read over both strands
        od = orc.oracle_match_dual(pg, reads, seed_len, kmax)
        assert_same_results(od, o, "dual scheme (oracle restatement)")
        assert ctr["redo_reads"] <= n // 4 + 50      # (reads whose falses bound exceeded the budget took the two passes)
    else:
        want = osc if ctr["screened"] else oe
        assert ctr["searched"] == want["searched"] and ctr["candidates"] == want["candidates"]
    # ... with the screen switched off, the early-stop oracle's; with both off, the reference's; forced on, the schedule's
    for knobs, want, screened in legs:
        os.environ.update(knobs)
        try:
            g0 = gpu_match("c", pg, reads, seed_len, kmax, kmin)
        finally:
            for k in knobs:
                del os.environ[k]
        assert_same_results(g0, o, str(knobs))
        c0 = g0["ctx"].counters()
        assert c0["screened"] == screened, knobs
        if want is not None:
            assert c0["searched"] == want["searched"] and c0["candidates"] == want["candidates"], knobs
    if orc.have_ref():
        r = orc.ref_match("c", pg, reads, seed_len, kmax, kmin)
        assert_same_results(g, r, "vs real reference")


def test_copmem_forward_only_and_ragged_sizes():
    for n in (0, 1, 63, 64, 65, 257):
        pg, reads = make_inputs(50000, max(n, 1), 100, seed=77 + n)
        reads = reads[:n]
        o = orc.oracle_match("c", pg, reads, 38, 2, 0, rev_compl=False)
        g = gpu_match("c", pg, reads, 38, 2, 0, rev_compl=False)
        assert_same_results(g, o, f"n={n}")


@pytest.mark.parametrize("stage", ["", "0"])
def test_copmem_staged_refills_window_edges(monkeypatch, stage):
    """The match kernel hands out reads from a 32-read window staged in LDS per wave (1024-read chunks): read counts
    around both sizes, reads with N inside the windows, both passes; PGRC_MATCH_STAGE=0 is the unstaged path."""
    if stage:
        monkeypatch.setenv("PGRC_MATCH_STAGE", stage)
    for n, nn in ((31, 0), (32, 3), (33, 0), (1023, 40), (1024, 0), (1025, 1), (2080, 0), (70001, 900)):
        pg, reads = make_inputs(120000, n, 100, seed=900 + n, n_with_n=nn)
        o = orc.oracle_match("c", pg, reads, 38, 33, 0)
        g = gpu_match("c", pg, reads, 38, 33, 0)
        assert_same_results(g, o, f"n={n} nn={nn} stage={stage!r}")


def test_copmem_reads_hanging_over_pg_ends_and_repeats():
    pg, reads = make_inputs(60000, 4000, 100, seed=5, pool_div=64)
    # reads that overlap the Pg ends: their seeds hit but the window is rejected (:517-520)
    for k in range(50):
        reads[k, :60] = pg[-60:]
        reads[50 + k, 40:] = pg[:60]
    o = orc.oracle_match("c", pg, reads, 38, 33, 0)
    g = gpu_match("c", pg, reads, 38, 33, 0)
    assert_same_results(g, o, "ends/repeats")


def test_copmem_n_reads_take_the_byte_path():
    pg, reads = make_inputs(200000, 6000, 100, seed=31, n_with_n=700)
    o = orc.oracle_match("c", pg, reads, 38, 33, 0)
    g = gpu_match("c", pg, reads, 38, 33, 0)
    assert_same_results(g, o, "N reads")
    if orc.have_ref():
        r = orc.ref_match("c", pg, reads, 38, 33, 0, n_nset=700)
        assert_same_results(g, r, "N reads vs reference")


@pytest.mark.parametrize("L,kmax", [(150, 3), (150, 50), (100, 6), (250, 5)])
@pytest.mark.parametrize("inline", ["1", "0"])
def test_dual_kernel_takes_reads_with_few_ns(monkeypatch, L, kmax, inline):
    """Round 4: a read with at most 4 N's is the dual kernel's own (nread_flag 3 + its N positions: the window hash is
    patched where an N falls into a window, every N counts as a mismatch); more N's go the byte path behind it.
    N's at the read's ends, several in one hash step, on fingerprint symbols, in every window; reads of N only; ASCII
    rows and the reference's LQ + N sum set; PGRC_NREAD_INLINE=0 sends every read with an N down the byte path."""
    monkeypatch.setenv("PGRC_DUAL", "1")
    monkeypatch.setenv("PGRC_NREAD_INLINE", inline)
    rng = np.random.default_rng(1000 + L + kmax)
    pg, reads = make_inputs(400_000, 6000, L, seed=500 + L + kmax, n_with_n=0, pool_div=32)
    n = reads.shape[0]
    first_n = n - 1500
    N = ord("N")
    for i in range(first_n, n):
        k = (i - first_n) % 12
        if k < 4:                                      # 1..4 N's anywhere
            for x in rng.choice(L, size=k + 1, replace=False):
                reads[i, x] = N
        elif k == 4:                                   # the read's first and last symbol
            reads[i, 0] = N; reads[i, L - 1] = N
        elif k == 5:                                   # several N's inside one 4-symbol hash step (hashed and fingerprint symbols)
            x = int(rng.integers(0, L // 4 - 1)) * 4
            reads[i, x:x + 4] = N
        elif k == 6:                                   # N's every 40 symbols: (nearly) every window holds one
            reads[i, 5::40][:4] = N
        elif k == 7:                                   # 5..12 N's: the byte path
            for x in rng.choice(L, size=int(rng.integers(5, 13)), replace=False):
                reads[i, x] = N
        elif k == 8:                                   # nothing but N
            reads[i, :] = N
        elif k == 9:                                   # an N in the tail part of the two-stage count
            reads[i, L - 1 - int(rng.integers(0, L % 8 + 1))] = N
        elif k == 10:                                  # an exact read from the text with one N
            st = int(rng.integers(0, pg.size - L)); reads[i] = pg[st:st + L]; reads[i, int(rng.integers(0, L))] = N
        else:                                          # its reverse complement with two
            st = int(rng.integers(0, pg.size - L)); reads[i] = revcomp(pg[st:st + L]); reads[i, rng.choice(L, size=2, replace=False)] = N
    o = orc.oracle_match("c", pg, reads, 38, kmax, 0)
    assert (o["mism"][first_n:] != 255).sum() > 100          # (reads with N do get matched here)
    g = gpu_match("c", pg, reads, 38, kmax, 0)
    assert_same_results(g, o, f"ASCII rows, inline={inline}")
    assert g["ctx"].counters()["screened"] == 2
    g2 = gpu_match("c", pg, reads, 38, kmax, 0, n_nset=n - first_n)
    assert_same_results(g2, o, f"LQ + N sum set, inline={inline}")
    g3 = gpu_match("c", pg, reads, 38, kmax, 0, devices=[0, 0])
    assert_same_results(g3, o, f"two shards, inline={inline}")
    if orc.have_ref() and inline == "1":
        r = orc.ref_match("c", pg, reads, 38, kmax, 0, n_nset=n - first_n)
        assert_same_results(g, r, "vs real reference")


def test_reference_packed_reads_entry_point():
    pg, reads = make_inputs(150000, 3000, 150, seed=8)
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    g = gpu_match("c", pg, reads, 38, 3, 0, packed_ref=True)
    assert_same_results(g, o, "set_reads_packed")


def test_bad_symbols_and_errors():
    from pgrc_amd import MatchContext, PgrcMatchError
    pg, reads = make_inputs(50000, 10, 100, seed=1)
    ctx = MatchContext(100, 38, 2, 0, "c")
    bad = pg.copy()
    bad[777] = ord("N")
    with pytest.raises(PgrcMatchError) as e:
        ctx.set_pg_ascii(bad)
    assert e.value.code == 5
    ctx.set_pg_ascii(pg)
    r2 = reads.copy()
    r2[3, 5] = ord("X")
    with pytest.raises(PgrcMatchError) as e:
        ctx.set_reads_ascii(r2)
    assert e.value.code == 5
    with pytest.raises(PgrcMatchError) as e:
        MatchContext(100, 20, 2, 0, "c")  # "Minimal matching length too short" CopMEMMatcher.cpp:77-80
    assert e.value.code == 2
    with pytest.raises(PgrcMatchError):
        MatchContext(100, 38, 2, 0, "x")


def test_mismatch_extraction():
    pg, reads = make_inputs(200000, 5000, 100, seed=44, n_with_n=300)
    g = gpu_match("c", pg, reads, 38, 33, 0)
    ctx = g["ctx"]
    n = reads.shape[0]
    for flags in (None, (g["rc"] != (np.arange(n) & 1)).astype(np.uint8)):
        cum, codes, offs = ctx.extract_mismatches(flags)
        cnt = np.where(g["mism"] == 255, 0, g["mism"]).astype(np.uint64)
        assert np.array_equal(np.diff(cum), cnt)
        for i in np.flatnonzero(cnt)[:1500]:
            rev = bool(g["rc"][i]) if flags is None else bool(flags[i])
            co, oo = orc.oracle_extract(pg, g["pos"][i], reads[i], g["rc"][i], rev, int(cnt[i]))
            s, e = int(cum[i]), int(cum[i + 1])
            assert np.array_equal(codes[s:e], co) and np.array_equal(offs[s:e], oo), i


def test_two_phase_continuation():
    """-l pre-matching flow: exact-ish copMEM first, then the approximate phase takes the results over
    (ReadsMatchers.cpp:749-779)."""
    from pgrc_amd import mapReadsIntoPg
    pg, reads = make_inputs(200000, 5000, 100, seed=91)
    bitmap, m = mapReadsIntoPg(pg, True, reads, 38, 50, "c", preReadsExactMatchingChars=100, preMatchingMode="c")
    o1 = orc.oracle_match("c", pg, reads, 100, 2, 0)
    # 2nd-phase minMismatches = targetMismatches(1st phase seed) + 1 = 100 // 100 - 1 + 1 (ReadsMatchers.cpp:713, :752)
    o2 = orc.oracle_match("c", pg, reads, 38, 2, 100 // 100 - 1 + 1, state=(o1["pos"], o1["rc"], o1["mism"]))
    assert np.array_equal(m.readMatchPos, o2["pos"])
    assert np.array_equal(m.readMismatchesCount, o2["mism"])
    assert np.array_equal(m.readMatchRC.astype(np.uint8), o2["rc"])
    assert np.array_equal(bitmap, o2["mism"] <= 254)


@pytest.mark.parametrize("screen", ["0", "1", "dual"])
def test_screened_schedule_hard_cases(monkeypatch, screen):
    """The screened schedule of a two-pass run (exact-match screen on the RC text first) where its side conditions
    bite: low-complexity text (capped buckets, falses budgets running out: flagged reads fall back to the real query),
    reads that match both strands exactly (reverse-palindromic inserts), reads with N, a run that continues from an
    earlier phase's results, and a sharded matcher.  Results against the reference-order oracle, work counters against
    the oracle's restatement of the schedule."""
    from pgrc_amd import MatchContext
    if screen != "dual":
        monkeypatch.setenv("PGRC_SCREEN", screen)
    else:                                # the dual kernel, one query per read over both strands (forced: these reads are short)
        monkeypatch.setenv("PGRC_DUAL", "1")
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    # (a) low-complexity text
    unit = rng.choice(acgt, size=37)
    pg = np.tile(unit, 3000)[:100000].copy()
    flips = rng.integers(0, pg.size, size=600)
    pg[flips] = rng.choice(acgt, size=flips.size)
    _, reads = make_inputs(100000, 3000, 100, seed=3)
    for i, st in enumerate(rng.integers(0, pg.size - 100, size=2000)):
        reads[i] = pg[st:st + 100]
        for _ in range(int(rng.integers(0, 4))):
            reads[i, int(rng.integers(0, 100))] = rng.choice(acgt)
    for i, st in enumerate(rng.integers(0, pg.size - 100, size=800)):          # ... and reads from its other strand
        reads[2000 + i] = revcomp(pg[st:st + 100])
        for _ in range(int(rng.integers(0, 3))):
            reads[2000 + i, int(rng.integers(0, 100))] = rng.choice(acgt)
    # (b) a text with reverse-palindromic stretches: reads from them match both strands exactly
    pg2, reads2 = make_inputs(150000, 4000, 100, seed=8, n_with_n=150)
    for k in range(40):
        half = rng.choice(acgt, size=100)
        pal = np.concatenate([half, np.array([comp[int(x)] for x in half[::-1]], dtype=np.uint8)])
        at = 1000 + 3000 * k
        pg2[at:at + 200] = pal
        reads2[k] = pal[50:150]
        reads2[40 + k] = pal[20:120]
    for name, (P, R, kmax) in {"lowcomplexity": (pg, reads, 5), "lowcomplexity_k2": (pg, reads, 2), "palindromes": (pg2, reads2, 2)}.items():
        o = orc.oracle_match("c", P, R, 38, kmax, 0)
        g = gpu_match("c", P, R, 38, kmax, 0)
        assert_same_results(g, o, name)
        ctr = g["ctx"].counters()
        assert ctr["screened"] == (2 if screen == "dual" else int(screen))
        want = orc.oracle_match_screened(P, R, 38, kmax, 0) if screen == "1" else orc.oracle_match("c", P, R, 38, kmax, 0, early_stop=True)
        if screen == "dual":
            assert_same_results(orc.oracle_match_dual(P, R, 38, kmax), o, name + " (oracle restatement of the dual scheme)")
        elif name != "palindromes":      # (reads with N go the byte path: the main kernel's counters leave them out)
            assert ctr["searched"] == want["searched"] and ctr["candidates"] == want["candidates"], name
        g3 = gpu_match("c", P, R, 38, kmax, 0, devices=[0, 0, 0])
        assert_same_results(g3, o, name + " (3 shards)")
    # (c) continuing from an earlier phase (set_results): exact phase first, then k <= 2 with min 0
    o1 = orc.oracle_match("c", pg2, reads2, 100, 0, 0)
    o2 = orc.oracle_match("c", pg2, reads2, 38, 2, 0, state=(o1["pos"], o1["rc"], o1["mism"]))
    ctx = MatchContext(100, 38, 2, 0, "c")
    ctx.set_pg_ascii(pg2)
    ctx.set_reads_ascii(reads2)
    ctx.set_results(o1["pos"], o1["rc"], o1["mism"])
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    assert np.array_equal(pos, o2["pos"]) and np.array_equal(rc, o2["rc"]) and np.array_equal(mism, o2["mism"])


def test_two_index_schedules_fall_back_when_the_second_index_does_not_fit(monkeypatch):
    """PGRC_TEST_NO_SECOND_INDEX makes the build of the second index set report out-of-memory: the run must free what it
    got, take the two passes in the reference's order, and stay on them for later runs of the context."""
    from pgrc_amd import MatchContext
    pg, reads = make_inputs(300000, 6000, 150, seed=77, n_with_n=100)
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    monkeypatch.setenv("PGRC_TEST_NO_SECOND_INDEX", "1")
    ctx = MatchContext(150, 38, 3, 0, "c")
    ctx.set_profiling(True)
    ctx.set_pg_ascii(pg)
    ctx.set_reads_ascii(reads)
    for run in range(2):
        if run == 1:
            monkeypatch.delenv("PGRC_TEST_NO_SECOND_INDEX")      # the context remembers
            ctx.reload_options()                                  # (the environment is only read when asked: the knob is really gone)
        ctx.init_results()
        ctx.run(True)
        pos, rc, mism, hist, matched = ctx.get_results()
        assert np.array_equal(pos, o["pos"]) and np.array_equal(rc, o["rc"]) and np.array_equal(mism, o["mism"]), run
        assert ctx.counters()["screened"] == 0 and ctx.counters()["schedule_downgraded"] == 1
    # another read set (or text) on the same context: the second set may fit now, so the schedule is tried again
    ctx.set_reads_ascii(reads[:5000])
    ctx.init_results()
    ctx.run(True)
    o5 = orc.oracle_match("c", pg, reads[:5000], 38, 3, 0)
    pos, rc, mism, hist, matched = ctx.get_results()
    assert np.array_equal(pos, o5["pos"]) and np.array_equal(rc, o5["rc"]) and np.array_equal(mism, o5["mism"])
    assert ctx.counters()["screened"] == 2 and ctx.counters()["schedule_downgraded"] == 0
    g = gpu_match("c", pg, reads, 38, 3, 0)                        # a fresh context takes the dual kernel
    assert g["ctx"].counters()["screened"] == 2
    assert_same_results(g, o, "fresh context")


def test_device_generators_equal_host_generators():
    import torch
    from pgrc_amd import MatchContext, synth
    G, n, L = 300000, 5000, 150
    g = synth.pg_params(G, seed=4, tandem_every=2)
    pg = synth.pg_host(g)
    rs = synth.reads_params(n, L, seed=4, paired=True)
    reads = synth.reads_host(g, pg, rs)
    d_pg = torch.zeros((G + 15) // 16 + 64, dtype=torch.int32, device="cuda")
    synth.pg_device(g, d_pg.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_pg.cpu().numpy().view(np.uint32)[: (G + 15) // 16], pack2(pg))
    nw, stride = (L + 15) // 16, 5056
    d_rd = torch.zeros(nw * stride, dtype=torch.int32, device="cuda")
    synth.reads_device(g, d_pg.data_ptr(), rs, 0, n, d_rd.data_ptr(), stride)
    torch.cuda.synchronize()
    ctx = MatchContext(L, 38, 3, 0, "c")
    ctx.set_pg_packed_device(d_pg.data_ptr(), G)
    ctx.set_reads_device(d_rd.data_ptr(), n, stride, keep=d_rd)
    ctx.init_results()
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    assert_same_results({"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": matched}, o, "device inputs")


SEED_CASES = [
    # mode, L, seed, M, kmin_shortcut, G, n, n_with_n
    ("e", 100, 100, 50, False, 300000, 15000, 0),
    ("e", 150, 150, 50, False, 300000, 10000, 0),
    ("d", 100, 38, 50, False, 300000, 15000, 0),
    ("d", 100, 38, 3, False, 300000, 15000, 0),
    ("d", 100, 38, 50, True, 300000, 15000, 0),
    ("d", 150, 38, 50, False, 300000, 10000, 0),
    ("d", 250, 45, 20, False, 300000, 6000, 0),
    ("d", 100, 25, 10, False, 200000, 8000, 0),
    ("i", 100, 38, 50, False, 300000, 15000, 0),
    ("i", 100, 38, 3, False, 300000, 15000, 0),
    ("i", 100, 38, 50, True, 300000, 15000, 0),
    ("i", 150, 38, 50, False, 300000, 10000, 0),
    ("i", 250, 45, 20, False, 300000, 6000, 0),
    ("i", 100, 25, 10, False, 200000, 8000, 0),
    ("d", 100, 38, 3, False, 200000, 8000, 600),
    ("i", 100, 38, 3, False, 200000, 8000, 600),
    ("e", 100, 100, 50, False, 200000, 8000, 600),
]


@pytest.mark.parametrize("mode,L,seed_len,M,shortcut,G,n,n_with_n", SEED_CASES)
def test_seedindex_modes_parity(mode, L, seed_len, M, shortcut, G, n, n_with_n):
    pg, reads = make_inputs(G, n, L, seed=2000 + L + seed_len + M + n_with_n)
    reads[100] = reads[50]
    reads[101] = reads[50]
    reads[200, : L // 2] = reads[200, L // 2: 2 * (L // 2)]
    for k in range(20):  # reads hanging over the Pg ends
        reads[300 + k, : L - 7] = pg[-(L - 7):]
        reads[330 + k, 7:] = pg[: L - 7]
    kmax = L // M
    kmin = kmax if shortcut else 0
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, kmin)
    g = gpu_match(mode, pg, reads, seed_len, kmax, kmin)
    assert_same_results(g, o, f"mode {mode} L={L} seed={seed_len} M={M} shortcut={shortcut}")
    # ONE scan of the forward text finds the (window, part) pairs with equal keys of both strands: as many as the oracle's two scans
    assert g["ctx"].counters()["candidates"] == o["candidates"], (g["ctx"].counters()["candidates"], o["candidates"])
    if orc.have_ref():
        r = orc.ref_match(mode, pg, reads, seed_len, kmax, kmin, n_nset=(n if n_with_n else 0))
        assert_same_results(g, r, "vs real reference")


@pytest.mark.parametrize("heavy", ["1", "32", "4096", "1-window", "32-window"])
@pytest.mark.parametrize("mode", ["d", "i", "e"])
def test_seedindex_hits_reduced_by_atomic_minimum(monkeypatch, mode, heavy):
    """Round 4: the reference's sequential rule over a read's hits is taken as what it amounts to -- a lexicographic minimum of
    (count <= kmin ? 0 : count, strand, scan order) -- one atomicMin per acceptable hit on a key per read, no sort and no hit
    records at all (seedidx.hip section 3b).  A window's entries are a range of an entry array: the windows of a stretch of the
    text are expanded by the block that holds them, those with more than PGRC_SEED_HEAVY entries (default 32) by the waves of a
    persistent grid -- 1 / 4096: (nearly) every window takes the one road / the other.  Repeat families and a tandem tract: dozens
    to thousands of hits per read, arriving in arbitrary order, many of them on the same key at once."""
    # (round 5: the heavy windows are grouped by their key and a thread keeps its entry while the key's windows pass; "-window":
    #  a wave per heavy window, the form of rounds 4-5a)
    monkeypatch.setenv("PGRC_SEED_HEAVY", heavy.split("-")[0])
    monkeypatch.setenv("PGRC_SEED_HEAVY_FORM", "window" if heavy.endswith("-window") else "grouped")
    L, seed_len = (100, 25) if mode != "e" else (100, 100)
    pg, reads = make_inputs(300_000, 9000, L, seed=4242, pool_div=4, tandem_every=2)      # repeat families: runs of dozens of hits
    rng = np.random.default_rng(3)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=7)
    pg[50_000:56_000] = np.tile(unit, 1000)[:6000]                                        # a tandem tract: runs of hundreds to thousands
    for k in range(40):
        st = 50_000 + int(rng.integers(0, 5000)); reads[k] = pg[st:st + L]
    kmax = 0 if mode == "e" else L // seed_len - 1
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    g = gpu_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same_results(g, o, f"hits reduced by minimum, mode {mode}")
    if mode != "e":                                   # ... and with a lower bound on the count that ends a read's walk (upper-case modes)
        o2 = orc.oracle_match(mode, pg, reads, seed_len, kmax, kmax)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, kmax), o2, f"kmin = kmax, mode {mode}")
        o3 = orc.oracle_match(mode, pg, reads, seed_len, kmax, 1)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 1), o3, f"kmin = 1, mode {mode}")


@pytest.mark.parametrize("mode,L,seed_len", [("d", 100, 38), ("d", 150, 38), ("i", 150, 38), ("i", 100, 25), ("e", 100, 100), ("e", 150, 150)])
def test_seedindex_reverse_palindromes(mode, L, seed_len):
    """One scan of the forward text serves both strands (canonical keys, seedidx.hip): a window that equals its own reverse
    complement has equal keys and is a hit of BOTH strands for the parts that equal it.  Stretches S + rc(S) planted in the text
    (every window centred on the junction is such a window), reads taken across the junctions in both orientations: results as
    the reference's two scans give them, and the same number of (window, part) pairs with equal keys per strand as the oracle's
    two scans find (its tables hold the HIP path's words for A C G T N: table-dependent collisions on tandem tracts included)."""
    pg, reads = make_inputs(200_000, 4000, L, seed=5150 + L + seed_len)
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    k = 0
    for st in range(10_000, 190_000, 9_000):
        half = rng.choice(acgt, size=2 * L)
        pg[st:st + 2 * L] = half
        pg[st + 2 * L:st + 4 * L] = revcomp(half)
        for d in (-L // 2, -L // 2 + 1, -L // 2 - 3, -19, -seed_len // 2 - L // 2 + L // 2):
            p = st + 2 * L + d - (L // 2 if d == -19 else 0)
            reads[k] = pg[p:p + L]; k += 1
            reads[k] = revcomp(pg[p:p + L]); k += 1
    kmax = 0 if mode == "e" else L // seed_len - 1
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    g = gpu_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same_results(g, o, f"reverse palindromes, mode {mode} L={L}")
    assert g["ctx"].counters()["candidates"] == o["candidates"], (g["ctx"].counters()["candidates"], o["candidates"])
    if orc.have_ref():
        assert_same_results(g, orc.ref_match(mode, pg, reads, seed_len, kmax, 0), "vs real reference")


@pytest.mark.parametrize("mode", ["d", "i", "e"])
def test_seedindex_candidates_at_ten_megabases(mode):
    """The one scan of the forward text against the oracle's two scans at a size where the synthetic Pg has thousands of planted
    repeats and tandem tracts (10 Mbp, 600 k reads of 150 bp): the results, and the number of (window, part) pairs with equal keys
    per strand to the last one -- the table-dependent collisions of the rolling hash on periodic runs included (table(complement) =
    bit reversal of table(symbol), seedidx.hip; with two arbitrary tables a few hundred pairs per 10^8 differed)."""
    L = 150
    pg, reads = make_inputs(10_000_000, 600_000, L, seed=777)
    seed_len, kmax = (L, 0) if mode == "e" else (38, 3)
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    g = gpu_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same_results(g, o, f"10 Mbp, mode {mode}")
    assert g["ctx"].counters()["candidates"] == o["candidates"], (g["ctx"].counters()["candidates"], o["candidates"])


@pytest.mark.parametrize("mode", ["d", "i", "e"])
def test_seedindex_hit_floods_on_low_complexity_text(mode):
    """A poly-A tract and reads taken from it: every window of the tract hits every part of those reads (~1.4 M hits
    from 8 kbp of text), which overflows the scan kernel's block-local hit buffer and the first guess of the hit
    array -- both slow paths must give the reference's result."""
    L = 100
    seed_len = L if mode == "e" else 38
    pg, reads = make_inputs(120000, 3000, L, seed=4242)
    pg[30000:38000] = ord("A")
    pg[60000:64000] = np.resize(np.frombuffer(b"AC", dtype=np.uint8), 4000)
    for k in range(90):
        reads[k] = ord("A")
        if mode != "e" and k % 3:
            reads[k, 5 + k % 50] = ord("C")          # approximate matches inside the tract
    for k in range(90, 130):
        reads[k] = np.resize(np.frombuffer(b"AC" if k % 2 else b"CA", dtype=np.uint8), L)
    kmax = 0 if mode == "e" else 2
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    g = gpu_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same_results(g, o, f"hit flood, mode {mode}")
    assert g["ctx"].counters()["candidates"] == o["candidates"]
    assert g["ctx"].counters()["candidates"][0] > 3 * 3000 * (L // seed_len) + 4096     # the guess was exceeded
    if orc.have_ref():
        assert_same_results(g, orc.ref_match(mode, pg, reads, seed_len, kmax, 0), "hit flood vs real reference")


def test_seedindex_cyclic_equivalence_candidates():
    """seeds longer than 32: the reference's CyclicHash collides deterministically for symbol swaps 32 apart;
    the GPU key must yield the same candidates (see tests/test_oracle_vs_ref.py)."""
    pg, reads0 = make_inputs(120000, 600, 100, seed=77)
    rng = np.random.default_rng(7)
    starts = rng.integers(0, pg.size - 100, size=600)
    for mode in ("d", "i"):
        reads = reads0.copy()
        for k in range(600):
            r = pg[starts[k]: starts[k] + 100].copy()
            q = k % 6
            pairs = [(q, q + 32), (38 + q, 38 + q + 32)] if mode == "d" else [(2 * q, 2 * (q + 32)), (1 + 2 * q, 1 + 2 * (q + 32))]
            for a, b in pairs:
                r[a], r[b] = r[b], r[a]
            reads[k] = r
        o = orc.oracle_match(mode, pg, reads, 38, 33, 0)
        g = gpu_match(mode, pg, reads, 38, 33, 0)
        assert_same_results(g, o, f"cyclic {mode}")
        assert g["matched"] > 550


def test_map_reads_into_pg_factory():
    """mapReadsIntoPg picks the matcher exactly like ReadsMatchers.cpp:715-740."""
    from pgrc_amd import (CopMEMReadsApproxMatcher, DefaultReadsApproxMatcher, DefaultReadsExactMatcher,
                          InterleavedReadsApproxMatcher, PgrcMatchError, mapReadsIntoPg)
    pg, reads = make_inputs(100000, 2000, 100, seed=12)
    for mode, seed_len, cls, okind in (("c", 38, CopMEMReadsApproxMatcher, "c"), ("d", 38, DefaultReadsApproxMatcher, "d"),
                                       ("i", 38, InterleavedReadsApproxMatcher, "i"), ("d", 100, DefaultReadsExactMatcher, "e"),
                                       ("c", 120, CopMEMReadsApproxMatcher, "c")):
        bitmap, m = mapReadsIntoPg(pg, True, reads, seed_len, 50, mode)
        assert type(m) is cls
        o = orc.oracle_match(okind, pg, reads, min(seed_len, 100), 2, 0)
        assert np.array_equal(m.readMatchPos, o["pos"]) and np.array_equal(m.readMatchRC.astype(np.uint8), o["rc"])
        assert m.matchedReadsCount == o["matched"] and np.array_equal(bitmap, o["pos"] != np.uint64(2**64 - 1))
    with pytest.raises(PgrcMatchError):
        mapReadsIntoPg(pg, True, reads, 38, 50, "x")


@pytest.mark.skipif(not orc.have_ref(), reason="needs oracle/_ref (the compiled reference)")
@pytest.mark.parametrize("mode,seed_len,n_nset", [("c", 38, 0), ("c", 38, 500), ("d", 38, 0), ("i", 38, 0), ("e", 100, 0),
                                                  # LQ + N sum set in modes d/i/e: the reference indexes no read
                                                  # (reference quirk 4) and the adapter keeps that
                                                  ("d", 38, 500), ("i", 38, 500), ("e", 100, 500),
                                                  # one ACGNT set holding every read: all of them are indexed
                                                  ("d", 38, 5000), ("i", 38, 5000)])
@pytest.mark.parametrize("entry", [0, 1])
def test_reference_adapter_drop_in(mode, seed_len, n_nset, entry):
    """integration/HipReadsMatcher inside the REFERENCE's matcher hierarchy (compiled against its headers, its own
    PackedConstantLengthReadsSet / SumOfConstantLengthReadsSets feeding it) against the reference's CPU matcher."""
    if not orc.have_adapter():
        pytest.skip("oracle/_ref was built without the adapter")
    pg, reads = make_inputs(200000, 5000, 100, seed=61 + n_nset, n_with_n=n_nset)
    kmax = 2 if mode != "e" else 0
    r = orc.ref_match(mode, pg, reads, seed_len, kmax, 0, n_nset=n_nset)
    a = orc.ref_match_via_adapter(mode, pg, reads, seed_len, kmax, 0, n_nset=n_nset, entry=entry)
    assert_same_results(a, r, f"adapter mode {mode} entry {entry}")


def test_index_of_low_complexity_text():
    """Poly-A / dinucleotide tracts: buckets with tens of thousands of candidates keep exactly their 13 smallest."""
    from pgrc_amd import MatchContext
    rng = np.random.default_rng(11)
    pg = rng.choice(list(b"ACGT"), size=300000).astype(np.uint8)
    pg[20000:120000] = ord("A")                                                   # 100 kbp poly-A
    pg[150000:200000] = np.resize(np.frombuffer(b"AC", dtype=np.uint8), 50000)    # dinucleotide tract
    _, cumm, positions = orc.oracle_index(pg, 38)
    ctx = MatchContext(100, 38, 2, 0, "c")
    ctx.set_pg_ascii(pg)
    c, p = ctx.export_index(0)
    assert np.array_equal(c, cumm) and np.array_equal(p, positions)
    # and matching over such an index
    _, reads = make_inputs(300000, 3000, 100, seed=3)
    for k in range(200):
        s = 20000 + 400 * k
        reads[k] = pg[s: s + 100]
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    ctx.set_reads_ascii(reads)
    ctx.init_results()
    ctx.run(True)
    pos, rc, mism, hist, matched = ctx.get_results()
    assert_same_results({"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": matched}, o, "low complexity")


def test_c_consumer_of_the_abi(tmp_path):
    """tests/c_abi_smoke.c (plain C99) drives the library end to end; its digest equals the oracle's."""
    import os
    import subprocess
    from pgrc_amd import _lib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi_smoke.c"), "-o",
                    str(exe), "-L", libdir, "-lpgrc_match", f"-Wl,-rpath,{libdir}"], check=True)
    env = dict(os.environ)
    tl = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")
    env["LD_LIBRARY_PATH"] = tl + ":" + env.get("LD_LIBRARY_PATH", "")  # same HIP runtime as the rest of the suite
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    g = synth.SynthPg(2024, 200000, 20000, 3000, 8, 4)
    pg = synth.pg_host(g)
    rs = synth.SynthReads(77, 4000, 100, 0, 0)
    reads = synth.reads_host(g, pg, rs)
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    fnv = 1469598103934665603
    for p, m, c in zip(o["pos"].tolist(), o["mism"].tolist(), o["rc"].tolist()):
        fnv = ((fnv ^ p ^ (m << 56) ^ (c << 48)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"matched {o['matched']} of 4000, exact {int(o['hist'][0])}, digest {fnv:016x}" in r.stdout
    mm = int(o["mism"][o["mism"] != 255].astype(np.int64).sum())
    assert f"two shards agree; export: 4000 entries, {mm} mismatches" in r.stdout   # create_multi, packed appends, export from C


@pytest.mark.parametrize("L,seed_len,M,shortcut", [(100, 38, 50, False), (150, 38, 3, False), (250, 45, 20, True), (64, 32, 10, False)])
def test_copmem_64bit_position_kernels_on_small_text(monkeypatch, L, seed_len, M, shortcut):
    """PGRC_FORCE_POS64 selects the kernels that a pseudogenome >= 4 Gi symbols needs (the reference's u64 index
    branch, CopMEMMatcher.cpp:579-586) on a small text; results must not change.  (The real thing is exercised by
    tests/fullscale_parity.py --workload P64, profiles/r01_fullscale_parity_P64.json.)"""
    monkeypatch.setenv("PGRC_FORCE_POS64", "1")
    pg, reads = make_inputs(300000, 12000, L, seed=4000 + L, n_with_n=300)
    kmax = L // M
    kmin = kmax if shortcut else 0
    o = orc.oracle_match("c", pg, reads, seed_len, kmax, kmin)
    g = gpu_match("c", pg, reads, seed_len, kmax, kmin)
    assert_same_results(g, o, f"pos64 L={L}")
