"""Row f3 on the GPU: making the packed read sets through the C ABI of include/pgrc_reads.h, against the oracle and the
compiled reference, and through integration/HipDividedReadsSets inside the reference's own classes."""
import numpy as np
import pytest

import oracle as orc
from divide_util import (COMBOS, make_fastq, make_records, oracle_divide, oracle_divide_fastq, ref_divide, ref_divide_files, same)

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
    # error_limit 0 in the simplified mode: the reference tests quality[L], the string's terminating 0 -- never "high":
    # every read without an N goes to the LQ set (round 4: accepted like the reference, round 3 refused it); and a quality
    # byte of 128 or more is a negative char there, never above '#'
    g0 = gpu_divide(reads, quals, 0.0, True, True, False)
    assert g0["n_hq"] == 0 and same(g0, oracle_divide(reads, quals, 0.0, True, True, False)) is None
    hi = quals.copy()
    hi[:, 90] = 200
    gh = gpu_divide(reads, hi, 0.1, True, True, False)
    assert same(gh, oracle_divide(reads, hi, 0.1, True, True, False)) is None and gh["n_hq"] == 0


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


def _concat(parts):
    """division results of consecutive pieces -> one result (indexes shifted by the records before)"""
    out = {k: parts[0][0][k] for k in ("symbols", "row_bytes")}
    base = 0
    acc = {k: [] for k in ("hq_rows", "lq_rows", "n_rows", "lq_index", "n_index")}
    tot = {"n_hq": 0, "n_lq": 0, "n_n": 0}
    for res, nrec in parts:
        for k in ("hq_rows", "lq_rows", "n_rows"):
            acc[k].append(res[k])
        acc["lq_index"].append(res["lq_index"] + np.uint32(base))
        acc["n_index"].append(res["n_index"] + np.uint32(base))
        for k in tot:
            tot[k] += res[k]
        base += nrec
    out.update(tot)
    out.update({k: np.concatenate(v) for k, v in acc.items()})
    return out, base


def gpu_divide_fastq(text, pair_text, rev, L, combo, piece=None):
    """the whole text in one call, or fed in pieces of `piece` bytes the way a file reader would (what a call leaves
    unconsumed is kept and the next piece appended)"""
    from pgrc_amd import DividedPCLReadsSets
    d = DividedPCLReadsSets(L, *combo)
    if piece is None:
        res, nrec, used, pused = d.divide_fastq(text, pair_text, rev, final=True)
        d.close()
        return res, nrec
    parts, buf, at = [], [b"", b""], [0, 0]
    src = [text, pair_text if pair_text is not None else b""]
    while True:
        for f in range(2 if pair_text is not None else 1):
            buf[f] += src[f][at[f]: at[f] + piece]
            at[f] += piece
        final = at[0] >= len(src[0]) and at[1] >= len(src[1])
        res, nrec, used, pused = d.divide_fastq(buf[0], buf[1] if pair_text is not None else None, rev, final=final)
        parts.append((res, nrec))
        buf[0], buf[1] = buf[0][used:], buf[1][pused:]
        if final:
            break
    d.close()
    return _concat(parts)


@pytest.mark.parametrize("paired,rev", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("crlf,trailing", [(False, True), (True, True), (False, False)])
def test_divide_fastq_text_parity(paired, rev, crlf, trailing):
    """FASTQ text parsed on the device: lines found by all threads, records' rows copied out, every second record of a pair
    reverse-complemented -- against the oracle's line reader; whole text at once and in pieces of odd sizes"""
    L, n = 100, 6000
    reads, quals = make_records(seed=13, n=n, L=L)
    a = make_fastq(reads[0::2] if paired else reads, quals[0::2] if paired else quals, seed=1, crlf=crlf, trailing_newline=trailing)
    b = make_fastq(reads[1::2], quals[1::2], seed=2, crlf=crlf, trailing_newline=trailing) if paired else None
    for combo in (COMBOS[1], COMBOS[5], COMBOS[8]):
        o = oracle_divide_fastq(a, b, rev, L, combo)
        g, nrec = gpu_divide_fastq(a, b, rev, L, combo)
        assert nrec == n and same(g, o) is None, (combo, same(g, o))
    for piece in (997, 65536, 300_001):
        g, nrec = gpu_divide_fastq(a, b, rev, L, COMBOS[5], piece=piece)
        assert nrec == n and same(g, oracle_divide_fastq(a, b, rev, L, COMBOS[5])) is None, piece


@pytest.mark.parametrize("n_a,n_b", [(20, 300), (300, 20), (150, 151), (0, 40)])
def test_divide_fastq_pair_files_of_different_lengths(n_a, n_b):
    """The reference stops at the first exhausted file of a pair.  Fed in pieces, the text of the shorter file ends while the
    other goes on (final_piece bits 2 / 4): the call that takes the last records says so (pgrc_divider_last_was_terminal)
    and the reader stops there -- it does not have to grow the longer file's window to its end (round 4)."""
    from pgrc_amd import DividedPCLReadsSets
    L = 80
    reads, quals = make_records(seed=77, n=n_a + n_b, L=L)
    a = make_fastq(reads[:n_a], quals[:n_a], seed=1) if n_a else b""
    b = make_fastq(reads[n_a:], quals[n_a:], seed=2)
    combo = COMBOS[5]
    o = oracle_divide_fastq(a, b, True, L, combo)
    for piece in (700, 4096, 1 << 20):
        d = DividedPCLReadsSets(L, *combo)
        parts, buf, at, src, calls = [], [b"", b""], [0, 0], [a, b], 0
        while True:
            bits = 0
            for f in range(2):
                buf[f] += src[f][at[f]: at[f] + piece]
                at[f] += piece
                if at[f] >= len(src[f]):
                    bits |= 2 << f
            res, nrec, used, pused = d.divide_fastq(buf[0], buf[1], True, final=bits)
            parts.append((res, nrec))
            buf[0], buf[1] = buf[0][used:], buf[1][pused:]
            calls += 1
            if d.last_was_terminal():
                break
            assert calls < 10_000
        d.close()
        g, nrec = _concat(parts)
        assert nrec == (2 * n_a if n_a <= n_b else 2 * n_b + 1), (piece, nrec)
        assert same(g, o) is None, (piece, same(g, o))
        # the reader stopped about where the shorter file ended, not at the end of the longer one
        assert max(at) <= (min(n_a, n_b) + 2) * (2 * L + 40) + 3 * piece, (piece, at)


def test_divide_fastq_edges():
    from pgrc_amd import DividedPCLReadsSets, PgrcMatchError
    L = 64
    reads, quals = make_records(seed=9, n=41, L=L)
    combo = (0.2, False, True, False)
    # the second file one record short: the reference stops after the first file's extra read
    a, b = make_fastq(reads[0::2], quals[0::2], seed=1), make_fastq(reads[1::2], quals[1::2], seed=2)
    g, nrec = gpu_divide_fastq(a, b, True, L, combo)
    assert nrec == 41 and same(g, oracle_divide_fastq(a, b, True, L, combo)) is None
    d = DividedPCLReadsSets(L, *combo)
    res, nrec, used, _ = d.divide_fastq(b"", None, False, final=True)                 # nothing at all
    assert nrec == 0 and used == 0
    res, nrec, used, _ = d.divide_fastq(a[:50], None, False, final=False)            # less than a record: wait for more
    assert nrec == 0 and used == 0
    cut = a[: a.rstrip(b"\n").rfind(b"\n+")]
    with pytest.raises(PgrcMatchError):
        d.divide_fastq(cut, None, False, final=True)                                  # ends inside a record
    longer = a.replace(reads[0].tobytes(), reads[0].tobytes() + b"A", 1)
    with pytest.raises(PgrcMatchError):
        d.divide_fastq(longer, None, False, final=True)                               # a read of another length
    shorter = a.replace(reads[2].tobytes(), reads[2].tobytes()[:-1], 1)
    with pytest.raises(PgrcMatchError):
        d.divide_fastq(shorter, None, False, final=True)
    res, nrec, _, _ = d.divide_fastq(a, None, False, final=True)                      # the context survives the errors
    assert nrec == 21 and same(res, oracle_divide_fastq(a, None, False, L, combo)) is None
    d.close()


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
@pytest.mark.parametrize("piece", [64, 5000, 1 << 20])
def test_divide_fastq_files_through_the_adapter(tmp_path, monkeypatch, piece):
    """HipDividedReadsSets::getQualityDivisionBasedReadsSetsFromFastq (files read in pieces, text parsed on the device) against
    the reference's managed iterator + factory on the same files: single file and pair, second file reverse-complemented"""
    if not hasattr(orc.ref(), "pgrc_ref_divide_files"):
        pytest.skip("oracle/_ref was built without the file harness")
    monkeypatch.setenv("PGRC_FASTQ_PIECE", str(piece))
    L, n = 100, 1200 if piece >= 5000 else 200
    reads, quals = make_records(seed=21, n=n, L=L)
    (tmp_path / "se.fq").write_bytes(make_fastq(reads, quals, seed=5))
    (tmp_path / "a.fq").write_bytes(make_fastq(reads[0::2], quals[0::2], seed=1, crlf=True))
    (tmp_path / "b.fq").write_bytes(make_fastq(reads[1::2], quals[1::2], seed=2, crlf=True, trailing_newline=False))
    for combo in (COMBOS[1], COMBOS[5], COMBOS[8]):
        for src, pair, rev in ((tmp_path / "se.fq", None, False), (tmp_path / "a.fq", tmp_path / "b.fq", True), (tmp_path / "a.fq", tmp_path / "b.fq", False)):
            a = ref_divide_files(src, pair, rev, L, n, combo, use_adapter=True)
            r = ref_divide_files(src, pair, rev, L, n, combo)
            assert same(a, r) is None, (combo, pair is not None, rev, same(a, r))


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
def test_divide_fasta_file_takes_the_reference_iterator(tmp_path):
    """a FASTA file through the adapter's file entry: not FASTQ (first byte), so the records come from the reference's own
    iterator and only classification + packing run on the device -- same sets as the reference"""
    if not hasattr(orc.ref(), "pgrc_ref_divide_files"):
        pytest.skip("oracle/_ref was built without the file harness")
    L, n = 100, 900
    reads, _ = make_records(seed=31, n=n, L=L)
    (tmp_path / "r.fa").write_bytes(b"".join(b">r%d\n" % i + reads[i].tobytes() + b"\n" for i in range(n)))
    for combo in COMBOS[:4]:
        a = ref_divide_files(tmp_path / "r.fa", None, False, L, n, combo, use_adapter=True)
        assert same(a, ref_divide_files(tmp_path / "r.fa", None, False, L, n, combo)) is None, combo
        assert a["n_hq"] + a["n_lq"] + a["n_n"] == n
