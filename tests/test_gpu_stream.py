"""The pipelined hand-over (pgrc_match_prepare_index / _stream_begin / _stream_end, pgrc_amd/csrc/stream.hip): blocks of
reads matched while the next ones are uploaded, results downloaded block by block, index builds started ahead of the run.
Everything must equal the plain sequence init_results + run(True) + get_results -- and so the oracle."""
import numpy as np
import pytest

import oracle as orc
from util import assert_same_results, gpu_match, make_inputs, pack_rows

pytestmark = pytest.mark.gpu


def _streamed(pg, reads, seed_len, kmax, n_nset=0, blocks=1, ascii_rows=False, prepare=True):
    from pgrc_amd import MatchContext
    ctx = MatchContext(reads.shape[1], seed_len, kmax, 0, "c")
    ctx.set_pg_ascii(pg)
    if prepare:
        ctx.prepare_index(True)
    n_lq = reads.shape[0] - n_nset
    if ascii_rows:
        sets = [(reads, reads.shape[0], 0)]
    else:
        sets = [(pack_rows(reads[:n_lq]), n_lq, 4)]
        if n_nset:
            sets.append((pack_rows(reads[n_lq:], b"ACGNT"), n_nset, 5))
    pos, rc, mism, hist, matched = ctx.match_streamed(sets, blocks=blocks)
    return {"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": matched, "ctx": ctx}


@pytest.mark.parametrize("L,kmax,n_nset,blocks,dual", [(150, 3, 0, 1, None), (150, 3, 300, 4, None), (150, 50, 200, 3, None),
                                                      (100, 2, 250, 5, None), (100, 2, 0, 2, "1"), (250, 5, 100, 3, None)])
def test_streamed_run_equals_plain_run(monkeypatch, L, kmax, n_nset, blocks, dual):
    if dual:
        monkeypatch.setenv("PGRC_DUAL", dual)
    pg, reads = make_inputs(400_000, 30_000, L, seed=L + kmax, n_with_n=n_nset, pool_div=8, tandem_every=2)
    o = orc.oracle_match("c", pg, reads, 38, kmax, 0)
    plain = gpu_match("c", pg, reads, 38, kmax, 0, n_nset=n_nset or None)
    assert_same_results(plain, o, "plain run")
    g = _streamed(pg, reads, 38, kmax, n_nset, blocks)
    assert_same_results(g, o, f"streamed, {blocks} blocks per set")
    # the context is left as a plain run leaves it: results on the device, counters, exports
    pos, rc, mism, hist, matched = g["ctx"].get_results()
    assert np.array_equal(pos, o["pos"]) and np.array_equal(rc, o["rc"]) and np.array_equal(mism, o["mism"]) and matched == o["matched"]
    assert g["ctx"].counters()["screened"] == plain["ctx"].counters()["screened"]
    cum, codes, offs = g["ctx"].extract_mismatches(None)
    cum2, codes2, offs2 = plain["ctx"].extract_mismatches(None)
    assert np.array_equal(cum, cum2) and np.array_equal(codes, codes2) and np.array_equal(offs, offs2)


def test_streamed_run_of_ascii_rows_and_without_prepare():
    pg, reads = make_inputs(300_000, 20_000, 150, seed=5, n_with_n=400)
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    assert_same_results(_streamed(pg, reads, 38, 3, blocks=3, ascii_rows=True, prepare=False), o, "ASCII rows, index built by stream_begin")


def test_prepared_index_serves_the_next_plain_run_once():
    """prepare_index builds both indexes ahead; the next two-strand run uses them (no index time), the one after builds
    its own again; a new text invalidates them"""
    from pgrc_amd import MatchContext
    pg, reads = make_inputs(300_000, 10_000, 150, seed=9)
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    ctx = MatchContext(150, 38, 3, 0, "c")
    ctx.set_profiling(True)
    ctx.set_pg_ascii(pg)
    ctx.set_reads_ascii(reads)
    ctx.prepare_index(True)
    for k in range(2):
        ctx.init_results()
        ctx.run(True)
        pos, rc, mism, hist, matched = ctx.get_results()
        assert np.array_equal(pos, o["pos"]) and np.array_equal(mism, o["mism"]) and np.array_equal(rc, o["rc"]), k
    pg2, _ = make_inputs(200_000, 1, 150, seed=10)
    ctx.prepare_index(True)
    ctx.set_pg_ascii(pg2)                      # the prepared indexes describe the old text: they must not be used
    ctx.init_results()
    ctx.run(True)
    o2 = orc.oracle_match("c", pg2, reads, 38, 3, 0)
    pos, rc, mism, hist, matched = ctx.get_results()
    assert np.array_equal(pos, o2["pos"]) and np.array_equal(mism, o2["mism"])


def test_streaming_refuses_what_it_does_not_cover():
    from pgrc_amd import MatchContext, PgrcMatchError
    pg, reads = make_inputs(100_000, 2000, 100, seed=3)
    ctx = MatchContext(100, 38, 2, 0, "d")
    ctx.set_pg_ascii(pg)
    with pytest.raises(PgrcMatchError):
        ctx.prepare_index(True)
    with pytest.raises(PgrcMatchError):
        ctx.match_streamed([(pack_rows(reads), reads.shape[0], 4)])
    ctx.set_reads_ascii(reads)                 # ... and is fine afterwards
    ctx.init_results()
    ctx.run(True)
    o = orc.oracle_match("d", pg, reads, 38, 2, 0)
    assert np.array_equal(ctx.get_results()[0], o["pos"])
    two = MatchContext(100, 38, 2, 0, "c", devices=[0, 0])
    two.set_pg_ascii(pg)
    with pytest.raises(PgrcMatchError):
        two.prepare_index(True)
