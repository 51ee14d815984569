"""One matcher over several devices in one process (pgrc_match_create_multi) and the packed read-set hand-over
(pgrc_match_append_reads_packed): results must equal the single-device path, the oracle and the reference.

A one-GPU box rehearses the sharded path with the same device listed several times (every shard is its own context
with its own copy of the text and the index; the all-gather then runs as device-to-device copies); the RCCL engine is
exercised with a one-rank communicator; with more than one visible device the real thing runs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle as orc
from pgrc_amd import MatchContext, PgrcMatchError
from pgrc_amd._lib import lib
from util import assert_same_results, gpu_match, make_inputs, pack2, pack_rows

pytestmark = pytest.mark.gpu


def _res(ctx):
    pos, rc, mism, hist, matched = ctx.get_results()
    return {"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": matched}


@pytest.mark.parametrize("mode,k", [("c", 2), ("c", 3), ("c", 8), ("d", 2), ("i", 3), ("e", 2)])
def test_sharded_matcher_equals_single_device(mode, k):
    L = 100
    pg, reads = make_inputs(300017, 9001, L, seed=40 + k, n_with_n=300, paired=True)
    seed_len, kmax = (L, 0) if mode == "e" else (38, 2)
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
    one = gpu_match(mode, pg, reads, seed_len, kmax, 0)
    assert_same_results(one, o, "single device")
    g = gpu_match(mode, pg, reads, seed_len, kmax, 0, devices=[0] * k)
    assert_same_results(g, o, f"{k} shards")
    sh = g["ctx"].shards()
    assert len(sh) == k and sh[0][1] == 0 and sum(s[2] for s in sh) == reads.shape[0]
    assert all(a[1] + a[2] == b[1] for a, b in zip(sh, sh[1:]))
    assert all(s[1] % 2 == 0 for s in sh if s[2])            # PE mates (2q, 2q+1) stay on one shard
    # every shard holds the whole packed text, both strands (gathered, then reverse-complemented locally)
    assert np.array_equal(g["ctx"].export_pg(0), pack2(pg))
    if mode == "c":
        ca, cb = one["ctx"].counters(), g["ctx"].counters()
        for key in ("searched", "candidates", "probes"):
            assert ca[key] == cb[key], key                   # the same work, only split


def test_more_shards_than_reads_and_no_reads_at_all():
    pg, reads = make_inputs(80000, 3, 100, seed=12)
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    g = gpu_match("c", pg, reads, 38, 2, 0, devices=[0] * 8)       # 2, 1 and 0 reads per shard
    assert_same_results(g, o, "3 reads over 8 shards")
    assert [s[2] for s in g["ctx"].shards()] == [2, 1, 0, 0, 0, 0, 0, 0]
    cum, codes, offs = g["ctx"].extract_mismatches()
    assert cum.size == 4 and int(cum[-1]) == int(g["mism"][g["mism"] != 255].sum())
    for mode in ("c", "d"):
        e = gpu_match(mode, pg, reads[:0], 38, 2, 0, devices=[0, 0])
        assert e["pos"].size == 0 and e["matched"] == 0 and int(e["hist"].sum()) == 0
        e = gpu_match(mode, pg, reads[:0], 38, 2, 0)
        assert e["pos"].size == 0 and e["matched"] == 0


def test_sharded_streamed_uploads_cross_shard_boundaries():
    L = 150
    pg, reads = make_inputs(200000, 5003, L, seed=77, n_with_n=211)
    n_n = 211
    n_lq = reads.shape[0] - n_n
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    # ASCII rows in odd-sized blocks
    ctx = MatchContext(L, 38, 3, 0, "c", devices=[0, 0, 0])
    ctx.set_pg_ascii(pg)
    assert lib.pgrc_match_begin_reads(ctx._h, reads.shape[0]) == 0
    at = 0
    for blk in (1, 1000, 777, 2500, 10**9):
        part = np.ascontiguousarray(reads[at:at + blk])
        assert lib.pgrc_match_append_reads_ascii(ctx._h, part.ctypes.data_as(C.c_void_p), part.shape[0]) == 0
        at += part.shape[0]
    assert lib.pgrc_match_end_reads(ctx._h) == 0
    ctx.n = reads.shape[0]
    ctx.init_results()
    ctx.run(True)
    assert_same_results(_res(ctx), o, "streamed ASCII")
    # the LQ + N sum set in the reference's packed layouts (the N set starts inside the last shard)
    g = gpu_match("c", pg, reads, 38, 3, 0, devices=[0, 0, 0], n_nset=n_n)
    assert_same_results(g, o, "packed sum set, sharded")
    # too many / too few rows are refused
    assert lib.pgrc_match_begin_reads(ctx._h, 10) == 0
    part = np.ascontiguousarray(reads[:11])
    assert lib.pgrc_match_append_reads_ascii(ctx._h, part.ctypes.data_as(C.c_void_p), 11) == 6      # E_STATE
    assert lib.pgrc_match_end_reads(ctx._h) == 6


@pytest.mark.parametrize("mode", ["c", "d", "i", "e"])
def test_packed_sum_set_entry_point(mode):
    """f3: the reads arrive in the reference's own packed sets (ACGT 4/byte + ACGNT 3/byte) and are unpacked on the
    device; same results as ASCII rows."""
    for L in (100, 150, 37 * 3, 255):
        if mode != "c" and L not in (100, 150):
            continue
        pg, reads = make_inputs(150000, 4000, L, seed=L, n_with_n=500)
        seed_len, kmax = (L, 0) if mode == "e" else (38 if L >= 100 else 24, 3)
        o = orc.oracle_match(mode, pg, reads, seed_len, kmax, 0)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0, n_nset=500), o, f"sum set L={L}")
        # one ACGNT set holding every read (the nReadsLQ configuration, DividedPCLReadsSets.cpp:10-21)
        assert_same_results(gpu_match(mode, pg, reads, seed_len, kmax, 0, n_nset=reads.shape[0]), o, f"all-ACGNT L={L}")
    # mismatch lists of reads with N come out the same whichever way the reads went in
    pg, reads = make_inputs(150000, 3000, 100, seed=3, n_with_n=400)
    a = gpu_match("c", pg, reads, 38, 33, 0)
    b = gpu_match("c", pg, reads, 38, 33, 0, n_nset=400)
    for x, y in zip(a["ctx"].extract_mismatches(), b["ctx"].extract_mismatches()):
        assert np.array_equal(x, y)


def test_packed_rows_outside_the_code_range_are_refused():
    pg, reads = make_inputs(50000, 100, 100, seed=1)
    ctx = MatchContext(100, 38, 2, 0, "c")
    ctx.set_pg_ascii(pg)
    rows = pack_rows(reads, b"ACGNT")
    rows[7, 3] = 125                                            # 5^3 = 125 codes: 0..124
    with pytest.raises(PgrcMatchError) as e:
        ctx.set_reads_packed_sets([(rows, 100, 5)])
    assert e.value.code == 5
    with pytest.raises(PgrcMatchError) as e:
        ctx.set_reads_packed_sets([(rows, 100, 3)])
    assert e.value.code == 1


def test_golden_nreads_case_through_the_packed_entry():
    from golden_util import load_case
    m, pg, reads, kind, sl, kmax, kmin, gold = load_case("c_L100_s38_M50_nreads")
    has_n = (reads == ord("N")).any(axis=1)
    n_n = int(has_n.sum())
    assert n_n > 0 and has_n[-n_n:].all()                       # the N reads are the tail = the N set of the sum
    assert_same_results(gpu_match(kind, pg, reads, sl, kmax, kmin, m["rev_compl"], n_nset=n_n), gold, "golden, packed")
    assert_same_results(gpu_match(kind, pg, reads, sl, kmax, kmin, m["rev_compl"], n_nset=n_n, devices=[0, 0]), gold,
                        "golden, packed, 2 shards")


def test_sharded_two_phase_flow_and_single_passes():
    L = 100
    pg, reads = make_inputs(250000, 6000, L, seed=9, n_with_n=100)
    first = orc.oracle_match("c", pg, reads, 64, 3, 0)
    want = orc.oracle_match("c", pg, reads, 32, 3, 1, state=(first["pos"], first["rc"], first["mism"]))
    a = gpu_match("c", pg, reads, 64, 3, 0, devices=[0, 0])
    assert_same_results(a, first, "phase 1")
    b = MatchContext(L, 32, 3, 1, "c", devices=[0, 0, 0])
    b.set_pg_ascii(pg)
    b.set_reads_ascii(reads)
    b.set_results(a["pos"], a["rc"], a["mism"])
    b.run_pass(0)
    b.run_pass(1)
    assert_same_results(_res(b), want, "phase 2 as two single passes")


def test_sharded_mismatch_lists():
    pg, reads = make_inputs(200000, 5000, 100, seed=21, n_with_n=333, paired=True)
    one = gpu_match("c", pg, reads, 38, 33, 0)
    many = gpu_match("c", pg, reads, 38, 33, 0, devices=[0, 0, 0])
    assert_same_results(many, one, "sharded")
    n = reads.shape[0]
    for flags in (None, (one["rc"] != (np.arange(n) & 1)).astype(np.uint8)):
        for x, y in zip(one["ctx"].extract_mismatches(flags), many["ctx"].extract_mismatches(flags)):
            assert np.array_equal(x, y)


def test_rccl_engine_with_a_one_rank_communicator(monkeypatch):
    """The RCCL path (dlopen, ncclCommInitAll, grouped in-place ncclAllGather) on one device; a device listed twice
    cannot join a communicator twice and must be refused loudly, not silently rerouted."""
    pg, reads = make_inputs(120000, 2000, 100, seed=2)
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    monkeypatch.setenv("PGRC_ALLGATHER", "rccl")
    g = gpu_match("c", pg, reads, 38, 2, 0, devices=[0])
    assert_same_results(g, o, "rccl, 1 rank")
    assert g["ctx"].counters()["ms_allgather"] > 0
    ctx = MatchContext(100, 38, 2, 0, "c", devices=[0, 0])
    with pytest.raises(PgrcMatchError) as e:
        ctx.set_pg_ascii(pg)
    assert e.value.code == 8 and "ncclCommInitAll" in str(e.value)


def test_all_visible_devices():
    """More than one GPU visible: the real multi-device run (RCCL all-gather over xGMI)."""
    cnt = C.c_int32(0)
    assert lib.pgrc_match_device_count(C.byref(cnt)) == 0 and cnt.value == torch.cuda.device_count()
    if cnt.value < 2:
        pytest.skip("one visible device")
    pg, reads = make_inputs(2_000_000, 200_000, 150, seed=6, n_with_n=1000, paired=True)
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    g = gpu_match("c", pg, reads, 38, 3, 0, devices=list(range(cnt.value)))
    assert_same_results(g, o, f"{cnt.value} devices")
    assert g["ctx"].counters()["ms_allgather"] > 0


def test_single_device_only_entry_points_and_device_restore():
    pg, reads = make_inputs(60000, 500, 100, seed=4)
    ctx = MatchContext(100, 38, 2, 0, "c", devices=[0, 0])
    assert lib.pgrc_match_set_stream(ctx._h, None) == 1
    ctx.set_pg_ascii(pg)
    ctx.set_reads_ascii(reads)
    with pytest.raises(PgrcMatchError):
        ctx.results_device_ptrs()
    with pytest.raises(PgrcMatchError):
        MatchContext(100, 38, 2, 0, "c", devices=[0, 99])
    with pytest.raises(PgrcMatchError):
        MatchContext(100, 38, 2, 0, "c", devices=[])
    assert torch.cuda.current_device() == 0


def test_adapter_honours_PGRC_DEVICES_and_takes_packed_sets(monkeypatch):
    """HipReadsMatcher inside the compiled reference: PGRC_DEVICES selects the multi-device matcher; the reference's
    own PackedConstantLengthReadsSet / SumOfConstantLengthReadsSets go over packed (no getRead)."""
    if not orc.have_adapter():
        pytest.skip("oracle/_ref was built without the adapter")
    r = orc.ref()
    r.pgrc_ref_packed_handovers.restype = C.c_uint64
    pg, reads = make_inputs(200000, 6000, 100, seed=15, n_with_n=400, paired=True)
    want = orc.ref_match("c", pg, reads, 38, 2, 0, n_nset=400)
    before = r.pgrc_ref_packed_handovers()
    got = orc.ref_match_via_adapter("c", pg, reads, 38, 2, 0, n_nset=400)
    assert_same_results(got, want, "sum set, one device")
    assert r.pgrc_ref_packed_handovers() == before + 1
    monkeypatch.setenv("PGRC_DEVICES", "0,0,0")
    got = orc.ref_match_via_adapter("c", pg, reads, 38, 2, 0, n_nset=400)
    assert_same_results(got, want, "sum set, PGRC_DEVICES=0,0,0")
    got = orc.ref_match_via_adapter("c", pg, reads, 38, 2, 0, n_nset=reads.shape[0], entry=0)
    assert_same_results(got, want, "one ACGNT set, PGRC_DEVICES=0,0,0")
    assert r.pgrc_ref_packed_handovers() == before + 3
    monkeypatch.setenv("PGRC_DEVICES", "all")
    got = orc.ref_match_via_adapter("d", pg, reads[:5000], 38, 2, 0)
    assert_same_results(got, orc.ref_match("d", pg, reads[:5000], 38, 2, 0), "mode d, PGRC_DEVICES=all")
