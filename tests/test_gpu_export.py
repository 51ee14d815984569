"""Row f1 on the GPU: the export streams made on the device (pgrc_match_export_pg_order / _export_entries) against the
oracle's restatement and -- through HipReadsMatcher inside the compiled reference -- against the reference's own
export: identical stream files and identical archive bytes."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
import export_util as xu
from util import gpu_match

pytestmark = pytest.mark.gpu

CASES = {
    "se": dict(seed=1),
    "pe_pairfile": dict(seed=2, paired=True),
    "short_list": dict(seed=3, short_list=True),
    "no_list": dict(seed=4, empty_list=True),
    "L250": dict(seed=5, L=250, n=6000, list_gap=110),
    "L37": dict(seed=6, L=37, n=5000, list_gap=20, n_with_n=100),
}


def _case(name):
    kw = dict(CASES[name])
    pair = kw.pop("paired", False)
    return xu.export_case(paired=pair, **kw), pair


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_streams_equal_oracle_streams(name):
    case, pair = _case(name)
    L = case["L"]
    seed_len = 38 if L >= 100 else 24
    kmax = L // 3
    n = case["reads"].shape[0]
    g = gpu_match("c", case["pg"], case["reads"], seed_len, kmax, 0, n_nset=case["n_n"])
    res = {k: g[k] for k in ("pos", "rc", "mism")}
    order = xu.stable_order(res["pos"])          # any order by position will do here: both sides get the same one
    for byte_mode in (True, False):
        for with_org in (True, False):
            want = xu.oracle_export_pg_order(case, res, order, pair_file=pair, byte_mode=byte_mode, with_read_org=with_org)
            got = g["ctx"].export_pg_order(order, case["list_off"], case["list_org"], case["list_rc"],
                                           case["read_org"] if with_org else None, pair, byte_mode)
            for k in xu.STREAMS:
                assert np.array_equal(got[k], want[k]), (name, byte_mode, with_org, k)
            assert got["last_pos"] == want["last_pos"]
    # the order made on the device (order = NULL): ascending position, reads at one position by ascending index
    want = xu.oracle_export_pg_order(case, res, order, pair_file=pair)
    got = g["ctx"].export_pg_order(None, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], pair, True)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), (name, "device-made order", k)
    # no RC stream on the old list
    c2 = dict(case, list_rc=None)
    want = xu.oracle_export_pg_order(c2, res, order, pair_file=pair)
    got = g["ctx"].export_pg_order(order, case["list_off"], case["list_org"], None, case["read_org"], pair, True)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), (name, "no rc", k)
    # original order: a caller-made entry list with fillers and unmatched reads left out
    er, eo = xu.original_order_entries(case["read_org"], res["mism"] != 255, case["total"], pair, n - case["n_n"])
    want = xu.oracle_export_entries(case, res, er, eo, pair_file=pair)
    got = g["ctx"].export_entries(er, eo, pair, True)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), (name, "entries", k)
    # ... and with the entry list made on the device from the reads' original indexes alone
    for byte_mode in (True, False):
        want = xu.oracle_export_entries(case, res, er, eo, pair_file=pair, byte_mode=byte_mode)
        for pair_mode in (False, True):       # pairFileMode (class-major order) is independent of the paired-file rule
            er2, eo2 = xu.original_order_entries(case["read_org"], res["mism"] != 255, case["total"], pair_mode, n - case["n_n"])
            want2 = want if pair_mode == pair else xu.oracle_export_entries(case, res, er2, eo2, pair_file=pair, byte_mode=byte_mode)
            got = g["ctx"].export_original_order(case["read_org"], case["total"], pair_mode, pair, byte_mode)
            for k in xu.STREAMS:
                assert np.array_equal(got[k], want2[k]), (name, "original order on the device", byte_mode, pair_mode, k)


def test_export_edge_cases():
    case, _ = _case("se")
    g = gpu_match("c", case["pg"], case["reads"], 38, 33, 0, n_nset=case["n_n"])
    res = {k: g[k] for k in ("pos", "rc", "mism")}
    # nothing matched + a list; matches + nothing else; nothing at all
    none = np.zeros(0, dtype=np.uint32)
    got = g["ctx"].export_pg_order(none, case["list_off"], case["list_org"], case["list_rc"], case["read_org"])
    want = xu.oracle_export_pg_order(case, res, none)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), k
    assert got["off"].size == case["list_off"].size and np.array_equal(got["off"], case["list_off"])
    got = g["ctx"].export_entries(none, none)
    assert all(got[k].size == 0 for k in xu.STREAMS)
    from pgrc_amd import MatchContext, PgrcMatchError
    # bad arguments come back as errors, not as device faults: an index beyond the read set, a read without a match in
    # order[], original indexes out of range or used twice
    n = case["reads"].shape[0]
    unmatched = np.flatnonzero(res["mism"] == 255)
    assert unmatched.size
    for bad in (np.array([n], dtype=np.uint32), unmatched[:1].astype(np.uint32)):
        with pytest.raises(PgrcMatchError):
            g["ctx"].export_pg_order(bad, case["list_off"], case["list_org"], case["list_rc"], case["read_org"])
    ro = case["read_org"].copy()
    with pytest.raises(PgrcMatchError):
        g["ctx"].export_original_order(ro, int(ro.max()))            # total too small
    ro[1] = ro[0]
    with pytest.raises(PgrcMatchError):
        g["ctx"].export_original_order(ro, case["total"])            # two reads, one original index
    many = MatchContext(100, 38, 33, 0, "c", devices=[0, 0])
    with pytest.raises(PgrcMatchError):
        many.export_entries(none, none)             # nothing matched yet


@pytest.mark.parametrize("name", ["se", "pe_pairfile", "L37"])
def test_export_from_a_matcher_over_several_shards(name):
    """a multi-device context exports from its first device after gathering the shards' results, reads and N side lists"""
    case, pair = _case(name)
    L = case["L"]
    seed_len, kmax = (38 if L >= 100 else 24), L // 3
    n = case["reads"].shape[0]
    one = gpu_match("c", case["pg"], case["reads"], seed_len, kmax, 0, n_nset=case["n_n"])
    many = gpu_match("c", case["pg"], case["reads"], seed_len, kmax, 0, n_nset=case["n_n"], devices=[0, 0, 0])
    order = xu.stable_order(one["pos"])
    a = one["ctx"].export_pg_order(order, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], pair, True)
    b = many["ctx"].export_pg_order(order, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], pair, True)
    er, eo = xu.original_order_entries(case["read_org"], one["mism"] != 255, case["total"], pair, n - case["n_n"])
    a2 = one["ctx"].export_entries(er, eo, pair, True)
    b2 = many["ctx"].export_entries(er, eo, pair, True)
    c2 = many["ctx"].export_original_order(case["read_org"], case["total"], pair, pair, True)
    for k in xu.STREAMS:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a2[k], b2[k]) and np.array_equal(a2[k], c2[k]), (name, k)


@pytest.mark.parametrize("name", ["se", "pe_pairfile", "short_list", "no_list"])
@pytest.mark.parametrize("preserve_order", [False, True])
def test_adapter_export_equals_reference_export(tmp_path, name, preserve_order):
    """HipReadsMatcher's device export inside the compiled reference vs the reference's own export: the six stream
    files and the compressed bytes handed to the archive stream."""
    if not orc.have_adapter():
        pytest.skip("oracle/_ref was built without the adapter")
    r = orc.ref()
    r.pgrc_ref_device_exports.restype = C.c_uint64
    case, pair = _case(name)
    kw = dict(kmax=33, preserve_order=preserve_order, pair_file_mode=pair, rev_compl_pair_file=pair)
    want = xu.ref_export_run(case, str(tmp_path / "cpu"), 0, **kw)
    before = r.pgrc_ref_device_exports()
    got = xu.ref_export_run(case, str(tmp_path / "gpu"), 1, **kw)
    assert r.pgrc_ref_device_exports() == before + 1          # the device path ran (no silent fallback)
    for k in list(xu.STREAMS) + ["archive"]:
        assert got[k] == want[k], (name, preserve_order, k)
    assert len(want["archive"]) > 1000


def test_adapter_export_with_a_parallel_sort(tmp_path):
    if not orc.have_adapter():
        pytest.skip("oracle/_ref was built without the adapter")
    case = xu.export_case(seed=7, n=60_000, G=500_000, dups=3000)
    want = xu.ref_export_run(case, str(tmp_path / "cpu"), 0, threads=8)
    got = xu.ref_export_run(case, str(tmp_path / "gpu"), 1, threads=8)
    for k in xu.STREAMS:
        assert got[k] == want[k], k


@pytest.mark.parametrize("n,G,dups", [(70_000, 400_000, 30_000), (9_000, 3_000, 0), (8_192, 200_000, 100), (1, 1_000, 0)])
def test_device_made_position_order(n, G, dups):
    """The hand-written stable radix sort behind pgrc_match_export_pg_order(order = NULL) (radix.hip): several tiles, tile
    edges, many reads at one position (copies of reads; a text shorter than the read count), one read.  The streams must be
    those of the stable order (position, then read index)."""
    L = 100 if G >= 1000 else 50
    case = xu.export_case(seed=900 + n % 97, G=G, n=n, L=L, n_with_n=min(200, n // 10), dups=dups, list_gap=40)
    g = gpu_match("c", case["pg"], case["reads"], 38, L // 3, 0, n_nset=case["n_n"])
    res = {k: g[k] for k in ("pos", "rc", "mism")}
    order = xu.stable_order(res["pos"])
    if n > 1:
        assert order.size > n // 2
    want = xu.oracle_export_pg_order(case, res, order)
    got = g["ctx"].export_pg_order(None, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], False, True)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), (n, G, k)
    assert got["last_pos"] == want["last_pos"]
    many = gpu_match("c", case["pg"], case["reads"], 38, L // 3, 0, n_nset=case["n_n"], devices=[0, 0, 0])
    got = many["ctx"].export_pg_order(None, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], False, True)
    for k in xu.STREAMS:
        assert np.array_equal(got[k], want[k]), (n, G, k, "three shards")


def test_order_null_is_only_accepted_without_matched_reads_or_with_the_flag():
    """pgrc_export_pg_order_args.order == NULL: with n_matched > 0 and without order_on_device it is a caller's
    mistake (PGRC_E_PARAM, not silently another tie order); with n_matched == 0 it means "no matched read" -- what
    HipReadsMatcher hands over on its host-sort road when nothing matched (an empty vector's data() is NULL) -- and the export
    is the old list alone; with order_on_device the library makes the order."""
    from pgrc_amd import _lib, PgrcMatchError
    case, pair = _case("se")
    L = case["L"]
    g = gpu_match("c", case["pg"], case["reads"], 38, L // 3, 0, n_nset=case["n_n"])
    ctx = g["ctx"]
    keep = [np.ascontiguousarray(case["list_off"], dtype=np.uint8), np.ascontiguousarray(case["list_org"], dtype=np.uint32)]

    def call(n_matched, on_device):
        a = _lib.ExportPgOrderArgs()
        a.n_matched, a.order_on_device = n_matched, on_device
        a.list_off, a.list_org_idx, a.list_count = keep[0].ctypes.data, keep[1].ctypes.data, keep[0].size
        a.byte_per_read_length = 1
        st = _lib.ExportStreams()
        rc = _lib.lib.pgrc_match_export_pg_order(ctx._h, C.byref(a), C.byref(st))
        n = int(st.n_entries)
        _lib.lib.pgrc_match_free_export(C.byref(st))
        return rc, n
    rc, _ = call(5, 0)
    assert rc == 1                                   # PGRC_E_PARAM
    rc, n = call(0, 0)
    assert rc == 0 and n == keep[0].size             # the old list alone
    rc, n = call(0, 1)
    assert rc == 0 and n == keep[0].size + int((g["mism"] != 255).sum())
    # a run in which no read matches at all, on both roads
    reads = np.full((64, L), ord("A"), dtype=np.uint8)
    reads[:, ::2] = ord("C")
    g0 = gpu_match("c", case["pg"], reads, 38, 0, 0)
    if int((g0["mism"] != 255).sum()) == 0:
        for order in (np.zeros(0, dtype=np.uint32), None):
            got = g0["ctx"].export_pg_order(order, case["list_off"], case["list_org"], case["list_rc"], None, False, True)
            assert got["org_idx"].size == keep[0].size
