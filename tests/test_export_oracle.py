"""Row f1 on the CPU: the oracle's restatement of the export against the compiled reference's own export
(stream files of SeparatedPseudoGenomeOutputBuilder::build), and the adapter's position order against the reference's
sort of the matched reads -- ties, reads with N, the paired-file rule, a reads list that ends early, no list at all."""
import numpy as np
import pytest

import oracle as orc
import export_util as xu

needs_ref = pytest.mark.skipif(not orc.have_adapter(), reason="oracle/_ref (with the adapter) not built")

CASES = {
    "se": dict(seed=1),
    "pe_pairfile": dict(seed=2, paired=True),
    "short_list": dict(seed=3, short_list=True),
    "no_list": dict(seed=4, empty_list=True),
    "L250": dict(seed=5, L=250, n=6000, list_gap=110),
}


@needs_ref
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_export_equals_reference_export(tmp_path, name):
    kw = dict(CASES[name])
    pair = kw.pop("paired", False)
    case = xu.export_case(paired=pair, **kw)
    n = case["reads"].shape[0]
    kmax = case["L"] // 3
    res = orc.ref_match("c", case["pg"], case["reads"], 38, kmax, 0, n_nset=case["n_n"])
    order = xu.position_order(res["pos"])
    # the order is a sort by position; with ties it is NOT the stable one (else the test would not pin anything)
    assert np.array_equal(res["pos"][order], np.sort(res["pos"][order]))
    if name == "se":
        assert not np.array_equal(order, xu.stable_order(res["pos"]))
    ref = xu.ref_export_run(case, str(tmp_path / "r"), 0, kmax=kmax, pair_file_mode=pair, rev_compl_pair_file=pair)
    got = xu.stream_bytes(xu.oracle_export_pg_order(case, res, order, pair_file=pair))
    for k in xu.STREAMS:
        assert got[k] == ref[k], (name, k)
    # original order (-o): entries listed by the numpy restatement of the walk
    ref = xu.ref_export_run(case, str(tmp_path / "o"), 0, kmax=kmax, preserve_order=True, pair_file_mode=pair,
                            rev_compl_pair_file=pair)
    er, eo = xu.original_order_entries(case["read_org"], res["mism"] != 255, case["total"], pair, n - case["n_n"])
    got = xu.stream_bytes(xu.oracle_export_entries(case, res, er, eo, pair_file=pair))
    for k in xu.STREAMS:
        assert got[k] == ref[k], (name, "original order", k)


@needs_ref
def test_position_order_equals_reference_sort_with_threads(tmp_path):
    """__gnu_parallel::sort on pairs == on indexes also when it really runs in parallel (the reference at -t 8)"""
    case = xu.export_case(seed=7, n=60_000, G=500_000, dups=3000)
    res = orc.ref_match("c", case["pg"], case["reads"], 38, 33, 0, n_nset=case["n_n"])
    order8 = xu.position_order(res["pos"], threads=8)
    ref = xu.ref_export_run(case, str(tmp_path / "t8"), 0, threads=8)
    got = xu.stream_bytes(xu.oracle_export_pg_order(case, res, order8))
    for k in xu.STREAMS:
        assert got[k] == ref[k], k
