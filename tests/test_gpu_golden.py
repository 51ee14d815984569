"""The HIP path against the committed golden vectors (outputs of the real reference)."""
import numpy as np
import pytest

from golden_util import HARD_CASES, INDEX_CASES, MANIFEST, MATCH_CASES, cumm_to_sparse, load_case, load_hard_case, load_index_case
from util import assert_same_results, gpu_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", MATCH_CASES)
def test_gpu_reproduces_reference_output(name):
    m, pg, reads, kind, sl, kmax, kmin, gold = load_case(name)
    g = gpu_match(kind, pg, reads, sl, kmax, kmin, m["rev_compl"])
    assert_same_results(g, gold, name)


@pytest.mark.parametrize("schedule", ["default", "screen", "two_pass", "full_loops"])
@pytest.mark.parametrize("name", HARD_CASES)
def test_gpu_reproduces_reference_output_on_hard_inputs(monkeypatch, name, schedule):
    """L = 150, seed 38 (the dual kernel is the default schedule here): repeat families, reverse palindromes, short-period
    texts, reads from both strands, against the real reference's output -- under every schedule the library has."""
    env = {"default": {}, "screen": {"PGRC_DUAL": "0"}, "two_pass": {"PGRC_DUAL": "0", "PGRC_SCREEN": "0"},
           "full_loops": {"PGRC_EARLY_STOP": "0"}}[schedule]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m, pg, reads, gold = load_hard_case(name)
    g = gpu_match("c", pg, reads, m["seed_len"], m["kmax"], 0, True, n_nset=m["n_nset"] or None)
    assert_same_results(g, gold, f"{name} [{schedule}]")
    c = g["ctx"].counters()
    assert c["screened"] == {"default": 2, "screen": 1, "two_pass": 0, "full_loops": 0}[schedule]
    if schedule == "default" and ("repeat" in name or "period" in name):
        assert c["redo_reads"] > 0          # the dual kernel met reads whose falses budget matters and redid them in order


@pytest.mark.parametrize("name", INDEX_CASES)
def test_gpu_index_reproduces_reference_index(name):
    from pgrc_amd import MatchContext
    m, pg, positions, buckets, counts = load_index_case(name)
    ctx = MatchContext(max(150, m["seed_len"]), m["seed_len"], 2, 0, "c")
    ctx.set_pg_ascii(pg)
    cumm, pos = ctx.export_index(0)
    nz, cnt = cumm_to_sparse(cumm)
    assert np.array_equal(pos, positions) and np.array_equal(nz, buckets) and np.array_equal(cnt, counts)


def test_gpu_mismatch_lists_reproduce_reference():
    import json
    import os
    import make_golden as mg
    from golden_util import GOLD
    pg, reads = mg.case_inputs(mg.CASES[8])
    with open(os.path.join(GOLD, "extract_c_nreads_M3.json")) as f:
        rows = json.load(f)
    g = gpu_match("c", pg, reads, 38, 33, 0)
    z = np.load(os.path.join(GOLD, "extract_c_nreads_M3.npz"))
    assert np.array_equal(g["pos"], z["pos"]) and np.array_equal(g["mism"], z["mism"])
    n = reads.shape[0]
    se = g["ctx"].extract_mismatches(None)
    pe = g["ctx"].extract_mismatches((g["rc"] != (np.arange(n) & 1)).astype(np.uint8))
    for i, pair_file, codes, offs in rows:
        cum, c, o = pe if pair_file else se
        s, e = int(cum[i]), int(cum[i + 1])
        assert c[s:e].tolist() == codes and o[s:e].tolist() == offs, (i, pair_file)


# ---- row f2: Pg-vs-Pg exact matching

import mem_golden_util as mg  # noqa: E402
import oracle as orc  # noqa: E402  (mem_dest: numpy only)


@pytest.mark.parametrize("name", mg.NAMES)
def test_gpu_mem_match_reproduces_reference_output(name):
    from pgrc_amd import CopMEMMatcher
    src, other, tl, ml, expected = mg.load(name)
    m = CopMEMMatcher(src, tl)
    for (dis, rc), want in expected.items():
        got = m.matchTexts(orc.mem_dest(src, other, dis, rc), dis, rc, ml)
        assert np.array_equal(got, want), (name, dis, rc)


# ---- row f1: export streams against the committed reference output

import export_util as xu  # noqa: E402


@pytest.mark.parametrize("name", xu.EXPORT_GOLDEN)
def test_gpu_export_reproduces_reference_streams(name):
    case, pair, kmax, res, order, gold = xu.load_export_golden(name)
    g = gpu_match("c", case["pg"], case["reads"], 38, kmax, 0, n_nset=case["n_n"])
    for k in ("pos", "rc", "mism"):
        assert np.array_equal(g[k], res[k]), k
    got = xu.stream_bytes(g["ctx"].export_pg_order(order, case["list_off"], case["list_org"], case["list_rc"], case["read_org"], pair, True))
    for k in xu.STREAMS:
        assert got[k] == gold["pg"][k], (name, k)
    n = case["reads"].shape[0]
    er, eo = xu.original_order_entries(case["read_org"], res["mism"] != 255, case["total"], pair, n - case["n_n"])
    got = xu.stream_bytes(g["ctx"].export_entries(er, eo, pair, True))
    dev = xu.stream_bytes(g["ctx"].export_original_order(case["read_org"], case["total"], pair, pair, True))
    for k in xu.STREAMS:
        assert got[k] == gold["org"][k], (name, "original order", k)
        assert dev[k] == gold["org"][k], (name, "original order, entry list made on the device", k)
