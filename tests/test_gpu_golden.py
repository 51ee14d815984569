"""The HIP path against the committed golden vectors (outputs of the real reference)."""
import numpy as np
import pytest

from golden_util import INDEX_CASES, MANIFEST, MATCH_CASES, cumm_to_sparse, load_case, load_index_case
from util import assert_same_results, gpu_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", MATCH_CASES)
def test_gpu_reproduces_reference_output(name):
    m, pg, reads, kind, sl, kmax, kmin, gold = load_case(name)
    g = gpu_match(kind, pg, reads, sl, kmax, kmin, m["rev_compl"])
    assert_same_results(g, gold, name)


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
