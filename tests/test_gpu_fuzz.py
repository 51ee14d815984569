"""Differential sweep over the parameter space (read lengths around every word / byte boundary, every copMEM K,
all modes, shortcut on/off, both strands or forward only, N reads) -- HIP path vs oracle, small inputs."""
import numpy as np
import pytest

import oracle as orc
from util import assert_same_results, gpu_match, make_inputs

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20260101)
    out = []
    # copMEM: every K the reference can pick (20,24,28,32,36,40,44,56) via the seed length, read lengths on and around
    # 8- and 16-symbol boundaries (head/tail split, last-word masks), limits from 0 up
    seeds = [24, 27, 28, 31, 32, 33, 38, 43, 45, 47, 50, 54, 60, 63, 64, 90, 111, 130]
    for k, seed_len in enumerate(seeds):
        for L in sorted({seed_len, seed_len + int(rng.integers(1, 40)), 16 * ((seed_len + 31) // 16), 8 * ((seed_len + 23) // 8) + 1}):
            if L > 255:
                continue
            M = int(rng.choice([1000, 50, 25, 10, 3]))
            out.append(("c", L, seed_len, M, bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0), int(rng.integers(0, 2)) * 150))
    for L in (255, 254, 249, 241, 240, 239):
        out.append(("c", L, 38, 40, False, True, 0))
    # read-side index modes
    for mode in ("d", "i"):
        for L, seed_len in ((60, 20), (64, 32), (65, 32), (97, 33), (100, 50), (128, 40), (150, 50), (200, 66), (255, 38)):
            M = int(rng.choice([1000, 50, 20, 5]))
            out.append((mode, L, seed_len, M, bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0), int(rng.integers(0, 2)) * 120))
    for L in (24, 33, 64, 100, 177, 255):
        out.append(("e", L, L, 1000, False, bool(rng.integers(0, 2)), 0))
    return out


@pytest.mark.parametrize("mode,L,seed_len,M,shortcut,rev,n_with_n", _cases())
def test_sweep(mode, L, seed_len, M, shortcut, rev, n_with_n):
    G = 60000 + 37 * L
    n = 2500
    pg, reads = make_inputs(G, n, L, seed=L * 1000 + seed_len * 7 + M, n_with_n=n_with_n)
    kmax = min(L // M, 247)
    kmin = kmax if shortcut else 0
    o = orc.oracle_match(mode, pg, reads, seed_len, kmax, kmin, rev)
    g = gpu_match(mode, pg, reads, seed_len, kmax, kmin, rev)
    assert_same_results(g, o, f"{mode} L={L} seed={seed_len} M={M} shortcut={shortcut} rev={rev} N={n_with_n}")
    if mode != "e":
        assert g["matched"] > 0.5 * n * (0.5 if not rev else 1.0) * 0.5


# ---- row f2: Pg-vs-Pg exact matching -- every K the reference can pick (through the target match length), texts with
# periodic / low-complexity / duplicated content, N runs, ragged lengths around the 256-window block size

from mem_util import mem_sweep_cases as _mem_cases, mem_sweep_texts as _mem_texts  # noqa: E402


@pytest.mark.parametrize("target,min_len,seed", _mem_cases())
def test_mem_sweep(target, min_len, seed):
    from pgrc_amd import CopMEMMatcher
    src, other = _mem_texts(seed, target)
    m = CopMEMMatcher(src, target)
    for dest_is_src, rev_compl in ((0, 1), (1, 1), (0, 0), (1, 0)):
        d = orc.mem_dest(src, other, dest_is_src, rev_compl)
        g = m.matchTexts(d, dest_is_src, rev_compl, min_len)
        o = orc.oracle_mem_match(src, d, dest_is_src, rev_compl, target, min_len)
        assert np.array_equal(g, o), f"target={target} min={min_len} seed={seed} destIsSrc={dest_is_src} rc={rev_compl}: {len(g)} vs {len(o)}"
