"""Shared helpers for the tests: synthetic inputs and numpy re-statements of the packed layouts."""
from __future__ import annotations

import numpy as np

from pgrc_amd import synth

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_inputs(G, n, L, seed, n_with_n=0, tandem_every=2, paired=False, pool_div=8):
    g = synth.pg_params(G, seed=seed, tandem_every=tandem_every, pool_div=pool_div)
    pg = synth.pg_host(g)
    rs = synth.reads_params(n, L, seed=seed, n_with_n=n_with_n, paired=paired)
    reads = synth.reads_host(g, pg, rs)
    return pg, reads


def pack2(ascii_1d: np.ndarray) -> np.ndarray:
    """2-bit packing used on the device: symbol i at bits 2*(i%16) of u32 word i/16."""
    codes = _CODE[ascii_1d].astype(np.uint64)
    assert (codes < 4).all()
    n = codes.size
    pad = (-n) % 16
    codes = np.concatenate([codes, np.zeros(pad, dtype=np.uint64)]).reshape(-1, 16)
    sh = (2 * np.arange(16, dtype=np.uint64))[None, :]
    return (codes << sh).sum(axis=1).astype(np.uint32)


def revcomp(ascii_1d: np.ndarray) -> np.ndarray:
    return _COMP[ascii_1d[::-1]]


def assert_same_results(a, b, what=""):
    for k in ("pos", "rc", "mism", "hist"):
        ak, bk = np.asarray(a[k]), np.asarray(b[k])
        assert np.array_equal(ak, bk), f"{what}: {k} differs at {np.flatnonzero(ak != bk)[:10]}"
    assert a["matched"] == b["matched"], what


def pack_rows(reads, alphabet=b"ACGT"):
    """The reference's packed rows (SymbolsPackingFacility layout) of ASCII reads, made by the oracle's restatement
    of packSequence (checked against the real one in tests/test_oracle_vs_ref.py)."""
    import ctypes as C
    import oracle as orc
    n, L = reads.shape
    spe = 4 if len(alphabet) == 4 else 3
    buf = np.zeros((n, (L + spe - 1) // spe), dtype=np.uint8)
    f = orc.oracle().pgrc_or_pack_read
    for i in range(n):
        f(reads[i].ctypes.data_as(C.c_void_p), L, alphabet, buf[i].ctypes.data_as(C.c_void_p))
    return buf


def gpu_match(mode, pg, reads, seed_len, kmax, kmin, rev_compl=True, packed_ref=False, devices=None, n_nset=None):
    """Runs the HIP path through the C ABI; returns the same dict shape as the oracle helpers.
    packed_ref: hand the reads over in the reference's ACGT packing; n_nset: ... as the LQ + N sum set (the last
    n_nset reads ACGNT-packed); devices: one matcher over these devices (pgrc_match_create_multi)."""
    from pgrc_amd import MatchContext
    ctx = MatchContext(reads.shape[1], seed_len, kmax, kmin, mode, devices=devices)
    ctx.set_pg_ascii(pg)
    if n_nset is not None:
        n_lq = reads.shape[0] - n_nset
        ctx.set_reads_packed_sets([(pack_rows(reads[:n_lq]), n_lq, 4), (pack_rows(reads[n_lq:], b"ACGNT"), n_nset, 5)])
    elif packed_ref:
        ctx.set_reads_packed(pack_rows(reads), reads.shape[0])
    else:
        ctx.set_reads_ascii(reads)
    ctx.init_results()
    ctx.run(rev_compl)
    pos, rc, mism, hist, matched = ctx.get_results()
    return {"pos": pos, "rc": rc, "mism": mism, "hist": hist, "matched": matched, "ctx": ctx}
