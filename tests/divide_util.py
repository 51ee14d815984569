"""Row f3 helpers (test infrastructure): FASTQ-like records as row arrays, and the read-set division through the oracle
restatement, the compiled reference (oracle/_ref, with or without integration/HipDividedReadsSets) and the HIP library."""
import ctypes as C

import numpy as np

import oracle as orc

_P = C.c_void_p
# (error_limit, simplified_suffix_mode, separateNReadsSet, nReadsLQ): what pgrc-encoder.cpp:254-282 can ask for
COMBOS = [(1.0, True, False, False), (1.0, True, True, False), (1.0, True, False, True), (1.0, True, True, True),
          (0.05, True, True, False), (0.05, False, True, False), (0.2, False, False, True), (0.5, True, False, False),
          (0.01, False, True, True), (0.12, False, False, False), (0.3, True, True, True)]


def make_records(seed, n, L, n_frac=0.04, low_quality_frac=0.3):
    """n reads of length L over ACGT with an N here and there, and quality rows: Phred 2..41 ('#'..'J'), a share of the
    reads with a bad tail, a share noisy all over -- so that both quality tests cut the set somewhere in the middle"""
    rng = np.random.default_rng(seed)
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))].copy()
    with_n = rng.random(n) < n_frac
    for i in np.flatnonzero(with_n):
        reads[i, rng.integers(0, L, size=int(rng.integers(1, 4)))] = ord("N")
    q = rng.integers(30, 42, size=(n, L))
    bad = rng.random(n) < low_quality_frac
    for i in np.flatnonzero(bad):
        if rng.random() < 0.5 and L > 1:
            k = int(rng.integers(1, L))
            q[i, k:] = rng.integers(2, 12, size=L - k)           # a bad tail
        else:
            q[i] = rng.integers(2, 42, size=L)                   # noisy all over
    quals = (q + 33).astype(np.uint8)
    return reads, quals


def _alloc(n, L):
    rb = (L + 2) // 3
    return (np.zeros(n * rb + 1, np.uint8), np.zeros(n * rb + 1, np.uint8), np.zeros(n * rb + 1, np.uint8),
            np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32))


def _result(bufs, counts, symbols, L):
    hq, lq, nn, li, ni = bufs
    rb = [((L + (4 if s == 4 else 3) - 1) // (4 if s == 4 else 3)) if s else 0 for s in symbols]
    return {"n_hq": int(counts[0]), "n_lq": int(counts[1]), "n_n": int(counts[2]), "symbols": tuple(int(s) for s in symbols),
            "row_bytes": tuple(rb), "hq_rows": hq[: counts[0] * rb[0]].copy(), "lq_rows": lq[: counts[1] * rb[1]].copy(),
            "n_rows": nn[: counts[2] * rb[2]].copy(), "lq_index": li[: counts[1]].copy(), "n_index": ni[: counts[2]].copy()}


def oracle_divide(reads, quals, error_limit, simplified, separate_n, n_reads_lq):
    n, L = reads.shape
    lib = orc.oracle()
    f = lib.pgrc_or_divide_reads
    f.argtypes = [_P, _P, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                  C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_uint32 * 3)]
    bufs = _alloc(n, L)
    counts, symbols = (C.c_uint64 * 3)(), (C.c_uint32 * 3)()
    r = np.ascontiguousarray(reads)
    q = np.ascontiguousarray(quals) if quals is not None else None
    e = f(r.ctypes.data, q.ctypes.data if q is not None else None, n, L, error_limit, int(simplified), int(separate_n), int(n_reads_lq),
          *[b.ctypes.data for b in bufs], C.byref(counts), C.byref(symbols))
    assert e == 0
    return _result(bufs, list(counts), list(symbols), L)


def ref_divide(reads, quals, error_limit, simplified, separate_n, n_reads_lq, use_adapter=False):
    n, L = reads.shape
    f = orc.ref().pgrc_ref_divide
    f.argtypes = [C.c_int, _P, _P, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                  C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_uint32 * 3)]
    bufs = _alloc(n, L)
    counts, symbols = (C.c_uint64 * 3)(), (C.c_uint32 * 3)()
    r = np.ascontiguousarray(reads)
    q = np.ascontiguousarray(quals) if quals is not None else None
    e = f(int(use_adapter), r.ctypes.data, q.ctypes.data if q is not None else None, n, L, error_limit, int(simplified), int(separate_n),
          int(n_reads_lq), *[b.ctypes.data for b in bufs], C.byref(counts), C.byref(symbols))
    assert e == 0, e
    return _result(bufs, list(counts), list(symbols), L)


def same(a, b):
    """first differing field of two division results, or None"""
    for k in ("n_hq", "n_lq", "n_n", "symbols", "row_bytes"):
        if a[k] != b[k]:
            return k
    for k in ("hq_rows", "lq_rows", "n_rows", "lq_index", "n_index"):
        if not np.array_equal(a[k], b[k]):
            return k
    return None


# ---- FASTQ text (the reference's FASTQReadsSourceIterator reads it line by line)

def make_fastq(reads, quals, seed=0, crlf=False, trailing_newline=True):
    """records with identifier lines of varying length (so that nothing lines up), optionally CR LF line ends"""
    rng = np.random.default_rng(seed)
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(reads.shape[0]):
        name = b"@r%d" % i + b"x" * int(rng.integers(0, 40)) + (b" len=%d" % reads.shape[1] if i % 3 == 0 else b"")
        parts += [name, nl, reads[i].tobytes(), nl, b"+" + (name[1:] if i % 5 == 0 else b""), nl, quals[i].tobytes(), nl]
    text = b"".join(parts)
    return text if trailing_newline else text[: -len(nl)]


def oracle_fastq_rows(text, pair_text, rev_compl_pair, L):
    f = orc.oracle().pgrc_or_fastq_records
    f.restype = C.c_int64
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, C.c_uint32, _P, _P, C.c_uint64]
    cap = (len(text) + (len(pair_text) if pair_text else 0)) // (2 * L + 4) + 4
    reads, quals = np.zeros((cap, L), np.uint8), np.zeros((cap, L), np.uint8)
    n = f(text, len(text), pair_text, len(pair_text) if pair_text is not None else 0, int(rev_compl_pair), L, reads.ctypes.data, quals.ctypes.data, cap)
    if n < 0:
        return None, None
    return reads[:n].copy(), quals[:n].copy()


def oracle_divide_fastq(text, pair_text, rev_compl_pair, L, combo):
    reads, quals = oracle_fastq_rows(text, pair_text, rev_compl_pair, L)
    if reads is None:
        return None
    return oracle_divide(reads, quals, *combo)


def ref_divide_files(src, pair, rev_compl_pair, L, n_max, combo, use_adapter=False):
    f = orc.ref().pgrc_ref_divide_files
    f.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                  C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_uint32 * 3)]
    bufs = _alloc(n_max, L)
    counts, symbols = (C.c_uint64 * 3)(), (C.c_uint32 * 3)()
    error_limit, simplified, separate_n, n_reads_lq = combo
    e = f(int(use_adapter), str(src).encode(), str(pair).encode() if pair else b"", int(rev_compl_pair), L, error_limit, int(simplified),
          int(separate_n), int(n_reads_lq), *[b.ctypes.data for b in bufs], C.byref(counts), C.byref(symbols))
    assert e == 0, e
    return _result(bufs, list(counts), list(symbols), L)
